"""link_amd -- MI355X-native (gfx950, hand-written HIP) implementation of LinK's linear-kernel
sparse 3-D aggregation hot path, behind the Python surface the reference's backbones call.

Reference: MCG-NJU/LinK (segmentation/core/models/utils.py:44-84,
segmentation/core/models/semantic_kitti/linkunet.py:94-185, detection/det3d/models/utils/ts_elk.py).
The HIP library (link_amd/lib/liblink_amd.so, ABI in include/link_amd.h) is mandatory: there is no
CPU or eager-PyTorch fallback.
"""
from . import _lib, backend, functional
from .aggregate import (aux_to_voxel, large_to_small, link_index_of, small_to_large_v2, upsample_voxel,
                        voxel_to_aux)
from .elk import (Conv3d, ELKBlock, ElkCoreBatch, ElkCorePlan, SparseConvTensor, TSELKBlock, elk_core_autograd, elk_core_fused,
                  invalidate_derived_weights, spconv2ts, ts2spconv)
from .detstage import ELKv3Stage, SparseBasicBlock, SparseConv3d, SpMiddleResNetFHDELKv3, SubMConv3d, to_dense
from .functional import calc_ti_weights, spcount, spdevoxelize, sphash, sphashquery, spvoxelize
from .index import BlockIndex, coords_bounds
from .modules import BatchNorm, LeakyReLU, ReLU, fapply, fuse_for_inference
from .pointvoxel import initial_voxelize, point_to_voxel, voxel_to_point
from .tensor import PointTensor, SparseTensor, cat
from .utils import get_kernel_offsets, make_ntuple

__version__ = "0.1.0"


def install_as_torchsparse() -> None:
    """Register link_amd under the module names the reference imports (`torchsparse`,
    `torchsparse.nn`, `torchsparse.nn.functional`, `torchsparse.nn.utils`, `torchsparse.utils`,
    `torchsparse.backend`) so that unmodified reference code such as
    `import torchsparse.nn.functional as F; from torchsparse import SparseTensor` resolves here."""
    import sys
    import types
    ts = types.ModuleType("torchsparse")
    ts.SparseTensor, ts.PointTensor, ts.cat = SparseTensor, PointTensor, cat
    nn_mod = types.ModuleType("torchsparse.nn")
    nn_mod.Conv3d, nn_mod.BatchNorm, nn_mod.ReLU, nn_mod.LeakyReLU = Conv3d, BatchNorm, ReLU, LeakyReLU
    nn_mod.functional = functional
    nn_utils = types.ModuleType("torchsparse.nn.utils")
    nn_utils.get_kernel_offsets = get_kernel_offsets
    nn_utils.fapply = fapply
    nn_mod.utils = nn_utils
    utils_mod = types.ModuleType("torchsparse.utils")
    utils_mod.make_ntuple = make_ntuple
    ts.nn, ts.utils, ts.backend = nn_mod, utils_mod, backend
    ts.__version__ = "1.4.0+link_amd"
    sys.modules.update({"torchsparse": ts, "torchsparse.nn": nn_mod, "torchsparse.nn.functional": functional,
                        "torchsparse.nn.utils": nn_utils, "torchsparse.utils": utils_mod,
                        "torchsparse.backend": backend})
