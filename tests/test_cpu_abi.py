"""CPU-only checks (run with -m "not gpu"): the C-ABI library loads and exports every symbol
include/link_amd.h declares, host-side logic, the data model, and loud failure on CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, golden_files, load_golden


def test_library_exports_every_declared_symbol():
    from link_amd import _lib as L
    from link_amd import build as hip_build
    hip_build.build()
    hdr = open(os.path.join(ROOT, "include", "link_amd.h")).read()
    declared = set(re.findall(r"\b(link_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"link_grid_t", "link_elk_desc_t", "link_elk_buffers_t"}
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    handle = ctypes.CDLL(L.SO_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert L.lib().link_abi_version() == L.ABI_VERSION
    # struct layouts agree with the header (sizes in bytes)
    assert ctypes.sizeof(L.LinkGrid) == 36 and ctypes.sizeof(L.LinkElkDesc) == 28
    assert ctypes.sizeof(L.LinkElkBuffers) == 29 * 8
    for which, cls in enumerate((L.LinkGrid, L.LinkElkDesc, L.LinkElkBuffers, L.LinkDcGrid, L.LinkDcTuning, L.LinkDcBuffers)):
        assert handle.link_abi_struct_size(which) == ctypes.sizeof(cls), cls.__name__
    assert handle.link_abi_struct_size(99) == -1
    # no process-global tuning state in the library: the setters of rounds 1-2 are gone from the ABI
    for gone in ("link_set_tuning", "link_conv_set_tuning", "link_dc_set_tuning", "link_dc_set_tuning2", "link_dc_set_debug_buffer"):
        assert not hasattr(handle, gone), gone


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from link_amd import _lib as L
    lib = L.lib()
    assert lib.link_hash(None, -1, None, None) == L.LINK_ERR_ARG
    assert lib.link_hash(None, 5, None, None) == L.LINK_ERR_ARG
    assert lib.link_hash(None, 0, None, None) == L.LINK_OK          # empty input is a no-op
    assert lib.link_kernel_hash(None, 0, None, 27, None, None) == L.LINK_OK
    assert lib.link_premix_ln(None, None, None, None, 10, 0, 1e-6, None, None) == L.LINK_ERR_ARG
    assert lib.link_premix_ln(None, None, None, None, 10, 512, 1e-6, None, None) == L.LINK_ERR_ARG
    assert lib.link_hash_query_workspace_bytes(1000) >= 2 * 1000 * 12
    assert lib.link_index_scratch_bytes(1000, 50000) >= 4 * 4000 + 4 * 50000
    # round-2 entries: the pair-list convolution, its map builders, the dense-cell step
    assert lib.link_conv_pairs_supported(64, 64) == 1 and lib.link_conv_pairs_supported(64, 128) == 1
    assert lib.link_conv_pairs_supported(48, 48) == 0 and lib.link_conv_pairs_supported(64, 256) == 0
    assert lib.link_conv_pairs_gemm(None, None, None, 0, None, 64, 64, None, None) == L.LINK_OK          # no pairs
    assert lib.link_conv_pairs_gemm(None, None, None, 100, None, 64, 64, None, None) == L.LINK_ERR_ARG   # not 128-row granules
    assert lib.link_conv_pairs_gemm(None, None, None, 128, None, 48, 48, None, None) == L.LINK_ERR_ARG   # width
    assert lib.link_conv_pairs_gemm(None, None, None, 128, None, 64, 64, None, None) == L.LINK_ERR_ARG   # null buffers
    assert lib.link_conv_pairs_sum(None, None, None, 0, 0, 64, None, None, None, 0.0, None, 0, None, None) == L.LINK_OK
    assert lib.link_conv_pairs_sum(None, None, None, 10, 11, 64, None, None, None, 0.0, None, 0, None, None) == L.LINK_ERR_ARG
    assert lib.link_conv_centre_sum(None, None, -1, None, 0, None, None, 10, 64, 64, None, None, None, 0.0, None, 0, None,
                                    None) == L.LINK_ERR_ARG
    assert lib.link_conv_pairs_gemm_io(None, 3, None, None, 0, None, 64, 64, None, None) == L.LINK_ERR_ARG    # row type
    assert lib.link_conv_pairs_gemm_io(None, L.IO_BF16, None, None, 0, None, 64, 64, None, None) == L.LINK_OK
    assert lib.link_conv_pairs_sum_io(None, None, None, 0, 0, 64, None, None, None, 0.0, None, 0, None, -1, None) == L.LINK_ERR_ARG
    assert lib.link_conv_pairs_sum_io(None, None, None, 0, 0, 64, None, None, None, 0.0, None, 0, None, L.IO_F16, None) == L.LINK_OK
    assert lib.link_conv_centre_sum_io(None, None, 13, None, 0, None, None, 0, 64, 64, None, None, None, 0.0, None, 0, None,
                                       7, None) == L.LINK_ERR_ARG
    assert lib.link_conv_centre_sum_io(None, None, 13, None, 0, None, None, 0, 64, 64, None, None, None, 0.0, None, 0, None,
                                       L.IO_F16, None) == L.LINK_OK
    assert lib.link_conv_pairs_gemm_amp(None, L.IO_F32, None, None, 0, None, 64, 64, None, L.IO_F32, None) == L.LINK_ERR_ARG   # 16-bit rows only
    assert lib.link_conv_pairs_gemm_amp(None, L.IO_F16, None, None, 0, None, 64, 64, None, L.IO_F16, None) == L.LINK_OK
    assert lib.link_conv_pairs_gemm_amp(None, L.IO_F16, None, None, 0, None, 64, 64, None, L.IO_BF16, None) == L.LINK_ERR_ARG  # contrib: fp32 or the row type
    assert lib.link_conv_pairs_gemm_amp(None, L.IO_F16, None, None, 128, None, 64, 64, None, L.IO_F32, None) == L.LINK_ERR_ARG  # null buffers
    assert lib.link_conv_centre_sum_amp(None, None, 13, None, L.IO_F32, 0, None, None, 0, 64, 64, None, None, None, 0.0, None, 0, None,
                                        L.IO_F32, None) == L.LINK_ERR_ARG
    assert lib.link_conv_centre_sum_amp(None, None, 13, None, L.IO_F16, 0, None, None, 0, 64, 64, None, None, None, 0.0, None, 0, None,
                                        L.IO_BF16, None) == L.LINK_ERR_ARG
    assert lib.link_conv_centre_sum_amp(None, None, 13, None, L.IO_BF16, 0, None, None, 0, 64, 64, None, None, None, 0.0, None, 0, None,
                                        L.IO_BF16, None) == L.LINK_OK
    assert lib.link_bn_partial_workgroups(100000, 64) >= 1 and lib.link_bn_partial_workgroups(100000, 6) == 0
    assert lib.link_bn_forward_stats(None, 10, 6, 1e-3, 0.1, None, None, None, None, None, None, None, None, None, None) == L.LINK_ERR_ARG   # C % 4
    assert lib.link_bn_forward_stats(None, 10, 64, 1e-3, 0.1, None, None, None, None, None, None, None, None, None, None) == L.LINK_ERR_ARG  # null buffers
    assert lib.link_bn_backward_reduce(None, None, None, None, 0, 64, None, None, None, None, None, None) == L.LINK_ERR_ARG                 # n < 1
    assert lib.link_pair_plan_count(None, 10, 65, None, None, None) == L.LINK_ERR_ARG                     # kvol > 64
    assert lib.link_pair_plan_count(None, 0, 27, None, None, None) == L.LINK_OK
    i3 = ctypes.c_int32 * 3
    assert lib.link_conv_out_candidate_count(i3(3, 3, 3), i3(2, 2, 2)) == 8
    assert lib.link_conv_out_candidate_count(i3(3, 1, 1), i3(2, 1, 1)) == 2
    assert lib.link_conv_out_candidate_count(i3(3, 3, 3), i3(1, 1, 1)) == 27
    assert lib.link_conv_out_candidates(None, 5, i3(2, 3, 3), i3(2, 2, 2), i3(1, 1, 1), i3(4, 4, 4), None, None) == L.LINK_ERR_ARG
    assert lib.link_dc_index_ids(None, 5, None, None, None, None, None, None) == L.LINK_ERR_ARG
    assert lib.link_dc_index_ids(None, 0, ctypes.byref(L.LinkDcGrid()), None, None, None, None, None) == L.LINK_OK
    assert lib.link_ti_weights(None, None, 10, 0.0, None, None) == L.LINK_ERR_ARG and lib.link_ti_weights(None, None, 0, 1.0, None, None) == L.LINK_OK
    assert lib.link_ti_weights(None, None, 10, 1.0, None, None) == L.LINK_ERR_ARG
    # end of round 3: one-call pair plans, the dense-layout probe, half rows through the tile form, aux_to_voxel with the scatter
    assert lib.link_pair_plan_build(None, 0, 27, 1, 10, *([None] * 13)) == L.LINK_ERR_ARG                  # n < 1
    assert lib.link_pair_plan_build(None, 10, 65, 1, 10, *([None] * 13)) == L.LINK_ERR_ARG                 # kvol > 64
    assert lib.link_pair_plan_build(None, 10, 27, 1, 10, *([None] * 13)) == L.LINK_ERR_ARG                 # null buffers
    assert lib.link_dc_index_probe(None, None, 5, None, None) == L.LINK_ERR_ARG
    assert lib.link_dc_index_probe(ctypes.byref(L.LinkDcBuffers()), ctypes.byref(L.LinkDcGrid()), 5, None, None) == L.LINK_ERR_ARG  # no stats
    desc = L.LinkElkDesc(L.OP_COS, 64, 32, 3, 1.0, 1e-6)
    assert lib.link_elk_premix_modsum_tiles_io(None, 3, *([None] * 9), ctypes.byref(desc), 10, 10, None, 0, None, None) == L.LINK_ERR_ARG   # row type
    assert lib.link_elk_premix_modsum_tiles_io(None, L.IO_F16, *([None] * 9), ctypes.byref(desc), 0, 10, None, 0, None, None) == L.LINK_OK   # empty frame
    assert lib.link_elk_premix_modsum_tiles_io(None, L.IO_F16, *([None] * 9), ctypes.byref(desc), 10, 10, None, 0, None, None) == L.LINK_ERR_ARG  # null buffers
    g = L.LinkGrid()
    assert lib.link_elk_gather_demod_tiles_io(*([None] * 6), ctypes.byref(g), *([None] * 5), ctypes.byref(desc), 10, 10, None, -1, None) == L.LINK_ERR_ARG
    assert lib.link_elk_gather_demod_tiles_io(*([None] * 6), ctypes.byref(g), *([None] * 5), ctypes.byref(desc), 0, 10, None, L.IO_BF16, None) == L.LINK_OK
    assert lib.link_aux_to_voxel_forward_scatter(*([None] * 4), ctypes.byref(g), None, None, None, 10, 10, 10, 3, None, None, None, None) == L.LINK_ERR_ARG   # width 10: neither 2 nor 3 parts
    assert lib.link_aux_to_voxel_forward_scatter(*([None] * 4), ctypes.byref(g), None, None, None, 10, 0, 128, 3, None, None, None, None) == L.LINK_OK       # no blocks
    assert lib.link_aux_to_voxel_forward_scatter(*([None] * 4), ctypes.byref(g), None, None, None, 10, 10, 128, 3, None, None, None, None) == L.LINK_ERR_ARG  # null buffers


def test_grid_from_bounds_host_logic():
    from link_amd import _lib as L
    g = L.grid_from_bounds((0, 0, 0, 0), (255, 255, 255, 0), 7)
    assert g.key() == (7, (0, 0, 0, 0), (37, 37, 37, 1)) and g.cells == 50653
    g = L.grid_from_bounds((-15, -1, 0, 1), (14, 20, 6, 3), 7)       # floor division for negatives
    assert tuple(g.lo) == (-3, -1, 0, 1) and tuple(g.dim) == (6, 4, 1, 3)
    with pytest.raises(L.LinkAmdError):
        L.grid_from_bounds((0, 0, 0, 0), (2 ** 31 - 1, 2 ** 31 - 1, 0, 0), 1)   # >= 2^30 cells
    with pytest.raises(L.LinkAmdError):
        L.grid_from_bounds((5, 0, 0, 0), (4, 0, 0, 0), 1)


def test_kernel_offsets_match_reference():
    import link_amd as la
    k = load_golden("g_koff.npz")
    for r in (2, 3, 4, 5):
        out = la.get_kernel_offsets(r)
        assert out.dtype == torch.int32 and np.array_equal(out.numpy(), k[f"r{r}"])
    assert la.get_kernel_offsets(3, stride=2)[0].tolist() == [-2, -2, -2]
    assert la.make_ntuple(3, 3) == (3, 3, 3) and la.make_ntuple([1, 2, 3], 3) == (1, 2, 3)
    assert la.make_ntuple(torch.tensor([4, 5, 6]), 3) == (4, 5, 6)


def test_sparse_tensor_surface():
    import link_amd as la
    f, c = torch.randn(5, 3), torch.zeros(5, 4, dtype=torch.int32)
    st = la.SparseTensor(f, c, 2)
    assert st.F is f and st.C is c and st.s == (2, 2, 2) and st.stride == (2, 2, 2)
    st.F = f * 2
    assert torch.equal(st.feats, f * 2)
    st.s = 4
    assert st.stride == (4, 4, 4)
    other = la.SparseTensor(f, c, 4)
    tot = st + other
    assert tot.cmaps is st.cmaps and tot.kmaps is st.kmaps and torch.equal(tot.F, f * 3)
    cat = la.cat([st, other])
    assert cat.F.shape == (5, 6) and cat.kmaps is st.kmaps
    pt = la.PointTensor(f, c.float())
    assert set(pt.additional_features) == {"idx_query", "counts"}


@pytest.mark.parametrize("name", golden_files("g_block_*_s3_r2.npz"))
def test_state_dict_compatible_with_reference(name):
    """Reference checkpoints must load: same keys, same shapes (checkpoint row)."""
    import link_amd as la
    g = load_golden(name)
    m = g["meta"]
    blk = la.ELKBlock(m["C"], m["C"], groups=m["groups"], baseop=m["baseop"], variant=m["variant"])
    ref = {k[4:]: tuple(v.shape) for k, v in g.items() if k.startswith("sd__")}
    mine = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    assert mine == ref
    det = la.TSELKBlock(16, 16).state_dict()
    assert tuple(det["pos_weight.0.weight"].shape) == (16, 3) and "local_mix.0.kernel" in det


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors, never fall back."""
    import link_amd as la
    from link_amd._lib import LinkAmdError
    c = torch.zeros((4, 4), dtype=torch.int32)
    with pytest.raises(LinkAmdError):
        la.sphash(c)
    with pytest.raises(LinkAmdError):
        la.BlockIndex(c, 3)
    with pytest.raises(LinkAmdError):
        la.spvoxelize(torch.zeros(4, 2), torch.zeros(4, dtype=torch.int32), torch.ones(1, dtype=torch.int32))
    with pytest.raises(Exception):
        la.voxel_to_aux(la.SparseTensor(torch.zeros(4, 2), c, 1), 3)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "link_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "link_oracle" not in src, fn


def test_install_as_torchsparse():
    import sys
    import link_amd as la
    la.install_as_torchsparse()
    import torchsparse
    import torchsparse.nn.functional as F
    from torchsparse.nn.utils import get_kernel_offsets
    from torchsparse.utils import make_ntuple
    assert torchsparse.SparseTensor is la.SparseTensor and F.sphash is la.sphash
    assert get_kernel_offsets is la.get_kernel_offsets and make_ntuple is la.make_ntuple
    import torchsparse.backend as B
    for name in ("hash_cuda", "kernel_hash_cuda", "hash_query_cuda", "count_cuda", "voxelize_forward_cuda",
                 "voxelize_backward_cuda", "devoxelize_forward_cuda", "devoxelize_backward_cuda",
                 "convolution_forward_cuda", "convolution_backward_cuda"):
        assert callable(getattr(B, name))
    assert callable(F.conv3d) and callable(F.spdownsample)      # nn/functional/conv.py:83, downsample.py:11
    for k in [k for k in sys.modules if k == "torchsparse" or k.startswith("torchsparse.")]:
        del sys.modules[k]


@pytest.mark.skipif(not os.path.isdir("/root/reference/segmentation/core/models"),
                    reason="needs the reference checkout (build container only)")
def test_reference_network_classes_build_on_the_aliased_surface():
    """Level-0 integration: with link_amd registered under the torchsparse module names, the
    reference's OWN network classes (segmentation/core/models/semantic_kitti/link{encoder,unet}.py, imported
    from where they lie, unmodified) construct on top of link_amd.Conv3d / BatchNorm / ReLU and their
    ELKBlock parameters carry the reference's state_dict names.  Construction only: running them needs a GPU,
    and the reference checkout does not travel to the GPU box."""
    import importlib
    import sys
    import link_amd as la
    la.install_as_torchsparse()
    sys.path.insert(0, "/root/reference/segmentation")
    try:
        enc = importlib.import_module("core.models.semantic_kitti.linkencoder")
        net = enc.ELKEncoder(num_classes=19, cr=1.0, baseop="cos_x", groups=1, s=3, r=2)
        assert isinstance(net.stem[0], la.Conv3d) and isinstance(net.down1[0].net[1], la.BatchNorm)
        keys = set(net.state_dict().keys())
        for k in ("elk1.alpha", "elk1.pos_weight.0.weight", "elk1.pre_mix.0.weight", "elk1.pre_mix.1.weight",
                  "elk1.local_mix.0.kernel", "elk1.norm_local.weight", "elk1.norm.bias", "stem.0.kernel",
                  "down1.0.net.0.kernel"):
            assert k in keys, k
        # our ELKBlock exposes exactly the parameter names/shapes of the reference's
        ref_blk = {k[len("elk1."):]: v.shape for k, v in net.state_dict().items() if k.startswith("elk1.")}
        ours = {k: v.shape for k, v in la.ELKBlock(64, 64, 1, baseop="cos_x", variant="encoder").state_dict().items()}
        assert ref_blk == ours
        unet = importlib.import_module("core.models.semantic_kitti.linkunet")
        net2 = unet.ELKUNet(num_classes=19, cr=0.5, baseop="cos", groups=2, s=7, r=3)
        assert any(isinstance(m, la.Conv3d) and m.transposed for m in net2.modules())      # the up-sampling path
    finally:
        sys.path.remove("/root/reference/segmentation")
        for k in [k for k in sys.modules if k == "torchsparse" or k.startswith("torchsparse.") or k == "core"
                  or k.startswith("core.")]:
            del sys.modules[k]


def test_bev_half_harness_shapes():
    """The plain-torch stand-in of the detection model's dense half (RPN + CenterHead; outside the hot path, used by
    `bench.py --workload cfg5 --bev`): six tasks, the config's head widths, predictions at the BEV map's resolution."""
    import torch
    from harness.bevhead import BevCenterHead, BevHalf
    m = BevHalf().eval()
    with torch.no_grad():
        out = m(torch.randn(1, 256, 20, 20))
    assert len(out) == 6
    for task, ncls in zip(out, BevCenterHead.NUSC_TASKS):
        assert {k: v.shape[1] for k, v in task.items()} == {"reg": 2, "height": 1, "dim": 3, "rot": 2, "vel": 2, "hm": ncls}
        assert all(v.shape[-2:] == (20, 20) for v in task.values())
    assert m.neck.out_channels == 512


def test_lean_form_entry_rejects_bad_arguments_without_gpu():
    """link_elk_core_lean_forward (round 4, ABI 9) validates before touching the device: widths, r, slot capacity, grid size,
    list capacity, missing buffers; the struct layout is the header's."""
    import ctypes
    from link_amd import _lib as L
    lib = L.lib()
    assert L.ABI_VERSION >= 9 and ctypes.sizeof(L.LinkLeanBuffers) == 21 * 8 + 8 + 4 * 4
    assert lib.link_abi_struct_size(6) == ctypes.sizeof(L.LinkLeanBuffers)
    grid = L.grid_from_bounds((0, 0, 0, 0), (63, 63, 63, 0), 7)
    desc = L.LinkElkDesc(L.OP_COS, 64, 32, 3, 1.0, 1e-6)
    b = L.LinkLeanBuffers()
    b.k, b.seg_cap, b.io_dtype = 343, 4096, L.IO_F32
    call = lambda n=1000, n_prev=0, build=1: lib.link_elk_core_lean_forward(ctypes.byref(b), ctypes.byref(grid), ctypes.byref(desc), n, n_prev, build, None)
    assert call() == L.LINK_ERR_ARG                                   # no buffers
    assert lib.link_elk_core_lean_forward(None, ctypes.byref(grid), ctypes.byref(desc), 10, 0, 1, None) == L.LINK_ERR_ARG
    assert call(n=-1) == L.LINK_ERR_ARG
    for field, bad in (("c", 48), ("r", 4), ("op", 7), ("cg", 0)):
        d2 = L.LinkElkDesc(L.OP_COS, 64, 32, 3, 1.0, 1e-6)
        setattr(d2, field, bad)
        assert lib.link_elk_core_lean_forward(ctypes.byref(b), ctypes.byref(grid), ctypes.byref(d2), 10, 0, 1, None) == L.LINK_ERR_ARG, field
    b.k = 353                                                         # beyond the slot capacity the form takes
    assert call() == L.LINK_ERR_ARG
    b.k, b.cnt_shift = 343, 6
    assert call() == L.LINK_ERR_ARG
    b.cnt_shift, b.seg_cap = 0, 8                                     # item lists too short for 1000 voxels
    assert call() == L.LINK_ERR_ARG
    big = L.LinkGrid()
    big.s = 1
    for a in range(4):
        big.lo[a], big.dim[a] = 0, (1024 if a < 3 else 1)           # 2^30 cells: an item is cell * 16 + chunk in 31 bits
    b.seg_cap = 4096
    assert lib.link_elk_core_lean_forward(ctypes.byref(b), ctypes.byref(big), ctypes.byref(desc), 10, 0, 1, None) == L.LINK_ERR_ARG


def test_block_driver_struct_and_arena_helper():
    """Section G (ABI 11): link_block_args_t as ctypes sees it is what the library was compiled with (checked at load through
    link_abi_struct_size(7)), and link_pair_plan_arena -- a host helper, no GPU -- lays the ten pieces out 16-byte aligned with the
    capacity link_pair_plan_build demands."""
    from link_amd import _lib as L
    lib = L.lib()
    assert L.ABI_VERSION >= 11 and lib.link_abi_struct_size(7) == ctypes.sizeof(L.LinkBlockArgs)
    offs = (ctypes.c_int64 * 11)()
    for n, kvol, skip in ((100000, 27, 1), (777, 27, 0), (5, 8, 0), (300000, 27, 1)):
        gran = lib.link_pair_plan_arena(n, kvol, skip, offs)
        nwg, cap = (n + 255) // 256, n * (kvol - skip)
        assert gran == (cap + 127 * kvol + 127) // 128
        o = [int(v) for v in offs]
        assert o[0] == 0 and all(v % 4 == 0 for v in o) and all(b > a for a, b in zip(o, o[1:]))
        sizes = [nwg * (kvol + 1), n, kvol + nwg * kvol + kvol + 1, nwg, gran, 8, n + 1, gran * 128, gran * 128, max(cap, 1)]
        assert all(o[i + 1] - o[i] >= sizes[i] for i in range(10))
    assert lib.link_pair_plan_arena(0, 27, 1, offs) == -1 and lib.link_pair_plan_arena(10, 65, 0, offs) == -1
    assert lib.link_elk_block_forward(None, None, None) == L.LINK_ERR_ARG       # argument validation before anything touches a device


def test_batch_kernels_resource_shape():
    """Section H (round 6): the three kernels of a batch call must fit ONE CU side by side -- a K1-role workgroup (4 waves), a
    K2-role workgroup (8 waves) and the insert's single-wave workgroups: per SIMD one K1 wave + two K2 waves + one insert wave within
    512 vector registers; no scratch (scratch traffic would break the K2 producers' counted vmcnt waits); the insert without LDS (the
    CU's 128 LDS granules of 1 280 bytes are taken by 65 (K1, padded) + 63 (K2): tools/coresidency_probe.hip).  Read from the built
    code object's metadata (tools/kernel_regs.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_regs import kernel_table
    from link_amd import build as hip_build
    hip_build.build()
    for obj in ("dense_batch.o", "dense_batch_f16.o", "dense_batch_bf16.o"):     # fp32 / fp16 / bf16 rows: one set of kernels each
        rows = [r for r in kernel_table(os.path.join(ROOT, "link_amd", "lib", "obj", obj))
                if "k_dc_batch" in r[0] and "k_dc_batch_spin" not in r[0] and "k_dc_batch_clear" not in r[0]]
        assert len(rows) == 7, (obj, [r[0] for r in rows])                   # insert + K1 x {cos, sin} + K2 x {cos, sin} x {r 2, 3}
        al = lambda v: (int(v) + 7) // 8 * 8                                 # vector registers are allocated in eights
        ins = [r for r in rows if "insert" in r[0]]
        k1 = [r for r in rows if "batch_k1" in r[0]]
        k2 = [r for r in rows if "batch_k2" in r[0]]
        assert len(ins) == 1 and len(k1) == 2 and len(k2) == 4
        for name, vgpr, agpr, sgpr, lds, scratch, wg in rows:
            assert int(scratch) == 0 and int(agpr) == 0, (name, scratch, agpr)
        assert int(ins[0][4]) == 0 and int(ins[0][6]) == 64                 # no LDS, single-wave workgroups
        worst = max(al(r[1]) for r in k1) + 2 * max(al(r[1]) for r in k2) + al(ins[0][1])
        assert worst <= 512, (obj, worst)
        # LDS granules: K1 static 256 + dynamic padded to 65 granules; K2 static 16 + its plane ring within 63
        assert all(int(r[4]) == 256 for r in k1) and all(int(r[4]) == 16 for r in k2)
