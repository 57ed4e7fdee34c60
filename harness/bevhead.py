"""Dense BEV half of the detection model BASELINE.json's configs[4] names ("backbone + CenterPoint head"): the RPN neck
and the CenterHead, restated in plain torch from the structure the reference builds them with
(detection/det3d/models/necks/rpn.py:24-160, detection/det3d/models/bbox_heads/center_head.py:67-112,170-250, configured by
detection/configs/nusc/voxelnet/nusc_centerpoint_voxelnet_0075voxel_fix_bn_z_elkv3.py:36-58).  Both are ordinary
Conv2d / BatchNorm2d / ReLU stacks in the reference too (no custom kernels: they run on the vendor's dense-convolution
library), OUTSIDE the LinK hot path (not a row); they exist here only so that `bench.py --workload
cfg5 --bev` can time the backbone together with what consumes its BEV map.  Random-init weights, attribute layout chosen
for this harness (not state_dict compatible with det3d; decoding / NMS / losses are not built)."""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn


def _cbr(cin: int, cout: int, k: int, stride: int = 1, pad: int = 0, eps: float = 1e-3, momentum: float = 0.01) -> List[nn.Module]:
    return [nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False), nn.BatchNorm2d(cout, eps=eps, momentum=momentum), nn.ReLU()]


class BevRPN(nn.Module):
    """rpn.py: per level a strided 3x3 convolution (explicit zero padding) + `layer_nums[i]` 3x3 convolutions, each with
    BatchNorm (eps 1e-3) + ReLU; every level is brought to the first level's resolution (transposed convolution with kernel =
    stride, or a 1x1 convolution at stride 1) and the results are concatenated along channels."""

    def __init__(self, layer_nums: Sequence[int] = (5, 5), ds_layer_strides: Sequence[int] = (1, 2),
                 ds_num_filters: Sequence[int] = (128, 256), us_layer_strides: Sequence[int] = (1, 2),
                 us_num_filters: Sequence[int] = (256, 256), num_input_features: int = 256):
        super().__init__()
        assert len(layer_nums) == len(ds_layer_strides) == len(ds_num_filters) == len(us_layer_strides) == len(us_num_filters)
        cin = [num_input_features, *ds_num_filters[:-1]]
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        for i, layers in enumerate(layer_nums):
            mods: List[nn.Module] = [nn.ZeroPad2d(1), *_cbr(cin[i], ds_num_filters[i], 3, stride=ds_layer_strides[i])]
            for _ in range(layers):
                mods += _cbr(ds_num_filters[i], ds_num_filters[i], 3, pad=1)
            self.blocks.append(nn.Sequential(*mods))
            up = int(us_layer_strides[i])
            if up > 1:
                first = nn.ConvTranspose2d(ds_num_filters[i], us_num_filters[i], up, stride=up, bias=False)
            else:
                first = nn.Conv2d(ds_num_filters[i], us_num_filters[i], 1, stride=1, bias=False)
            self.deblocks.append(nn.Sequential(first, nn.BatchNorm2d(us_num_filters[i], eps=1e-3, momentum=0.01), nn.ReLU()))
        self.out_channels = int(sum(us_num_filters))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ups = []
        for block, deblock in zip(self.blocks, self.deblocks):
            x = block(x)
            ups.append(deblock(x))
        return torch.cat(ups, dim=1)


class BevCenterHead(nn.Module):
    """center_head.py: a shared 3x3 convolution (+ BatchNorm + ReLU) to 64 channels, then per task one small stack per output
    (`common_heads` + the class heat map): (num_conv - 1) x [3x3 convolution, BatchNorm, ReLU] and a final 3x3 convolution
    with bias; the heat map's final bias starts at -2.19."""

    NUSC_TASKS = (1, 2, 2, 1, 2, 2)                    # classes per task (the config's six groups of nuScenes classes)
    COMMON_HEADS = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}

    def __init__(self, in_channels: int = 512, num_classes: Sequence[int] = NUSC_TASKS, common_heads: Dict[str, Tuple[int, int]] = None,
                 share_conv_channel: int = 64, head_conv: int = 64, num_hm_conv: int = 2, init_bias: float = -2.19):
        super().__init__()
        common_heads = dict(common_heads or self.COMMON_HEADS)
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, 3, padding=1, bias=True),
                                         nn.BatchNorm2d(share_conv_channel), nn.ReLU(inplace=True))
        self.tasks = nn.ModuleList()
        for ncls in num_classes:
            heads = dict(common_heads)
            heads["hm"] = (int(ncls), num_hm_conv)
            task = nn.ModuleDict()
            for name, (classes, num_conv) in heads.items():
                mods: List[nn.Module] = []
                for _ in range(num_conv - 1):
                    mods += [nn.Conv2d(share_conv_channel, head_conv, 3, padding=1, bias=True), nn.BatchNorm2d(head_conv), nn.ReLU()]
                mods.append(nn.Conv2d(head_conv, classes, 3, padding=1, bias=True))
                if name == "hm":
                    nn.init.constant_(mods[-1].bias, init_bias)
                task[name] = nn.Sequential(*mods)
            self.tasks.append(task)

    def forward(self, x: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        x = self.shared_conv(x)
        return [{name: head(x) for name, head in task.items()} for task in self.tasks]


class BevHalf(nn.Module):
    """RPN + CenterHead on the backbone's BEV map [B, 256, H, W] -> per-task prediction maps at the same resolution."""

    def __init__(self, num_input_features: int = 256):
        super().__init__()
        self.neck = BevRPN(num_input_features=num_input_features)
        self.head = BevCenterHead(in_channels=self.neck.out_channels)

    def forward(self, bev: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
        return self.head(self.neck(bev))
