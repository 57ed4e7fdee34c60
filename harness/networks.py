"""link_amd/networks.py -- the reference's segmentation networks assembled from link_amd modules with the REFERENCE'S
attribute names, so that a reference checkpoint loads with strict=True and `forward` follows the reference line by line:

  build_reference_shaped_encoder   encoder half of ELKEncoder (segmentation/core/models/semantic_kitti/linkencoder.py:186-290,
                                   forward :339-368 up to x4) -- what bench.py --workload cfg3 / cfg4 time
  build_reference_shaped_unet      the whole ELKUNet (linkunet.py:186-385): stem, four encoder stages (down-conv, two residual
                                   blocks + tail || ELKBlock + tail, add, ReLU), four decoder stages (transposed conv, cat with the
                                   skip, two residual blocks), classifier

The reference's own classes construct on the aliased surface too (link_amd.install_as_torchsparse(); tests/test_cpu_abi.py);
these builders exist because /root/reference does not travel to the GPU box.  Only parameter names / shapes and the order of
operations are the reference's (interface-forced); every module is link_amd's.
"""
import torch
import torch.nn as nn


def _la(la):
    if la is None:
        import link_amd as la
    return la


def _blocks(la):
    spnn = la

    class BasicConvolutionBlock(nn.Module):                      # linkunet.py:18-35
        def __init__(self, inc, outc, ks=3, stride=1):
            super().__init__()
            self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride), spnn.BatchNorm(outc),
                                     spnn.ReLU(True))

        def forward(self, x):
            return self.net(x)

    class BasicDeconvolutionBlock(nn.Module):                    # linkunet.py:38-54
        def __init__(self, inc, outc, ks=3, stride=1):
            super().__init__()
            self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride, transposed=True),
                                     spnn.BatchNorm(outc), spnn.ReLU(True))

        def forward(self, x):
            return self.net(x)

    class ResidualBlock(nn.Module):                              # linkunet.py:57-92
        def __init__(self, inc, outc, ks=3):
            super().__init__()
            self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=1), spnn.BatchNorm(outc),
                                     spnn.ReLU(True), spnn.Conv3d(outc, outc, kernel_size=ks, stride=1),
                                     spnn.BatchNorm(outc))
            self.downsample = nn.Sequential() if inc == outc else nn.Sequential(
                spnn.Conv3d(inc, outc, kernel_size=1, stride=1), spnn.BatchNorm(outc))
            self.relu = spnn.ReLU(True)

        def forward(self, x):
            return self.relu(self.net(x) + self.downsample(x))

    return BasicConvolutionBlock, BasicDeconvolutionBlock, ResidualBlock


def build_reference_shaped_unet(la=None, cr=1.0, baseop="cos_x", groups=1, s=3, r=2, num_classes=19):
    """ELKUNet (linkunet.py:186-385) from link_amd modules, reference attribute names; forward(x) -> logits [N, classes]
    (also keeps the intermediate tensors of the last call in `self.trace` for the parity tests)."""
    la = _la(la)
    spnn = la
    BasicConvolutionBlock, BasicDeconvolutionBlock, ResidualBlock = _blocks(la)
    cs = [int(cr * x) for x in [64] * 9]

    class ELKUNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.s, self.r = s, r
            self.stem = nn.Sequential(spnn.Conv3d(4, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True),
                                      spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True))
            for i in (1, 2, 3, 4):
                setattr(self, f"down{i}", nn.Sequential(BasicConvolutionBlock(cs[i - 1], cs[i - 1], ks=2, stride=2)))
                setattr(self, f"stage{i}", nn.Sequential(ResidualBlock(cs[i - 1], cs[i]), ResidualBlock(cs[i], cs[i])))
                setattr(self, f"stage{i}_tail", nn.Sequential(spnn.Conv3d(cs[i], cs[i], kernel_size=3, stride=1), spnn.BatchNorm(cs[i])))
                setattr(self, f"elk{i}", la.ELKBlock(cs[i - 1], cs[i - 1], groups, baseop=baseop))
                setattr(self, f"elk{i}_tail", nn.Sequential(spnn.Conv3d(cs[i - 1], cs[i], kernel_size=3, stride=1), spnn.BatchNorm(cs[i])))
                setattr(self, f"activate{i}", nn.ReLU(True))
            for i, skip in zip((1, 2, 3, 4), (3, 2, 1, 0)):
                setattr(self, f"up{i}", nn.ModuleList([
                    BasicDeconvolutionBlock(cs[3 + i], cs[4 + i], ks=2, stride=2),
                    nn.Sequential(ResidualBlock(cs[4 + i] + cs[skip], cs[4 + i]), ResidualBlock(cs[4 + i], cs[4 + i]))]))
            self.classifier = nn.Sequential(nn.Linear(cs[8], num_classes))
            self.trace = {}

        def forward(self, x):
            x0 = self.stem(x)
            skips, prev = [x0], x0
            for i in (1, 2, 3, 4):                                # linkunet.py:345-364
                d = getattr(self, f"down{i}")(prev)
                xi = getattr(self, f"stage{i}_tail")(getattr(self, f"stage{i}")(d))
                lk = getattr(self, f"elk{i}_tail")(getattr(self, f"elk{i}")(d, d.s[0] * self.s, self.r))
                xi.F = getattr(self, f"activate{i}")(xi.F + lk.F)
                skips.append(xi)
                prev = xi
            y = prev
            self.trace = {f"x{i}": t for i, t in enumerate(skips)}
            for i, skip in zip((1, 2, 3, 4), (3, 2, 1, 0)):      # linkunet.py:367-381
                up = getattr(self, f"up{i}")
                y = up[1](la.cat([up[0](y), skips[skip]]))
                self.trace[f"y{i}"] = y
            return self.classifier(y.F)

    return ELKUNet()


def build_reference_shaped_encoder(la=None, c=16, baseop="cos_x", groups=1):
    """The encoder half of the reference's ELKEncoder (linkencoder.py:186-290) from link_amd modules with the
    reference's attribute names, so that the reference's own state_dict loads with strict=True
    (tests/golden/g_encoder_*.npz).  Forward = linkencoder.py:339-368 up to x4."""
    la = _la(la)
    spnn = la

    class BasicConvolutionBlock(nn.Module):                      # linkencoder.py:23-39
        def __init__(self, inc, outc, ks=3, stride=1):
            super().__init__()
            self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride), spnn.BatchNorm(outc),
                                     spnn.ReLU(True))

        def forward(self, x):
            return self.net(x)

    class ResidualBlock(nn.Module):                              # linkencoder.py:61-92
        def __init__(self, inc, outc, ks=3):
            super().__init__()
            self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=1), spnn.BatchNorm(outc),
                                     spnn.ReLU(True), spnn.Conv3d(outc, outc, kernel_size=ks, stride=1),
                                     spnn.BatchNorm(outc))
            self.downsample = nn.Sequential() if inc == outc else nn.Sequential(
                spnn.Conv3d(inc, outc, kernel_size=1, stride=1), spnn.BatchNorm(outc))
            self.relu = spnn.ReLU(True)

        def forward(self, x):
            y = self.net(x)
            sc = self.downsample(x)
            out = la.SparseTensor(y.F + sc.F, y.C, y.s)
            out.cmaps, out.kmaps = x.cmaps, x.kmaps
            return self.relu(out)

    class Encoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Sequential(spnn.Conv3d(4, c, kernel_size=3, stride=1), spnn.BatchNorm(c), spnn.ReLU(True),
                                      spnn.Conv3d(c, c, kernel_size=3, stride=1), spnn.BatchNorm(c), spnn.ReLU(True))
            for i in (1, 2, 3, 4):
                setattr(self, f"down{i}", nn.Sequential(BasicConvolutionBlock(c, c, ks=2, stride=2)))
                setattr(self, f"stage{i}", nn.Sequential(ResidualBlock(c, c), ResidualBlock(c, c)))
                setattr(self, f"stage{i}_tail", nn.Sequential(spnn.Conv3d(c, c, kernel_size=3, stride=1), spnn.BatchNorm(c)))
                setattr(self, f"elk{i}", la.ELKBlock(c, c, groups, baseop=baseop, variant="encoder"))
                setattr(self, f"elk{i}_tail", nn.Sequential(spnn.Conv3d(c, c, kernel_size=3, stride=1), spnn.BatchNorm(c)))

        def forward(self, x, s, r):
            x0 = self.stem(x)
            prev, outs = x0, []
            for i in (1, 2, 3, 4):
                d = getattr(self, f"down{i}")(prev)
                xi = getattr(self, f"stage{i}_tail")(getattr(self, f"stage{i}")(d))
                lk = getattr(self, f"elk{i}_tail")(getattr(self, f"elk{i}")(d, d.s[0] * s, r))
                xi.F = torch.relu(xi.F + lk.F)
                outs.append(xi)
                prev = xi
            return x0, outs

    return Encoder()
