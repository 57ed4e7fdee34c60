import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, link_amd as la, link_encoder as LE
from link_amd.synth import s_kitti
dev = torch.device("cuda", 0)
co, fe = s_kitti(0)
coords, feats = torch.from_numpy(co).to(dev), torch.from_numpy(fe).to(dev)
torch.manual_seed(0)
net = LE.build_reference_shaped_encoder(la, 64, "cos_x", 1).to(dev).train()
st0 = la.SparseTensor(feats, coords, 1)
with torch.no_grad(): net(st0, 3, 2)
for _ in range(8):
    f = feats.detach().requires_grad_(True)
    x = la.SparseTensor(f, coords, 1); x.kmaps, x.cmaps = st0.kmaps, st0.cmaps
    net(x, 3, 2)[1][-1].F.square().sum().backward()
torch.cuda.synchronize()
