# fp64 VGG16, N=2: perturb the output of features.21 (pre-pool4) by 1e-6 relative noise and see how much the
# gradients of the earlier layers move (relative L2).  A smooth network would move ~1e-6.
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch.nn.functional as F
from umpr_amd.synthetic import make_param_state
torch.set_num_threads(8)
P = make_param_state(71, 50, 10, 1, False)
pre = "visual_net.vgg16.0."
cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
g = torch.Generator().manual_seed(3)
x = torch.rand(2, 3, 224, 224, generator=g).double()
def run(noise):
    vp = {k: v.detach().double().requires_grad_(True) for k, v in P.items() if k.startswith(pre + "features")}
    h = x; i = 0
    for c in cfg:
        if c == 'M':
            h = F.max_pool2d(h, 2, 2); i += 1
        else:
            h = F.conv2d(h, vp[f"{pre}features.{i}.weight"], vp[f"{pre}features.{i}.bias"], padding=1)
            if i == 21 and noise:
                gn = torch.Generator().manual_seed(9)
                h = h + noise * h.detach().abs().mean() * torch.randn(h.shape, generator=gn, dtype=torch.float64)
            h = F.relu(h); i += 2
    gout = torch.randn(h.shape, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    (h * gout).sum().backward()
    return {k: v.grad for k, v in vp.items()}
g0 = run(0.0)
for nz in (1e-7, 1e-6, 4e-6):
    g1 = run(nz)
    print("noise", nz, flush=True)
    for i in (0, 5, 12, 14, 17, 19, 21, 24):
        k = f"{pre}features.{i}.weight"
        print(f"  features.{i}.weight rel L2 move {float((g1[k]-g0[k]).norm()/g0[k].norm()):.3e}", flush=True)
