"""Which torch-level copies / fills / AccumulateGrad nodes does one full-model training step issue?  torch.profiler over four steps (batch 16).
    python tools/prof_copies.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from umpr_amd.config import Config
from umpr_amd.model import UMPR
from umpr_amd.optim import FusedAdam
from umpr_amd.synthetic import make_batch, make_param_state
from umpr_amd.train import train_step
dev = torch.device("cuda:0")
Config.extend({"dtype": "fp32"})
cfg = Config(argv=[]); cfg.views = ["unknown"]
P = make_param_state(301, 50, 600, 1, False, m_scale=0.05)
b = make_batch(310, 16, 600, 1, full_pad=True)
b = tuple(t.to(dev) if i not in (3, 4, 5) else t for i, t in enumerate(b))
m = UMPR(cfg, P["embedding.weight"].numpy()); m.load_state_dict(P); m = m.to(dev)
opt = FusedAdam(m, 1e-3, 1e-3)
for _ in range(3): train_step(m, opt, b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(4): train_step(m, opt, b)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if any(k in e.key for k in ("copy", "Memcpy", "memcpy", "fill", "zero", "Memset", "contiguous", "clone", "to"))]
for e in sorted(rows, key=lambda e: -e.count)[:40]:
    print(f"{e.count/4:6.1f}/step  {e.key[:50]:50s} {str(e.input_shapes)[:90]}")
