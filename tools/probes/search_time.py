"""The scale search alone, 200 times on a 4096 x 11008 weight's histogram (20 000 bins), for a per-kernel profile."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from slim_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
W = (torch.randn(4096, 11008, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
h = ops.hist_abs(W)
for _ in range(200):
    out = ops.quant_search(h, 20000, 4)
torch.cuda.synchronize()
print(float(out[0]), int(out[1]))
