"""The statistics kernel alone (128 samples x 2048 tokens, K from argv, default 4096), 20 calls, for a per-kernel profile."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from slim_amd import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(128 * 2048, K, device="cuda").to(torch.bfloat16)
for _ in range(20):
    out = ops.colsumsq(x, 128)
torch.cuda.synchronize()
print(float(out.sum()))
