"""Wall time of the adapter-pruning optimiser (slim_amd.lora.prune_and_optimize_lora) at full layer sizes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slim_amd import lora
g = torch.Generator().manual_seed(1)
for (K, r, N) in ((4096, 512, 4096), (4096, 512, 11008)):
    L = (torch.randn(K, r, generator=g) * 0.05).to(torch.bfloat16).cuda()
    R = (torch.randn(r, N, generator=g) * 0.3).to(torch.bfloat16).cuda()
    lora.prune_and_optimize_lora(L.clone(), R.clone(), 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lora.prune_and_optimize_lora(L, R, 200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(K, r, N, "200 iterations", round(dt * 1e3, 1), "ms ->", round(dt * 5, 2), "s per 1000;", round(6.0 * K * r * N * 200 / dt / 1e12, 1), "TFLOP/s fp32")
