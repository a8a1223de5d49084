"""Time of the eigen-stage alternatives on 4 problems of 576 x 576 (one chain of the benchmark block)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from slim_amd import ops
g = torch.Generator().manual_seed(0)
l, B = 576, 4
Q, _ = torch.linalg.qr(torch.randn(l, l, generator=g, dtype=torch.float64))
w = torch.linspace(1140, 3960, l, dtype=torch.float64)
M = ((Q * w) @ Q.T).float().cuda().unsqueeze(0).repeat(B, 1, 1).contiguous()
for name, fn in (("ritz_split", lambda: ops.ritz_split(M, 512, [508] * B)),
                 ("tridiag+tail", lambda: (lambda A: (lambda deq: ops.eigh_tail(deq[0], deq[1], deq[2], A, 512))(ops.tridiagonalize_batched(A)))(M.clone()))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(name, "ms per call", (time.perf_counter() - t0) * 100)
