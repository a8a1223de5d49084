"""Time of the fit's front half (Gram matrix, subspace iteration, projection, Rayleigh-Ritz matrix) for B same-shape problems
in lockstep, alone on the chip: how the launch chain scales with the batch.  Run on the GPU box."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from slim_amd import ops

dev = torch.device("cuda:0")
N = K = int(os.environ.get("PROBE_N", "4096"))
r, over, iters = 512, 64, 10
l = ops.fit_subspace(N, K, r, over)
for ns in (2, 3):
    for split in (True, False):
        for B in (1, 2, 3, 4, 7):
            g = torch.Generator(device=dev).manual_seed(5)
            E = (torch.randn(B, N, K, device=dev, generator=g) * torch.rand(1, 1, K, device=dev, generator=g).add(0.5)).to(torch.bfloat16)
            Et = E.transpose(1, 2).contiguous()
            M = torch.empty(B, l, l, dtype=torch.float32, device=dev)
            flags = ops.fit_flags(False, False, False, split)
            for _ in range(2):
                ctx = ops.lowrank_fit_front(E, Et, r, over, iters, M, seed=0, flags=flags, ns_steps=ns)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ctx = ops.lowrank_fit_front(E, Et, r, over, iters, M, seed=0, flags=flags, ns_steps=ns)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print(f"N=K={N} cond_e {ns} sketch_split {split} B={B}: {dt * 1e3:.2f} ms ({dt * 1e3 / B:.2f} per problem)", flush=True)
            del E, Et, M, ctx
