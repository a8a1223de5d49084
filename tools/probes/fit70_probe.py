"""Excess of the randomized fit over the truncated SVD at the two MLP shapes of a 70B-shaped model (rank 1024), for
several iteration counts and oversampling ratios, with the fit's time.  Run on the GPU box."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import slim_amd.lora as LORA
from slim_amd import ops
import test_fullsize_gpu as T
import time
for N, K, rank in [(8192, 28672, 1024), (28672, 8192, 1024)]:
    E, sr = T._error_matrix(ops, N, K, 300 + N % 97)
    outl, spread = LORA.outlier_columns(sr, rank=rank)
    opt, cond = T._optimum_from_gram(E, rank)
    Et = ops.transpose(E)
    for iters, div in [(10, 8), (12, 8), (14, 8), (10, 4), (12, 4)]:
        LORA.FIT_ITERS, LORA.FIT_OVERSAMPLE_DIV, LORA.FIT_ITERS_LARGE_EXTRA = iters, div, 0
        LORA.fit_error_matrix(E, Et, rank, None, outl, spread)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        U, Ut, L, info = LORA.fit_error_matrix(E, Et, rank, None, outl, spread)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        err2 = 0.0
        Uf, Lf = U[:, :rank].double(), L[:, :rank].double()
        for r0 in range(0, N, 4096):
            D = E[r0:r0 + 4096].double() - Uf[r0:r0 + 4096] @ Lf.t()
            err2 += float((D * D).sum())
        print(N, K, "iters", iters, "over r/%d" % div, "excess %+.2e" % (err2 ** 0.5 / opt - 1), "%.1f ms" % (dt * 1e3), flush=True)
