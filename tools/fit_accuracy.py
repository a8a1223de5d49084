"""Accuracy of the randomized low-rank fit on a full-size layer against the exact truncated SVD (computed on the GPU
with torch.linalg.svdvals), for several subspace sizes and iteration counts:

    python tools/fit_accuracy.py --n 4096 --k 4096
Prints, per setting, the relative excess of ||E - U L^T||_F over the optimum sqrt(sum_{i>r} sigma_i^2) and the time."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from slim_amd import ops, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--rank", type=int, default=512)
    ap.add_argument("--configs", default="128:8:7,128:8:5,128:10:7,128:10:5,128:10:4,128:10:3",
                    help="comma list of oversample:iterations:log2 of the conditioning the orthonormalisations cover")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    _lib.load()
    N, K, r = args.n, args.k, args.rank
    W = B.synth_weight(N, K, 100, dev)
    X = B.synth_acts(16, 2048, K, 1, dev)
    sr = torch.zeros(K, device=dev)
    ops.scaler_row_accum(X.reshape(-1, K), sr, 0, 16, 1)
    sr = ops.shift_zeros(sr)
    s = ops.sqrt(sr)
    mask = ops.mask_nm(W, s, 2, 4)
    hist = ops.hist_abs(W)
    sf = ops.quant_search(hist, max(512, min(W.numel() // 1000, 20000)), 4)[2].reshape(1)
    _, E = ops.error_matrix(W, mask, s, sf, 4)
    Et = ops.transpose(E)
    Ef = E.float()
    sv = torch.linalg.svdvals(Ef)
    opt = float(torch.sqrt((sv[r:] ** 2).sum()))
    print(f"{N}x{K} rank {r}: sigma_1/sigma_r = {float(sv[0] / sv[r - 1]):.2f}, sigma_r/sigma_(r+128) = {float(sv[r - 1] / sv[r + 127]):.4f}, optimum {opt:.4f}")
    for cfg in args.configs.split(","):
        over, iters, ns = (int(v) for v in cfg.split(":"))
        ops.lowrank_fit(E, Et, r, over, 1, seed=0, ns_steps=ns)       # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit = ops.lowrank_fit(E, Et, r, over, iters, seed=0, ns_steps=ns)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        err = float(torch.norm(Ef - fit["U"].float() @ fit["L"].float().t()))
        print(f"  oversample {over:4d} iterations {iters} conditioning 2^{ns}: excess over optimum {err / opt - 1:.2e}   {dt * 1e3:.1f} ms  info {int(fit['info'].abs().sum())}")


if __name__ == "__main__":
    main()
