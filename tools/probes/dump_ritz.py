"""Dump the Rayleigh-Ritz matrices (l x l, fp32) the fit's front half produces on full-size synthetic error matrices, for
offline experiments with the eigen-stage (gpurun_out/ritz_*.npy)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fullsize_gpu as T
from slim_amd import ops
from slim_amd.lora import fit_error_matrix_front, fit_plan, outlier_columns

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for N, K, rank in [(4096, 4096, 512), (4096, 11008, 512), (1024, 8192, 256), (768, 768, 256), (3072, 768, 256)]:
    E, sr = T._error_matrix(ops, N, K, 100 + N % 97 + K % 89)
    outl, spread = outlier_columns(sr)
    rf, over, l, t = fit_plan(N, K, rank, outl)
    M = torch.empty(l, l, dtype=torch.float32, device="cuda:0")
    p = fit_error_matrix_front(E, ops.transpose(E), rank, outl, spread, M)
    torch.cuda.synchronize()
    sv = torch.linalg.svdvals(E.float())
    np.save(os.path.join(ROOT, "gpurun_out", f"ritz_{N}x{K}_r{rank}_l{l}.npy"), M.cpu().numpy())
    np.save(os.path.join(ROOT, "gpurun_out", f"sv_{N}x{K}.npy"), sv.cpu().numpy())
    print(N, K, rank, "l", l, "rf", rf, "t", t, "E norm^2", float((E.float() ** 2).sum()), flush=True)
