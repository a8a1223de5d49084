"""Steady-state throughput of the library's NT bf16 GEMM variants on a large square product (every CU holds its full
complement of workgroups), beside torch.matmul (hipBLASLt).
    python tools/gemm_peak.py [--n 8192] [--tiles 0,4,10,5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_probe import timed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--tiles", default="0,4,10,5")
    args = ap.parse_args()
    from slim_amd import _lib, ops
    n = args.n
    A = (torch.randn(n, n, device="cuda") * 0.1).to(torch.bfloat16)
    B = (torch.randn(n, n, device="cuda") * 0.1).to(torch.bfloat16)
    flops = 2.0 * n ** 3
    us = timed(lambda: torch.matmul(A, B.t()), iters=20, warmup=3)
    print(f"hipBLASLt: {us:.1f} us  {flops / us / 1e6:.0f} TFLOP/s")
    for t in [int(x) for x in args.tiles.split(",")]:
        _lib.query("slim_gemm_set_tile", t)
        us = timed(lambda: ops.gemm_nt(A, B, out_bf16=True), iters=20, warmup=3)
        print(f"tile {t}: {us:.1f} us  {flops / us / 1e6:.0f} TFLOP/s")
    _lib.query("slim_gemm_set_tile", -1)


if __name__ == "__main__":
    main()
