"""Microbenchmark of the calibration Gram update H <- H n/(n+b) + (2/(n+b)) X^T X (slim_gram_accum, the statistic of
SparseGPT-style wrappers, reference slim/sparsegpt.py add_batch) for the input widths of a LLaMA-2-7B block, against the
dense bf16 MFMA peak.  Prints one JSON line per width.

    python tools/gram_bench.py --rows 16384 --widths 4096,11008
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MFMA_BF16_PEAK_TF = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384, help="tokens per update (8 samples x 2048)")
    ap.add_argument("--widths", default="4096,11008")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from slim_amd import ops
    dev = torch.device("cuda:0")
    for K in [int(v) for v in args.widths.split(",")]:
        x = torch.randn(args.rows, K, device=dev, dtype=torch.bfloat16)
        H = torch.zeros(K, K, device=dev, dtype=torch.float32)
        for _ in range(2):
            ops.gram_accum(x, H, 0, 8)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(args.iters):
            ops.gram_accum(x, H, 8 * (i + 1), 8)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.iters
        flops = 2.0 * args.rows * K * K              # the full product; the kernel computes the upper triangle and mirrors it
        ref = torch.zeros(K, K, device=dev, dtype=torch.float32)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xf = x.float()
        torch.cuda.synchronize()
        t0.record()
        ref = xf.t() @ xf
        t1.record()
        torch.cuda.synchronize()
        print(json.dumps({"K": K, "rows": args.rows, "ms": round(ms, 4), "TFLOPs_full_product": round(flops / ms / 1e9, 1),
                          "frac_of_bf16_mfma_peak": round(flops / ms / 1e9 / MFMA_BF16_PEAK_TF, 3),
                          "bytes_GB": round((args.rows * K * 2 + 2 * K * K * 4) / 1e9, 3),
                          "torch_fp32_matmul_ms": round(t0.elapsed_time(t1), 3)}), flush=True)


if __name__ == "__main__":
    main()
