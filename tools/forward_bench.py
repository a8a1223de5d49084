"""Forward microbenchmark of one compressed Linear (2:4 sparse 4-bit weight + low-rank adapters):
the reference's emulated forward (dense bf16 nn.Linear + adapter hook, torch GEMMs) against the fused
forward from packed storage, over a range of batch sizes.  Prints one JSON line per batch size.

    python tools/forward_bench.py --out-features 28672 --in-features 8192 --rank-ratio 0.1 --batches 1,16,256,4096
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_layer(N, K, r, quant_block, dev):
    from slim_amd.quantization.quantization import Quantizer
    g = torch.Generator(device=dev).manual_seed(0)
    codes = torch.randint(-8, 8, (N, K), generator=g, device=dev).float()
    order = torch.rand(N, K // 4, 4, generator=g, device=dev).argsort(dim=-1)
    pruned = (order < 2).reshape(N, K)
    sf = torch.tensor([37.3], dtype=torch.float32, device=dev)
    W = (codes / sf).to(torch.bfloat16)
    W[pruned] = 0
    lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.bfloat16)
    lin.weight.data = W
    lin.scaling_factor = sf.reshape(1, 1)
    if r:
        lin.lora_rank = torch.tensor(r)
        lin.lora_left = torch.nn.Parameter((torch.randn(K, r, generator=g, device=dev) * 0.02 * math.sqrt(r)).to(torch.bfloat16))
        lin.lora_right = torch.nn.Parameter((torch.randn(r, N, generator=g, device=dev) * 0.02 * math.sqrt(r)).to(torch.bfloat16))
        if quant_block:
            lin.lora_quantizer = Quantizer("weight", num_bits=4, block_quantization=True, block_dim=quant_block)
    return lin


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-features", type=int, default=28672)
    ap.add_argument("--in-features", type=int, default=8192)
    ap.add_argument("--rank-ratio", type=float, default=0.1)
    ap.add_argument("--lora-tile", type=int, default=256, help="0: unquantised adapters of rank int(ratio*min(N,K))")
    ap.add_argument("--batches", default="1,16,64,256,1024,4096")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from slim_amd.lora import lora_rank
    from slim_amd.packed import PackedLinear
    from slim_amd.prune import add_lora_hook
    N, K = args.out_features, args.in_features
    r = lora_rank((N, K), args.rank_ratio, args.lora_tile or None) if args.rank_ratio > 0 else 0
    lin = make_layer(N, K, r, args.lora_tile or None, dev)
    packed = PackedLinear(lin)
    dense_bytes = N * K * 2 + (2 * (N + K) * r * 2 if r else 0)
    for M in [int(v) for v in args.batches.split(",")]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)

        def emulated():
            y = torch.nn.functional.linear(x, lin.weight.data)
            if r:
                add_lora_hook(lin, (x,), y)
            return y

        def fused():
            return packed(x)

        y_ref, y = emulated(), fused()
        same = (y == y_ref).float().mean().item()
        t_ref = timed(emulated, args.iters)
        t_new = timed(fused, args.iters)
        flops = 2.0 * M * (N * K + (N + K) * r)
        print(json.dumps({
            "N": N, "K": K, "rank": r, "adapters": "int4-group" if (r and args.lora_tile) else ("bf16" if r else "none"),
            "batch": M, "emulated_ms": round(t_ref, 4), "fused_ms": round(t_new, 4), "speedup": round(t_ref / t_new, 3),
            "packed_bytes": packed.packed_bytes(), "dense_bytes": dense_bytes,
            "fused_GBps_packed": round(packed.packed_bytes() / t_new / 1e6, 1),
            "fused_TFLOPs": round(flops / t_new / 1e9, 1), "emulated_TFLOPs": round(flops / t_ref / 1e9, 1),
            "outputs_identical_frac": round(same, 5)}), flush=True)


if __name__ == "__main__":
    main()
