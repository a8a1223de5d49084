"""Back-to-back timing of the library's NT bf16 GEMM on the shape classes of the low-rank fit, beside torch.matmul
(hipBLASLt) on the same operands as a known-good reference for what the hardware does at these sizes.

    python tools/gemm_probe.py [--l 640] [--rows 4096,11008]
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters=200, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--l", type=int, default=640)
    ap.add_argument("--rows", default="4096,11008")
    ap.add_argument("--tiles", default="-1", help="comma list of forced tile shapes (0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64, -1: auto)")
    ap.add_argument("--no-lib", action="store_true", help="skip the torch.matmul reference")
    args = ap.parse_args()
    from slim_amd import _lib, ops
    from slim_amd.ops import _p, _stream
    dev = torch.device("cuda:0")
    l = args.l
    shapes = [("small", l, l, l)]
    for rows in [int(v) for v in args.rows.split(",")]:
        shapes += [("apply", rows, l, l), ("gram", l, l, rows), ("sketch", l, rows, 4096 if rows != 4096 else 11008)]
    for name, M, N, Kd in shapes:
        A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
        Bt = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
        C32 = torch.empty(M, N, device=dev, dtype=torch.float32)
        C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        Ct = torch.empty(N, M, device=dev, dtype=torch.bfloat16)
        slab = torch.empty(16 * M * N if name == "gram" else 1, device=dev, dtype=torch.float32)
        st = _stream()
        outs = {"small": (C32, None, None), "apply": (None, C16, Ct), "gram": (C32, C16, None), "sketch": (None, C16, Ct)}[name]

        def ours():
            if name == "gram":
                tiles = ((l + 127) // 128) ** 2
                splits = max(1, min(16, (512 + tiles - 1) // tiles, (Kd + 63) // 64 // 4))
                _lib.call("slim_gemm_nt_bf16_splitk", _p(A), Kd, _p(Bt), Kd, M, N, Kd, 1.0, _p(C32), N, _p(C16), N, splits,
                          _p(slab), st)
            else:
                _lib.call("slim_gemm_nt_bf16", _p(A), Kd, _p(Bt), Kd, M, N, Kd, 1.0, 0.0, _p(outs[0]), N, _p(outs[1]), N,
                          _p(outs[2]), M, st)

        def lib():
            return torch.matmul(A, Bt.t())

        fl = 2.0 * M * N * Kd
        t_l = None if args.no_lib else timed(lib)
        for tile in [int(v) for v in args.tiles.split(",")]:
            _lib.query("slim_gemm_set_tile", tile)
            t_o = timed(ours)
            print(json.dumps({"shape": name, "M": M, "N": N, "Kd": Kd, "tile": tile, "ours_us": round(t_o, 2),
                              "hipblaslt_us": t_l and round(t_l, 2), "ours_TFLOPs": round(fl / t_o / 1e6, 1)}), flush=True)
        _lib.query("slim_gemm_set_tile", -1)


if __name__ == "__main__":
    main()
