"""Per-workgroup phase timing of one GEMM launch (entry / first tile staged / K-loop done / epilogue done) from the
library's in-kernel timestamps.   python tools/gemm_trace.py M N Kd [variant ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from slim_amd import _lib, ops
    from slim_amd.ops import _p
    M, N, Kd = (int(v) for v in sys.argv[1:4])
    variants = [int(v) for v in sys.argv[4:]] or [0]
    outs = os.environ.get("OUTS", "bt")     # f: fp32 C, b: bf16 C, t: transposed bf16 C
    kw = dict(out_f32="f" in outs, out_bf16="b" in outs, out_t="t" in outs)
    A = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    Bt = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
    cap = 4096
    for variant in variants:
        _lib.query("slim_gemm_set_tile", variant)
        for _ in range(3):
            ops.gemm_nt(A, Bt, **kw)
        buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        _lib.query("slim_gemm_set_trace", _p(buf), cap)
        ops.gemm_nt(A, Bt, **kw)
        torch.cuda.synchronize()
        _lib.query("slim_gemm_set_trace", None, 0)
        t = buf.view(cap, 8).cpu()
        t = t[t[:, 0] > 0].double() * 0.01          # us
        t0 = t[:, 0].min()
        rel = t - t0
        print(f"variant {variant}: {t.shape[0]} workgroups; kernel span {rel[:, 3].max():.2f} us; entry spread "
              f"{rel[:, 0].max():.2f} us; per-workgroup mean: stage-first {(t[:, 1] - t[:, 0]).mean():.2f}  k-loop "
              f"{(t[:, 2] - t[:, 1]).mean():.2f}  epilogue {(t[:, 3] - t[:, 2]).mean():.2f}  total {(t[:, 3] - t[:, 0]).mean():.2f}"
              f" (max {(t[:, 3] - t[:, 0]).max():.2f}); epilogue parts: to-LDS {(t[:, 4] - t[:, 2]).mean():.2f} rows {(t[:, 5] - t[:, 4]).mean():.2f} transposed {(t[:, 3] - t[:, 5]).mean():.2f}")
    _lib.query("slim_gemm_set_tile", -1)


if __name__ == "__main__":
    main()
