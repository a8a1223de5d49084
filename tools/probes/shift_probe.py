"""Subspace iteration on the shifted Gram matrix C - sigma I: excess of the fit over the truncated SVD against the number of
iterations, for sigma = f x (smallest wanted eigenvalue).  Built from the Gram-side pieces (gramfit_iterate, gramfit_project,
the eigen-solver, gramfit_factors) after the outlier columns have been split off.  Run on the GPU box."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import slim_amd.lora as LORA
from slim_amd import ops
import test_fullsize_gpu as T

shapes = [(4096, 11008, 512), (4096, 4096, 512), (11008, 4096, 512)]
if os.environ.get("PROBE_BIG"):
    shapes += [(8192, 28672, 1024)]
for N, K, rank in shapes:
    E, sr = T._error_matrix(ops, N, K, 100 + N % 97 + K % 89)
    outl, spread = LORA.outlier_columns(sr, rank=rank)
    rf, over, l, t = LORA.fit_plan(N, K, rank, outl)
    Et = ops.transpose(E)
    if t:
        E2, UO, Ct, _, _ = LORA.deflate_outliers(E, Et, outl)
        Et2 = ops.transpose(E2)
    else:
        E2, Et2 = E, Et
    right = K <= N
    X = Et2 if right else E2                 # [Dm, Dx]
    Dm = X.shape[0]
    C32 = ops.gemm_nt(X, X, out_f32=True)[0]
    ev = torch.linalg.eigvalsh(C32.double()).flip(0)     # descending
    mean = float(ev.sum() / Dm)
    lam_l, lam_r = float(ev[l - 1]), float(ev[rf - 1])
    opt = float(ev[rf:].sum().sqrt())
    print(f"{N}x{K}: l {l} rf {rf} t {t} lambda_1/lambda_r {float(ev[0]) / lam_r:.1f} lambda_r/lambda_l {lam_r / lam_l:.4f} "
          f"lambda_l/mean {lam_l / mean:.3f} cond_e {LORA.fit_ns_steps(spread)}", flush=True)
    cond = LORA.fit_ns_steps(spread)
    for f, iters_list in [(0.0, (10, 6)), (0.4, (7, 6, 5)), (0.5, (6, 5, 4))]:
        for iters in iters_list:
            C16 = C32.clone()
            C16.diagonal().sub_(f * lam_l)
            C16 = C16.to(torch.bfloat16)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            P, Pt, info = ops.gramfit_iterate(C16, l, iters, seed=0, cond_e=cond + (1 if f else 0))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            Z, Zt, M = ops.gramfit_project(E2 if right else Et2, Pt)
            Mb = M.reshape(1, l, l).clone()
            d, e, Q = ops.tridiagonalize_batched(Mb)
            ops.eigh_tail(d, e, Q, Mb, rf)
            dm, _, rows, _, S = ops.gramfit_factors(right, P, Z, Mb[0], d[0], rf)
            U, L = (rows, dm) if right else (dm, rows)
            err2 = 0.0
            for r0 in range(0, N, 2048):
                D = E2[r0:r0 + 2048].double() - U[r0:r0 + 2048].double() @ L.double().t()
                err2 += float((D * D).sum())
            print(f"   shift {f:.1f} lambda_l iters {iters}: excess {err2 ** 0.5 / opt - 1:+.2e}  smallest Ritz/lambda_l "
                  f"{float(d[0, 0]) / lam_l:.3f} info {int(info[:-1].abs().sum())} iterate {dt * 1e3:.2f} ms", flush=True)
