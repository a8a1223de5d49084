"""Times rocSOLVER's symmetric eigen-solvers on an l x l SPD matrix like the Rayleigh-Ritz matrix of the fit
(decaying spectrum), and counts their kernel launches under rocprofv3 when run under it."""
import ctypes as C, sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
which = sys.argv[2] if len(sys.argv) > 2 else "all"
lib = C.CDLL("librocsolver.so.0", mode=C.RTLD_GLOBAL)
h = C.c_void_p()
assert lib.rocblas_create_handle(C.byref(h)) == 0
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = torch.randn(n, 4096, device=dev) * torch.logspace(0, -1, 4096, device=dev)
G0 = (B @ B.t()).contiguous()
st = torch.cuda.current_stream().cuda_stream
lib.rocblas_set_stream(h, C.c_void_p(st))
vp = C.c_void_p
def p(t): return vp(t.data_ptr())
D = torch.empty(n, device=dev); E = torch.empty(n, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
res = torch.zeros(1, device=dev); nsw = torch.zeros(1, dtype=torch.int32, device=dev)
ref_w = torch.linalg.eigvalsh(G0.double()).float()
def run(name, fn, reps=5):
    ts = []
    for _ in range(reps):
        A = G0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = fn(A)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    w = D.clone()
    V = A.t()      # column-major eigenvectors
    resid = (G0 @ V - V * w[None, :]).norm() / G0.norm()
    print(f"{name}: rc={rc} info={int(info)} min {1e3*min(ts):.2f} ms  eigval err {float((w.sort().values-ref_w).abs().max()/ref_w.max()):.2e}  residual {float(resid):.2e}"
          + (f" sweeps {int(nsw)}" if "j" in name else ""))
if which in ("all", "syevd"):
    run("syevd", lambda A: lib.rocsolver_ssyevd(h, 211, 121, n, p(A), n, p(D), p(E), p(info)))
if which in ("all", "syevj"):
    for tol in (1e-4, 1e-6):
        run(f"syevj tol={tol}", lambda A: lib.rocsolver_ssyevj(h, 252, 211, 121, n, p(A), n, C.c_float(tol), p(res), 20, p(nsw), p(D), p(info)))
if which in ("all", "syevdj"):
    run("syevdj", lambda A: lib.rocsolver_ssyevdj(h, 211, 121, n, p(A), n, p(D), p(info)))
if which in ("all", "syev"):
    run("syev", lambda A: lib.rocsolver_ssyev(h, 211, 121, n, p(A), n, p(D), p(E), p(info)))
