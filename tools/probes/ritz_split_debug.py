import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from slim_amd import ops
def al(x): return (x + 63) & ~63
g = torch.Generator().manual_seed(5)
l, keep, top = 320, 256, 256
w = torch.linspace(1, 2, l, dtype=torch.float64)
Q, _ = torch.linalg.qr(torch.randn(l, l, generator=g, dtype=torch.float64))
M = ((Q * w) @ Q.T).float().cuda().unsqueeze(0).contiguous()
for degs in ([1], [2], [4], [8], [8, 12], [8, 12, 20]):
    vecs, evals, info, ws = ops.ritz_split(M, top, [keep], degs, _workspace=True)
    torch.cuda.synchronize()
    c = l - keep; cp = (c + 3) & ~3
    blk = cp * l
    off = 0
    offs = {}
    for name, n in [("Y0", blk), ("Y1", blk), ("Y2", blk), ("V", blk), ("tau", cp), ("Bm", cp * cp), ("P0", l * l), ("P1", l * l), ("scal", 16)]:
        offs[name] = off; off += al(n * 4)
    scal = ws[offs["scal"]:offs["scal"] + 64].view(torch.float32).cpu().numpy()
    Md = M[0].double().cpu().numpy()
    ev = np.linalg.eigvalsh(Md)
    W = vecs[0, l - keep:].double().cpu().numpy()
    kept = np.trace(W @ Md @ W.T)
    print("degs", degs, "s", scal[0], "n3", scal[1], "b", scal[2], "a", scal[3], "coefA", scal[4:7], "coefB", scal[8:11], "lo", scal[12], "hi", scal[13],
          "| lam_max", ev[-1], "lam_c", ev[c - 1], "| dropped excess", (ev.sum() - kept) / ev[:c].sum() - 1)
# reference of one Chebyshev step against the kernel: degree 1 output lives in Y1
vecs, evals, info, ws = ops.ritz_split(M, top, [keep], [2], _workspace=True)
torch.cuda.synchronize()
Y = [ws[offs[n]:offs[n] + blk * 4].view(torch.float32).reshape(cp, l).double().cpu().numpy() for n in ("Y0", "Y1", "Y2")]
scal = ws[offs["scal"]:offs["scal"] + 64].view(torch.float32).cpu().numpy()
# after the run the bounds were overwritten? (single round: no) coefA/B as used
cA, cB = scal[4:7], scal[8:11]
Y1 = cA[0] * (Y[0] @ Md) + cA[1] * Y[0]
Y2 = cB[0] * (Y1 @ Md) + cB[1] * Y1 + cB[2] * Y[0]
print("step 1 rel err", np.abs(Y[1] - Y1).max() / np.abs(Y1).max(), "step 2 rel err", np.abs(Y[2] - Y2).max() / np.abs(Y2).max())
vecs, evals, info, ws = ops.ritz_split(M, top, [keep], [1], _workspace=True)
torch.cuda.synchronize()
V = ws[offs["V"]:offs["V"] + blk * 4].view(torch.float32).reshape(cp, l).cpu().numpy()
tau = ws[offs["tau"]:offs["tau"] + cp * 4].view(torch.float32).cpu().numpy()
Y1 = ws[offs["Y1"]:offs["Y1"] + blk * 4].view(torch.float32).reshape(cp, l).double().cpu().numpy()
print("Y1 finite", np.isfinite(Y1).all(), "V finite", np.isfinite(V).all(), "tau", tau[:6], "nan rows of V", np.where(~np.isfinite(V).all(1))[0][:10])
# reference Householder QR of Y1^T
A = Y1.T.copy()
for j in range(3):
    x = A[j:, j]; nrm = np.linalg.norm(x); beta = -np.sign(x[0]) * nrm if x[0] != 0 else -nrm
    t = (beta - x[0]) / beta; v = x / (x[0] - beta); v[0] = 1
    print("ref step", j, "tau", t, "v[1:4]", v[1:4], "gpu v", V[j, j + 1:j + 4], "gpu v[j]", V[j, j], "above", np.abs(V[j, :j]).max() if j else 0)
    A[j:, :] -= t * np.outer(v, v @ A[j:, :])
print("vecs finite", torch.isfinite(vecs).all().item())
