import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from slim_amd import ops, _lib
dev = torch.device("cuda:0"); _lib.load()
N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
r = N // 8
W = B.synth_weight(N, K, 100, dev); X = B.synth_acts(8, 2048, K, 1, dev)
sr = torch.zeros(K, device=dev); ops.scaler_row_accum(X.reshape(-1, K), sr, 0, 8, 1); sr = ops.shift_zeros(sr); s = ops.sqrt(sr)
mask = ops.mask_nm(W, s, 2, 4); sf = ops.quant_search(ops.hist_abs(W), max(512, min(W.numel() // 1000, 20000)), 4)[2].reshape(1)
_, E = ops.error_matrix(W, mask, s, sf, 4); Et = ops.transpose(E); Ef = E.float()
sv = torch.linalg.svdvals(Ef); opt = float(torch.sqrt((sv[r:] ** 2).sum()))
print(f"{N}x{K} r={r}: s1/sr {float(sv[0]/sv[r-1]):.1f}")
for kw in (dict(cholesky_qr=True), dict(ns_steps=4), dict(ns_steps=12)):
    f = ops.lowrank_fit(E, Et, r, r // 4, 10, seed=0, **kw)
    U, L = f["U"].float(), f["L"].float()
    err = float(torch.norm(Ef - U @ L.t())) / opt - 1
    G = U.t() @ U
    print(kw, f"excess {err:+.2e}  max|U^T U - I| {float((G - torch.eye(r, device=dev)).abs().max()):.3e}  eig(U^T U) in [{float(torch.linalg.eigvalsh(G)[0]):.4f}, {float(torch.linalg.eigvalsh(G)[-1]):.4f}]  S0/sv0 {float(f['S'][0]/sv[0]):.4f} S_r/sv_r {float(f['S'][r-1]/sv[r-1]):.4f}")
