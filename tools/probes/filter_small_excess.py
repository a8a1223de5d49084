"""Per-layer excess of the fit over the truncated SVD on the 4-block hidden-512 fixture model, as the driver runs it (hints
from block to block, so blocks 1.. are Chebyshev-filtered unless SLIM_FIT_FILTER=0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import slim_amd.prune as P, slim_amd.lora as LORA
import test_zy_perplexity_gpu as T
g = {k: torch.from_numpy(v) if isinstance(v, np.ndarray) and v.dtype.kind in "fiub" else v
     for k, v in np.load(os.path.join(os.path.dirname(T.__file__), "golden", "g16_e2e_large.npz")).items()}
class G(dict):
    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return v
g = G(g)
seen = []
real_finish = LORA.add_lora_finish
def finish(c, prescale=False):
    U, Ut, L, info = c.fit
    Ef = c.E.float()
    sv = torch.linalg.svdvals(Ef.double())
    r = c.rank
    opt = float(torch.sqrt((sv[r:] ** 2).sum()))
    s16 = c.s.to(torch.bfloat16).float().reshape(-1, 1)
    err = float(torch.norm(Ef.double() - U[:, :r].double() @ (L[:, :r].float() * s16).double().t()))
    seen.append((tuple(c.E.shape), r, err / opt - 1, int(info[-1]), int(info[-2]), int(info[-3]), c.level_hint))
    return real_finish(c, prescale=prescale)
LORA.add_lora_finish = finish
import slim_amd.prune
for seed in range(int(os.environ.get('NSEEDS', '2'))):
    seen.clear()
    m = T._compress_large(g, True, seed)
    for i, x in enumerate(seen if os.environ.get("VERBOSE") else []):
        print(seed, i // 7, x[0], "rank", x[1], f"excess {x[2]:+.2e}", "words", x[3:6], "hint", hex(x[6]))
    ex = [x[2] for x in seen]
    print("seed", seed, "block 0 mean", sum(ex[:7]) / 7, "later blocks mean", sum(ex[7:]) / len(ex[7:]), "max", max(ex))
    evalids = g["eval_ids"]
    print("ppl", T._ppl(m, evalids))
