"""Host cost per torch op / library call on this box."""
import time, torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from slim_amd import ops
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
def t(name, fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    print(f"{name}: host {h / n * 1e6:.1f} us/op, with GPU {w / n * 1e6:.1f} us/op", flush=True)
t("torch.empty", lambda: torch.empty(1024, device="cuda"))
t("x.ne(0)", lambda: x.ne(0))
t("x+1", lambda: x + 1)
W = torch.zeros(256, 256, dtype=torch.bfloat16, device="cuda")
h = torch.zeros(32768, dtype=torch.int32, device="cuda")
t("ops.hist_abs", lambda: ops.hist_abs(W, h))
t("ops.transpose", lambda: ops.transpose(W))
t("ops.quant_search", lambda: ops.quant_search(h, 512, 4))
