import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/slim_amd") else os.getcwd())
os.environ["SLIM_FWD_WIDE"] = "1"; os.environ["EXP_STAMP"] = "1"
from slim_amd import ops, _lib
from tools.forward_bench import make_layer
from slim_amd.packed import PackedLinear
from slim_amd.lora import lora_rank
dev = torch.device("cuda:0")
N, K, M = 28672, 8192, 4096
r = lora_rank((N, K), 0.1, 256)
lin = make_layer(N, K, r, 256, dev)
p = PackedLinear(lin)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
need = _lib.query("slim_linear24_workspace_bytes", M, N, K, r)
ws = torch.zeros(need + 4096, dtype=torch.uint8, device=dev)
for _ in range(3):
    y = ops.linear24_forward(x, p.vals, p.meta, p.sf, N, None, p.LeffT, p.ReffT, r, p.a_div, ws=ws)
torch.cuda.synchronize()
r128 = -(-r // 128) * 128
off = (M * r128 * 2 + 255) // 256 * 256 + (M * K * 2 + 255) // 256 * 256
st = ws[off:off + 64].view(torch.int64).cpu().tolist()
print("stamps", st[:7])
print("deltas (memtime ticks)", [st[i + 1] - st[i] for i in range(6)])
