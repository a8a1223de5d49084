"""Single-threaded pass over the kernels of the path on one 4096 x 4096 linear (statistics over 16 x 2048 tokens,
mask, histograms + scale searches, error matrix, low-rank fit, residual, re-quantisation, packing, forward), meant to
be run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one counter per run).  Everything is enqueued from
the main thread on one stream: counter collection serialises dispatches and is not exercised with several
launching threads here.  tools/pmc_summarize.py turns the two CSVs into per-kernel HBM bytes per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLIM_COMPRESS_STREAMS"] = "1"
import bench as B
from slim_amd import ops, _lib
from slim_amd import prune as P
from slim_amd.packed import PackedLinear
from slim_amd.quantization.quantization import Quantizer

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
_lib.load()
N = K = 4096
lin = torch.nn.Linear(K, N, bias=False, device=dev, dtype=torch.bfloat16)
lin.weight.requires_grad_(False)
lin.weight.data = B.synth_weight(N, K, 100, dev)
X = B.synth_acts(16, 2048, K, 1, dev)
for rep in range(2):                       # first pass warms code objects and allocator
    lin._forward_hooks.clear()
    lin.weight.data = B.synth_weight(N, K, 100, dev)
    wrapped = B.collect_stats(lin, X)
    wrapped.scaler_row = ops.shift_zeros(wrapped.scaler_row)
    P.block_stats([wrapped])
    P.compress_linears([("probe", lin, wrapped)], dev, 2, 4, sparsity_ratio=0.5,
                       quantizer=Quantizer("weight", num_bits=4, slim_quant=True), lora_rank=0.1, slim_lora=True,
                       quantize_lora=True, lora_tile_size=256, n_streams=1)
    packed = PackedLinear(lin)
    for M in (1, 16, 4096):
        packed(torch.randn(M, K, device=dev, dtype=torch.bfloat16))
    torch.cuda.synchronize()
print("done")
