"""Where one bench step (one LLaMA-2-7B block) spends its wall-clock: statistics pass, host enqueue time of the
compression jobs per linear, and the GPU drain after the host is done."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from slim_amd import ops, _lib
from slim_amd import prune as P
from slim_amd.quantization.quantization import Quantizer

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
_lib.load()
lins, acts = {}, {}
for i, (name, n, k, src) in enumerate(B.LLAMA2_7B_BLOCK):
    lins[i] = torch.nn.Linear(k, n, bias=False, device=dev, dtype=torch.bfloat16)
    lins[i].weight.requires_grad_(False)
    if src not in acts:
        acts[src] = B.synth_acts(128, 2048, k, 1 + len(acts), dev)
W0 = {i: B.synth_weight(n, k, 100 + i, dev) for i, (_, n, k, _) in enumerate(B.LLAMA2_7B_BLOCK)}
only = os.environ.get("ONLY")
idxs = [int(v) for v in only.split(",")] if only else list(range(len(B.LLAMA2_7B_BLOCK)))
streams = int(os.environ.get("STREAMS", "7"))

def step(verbose):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wrapped = {}
    for i in idxs:
        lins[i].weight.data = W0[i]
        lins[i]._forward_hooks.clear()
        wrapped[i] = B.collect_stats(lins[i], acts[B.LLAMA2_7B_BLOCK[i][3]])
        wrapped[i].scaler_row = ops.shift_zeros(wrapped[i].scaler_row)
    t1 = time.perf_counter()
    P.block_stats([wrapped[i] for i in idxs])
    t2 = time.perf_counter()
    marks = {}
    def cb(name, lin, w):
        marks[name] = time.perf_counter()
    P.compress_linears([(B.LLAMA2_7B_BLOCK[i][0], lins[i], wrapped[i]) for i in idxs], dev, 2, 4,
                       quantizer=Quantizer("weight", num_bits=4, slim_quant=True), lora_rank=0.1, slim_lora=True,
                       quantize_lora=True, lora_tile_size=256, n_streams=streams, callback=cb)
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if verbose:
        print(f"stats enqueue {1e3*(t1-t0):.1f} ms | block_stats (sync) {1e3*(t2-t1):.1f} | compress host {1e3*(t3-t2):.1f} | "
              f"gpu drain {1e3*(t4-t3):.1f} | total {1e3*(t4-t0):.1f}")
        print("  host finish times after compress start:", {k: round(1e3*(v-t2), 1) for k, v in marks.items()})
for it in range(4):
    step(it >= 1)
