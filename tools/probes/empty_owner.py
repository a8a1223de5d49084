import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from slim_amd.prune import compress_linears, block_stats
from slim_amd.quantization.quantization import Quantizer
from slim_amd import parallel
dev = torch.device("cuda", 0)
print("block_stats([]) ->", block_stats([]))
print("compress_linears([]) ->", compress_linears([], dev, 2, 4, quantizer=Quantizer("weight", num_bits=4, slim_quant=True), n_streams=7,
      level_hints={}, sparsity_ratio=0.5, lora_rank=0.1, slim_lora=True, quantize_lora=True, lora_tile_size=256))
parallel.broadcast_coalesced([], 0)
print("ok")
