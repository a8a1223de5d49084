#!/usr/bin/env python3
"""Golden vectors for the adapter-pruning branch (prune_lora=True): slim.lora.prune_and_optimize_lora
(slim/lora.py:7-53) run as it is, on the CPU, by importing the reference (read-only at /root/reference) in the
development container.  The learning rate of that routine is 1e6 / min(K, N)^2, so the shapes are chosen wide
(min(K, N) = 4096 -> lr 0.0596) with a small rank to keep the CPU run short; a few iterations only, because the
Adam trajectory amplifies summation-order differences.  Only data is written.
    python tests/golden/make_golden_prune_lora.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.chdir(REF)
sys.path.insert(0, REF)

import numpy as np
import torch


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    from slim.lora import prune_and_optimize_lora
    out = {}
    for tag, (K, r, N, iters) in {"a": (4096, 8, 4096, 12), "b": (8192, 16, 4096, 6)}.items():
        g = torch.Generator().manual_seed(17 + K + r)
        L = (torch.randn(K, r, generator=g) * 0.05).to(torch.bfloat16)
        R = (torch.randn(r, N, generator=g) * 0.3).to(torch.bfloat16)
        L_in, R_in = L.clone(), R.clone()
        mask = prune_and_optimize_lora(L, R, num_iters=iters, lr_end_factor=1e-4)
        target = torch.matmul(L_in, R_in).float()
        loss = (torch.norm(torch.matmul(L.float(), R.float()) - target) / torch.norm(target)).item()
        Lp = L_in.clone()
        Lp[mask] = 0
        loss0 = (torch.norm(torch.matmul(Lp, R_in).float() - target) / torch.norm(target)).item()
        out.update({f"{tag}_L_in_bf16": bits(L_in), f"{tag}_R_in_bf16": bits(R_in), f"{tag}_L_out_bf16": bits(L),
                    f"{tag}_R_out_bf16": bits(R), f"{tag}_mask": mask.numpy().astype(np.uint8),
                    f"{tag}_iters": np.asarray(iters), f"{tag}_final_loss": np.asarray(loss),
                    f"{tag}_initial_loss": np.asarray(loss0)})
        print(tag, "initial", loss0, "final", loss)
    np.savez_compressed(os.path.join(HERE, "g12_prune_lora.npz"), **out)
    print("wrote g12_prune_lora", len(out), "arrays")


if __name__ == "__main__":
    main()
