#!/usr/bin/env python3
"""Golden vectors for the variants of the path next to the headline recipe, generated like make_golden.py by
importing the reference (read-only at /root/reference) in the development container: the tiny LLaMA model of
g8_e2e_llama run through the reference's prune_and_quantize with
  g11_unstructured : sparsity_type="unstructured" (per-row stable sort, slim/prune.py:275-279), 4-bit SLiM-Quant,
                     adapters of rank 0.1, tile 128
  g11_nolora       : 2:4, 4-bit SLiM-Quant, lora_rank = 0 (quantise, then mask; slim/prune.py:330-357)
  g11_merged       : 2:4, 4-bit SLiM-Quant, rank 0.1 adapters folded back into the weight (separate_lora=False)
  g11_tiled        : (with --tiled, TRITON_INTERPRET=1, slow) weight_tiled_quantization=True, tile 128
Only data is written.  Run:  python tests/golden/make_golden_variants.py [--tiled]
"""
import os
import sys
import argparse

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
ap = argparse.ArgumentParser()
ap.add_argument("--tiled", action="store_true")
args = ap.parse_args()
if args.tiled:
    os.environ["TRITON_INTERPRET"] = "1"
os.chdir(REF)
sys.path.insert(0, REF)

import numpy as np
import torch

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bfloat16:
                out[k + "_bf16"] = bits(v)
            elif v.dtype == torch.bool:
                out[k] = v.numpy().astype(np.uint8)
            else:
                out[k] = v.detach().numpy().copy()
        else:
            out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(out), "arrays")


def main():
    from slim.utils import find_layers, get_layers_list, check_sparsity
    import slim.prune as PRUNE
    from transformers import LlamaConfig, LlamaForCausalLM
    seqlen, vocab = 32, 256

    def fp32_loss_ppl(model, ids):
        n = ids.numel() // seqlen
        nll = 0.0
        for i in range(n):
            inp = ids[:, i * seqlen:(i + 1) * seqlen]
            lg = model(inp).logits.float()
            nll += torch.nn.functional.cross_entropy(lg[0, :-1], inp[0, 1:], reduction="mean").item() * seqlen
        return float(np.exp(nll / (n * seqlen)))

    def run(tag, **kw):
        out = {}
        torch.manual_seed(0)
        model = LlamaForCausalLM(LlamaConfig(vocab_size=vocab, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                             num_attention_heads=4, num_key_value_heads=4,
                                             max_position_embeddings=seqlen)).to(torch.bfloat16)
        model.eval()
        model.config.max_position_embeddings = seqlen
        # same seeds as g8_e2e_llama: its fixture holds the initial weights and the token ids
        g = torch.Generator().manual_seed(1)
        calib = [(torch.randint(0, vocab, (1, seqlen), generator=g), None) for _ in range(8)]
        g = torch.Generator().manual_seed(2)
        evalids = torch.randint(0, vocab, (1, 8 * seqlen), generator=g)
        PRUNE.get_loaders = lambda *a, **k: (calib, None)
        PRUNE.prune_and_quantize(model, None, nsamples=8, **kw)
        layers = get_layers_list(model)
        for i in range(len(layers)):
            for name, lin in find_layers(layers[i]).items():
                p = f"L{i}/{name}/"
                out[p + "weight"] = lin.weight.data
                sf = getattr(lin, "scaling_factor", None)
                if sf is not None:
                    out[p + "sf"] = sf.float()
                if hasattr(lin, "lora_left"):
                    out[p + "lora_left"] = lin.lora_left.data
                    out[p + "lora_right"] = lin.lora_right.data
                    out[p + "lora_rank"] = lin.lora_rank
        with torch.no_grad():
            out["ppl_fp32loss"] = fp32_loss_ppl(model, evalids)
            out["sparsity"] = check_sparsity(model)
        save("g11_" + tag, **out)

    if args.tiled:
        run("tiled", bitwidth=4, slim_quant=False, weight_tiled_quantization=True, weight_tile_size=128, prune_method="wanda",
            sparsity_ratio=0.5, sparsity_type="2:4", quantize_weight=True, lora_rank=0.1, slim_lora=True,
            quantize_lora=False)
        return
    run("unstructured", bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="unstructured",
        quantize_weight=True, lora_rank=0.1, slim_lora=True, quantize_lora=True, lora_tile_size=128)
    run("merged", bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="2:4",
        quantize_weight=True, lora_rank=0.1, slim_lora=True, separate_lora=False)
    run("nolora", bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="2:4",
        quantize_weight=True, lora_rank=0.)


if __name__ == "__main__":
    main()
