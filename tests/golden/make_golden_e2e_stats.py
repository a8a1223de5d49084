#!/usr/bin/env python3
"""Reference-side per-linear state during the end-to-end runs of tests/golden/g8_e2e_*.npz: for every
linear of every decoder block the statistics add_lora received (the shifted scaler_row and its square
root as this host's torch.sqrt evaluates it), the 2:4 mask and the first-pass quantisation (codes +
scaling factor) of the reference, recorded by wrapping the reference's add_lora while the reference
runs on the g8 models (same initial weights, same calibration ids).  Written to
tests/golden/g13_e2e_stats_{opt,llama}.npz (keys "L<block>/<linear>/..."); only data travels.

Run:  python tests/golden/make_golden_e2e_stats.py

Harness conditions are those of make_golden.py (cwd = /root/reference, identity .cuda(), synthetic
calibration ids instead of the C4 download).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.chdir(REF)
sys.path.insert(0, REF)

import numpy as np
import torch

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


def load_g8(tag):
    d = np.load(os.path.join(HERE, f"g8_e2e_{tag}.npz"), allow_pickle=False)

    def get(k):
        if k + "_bf16" in d.files:
            return torch.from_numpy(d[k + "_bf16"].view(np.int16).copy()).view(torch.bfloat16)
        return torch.from_numpy(np.ascontiguousarray(d[k]))
    return d, get


def make(tag):
    from transformers import OPTConfig, OPTForCausalLM, LlamaConfig, LlamaForCausalLM
    if tag == "opt":
        m = OPTForCausalLM(OPTConfig(vocab_size=256, hidden_size=256, ffn_dim=512, num_hidden_layers=2,
                                     num_attention_heads=4, max_position_embeddings=32, word_embed_proj_dim=256))
    else:
        m = LlamaForCausalLM(LlamaConfig(vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                         num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=32))
    m = m.to(torch.bfloat16).eval()
    m.config.max_position_embeddings = 32
    return m


def main():
    import slim.prune as PRUNE
    import slim.quantization.quantization as Q
    from slim.utils import find_layers, get_layers_list

    for tag in ("opt", "llama"):
        d, get = load_g8(tag)
        model = make(tag)
        model.load_state_dict({k: get("init/" + k) for k in d["state_keys"].tolist()})
        calib = [(get("calib_ids")[i:i + 1], None) for i in range(8)]
        PRUNE.get_loaders = lambda *a, **k: (calib, None)
        name_of = {}
        for i, layer in enumerate(get_layers_list(model)):
            for name, lin in find_layers(layer).items():
                name_of[id(lin)] = f"L{i}/{name}"
        out = {}
        real_add_lora = PRUNE.add_lora

        def recording_add_lora(module, W_mask=None, activations=None, quantizer=None, **kw):
            name = name_of.get(id(module))
            if name is not None and (name + "/mask") not in out:
                W = module.weight.data.clone()
                out[name + "/scaler_row"] = activations.scaler_row.clone().numpy()
                out[name + "/s"] = torch.sqrt(activations.scaler_row).numpy()          # slim/prune.py:264-265
                out[name + "/mask"] = np.packbits(W_mask.numpy().astype(np.uint8), axis=1)
                # the first pass of add_lora (slim/lora.py:149) on the weight as it stands, with a quantizer of its own
                q1 = Q.Quantizer("weight", num_bits=quantizer.num_bits, slim_quant=quantizer.slim_quant)
                out[name + "/codes1"] = q1.quantize_weight(W).numpy().astype(np.int8)
                out[name + "/sf1"] = q1.scaling_factor.numpy().astype(np.float32)
            return real_add_lora(module, W_mask=W_mask, activations=activations, quantizer=quantizer, **kw)

        PRUNE.add_lora = recording_add_lora
        try:
            PRUNE.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                                     sparsity_type="2:4", quantize_weight=True, nsamples=8, lora_rank=0.1, slim_lora=True,
                                     quantize_lora=True, lora_tile_size=128)
        finally:
            PRUNE.add_lora = real_add_lora
        # the run must be the one the g8 fixture recorded
        layers = get_layers_list(model)
        for i, layer in enumerate(layers):
            for name, lin in find_layers(layer).items():
                assert torch.equal(lin.weight.data, get(f"L{i}/{name}/weight")), name
                assert torch.equal(lin.lora_left.data, get(f"L{i}/{name}/lora_left")), name
        out["names"] = np.array(list(find_layers(layers[0])))
        np.savez_compressed(os.path.join(HERE, f"g13_e2e_stats_{tag}.npz"), **out)
        print("wrote g13_e2e_stats_" + tag, sorted(out)[:8], "...")


if __name__ == "__main__":
    main()
