#!/usr/bin/env python3
"""Reference-side state of the two helpers around the adapter fine-tune (slim/fine_tune.py: disable_linear_layer_grads
:24-48, requantize :51-67) on a toy module with the attributes the compression leaves: a linear with adapters, one with
pruned adapters (lora_left_mask), one without adapters but with a scaling factor, one plain linear with a bias.
Recorded: requires_grad flags and dtypes after disable_linear_layer_grads, the tensors after one forward (the masking
pre-hooks have run on deliberately dirtied values), the weights after requantize.  Written to
tests/golden/g14_finetune.npz; only data travels.

Run:  python tests/golden/make_golden_finetune.py

Harness conditions: cwd = /root/reference (an import-time default there reads a relative path), CPU tensors.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.chdir(REF)
sys.path.insert(0, REF)

import numpy as np
import torch


def bf16_bits(t):
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def build(seed=0):
    """Toy model (same construction in tests/test_finetune.py)."""
    g = torch.Generator().manual_seed(seed)
    m = torch.nn.Sequential()
    sf = torch.tensor([[37.5]], dtype=torch.float32)

    def levels(n, k):
        w = torch.randint(-8, 8, (n, k), generator=g).float()
        w[torch.rand(n, k, generator=g) < 0.5] = 0
        return (w / sf).to(torch.bfloat16)

    a = torch.nn.Linear(16, 24, bias=False)
    a.weight.data = levels(24, 16)
    a.lora_left = torch.nn.Parameter((torch.randn(16, 8, generator=g) * 0.1).to(torch.bfloat16))
    a.lora_right = torch.nn.Parameter((torch.randn(8, 24, generator=g) * 0.1).to(torch.bfloat16))
    a.scaling_factor = sf.clone()
    b = torch.nn.Linear(24, 16, bias=True)
    b.weight.data = levels(16, 24)
    b.bias.data = (torch.randn(16, generator=g) * 0.1).to(torch.bfloat16)
    b.lora_left = torch.nn.Parameter((torch.randn(24, 8, generator=g) * 0.1).to(torch.bfloat16))
    b.lora_right = torch.nn.Parameter((torch.randn(8, 16, generator=g) * 0.1).to(torch.bfloat16))
    b.lora_left_mask = torch.rand(24, 8, generator=g) < 0.5
    b.scaling_factor = sf.clone()
    c = torch.nn.Linear(16, 16, bias=False)
    c.weight.data = levels(16, 16)
    c.scaling_factor = sf.clone()
    d = torch.nn.Linear(16, 8, bias=True)
    d.weight.data = (torch.randn(8, 16, generator=g) * 0.1).to(torch.bfloat16)
    d.weight.data[torch.rand(8, 16, generator=g) < 0.3] = 0
    d.bias.data = d.bias.data.to(torch.bfloat16)
    # no adapters, abs-max style scale: a bf16 (1, 1) tensor, so that `weight * scaling_factor` stays bf16 (requantize then
    # rounds the product to bf16 before it rounds to an integer); values up to 7 / sf, where bf16 spacing is 2^-5 after scaling
    e = torch.nn.Linear(16, 16, bias=False)
    sf16 = torch.tensor([[37.5]], dtype=torch.bfloat16)
    we = torch.randint(-7, 8, (16, 16), generator=g).float()
    we[torch.rand(16, 16, generator=g) < 0.5] = 0
    e.weight.data = (we / sf16.float()).to(torch.bfloat16)
    e.scaling_factor = sf16
    for name, mod in zip("abcde", (a, b, c, d, e)):
        m.add_module(name, mod)
    return m, g


def main():
    from slim.fine_tune import disable_linear_layer_grads, requantize
    m, g = build()
    out = {}
    disable_linear_layer_grads(m)
    for name, mod in m.named_children():
        out[f"{name}/weight_requires_grad"] = np.array(mod.weight.requires_grad)
        out[f"{name}/weight_dtype"] = np.array(str(mod.weight.dtype))
        if mod.bias is not None:
            out[f"{name}/bias_requires_grad"] = np.array(mod.bias.requires_grad)
            out[f"{name}/bias_dtype"] = np.array(str(mod.bias.dtype))
        out[f"{name}/has_weight_mask"] = np.array(hasattr(mod, "weight_mask"))
        if hasattr(mod, "weight_mask"):
            out[f"{name}/weight_mask"] = mod.weight_mask.numpy()
        out[f"{name}/n_pre_hooks"] = np.array(len(mod._forward_pre_hooks))
    # dirty what the masking hooks guard (a training step would), then run the pre-hooks the way a forward does
    drift = {}
    for name, mod in m.named_children():
        # (module e: drifts of the order of half a level, so that products land near the half-way points)
        drift[name] = (torch.randn(mod.weight.shape, generator=g) * (0.013 if name == "e" else 0.01))
        if not hasattr(mod, "lora_left"):
            mod.weight.data = (mod.weight.data.float() + drift[name]).to(mod.weight.dtype)
        out[f"{name}/drift"] = drift[name].numpy()
    m.b.lora_left.data = m.b.lora_left.data + 1.0
    for name, mod in m.named_children():
        for hook in mod._forward_pre_hooks.values():
            hook(mod, (None,))
        out[f"{name}/weight_after_hooks_bf16"] = bf16_bits(mod.weight.data.float())
    out["b/lora_left_after_hooks_bf16"] = bf16_bits(m.b.lora_left.data)
    requantize(m)
    for name, mod in m.named_children():
        out[f"{name}/weight_after_requantize_bf16"] = bf16_bits(mod.weight.data.float())
        out[f"{name}/weight_dtype_after_requantize"] = np.array(str(mod.weight.dtype))
    # group quantisation (scaling_factor None) is refused by both helpers
    e = torch.nn.Sequential()
    lin = torch.nn.Linear(8, 8, bias=False)
    lin.scaling_factor = None
    e.add_module("x", lin)
    for fn, key in ((disable_linear_layer_grads, "err_disable"), (requantize, "err_requantize")):
        try:
            fn(e)
            out[key] = np.array("")
        except Exception as ex:   # noqa: BLE001
            out[key] = np.array(f"{type(ex).__name__}: {ex}")
    np.savez_compressed(os.path.join(HERE, "g14_finetune.npz"), **out)
    print({k: (v.shape if v.ndim else v.item()) for k, v in out.items() if "bf16" not in k and "drift" not in k and "mask" not in k[-4:]})


if __name__ == "__main__":
    main()
