#!/usr/bin/env python3
"""Golden vectors for scale_important_weights (slim/lora.py:132-138, slim/quantization/quantization.py:133-170,
:192, :215), generated like make_golden.py by importing the reference (read-only at /root/reference) in the
development container.  Only data is written.

  g15_important        : add_lora on single layers with scale_important_weights=True, per-tensor quantisers
                         (SLiM-Quant, abs-max); torch.svd is wrapped so that the error matrix it receives is recorded
  g15_important_tiled  : (--tiled, TRITON_INTERPRET=1, slow) the same with the tiled group quantiser
  g11_important        : the tiny LLaMA model of g8_e2e_llama through prune_and_quantize with
                         scale_important_weights=True (SLiM-Quant, 2:4, rank 0.1)
  g11_important_tiled  : (--tiled) the same model with weight_tiled_quantization=True, tile 128

Run:  python tests/golden/make_golden_important.py [--tiled]
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


def synth_weight(n, k, seed, sigma=0.02):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n, k, generator=g) * sigma
    m = torch.rand(n, k, generator=g) < 1e-3
    return torch.where(m, w * 20, w).to(torch.bfloat16)


def synth_scaler(k, seed):
    g = torch.Generator().manual_seed(seed)
    s = torch.exp(torch.randn(k, generator=g) * 0.5) ** 2 * 2048.0
    s[torch.randperm(k, generator=g)[:4]] *= 30.0
    return s.float()


class Act:
    pass


def layer_cases(tiled):
    import slim.lora as LORA
    import slim.quantization.quantization as Q
    from slim.utils import prune_nm
    out = {}
    recorded = {}
    real_svd = torch.svd

    def svd(x, *a, **k):
        recorded["E"] = x.clone()
        return real_svd(x, *a, **k)
    torch.svd = svd
    if tiled:
        cases = {"t128": (256, 512, dict(slim_quant=False, block_quantization=True, block_dim=128), True),
                 "t128_maskfirst": (256, 256, dict(slim_quant=False, block_quantization=True, block_dim=128), False)}
    else:
        cases = {"slim": (256, 512, dict(slim_quant=True), True),
                 "slim_tall": (512, 256, dict(slim_quant=True), True),
                 "absmax": (256, 512, dict(slim_quant=False), True),
                 "slim_maskfirst": (256, 512, dict(slim_quant=True), False)}
    for tag, (N, K, qkw, quantize_first) in cases.items():
        W = synth_weight(N, K, 11 + len(tag))
        sr = synth_scaler(K, 5 + len(tag))
        lin = torch.nn.Linear(K, N, bias=False).to(torch.bfloat16)
        lin.weight.data = W.clone()
        act = Act()
        act.scaler_row = sr.clone()
        mask = prune_nm(torch.abs(W) * torch.sqrt(sr.reshape(1, -1)), 2, 4)
        metric = sr * W.abs().sum(dim=0)
        important = metric.topk(int(0.01 * metric.numel()), largest=True, sorted=False)[1]
        quantizer = Q.Quantizer("weight", num_bits=4, **qkw)
        # the first quantisation by itself (what add_lora does at lora.py:149-150 / :153-155)
        src = W.clone()
        if not quantize_first:
            src[mask] = 0
        codes1 = quantizer.quantize_weight(src, important)
        assert torch.equal(src, W if quantize_first else W.masked_fill(mask, 0)), "the scaling must be undone in place"
        deq1 = quantizer.dequantize_absmax(codes1)
        sf1 = quantizer.scaling_factor
        LORA.add_lora(lin, W_mask=mask, rank_ratio=0.1, slim_lora=True, activations=act, quantizer=quantizer,
                      prune_lora=False, separate_lora=True, lora_tile_size=None, quantize_first=quantize_first,
                      scale_important_weights=True)
        p = tag + "/"
        out[p + "W"], out[p + "scaler_row"], out[p + "mask"] = W, sr, mask
        out[p + "colsum"] = W.abs().sum(dim=0)
        out[p + "metric"] = metric
        out[p + "important"] = torch.sort(important)[0]
        out[p + "codes1"], out[p + "deq1"] = codes1, deq1
        out[p + "sf1"] = sf1.float() if sf1.shape == (1, 1) else sf1
        out[p + "E"] = recorded["E"]
        out[p + "weight"] = lin.weight.data
        out[p + "lora_left"], out[p + "lora_right"] = lin.lora_left.data, lin.lora_right.data
        sf2 = quantizer.scaling_factor
        out[p + "sf2"] = sf2.float() if sf2.shape == (1, 1) else sf2
        out[p + "quantize_first"] = int(quantize_first)
        print(tag, "important", out[p + "important"].tolist(), "E norm", float(recorded["E"].norm()))
    torch.svd = real_svd
    save("g15_important" + ("_tiled" if tiled else ""), **out)


def model_case(tiled):
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

    out = {}
    torch.manual_seed(0)
    model = LlamaForCausalLM(LlamaConfig(vocab_size=vocab, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                         num_attention_heads=4, num_key_value_heads=4,
                                         max_position_embeddings=seqlen)).to(torch.bfloat16)
    model.eval()
    model.config.max_position_embeddings = seqlen
    g = torch.Generator().manual_seed(1)
    calib = [(torch.randint(0, vocab, (1, seqlen), generator=g), None) for _ in range(8)]
    g = torch.Generator().manual_seed(2)
    evalids = torch.randint(0, vocab, (1, 8 * seqlen), generator=g)
    PRUNE.get_loaders = lambda *a, **k: (calib, None)
    kw = dict(bitwidth=4, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="2:4", quantize_weight=True,
              lora_rank=0.1, slim_lora=True, quantize_lora=False, scale_important_weights=True)
    if tiled:
        kw.update(slim_quant=False, weight_tiled_quantization=True, weight_tile_size=128)
    else:
        kw.update(slim_quant=True)
    PRUNE.prune_and_quantize(model, None, nsamples=8, **kw)
    layers = get_layers_list(model)
    for i in range(len(layers)):
        for name, lin in find_layers(layers[i]).items():
            p = f"L{i}/{name}/"
            out[p + "weight"] = lin.weight.data
            sf = getattr(lin, "scaling_factor", None)
            if sf is not None:
                out[p + "sf"] = sf.float()
            out[p + "lora_left"] = lin.lora_left.data
            out[p + "lora_right"] = lin.lora_right.data
            out[p + "lora_rank"] = lin.lora_rank
    with torch.no_grad():
        out["ppl_fp32loss"] = fp32_loss_ppl(model, evalids)
        out["sparsity"] = check_sparsity(model)
    save("g11_important" + ("_tiled" if tiled else ""), **out)


if __name__ == "__main__":
    layer_cases(args.tiled)
    model_case(args.tiled)
