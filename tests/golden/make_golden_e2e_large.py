#!/usr/bin/env python3
"""A larger end-to-end fixture for the perplexity comparison, generated like make_golden.py by importing the reference
(read-only at /root/reference) in the development container: a 4-block LLaMA-shaped model (hidden 512, MLP 1024, 8
heads, vocabulary 512, sequences of 64 tokens) through the reference's prune_and_quantize with the headline recipe
(wanda 2:4, 4-bit SLiM-Quant, rank 0.1 adapters padded to a tile of 64), 16 calibration samples, and an evaluation
stream of 64 windows (4096 tokens: sixteen times the stream of g8_e2e_*, so the perplexity is a much smoother function
of the compressed tensors).

To keep the file small the initial weights are NOT stored: the model is built from torch.manual_seed(0) by the same
constructor call on both sides and the fixture holds a SHA-256 of its state dict, which the test checks.  The reference's
compressed weights are stored as their int8 codes and scale (weight == bf16(codes / scale), verified here), the adapters
as bf16 bit patterns.  Every linear's scaler_row (after shift_zeros) is recorded so that the test can inject the
reference's statistics.

Run:  python tests/golden/make_golden_e2e_large.py
"""
import hashlib
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

SEQ, VOCAB, NCAL, NEVAL = 64, 512, 16, 64


def build_model():
    """Same call in tests/test_zy_perplexity_gpu.py."""
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    m = LlamaForCausalLM(LlamaConfig(vocab_size=VOCAB, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                                     num_attention_heads=8, num_key_value_heads=8, max_position_embeddings=SEQ))
    m = m.to(torch.bfloat16).eval()
    m.config.max_position_embeddings = SEQ
    return m


def state_hash(model):
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(v.contiguous().view(torch.int16).numpy().tobytes() if v.dtype == torch.bfloat16 else v.contiguous().numpy().tobytes())
    return h.hexdigest()


def token_ids():
    g = torch.Generator().manual_seed(1)
    calib = torch.cat([torch.randint(0, VOCAB, (1, SEQ), generator=g) for _ in range(NCAL)], 0)
    g = torch.Generator().manual_seed(2)
    return calib, torch.randint(0, VOCAB, (1, NEVAL * SEQ), generator=g)


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    from slim.utils import find_layers, get_layers_list, check_sparsity, shift_zeros
    import slim.prune as PRUNE
    import slim.eval as EVAL
    import slim.layerwrapper as LW

    class Enc:
        def __init__(self, ids):
            self.input_ids = ids

    def fp32_loss_ppl(model, ids):
        n = ids.numel() // SEQ
        nll = 0.0
        for i in range(n):
            inp = ids[:, i * SEQ:(i + 1) * SEQ]
            lg = model(inp).logits.float()
            nll += torch.nn.functional.cross_entropy(lg[0, :-1], inp[0, 1:], reduction="mean").item() * SEQ
        return float(np.exp(nll / (n * SEQ)))

    out = {}
    model = build_model()
    out["state_sha256"] = np.array(state_hash(model))
    calib, evalids = token_ids()
    out["calib_ids"], out["eval_ids"] = calib.numpy(), evalids.numpy()
    PRUNE.get_loaders = lambda *a, **k: ([(calib[i:i + 1], None) for i in range(NCAL)], None)
    with torch.no_grad():
        out["ppl_dense"] = EVAL.eval_ppl_wikitext(model, Enc(evalids), 4, "cpu")
        out["ppl_dense_fp32loss"] = fp32_loss_ppl(model, evalids)
    # record every wrapper's statistics as add_lora sees them (after shift_zeros): add_lora is wrapped
    import slim.lora as LORA
    real = LORA.add_lora
    order = []

    def add_lora(module, **kw):
        order.append(kw["activations"].scaler_row.clone())
        return real(module, **kw)
    PRUNE.add_lora = add_lora
    PRUNE.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, nsamples=NCAL, lora_rank=0.1, slim_lora=True,
                             quantize_lora=True, lora_tile_size=64)
    PRUNE.add_lora = real
    layers = get_layers_list(model)
    j = 0
    for i in range(len(layers)):
        for name, lin in find_layers(layers[i]).items():
            p = f"L{i}/{name}/"
            sf = lin.scaling_factor.float()
            codes = torch.round(lin.weight.data.float() * sf).to(torch.int8)
            assert torch.equal((codes / sf).to(torch.bfloat16), lin.weight.data), p
            out[p + "codes"] = codes.numpy()
            out[p + "sf"] = sf.numpy()
            out[p + "lora_left_bf16"] = bits(lin.lora_left.data)
            out[p + "lora_right_bf16"] = bits(lin.lora_right.data)
            out[p + "lora_rank"] = lin.lora_rank.numpy()
            out[p + "scaler_row"] = order[j].numpy()
            j += 1
    assert j == len(order)
    with torch.no_grad():
        out["ppl_slim"] = EVAL.eval_ppl_wikitext(model, Enc(evalids), 4, "cpu")
        out["ppl_slim_fp32loss"] = fp32_loss_ppl(model, evalids)
        out["sparsity"] = check_sparsity(model)
    np.savez_compressed(os.path.join(HERE, "g16_e2e_large.npz"), **out)
    print({k: out[k] for k in ("ppl_dense", "ppl_dense_fp32loss", "ppl_slim", "ppl_slim_fp32loss", "sparsity")})
    print("bytes", os.path.getsize(os.path.join(HERE, "g16_e2e_large.npz")))


if __name__ == "__main__":
    main()
