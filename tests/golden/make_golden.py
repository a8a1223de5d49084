#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by importing the
reference implementation (read-only at /root/reference) in the development
container.  Only data (inputs and the reference's outputs) is written; no
reference source travels.

Run:  python tests/golden/make_golden.py            (everything except G6)
      TRITON_INTERPRET=1 python tests/golden/make_golden.py --group-quant   (G6, slow)

Harness conditions (all confined to this process):
  * cwd = /root/reference: slim/prune.py evaluates a default argument that opens
    ./data/best_tree.json at import time.
  * the container has no GPU, so Tensor.cuda / Module.cuda are made identity.
  * slim.prune.get_loaders is replaced by synthetic token ids (no network).
bf16 tensors are stored as their uint16 bit patterns (suffix _bf16).
"""
import os
import sys
import argparse

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

ap = argparse.ArgumentParser()
ap.add_argument("--group-quant", action="store_true")
args = ap.parse_args()
if args.group_quant:
    os.environ["TRITON_INTERPRET"] = "1"

os.chdir(REF)
sys.path.insert(0, REF)

import itertools
import numpy as np
import torch

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self


def bits(t):
    assert t.dtype == torch.bfloat16
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
    print("wrote", name, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


def synth_weight(n, k, seed, sigma=0.02, heavy=False):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n, k, generator=g) * sigma
    if heavy:
        m = torch.rand(n, k, generator=g) < 1e-3
        w = torch.where(m, w * 20, w)
    return w.to(torch.bfloat16)


def synth_scaler(k, seed, zeros=0):
    g = torch.Generator().manual_seed(seed)
    s = torch.exp(torch.randn(k, generator=g) * 0.5) ** 2 * 2048.0
    idx = torch.randperm(k, generator=g)[:4]
    s[idx] *= 900.0
    if zeros:
        s[torch.randperm(k, generator=g)[:zeros]] = 0.0
    return s.float()


def main_group_quant():
    from slim.quantization.utils import compute_quantization_params, quantize_tensor, dequantize_tensor
    out = {}
    for tag, (m, n, blk) in {"a": (512, 256, 256), "b": (256, 512, 128), "c": (256, 64, 64)}.items():
        g = torch.Generator().manual_seed(hash(tag) % 1000 + 7)
        x = (torch.randn(m, n, generator=g) * torch.exp(torch.randn(1, n, generator=g))).to(torch.bfloat16)
        # force exact ties at .5 of the scaled value and an all-zero group column
        x[:, 0] = 0
        x[0:blk, 1] = torch.tensor([1.0, 0.5, 0.25, 3.5 / 7, 2.5 / 7, -2.5 / 7, -1.0, 6.5 / 7] * (blk // 8)).to(torch.bfloat16)
        out[f"{tag}_x"] = x
        out[f"{tag}_blk"] = blk
        for q in (4, 8):
            alphas, _ = compute_quantization_params(x, blk, 1, symmetric=True)
            codes = quantize_tensor(x, alphas, None, q)
            deq = dequantize_tensor(codes, alphas, None, q, dtype=x.dtype)
            out[f"{tag}_q{q}_alphas"] = alphas
            out[f"{tag}_q{q}_codes"] = codes
            out[f"{tag}_q{q}_deq"] = deq
    save("g6_group_quant", **out)


def main():
    from slim.layerwrapper import WrappedGPT
    from slim.sparsegpt import SparseGPT
    from slim.utils import shift_zeros, prune_nm, find_layers, get_layers_list
    import slim.quantization.quantization as Q
    import slim.lora as LORA
    import slim.prune as PRUNE
    import slim.eval as EVAL

    # ---- G1: calibration statistics -------------------------------------------------
    g = torch.Generator().manual_seed(1)
    X = (torch.randn(4, 1, 16, 64, generator=g) * torch.exp(torch.randn(64, generator=g) * 0.5)).to(torch.bfloat16)
    lin = torch.nn.Linear(64, 8, bias=False).to(torch.bfloat16)
    wg, sg = WrappedGPT(lin), SparseGPT(lin)
    for j in range(4):
        wg.add_batch(X[j], None)
        sg.add_batch(X[j], None)
    # a second wrapper fed two batches of 2 samples each, exercising tmp > 1
    wg2 = WrappedGPT(lin)
    wg2.add_batch(X[:2, 0], None)
    wg2.add_batch(X[2:, 0], None)
    save("g1_stats", X=X, scaler_row=wg.scaler_row, nsamples=wg.nsamples, inp_sum=wg.inp_sum,
         H=sg.H, scaler_row_b2=wg2.scaler_row, nsamples_b2=wg2.nsamples)

    # ---- G2: shift_zeros ------------------------------------------------------------
    v = torch.tensor([0.0, 3.0, 0.5, 0.0, 7.25, 0.125, 2.0, 0.0])
    v2 = torch.tensor([4.0, 3.0, 0.5, 1.0])
    v3 = torch.zeros(6)
    save("g2_shift_zeros", v=v, out=shift_zeros(v), v2=v2, out2=shift_zeros(v2), v3=v3, out3=shift_zeros(v3))

    # ---- G3: N:M masks ---------------------------------------------------------------
    tups = torch.tensor(list(itertools.product(range(4), repeat=4))).float()  # all weak orderings of 4 values
    out = {"tuples": tups}
    for n in (1, 2, 3):
        out[f"mask_n{n}"] = prune_nm(tups, n, 4)
    g = torch.Generator().manual_seed(3)
    R = torch.rand(64, 256, generator=g)
    out["rand"] = R
    out["rand_mask_2_4"] = prune_nm(R, 2, 4)
    W = synth_weight(96, 512, 11)
    W[5, :64] = 0
    W[6, 8:12] = torch.tensor([0.01, 0.0, 0.0, -0.01]).to(torch.bfloat16)
    sc = synth_scaler(512, 12)
    sal = torch.abs(W) * torch.sqrt(sc.reshape(1, -1))
    out["W"] = W
    out["scaler_row"] = sc
    out["saliency"] = sal
    out["sal_mask_2_4"] = prune_nm(sal, 2, 4)
    out["sal_mask_1_4"] = prune_nm(sal, 1, 4)
    out["sal_mask_3_4"] = prune_nm(sal, 3, 4)
    special = torch.tensor([[float("nan"), 1.0, 2.0, 3.0], [1.0, float("nan"), float("nan"), 0.5],
                            [float("inf"), float("nan"), 1.0, float("inf")], [0.0, -0.0, 1.0, 0.0],
                            [float("nan")] * 4, [-1.0, -2.0, -1.0, 3.0]])
    out["special"] = special
    out["special_mask_2_4"] = prune_nm(special, 2, 4)
    save("g3_prune_nm", **out)

    # ---- G4/G5: SLiM-Quant scale search + per-tensor codes ----------------------------
    out = {}
    rec = {}
    orig_cae = Q.compute_average_error

    def recording_cae(pdf, val, index, q):
        r = orig_cae(pdf, val, index, q)
        rec.setdefault("idx", []).append(int(index))
        rec.setdefault("loss", []).append(float(r))
        rec["pdf"], rec["val"] = pdf, val
        return r

    Q.compute_average_error = recording_cae
    cases = {
        "bins512": synth_weight(256, 512, 21),                    # 131072 elems -> 512 bins
        "bins589": synth_weight(768, 768, 22),                    # OPT-125m attention
        "bins2359": synth_weight(768, 3072, 23),                  # OPT-125m fc
        "heavy": synth_weight(512, 1024, 24, heavy=True),         # heavy tail, 524 bins
        "bins20000": synth_weight(4096, 5120, 25),                # 20000 bins, 10 coarse points
        "uniform_edge": (torch.rand(4096, 5120, generator=torch.Generator().manual_seed(26)) * 2 - 1).to(torch.bfloat16),
        "masked": None,
    }
    wm = synth_weight(512, 512, 27).clone()
    wm[prune_nm(wm.abs().float(), 2, 4)] = 0
    cases["masked"] = wm
    for tag, w in cases.items():
        rec.clear()
        quant = Q.Quantizer("weight", num_bits=4, slim_quant=True)
        codes = quant.quantize_weight(w)
        deq = quant.dequantize_absmax(codes)
        big = w.numel() > 2_000_000
        if not big:
            out[f"{tag}_W"] = w
            out[f"{tag}_codes"] = codes
            out[f"{tag}_deq_bf16cast"] = deq.to(torch.bfloat16)
        else:
            out[f"{tag}_seed_shape"] = np.array([{"bins2359": 23, "bins20000": 25, "uniform_edge": 26}[tag], w.shape[0], w.shape[1]])
            out[f"{tag}_codes_sum"] = codes.long().sum()
            out[f"{tag}_codes_hist"] = torch.bincount((codes.long() + 8).flatten(), minlength=16)
        out[f"{tag}_sf"] = quant.scaling_factor
        out[f"{tag}_alpha"] = (7 / quant.scaling_factor).reshape(())
        out[f"{tag}_eval_idx"] = np.array(rec["idx"])
        out[f"{tag}_eval_loss"] = np.array(rec["loss"], dtype=np.float32)
        if True:
            out[f"{tag}_pdf"] = rec["pdf"]
            out[f"{tag}_val"] = rec["val"]
    Q.compute_average_error = orig_cae
    # absmax (slim_quant=False) and 8-bit variants on one small case
    w = cases["bins512"]
    for nb, sq in ((4, False), (8, True), (8, False)):
        quant = Q.Quantizer("weight", num_bits=nb, slim_quant=sq)
        codes = quant.quantize_weight(w)
        out[f"bins512_b{nb}_sq{int(sq)}_codes"] = codes
        out[f"bins512_b{nb}_sq{int(sq)}_sf"] = quant.scaling_factor
    # explicit rounding edge cases for a fixed scale: ties after the bf16 product, clamp ends
    wt = torch.tensor([[0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 7.5, 8.5, -8.5, -9.0, 100.0, -100.0, 0.0, 6.5, 7.49, -7.51]]).to(torch.bfloat16)
    sf = torch.tensor(1.0)
    codes = torch.clamp(torch.round((wt * sf).float()), -8, 7).to(torch.int8)
    out["ties_W"] = wt
    out["ties_codes_sf1"] = codes
    save("g4_g5_slim_quant", **out)

    # ---- G7: add_lora -----------------------------------------------------------------
    out = {}
    for tag, (n, k, tile, zeros) in {"a": (256, 512, 128, 0), "b": (512, 256, 128, 0), "c": (512, 512, 256, 0),
                                      "naive": (256, 256, 64, 3)}.items():
        w = synth_weight(n, k, 31 + n + k + zeros)
        sc = synth_scaler(k, 32 + n + k, zeros=zeros)
        lin = torch.nn.Linear(k, n, bias=False).to(torch.bfloat16)
        lin.weight.data = w.clone()
        act = WrappedGPT(lin)
        act.scaler_row = shift_zeros(sc) if not zeros else sc.clone()
        sal = torch.abs(w) * torch.sqrt(act.scaler_row.reshape(1, -1))
        mask = prune_nm(sal, 2, 4)
        quant = Q.Quantizer("weight", num_bits=4, slim_quant=True)
        LORA.add_lora(lin, W_mask=mask, rank_ratio=0.1, slim_lora=True, activations=act, quantizer=quant,
                      prune_lora=False, separate_lora=True, lora_tile_size=tile, quantize_first=True)
        out[f"{tag}_W"] = w
        out[f"{tag}_scaler_row"] = act.scaler_row
        out[f"{tag}_mask"] = mask
        out[f"{tag}_tile"] = tile
        out[f"{tag}_new_weight"] = lin.weight.data
        out[f"{tag}_lora_left"] = lin.lora_left.data
        out[f"{tag}_lora_right"] = lin.lora_right.data
        out[f"{tag}_sf2"] = quant.scaling_factor
    save("g7_add_lora", **out)

    # ---- G9: input quantization hooks ------------------------------------------------
    out = {}
    g = torch.Generator().manual_seed(9)
    for tag, scale in (("small", 3.0), ("large", 300.0)):
        x = (torch.randn(2, 8, 128, generator=g) * scale).to(torch.bfloat16)
        for nb, gs in ((8, -1), (4, 64), (6, -1)):
            lin = torch.nn.Linear(128, 16, bias=False).to(torch.bfloat16)

            class M(torch.nn.Module):
                def __init__(s):
                    super().__init__()
                    s.model = torch.nn.Module()
                    blk = torch.nn.Module()
                    blk.lin = lin
                    s.model.layers = torch.nn.ModuleList([blk])
            mdl = M()
            Q.attach_input_quantization_hooks(mdl, num_bits=nb, input_group_size=gs)
            xin = x.clone()
            hook = list(lin._forward_pre_hooks.values())[0]
            hook(lin, (xin,))
            out[f"{tag}_b{nb}_g{gs}"] = xin
        out[f"{tag}_x"] = x
    save("g9_input_quant", **out)

    # ---- G8: end-to-end tiny models ---------------------------------------------------
    from transformers import OPTConfig, OPTForCausalLM, LlamaConfig, LlamaForCausalLM

    class Enc:
        def __init__(self, ids):
            self.input_ids = ids

    def fp32_loss_ppl(model, ids, seqlen):
        n = ids.numel() // seqlen
        nll = 0.0
        for i in range(n):
            inp = ids[:, i * seqlen:(i + 1) * seqlen]
            lg = model(inp).logits.float()
            nll += torch.nn.functional.cross_entropy(lg[0, :-1], inp[0, 1:], reduction="mean").item() * seqlen
        return float(np.exp(nll / (n * seqlen)))

    def run_e2e(tag, make_model, seqlen, vocab):
        out = {}
        torch.manual_seed(0)
        model = make_model().to(torch.bfloat16)
        model.eval()
        model.config.max_position_embeddings = seqlen
        out["state_keys"] = np.array(list(model.state_dict().keys()))
        for k, v in model.state_dict().items():
            out["init/" + k] = v
        g = torch.Generator().manual_seed(1)
        calib = [(torch.randint(0, vocab, (1, seqlen), generator=g), None) for _ in range(8)]
        out["calib_ids"] = torch.cat([c[0] for c in calib], 0)
        g = torch.Generator().manual_seed(2)
        evalids = torch.randint(0, vocab, (1, 8 * seqlen), generator=g)
        out["eval_ids"] = evalids
        PRUNE.get_loaders = lambda *a, **k: (calib, None)
        with torch.no_grad():
            out["ppl_dense"] = EVAL.eval_ppl_wikitext(model, Enc(evalids), 2, "cpu")
            out["ppl_dense_fp32loss"] = fp32_loss_ppl(model, evalids, seqlen)
        PRUNE.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                                 sparsity_type="2:4", quantize_weight=True, nsamples=8, lora_rank=0.1, slim_lora=True,
                                 quantize_lora=True, lora_tile_size=128)
        layers = get_layers_list(model)
        for i in range(len(layers)):
            for name, lin in find_layers(layers[i]).items():
                p = f"L{i}/{name}/"
                out[p + "weight"] = lin.weight.data
                out[p + "sf"] = lin.scaling_factor
                out[p + "lora_left"] = lin.lora_left.data
                out[p + "lora_right"] = lin.lora_right.data
                out[p + "lora_rank"] = lin.lora_rank
        with torch.no_grad():
            out["ppl_slim"] = EVAL.eval_ppl_wikitext(model, Enc(evalids), 2, "cpu")
            out["ppl_slim_fp32loss"] = fp32_loss_ppl(model, evalids, seqlen)
            out["sparsity"] = __import__("slim.utils").utils.check_sparsity(model)
        Q.attach_input_quantization_hooks(model, num_bits=8, input_group_size=-1)
        with torch.no_grad():
            out["ppl_slim_in8"] = EVAL.eval_ppl_wikitext(model, Enc(evalids), 2, "cpu")
            out["ppl_slim_in8_fp32loss"] = fp32_loss_ppl(model, evalids, seqlen)
        save("g8_e2e_" + tag, **out)

    run_e2e("opt", lambda: OPTForCausalLM(OPTConfig(vocab_size=256, hidden_size=256, ffn_dim=512, num_hidden_layers=2,
                                                     num_attention_heads=4, max_position_embeddings=32,
                                                     word_embed_proj_dim=256)), 32, 256)
    run_e2e("llama", lambda: LlamaForCausalLM(LlamaConfig(vocab_size=256, hidden_size=256, intermediate_size=512,
                                                           num_hidden_layers=2, num_attention_heads=4,
                                                           num_key_value_heads=4, max_position_embeddings=32)), 32, 256)

    # ---- dispatcher error matrix -------------------------------------------------------
    errs = []
    combos = [
        dict(sparsity_type="dense", quantize_weight=True, lora_rank=0.1),
        dict(prune_method="magnitude", lora_rank=0.1),
        dict(prune_method="sparsegpt", lora_rank=0.1),
        dict(prune_method="sparsegpt", slim_quant=True),
        dict(prune_method="joint_pq", weight_tiled_quantization=False),
        dict(prune_method="nonsense"),
        dict(sparsity_type="2:4", sparsity_ratio=0.3),
        dict(prune_lora=True, sparsity_type="1:4", sparsity_ratio=0.75),
    ]
    for c in combos:
        kw = dict(bitwidth=4, slim_quant=c.get("slim_quant", False))
        kw.update(c)
        try:
            PRUNE.prune_and_quantize(None, None, **kw)
            errs.append("none")
        except Exception as e:  # noqa
            errs.append(type(e).__name__)
    save("g10_dispatch_errors", combos=np.array([repr(sorted(c.items())) for c in combos]), errors=np.array(errs))


if __name__ == "__main__":
    if args.group_quant:
        main_group_quant()
    else:
        main()
