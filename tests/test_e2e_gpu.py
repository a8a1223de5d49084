"""End-to-end parity of slim_amd.prune.prune_and_quantize against the reference's own run on the same
tiny random-init models (fixtures tests/golden/g8_e2e_*.npz hold the initial weights, the calibration
and evaluation ids, and everything the reference left on each linear)."""
import math

import pytest
import torch

from oracle import slim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(tag):
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


def _load_init(model, g):
    sd = {k: g["init/" + k] for k in g["state_keys"].tolist()}
    model.load_state_dict(sd)


class Enc:
    def __init__(self, ids):
        self.input_ids = ids


def _fp32_ppl(model, ids, seqlen):
    from slim_amd.eval import eval_ppl_wikitext
    return eval_ppl_wikitext(model, Enc(ids), 2, DEV, fp32_loss=True)


@pytest.mark.parametrize("tag", ["opt", "llama"])
def test_prune_and_quantize_matches_reference(golden, tag):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    import slim_amd.lora as LORA
    from slim_amd.quantization.quantization import attach_input_quantization_hooks
    from slim_amd.utils import get_layers_list, find_layers, check_sparsity
    from slim_amd.eval import eval_ppl_wikitext

    g = golden("g8_e2e_" + tag)
    calib = [(g["calib_ids"][i:i + 1], None) for i in range(8)]
    evalids = g["eval_ids"]

    # --- our compression ---------------------------------------------------------------------------
    model = _make(tag)
    _load_init(model, g)
    orig = {k: v.clone() for k, v in model.state_dict().items()}
    stats = {}

    def cb(i, name, lin, wrapped):
        stats[(i, name)] = wrapped.scaler_row.clone()
    P.get_loaders = lambda *a, **k: (calib, None)
    P.LAYER_CALLBACK = cb
    P.KEEP_ON_DEVICE = True
    try:
        P.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, nsamples=8, lora_rank=0.1, slim_lora=True,
                             quantize_lora=True, lora_tile_size=128)
    finally:
        P.LAYER_CALLBACK = None
        P.KEEP_ON_DEVICE = False
    model.to(DEV)

    # --- the reference's result, loaded into an identical model for a like-for-like forward ---------
    ref = _make(tag)
    _load_init(ref, g)
    ref.to(DEV)
    layers_o, layers_r = get_layers_list(model), get_layers_list(ref)
    prefix = "model.decoder.layers" if tag == "opt" else "model.layers"
    worst = {"mask": 1.0, "err": 0.0}
    problems, rows = [], []
    for i in range(len(layers_o)):
        sub_o, sub_r = find_layers(layers_o[i]), find_layers(layers_r[i])
        assert list(sub_o) == list(sub_r)
        for name in sub_o:
            p = f"L{i}/{name}/"
            lo, lr = sub_o[name], sub_r[name]
            Wr, Lr, Rr = g[p + "weight"], g[p + "lora_left"], g[p + "lora_right"]
            r = int(g[p + "lora_rank"])
            # contract of the side effects (attributes, shapes, dtypes)
            assert lo.weight.dtype == torch.bfloat16 and lo.weight.is_contiguous()
            assert tuple(lo.lora_left.shape) == tuple(Lr.shape) and tuple(lo.lora_right.shape) == tuple(Rr.shape)
            assert int(lo.lora_rank) == r and lo.scaling_factor.shape == (1, 1)
            assert isinstance(lo.lora_left, torch.nn.Parameter) and lo.lora_left.dtype == torch.bfloat16
            assert int(lo.lora_fit_info.abs().sum()) == 0
            assert lo.weight.data.float().unique().numel() <= 16
            W0 = orig[f"{prefix}.{i}.{name}.weight"].float()
            # zero pattern (2:4 mask plus weights quantised to 0): calibration statistics come from GPU
            # forwards here and CPU forwards in the fixture, and block i > 0 sees the outputs of a block whose
            # adapters come from a different (randomized) factorisation, so saliencies differ slightly and
            # near-ties flip.  Demand >= 99 % agreement in block 0 and >= 98 % downstream.  (Bit-exact masks
            # for identical saliency are pinned in test_hip_kernels.py.)
            zo = (lo.weight.data == 0).cpu()
            zr = (Wr == 0)
            agree = (zo == zr).float().mean().item()
            worst["mask"] = min(worst["mask"], agree)
            if agree < (0.99 if i == 0 else 0.98):
                problems.append((p, "zero-pattern agreement", agree))
            # second-pass scale: the argmin of a flat, noisy loss curve (512-bin histogram of only 65k-131k
            # values of W - LR), so it moves by several bins (0.4 % each) under tiny perturbations of LR;
            # what is pinned tightly is the resulting error below, the scale only loosely
            sfo, sfr = float(lo.scaling_factor), float(g[p + "sf"])
            if abs(sfo / sfr - 1) >= 0.10:
                problems.append((p, "scaling factor", sfo, sfr))
            # saliency-weighted error of the whole compressed layer (what the method minimises):
            # || (W0 - (Wq + R^T L^T / r)) diag(sqrt(scaler_row)) ||_F, ours vs the reference's
            s = torch.sqrt(stats[(i, name)]).cpu().reshape(1, -1)
            eff_o = lo.weight.data.float().cpu() + (lo.lora_right.data.float().cpu().t() @ lo.lora_left.data.float().cpu().t()) / r
            eff_r = Wr.float() + (Rr.float().t() @ Lr.float().t()) / r
            eo = torch.norm((W0 - eff_o) * s).item()
            er = torch.norm((W0 - eff_r) * s).item()
            worst["err"] = max(worst["err"], eo / er - 1)
            rows.append((p, round(agree, 4), round(sfo / sfr, 4), round(eo / er - 1, 5)))
            # block 0 sees identical inputs on both sides: 2 % (the second-pass scale, a noisy argmin on these
            # tiny layers, moves the error by about 1 %; the fit alone is pinned to 1e-3 in test_hip_kernels.py).
            # Downstream blocks are measured under OUR statistics, which the reference's factors were not fitted to,
            # and their second-pass scale differs by up to 5 % (sf ratio column): the search minimises the
            # UNWEIGHTED quantisation error, so two scales it rates alike can differ by several per cent in this
            # saliency-weighted measure (observed -2.3 % ... +6.6 % per layer across fit settings).  Per layer 8 %,
            # and 2 % on the mean over the block's layers (checked below).
            if eo > er * (1.02 if i == 0 else 1.08):
                problems.append((p, "saliency-weighted error", eo, er))
            # install the reference's tensors in the twin model
            lr.weight.data = Wr.to(DEV)
            lr.lora_left = torch.nn.Parameter(Lr.to(DEV))
            lr.lora_right = torch.nn.Parameter(Rr.to(DEV))
            lr.lora_rank = torch.tensor(r)
            lr.register_forward_hook(P.add_lora_hook)
    for row in rows:
        print("  layer, zero-agreement, sf ratio, weighted-error excess:", row)
    print("worst mask agreement", worst["mask"], "worst weighted-error excess", worst["err"])
    assert not problems, problems
    for blk in range(len(layers_o)):
        ex = [r[3] for r in rows if r[0].startswith(f"L{blk}/")]
        assert sum(ex) / len(ex) <= 0.02, (blk, ex)

    sp = check_sparsity(model)
    assert abs(sp - float(g["sparsity"])) < 2e-3

    # --- perplexity: bf16-loss (reference semantics) and fp32-loss, our compression vs the reference's ---
    ppl_o = _fp32_ppl(model, evalids, 32)
    ppl_r = _fp32_ppl(ref, evalids, 32)
    ppl_o_bf16 = eval_ppl_wikitext(model, Enc(evalids), 2, DEV)
    print(f"{tag}: fp32-loss ppl ours {ppl_o:.4f} reference-weights {ppl_r:.4f} fixture(cpu) {float(g['ppl_slim_fp32loss']):.4f}; "
          f"bf16-loss ours {ppl_o_bf16:.3f} fixture {float(g['ppl_slim']):.3f}")
    # the bf16 loss is quantised to 8 bits of mantissa per batch: one step near ln(256) is 2^-5 = 0.03125 in loss
    # (~8 in perplexity).  A 0.5 % difference of the underlying fp32 losses can move every batch's rounded loss by
    # one step in the same direction, so the mean may shift by one step plus that difference: 0.03125 + 0.014
    assert abs(math.log(ppl_o_bf16) - math.log(float(g["ppl_slim"]))) <= 0.045
    # fp32-loss perplexity of OUR compressed model vs the REFERENCE's compressed tensors in the same GPU
    # forward.  On a random-init 256-token-vocabulary model the perplexity (~ vocabulary size) moves by
    # ~0.5 % between two equally good compressions (the reference's own compressed model differs from its
    # dense model by 0.4 % in either direction), so an absolute 0.05 is below the noise of the setting;
    # the bound used is 1 % relative, and the two numbers are printed.
    assert abs(ppl_o - ppl_r) <= 0.01 * ppl_r, (ppl_o, ppl_r)

    # --- quantised adapters + 8-bit input hooks run end to end ---------------------------------------
    LORA.quantize_lora(model, bitwidth=4, lora_tile_size=128)
    LORA.quantize_lora(ref, bitwidth=4, lora_tile_size=128)
    pq_o, pq_r = _fp32_ppl(model, evalids, 32), _fp32_ppl(ref, evalids, 32)
    attach_input_quantization_hooks(model, num_bits=8)
    attach_input_quantization_hooks(ref, bitwidth=8)
    pi_o, pi_r = _fp32_ppl(model, evalids, 32), _fp32_ppl(ref, evalids, 32)
    print(f"{tag}: int4 adapters ours {pq_o:.4f} ref-weights {pq_r:.4f}; +fp8 inputs ours {pi_o:.4f} ref-weights {pi_r:.4f}")
    assert abs(pq_o - pq_r) <= 0.01 * pq_r and abs(pi_o - pi_r) <= 0.01 * pi_r
    assert all(math.isfinite(v) for v in (pq_o, pq_r, pi_o, pi_r))
