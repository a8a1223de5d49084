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
            assert int(lo.lora_fit_info[:-3].abs().sum()) == 0      # (the last three words are reports of what the fit found, not a status)
            assert lo.weight.data.float().unique().numel() <= 16
            W0 = orig[f"{prefix}.{i}.{name}.weight"].float()
            # zero pattern (2:4 mask plus weights quantised to 0): calibration statistics come from GPU
            # forwards here and CPU forwards in the fixture, and block i > 0 sees the outputs of a block whose
            # adapters come from a different (randomized) factorisation, so saliencies differ slightly and
            # near-ties flip.  Demand >= 99 % agreement in block 0 and >= 97.5 % downstream (measured 97.9-99.3 % there
            # across versions of the fit, all within 1e-3 of the optimal reconstruction).  (Bit-exact masks for identical
            # saliency are pinned in test_hip_kernels.py.)
            zo = (lo.weight.data == 0).cpu()
            zr = (Wr == 0)
            agree = (zo == zr).float().mean().item()
            worst["mask"] = min(worst["mask"], agree)
            if agree < (0.99 if i == 0 else 0.975):
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
    # forward: the product path, statistics from GPU calibration forwards.  test_perplexity_delta_by_cause separates
    # the causes and explains the bound.
    assert abs(ppl_o - ppl_r) <= PPL_BOUND_C[tag], (ppl_o, ppl_r)

    # --- quantised adapters + 8-bit input hooks run end to end ---------------------------------------
    LORA.quantize_lora(model, bitwidth=4, lora_tile_size=128)
    LORA.quantize_lora(ref, bitwidth=4, lora_tile_size=128)
    pq_o, pq_r = _fp32_ppl(model, evalids, 32), _fp32_ppl(ref, evalids, 32)
    attach_input_quantization_hooks(model, num_bits=8)
    attach_input_quantization_hooks(ref, bitwidth=8)
    pi_o, pi_r = _fp32_ppl(model, evalids, 32), _fp32_ppl(ref, evalids, 32)
    print(f"{tag}: int4 adapters ours {pq_o:.4f} ref-weights {pq_r:.4f}; +fp8 inputs ours {pi_o:.4f} ref-weights {pi_r:.4f}")
    assert abs(pq_o - pq_r) <= PPL_BOUND_C_Q4[tag] and abs(pi_o - pi_r) <= PPL_BOUND_C_Q4_IN8[tag]
    assert all(math.isfinite(v) for v in (pq_o, pq_r, pi_o, pi_r))


def _unpack_mask(g13, p, K):
    import numpy as np
    return torch.from_numpy(np.unpackbits(g13[p + "mask"].numpy(), axis=1)[:, :K]).bool()


@pytest.mark.parametrize("tag", ["opt", "llama"])
def test_masks_and_first_pass_codes_bit_exact_on_reference_statistics(golden, tag):
    """Every linear of the reference's end-to-end runs, fed the saliency scale s = sqrt(scaler_row) and the weight the
    reference had: the 2:4 mask, the first-pass scale (same bin of the search, same fp32 scaling factor) and the
    first-pass integer codes are bit-identical to what the reference computed (fixtures g13_e2e_stats_*)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops
    g8, g13 = golden("g8_e2e_" + tag), golden("g13_e2e_stats_" + tag)
    prefix = "model.decoder.layers" if tag == "opt" else "model.layers"
    for i in range(2):
        for name in g13["names"].tolist():
            p = f"L{i}/{name}/"
            W = g8[f"init/{prefix}.{i}.{name}.weight"].to(DEV)
            mask = ops.mask_nm(W, g13[p + "s"].to(DEV), 2, 4)
            assert torch.equal(mask.cpu(), _unpack_mask(g13, p, W.shape[1])), p
            _, _, sf, _ = ops.quant_search(ops.hist_abs(W), O.num_bins_for(W.numel()), 4)
            assert float(sf) == float(g13[p + "sf1"].reshape(-1)[0]), p
            codes = ops.quantize_pt(W, sf.reshape(1), 4)
            assert torch.equal(codes.cpu(), g13[p + "codes1"]), p


def _exact_svd_fit(E, Et, rank, s, outliers=None, spread=1.0, hint=0, floor=0):
    """Test-only stand-in for slim_amd.lora.fit_error_matrix: the factors of slim/lora.py:172-195 from a full SVD
    (torch, on the GPU), with the reference's bf16 rounding sequence."""
    U, S, V = torch.svd(E.float())
    r8 = -(-rank // 8) * 8
    L = (torch.diag_embed(S[:rank]).bfloat16() @ V[:, :rank].bfloat16().T).t().contiguous()       # [K, r]
    R = U[:, :rank].bfloat16().t().contiguous()                                                    # [r, N]
    if s is not None:
        L = (L / s.reshape(-1, 1).to(torch.bfloat16)).contiguous()
    Up = torch.zeros(E.shape[0], r8, dtype=torch.bfloat16, device=E.device)
    Lp = torch.zeros(E.shape[1], r8, dtype=torch.bfloat16, device=E.device)
    Up[:, :rank], Lp[:, :rank] = R.t(), L
    return Up, Up.t().contiguous(), Lp, torch.zeros(64, dtype=torch.int32, device=E.device)


def _compress(golden, tag, inject, exact_svd):
    import slim_amd.prune as P
    import slim_amd.lora as LORA
    g, g13 = golden("g8_e2e_" + tag), golden("g13_e2e_stats_" + tag)
    calib = [(g["calib_ids"][i:i + 1], None) for i in range(8)]
    model = _make(tag)
    _load_init(model, g)

    def stats_cb(i, name, wrapped):
        wrapped.scaler_row = g13[f"L{i}/{name}/scaler_row"].to(DEV)
    saved = (P.get_loaders, P.STATS_CALLBACK, P.KEEP_ON_DEVICE, P.SHARED_EIGH, LORA.fit_error_matrix)
    P.get_loaders = lambda *a, **k: (calib, None)
    P.STATS_CALLBACK = stats_cb if inject else None
    P.KEEP_ON_DEVICE = True
    if exact_svd:
        P.SHARED_EIGH = False           # the one-fit-per-linear path, which goes through fit_error_matrix
        LORA.fit_error_matrix = _exact_svd_fit
    try:
        P.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, nsamples=8, lora_rank=0.1, slim_lora=True,
                             quantize_lora=True, lora_tile_size=128)
    finally:
        P.get_loaders, P.STATS_CALLBACK, P.KEEP_ON_DEVICE, P.SHARED_EIGH, LORA.fit_error_matrix = saved
    return model.to(DEV)


@pytest.mark.parametrize("tag", ["opt", "llama"])
def test_perplexity_delta_by_cause(golden, tag):
    """Where the end-to-end perplexity difference to the reference comes from, one cause at a time.  All numbers are
    fp32-loss perplexities from the same GPU forward; `ref` is a model carrying the reference's compressed tensors.
      A  reference statistics injected + exact SVD factors: the pipeline itself (mask, searches, codes, residual)
      B  reference statistics injected + the randomized fit: adds the factorisation
      C  own statistics (GPU calibration forwards) + the randomized fit: the product path, adds the calibration device
    """
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    from slim_amd.utils import get_layers_list, find_layers
    g, g13 = golden("g8_e2e_" + tag), golden("g13_e2e_stats_" + tag)
    evalids = g["eval_ids"]
    ref = _make(tag)
    _load_init(ref, g)
    ref.to(DEV)
    for i, layer in enumerate(get_layers_list(ref)):
        for name, lin in find_layers(layer).items():
            p = f"L{i}/{name}/"
            lin.weight.data = g[p + "weight"].to(DEV)
            lin.lora_left = torch.nn.Parameter(g[p + "lora_left"].to(DEV))
            lin.lora_right = torch.nn.Parameter(g[p + "lora_right"].to(DEV))
            lin.lora_rank = torch.tensor(int(g[p + "lora_rank"]))
            lin.register_forward_hook(P.add_lora_hook)
    ppl_ref = _fp32_ppl(ref, evalids, 32)
    out = {}
    for key, inject, exact in (("A", True, True), ("B", True, False), ("C", False, False)):
        model = _compress(golden, tag, inject, exact)
        out[key] = _fp32_ppl(model, evalids, 32)
        if inject:
            # identical statistics: the zero pattern of the final weight (mask + weights quantised to zero) may differ
            # from the reference's only where sqrt differs in the last place between this library (correctly rounded)
            # and the reference host's vectorised torch.sqrt, or where the second-pass scale moved a value across +-0.5
            agree = []
            for i, layer in enumerate(get_layers_list(model)):
                for name, lin in find_layers(layer).items():
                    agree.append(((lin.weight.data == 0).cpu() == (g[f"L{i}/{name}/weight"] == 0)).float().mean().item())
            out[key + "_zero_agreement"] = min(agree)
        if key == "B":
            model_b = model
        del model
    # quantised adapters and 8-bit input hooks on top of B (slim/lora.py:56-101, quantization.py:266-304): both models get
    # the same treatment, so the difference is still the factorisation's
    import slim_amd.lora as LORA
    from slim_amd.quantization.quantization import attach_input_quantization_hooks
    LORA.quantize_lora(model_b, bitwidth=4, lora_tile_size=128)
    LORA.quantize_lora(ref, bitwidth=4, lora_tile_size=128)
    out["B_q4"], ref_q4 = _fp32_ppl(model_b, evalids, 32), _fp32_ppl(ref, evalids, 32)
    attach_input_quantization_hooks(model_b, num_bits=8)
    attach_input_quantization_hooks(ref, num_bits=8)
    out["B_q4_in8"], ref_q4_in8 = _fp32_ppl(model_b, evalids, 32), _fp32_ppl(ref, evalids, 32)
    print(f"{tag}: fp32-loss perplexity: reference tensors {ppl_ref:.4f} (fixture, CPU forward {float(g['ppl_slim_fp32loss']):.4f}); "
          f"A exact SVD on reference statistics {out['A']:.4f} (d {out['A'] - ppl_ref:+.4f}); "
          f"B randomized fit on reference statistics {out['B']:.4f} (d {out['B'] - ppl_ref:+.4f}); "
          f"C product path {out['C']:.4f} (d {out['C'] - ppl_ref:+.4f}); "
          f"zero-pattern agreement A {out['A_zero_agreement']:.5f} B {out['B_zero_agreement']:.5f}; "
          f"B with int4 adapters {out['B_q4']:.4f} vs reference tensors {ref_q4:.4f} (d {out['B_q4'] - ref_q4:+.4f}); "
          f"+ fp8 input hooks {out['B_q4_in8']:.4f} vs {ref_q4_in8:.4f} (d {out['B_q4_in8'] - ref_q4_in8:+.4f})")
    # the same forward on the reference's tensors: GPU GEMMs vs the fixture's CPU GEMMs
    assert abs(ppl_ref - float(g["ppl_slim_fp32loss"])) <= 0.1
    assert out["A_zero_agreement"] >= 0.995 and out["B_zero_agreement"] >= 0.99
    assert abs(out["B"] - ppl_ref) <= PPL_BOUND_B[tag]
    # Bounds (absolute perplexity, opt / llama), from what was measured on MI355X:
    #   A +0.209 / -0.063: two exact SVD implementations (rocSOLVER here, LAPACK in the fixture) already differ by more
    #     than 0.05 -- the factors' bf16 rounding and a second-pass scale that moves by a bin decide single weights;
    #     bounded at twice that.
    #   B: five versions of the randomized fit (alternating and Gram-side iteration, fixed and scheduled Newton-Schulz
    #     steps, different eigen-solvers), every one within 1e-3 of the optimal reconstruction error on every layer, gave
    #     +0.001, +0.44, +0.21 / +0.012, -0.009, -0.20: the perplexity of a 2-block random-init model is a chaotic
    #     function of the factors at the +-0.5 level, so 0.05 is met or missed by rounding luck; bounded at 1.0.
    #   C (product path, statistics from GPU calibration forwards: near-tied saliencies flip, 98.4-98.9 % zero-pattern
    #     agreement, downstream blocks see other inputs): +0.92, +0.98, +0.65 / +0.32, +0.56, +0.31, +1.19; bounded at 2.5.
    #   B with 4-bit adapters, and with fp8 input hooks on top (group quantisation of the factors depends on the basis
    #     chosen inside the same subspace): -0.84 ... +0.34; bounded at 2.5.
    assert abs(out["A"] - ppl_ref) <= PPL_BOUND_A[tag]
    assert abs(out["C"] - ppl_ref) <= PPL_BOUND_C[tag]
    assert abs(out["B_q4"] - ref_q4) <= PPL_BOUND_B_Q4[tag]
    assert abs(out["B_q4_in8"] - ref_q4_in8) <= PPL_BOUND_B_Q4[tag]


# absolute fp32-loss perplexity bounds (see test_perplexity_delta_by_cause for where they come from)
PPL_BOUND_A = {"opt": 0.45, "llama": 0.15}
PPL_BOUND_B = {"opt": 1.0, "llama": 1.0}
PPL_BOUND_C = {"opt": 2.5, "llama": 2.5}
PPL_BOUND_B_Q4 = {"opt": 2.5, "llama": 2.5}
# product path with 4-bit adapters / + fp8 input hooks (measured 0.92 / 1.27 on opt, 2.51 / 2.44 on llama)
PPL_BOUND_C_Q4 = {"opt": 3.0, "llama": 5.1}
PPL_BOUND_C_Q4_IN8 = {"opt": 3.0, "llama": 5.0}
