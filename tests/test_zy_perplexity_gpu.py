"""Perplexity of the compressed model against the reference's, measured instead of argued: the randomized fit's seed is
swept (8 seeds), on the two 2-block fixtures (g8_e2e_*, 256 evaluation tokens) and on a 4-block, hidden-512 LLaMA-shaped
fixture with an evaluation stream of 4096 tokens (g16_e2e_large, made by tests/golden/make_golden_e2e_large.py from the
reference).  Every number is an fp32-loss perplexity from the same GPU forward (slim/eval.py:94-113 with the loss in
fp32); `reference` is a model carrying the tensors the reference left on every linear.

  B: the reference's statistics injected, the randomized fit -> what the factorisation alone changes
  C: the product path (statistics from GPU calibration forwards)
"""
import hashlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEEDS = list(range(8))


class Enc:
    def __init__(self, ids):
        self.input_ids = ids


def _ppl(model, ids, bs=4):
    from slim_amd.eval import eval_ppl_wikitext
    return eval_ppl_wikitext(model, Enc(ids), bs, DEV, fp32_loss=True)


def _stats(xs):
    m = sum(xs) / len(xs)
    sd = math.sqrt(sum((x - m) ** 2 for x in xs) / (len(xs) - 1))
    return m, sd


# ---- the large fixture ---------------------------------------------------------------------------------------------
SEQ, VOCAB, NCAL = 64, 512, 16


def build_large():
    """Same call as tests/golden/make_golden_e2e_large.py:build_model."""
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


def _compress_large(g, inject, seed):
    import slim_amd.prune as P
    import slim_amd.lora as LORA
    model = build_large()
    calib = [(g["calib_ids"][i:i + 1], None) for i in range(NCAL)]

    def stats_cb(i, name, wrapped):
        wrapped.scaler_row = g[f"L{i}/{name}/scaler_row"].to(DEV)
    saved = (P.get_loaders, P.STATS_CALLBACK, P.KEEP_ON_DEVICE, LORA.FIT_SEED)
    P.get_loaders = lambda *a, **k: (calib, None)
    P.STATS_CALLBACK = stats_cb if inject else None
    P.KEEP_ON_DEVICE = True
    LORA.FIT_SEED = seed
    try:
        P.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, nsamples=NCAL, lora_rank=0.1, slim_lora=True,
                             quantize_lora=True, lora_tile_size=64)
    finally:
        P.get_loaders, P.STATS_CALLBACK, P.KEEP_ON_DEVICE, LORA.FIT_SEED = saved
    return model.to(DEV)


def test_perplexity_seed_sweep_large_fixture(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    from slim_amd.utils import get_layers_list, find_layers
    g = golden("g16_e2e_large")
    ref = build_large()
    assert state_hash(ref) == str(g["state_sha256"]), "the seeded initial weights differ from the fixture's"
    evalids = g["eval_ids"]
    dense = _ppl(ref.to(DEV), evalids)
    for i, layer in enumerate(get_layers_list(ref)):
        for name, lin in find_layers(layer).items():
            p = f"L{i}/{name}/"
            lin.weight.data = (g[p + "codes"] / g[p + "sf"]).to(torch.bfloat16).to(DEV)
            lin.lora_left = torch.nn.Parameter(g[p + "lora_left"].to(DEV))
            lin.lora_right = torch.nn.Parameter(g[p + "lora_right"].to(DEV))
            lin.lora_rank = torch.tensor(int(g[p + "lora_rank"]))
            lin.register_forward_hook(P.add_lora_hook)
    ppl_ref = _ppl(ref, evalids)
    # the same tensors in the fixture's CPU forward
    assert abs(dense / float(g["ppl_dense_fp32loss"]) - 1) < 2e-4 and abs(ppl_ref / float(g["ppl_slim_fp32loss"]) - 1) < 2e-4
    res = {}
    for key, inject in (("B", True), ("C", False)):
        res[key] = [_ppl(_compress_large(g, inject, seed), evalids) for seed in SEEDS]
    (mb, sb), (mc, sc) = _stats(res["B"]), _stats(res["C"])
    print(f"large fixture: dense {dense:.3f}, reference tensors {ppl_ref:.3f} (compression costs {ppl_ref - dense:+.3f}); "
          f"B mean {mb:.3f} (d {mb - ppl_ref:+.3f}, sd {sb:.3f}, range {min(res['B']) - ppl_ref:+.3f} .. {max(res['B']) - ppl_ref:+.3f}); "
          f"C mean {mc:.3f} (d {mc - ppl_ref:+.3f}, sd {sc:.3f}, range {min(res['C']) - ppl_ref:+.3f} .. {max(res['C']) - ppl_ref:+.3f})")
    LARGE_RESULTS.update(dense=dense, ref=ppl_ref, B=res["B"], C=res["C"])
    # Measured on MI355X (perplexity 564.9; the whole compression costs +2.8), the fits of blocks 1-3 Chebyshev-filtered as the
    # driver runs them: B mean +0.52, sd over the seeds 0.58, single runs -0.26 .. +1.39 (with unfiltered fits, SLIM_FIT_FILTER=0:
    # mean +0.21, sd 0.38, -0.61 .. +0.59; with four filter rounds instead of three: mean +0.38, sd 0.35; every fit of all
    # three within 1.3e-4 of the truncated SVD, so what moves the perplexity is not the quality of the fit but which way
    # roundings tip after it); C mean +0.35, sd 0.92, single runs -1.34 .. +1.38.  The seed of the fit alone moves the
    # perplexity by +-0.1 to 0.2 %, so an absolute 0.05 (0.009 % here) is met or missed by the draw; the tolerance is therefore
    # stated relative to the perplexity (0.05 on a WikiText-2 perplexity of 5.5, where the reference states it, is 0.9 %):
    # single runs within 0.3 % (B) / 0.4 % (C), the mean over the seeds within 0.15 %, and the reference's value inside the
    # distribution of ours (within three standard deviations of the mean: no bias beyond the seed's noise).
    assert max(abs(x / ppl_ref - 1) for x in res["B"]) <= 3e-3, (res["B"], ppl_ref)
    assert max(abs(x / ppl_ref - 1) for x in res["C"]) <= 4e-3, (res["C"], ppl_ref)
    assert abs(mb / ppl_ref - 1) <= 1.5e-3 and abs(mc / ppl_ref - 1) <= 1.5e-3
    assert abs(mb - ppl_ref) <= 3 * sb and abs(mc - ppl_ref) <= 3 * sc


LARGE_RESULTS = {}


# ---- the two small fixtures -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["opt", "llama"])
def test_perplexity_seed_sweep_small_fixtures(golden, tag):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    import slim_amd.lora as LORA
    import test_e2e_gpu as E
    from slim_amd.utils import get_layers_list, find_layers
    g = golden("g8_e2e_" + tag)
    evalids = g["eval_ids"]
    ref = E._make(tag)
    E._load_init(ref, g)
    ref.to(DEV)
    for i, layer in enumerate(get_layers_list(ref)):
        for name, lin in find_layers(layer).items():
            p = f"L{i}/{name}/"
            lin.weight.data = g[p + "weight"].to(DEV)
            lin.lora_left = torch.nn.Parameter(g[p + "lora_left"].to(DEV))
            lin.lora_right = torch.nn.Parameter(g[p + "lora_right"].to(DEV))
            lin.lora_rank = torch.tensor(int(g[p + "lora_rank"]))
            lin.register_forward_hook(P.add_lora_hook)
    ppl_ref = _ppl(ref, evalids, 2)
    res = {"B": [], "C": []}
    saved = LORA.FIT_SEED
    try:
        for seed in SEEDS:
            LORA.FIT_SEED = seed
            res["B"].append(_ppl(E._compress(golden, tag, True, False), evalids, 2))
            res["C"].append(_ppl(E._compress(golden, tag, False, False), evalids, 2))
    finally:
        LORA.FIT_SEED = saved
    (mb, sb), (mc, sc) = _stats(res["B"]), _stats(res["C"])
    print(f"{tag}: reference tensors {ppl_ref:.3f}; B mean {mb:.3f} (d {mb - ppl_ref:+.3f}, sd {sb:.3f}, range "
          f"{min(res['B']) - ppl_ref:+.3f} .. {max(res['B']) - ppl_ref:+.3f}); C mean {mc:.3f} (d {mc - ppl_ref:+.3f}, sd {sc:.3f}, "
          f"range {min(res['C']) - ppl_ref:+.3f} .. {max(res['C']) - ppl_ref:+.3f})")
    # 256 evaluation tokens.  Measured (opt / llama, perplexity 262.6 / 268.2): B mean +0.14 / -0.22, sd 0.32 / 0.24; C mean
    # +0.44 / +0.45, sd 0.36 / 0.51, single runs up to +1.27.  Relative to the perplexity: single runs within 0.8 %, the
    # mean over the seeds within 0.3 %; B, which differs from the reference in the factorisation only, within three
    # standard deviations of its seeds.
    for key in ("B", "C"):
        assert max(abs(x / ppl_ref - 1) for x in res[key]) <= 8e-3, (key, res[key], ppl_ref)
    assert abs(mb / ppl_ref - 1) <= 3e-3 and abs(mc / ppl_ref - 1) <= 3e-3
    assert abs(mb - ppl_ref) <= 3 * sb
