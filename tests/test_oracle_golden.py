"""The CPU oracle (oracle/slim_oracle.py) against golden vectors produced by the reference."""
import numpy as np
import pytest
import torch

from oracle import slim_oracle as O
from conftest import synth_weight


def test_g1_scaler_row_and_gram(golden):
    g = golden("g1_stats")
    X = g["X"]
    sr, n = torch.zeros(64), 0
    H, nh = torch.zeros(64, 64), 0
    for j in range(4):
        sr, n = O.add_batch_scaler_row(sr, n, X[j])
        H, nh = O.add_batch_gram(H, nh, X[j])
    assert n == int(g["nsamples"])
    assert torch.equal(sr, g["scaler_row"])
    assert torch.equal(H, g["H"])
    # diag(H)/2 is the same statistic
    torch.testing.assert_close(torch.diag(H) / 2, sr, rtol=1e-5, atol=0)
    sr2, n2 = torch.zeros(64), 0
    sr2, n2 = O.add_batch_scaler_row(sr2, n2, X[:2, 0])
    sr2, n2 = O.add_batch_scaler_row(sr2, n2, X[2:, 0])
    assert n2 == int(g["nsamples_b2"]) and torch.equal(sr2, g["scaler_row_b2"])


def test_g2_shift_zeros(golden):
    g = golden("g2_shift_zeros")
    for a, b in (("v", "out"), ("v2", "out2"), ("v3", "out3")):
        assert torch.equal(O.shift_zeros(g[a]), g[b])


def test_g3_prune_nm(golden):
    g = golden("g3_prune_nm")
    for n in (1, 2, 3):
        assert torch.equal(O.prune_nm(g["tuples"], n, 4), g[f"mask_n{n}"].bool())
    assert torch.equal(O.prune_nm(g["rand"], 2, 4), g["rand_mask_2_4"].bool())
    sal = O.saliency(g["W"], g["scaler_row"])
    assert torch.equal(sal, g["saliency"])
    for n in (1, 2, 3):
        assert torch.equal(O.prune_nm(sal, n, 4), g[f"sal_mask_{n}_4"].bool())
    assert torch.equal(O.prune_nm(g["special"], 2, 4), g["special_mask_2_4"].bool())


def _case_weight(g, tag):
    if tag + "_seed_shape" in g:
        seed, n, k = [int(v) for v in g[tag + "_seed_shape"]]
        if tag == "uniform_edge":
            return (torch.rand(n, k, generator=torch.Generator().manual_seed(seed)) * 2 - 1).to(torch.bfloat16)
        return synth_weight(n, k, seed)
    return g[tag + "_W"]


@pytest.mark.parametrize("tag", ["bins512", "bins589", "bins2359", "heavy", "masked", "bins20000", "uniform_edge"])
def test_g4_scale_search(golden, tag):
    g = golden("g4_g5_slim_quant")
    W = _case_weight(g, tag)
    alpha, idx, pdf, val, trace = O.find_optimal_quantization_cap(W, 4, O.num_bins_for(W.numel()), return_trace=True)
    assert torch.equal(pdf, g[tag + "_pdf"]) and torch.equal(val, g[tag + "_val"])
    assert [t[0] for t in trace] == g[tag + "_eval_idx"].tolist()
    assert np.array_equal(np.array([t[1] for t in trace], dtype=np.float32), g[tag + "_eval_loss"].numpy())
    # the fixture's "alpha" is 7 / scaling_factor (a round trip), so the pin is the scaling factor itself
    assert abs(float(alpha) - float(g[tag + "_alpha"])) <= 1e-6 * float(alpha)
    codes, sf = O.quantize_per_tensor(W, 4, abs_max=alpha)
    assert torch.equal(sf, g[tag + "_sf"])
    if tag + "_codes" in g:
        assert torch.equal(codes, g[tag + "_codes"])
        assert torch.equal(O.dequantize_per_tensor(codes, sf).to(torch.bfloat16), g[tag + "_deq_bf16cast"])
    else:
        assert int(codes.long().sum()) == int(g[tag + "_codes_sum"])
        assert torch.equal(torch.bincount((codes.long() + 8).flatten(), minlength=16), g[tag + "_codes_hist"])


def test_g5_variants_and_ties(golden):
    g = golden("g4_g5_slim_quant")
    W = g["bins512_W"]
    c, sf = O.quantize_per_tensor(W, 4, slim_quant=False)
    assert torch.equal(c, g["bins512_b4_sq0_codes"]) and torch.equal(sf, g["bins512_b4_sq0_sf"])
    c, sf = O.quantize_per_tensor(W, 8, slim_quant=True)
    assert torch.equal(c, g["bins512_b8_sq1_codes"]) and torch.equal(sf, g["bins512_b8_sq1_sf"])
    c, sf = O.quantize_per_tensor(W, 8, slim_quant=False)
    assert torch.equal(c, g["bins512_b8_sq0_codes"]) and torch.equal(sf, g["bins512_b8_sq0_sf"])
    c, _ = O.quantize_per_tensor(g["ties_W"], 4, abs_max=torch.tensor(7.0))
    assert torch.equal(c, g["ties_codes_sf1"])


def test_g6_group_quant(golden):
    g = golden("g6_group_quant")
    for tag in ("a", "b", "c"):
        x, blk = g[tag + "_x"], int(g[tag + "_blk"])
        for q in (4, 8):
            al = O.group_quant_params(x, blk)
            assert torch.equal(al, g[f"{tag}_q{q}_alphas"])
            codes = O.group_quantize(x, al, q)
            assert torch.equal(codes, g[f"{tag}_q{q}_codes"])
            deq = O.group_dequantize(codes, al, q)
            ref = g[f"{tag}_q{q}_deq"]
            # the golden comes from the Triton CPU interpreter, which truncates fp32 -> bf16:
            # allow one bf16 ulp, and require exactness wherever the fp32 value is representable
            d = (deq.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
            assert int(d.max()) <= 1


@pytest.mark.parametrize("tag", ["a", "b", "c", "naive"])
def test_g7_add_lora(golden, tag):
    g = golden("g7_add_lora")
    W, sr, mask, tile = g[tag + "_W"], g[tag + "_scaler_row"], g[tag + "_mask"].bool(), int(g[tag + "_tile"])
    out = O.add_lora(W, mask, sr, 0.1, num_bits=4, tile=tile)
    assert torch.equal(out["weight"], g[tag + "_new_weight"])
    assert torch.equal(out["lora_left"], g[tag + "_lora_left"])
    assert torch.equal(out["lora_right"], g[tag + "_lora_right"])
    assert torch.equal(out["scaling_factor"], g[tag + "_sf2"])


def test_g9_input_quant(golden):
    g = golden("g9_input_quant")
    for tag in ("small", "large"):
        x = g[tag + "_x"]
        assert torch.equal(O.quantize_input_fp8(x.clone()), g[f"{tag}_b8_g-1"])
        assert torch.equal(O.quantize_input_int(x.clone(), 4, 64), g[f"{tag}_b4_g64"])
        assert torch.equal(O.quantize_input_int(x.clone(), 6, -1), g[f"{tag}_b6_g-1"])


@pytest.mark.parametrize("tag", ["opt", "llama"])
def test_g13_oracle_reproduces_reference_end_to_end_layers(golden, tag):
    """Every linear of the reference's end-to-end runs (g8), from the statistics the reference's add_lora received
    (g13): mask, first-pass codes and scale, final weight, adapters and second scale, all bit for bit."""
    import math
    g8, g13 = golden("g8_e2e_" + tag), golden("g13_e2e_stats_" + tag)
    prefix = "model.decoder.layers" if tag == "opt" else "model.layers"
    for i in range(2):
        for name in g13["names"].tolist():
            p = f"L{i}/{name}/"
            W = g8[f"init/{prefix}.{i}.{name}.weight"]
            sr = g13[p + "scaler_row"]
            mask_ref = torch.from_numpy(np.unpackbits(g13[p + "mask"].numpy(), axis=1)[:, :W.shape[1]]).bool()
            sal = torch.abs(W) * g13[p + "s"].reshape(1, -1)
            assert torch.equal(O.prune_nm(sal, 2, 4), mask_ref), p
            codes, sf = O.quantize_per_tensor(W, 4)
            assert torch.equal(codes, g13[p + "codes1"]) and torch.equal(sf.reshape(-1), g13[p + "sf1"].reshape(-1)), p
            out = O.add_lora(W, mask_ref, sr, 0.1, num_bits=4, tile=128)
            r = int(g8[p + "lora_rank"])
            assert out["rank"] == r
            assert torch.equal(out["weight"], g8[p + "weight"]), p
            assert torch.equal(out["scaling_factor"].reshape(-1), g8[p + "sf"].reshape(-1)), p
            # the driver multiplies the adapters by sqrt(rank) (slim/prune.py:324-326)
            sq = torch.sqrt(torch.tensor(r))
            assert torch.equal(out["lora_left"] * sq, g8[p + "lora_left"]), p
            assert torch.equal(out["lora_right"] * sq, g8[p + "lora_right"]), p


# ---- scale_important_weights (g15: the reference's add_lora with its error matrix recorded) ------------------------
@pytest.mark.parametrize("tag", ["slim", "slim_tall", "absmax", "slim_maskfirst"])
def test_important_columns_per_tensor(golden, tag):
    g = golden("g15_important")
    p = tag + "/"
    W, sr, mask = g[p + "W"], g[p + "scaler_row"], g[p + "mask"].bool()
    qf = bool(int(g[p + "quantize_first"]))
    imp = O.important_columns(W, sr)
    assert torch.equal(torch.sort(imp)[0], g[p + "important"])
    assert torch.equal(W.abs().sum(dim=0), g[p + "colsum"])
    slim_quant = tag != "absmax"
    src = W if qf else W.masked_fill(mask, 0)
    codes, sf = O.quantize_per_tensor(src, 4, slim_quant=slim_quant, important=imp)
    assert torch.equal(codes, g[p + "codes1"])
    assert torch.equal(sf.float(), g[p + "sf1"])
    assert torch.equal(O.dequantize_per_tensor(codes, sf, imp), g[p + "deq1"])
    out = O.add_lora(W, mask, sr, 0.1, num_bits=4, tile=None, scale_important_weights=True, slim_quant=slim_quant,
                     quantize_first=qf)
    assert torch.equal(out["error_mat"], g[p + "E"])
    # the important columns carry levels of half the size: up to 32 distinct values in the weight instead of 16
    assert torch.equal(out["weight"], g[p + "weight"])
    assert torch.equal(out["scaling_factor"].float(), g[p + "sf2"])
    assert torch.equal(out["lora_left"], g[p + "lora_left"]) and torch.equal(out["lora_right"], g[p + "lora_right"])
    assert torch.unique(out["weight"][:, imp]).numel() <= 16
    assert torch.unique(out["weight"]).numel() <= 32


@pytest.mark.parametrize("tag", ["t128", "t128_maskfirst"])
def test_important_columns_tiled(golden, tag):
    g = golden("g15_important_tiled")
    p = tag + "/"
    W, sr, mask = g[p + "W"], g[p + "scaler_row"], g[p + "mask"].bool()
    qf = bool(int(g[p + "quantize_first"]))
    imp = O.important_columns(W, sr)
    assert torch.equal(torch.sort(imp)[0], g[p + "important"])
    src = W if qf else W.masked_fill(mask, 0)
    deq, alphas = O.group_fake_quantize(src, 128, 4, imp)
    assert torch.equal(alphas, g[p + "sf1"])
    # the Triton CPU interpreter truncates fp32 -> bf16 where a GPU build rounds to nearest: <= 1 bf16 unit
    d = (deq.view(torch.int16).int() - g[p + "deq1"].view(torch.int16).int()).abs()
    assert int(d.max()) <= 1
    # the error matrix the reference handed to its SVD is exactly W * s - mask0(deq * s) of ITS dequantised values; the
    # oracle's differs from it only through the interpreter's truncation (a bf16 unit of deq is ~10 % of a quantisation
    # error, so the matrices agree to ~1 % and not to the bit)
    s = torch.sqrt(sr.reshape(1, -1))
    E_ref = W * s - (g[p + "deq1"] * s).masked_fill(mask, 0)
    assert torch.equal(E_ref, g[p + "E"])
    out = O.add_lora(W, mask, sr, 0.1, num_bits=4, tile=None, scale_important_weights=True, block=128, quantize_first=qf)
    rel = (torch.norm(out["error_mat"] - g[p + "E"]) / torch.norm(g[p + "E"])).item()
    assert rel < 2e-2, rel
