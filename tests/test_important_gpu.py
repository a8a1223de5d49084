"""scale_important_weights (slim/lora.py:132-138, slim/quantization/quantization.py:133-170, :192, :215) on the GPU
against the reference-made fixtures g15_important* (single layers through the reference's add_lora with the error
matrix it handed to its SVD recorded) and g11_important* (the tiny LLaMA model through prune_and_quantize)."""
import pytest
import torch

from oracle import slim_oracle as O
from conftest import synth_weight, synth_scaler

DEV = "cuda:0"
pytestmark = pytest.mark.gpu

PT = ["slim", "slim_tall", "absmax", "slim_maskfirst"]
TILED = ["t128", "t128_maskfirst"]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops as _ops
    return _ops


def _case(golden, tag):
    g = golden("g15_important_tiled" if tag in TILED else "g15_important")
    p = tag + "/"
    return g, p, g[p + "W"], g[p + "scaler_row"], g[p + "mask"].bool(), bool(int(g[p + "quantize_first"]))


@pytest.mark.parametrize("tag", PT + TILED)
def test_important_columns_kernel_matches_reference(ops, golden, tag):
    g, p, W, sr, _, _ = _case(golden, tag)
    flags = ops.important_columns(W.to(DEV), sr.to(DEV)).cpu()
    assert flags.dtype == torch.bool and int(flags.sum()) == int(0.01 * W.shape[1])
    assert torch.equal(flags.nonzero().flatten(), g[p + "important"])


@pytest.mark.parametrize("N,K", [(1024, 4096), (4096, 11008), (300, 28672), (64, 96)])
def test_important_columns_kernel_matches_oracle(ops, N, K):
    W, sr = synth_weight(N, K, N + K), O.shift_zeros(synth_scaler(K, K))
    want = torch.zeros(K, dtype=torch.bool)
    want[O.important_columns(W, sr)] = True
    flags = ops.important_columns(W.to(DEV), sr.to(DEV)).cpu()
    assert int(flags.sum()) == int(0.01 * K)
    diff = (flags != want).nonzero().flatten()
    if diff.numel():
        # The column sums are rounded to bf16 from an fp32 accumulation whose order differs between the two sides: a sum that
        # lands on the other side of a rounding boundary moves its metric by one bf16 unit, which may swap two columns
        # at the threshold.  Anything else is a bug.
        metric = sr * W.abs().sum(dim=0)
        thr = metric[want].min()
        assert diff.numel() <= 2 and bool(torch.all((metric[diff] / thr - 1).abs() < 1e-2)), (diff, metric[diff], thr)


def test_important_columns_ties_take_the_lower_index(ops):
    W = torch.ones(8, 400, dtype=torch.bfloat16, device=DEV)
    sr = torch.ones(400, device=DEV)
    sr[[7, 300]] = 2.0
    flags = ops.important_columns(W, sr).cpu()          # 4 columns: the two larger, then ties 0, 1
    assert flags.nonzero().flatten().tolist() == [0, 1, 7, 300]
    assert not bool(ops.important_columns(W[:, :96].contiguous(), sr[:96].contiguous()).any())       # int(0.96) == 0


@pytest.mark.parametrize("tag", PT + TILED)
def test_quantizer_with_important_columns_matches_reference(ops, golden, tag):
    from slim_amd.quantization.quantization import Quantizer
    g, p, W, sr, mask, qf = _case(golden, tag)
    src = (W if qf else W.masked_fill(mask, 0)).to(DEV)
    before = src.clone()
    if tag in TILED:
        q = Quantizer("weight", num_bits=4, slim_quant=False, block_quantization=True, block_dim=128)
    else:
        q = Quantizer("weight", num_bits=4, slim_quant=(tag != "absmax"))
    codes = q.quantize_weight(src, g[p + "important"].to(DEV))
    assert torch.equal(src, before)                       # (the reference scales in place and undoes it)
    assert torch.equal(codes.cpu(), g[p + "codes1"])
    assert torch.equal(q.scaling_factor.float().cpu(), g[p + "sf1"].float())
    deq = q.dequantize_absmax(codes).cpu()
    if tag in TILED:
        # the fixture's values carry the Triton CPU interpreter's truncation to bf16; a GPU build rounds to nearest
        imp = g[p + "important"]
        want, _ = O.group_fake_quantize(src.cpu(), 128, 4, imp)
        assert torch.equal(deq, want)
        d = (deq.view(torch.int16).int() - g[p + "deq1"].view(torch.int16).int()).abs()
        assert int(d.max()) <= 1
    else:
        assert deq.dtype == g[p + "deq1"].dtype and torch.equal(deq, g[p + "deq1"])
    # the same call with the flags instead of the indices
    flags = torch.zeros(W.shape[1], dtype=torch.bool, device=DEV)
    flags[g[p + "important"].to(DEV)] = True
    assert torch.equal(q.quantize_weight(src, flags), codes)


@pytest.mark.parametrize("tag", PT + TILED)
def test_add_lora_with_important_columns(ops, golden, tag):
    """Error matrix bit-exact (per-tensor quantisers) and the whole layer against the reference's result: same important
    columns, the saliency-weighted error of the compressed layer within 1 % of the reference's."""
    from slim_amd.lora import add_lora
    from slim_amd.quantization.quantization import Quantizer
    import slim_amd.lora as L
    g, p, W, sr, mask, qf = _case(golden, tag)
    lin = torch.nn.Linear(W.shape[1], W.shape[0], bias=False).to(torch.bfloat16).to(DEV)
    lin.weight.data = W.to(DEV).clone()

    class Act:
        scaler_row = sr.to(DEV)
    if tag in TILED:
        q = Quantizer("weight", num_bits=4, slim_quant=False, block_quantization=True, block_dim=128)
    else:
        q = Quantizer("weight", num_bits=4, slim_quant=(tag != "absmax"))
    c = L.add_lora_prepare(lin, mask.to(DEV), 0.1, True, Act, q, False, True, None, qf, True)
    assert torch.equal(c.important.nonzero().flatten().cpu(), g[p + "important"])
    E_ref = g[p + "E"]
    if tag in TILED:
        rel = (torch.norm(c.E.float().cpu() - E_ref) / torch.norm(E_ref)).item()
        assert rel < 2e-2, rel              # the interpreter's truncation of the fixture's dequantised values (see the oracle test)
        want = O.add_lora(W, mask, sr, 0.1, num_bits=4, scale_important_weights=True, block=128, quantize_first=qf)["error_mat"]
        assert torch.equal(c.E.cpu(), want.to(torch.bfloat16))
    else:
        assert torch.equal(c.E.cpu(), E_ref.to(torch.bfloat16))
    L.add_lora_fit([c])
    L.add_lora_finish(c)
    r = lin.lora_left.shape[1]
    assert tuple(lin.lora_left.shape) == tuple(g[p + "lora_left"].shape) and tuple(lin.lora_right.shape) == tuple(g[p + "lora_right"].shape)
    s = torch.sqrt(sr).reshape(1, -1)
    W0 = W.float()

    def weighted_error(weight, left, right):
        return torch.norm((W0 - weight.float() - right.float().t() @ left.float().t()) * s).item()
    eo = weighted_error(lin.weight.data.cpu(), lin.lora_left.data.cpu(), lin.lora_right.data.cpu())
    er = weighted_error(g[p + "weight"], g[p + "lora_left"], g[p + "lora_right"])
    print(tag, "weighted-error excess", eo / er - 1)
    assert eo <= er * 1.01, (eo, er)
    agree = ((lin.weight.data.cpu() == 0) == (g[p + "weight"] == 0)).float().mean().item()
    assert agree > 0.99, agree
    if tag not in TILED:
        sf2 = q.scaling_factor.float().cpu()
        assert abs(float(sf2) / float(g[p + "sf2"]) - 1) < 0.05
        # levels: multiples of 1/sf outside the important columns, of 1/(2 sf) inside
        imp = g[p + "important"]
        lv = lin.weight.data.float().cpu() * sf2
        other = torch.ones(W.shape[1], dtype=torch.bool)
        other[imp] = False
        # (the weight is a bf16 tensor: a level of size 8 / sf carries a rounding of up to 8 x 2^-8)
        assert float((lv[:, other] - lv[:, other].round()).abs().max()) < 0.035
        assert float((2 * lv[:, imp] - (2 * lv[:, imp]).round()).abs().max()) < 0.07
        assert float(lv[:, imp].abs().max()) <= 4.01


@pytest.mark.parametrize("slim_quant,block", [(True, None), (False, None), (False, 128)])
def test_second_quantisation_given_identical_inputs(ops, slim_quant, block):
    from slim_amd.lora import requantize_weight
    from slim_amd.quantization.quantization import Quantizer
    W, sr = synth_weight(256, 512, 3, heavy=True), synth_scaler(512, 4)
    imp = O.important_columns(W, sr)
    flags = torch.zeros(512, dtype=torch.bool)
    flags[imp] = True
    q = Quantizer("weight", num_bits=4, slim_quant=slim_quant, block_quantization=block is not None, block_dim=block or 16)
    got, sf = requantize_weight(W.to(DEV), q, 4, flags.to(DEV))
    if block is None:
        codes, sf_ref = O.quantize_per_tensor(W, 4, slim_quant=slim_quant, important=imp)
        want = O.dequantize_per_tensor(codes, sf_ref, imp).to(torch.bfloat16)
        assert torch.equal(sf.float().cpu(), sf_ref.float().reshape(1, 1))
    else:
        want, _ = O.group_fake_quantize(W, block, 4, imp)
        assert sf is None
    assert torch.equal(got.cpu(), want)


def _tiny_llama(golden):
    from transformers import LlamaConfig, LlamaForCausalLM
    g = golden("g8_e2e_llama")
    m = LlamaForCausalLM(LlamaConfig(vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                     num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=32))
    m = m.to(torch.bfloat16).eval()
    m.config.max_position_embeddings = 32
    m.load_state_dict({k: g["init/" + k] for k in g["state_keys"].tolist()})
    return m, g


@pytest.mark.parametrize("variant", ["important", "important_tiled"])
def test_driver_with_important_columns_matches_reference(ops, golden, variant):
    import slim_amd.prune as P
    from slim_amd.eval import eval_ppl_wikitext
    from slim_amd.utils import get_layers_list, find_layers, check_sparsity
    g = golden("g11_" + variant)
    kw = dict(bitwidth=4, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="2:4", quantize_weight=True,
              lora_rank=0.1, slim_lora=True, quantize_lora=False, scale_important_weights=True)
    if variant == "important_tiled":
        kw.update(slim_quant=False, weight_tiled_quantization=True, weight_tile_size=128)
    else:
        kw.update(slim_quant=True)
    model, g8 = _tiny_llama(golden)
    orig = {k: v.clone() for k, v in model.state_dict().items()}
    calib = [(g8["calib_ids"][i:i + 1], None) for i in range(8)]
    P.get_loaders = lambda *a, **k: (calib, None)
    P.KEEP_ON_DEVICE = True
    stats = {}
    P.LAYER_CALLBACK = lambda i, name, lin, wrapped: stats.__setitem__((i, name), wrapped.scaler_row.clone())
    try:
        P.prune_and_quantize(model, None, nsamples=8, **kw)
    finally:
        P.KEEP_ON_DEVICE = False
        P.LAYER_CALLBACK = None
    model = model.to(DEV)
    problems = []
    for i, layer in enumerate(get_layers_list(model)):
        for name, lin in find_layers(layer).items():
            p = f"L{i}/{name}/"
            Wr, Lr, Rr, r = g[p + "weight"], g[p + "lora_left"], g[p + "lora_right"], int(g[p + "lora_rank"])
            assert tuple(lin.lora_left.shape) == tuple(Lr.shape) and int(lin.lora_rank) == r
            agree = ((lin.weight.data == 0).cpu() == (Wr == 0)).float().mean().item()
            # (block 1 sees the GPU's own block-0 outputs; kept weights that quantise to code 0 follow the second scale,
            # which the two-stage search moves in steps: measured 0.979-0.993 there, 0.999 in block 0)
            if agree < (0.99 if i == 0 else 0.97):
                problems.append((p, "zero-pattern agreement", agree))
            s = torch.sqrt(stats[(i, name)]).cpu().reshape(1, -1)
            W0 = orig[f"model.layers.{i}.{name}.weight"].float()
            eff_o = lin.weight.data.float().cpu() + (lin.lora_right.data.float().cpu().t() @ lin.lora_left.data.float().cpu().t()) / r
            eff_r = Wr.float() + (Rr.float().t() @ Lr.float().t()) / r
            eo, er = torch.norm((W0 - eff_o) * s).item(), torch.norm((W0 - eff_r) * s).item()
            print(p, "zero agreement", round(agree, 4), "weighted-error excess", round(eo / er - 1, 5))
            if eo > er * (1.02 if i == 0 else 1.08):
                problems.append((p, "saliency-weighted error", eo, er))
            if variant == "important":
                # more than 16 levels: the important columns carry half steps
                assert lin.scaling_factor.shape == (1, 1)
                assert torch.unique(lin.weight.data).numel() > 16
    assert not problems, problems
    assert abs(check_sparsity(model) - float(g["sparsity"])) < 3e-3

    class Enc:
        input_ids = g8["eval_ids"]
    ppl = eval_ppl_wikitext(model, Enc, 2, DEV, fp32_loss=True)
    print(variant, "fp32-loss ppl ours", ppl, "reference (cpu)", float(g["ppl_fp32loss"]))
    assert abs(ppl - float(g["ppl_fp32loss"])) <= 0.01 * float(g["ppl_fp32loss"])


def test_no_adapters_with_important_columns_raises_like_the_reference(ops, golden):
    """slim/prune.py:333 reads `scaler_row` off the nn.Linear: AttributeError in the reference, and here."""
    import slim_amd.prune as P
    model, g8 = _tiny_llama(golden)
    calib = [(g8["calib_ids"][i:i + 1], None) for i in range(8)]
    P.get_loaders = lambda *a, **k: (calib, None)
    with pytest.raises(AttributeError, match="scaler_row"):
        P.prune_and_quantize(model, None, nsamples=8, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, lora_rank=0., scale_important_weights=True)


def test_packed_form_refuses_half_levels(ops):
    """A weight compressed with scale_important_weights carries, in the flagged columns, levels of half the step (k / (2 sf)):
    it is not a 4-bit code of the layer's scale, and the 2:4 int4 packing (pack_model, the multi-GPU exchange) must say so
    instead of storing the nearest level."""
    g = torch.Generator().manual_seed(3)
    N, K, sf = 128, 256, 64.0
    codes = torch.randint(-8, 8, (N, K), generator=g).float()
    keep = torch.zeros(N, K // 4, 4, dtype=torch.bool)
    keep[:, :, :2] = True                                             # two of every four input channels kept
    W = (codes * keep.reshape(N, K) / sf).to(torch.bfloat16).to(DEV)
    scale = torch.tensor([sf], device=DEV)
    vals, meta = ops.pack24(W, scale)                                 # plain levels: fine
    assert torch.equal(ops.unpack24(vals, meta, scale, N, K), W)
    W2 = W.clone()
    W2[:, 0] = (3.5 / sf)                                             # an odd multiple of half a step, as a doubled column leaves
    with pytest.raises(ValueError, match="not 4-bit quantisation levels"):
        ops.pack24(W2, scale)
