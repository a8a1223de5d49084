"""The driver at the shapes of OPT-125m (config 1 of the benchmark: q, k, v, out 768 x 768; fc1 3072 x 768; fc2 768 x 3072,
all with bias; rank int(0.1 * 768) = 76 padded to the 256 tile, so the fit's subspace is l = 320; 589 / 2359 histogram bins,
slim/lora.py:174-181, slim/quantization/quantization.py:187) with the README recipe: wanda 2:4 + 4-bit SLiM-Quant +
lora_rank 0.1 + quantize_lora with tile 256.

What is held against the oracle, on the statistics the driver itself gathered (so both sides see identical saliency):
the 2:4 mask, the first-pass scale and the error matrix (which is a function of mask, scale and first-pass codes) bit
for bit; the fit within 1e-3 of the truncated SVD of that error matrix; the attribute contract; a forward pass."""
import pytest
import torch

from oracle import slim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEQ, NSAMPLES, VOCAB = 128, 16, 512


def _model(layers=1):
    from transformers import OPTConfig, OPTForCausalLM
    torch.manual_seed(0)
    m = OPTForCausalLM(OPTConfig(vocab_size=VOCAB, hidden_size=768, ffn_dim=3072, num_hidden_layers=layers, num_attention_heads=12,
                                 max_position_embeddings=SEQ, word_embed_proj_dim=768))
    m = m.to(torch.bfloat16).eval()
    m.config.max_position_embeddings = SEQ
    return m


def test_opt125m_shaped_block_through_the_driver():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import slim_amd.prune as P
    import slim_amd.lora as LORA
    from slim_amd import ops
    from slim_amd.utils import get_layers_list, find_layers, check_sparsity
    model = _model()
    orig = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    calib = [(torch.randint(0, VOCAB, (1, SEQ), generator=g), None) for _ in range(NSAMPLES)]
    seen = {}
    real_prepare, real_finish = LORA.add_lora_prepare_fused, LORA.add_lora_finish

    def prepare(module, *a, **k):
        c = real_prepare(module, *a, **k)
        seen[id(module)] = dict(mask=c.W_mask.clone(), E=c.E.clone(), s=c.s.clone(), rank=c.rank)
        return c

    def finish(c, prescale=False):
        U, Ut, L, info = c.fit
        seen[id(c.module)].update(U=U.clone(), L=L.clone(), info=info.clone())
        return real_finish(c, prescale=prescale)
    stats = {}
    saved = (P.get_loaders, P.LAYER_CALLBACK, P.KEEP_ON_DEVICE)
    P.get_loaders = lambda *a, **k: (calib, None)
    P.LAYER_CALLBACK = lambda i, name, lin, wrapped: stats.__setitem__((i, name), wrapped.scaler_row.clone())
    P.KEEP_ON_DEVICE = True
    LORA.add_lora_prepare_fused, LORA.add_lora_finish = prepare, finish
    try:
        P.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, nsamples=NSAMPLES, lora_rank=0.1, slim_lora=True,
                             quantize_lora=True, lora_tile_size=256)
    finally:
        P.get_loaders, P.LAYER_CALLBACK, P.KEEP_ON_DEVICE = saved
        LORA.add_lora_prepare_fused, LORA.add_lora_finish = real_prepare, real_finish
    subset = find_layers(get_layers_list(model)[0])
    assert sorted(subset) == sorted(["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj", "fc1", "fc2"])
    for name, lin in subset.items():
        N, K = lin.weight.shape
        assert (N, K) in ((768, 768), (3072, 768), (768, 3072)) and lin.bias is not None
        rec = seen[id(lin)]
        W0 = orig[f"model.decoder.layers.0.{name}.weight"]
        assert torch.equal(orig[f"model.decoder.layers.0.{name}.bias"], lin.bias.data.cpu())       # the bias is not touched
        # ---- identical saliency on both sides: the statistics and their square root as the driver had them
        sr = stats[(0, name)]
        s = rec["s"]
        assert torch.equal(s, ops.sqrt(sr))
        s_cpu = s.cpu().reshape(1, -1)
        mask = O.prune_nm(W0.abs() * s_cpu, 2, 4)
        assert torch.equal(rec["mask"].cpu(), mask), name
        bins = O.num_bins_for(W0.numel())
        assert bins == (589 if N == K else 2359)
        codes, sf1 = O.quantize_per_tensor(W0, 4)
        E = W0 * s_cpu - (O.dequantize_per_tensor(codes, sf1) * s_cpu).masked_fill(mask, 0)
        assert torch.equal(rec["E"].cpu(), E.to(torch.bfloat16)), name
        # ---- the fit: rank 256 out of a subspace of 320, within 1e-3 of the truncated SVD of the error matrix
        assert rec["rank"] == 256 and int(lin.lora_rank) == 256 and int(rec["info"][:-3].abs().sum()) == 0
        Ef = rec["E"].float()
        sv = torch.linalg.svdvals(Ef.double())
        opt = float(torch.sqrt((sv[256:] ** 2).sum()))
        s16 = s.to(torch.bfloat16).float().reshape(-1, 1)
        err = float(torch.norm(Ef.double() - rec["U"][:, :256].double() @ (rec["L"][:, :256].float() * s16).double().t()))
        print(f"{name} {N}x{K}: excess over the truncated SVD {err / opt - 1:+.2e}, sigma_1/sigma_256 {float(sv[0] / sv[255]):.1f}")
        assert err / opt - 1 <= 1e-3, (name, err / opt - 1)
        # ---- contract
        assert tuple(lin.lora_left.shape) == (K, 256) and tuple(lin.lora_right.shape) == (256, N)
        assert lin.scaling_factor.shape == (1, 1) and lin.scaling_factor.dtype == torch.float32
        assert lin.weight.data.float().unique().numel() <= 16
        assert bool((lin.weight.data[rec["mask"]] == 0).all())
    assert abs(check_sparsity(model) - 0.5) < 0.02
    # quantised adapters and a forward pass of the compressed model
    model.to(DEV)
    LORA.quantize_lora(model, bitwidth=4, lora_tile_size=256)
    with torch.no_grad():
        out = model(calib[0][0].to(DEV)).logits
    assert out.shape == (1, SEQ, VOCAB) and bool(torch.isfinite(out).all())
