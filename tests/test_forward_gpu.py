"""Packed 2:4 int4 storage and the fused forward pass, against the reference's emulated forward
(nn.Linear on the dequantised bf16 weight + add_lora_hook, slim/prune.py:303-321) run with torch on the
same GPU, and against an fp64 evaluation of the same expression."""
import functools
import math

import pytest
import torch

from oracle import slim_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops as _ops
    return _ops


@functools.lru_cache(maxsize=2)
def make_weight(N, K, seed, sf=None):
    """A weight of the form the compression leaves behind: 2:4 sparse, values bf16(c / sf), c in [-8, 7]."""
    g = torch.Generator().manual_seed(seed)
    codes = torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8).float()
    keep = O.prune_nm(torch.rand(N, K, generator=g), 2, 4)        # True = pruned
    sf = torch.tensor(sf if sf is not None else 37.3, dtype=torch.float32)
    W = (codes / sf).to(torch.bfloat16)
    W[keep] = 0
    return W, sf.reshape(1)


def test_pack_unpack_round_trip(ops):
    for (N, K, seed) in [(128, 64, 1), (256, 512, 2), (1024, 4096, 3)]:
        W, sf = make_weight(N, K, seed)
        vals, meta = ops.pack24(W.to(DEV), sf.to(DEV))
        assert vals.numel() == N * K // 4 and meta.numel() == N * K // 8
        back = ops.unpack24(vals, meta, sf.to(DEV), N, K).cpu()
        assert torch.equal(back, W)
        # bit-identical except that a dequantised -0 is stored as +0
        assert torch.equal(back.view(torch.int16)[W != 0], W.view(torch.int16)[W != 0])


def test_pack_rejects_bad_weights(ops):
    W, sf = make_weight(128, 64, 4)
    bad = W.clone()
    bad[3, 0:4] = torch.tensor([1, 2, 3, 0], dtype=torch.float32).div(sf).to(torch.bfloat16)
    with pytest.raises(ValueError, match="2:4"):
        ops.pack24(bad.to(DEV), sf.to(DEV))
    bad = W.clone()
    bad[5, 8:12] = torch.tensor([0.01234, 0, 0, 0]).to(torch.bfloat16)
    with pytest.raises(ValueError, match="levels"):
        ops.pack24(bad.to(DEV), sf.to(DEV))
    with pytest.raises(ValueError):
        ops.pack24(W[:100].contiguous().to(DEV), sf.to(DEV))      # N not a multiple of 128


def reference_forward(lin, x):
    """What the reference runs: nn.Linear on the dequantised weight, then the adapter hook (bf16, torch GEMMs)."""
    from slim_amd.prune import add_lora_hook
    y = torch.nn.functional.linear(x, lin.weight.data, lin.bias)
    if hasattr(lin, "lora_left"):
        add_lora_hook(lin, (x,), y)
    return y


def bf16r(t):
    return t.to(torch.bfloat16).double()


def rounded_fp64_forward(lin, x, ops):
    """The same expression with the same bf16 rounding points, every matmul accumulated in fp64."""
    y = bf16r(x.double() @ lin.weight.data.double().t() + (0 if lin.bias is None else lin.bias.data.double()))
    if hasattr(lin, "lora_left"):
        sqrt_r = torch.sqrt(lin.lora_rank).to(x.device)
        q = getattr(lin, "lora_quantizer", None)
        if q is not None:
            Lq = ops.group_fake_quant(lin.lora_left.data.contiguous(), q.block_dim, q.num_bits)
            Rq = ops.group_fake_quant(lin.lora_right.data.contiguous(), q.block_dim, q.num_bits)
            t = bf16r((x / sqrt_r).double() @ Lq.double())
            ad = bf16r((t.to(torch.bfloat16) / sqrt_r).double() @ Rq.double())
        else:
            t = bf16r(x.double() @ (lin.lora_left.data / sqrt_r).double())
            ad = bf16r(t @ (lin.lora_right.data / sqrt_r).double())
        y = bf16r(y + ad)
    return y


def build_layer(N, K, r, seed, bias, quant_block=None):
    from slim_amd.quantization.quantization import Quantizer
    W, sf = make_weight(N, K, seed)
    lin = torch.nn.Linear(K, N, bias=bias).to(torch.bfloat16).to(DEV)
    lin.weight.data = W.to(DEV)
    lin.scaling_factor = sf.reshape(1, 1).to(DEV)
    g = torch.Generator().manual_seed(seed + 100)
    if bias:
        lin.bias.data = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    if r:
        lin.lora_rank = torch.tensor(r)
        lin.lora_left = torch.nn.Parameter((torch.randn(K, r, generator=g) * 0.05 * math.sqrt(r)).to(torch.bfloat16).to(DEV))
        lin.lora_right = torch.nn.Parameter((torch.randn(r, N, generator=g) * 0.05 * math.sqrt(r)).to(torch.bfloat16).to(DEV))
        if quant_block:
            lin.lora_quantizer = Quantizer("weight", num_bits=4, block_quantization=True, block_dim=quant_block)
    return lin


CASES = [
    # N, K, r, bias, quantised-adapter block, batch sizes
    (8192, 8192, 256, False, 256, [1, 7, 16]),          # large enough for the register-streaming small-batch kernel
    (16384, 4096, 200, True, None, [3, 16]),
    (128, 64, 0, False, None, [1, 16, 130]),
    (256, 512, 51, True, None, [1, 5, 17, 33, 100, 128, 300]),
    (256, 512, 128, False, 128, [1, 16, 64, 200]),
    (1024, 4096, 409, False, None, [1, 16, 48, 1000]),
    (4096, 1024, 256, True, 128, [2, 128, 513]),
    # 9 tiles of 128 rows: the last 256-row panel of the wide tile is half full
    (1152, 512, 51, True, None, [100, 300, 513]),
    (1152, 512, 128, False, 128, [129, 640]),
    # the forward microbenchmark's layer (LLaMA-3-70B up/gate projection, 10 % rank adapters), quantised and bf16 adapters
    # up to the microbenchmark's largest batch (the wide tile's full-grid regime)
    (28672, 8192, 1024, False, 256, [1, 16, 256, 512, 1024, 2048, 4096]),
    (28672, 8192, 1024, False, None, [1, 16, 256, 1024, 4096]),
]


@pytest.mark.parametrize("q4", [False, True])
@pytest.mark.parametrize("N,K,r,bias,qblock,batches", CASES)
def test_packed_forward_matches_emulated_forward(ops, N, K, r, bias, qblock, batches, q4):
    from slim_amd.packed import PackedLinear
    import slim_amd.packed as P
    if q4 and not qblock:
        pytest.skip("4-bit adapter operands exist for quantised adapters only")
    lin = build_layer(N, K, r, seed=N + K + r, bias=bias, quant_block=qblock)
    saved_q4, P.Q4_ADAPTERS = P.Q4_ADAPTERS, q4
    try:
        packed = PackedLinear(lin)
    finally:
        P.Q4_ADAPTERS = saved_q4
    assert (packed.qL is not None) == bool(q4)
    assert torch.equal(packed.dense_weight(), lin.weight.data)
    dense_bytes = N * K * 2
    assert packed.vals.numel() + packed.meta.numel() == dense_bytes * 3 // 16
    for M in batches:
        g = torch.Generator().manual_seed(M)
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
        y = packed(x)
        assert y.shape == (M, N) and y.dtype == torch.bfloat16
        ref = reference_forward(lin, x)
        exact = rounded_fp64_forward(lin, x, ops)
        # Both sides round the same expression at the same points but accumulate each matmul in a different
        # order, so individual outputs may differ by a bf16 unit of one of the rounded terms.  Bound: every
        # output within 2^-6 of the output scale of the torch result, and our distance to the fp64-accumulated
        # value no larger than the torch path's (x1.25, plus the slack below).
        scale = exact.abs().max().item()
        diff = (y.double() - ref.double()).abs().max().item()
        err_ours = (y.double() - exact).norm().item()
        err_ref = (ref.double() - exact).norm().item()
        print(f"N={N} K={K} r={r} M={M}: max|ours-torch|={diff:.4g} scale={scale:.4g} "
              f"err ours={err_ours:.4g} torch={err_ref:.4g} identical={(y == ref).float().mean().item():.4f}")
        assert diff <= scale * 2.0 ** -6, (M, diff, scale)
        # slack: up to 0.1 % of the outputs one bf16 unit (of the output scale) away from the fp64-accumulated value
        slack = scale * 2.0 ** -8 * math.sqrt(1e-3 * y.numel())
        assert err_ours <= err_ref * 1.25 + slack, (M, err_ours, err_ref)
        if M > 64:
            # the 128 x 128 tile and the 128 x 256 tile (with the first adapter product as an NT GEMM), whichever of them
            # the library's rule picked above: the same bounds, and identical outputs wherever neither path splits a
            # reduction into slices (both then accumulate in the same order; the small layers with short reductions here)
            from slim_amd import _lib
            for mode in (0, 1):
                try:
                    _lib.query("slim_linear24_set_wide", mode)
                    y_m = packed(x)
                finally:
                    _lib.query("slim_linear24_set_wide", -1)
                assert (y_m.double() - ref.double()).abs().max().item() <= scale * 2.0 ** -6, (M, mode)
                assert (y_m.double() - exact).norm().item() <= err_ref * 1.25 + slack, (M, mode)
                if N <= 4096 and K < 2048:
                    assert torch.equal(y_m, y), (M, mode, (y_m != y).sum().item())
        if M >= 512:
            # the optional transient dense expansion + library GEMMs: the reference's own operations, bit for bit
            saved, P.LARGE_BATCH = P.LARGE_BATCH, M
            try:
                y2 = packed(x)
            finally:
                P.LARGE_BATCH = saved
            assert torch.equal(y2, ref)


def test_packed_forward_leading_dims_and_hooks(ops):
    from slim_amd.packed import PackedLinear
    from slim_amd.quantization.quantization import attach_input_quantization_hooks
    lin = build_layer(256, 512, 64, seed=9, bias=False)
    packed = PackedLinear(lin)
    x = torch.randn(2, 7, 512, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(DEV)
    y = packed(x)
    assert y.shape == (2, 7, 256)
    assert torch.equal(y.reshape(14, 256), packed(x.reshape(14, 512)))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear24_forward(x.reshape(14, 512).cpu(), packed.vals, packed.meta, packed.sf, 256)


def test_packed_checkpoint_round_trip(ops, tmp_path):
    """pack_model -> save_packed -> load_packed into a freshly initialised model of the same architecture:
    identical tensors and identical logits."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from slim_amd.packed import PackedLinear, pack_model, save_packed, load_packed
    from slim_amd.utils import get_layers_list, find_layers
    cfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, max_position_embeddings=32)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).to(DEV).eval()
    # give every block linear the compressed form (2:4 sparse 4-bit levels + adapters)
    for layer in get_layers_list(model):
        for name, lin in find_layers(layer).items():
            N, K = lin.weight.shape
            src = build_layer(N, K, 16, seed=N + K, bias=False)
            lin.weight.data = src.weight.data
            lin.scaling_factor, lin.lora_rank = src.scaling_factor, src.lora_rank
            lin.lora_left, lin.lora_right = src.lora_left, src.lora_right
    dense, packed = pack_model(model)
    # (at these toy sizes the adapter operands, padded to 128 rows, outweigh the saving on the weight)
    assert dense > 0 and packed > 0 and sum(isinstance(m, PackedLinear) for m in model.modules()) == 14
    ids = torch.randint(0, 128, (2, 16), generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        ref = model(ids).logits
    path = str(tmp_path / "packed.safetensors")
    index = save_packed(model, path)
    assert len(index) == 14
    torch.manual_seed(1)
    fresh = LlamaForCausalLM(cfg).to(torch.bfloat16).to(DEV).eval()
    fresh.load_state_dict({k: v for k, v in model.state_dict().items()}, strict=False)   # embeddings, norms, head
    assert load_packed(fresh, path, DEV) == 14
    for (ka, a), (kb, b) in zip(sorted((k, m) for k, m in model.named_modules() if isinstance(m, PackedLinear)),
                                sorted((k, m) for k, m in fresh.named_modules() if isinstance(m, PackedLinear))):
        assert ka == kb and a.rank == b.rank and a.a_div == b.a_div
        sa, sb = a.state(), b.state()
        assert sa.keys() == sb.keys() and {"vals", "meta", "sf", "LeffT", "ReffT"} <= sa.keys()
        for t in sa:
            assert torch.equal(sa[t], sb[t]), (ka, t)
    with torch.no_grad():
        assert torch.equal(fresh(ids).logits, ref)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("pad_x,pad_y,off_y", [(8, 8, 0), (16, 4, 0), (8, 3, 0), (8, 8, 4)])
def test_forward_with_strided_activations_and_outputs(ops, mode, pad_x, pad_y, off_y):
    """slim_linear24_forward through the C ABI with row strides larger than the row length (x: a multiple of 8 elements;
    y: any) and an output pointer that is only 8-byte aligned, on both tiles: the same values as the contiguous call
    (the kernels' 16-, 8- and 2-byte store paths), and nothing written between the rows."""
    from slim_amd import _lib
    from slim_amd.packed import PackedLinear
    N, K, r, M = 1152, 512, 51, 300
    lin = build_layer(N, K, r, seed=77, bias=True)
    p = PackedLinear(lin)
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(DEV)
    try:
        _lib.query("slim_linear24_set_wide", mode)
        ref = p(x)
        xs = torch.zeros(M, K + pad_x, dtype=torch.bfloat16, device=DEV)
        xs[:, :K] = x
        ybuf = torch.full((M * (N + pad_y) + 16,), 7.0, dtype=torch.bfloat16, device=DEV)
        need = _lib.query("slim_linear24_workspace_bytes", M, N, K, r)
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        _lib.call("slim_linear24_forward", xs.data_ptr(), K + pad_x, M, K, p.vals.data_ptr(), p.meta.data_ptr(),
                  p.sf.data_ptr(), N, p.bias.data_ptr(), p.LeffT.data_ptr(), p.ReffT.data_ptr(), r, float(p.a_div),
                  ybuf.data_ptr() + 2 * off_y, N + pad_y, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        _lib.query("slim_linear24_set_wide", -1)
    y = ybuf[off_y:off_y + M * (N + pad_y)].reshape(M, N + pad_y)
    assert torch.equal(y[:, :N], ref)
    assert bool((y[:, N:] == 7.0).all()) and bool((ybuf[:off_y] == 7.0).all())
