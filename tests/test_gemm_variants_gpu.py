"""Every workgroup-tile variant of the NT bf16 GEMM (128 x 128, 128 x 64, 64 x 128, 64 x 64) against a float64 product of
the same bf16 operands, on full, ragged and tiny shapes, with every output form (fp32, bf16, transposed bf16, split-K)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N_VARIANTS = 4
SHAPES = [(128, 128, 64), (256, 384, 640), (200, 136, 72), (8, 8, 8), (640, 640, 520), (1000, 72, 1096), (64, 4096, 264),
          (130, 258, 2056)]


@pytest.fixture(autouse=True)
def _reset_tile():
    from slim_amd import _lib
    yield
    _lib.query("slim_gemm_set_tile", -1)


@pytest.mark.parametrize("variant", list(range(N_VARIANTS)) + [-1])
def test_gemm_variant_matches_reference(variant):
    from slim_amd import _lib, ops
    _lib.query("slim_gemm_set_tile", variant)
    g = torch.Generator(device="cuda").manual_seed(variant + 5)
    for M, N, Kd in SHAPES:
        A = torch.randn(M, Kd, device="cuda", generator=g).to(torch.bfloat16)
        Bt = torch.randn(N, Kd, device="cuda", generator=g).to(torch.bfloat16)
        ref = A.double() @ Bt.double().t()
        C32, C16, Ct = ops.gemm_nt(A, Bt, alpha=0.5, out_f32=True, out_bf16=True, out_t=True)
        torch.cuda.synchronize()
        tol = 2e-6 * Kd ** 0.5 * 4 + 1e-6
        assert (C32.double() - 0.5 * ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (variant, M, N, Kd)
        assert torch.equal(C16, C32.to(torch.bfloat16)), (variant, M, N, Kd)
        assert torch.equal(Ct, C16.t()), (variant, M, N, Kd)
        # accumulate into an existing fp32 matrix
        C0 = torch.randn(M, N, device="cuda", generator=g)
        out = ops.gemm_nt(A, Bt, alpha=1.0, beta=2.0, C32=C0.clone(), out_f32=True)
        out = out[0] if isinstance(out, (tuple, list)) else out
        assert (out.double() - (ref + 2.0 * C0.double())).abs().max().item() <= tol * max(1.0, ref.abs().max().item()) * 2
        if Kd >= 256:
            S32, S16 = ops.gemm_nt_splitk(A, Bt, splits=3)
            assert (S32.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (variant, M, N, Kd, "split")
            assert torch.equal(S16, S32.to(torch.bfloat16))
            T16, Tt = ops.gemm_nt_splitk_t(A, Bt, splits=3, alpha=0.5)
            assert torch.equal(Tt, T16.t()), (variant, M, N, Kd, "split, transposed")
            assert (T16.double() - 0.5 * ref).abs().max().item() <= (tol + 2 ** -8) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,kd,splits", [(504, 4096, 2), (640, 1032, 4), (128, 512, 2), (200, 2048, 16)])
def test_symmetric_splitk_equals_full_product(n, kd, splits):
    """X X^T through the split-K entry point computes the tiles on and above the diagonal and mirrors the rest: the
    result is bit-identical to the same product with the second operand passed as a separate copy (all tiles computed)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n + kd)
    X = torch.randn(n, kd, generator=g, device="cuda").to(torch.bfloat16)
    full32, full16 = ops.gemm_nt_splitk(X, X.clone(), splits)
    sym32, sym16 = ops.gemm_nt_splitk(X, X, splits)
    torch.cuda.synchronize()
    assert torch.equal(sym32, full32) and torch.equal(sym16, full16)
    assert torch.equal(sym32, sym32.t())


@pytest.mark.parametrize("M,N,Kd", [(576, 4096, 4096), (576, 2056, 1600), (1152, 2048, 2048), (288, 1664, 1544), (320, 2048, 2048)])
def test_tall_tile_equals_square_tile(M, N, Kd):
    """The 288 x 128 tile (512 threads) the launcher picks for the (l x Dm) products of the fit (bf16 C and C^T, M the subspace
    dimension) against the 128 x 128 tile forced on the same operands: the same fp32 sums in the same order, so identical
    outputs.  (320 rows: the tall tile would pad more than the square one and is not picked; the case keeps the rule honest.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn(M, Kd, device="cuda", generator=g).to(torch.bfloat16)
    Bt = torch.randn(N, Kd, device="cuda", generator=g).to(torch.bfloat16)
    _, C16, Ct = ops.gemm_nt(A, Bt, out_bf16=True, out_t=True)                   # launcher's choice
    _lib.query("slim_gemm_set_tile", 0)
    _, R16, Rt = ops.gemm_nt(A, Bt, out_bf16=True, out_t=True)
    torch.cuda.synchronize()
    assert torch.equal(C16, R16) and torch.equal(Ct, Rt)
    assert torch.equal(Ct, C16.t())
    ref = A.double() @ Bt.double().t()
    assert (C16.double() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()


@pytest.mark.parametrize("variant", [5, 6, 7, 8])
def test_dma_tiles_equal_square_tile(variant):
    """The tiles fed by LDS-DMA several K-steps ahead (288 x 128, 128 x 128, 256 x 128) against the register-staged 128 x 128
    tile: the same fp32 sums in the same order, so identical outputs in every form -- full and ragged shapes, reductions
    shorter than the pipeline is deep, split-K; a reduction that is no multiple of 64 falls back to the register-staged tile."""
    from slim_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(variant)
    for M, N, Kd in [(576, 4096, 4096), (576, 2056, 1600), (300, 1000, 64), (128, 128, 128), (1000, 200, 192), (8, 8, 64),
                     (4096, 576, 576), (640, 640, 520), (512, 512, 2048)]:
        A = torch.randn(M, Kd, device="cuda", generator=g).to(torch.bfloat16)
        Bt = torch.randn(N, Kd, device="cuda", generator=g).to(torch.bfloat16)
        C0 = torch.randn(M, N, device="cuda", generator=g)
        outs = []
        for v in (0, variant):
            _lib.query("slim_gemm_set_tile", v)
            o = list(ops.gemm_nt(A, Bt, alpha=0.5, out_f32=True, out_bf16=True, out_t=True))
            o += list(ops.gemm_nt(A, Bt, out_bf16=True, out_t=True))[1:]          # (the form the 288-row tile serves)
            o.append(ops.gemm_nt(A, Bt, alpha=1.0, beta=2.0, C32=C0.clone(), out_f32=True)[0])
            if Kd >= 128:
                o += list(ops.gemm_nt_splitk(A, Bt, splits=2))
            if M == N:
                o += list(ops.gemm_nt_splitk(A, A, splits=2)) if Kd >= 128 else []
                H = C0[:, :M].clone() + C0[:, :M].t()
                o.append(ops.gemm_nt(A, A, alpha=1.0, beta=0.5, C32=H, out_f32=True)[0])
            outs.append(o)
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(*outs)):
            assert torch.equal(a, b), (variant, M, N, Kd, k)
        ref = 0.5 * (A.double() @ Bt.double().t())
        assert (outs[1][0].double() - ref).abs().max().item() <= 1e-5 * Kd ** 0.5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,kd", [(4096, 4096), (1024, 192), (1280, 640), (1100, 256), (2048, 11008)])
def test_mirrored_product_rectangular_tiles_equal_square_tiles(n, kd):
    """X X^T with mirrored bf16 output (the fit's Gram matrices C = E^T E): the 256 x 128 tiles fed by LDS-DMA -- rows of
    tiles from the diagonal on, the two leftmost of a row straddling it -- against the square register-staged tiles and
    against the full product: identical, and symmetric."""
    from slim_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(n + kd)
    X = torch.randn(n, kd, generator=g, device="cuda").to(torch.bfloat16)
    outs = {}
    for v in (0, 7, -1):
        _lib.query("slim_gemm_set_tile", v)
        outs[v] = ops.gemm_nt(X, X, mirrored=True)[1]
    _lib.query("slim_gemm_set_tile", 0)
    full = ops.gemm_nt(X, X.clone(), out_bf16=True)[1]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], full) and torch.equal(outs[7], full) and torch.equal(outs[-1], full)
    assert torch.equal(full, full.t())


@pytest.mark.parametrize("K,T", [(1280, 4096), (2048, 8192), (1096, 2048)])
def test_gram_accumulation_rectangular_tiles_equal_square_tiles(K, T):
    """slim_gram_accum (H = (2/n) sum X^T X, fp32, accumulated into: slim/sparsegpt.py:28-39) on the 256 x 128 tiles fed by
    LDS-DMA against the square register-staged tiles, two calls in a row: identical H, symmetric -- the tiles that straddle
    the diagonal compute their part below it themselves and nothing is mirrored into a position that is also computed."""
    from slim_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(K + T)
    X = [(torch.randn(T, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(2)]
    H = {}
    for v in (0, -1, 7):
        _lib.query("slim_gemm_set_tile", v)
        h = torch.zeros(K, K, device="cuda")
        for j in range(2):
            ops.gram_accum(X[j], h, j, 1)
        H[v] = h
    torch.cuda.synchronize()
    assert torch.equal(H[0], H[-1]) and torch.equal(H[0], H[7])
    assert torch.equal(H[0], H[0].t())
    ref = sum(x.double().t() @ x.double() for x in X) * (2.0 / 2)
    assert (H[0].double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
