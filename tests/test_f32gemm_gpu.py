"""The fp32 matrix-core GEMM (slim_gemm_f32: the eigen-solver's back-product) against float64, all four operand layouts,
ragged sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,N,Kd", [(64, 64, 32), (4, 8, 4), (100, 260, 36), (512, 640, 640), (640, 640, 644), (1120, 1400, 1400)])
@pytest.mark.parametrize("a_kmajor,b_kmajor", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_f32_matches_float64(M, N, Kd, a_kmajor, b_kmajor):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slim_amd import _lib
    g = torch.Generator().manual_seed(M + N + Kd)
    a = torch.randn(M, Kd, generator=g)
    b = torch.randn(Kd, N, generator=g)
    a[0] = torch.arange(Kd).float() / Kd            # asymmetric content
    A = (a.t() if a_kmajor else a).contiguous().to(DEV)
    B = (b if b_kmajor else b.t()).contiguous().to(DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    _lib.call("slim_gemm_f32", A.data_ptr(), A.shape[1], a_kmajor, B.data_ptr(), B.shape[1], b_kmajor, M, N, Kd, 0.5,
              C.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
    ref = 0.5 * (a.double() @ b.double())
    # fp32 products and fp32 accumulation over Kd terms
    tol = 4e-7 * Kd ** 0.5 * float(a.abs().max() * b.abs().max()) * 4
    assert float((C.cpu().double() - ref).abs().max()) <= tol
