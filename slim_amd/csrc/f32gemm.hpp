// fp32 MFMA GEMM for the small square problems of the low-rank fit (f32gemm.hip): the l x l polynomial iterations of the
// Gram-space orthonormalisation and the back-product of the eigen-solver, where bf16 operands are not accurate enough.
#pragma once
#include "common.hpp"

struct F32GemmArgs {
    // C [M, N] (row-major, ldc) = alpha * sum_k a(i, k) * b(k, j).
    // a_kmajor: A is stored [Kd, M] (a(i, k) = A[k * lda + i]), else [M, Kd] (a(i, k) = A[i * lda + k]);
    // b_kmajor: B is stored [Kd, N] (b(k, j) = B[k * ldb + j]), else [N, Kd] (b(k, j) = B[j * ldb + k]).
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    bf16_t* C16 = nullptr;  // optional bf16 copy of C (row stride ldc)
    int M = 0, N = 0, Kd = 0;
    int lda = 0, ldb = 0, ldc = 0;
    int a_kmajor = 0, b_kmajor = 0;
    float alpha = 1.f;
    // gridDim.z = batch * nA problems: problem z = p * nA + w multiplies A + p * sA + w * sAw with B + p * sB into
    // C + p * sC + w * sCw (several left operands against one right operand in one launch)
    int batch = 1, nA = 1;
    int64_t sA = 0, sAw = 0, sB = 0, sC = 0, sCw = 0;
    // optional polynomial epilogue (square problems): C = pa * I + pb * Gin + pq * acc, Gin [M, N] row-major (ld N),
    // one per p (stride sG)
    const float* poly_G = nullptr;
    int64_t sG = 0;
    float pa = 0.f, pb = 0.f, pq = 0.f;
    // optional three-term epilogue with coefficients in device memory (a recurrence whose constants a previous kernel
    // computed): C = coef[0] * acc + coef[1] * D1 + coef[2] * D2, coef + p * coef_stride per problem p; D1 / D2 (either may
    // be NULL, its term is then dropped) are [M, N] matrices with C's row stride, one per problem (strides sD1 / sD2)
    const float* coef = nullptr;
    int coef_stride = 0;
    const float* D1 = nullptr;
    const float* D2 = nullptr;
    int64_t sD1 = 0, sD2 = 0;
    int coef_diag = 0;  // != 0: the third term is coef[2] * I (square products) instead of coef[2] * D2
    // optional scale in device memory: the product is divided by x * x, x = inv_sq_scale[p * iss_stride]
    const float* inv_sq_scale = nullptr;
    int iss_stride = 0;
};

int slim_f32gemm_launch(const F32GemmArgs& g, void* stream);
