// The Rayleigh-Ritz step of the low-rank fit without an eigen-decomposition.
//
// The fit needs, from the l x l Rayleigh-Ritz matrix M = Z^T Z of its subspace, the `keep`-dimensional invariant subspace
// of the largest eigenvalues (the factors are Z W and P W for ANY orthonormal basis W of it: the product, which is all the
// adapters are judged by, does not depend on the basis).  l - keep, the part to drop, is small (the oversampling: 64-80
// of 576 directions), and the spectrum of M is flat (largest / smallest eigenvalue 2-10 on SLiM error matrices).  So
// instead of all eigenpairs of M (a chain of l dependent Householder steps) this file finds the SMALL complementary
// subspace and takes its orthogonal complement:
//   1. bounds: b >= lambda_max(M) from || (M / ||M||)^8 ||_inf^(1/8) (three squarings on the matrix cores; an upper bound
//      by construction, 5-10 % above the eigenvalue), a >= lambda_c(M) (c = l - keep, the largest eigenvalue to drop) from
//      the same bound on a c x c compression of M (Cauchy interlacing: the largest eigenvalue of Q^T M Q is >= lambda_c for
//      any orthonormal l x c Q; first the leading principal block, then the current iterate);
//   2. Chebyshev-filtered subspace iteration on c vectors: the polynomial that is bounded on [a, b] and grows fastest below
//      a, evaluated by its three-term recurrence as l x l x c products on the fp32 matrix cores (coefficients read from
//      device memory, no host round trip), in rounds of increasing degree with a Householder QR and a tightened `a` between;
//   3. the complement: with Y = Q R the Householder QR of the filtered block, the columns c .. l-1 of Q = H_0 ... H_(c-1)
//      are an orthonormal basis of the orthogonal complement of span(Y): the wanted basis, exact to rounding.
// About sixty launches of fixed work against l dependent ones.  What the split costs in reconstruction error is what the
// dropped subspace holds above the c smallest eigenvalues; with the default degrees that is below 4e-4 of the error on the
// shapes of the benchmark (tests/test_ritz_split_gpu.py measures it against an exact eigen-decomposition).
// The outputs have the layout of the eigen-solver (slim_eigh_tail_f32_batched), so the fit's second half runs unchanged:
// row l-1-j of vecs = j-th basis vector (j < keep; rows up to `top` zero), evals[l-1-j] = its Rayleigh quotient.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include "f32gemm.hpp"

#define RS_MAX_BATCH 16
#define RS_LPER 9                       // 64-lane pieces of a vector: l <= 576
#define RS_MAX_L (RS_LPER * 64)
#define RS_CPW 5                        // vectors a wave of the QR kernel owns: c <= 80
#define RS_WAVES 16
#define RS_MAX_C (RS_CPW * RS_WAVES)
#define RS_SQUARINGS_SMALL 4
#define RS_SCAL 24                      // floats of per-problem scalars
#define CHK(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__ != SLIM_OK) return rc__; \
    } while (0)

struct RsKeep {
    int c[RS_MAX_BATCH];                // vectors to drop per problem (l - keep)
};

// SC_ONE: {1, *, *};  SC_T2: T_2(X) = 2 X^2 - I as coefficients of {M M, M, I};  SC_TWO: {2, 0, -1}
enum { SC_S = 0, SC_N3 = 1, SC_B = 2, SC_A = 3, SC_CA = 4, SC_ONE = 7, SC_CB = 8, SC_LO = 12, SC_HI = 13, SC_T2 = 16, SC_TWO = 20 };

// Sum over the 64 lanes, the result in every lane: data-parallel-primitive moves inside the rows of 16 lanes and two row
// broadcasts instead of six trips through the LDS crossbar (the QR kernel's critical path is a chain of these sums).
__device__ __forceinline__ float rs_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm 1,0,3,2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm 2,3,0,1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    // every lane holds its row's sum; rows 0..3 -> one value
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ uint32_t rs_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// Yt [B][c_pad][l]: standard-normal rows k < c[b], zero rows beyond (every problem receives the same numbers)
__global__ void rs_init_kernel(float* __restrict__ Yt, int l, int c_pad, RsKeep keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    if (i >= l) return;
    const uint32_t n = (uint32_t)(k * l + i);
    const uint32_t a = rs_mix32(n * 2u + 0x9e3779b9u), bb = rs_mix32(n * 2u + 1u + 0x85ebca6bu * 7u);
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(bb >> 8) * (1.0f / 16777216.0f);
    const float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
    Yt[((int64_t)b * c_pad + k) * l + i] = k < keep.c[b] ? z : 0.f;
}

// scal[b][slot] = max row sum of |X[b]| (X symmetric: column sums).  grid (ceil(n / 16), B), 256 threads.  The slot must be
// zero before the launch (non-negative floats compare like their bit patterns).
__global__ __launch_bounds__(256) void rs_infnorm_kernel(const float* __restrict__ X, int n, float* __restrict__ scal, int slot) {
    __shared__ float part[16][16];
    const int b = blockIdx.y, cc = threadIdx.x & 15, col = blockIdx.x * 16 + cc, g = threadIdx.x >> 4;
    const float* Xb = X + (int64_t)b * n * n;
    float s = 0.f;
    if (col < n) {
        int r = g;
        for (; r + 48 < n; r += 64) {
            const float x0 = Xb[(int64_t)r * n + col], x1 = Xb[(int64_t)(r + 16) * n + col], x2 = Xb[(int64_t)(r + 32) * n + col],
                        x3 = Xb[(int64_t)(r + 48) * n + col];
            s += (fabsf(x0) + fabsf(x1)) + (fabsf(x2) + fabsf(x3));
        }
        for (; r < n; r += 16) s += fabsf(Xb[(int64_t)r * n + col]);
    }
    part[g][cc] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        s = 0.f;
        for (int q = 0; q < 16; ++q) s += part[q][threadIdx.x];
        for (int o = 8; o > 0; o >>= 1) s = fmaxf(s, __shfl_xor(s, o, 64));
        if (threadIdx.x == 0) atomicMax((unsigned*)(scal + (int64_t)b * RS_SCAL + slot), __float_as_uint(s));
    }
}

// out [B][c_pad][l] = c0 * (Y M) + c1 * D1 + c2 * D2 for the skinny block Y [B][c_pad][l] (c_pad <= 80) and the symmetric
// M [B][l][l]: a step of the Chebyshev recurrence (coef: three floats per problem in device memory; D1 / D2 may be NULL).
// A workgroup owns 16 output columns: their panel of M (16 rows of the symmetric matrix, read contiguously) sits in LDS,
// the four waves split the reduction into quarters, every wave keeps five 16 x 16 accumulators (80 rows of Y) and takes its
// pieces of Y straight from memory as 16-byte loads three steps ahead; the quarters meet in LDS.  One step of 16 reduction
// indices is four v_mfma_f32_16x16x4_f32 per accumulator: the lane (i, kq) supplies Y[i][16 s + 4 kq + u] and
// M[16 s + 4 kq + u][j] for u = 0..3, the same assignment on both operands.  grid (l / 16, B), 256 threads.
#define RS_PANEL 16
#define RS_RB 5
__global__ __launch_bounds__(256) void rs_cheb_kernel(const float* __restrict__ Y, const float* __restrict__ M, int l, int c_pad,
                                                      const float* __restrict__ coef, int coef_stride, const float* __restrict__ D1,
                                                      const float* __restrict__ D2, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y, j0 = blockIdx.x * RS_PANEL, t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
    const int ldm = l + 4;                                  // 16-byte aligned rows, four banks apart
    float* Ms = lds;                                        // [16][ldm]: Ms[jj][k] = M[j0 + jj][k] (= M[k][j0 + jj])
    float* red = lds + RS_PANEL * ldm;                      // [4 waves][RS_RB][256]
    const float* Mb = M + (int64_t)b * l * l;
    const float* Yb = Y + (int64_t)b * c_pad * l;
    const int steps = (l + 15) / 16, per = (steps + 3) / 4, s0 = wave * per, s1 = s0 + per < steps ? s0 + per : steps;
    // this lane's pieces of Y: rows rb * 16 + fr, reduction indices 16 s + 4 fq ..+3
    auto load_a = [&](int s, f32x4 (&a)[RS_RB]) {
        const int k = 16 * s + 4 * fq;
#pragma unroll
        for (int rb = 0; rb < RS_RB; ++rb) {
            const int row = rb * 16 + fr;
            a[rb] = (s < s1 && row < c_pad && k < l) ? *(const f32x4*)(Yb + (int64_t)row * l + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 a0[RS_RB], a1[RS_RB], a2[RS_RB];
    load_a(s0, a0);
    load_a(s0 + 1, a1);
    load_a(s0 + 2, a2);
    for (int e = t; e < RS_PANEL * (l / 4); e += 256) {
        const int jj = e / (l / 4), k4 = e % (l / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j0 + jj < l) v = *(const f32x4*)(Mb + (int64_t)(j0 + jj) * l + 4 * k4);
        *(f32x4*)(Ms + jj * ldm + 4 * k4) = v;
    }
    __syncthreads();
    f32x4 acc[RS_RB];
#pragma unroll
    for (int rb = 0; rb < RS_RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto step = [&](int s, const f32x4 (&a)[RS_RB]) {
        const int k = 16 * s + 4 * fq;
        const f32x4 bv = k < l ? *(const f32x4*)(Ms + fr * ldm + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rb = 0; rb < RS_RB; ++rb)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][u], bv[u], acc[rb], 0, 0, 0);
    };
    for (int s = s0; s < s1; s += 3) {
        step(s, a0);
        load_a(s + 3, a0);
        if (s + 1 < s1) step(s + 1, a1);
        load_a(s + 4, a1);
        if (s + 2 < s1) step(s + 2, a2);
        load_a(s + 5, a2);
    }
    // acc[rb][r]: row rb * 16 + 4 fq + r, column fr
#pragma unroll
    for (int rb = 0; rb < RS_RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * RS_RB + rb) * 256 + (4 * fq + r) * 16 + fr] = acc[rb][r];
    __syncthreads();
    const float c0 = coef[(int64_t)b * coef_stride], c1 = coef[(int64_t)b * coef_stride + 1], c2 = coef[(int64_t)b * coef_stride + 2];
    for (int e = t; e < RS_RB * 256; e += 256) {
        const int row = e / 16, col = e % 16;
        if (row >= c_pad || j0 + col >= l) continue;
        const int rb = row / 16, rr = row % 16;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[(w * RS_RB + rb) * 256 + rr * 16 + col];
        const int64_t o = ((int64_t)b * c_pad + row) * l + j0 + col;
        v *= c0;
        if (D1) v += c1 * D1[o];
        if (D2) v += c2 * D2[o];
        out[o] = v;
    }
}

// One workgroup per problem.  Bm: a symmetric c x c matrix (row stride ldb; the c_pad - c trailing rows / columns of a
// padded matrix are ignored).  Upper bound of its largest eigenvalue by RS_SQUARINGS_SMALL normalised squarings in LDS ->
// a; first call (first != 0) also b from the squarings of M (scal[SC_S], scal[SC_N3]); then the recurrence coefficients of
// the Chebyshev polynomial on [a, b] for a round of `degree` steps:
//   first step   Y1 = (Y0 M - mid Y0) / half                 -> coefA = {1 / half, -mid / half, 0}
//   later steps  Y2 = 2 (Y1 M - mid Y1) / half - Y0          -> coefB = {2 / half, -2 mid / half, -1}
// a is kept far enough below b that |T_degree| stays below 1e30 for eigenvalues down to zero (M is positive
// semi-definite): (mid / half) <= X with (2 X)^degree = 1e30.
__global__ __launch_bounds__(1024) void rs_bounds_kernel(const float* __restrict__ Bm, int64_t sB, int ldb, RsKeep keep, int first,
                                                         int degree, float* __restrict__ scal) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, c = keep.c[b], t = threadIdx.x;
    float* sc = scal + (int64_t)b * RS_SCAL;
    float* X = lds;
    float* Y = lds + RS_MAX_C * RS_MAX_C;
    __shared__ float red[RS_WAVES];
    __shared__ float s_acc, s_trace;
    const float* B0 = Bm + (int64_t)b * sB;
    auto block_max = [&](float v) {
        v = wave_max(v);
        if ((t & 63) == 0) red[t >> 6] = v;
        __syncthreads();
        float m = red[0];
        for (int w = 1; w < RS_WAVES; ++w) m = fmaxf(m, red[w]);
        __syncthreads();
        return m;
    };
    float a = 0.f, tr = 0.f;
    if (c > 0) {
        // load, row sums, trace
        for (int e = t; e < c * c; e += 1024) X[e] = B0[(int64_t)(e / c) * ldb + e % c];
        __syncthreads();
        float rs = 0.f;
        if (t < c) {
            for (int j = 0; j < c; ++j) rs += fabsf(X[t * c + j]);
            tr = X[t * c + t];
        }
        float s0 = block_max(t < c ? rs : 0.f);
        // trace by the first two waves (c <= 80 < 128)
        {
            float v = wave_sum(t < c ? tr : 0.f);
            if ((t & 63) == 0) red[t >> 6] = v;
            __syncthreads();
            if (t == 0) s_trace = red[0] + red[1];
            __syncthreads();
        }
        // rho(X_0) <= exp(sum_m 2^-m log n_m): X_0 = B / s0, X_(m+1) = X_m^2 / n_m^2, n_m = ||X_m||_inf (n_0 = 1)
        if (t == 0) s_acc = 0.f;
        if (s0 > 0.f) {
            const float inv = 1.f / s0;
            for (int e = t; e < c * c; e += 1024) X[e] *= inv;
            __syncthreads();
            float* src = X;
            float* dst = Y;
            float nk = 1.f;
            for (int k = 0; k < RS_SQUARINGS_SMALL; ++k) {
                const float scale = 1.f / (nk * nk);
                for (int e = t; e < c * c; e += 1024) {
                    const int i = e / c, j = e % c;
                    float acc = 0.f;
                    int q = 0;
                    for (; q + 8 <= c; q += 8) {
                        float pa[8], pb[8];
#pragma unroll
                        for (int z = 0; z < 8; ++z) pa[z] = src[i * c + q + z], pb[z] = src[(q + z) * c + j];
#pragma unroll
                        for (int z = 0; z < 8; ++z) acc += pa[z] * pb[z];
                    }
                    for (; q < c; ++q) acc += src[i * c + q] * src[q * c + j];
                    dst[e] = acc * scale;
                }
                __syncthreads();
                float r2 = 0.f;
                if (t < c)
                    for (int j = 0; j < c; ++j) r2 += fabsf(dst[t * c + j]);
                nk = block_max(t < c ? r2 : 0.f);
                if (t == 0) s_acc += ldexpf(logf(fmaxf(nk, 1e-37f)), -(k + 1));
                float* tmp = src;
                src = dst;
                dst = tmp;
                if (!(nk > 0.f)) break;            // (uniform: every thread holds the same nk)
            }
            __syncthreads();
            a = s0 * expf(s_acc);
        }
        tr = s_trace;
    }
    if (t == 0) {
        float bb = sc[SC_B];
        if (first) {
            // rho(M) <= s * || (M^2 / s^2)^4 ||_inf^(1/8)
            const float s = sc[SC_S], n3 = sc[SC_N3];
            bb = s > 0.f ? s * powf(fmaxf(n3, 1e-37f), 0.125f) * 1.01f : 0.f;
            sc[SC_B] = bb;
            sc[SC_HI] = bb / 1.05f;
        }
        // keep a below b: (mid / half) <= X with (2 X)^degree = 1e30
        const float X = 0.5f * powf(10.f, 30.f / (float)(degree > 1 ? degree : 1));
        const float amax = bb * (X - 1.f) / (X + 1.f);
        a = fminf(a, amax);
        if (!(a > 0.f)) a = 0.5f * bb;
        const float half = 0.5f * (bb - a), mid = 0.5f * (bb + a);
        const float ih = half > 0.f ? 1.f / half : 0.f;
        sc[SC_A] = a;
        sc[SC_CA] = ih, sc[SC_CA + 1] = -mid * ih, sc[SC_CA + 2] = 0.f;
        sc[SC_ONE] = 1.f;
        sc[SC_CB] = 2.f * ih, sc[SC_CB + 1] = -2.f * mid * ih, sc[SC_CB + 2] = -1.f;
        // X = p M + q I with p = 1 / half, q = -mid / half:  T_2(X) = 2 X^2 - I = 2 p^2 M M + 4 p q M + (2 q^2 - 1) I
        const float q = -mid * ih;
        sc[SC_T2] = 2.f * ih * ih, sc[SC_T2 + 1] = 4.f * ih * q, sc[SC_T2 + 2] = 2.f * q * q - 1.f;
        sc[SC_TWO] = 2.f, sc[SC_TWO + 1] = 0.f, sc[SC_TWO + 2] = -1.f;
        if (c > 0) sc[SC_LO] = tr / (float)c;        // mean Rayleigh quotient of the block: where the dropped eigenvalues sit
    }
}

// Householder QR of the block Yt [B][c_pad][l] (row k = vector k) in registers: one workgroup of 16 waves per problem, wave w
// owns the vectors w, w + 16, ...; per step the pivot vector travels through LDS (two alternating slots: one barrier per
// step).  Out: V [B][c_pad][l] (reflector j: zeros above row j, 1 at row j), tau [B][c_pad]; vectors k >= c[b] (and a pivot
// that is zero from its row down) give tau = 0.
__global__ __launch_bounds__(1024) void rs_qr_kernel(const float* __restrict__ Yt, int l, int c_pad, RsKeep keep, float* __restrict__ V,
                                                     float* __restrict__ tau) {
    __shared__ float pivot[2][RS_MAX_L];
    const int b = blockIdx.x, c = keep.c[b], lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* Yb = Yt + (int64_t)b * c_pad * l;
    float* Vb = V + (int64_t)b * c_pad * l;
    float y[RS_CPW][RS_LPER];
#pragma unroll
    for (int q = 0; q < RS_CPW; ++q) {
        const int k = w + RS_WAVES * q;
#pragma unroll
        for (int u = 0; u < RS_LPER; ++u) {
            const int i = u * 64 + lane;
            y[q][u] = (k < c && i < l) ? Yb[(int64_t)k * l + i] : 0.f;
        }
    }
    for (int j = 0; j < c_pad; ++j) {
        const int owner = j % RS_WAVES, oq = j / RS_WAVES;
        if (w == owner) {
#pragma unroll
            for (int q = 0; q < RS_CPW; ++q)
                if (q == oq)
#pragma unroll
                    for (int u = 0; u < RS_LPER; ++u) pivot[j & 1][u * 64 + lane] = y[q][u];
        }
        __syncthreads();
        float v[RS_LPER];
        float sigma = 0.f;
        const float x0 = pivot[j & 1][j];
#pragma unroll
        for (int u = 0; u < RS_LPER; ++u) {
            const int i = u * 64 + lane;
            const float x = pivot[j & 1][i];
            v[u] = i > j ? x : 0.f;
            sigma += v[u] * v[u];
        }
        sigma = rs_wave_sum(sigma);
        float tj = 0.f;
        if (j < c && sigma > 0.f) {
            const float nrm = sqrtf(x0 * x0 + sigma), beta = x0 >= 0.f ? -nrm : nrm;
            tj = (beta - x0) / beta;
            const float inv = 1.f / (x0 - beta);
#pragma unroll
            for (int u = 0; u < RS_LPER; ++u) v[u] *= inv;
        } else {
#pragma unroll
            for (int u = 0; u < RS_LPER; ++u) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < RS_LPER; ++u)
            if (u * 64 + lane == j) v[u] = 1.f;
        if (w == owner) {
#pragma unroll
            for (int u = 0; u < RS_LPER; ++u) {
                const int i = u * 64 + lane;
                if (i < l) Vb[(int64_t)j * l + i] = v[u];
            }
            if (lane == 0) tau[(int64_t)b * c_pad + j] = tj;
        }
        if (tj != 0.f) {
#pragma unroll
            for (int q = 0; q < RS_CPW; ++q) {
                const int k = w + RS_WAVES * q;
                if (k > j && k < c) {
                    float d = 0.f;
#pragma unroll
                    for (int u = 0; u < RS_LPER; ++u) d += v[u] * y[q][u];
                    d = rs_wave_sum(d) * tj;
#pragma unroll
                    for (int u = 0; u < RS_LPER; ++u) y[q][u] -= d * v[u];
                }
            }
        }
    }
}

// Columns of Q = H_0 ... H_(c-1), one per wave, the reflectors staged through LDS sixteen at a time.
// thin != 0: column k = blockIdx.x * 16 + wave (k < c_pad) -> row k of out [B][c_pad][l] (zero for k >= c[b]): the
//            orthonormal basis of the block.
// thin == 0: column c[b] + j, j = blockIdx.x * 16 + wave (j < top) -> row l - 1 - j of out [B][l][l] (zero for
//            j >= l - c[b]): the basis of the orthogonal complement in the eigen-solver's layout.
#define RS_CHUNK 16
__global__ __launch_bounds__(1024) void rs_apply_kernel(const float* __restrict__ V, const float* __restrict__ tau, int l, int c_pad,
                                                        RsKeep keep, int thin, int top, float* __restrict__ out) {
    __shared__ float vs[RS_CHUNK][RS_MAX_L];
    __shared__ float ts[RS_CHUNK];
    const int b = blockIdx.y, c = keep.c[b], lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = blockIdx.x * RS_WAVES + w;
    const float* Vb = V + (int64_t)b * c_pad * l;
    const int col = thin ? idx : c + idx;
    const bool live = thin ? idx < c : (idx < top && col < l);
    float y[RS_LPER];
#pragma unroll
    for (int u = 0; u < RS_LPER; ++u) y[u] = (live && u * 64 + lane == col) ? 1.f : 0.f;
    for (int j1 = c; j1 > 0; j1 -= RS_CHUNK) {             // reflectors j1 - 1 down to j0
        const int j0 = j1 - RS_CHUNK > 0 ? j1 - RS_CHUNK : 0;
        __syncthreads();
        for (int e = threadIdx.x; e < (j1 - j0) * l; e += 1024) vs[e / l][e % l] = Vb[(int64_t)(j0 + e / l) * l + e % l];
        if (threadIdx.x < j1 - j0) ts[threadIdx.x] = tau[(int64_t)b * c_pad + j0 + threadIdx.x];
        __syncthreads();
        if (!live) continue;
        for (int j = j1 - 1; j >= j0; --j) {
            if (thin && j > col) continue;                 // v_j is zero at and above row col < j: H_j e_col = e_col
            const float tj = ts[j - j0];
            if (tj == 0.f) continue;
            float d = 0.f, vv[RS_LPER];
#pragma unroll
            for (int u = 0; u < RS_LPER; ++u) {
                vv[u] = u * 64 + lane < l ? vs[j - j0][u * 64 + lane] : 0.f;       // (the rows of vs are filled up to l only)
                d += vv[u] * y[u];
            }
            d = rs_wave_sum(d) * tj;
#pragma unroll
            for (int u = 0; u < RS_LPER; ++u) y[u] -= d * vv[u];
        }
    }
    if (thin) {
        if (idx >= c_pad) return;
        float* o = out + ((int64_t)b * c_pad + idx) * l;
#pragma unroll
        for (int u = 0; u < RS_LPER; ++u)
            if (u * 64 + lane < l) o[u * 64 + lane] = y[u];
    } else {
        if (idx >= top) return;
        float* o = out + ((int64_t)b * l + (l - 1 - idx)) * l;
#pragma unroll
        for (int u = 0; u < RS_LPER; ++u)
            if (u * 64 + lane < l) o[u * 64 + lane] = y[u];
    }
}

// evals[b][l - 1 - j] = <row l-1-j of W M, row l-1-j of W> for j < top (0 beyond keep), one wave per row; then the two
// slots the fit's conditioning check reads: evals[l - 1] = estimate of the largest eigenvalue of M, evals[(l - top) / 2] =
// mean Rayleigh quotient of the dropped block, evals[l - top - 1] = estimate of the largest dropped eigenvalue.  info[b] = 1 when a quotient is not finite.
__global__ __launch_bounds__(256) void rs_evals_kernel(const float* __restrict__ WM, const float* __restrict__ vecs, int l, int top,
                                                       RsKeep keep, const float* __restrict__ scal, float* __restrict__ evals,
                                                       int32_t* __restrict__ info) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= top) return;
    const int kept = l - keep.c[b], row = l - 1 - j;
    float d = 0.f;
    if (j < kept) {
        const float* a = WM + ((int64_t)b * top + (top - 1 - j)) * l;      // WM row index: rows l-top .. l-1 of vecs, in order
        const float* v = vecs + ((int64_t)b * l + row) * l;
        for (int i = lane; i < l; i += 64) d += a[i] * v[i];
        d = wave_sum(d);
    }
    if (lane == 0) {
        const float* sc = scal + (int64_t)b * RS_SCAL;
        float out = j < kept ? d : 0.f;
        if (!(out == out) || fabsf(out) > 3e38f) info[b] = 1;
        if (j == 0) out = fmaxf(sc[SC_HI], out);
        evals[(int64_t)b * l + row] = out;
        if (j == 0 && (l - top) / 2 < keep.c[b]) evals[(int64_t)b * l + (l - top) / 2] = sc[SC_LO];
        // the largest eigenvalue that is dropped: the last bound of the iteration lies 3-8 % above it
        if (j == 0 && l - top - 1 >= 0 && l - top - 1 < keep.c[b] && l - top - 1 != (l - top) / 2)
            evals[(int64_t)b * l + l - top - 1] = sc[SC_A] / 1.05f;
    }
}

static inline int64_t rs_al(int64_t x) { return (x + 63) & ~(int64_t)63; }

struct RsWs {
    float *Y[3], *V, *tau, *Bm, *P[2], *scal;
    int64_t total;
};

static RsWs rs_ws(char* base, int l, int c_pad, int batch) {
    RsWs w;
    int64_t off = 0;
    auto take = [&](int64_t floats) {
        float* p = (float*)(base + off);
        off += rs_al(floats * 4);
        return p;
    };
    const int64_t blk = (int64_t)batch * (c_pad > 0 ? c_pad : 4) * l;
    for (int i = 0; i < 3; ++i) w.Y[i] = take(blk);
    w.V = take(blk);
    w.tau = take((int64_t)batch * (c_pad > 0 ? c_pad : 4));
    w.Bm = take((int64_t)batch * (c_pad > 0 ? c_pad * c_pad : 16));
    w.P[0] = take((int64_t)batch * l * l);
    w.P[1] = take((int64_t)batch * l * l);
    w.scal = take((int64_t)batch * RS_SCAL);
    w.total = off;
    return w;
}

static int rs_cpad(const int32_t* keep, int l, int b0, int nb) {
    int c = 0;
    for (int b = 0; b < nb; ++b) c = l - keep[b0 + b] > c ? l - keep[b0 + b] : c;
    return (c + 3) & ~3;
}

extern "C" int slim_ritz_split_supported(int l, int drop_max) {
    return l >= 8 && l % 8 == 0 && l <= RS_MAX_L && drop_max >= 0 && ((drop_max + 3) & ~3) <= RS_MAX_C && drop_max < l;
}

extern "C" int64_t slim_ritz_split_workspace_bytes(int l, int drop_max, int batch) {
    if (!slim_ritz_split_supported(l, drop_max) || batch < 1) return 0;
    return rs_ws(nullptr, l, (drop_max + 3) & ~3, batch < RS_MAX_BATCH ? batch : RS_MAX_BATCH).total;
}

// SLIM_RS_STRIDE4=0: every Chebyshev step as a launch of its own (A/B aid)
static const bool g_stride4 = [] {
    const char* v = getenv("SLIM_RS_STRIDE4");
    return !(v && v[0] == '0');
}();

static int rs_run(const float* M, int l, int nb, int top, const int32_t* keep_h, const int* degs, int ndeg, float* vecs, float* evals,
                  int32_t* info, char* ws_base, hipStream_t st) {
    RsKeep keep;
    int cmax = 0;
    for (int b = 0; b < RS_MAX_BATCH; ++b) keep.c[b] = 0;
    for (int b = 0; b < nb; ++b) {
        keep.c[b] = l - keep_h[b];
        cmax = keep.c[b] > cmax ? keep.c[b] : cmax;
    }
    const int c_pad = (cmax + 3) & ~3;
    RsWs w = rs_ws(ws_base, l, c_pad, nb);
    const int64_t sM = (int64_t)l * l, sY = (int64_t)c_pad * l;
    SLIM_HIP(hipMemsetAsync(w.scal, 0, (size_t)nb * RS_SCAL * sizeof(float), st));
    SLIM_HIP(hipMemsetAsync(info, 0, (size_t)nb * sizeof(int32_t), st));
    auto gemm = [&](const float* A, int64_t sA, int lda, int M_, const float* B, int64_t sB, int ldb, int b_kmajor, int N_, int Kd,
                    float* C, int64_t sC, int ldc) {
        F32GemmArgs g;
        g.A = A, g.B = B, g.C = C;
        g.M = M_, g.N = N_, g.Kd = Kd;
        g.lda = lda, g.ldb = ldb, g.ldc = ldc;
        g.a_kmajor = 0, g.b_kmajor = b_kmajor;
        g.batch = nb;
        g.sA = sA, g.sB = sB, g.sC = sC;
        return g;
    };
    {
        // ---- b >= lambda_max(M)
        {
            SLIM_PROF("rs_infnorm", st, 4.0 * nb * l * l, 2.0 * nb * l * l);
            rs_infnorm_kernel<<<dim3((unsigned)((l + 15) / 16), (unsigned)nb), 256, 0, st>>>(M, l, w.scal, SC_S);
            SLIM_LAUNCH_CHECK();
        }
        {
            F32GemmArgs g = gemm(M, sM, l, l, M, sM, l, 1, l, l, w.P[0], sM, l);
            g.inv_sq_scale = w.scal + SC_S;
            g.iss_stride = RS_SCAL;
            CHK(slim_f32gemm_launch(g, st));
            g = gemm(w.P[0], sM, l, l, w.P[0], sM, l, 1, l, l, w.P[1], sM, l);
            CHK(slim_f32gemm_launch(g, st));
            g = gemm(w.P[1], sM, l, l, w.P[1], sM, l, 1, l, l, w.P[0], sM, l);
            CHK(slim_f32gemm_launch(g, st));
        }
        {
            SLIM_PROF("rs_infnorm", st, 4.0 * nb * l * l, 2.0 * nb * l * l);
            rs_infnorm_kernel<<<dim3((unsigned)((l + 15) / 16), (unsigned)nb), 256, 0, st>>>(w.P[0], l, w.scal, SC_N3);
            SLIM_LAUNCH_CHECK();
        }
    }
    const int lds_bounds = 2 * RS_MAX_C * RS_MAX_C * (int)sizeof(float);
    SLIM_HIP(slim_allow_dynamic_lds((const void*)rs_bounds_kernel, lds_bounds));
    // a from the leading principal block of M (and b, the estimate of the largest eigenvalue)
    {
        SLIM_PROF("rs_bounds", st, 0.0, 0.0);
        rs_bounds_kernel<<<nb, 1024, lds_bounds, st>>>(M, sM, l, keep, 1, degs[0], w.scal);
        SLIM_LAUNCH_CHECK();
    }
    if (c_pad > 0) {
        // ---- the block
        {
            SLIM_PROF("rs_init", st, 4.0 * nb * c_pad * l, 0.0);
            rs_init_kernel<<<dim3((unsigned)((l + 255) / 256), (unsigned)c_pad, (unsigned)nb), 256, 0, st>>>(w.Y[0], l, c_pad, keep);
            SLIM_LAUNCH_CHECK();
        }
        const dim3 cheb_grid((unsigned)((l + RS_PANEL - 1) / RS_PANEL), (unsigned)nb);
        const int cheb_lds = (RS_PANEL * (l + 4) + 4 * RS_RB * 256) * (int)sizeof(float);
        SLIM_HIP(slim_allow_dynamic_lds((const void*)rs_cheb_kernel, cheb_lds));
        int cur = 0;
        for (int r = 0; r < ndeg; ++r) {
            // Chebyshev recurrence of degree degs[r] on the rows of Y[cur]
            int y0 = cur, y1 = (cur + 1) % 3;
            if (degs[r] % 4 == 0 && g_stride4) {
                // T_(m+n) = 2 T_m T_n - T_|m-n|: with P = T_4(X) formed once per round (two l x l products: T_2 = 2 X^2 - I,
                // T_4 = 2 T_2^2 - I) the blocks Z_j = T_(4j)(X) Y obey the same recurrence, Z_(j+1) = 2 Z_j P - Z_(j-1),
                // in a quarter of the launches (each is latency-bound: a skinny product at the launch floor)
                F32GemmArgs g = gemm(M, sM, l, l, M, sM, l, 1, l, l, w.P[1], sM, l);
                g.coef = w.scal + SC_T2, g.coef_stride = RS_SCAL, g.D1 = M, g.sD1 = sM, g.coef_diag = 1;
                CHK(slim_f32gemm_launch(g, st));
                g = gemm(w.P[1], sM, l, l, w.P[1], sM, l, 1, l, l, w.P[0], sM, l);
                g.coef = w.scal + SC_TWO, g.coef_stride = RS_SCAL, g.coef_diag = 1;
                CHK(slim_f32gemm_launch(g, st));
                for (int k = 1; k <= degs[r] / 4; ++k) {
                    SLIM_PROF("rs_cheb", st, 4.0 * nb * ((double)l * l + 3.0 * c_pad * l), 2.0 * nb * c_pad * (double)l * l);
                    if (k == 1) {
                        rs_cheb_kernel<<<cheb_grid, 256, cheb_lds, st>>>(w.Y[y0], w.P[0], l, c_pad, w.scal + SC_ONE, RS_SCAL, nullptr, nullptr, w.Y[y1]);
                        SLIM_LAUNCH_CHECK();
                    } else {
                        const int y2 = 3 - y0 - y1;
                        rs_cheb_kernel<<<cheb_grid, 256, cheb_lds, st>>>(w.Y[y1], w.P[0], l, c_pad, w.scal + SC_TWO, RS_SCAL, nullptr, w.Y[y0], w.Y[y2]);
                        SLIM_LAUNCH_CHECK();
                        y0 = y1;
                        y1 = y2;
                    }
                }
            } else {
                {
                    SLIM_PROF("rs_cheb", st, 4.0 * nb * ((double)l * l + 3.0 * c_pad * l), 2.0 * nb * c_pad * (double)l * l);
                    rs_cheb_kernel<<<cheb_grid, 256, cheb_lds, st>>>(w.Y[y0], M, l, c_pad, w.scal + SC_CA, RS_SCAL, w.Y[y0], nullptr, w.Y[y1]);
                    SLIM_LAUNCH_CHECK();
                }
                for (int k = 2; k <= degs[r]; ++k) {
                    const int y2 = 3 - y0 - y1;
                    {
                        SLIM_PROF("rs_cheb", st, 4.0 * nb * ((double)l * l + 3.0 * c_pad * l), 2.0 * nb * c_pad * (double)l * l);
                        rs_cheb_kernel<<<cheb_grid, 256, cheb_lds, st>>>(w.Y[y1], M, l, c_pad, w.scal + SC_CB, RS_SCAL, w.Y[y1], w.Y[y0], w.Y[y2]);
                        SLIM_LAUNCH_CHECK();
                    }
                    y0 = y1;
                    y1 = y2;
                }
            }
            {
                SLIM_PROF("rs_qr", st, 8.0 * nb * c_pad * l, 4.0 * nb * (double)c_pad * c_pad * l);
                rs_qr_kernel<<<nb, 1024, 0, st>>>(w.Y[y1], l, c_pad, keep, w.V, w.tau);
                SLIM_LAUNCH_CHECK();
            }
            if (r + 1 == ndeg) break;
            // orthonormal basis of the block, its compression of M, the tightened bound
            const int q = (y1 + 1) % 3, t1 = (y1 + 2) % 3;
            {
                SLIM_PROF("rs_apply", st, 0.0, 0.0);
                rs_apply_kernel<<<dim3((unsigned)((c_pad + RS_WAVES - 1) / RS_WAVES), (unsigned)nb), 1024, 0, st>>>(w.V, w.tau, l, c_pad, keep, 1,
                                                                                                                0, w.Y[q]);
                SLIM_LAUNCH_CHECK();
            }
            {
                SLIM_PROF("rs_cheb", st, 4.0 * nb * ((double)l * l + 3.0 * c_pad * l), 2.0 * nb * c_pad * (double)l * l);
                rs_cheb_kernel<<<cheb_grid, 256, cheb_lds, st>>>(w.Y[q], M, l, c_pad, w.scal + SC_ONE, RS_SCAL, nullptr, nullptr, w.Y[t1]);
                SLIM_LAUNCH_CHECK();
            }
            CHK(slim_f32gemm_launch(gemm(w.Y[t1], sY, l, c_pad, w.Y[q], sY, l, 0, c_pad, l, w.Bm, (int64_t)c_pad * c_pad, c_pad), st));
            {
                SLIM_PROF("rs_bounds", st, 0.0, 0.0);
                rs_bounds_kernel<<<nb, 1024, lds_bounds, st>>>(w.Bm, (int64_t)c_pad * c_pad, c_pad, keep, 0, degs[r + 1], w.scal);
                SLIM_LAUNCH_CHECK();
            }
            cur = q;
        }
    }
    // ---- the complement basis in the eigen-solver's layout, its Rayleigh quotients
    {
        SLIM_PROF("rs_apply", st, 0.0, 0.0);
        rs_apply_kernel<<<dim3((unsigned)((top + RS_WAVES - 1) / RS_WAVES), (unsigned)nb), 1024, 0, st>>>(w.V, w.tau, l, c_pad, keep, 0, top,
                                                                                                      vecs);
        SLIM_LAUNCH_CHECK();
    }
    CHK(slim_f32gemm_launch(gemm(vecs + (int64_t)(l - top) * l, sM, l, top, M, sM, l, 1, l, l, w.P[0], (int64_t)top * l, l), st));
    {
        SLIM_PROF("rs_evals", st, 8.0 * nb * top * l, 2.0 * nb * top * l);
        rs_evals_kernel<<<dim3((unsigned)((top + 3) / 4), (unsigned)nb), 256, 0, st>>>(w.P[0], vecs, l, top, keep, w.scal, evals, info);
        SLIM_LAUNCH_CHECK();
    }
    return SLIM_OK;
}

extern "C" int slim_ritz_split_f32_batched(const float* M, int l, int batch, int top, const int32_t* keep, const int32_t* degrees,
                                           int n_rounds, float* vecs, float* evals, int32_t* info, void* workspace,
                                           int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(M && keep && degrees && vecs && evals && info && workspace, "slim_ritz_split_f32_batched: null pointer");
    SLIM_REQUIRE(batch >= 1 && top >= 4 && top % 4 == 0 && top <= l && n_rounds >= 1 && n_rounds <= 8, "slim_ritz_split_f32_batched: bad sizes");
    int drop_max = 0;
    for (int b = 0; b < batch; ++b) {
        SLIM_REQUIRE(keep[b] >= 1 && keep[b] <= top, "slim_ritz_split_f32_batched: keep must lie in [1, top]");
        drop_max = l - keep[b] > drop_max ? l - keep[b] : drop_max;
    }
    SLIM_REQUIRE(slim_ritz_split_supported(l, drop_max), "slim_ritz_split_f32_batched: l <= 576 (a multiple of 8) and at most 80 dropped directions");
    int degs[8];
    for (int r = 0; r < n_rounds; ++r) {
        SLIM_REQUIRE(degrees[r] >= 1 && degrees[r] <= 64, "slim_ritz_split_f32_batched: a round's degree must lie in [1, 64]");
        degs[r] = degrees[r];
    }
    SLIM_REQUIRE(workspace_bytes >= slim_ritz_split_workspace_bytes(l, drop_max, batch), "slim_ritz_split_f32_batched: workspace too small");
    SLIM_REQUIRE(((uintptr_t)M % 16) == 0 && ((uintptr_t)vecs % 16) == 0 && ((uintptr_t)workspace % 256) == 0,
                 "slim_ritz_split_f32_batched: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < batch; b0 += RS_MAX_BATCH) {
        const int nb = batch - b0 < RS_MAX_BATCH ? batch - b0 : RS_MAX_BATCH;
        CHK(rs_run(M + (int64_t)b0 * l * l, l, nb, top, keep + b0, degs, n_rounds, vecs + (int64_t)b0 * l * l, evals + (int64_t)b0 * l,
                   info + b0, (char*)workspace, st));
    }
    return SLIM_OK;
}
