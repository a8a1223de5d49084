// fp32 GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) for the small problems of the low-rank fit whose operands
// cannot be rounded to bf16: the back-product of the symmetric eigen-solver (eigenvectors of the tridiagonal matrix times
// the accumulated Householder reflectors, n x n x top) and l x l products of fp32 polynomial iterations.
//
// 64 x 64 output tile per 256-thread workgroup (2 x 2 waves, 32 x 32 per wave as 2 x 2 fragments of 16 x 16), reduction
// in steps of 32 through LDS, operand tiles fetched two steps ahead into registers.  Either operand may be stored with
// the reduction index as its row ("k-major", rows of the tile are contiguous along the output index) or as its column
// ("i-major", contiguous along the reduction); the LDS image keeps the global orientation, with row strides chosen so
// that the fragment reads are conflict-free: a fragment read has lanes 0-15 on consecutive output indices and lane
// groups 0-3 on consecutive k, i.e. k-major rows 80 floats apart put the two k of a half-wave on disjoint halves of the
// 32 banks, and i-major rows 34 floats apart put even and odd k on even and odd banks.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include "f32gemm.hpp"

#define FT 64
#define FBK 32
#define LDK 80
#define LDI 34
#define TILE_FLOATS (FBK * LDK)            // 2560 >= FT * LDI = 2176
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <bool KMAJOR>
__device__ __forceinline__ void f32_load(const float* __restrict__ P, int ld, int extent, int Kd, int i0, int k0, f32x4 (&r)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        r[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (KMAJOR) {
            const int k = k0 + t / 16 + 16 * it, i = i0 + (t % 16) * 4;
            if (k < Kd && i < extent) r[it] = *(const f32x4*)(P + (int64_t)k * ld + i);
        } else {
            const int i = i0 + t / 8 + 32 * it, k = k0 + (t % 8) * 4;
            if (i < extent && k < Kd) r[it] = *(const f32x4*)(P + (int64_t)i * ld + k);
        }
    }
}

template <bool KMAJOR>
__device__ __forceinline__ void f32_store(float* tile, const f32x4 (&r)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (KMAJOR) {
            *(f32x4*)(tile + (t / 16 + 16 * it) * LDK + (t % 16) * 4) = r[it];
        } else {
            float* p = tile + (t / 8 + 32 * it) * LDI + (t % 8) * 4;      // 8-byte aligned (LDI even)
            *(f32x2*)p = f32x2{r[it][0], r[it][1]};
            *(f32x2*)(p + 2) = f32x2{r[it][2], r[it][3]};
        }
    }
}

// element (i, k) of an operand tile
template <bool KMAJOR>
__device__ __forceinline__ float f32_frag(const float* tile, int i, int k) {
    return KMAJOR ? tile[k * LDK + i] : tile[i * LDI + k];
}

template <bool AK, bool BK_>
__global__ __launch_bounds__(256) void f32_gemm_kernel(F32GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE_FLOATS];      // [stage][A | B]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tiles_n = (g.N + FT - 1) / FT;
    const int row0 = ((int)blockIdx.x / tiles_n) * FT, col0 = ((int)blockIdx.x % tiles_n) * FT;
    {
        const int64_t p = blockIdx.z / g.nA, w = blockIdx.z % g.nA;
        g.A += p * g.sA + w * g.sAw;
        g.B += p * g.sB;
        g.C += p * g.sC + w * g.sCw;
        if (g.C16) g.C16 += p * g.sC + w * g.sCw;
        if (g.poly_G) g.poly_G += p * g.sG;
        if (g.coef) g.coef += p * g.coef_stride;
        if (g.D1) g.D1 += p * g.sD1;
        if (g.D2) g.D2 += p * g.sD2;
        if (g.inv_sq_scale) {
            const float x = g.inv_sq_scale[p * g.iss_stride];
            g.alpha = x > 0.f ? g.alpha / (x * x) : 0.f;
        }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (g.Kd + FBK - 1) / FBK;
    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
    auto As = [&](int s) { return smem + s * 2 * TILE_FLOATS; };
    auto Bs = [&](int s) { return smem + s * 2 * TILE_FLOATS + TILE_FLOATS; };
    auto compute = [&](const float* at, const float* bt) {
#pragma unroll
        for (int kc = 0; kc < FBK / 4; ++kc) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = f32_frag<AK>(at, wr * 32 + i * 16 + fr, kc * 4 + fq);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = f32_frag<BK_>(bt, wc * 32 + j * 16 + fr, kc * 4 + fq);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    f32_load<AK>(g.A, g.lda, g.M, g.Kd, row0, 0, ra0);
    f32_load<BK_>(g.B, g.ldb, g.N, g.Kd, col0, 0, rb0);
    if (nk > 1) {
        f32_load<AK>(g.A, g.lda, g.M, g.Kd, row0, FBK, ra1);
        f32_load<BK_>(g.B, g.ldb, g.N, g.Kd, col0, FBK, rb1);
    }
    f32_store<AK>(As(0), ra0);
    f32_store<BK_>(Bs(0), rb0);
    __syncthreads();
    // two K-steps per iteration so that the two register sets alternate statically: while step kt is multiplied out of
    // LDS stage 0, step kt + 1 sits in registers (fetched one iteration ago) and step kt + 2 is being fetched
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk) {
            f32_load<AK>(g.A, g.lda, g.M, g.Kd, row0, (kt + 2) * FBK, ra0);
            f32_load<BK_>(g.B, g.ldb, g.N, g.Kd, col0, (kt + 2) * FBK, rb0);
        }
        compute(As(0), Bs(0));
        if (kt + 1 >= nk) break;
        f32_store<AK>(As(1), ra1);
        f32_store<BK_>(Bs(1), rb1);
        __syncthreads();
        if (kt + 3 < nk) {
            f32_load<AK>(g.A, g.lda, g.M, g.Kd, row0, (kt + 3) * FBK, ra1);
            f32_load<BK_>(g.B, g.ldb, g.N, g.Kd, col0, (kt + 3) * FBK, rb1);
        }
        compute(As(1), Bs(1));
        if (kt + 2 < nk) {
            f32_store<AK>(As(0), ra0);
            f32_store<BK_>(Bs(0), rb0);
        }
        __syncthreads();
    }
    // epilogue: the tile through LDS ([64][65] fp32), rows leave as 16-byte pieces
    __syncthreads();
    float* T = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(wr * 32 + i * 16 + fq * 4 + r) * 65 + wc * 32 + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int c = (t % 16) * 4, col = col0 + c;
    if (col >= g.N) return;
    float gin[4][4];
    if (g.poly_G) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = row0 + t / 16 + 16 * it;
#pragma unroll
            for (int u = 0; u < 4; ++u) gin[it][u] = 0.f;
            if (row < g.M && g.pb != 0.f) {
                const f32x4 v = *(const f32x4*)(g.poly_G + (int64_t)row * g.N + col);
#pragma unroll
                for (int u = 0; u < 4; ++u) gin[it][u] = v[u];
            }
        }
    }
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (g.coef) c0 = g.coef[0], c1 = g.coef[1], c2 = g.coef[2];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = t / 16 + 16 * it, row = row0 + r;
        if (row >= g.M) continue;
        float v[4];
        f32x4 d1 = f32x4{0.f, 0.f, 0.f, 0.f}, d2 = d1;
        if (g.coef && g.D1) d1 = *(const f32x4*)(g.D1 + (int64_t)row * g.ldc + col);
        if (g.coef && g.D2 && !g.coef_diag) d2 = *(const f32x4*)(g.D2 + (int64_t)row * g.ldc + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a = T[r * 65 + c + u];
            if (g.coef) v[u] = c0 * a + c1 * d1[u] + c2 * (g.coef_diag ? (row == col + u ? 1.f : 0.f) : d2[u]);
            else v[u] = g.poly_G ? g.pb * gin[it][u] + g.pq * a + (row == col + u ? g.pa : 0.f) : g.alpha * a;
        }
        *(f32x4*)(g.C + (int64_t)row * g.ldc + col) = f32x4{v[0], v[1], v[2], v[3]};
        if (g.C16)
            *(u16x4*)(g.C16 + (int64_t)row * g.ldc + col) = u16x4{f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
    }
}

int slim_f32gemm_launch(const F32GemmArgs& g, void* stream) {
    SLIM_REQUIRE(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.Kd > 0, "slim_gemm_f32: bad arguments");
    SLIM_REQUIRE(g.M % 4 == 0 && g.N % 4 == 0 && g.Kd % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && g.ldc % 4 == 0,
                 "slim_gemm_f32: sizes and leading dimensions must be multiples of 4");
    SLIM_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0 && ((uintptr_t)g.C % 16) == 0 &&
                     g.sA % 4 == 0 && g.sAw % 4 == 0 && g.sB % 4 == 0 && g.sC % 4 == 0 && g.sCw % 4 == 0,
                 "slim_gemm_f32: operands and strides must be 16-byte aligned");
    SLIM_REQUIRE(g.batch >= 1 && g.nA >= 1 && (int64_t)g.batch * g.nA <= 65535, "slim_gemm_f32: bad batch");
    SLIM_REQUIRE(!g.poly_G || (g.M == g.N && g.sG % 4 == 0), "slim_gemm_f32: the polynomial epilogue needs a square product");
    const int64_t tiles = (int64_t)((g.M + FT - 1) / FT) * ((g.N + FT - 1) / FT);
    SLIM_REQUIRE(tiles < ((int64_t)1 << 31), "slim_gemm_f32: too many tiles");
    const double np = (double)g.batch * g.nA;
    SLIM_PROF("gemm_f32", stream, 4.0 * ((double)(g.M + g.N) * g.Kd + (double)g.M * g.N) * np, 2.0 * g.M * (double)g.N * g.Kd * np);
    const dim3 grid((unsigned)tiles, 1, (unsigned)(g.batch * g.nA));
    hipStream_t st = (hipStream_t)stream;
    if (g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((f32_gemm_kernel<true, true>), grid, dim3(256), 0, st, g);
    else if (g.a_kmajor) hipLaunchKernelGGL((f32_gemm_kernel<true, false>), grid, dim3(256), 0, st, g);
    else if (g.b_kmajor) hipLaunchKernelGGL((f32_gemm_kernel<false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((f32_gemm_kernel<false, false>), grid, dim3(256), 0, st, g);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// C [M, N] = alpha * op(A) op(B) in fp32 on the matrix cores; see slim_hip.h
extern "C" int slim_gemm_f32(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, int M, int N,
                             int Kd, float alpha, float* C, int64_t ldc, void* stream) {
    F32GemmArgs g;
    g.A = A;
    g.B = B;
    g.C = C;
    g.M = M;
    g.N = N;
    g.Kd = Kd;
    g.lda = (int)lda;
    g.ldb = (int)ldb;
    g.ldc = (int)ldc;
    g.a_kmajor = a_kmajor;
    g.b_kmajor = b_kmajor;
    g.alpha = alpha;
    return slim_f32gemm_launch(g, stream);
}
