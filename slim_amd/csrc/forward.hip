// Forward pass of a SLiM-compressed Linear from its packed form:
//
//     y = bf16( bf16(x W^T + b) + bf16( t R_eff ) ),    t = bf16( x L_eff )
//
// where W is the 2:4-sparse, 4-bit, per-tensor-scaled weight the compression leaves in
// `weight.data` (slim/prune.py:281-301, slim/lora.py:199-204) and L_eff / R_eff are the adapters
// as the forward hook applies them (slim/prune.py:303-321).  The rounding points are those of the
// reference's emulated forward: F.linear rounds its fp32 accumulation (+ bias) to bf16, each
// adapter matmul rounds to bf16, and `output += xlr` is a bf16 addition.
//
// Packed weight.  Per group of 4 consecutive input channels two values survive the 2:4 mask:
// 8 bits of codes (two signed 4-bit integers, lower input channel in the low nibble) and 4 bits
// of metadata (two 2-bit positions), 0.375 bytes per dense element instead of 2.  Both arrays
// are stored tile-major so that the 256 threads of a workgroup read them fully coalesced: the
// weight is cut into tiles of 128 output rows x 64 input channels, a tile into two 32-channel
// halves, and
//     vals[tile][half][row][8 bytes]   meta[tile][half][row][4 bytes],   tile = tn * (K/64) + kt.
// A thread expands its 32 channels (8 groups) into bf16 through two small LDS tables -- a
// 256-entry table of code-byte -> two bf16 values (built from the 16 dequantised levels
// bf16(c / sf), so the expanded tile is bit-identical to the reference's dequantised weight)
// and a 16-entry table of byte-permute selectors that drop the two values at their positions --
// and writes them to the LDS tile the MFMA loop consumes.  HBM traffic per weight element falls
// from 2 bytes to 0.375.
//
// Small batches (decode) are bound by streaming the packed weight, large ones by MFMA; the
// kernel is templated on the x-tile height (16 .. 128 rows) and can split the reduction over
// blockIdx.y (partial sums in an fp32 slab, combined by a second small kernel) so that a
// 4096 x 4096 layer at batch 1 still fills 256 CUs.
#include "common.hpp"
#include "../../include/slim_hip.h"

#define FBN 128
#define FBK 64
#define FSTRIDE (FBK + 8)

// ---------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int64_t packed_unit(int64_t row, int64_t k32, int64_t K) {
    // index of the (row, 32-channel chunk) unit in tile-major order
    const int64_t tn = row >> 7, r = row & 127, kt = k32 >> 1, half = k32 & 1;
    return ((tn * (K >> 6) + kt) * 2 + half) * 128 + r;
}

// one thread per (row, 32-channel chunk); status bit 0: a group with more than two non-zeros,
// bit 1: a value that is not bf16(c / sf) for an integer c in [-8, 7]
__global__ void pack24_kernel(const bf16_t* __restrict__ W, const float* __restrict__ sf_ptr, int64_t N, int64_t K,
                              uint8_t* __restrict__ vals, uint8_t* __restrict__ meta, int* __restrict__ status) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t chunks = K >> 5;
    if (u >= N * chunks) return;
    const int64_t row = u / chunks, k32 = u - row * chunks;
    const float sf = *sf_ptr;
    const bf16_t* w = W + row * K + k32 * 32;
    uint32_t v[2] = {0u, 0u}, m = 0u;
    int bad = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        int pos[2] = {0, 1}, code[2] = {0, 0}, n = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t raw = w[g * 4 + e];
            if ((raw & 0x7fffu) == 0) continue;
            const float f = bf16_to_f32(raw);
            const float c = rintf(f * sf);
            if (!(c >= -8.f && c <= 7.f) || f32_to_bf16(c / sf) != raw) bad |= 2;
            if (n < 2) {
                pos[n] = e;
                code[n] = (int)c;
            }
            ++n;
        }
        if (n > 2) bad |= 1;
        if (n == 1) pos[1] = pos[0] == 3 ? 2 : 3, code[1] = 0;     // any other position, code 0
        if (n == 1 && pos[1] < pos[0]) {                             // keep positions ascending
            const int p = pos[0], c = code[0];
            pos[0] = pos[1], code[0] = code[1], pos[1] = p, code[1] = c;
        }
        const uint32_t byte = (uint32_t)(code[0] & 15) | ((uint32_t)(code[1] & 15) << 4);
        v[g >> 2] |= byte << (8 * (g & 3));
        m |= ((uint32_t)pos[0] | ((uint32_t)pos[1] << 2)) << (4 * g);
    }
    const int64_t o = packed_unit(row, k32, K);
    *(uint2*)(vals + o * 8) = make_uint2(v[0], v[1]);
    *(uint32_t*)(meta + o * 4) = m;
    if (bad) atomicOr(status, bad);
}

__global__ void unpack24_kernel(const uint8_t* __restrict__ vals, const uint8_t* __restrict__ meta,
                                const float* __restrict__ sf_ptr, int64_t N, int64_t K, bf16_t* __restrict__ W) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t chunks = K >> 5;
    if (u >= N * chunks) return;
    const int64_t row = u / chunks, k32 = u - row * chunks;
    const float sf = *sf_ptr;
    const int64_t o = packed_unit(row, k32, K);
    const uint2 v = *(const uint2*)(vals + o * 8);
    const uint32_t m = *(const uint32_t*)(meta + o * 4);
    bf16_t* w = W + row * K + k32 * 32;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const uint32_t byte = ((g < 4 ? v.x : v.y) >> (8 * (g & 3))) & 0xffu;
        const uint32_t nib = (m >> (4 * g)) & 0xfu;
        const int c0 = ((int)(byte << 28)) >> 28, c1 = ((int)(byte << 24)) >> 28;
        bf16_t out[4] = {0, 0, 0, 0};
        out[nib & 3] = f32_to_bf16((float)c0 / sf);
        out[nib >> 2] = f32_to_bf16((float)c1 / sf);
        // a zero code dequantises to +0; (float)0 / sf is +0 for sf > 0
#pragma unroll
        for (int e = 0; e < 4; ++e) w[g * 4 + e] = out[e];
    }
}

extern "C" int slim_packed24_bytes(int64_t N, int64_t K, int64_t* vals_bytes, int64_t* meta_bytes) {
    SLIM_REQUIRE(N > 0 && K > 0 && N % 128 == 0 && K % 64 == 0,
                 "slim_packed24_bytes: N must be a multiple of 128 and K of 64");
    if (vals_bytes) *vals_bytes = N * K / 4;
    if (meta_bytes) *meta_bytes = N * K / 8;
    return SLIM_OK;
}

extern "C" int slim_pack24_int4(const void* W, const float* sf, int64_t N, int64_t K, void* vals, void* meta,
                                int32_t* status, void* stream) {
    SLIM_REQUIRE(W && sf && vals && meta && status, "slim_pack24_int4: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && N % 128 == 0 && K % 64 == 0,
                 "slim_pack24_int4: N must be a multiple of 128 and K of 64");
    SLIM_PROF("pack24", stream, (double)N * K * 2.375);
    SLIM_HIP(hipMemsetAsync(status, 0, sizeof(int), (hipStream_t)stream));
    const int64_t units = N * (K >> 5);
    pack24_kernel<<<(unsigned)ceil_div64(units, 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)W, sf, N, K, (uint8_t*)vals, (uint8_t*)meta, status);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_unpack24_int4(const void* vals, const void* meta, const float* sf, int64_t N, int64_t K, void* W,
                                  void* stream) {
    SLIM_REQUIRE(W && sf && vals && meta, "slim_unpack24_int4: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && N % 128 == 0 && K % 64 == 0,
                 "slim_unpack24_int4: N must be a multiple of 128 and K of 64");
    SLIM_PROF("unpack24", stream, (double)N * K * 2.375);
    const int64_t units = N * (K >> 5);
    unpack24_kernel<<<(unsigned)ceil_div64(units, 256), 256, 0, (hipStream_t)stream>>>(
        (const uint8_t*)vals, (const uint8_t*)meta, sf, N, K, (bf16_t*)W);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
    // main product x [M, K] * W^T, W packed (vals == nullptr: no main product)
    const bf16_t* x;
    int64_t ldx;
    int M, N, K;
    const uint8_t* vals;
    const uint8_t* meta;
    const float* sf;       // device scalar: the dequantised levels are bf16(c / sf)
    const bf16_t* bias;    // [N] or nullptr
    // second product a [M, Ka] * Bt[N, Ka]^T with bf16 operands (Bt == nullptr: none); every element of
    // `a` is replaced by bf16(a / a_div) on load when a_div != 1 (the hook's division by sqrt(rank))
    const bf16_t* a;
    int64_t lda;
    const bf16_t* Bt;
    int64_t ldb;
    int Ka;
    float a_div;
    bf16_t* y;
    int64_t ldy;
    // split reduction: blockIdx.y < splits_main covers main K-steps, the rest cover the second product
    int splits_main, splits_ad, ksteps_main, ksteps_ad;
    float* slab;           // [splits_main + splits_ad][M][N]
};

template <int TBM>
__device__ __forceinline__ void load_a_tile(const bf16_t* __restrict__ P, int64_t ld, int rows, int Kd, int row0, int k0,
                                            float div, u16x8 (&r)[(TBM * 8 + 255) / 256]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < (TBM * 8 + 255) / 256; ++i) {
        const int c = t + 256 * i;
        if (TBM * 8 < 256 && c >= TBM * 8) break;
        int row = row0 + (c >> 3);
        row = row < rows ? row : rows - 1;
        const int k = k0 + (c & 7) * 8;
        if (k < Kd) {
            r[i] = *(const u16x8*)(P + (int64_t)row * ld + k);
            if (div != 1.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[i][j] = f32_to_bf16(bf16_to_f32(r[i][j]) / div);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[i][j] = 0;
        }
    }
}

template <int TBM>
__device__ __forceinline__ void store_a_tile(bf16_t* lds, const u16x8 (&r)[(TBM * 8 + 255) / 256]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < (TBM * 8 + 255) / 256; ++i) {
        const int c = t + 256 * i;
        if (TBM * 8 < 256 && c >= TBM * 8) break;
        *(u16x8*)(lds + (c >> 3) * FSTRIDE + (c & 7) * 8) = r[i];
    }
}

__device__ __forceinline__ void load_b_tile(const bf16_t* __restrict__ P, int64_t ld, int rows, int Kd, int row0, int k0,
                                            u16x8 (&r)[4]) {
    const int t = threadIdx.x;
    const int kc = (t & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = row0 + (t >> 3) + 32 * i;
        row = row < rows ? row : rows - 1;
        const int k = k0 + kc;
        if (k < Kd) r[i] = *(const u16x8*)(P + (int64_t)row * ld + k);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[i][j] = 0;
        }
    }
}

__device__ __forceinline__ void store_b_tile(bf16_t* lds, const u16x8 (&r)[4]) {
    const int t = threadIdx.x;
    const int kc = (t & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u16x8*)(lds + ((t >> 3) + 32 * i) * FSTRIDE + kc) = r[i];
}

// expand one thread's 32 packed channels into the LDS B tile
__device__ __forceinline__ void store_packed_tile(bf16_t* lds, uint2 v, uint32_t m, const uint32_t* __restrict__ pair_lut,
                                                  const uint2* __restrict__ sel_lut) {
    const int t = threadIdx.x;
    const int row = t & 127, half = t >> 7;
    uint32_t out[16];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const uint32_t byte = ((g < 4 ? v.x : v.y) >> (8 * (g & 3))) & 0xffu;
        const uint32_t pair = pair_lut[byte];
        const uint2 sel = sel_lut[(m >> (4 * g)) & 0xfu];
        out[2 * g] = __builtin_amdgcn_perm(pair, pair, sel.x);
        out[2 * g + 1] = __builtin_amdgcn_perm(pair, pair, sel.y);
    }
    uint4* dst = (uint4*)(lds + row * FSTRIDE + half * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
}

template <int WAVES_M, int FR_M>
__global__ __launch_bounds__(256) void linear24_kernel(FwdArgs g) {
    constexpr int TBM = WAVES_M * FR_M * 16;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int FR_N = FBN / (WAVES_N * 16);
    constexpr int NA = (TBM * 8 + 255) / 256;
    constexpr int BUF = (TBM + FBN) * FSTRIDE;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    auto As = [&](int b) { return smem + b * BUF; };
    auto Bs = [&](int b) { return smem + b * BUF + TBM * FSTRIDE; };
    uint32_t* pair_lut = (uint32_t*)(smem + 2 * BUF);
    uint2* sel_lut = (uint2*)(pair_lut + 256);

    // consecutive workgroups (one XCD's share of them, see below) walk down the x tiles of one weight
    // column panel, so the packed panel is read from HBM once and then served by that XCD's L2
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
    const int tiles_m = (g.M + TBM - 1) / TBM;
    const int tn = wgid / tiles_m, tm = wgid % tiles_m;
    const int row0 = tm * TBM, col0 = tn * FBN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;

    if (g.vals) {
        const float sf = *g.sf;
        const int t = threadIdx.x;
        const int c0 = ((int)((uint32_t)t << 28)) >> 28, c1 = ((int)((uint32_t)t << 24)) >> 28;
        pair_lut[t] = (uint32_t)f32_to_bf16((float)c0 / sf) | ((uint32_t)f32_to_bf16((float)c1 / sf) << 16);
        if (t < 16) {
            // element e of the group takes bytes (0,1) of the pair if it is the first kept position,
            // (2,3) if the second, and the constant-zero selector 0x0c otherwise
            const int p0 = t & 3, p1 = t >> 2;
            uint32_t s[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = (e == p0) ? 0x0100u : (e == p1) ? 0x0302u : 0x0c0cu;
            sel_lut[t] = make_uint2(s[0] | (s[1] << 16), s[2] | (s[3] << 16));
        }
        __syncthreads();
    }

    f32x4 acc[FR_M][FR_N];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < FR_M; ++i)
#pragma unroll
            for (int j = 0; j < FR_N; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // All fragment reads of a K-step are issued before its MFMAs (and pinned there): left to itself the compiler
    // interleaves four read-wait-MFMA rounds to save registers, and with one or two waves per SIMD each round's LDS
    // latency is exposed.
    auto mma = [&](int cur) {
        const bf16_t* a_base = As(cur) + (wr * FR_M * 16 + fr) * FSTRIDE + fq * 8;
        const bf16_t* b_base = Bs(cur) + (wc * FR_N * 16 + fr) * FSTRIDE + fq * 8;
        bf16x8 af[FBK / 32][FR_M], bfr[FBK / 32][FR_N];
#pragma unroll
        for (int ks = 0; ks < FBK / 32; ++ks) {
#pragma unroll
            for (int i = 0; i < FR_M; ++i) af[ks][i] = *(const bf16x8*)(a_base + i * 16 * FSTRIDE + ks * 32);
#pragma unroll
            for (int j = 0; j < FR_N; ++j) bfr[ks][j] = *(const bf16x8*)(b_base + j * 16 * FSTRIDE + ks * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < FBK / 32; ++ks)
#pragma unroll
            for (int i = 0; i < FR_M; ++i)
#pragma unroll
                for (int j = 0; j < FR_N; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // main product over K-steps [kb, ke)
    auto run_main = [&](int kb, int ke) {
        if (kb >= ke) return;
        u16x8 ra[NA];
        uint2 pv;
        uint32_t pm;
        const int64_t ktiles = g.K / FBK;
        auto load_packed = [&](int kt) {
            const int64_t unit = (((int64_t)tn * ktiles + kt) * 2 + (threadIdx.x >> 7)) * 128 + (threadIdx.x & 127);
            pv = *(const uint2*)(g.vals + unit * 8);
            pm = *(const uint32_t*)(g.meta + unit * 4);
        };
        load_a_tile<TBM>(g.x, g.ldx, g.M, g.K, row0, kb * FBK, 1.f, ra);
        load_packed(kb);
        store_a_tile<TBM>(As(0), ra);
        store_packed_tile(Bs(0), pv, pm, pair_lut, sel_lut);
        __syncthreads();
        for (int kt = kb; kt < ke; ++kt) {
            const int cur = (kt - kb) & 1;
            if (kt + 1 < ke) {
                load_a_tile<TBM>(g.x, g.ldx, g.M, g.K, row0, (kt + 1) * FBK, 1.f, ra);
                load_packed(kt + 1);
            }
            mma(cur);
            if (kt + 1 < ke) {
                store_a_tile<TBM>(As(cur ^ 1), ra);
                store_packed_tile(Bs(cur ^ 1), pv, pm, pair_lut, sel_lut);
            }
            __syncthreads();
        }
    };
    // second product over its K-steps [kb, ke)
    auto run_ad = [&](int kb, int ke) {
        if (kb >= ke) return;
        u16x8 ra[NA], rb[4];
        load_a_tile<TBM>(g.a, g.lda, g.M, g.Ka, row0, kb * FBK, g.a_div, ra);
        load_b_tile(g.Bt, g.ldb, g.N, g.Ka, col0, kb * FBK, rb);
        store_a_tile<TBM>(As(0), ra);
        store_b_tile(Bs(0), rb);
        __syncthreads();
        for (int kt = kb; kt < ke; ++kt) {
            const int cur = (kt - kb) & 1;
            if (kt + 1 < ke) {
                load_a_tile<TBM>(g.a, g.lda, g.M, g.Ka, row0, (kt + 1) * FBK, g.a_div, ra);
                load_b_tile(g.Bt, g.ldb, g.N, g.Ka, col0, (kt + 1) * FBK, rb);
            }
            mma(cur);
            if (kt + 1 < ke) {
                store_a_tile<TBM>(As(cur ^ 1), ra);
                store_b_tile(Bs(cur ^ 1), rb);
            }
            __syncthreads();
        }
    };

    const int nk_main = g.vals ? g.K / FBK : 0;
    const int nk_ad = g.Bt ? (g.Ka + FBK - 1) / FBK : 0;

    if (g.slab) {
        // split reduction: this workgroup adds one slice of one of the two products into its slab
        const int z = blockIdx.y;
        if (z < g.splits_main) run_main(z * g.ksteps_main, min((z + 1) * g.ksteps_main, nk_main));
        else run_ad((z - g.splits_main) * g.ksteps_ad, min((z - g.splits_main + 1) * g.ksteps_ad, nk_ad));
        float* sl = g.slab + (int64_t)z * g.M * g.N;
#pragma unroll
        for (int i = 0; i < FR_M; ++i)
#pragma unroll
            for (int j = 0; j < FR_N; ++j) {
                const int col = col0 + wc * FR_N * 16 + j * 16 + fr;
                const int rbase = row0 + wr * FR_M * 16 + i * 16 + fq * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rbase + r < g.M) sl[(int64_t)(rbase + r) * g.N + col] = acc[i][j][r];
            }
        return;
    }

    run_main(0, nk_main);
    // first term rounded to bf16 (with the bias) and parked as packed halves while the accumulators are reused
    uint32_t parked[FR_M][FR_N][2];
    if (nk_main) {
#pragma unroll
        for (int i = 0; i < FR_M; ++i)
#pragma unroll
            for (int j = 0; j < FR_N; ++j) {
                const int col = col0 + wc * FR_N * 16 + j * 16 + fr;
                const float b = g.bias ? bf16_to_f32(g.bias[col]) : 0.f;
                bf16_t h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = f32_to_bf16(acc[i][j][r] + b);
                parked[i][j][0] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                parked[i][j][1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
            }
    }
    if (nk_ad) {
        if (nk_main) {
            zero_acc();
            __syncthreads();
        }
        run_ad(0, nk_ad);
    }
#pragma unroll
    for (int i = 0; i < FR_M; ++i)
#pragma unroll
        for (int j = 0; j < FR_N; ++j) {
            const int col = col0 + wc * FR_N * 16 + j * 16 + fr;
            const int rbase = row0 + wr * FR_M * 16 + i * 16 + fq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (rbase + r >= g.M) continue;
                bf16_t o;
                if (nk_main && nk_ad) {
                    const bf16_t mh = (bf16_t)((parked[i][j][r >> 1] >> (16 * (r & 1))) & 0xffffu);
                    o = f32_to_bf16(bf16_to_f32(mh) + round_bf16(acc[i][j][r]));
                } else if (nk_main) {
                    o = (bf16_t)((parked[i][j][r >> 1] >> (16 * (r & 1))) & 0xffffu);
                } else {
                    o = f32_to_bf16(acc[i][j][r]);
                }
                g.y[(int64_t)(rbase + r) * g.ldy + col] = o;
            }
        }
}

// y = bf16( bf16(sum of main slices + bias) + bf16(sum of second-product slices) )
__global__ void linear24_reduce_kernel(const float* __restrict__ slab, int sm, int sa, int M, int N,
                                       const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)M * N;
    if (i >= n) return;
    const int64_t row = i / N, col = i - row * N;
    float a = 0.f, b = 0.f;
    for (int z = 0; z < sm; ++z) a += slab[(int64_t)z * n + i];
    for (int z = 0; z < sa; ++z) b += slab[(int64_t)(sm + z) * n + i];
    float out;
    if (sm && sa) out = round_bf16(a + (bias ? bf16_to_f32(bias[col]) : 0.f)) + round_bf16(b);
    else if (sm) out = a + (bias ? bf16_to_f32(bias[col]) : 0.f);
    else out = b;
    y[row * ldy + col] = f32_to_bf16(out);
}

template <int WAVES_M, int FR_M>
static int launch_linear24(const FwdArgs& g, dim3 grid, hipStream_t st) {
    constexpr int TBM = WAVES_M * FR_M * 16;
    const int lds = 2 * (TBM + FBN) * FSTRIDE * (int)sizeof(bf16_t) + 256 * 4 + 16 * 8;
    SLIM_HIP(slim_allow_dynamic_lds((const void*)linear24_kernel<WAVES_M, FR_M>, lds));
    linear24_kernel<WAVES_M, FR_M><<<grid, 256, lds, st>>>(g);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Slices of the two products' reductions for one launch.  Splitting buys occupancy when the tile grid alone
// is far below the 256 CUs, and costs 8 bytes of slab traffic per slice and output element; the slice count
// aims at >= 512 workgroups, is capped so that the slab traffic stays below half of the weight bytes
// streamed, and never exceeds the number of K-steps.
static void plan_splits(int tiles, int64_t M, int64_t N, int nk_main, int nk_ad, int* sm, int* sa) {
    *sm = nk_main ? 1 : 0;
    *sa = nk_ad ? 1 : 0;
    if (tiles >= 256) return;
    const double weight_bytes = (double)N * FBK * (0.375 * nk_main + 2.0 * nk_ad);
    int64_t want = (512 + tiles - 1) / tiles;
    const int64_t cap = (int64_t)(weight_bytes / (16.0 * (double)M * (double)N));
    if (want > cap) want = cap;
    const int total = nk_main + nk_ad;
    if (want <= *sm + *sa || total == 0) return;
    if (nk_main) {
        const int64_t v = (want * nk_main + total - 1) / total;
        *sm = (int)(v < 1 ? 1 : (v > nk_main ? nk_main : v));
    }
    if (nk_ad) {
        const int64_t v = (want * nk_ad + total - 1) / total;
        *sa = (int)(v < 1 ? 1 : (v > nk_ad ? nk_ad : v));
    }
}

// ---------------------------------------------------------------------------------------------
// small-batch kernel (M <= 16): weights straight to registers
// ---------------------------------------------------------------------------------------------
// At M <= 16 the weight is streamed once and shared by nobody, so the round trip through LDS and the
// barrier per K-step of the tiled kernel are pure overhead.  Here a wave owns 32 output rows (two MFMA
// row blocks) and walks its slice of the reduction in blocks of 128 channels.  The weight is the A operand:
// lane (n = lane & 15, q = lane >> 4) owns row n and takes, for every block, the 32-channel chunk
// 128 * kb + 32 * q -- one packed unit (8 B of codes + 4 B of positions), read 128 B-contiguous by the 16
// lanes of a q group -- which it expands in registers into the four 8-channel A fragments of four MFMAs.
// The activations are the B operand: lane (m = lane & 15, q) reads x[m][128 * kb + 32 * q + 8 * o ..] for MFMA
// o, i.e. the channel order inside a block is permuted the same way on both operands, which a dot product
// does not notice.  C[n][m]: a lane ends with four consecutive output rows of one activation row.
// Each workgroup (4 waves = 128 rows = one packed tile row) writes fp32 partial sums for its slice of the
// reduction; linear24_reduce_kernel combines the slices of the main and the adapter product with the
// rounding points of the reference.
struct DecArgs {
    const bf16_t* x;       // [M, Kd] activations (row stride ldx)
    int64_t ldx;
    int M, N, Kd;
    float a_div;           // every activation element becomes bf16(x / a_div) on load when != 1
    const uint8_t* vals;   // PACKED: tile-major codes / positions of W [N, Kd]
    const uint8_t* meta;
    const float* sf;
    const bf16_t* Wd;      // MODE_DENSE: dense bf16 [N, Kd], row stride ldw
    int64_t ldw;
    const uint8_t* qcodes; // MODE_Q4: 4-bit codes [N, Kd / 2] (even channel in the low nibble) ...
    const bf16_t* qalpha;  // ... and group scales [N, Kd / qblock]; value = bf16(code / 7 * alpha)
    int qblock;
    int kb_per_slice;      // blocks of 128 channels per reduction slice
    float* slab;           // [slices][M][N]; this launch writes slices slice0 .. slice0 + gridDim.y - 1
    int slice0;
};

enum { MODE_DENSE = 0, MODE_PACKED24 = 1, MODE_Q4 = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void linear24_decode_kernel(DecArgs g) {
    constexpr bool PACKED = MODE == MODE_PACKED24;
    __shared__ uint32_t pair_lut[256];
    __shared__ uint2 sel_lut[16];
    __shared__ float q4_level[16];     // code / 7 for the sixteen 4-bit codes (index = code & 15)
    const int t = threadIdx.x;
    if (MODE == MODE_Q4) {
        if (t < 16) q4_level[t] = (float)((int)((uint32_t)t << 28) >> 28) / 7.0f;
        __syncthreads();
    }
    if (PACKED) {
        const float sf = *g.sf;
        const int c0 = ((int)((uint32_t)t << 28)) >> 28, c1 = ((int)((uint32_t)t << 24)) >> 28;
        pair_lut[t] = (uint32_t)f32_to_bf16((float)c0 / sf) | ((uint32_t)f32_to_bf16((float)c1 / sf) << 16);
        if (t < 16) {
            const int p0 = t & 3, p1 = t >> 2;
            uint32_t sv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) sv[e] = (e == p0) ? 0x0100u : (e == p1) ? 0x0302u : 0x0c0cu;
            sel_lut[t] = make_uint2(sv[0] | (sv[1] << 16), sv[2] | (sv[3] << 16));
        }
        __syncthreads();
    }
    const int lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int tn = blockIdx.x;
    const int nkb = (g.Kd + 127) / 128;
    const int kb0 = blockIdx.y * g.kb_per_slice;
    const int kb1 = min(kb0 + g.kb_per_slice, nkb);
    const int m_row = fr < g.M ? fr : g.M - 1;
    const bf16_t* xrow = g.x + (int64_t)m_row * g.ldx;
    const int chunks = (g.Kd + 31) / 32;                // Kd is a multiple of 8 (of 64 when packed)
    const int64_t ktiles = g.Kd / 64;

    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    struct Stage {              // operands of one block of 128 channels, as loaded
        uint2 pv[2];
        uint32_t pm[2];
        u16x8 wd[2][4];
        u16x8 xb[4];
        uint4 qc[2];            // MODE_Q4: 32 codes of the lane's chunk
        float qa[2];            //          and their group scale
    };
    auto load_block = [&](Stage& sg, int kb) {
        const int chunk = 4 * kb + q;
        const bool ok = chunk < chunks;
        const int k = chunk * 32;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (k + 8 * o < g.Kd) sg.xb[o] = *(const u16x8*)(xrow + k + 8 * o);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) sg.xb[o][j] = 0;
            }
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r_in_tile = wave * 32 + rb * 16 + fr;
            if (PACKED) {
                if (ok) {
                    const int64_t unit = (((int64_t)tn * ktiles + (chunk >> 1)) * 2 + (chunk & 1)) * 128 + r_in_tile;
                    sg.pv[rb] = *(const uint2*)(g.vals + unit * 8);
                    sg.pm[rb] = *(const uint32_t*)(g.meta + unit * 4);
                } else {
                    sg.pv[rb] = make_uint2(0u, 0u);
                    sg.pm[rb] = 0x44444444u;          // positions (0, 1) with zero codes
                }
            } else if (MODE == MODE_Q4) {
                const int64_t row = (int64_t)tn * 128 + r_in_tile;
                if (ok) {
                    sg.qc[rb] = *(const uint4*)(g.qcodes + row * (g.Kd / 2) + k / 2);
                    sg.qa[rb] = bf16_to_f32(g.qalpha[row * (g.Kd / g.qblock) + k / g.qblock]);
                } else {
                    sg.qc[rb] = make_uint4(0u, 0u, 0u, 0u);
                    sg.qa[rb] = 0.f;
                }
            } else {
                const bf16_t* wrow = g.Wd + ((int64_t)tn * 128 + r_in_tile) * g.ldw + k;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (k + 8 * o < g.Kd) sg.wd[rb][o] = *(const u16x8*)(wrow + 8 * o);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) sg.wd[rb][o][j] = 0;
                    }
                }
            }
        }
    };
    auto compute_block = [&](const Stage& sg) {
        bf16x8 bx[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            u16x8 v = sg.xb[o];
            if (g.a_div != 1.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f32_to_bf16(bf16_to_f32(v[j]) / g.a_div);
            }
            bx[o] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                bf16x8 a;
                if (PACKED) {
                    uint32_t w[4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int grp = 2 * o + h;
                        const uint32_t byte = ((grp < 4 ? sg.pv[rb].x : sg.pv[rb].y) >> (8 * (grp & 3))) & 0xffu;
                        const uint32_t pair = pair_lut[byte];
                        const uint2 sel = sel_lut[(sg.pm[rb] >> (4 * grp)) & 0xfu];
                        w[2 * h] = __builtin_amdgcn_perm(pair, pair, sel.x);
                        w[2 * h + 1] = __builtin_amdgcn_perm(pair, pair, sel.y);
                    }
                    const uint4 packed4 = make_uint4(w[0], w[1], w[2], w[3]);
                    a = __builtin_bit_cast(bf16x8, packed4);
                } else if (MODE == MODE_Q4) {
                    const uint32_t word = o == 0 ? sg.qc[rb].x : o == 1 ? sg.qc[rb].y : o == 2 ? sg.qc[rb].z : sg.qc[rb].w;
                    u16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = f32_to_bf16(q4_level[(word >> (4 * j)) & 15u] * sg.qa[rb]);
                    a = __builtin_bit_cast(bf16x8, v);
                } else {
                    a = __builtin_bit_cast(bf16x8, sg.wd[rb][o]);
                }
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bx[o], acc[rb], 0, 0, 0);
            }
        }
    };
    // two register stages: the loads of the next block are in flight while the current one is expanded and multiplied
    Stage s0, s1;
    if (kb0 < kb1) load_block(s0, kb0);
    for (int kb = kb0; kb < kb1; kb += 2) {
        if (kb + 1 < kb1) load_block(s1, kb + 1);
        compute_block(s0);
        if (kb + 1 < kb1) {
            if (kb + 2 < kb1) load_block(s0, kb + 2);
            compute_block(s1);
        }
    }
    // C[n][m]: lane holds m = fr, rows n = 4 * q + reg of each row block
    if (fr < g.M) {
        float* sl = g.slab + ((int64_t)(g.slice0 + blockIdx.y) * g.M + fr) * g.N + (int64_t)tn * 128 + wave * 32 + 4 * q;
        *(f32x4*)(sl) = acc[0];
        *(f32x4*)(sl + 16) = acc[1];
    }
}

#define DECODE_MAX_M 16
#define DECODE_MIN_WEIGHT ((int64_t)1 << 26)
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// slices of a decode launch: enough workgroups to fill the chip four times over, at least two blocks per
// slice, slab traffic (8 B per slice and output) below half of the streamed weight bytes
static int decode_slices(int64_t M, int64_t N, int64_t Kd, double bytes_per_weight) {
    const int nkb = (int)((Kd + 127) / 128);
    const int64_t tiles = N / 128;
    int64_t want = (1024 + tiles - 1) / tiles;
    const int64_t cap = (int64_t)((double)N * (double)Kd * bytes_per_weight / (16.0 * (double)M * (double)N));
    if (want > cap) want = cap;
    if (want > nkb / 2) want = nkb / 2;
    if (want < 1) want = 1;
    const int per = (int)((nkb + want - 1) / want);
    return (nkb + per - 1) / per;
}

template <int MODE>
static int launch_decode(DecArgs g, int slices, hipStream_t st) {
    const int nkb = (g.Kd + 127) / 128;
    g.kb_per_slice = (nkb + slices - 1) / slices;
    dim3 grid((unsigned)(g.N / 128), (unsigned)slices);
    linear24_decode_kernel<MODE><<<grid, 256, 0, st>>>(g);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Rows of the activation tile.  A 256-row tile (instantiated below, 110 KB of LDS, one workgroup per CU) would halve the
// per-output cost of expanding the packed weight tile, but measured slower at every batch size (0.50 vs 0.36 ms at 256
// rows of a 28672 x 8192 layer): two resident workgroups per CU hide more latency than the saved expansion is worth.
static inline int pick_tbm(int64_t M) { return M <= 16 ? 16 : M <= 32 ? 32 : M <= 64 ? 64 : 128; }

static int run_linear24(FwdArgs g, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    const int TBM = pick_tbm(g.M);
    const int tiles = (g.N / FBN) * ((g.M + TBM - 1) / TBM);
    const int nk_main = g.vals ? g.K / FBK : 0;
    const int nk_ad = g.Bt ? (g.Ka + FBK - 1) / FBK : 0;
    int sm, sa;
    plan_splits(tiles, g.M, g.N, nk_main, nk_ad, &sm, &sa);
    const bool split = (sm + sa) > ((nk_main ? 1 : 0) + (nk_ad ? 1 : 0)) &&
                       workspace && workspace_bytes >= (int64_t)(sm + sa) * g.M * g.N * 4;
    dim3 grid((unsigned)tiles, 1);
    if (split) {
        g.splits_main = sm;
        g.splits_ad = sa;
        g.ksteps_main = sm ? (nk_main + sm - 1) / sm : 0;
        g.ksteps_ad = sa ? (nk_ad + sa - 1) / sa : 0;
        // drop slices left empty by the rounding
        if (sm) g.splits_main = (nk_main + g.ksteps_main - 1) / g.ksteps_main;
        if (sa) g.splits_ad = (nk_ad + g.ksteps_ad - 1) / g.ksteps_ad;
        g.slab = (float*)workspace;
        grid.y = (unsigned)(g.splits_main + g.splits_ad);
    } else {
        g.slab = nullptr;
    }
    int rc;
    switch (TBM) {
        case 16: rc = launch_linear24<1, 1>(g, grid, st); break;
        case 32: rc = launch_linear24<1, 2>(g, grid, st); break;
        case 64: rc = launch_linear24<1, 4>(g, grid, st); break;
        case 128: rc = launch_linear24<2, 4>(g, grid, st); break;
        default: rc = launch_linear24<2, 8>(g, grid, st); break;
    }
    if (rc != SLIM_OK) return rc;
    if (split) {
        const int64_t n = (int64_t)g.M * g.N;
        linear24_reduce_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(g.slab, g.splits_main, g.splits_ad, g.M, g.N,
                                                                             g.bias, g.y, g.ldy);
        SLIM_LAUNCH_CHECK();
    }
    return SLIM_OK;
}


// Workspace: t [M, r128] bf16 (r128 = r rounded up to 128, the width of the tile grid of the first launch)
// followed by the split-reduction slab, sized for the larger of the two launches.
// Small-batch forward.  The adapter operands are either dense bf16 (LeffT / ReffT) or 4-bit group-quantised
// (qL_* / qR_*: codes [rows, Kd / 2] and scales [rows, Kd / qblock]), never both.
static int decode_path(const void* x, int64_t ldx, int64_t M, int64_t K, const void* vals, const void* meta, const float* sf,
                       int64_t N, const void* bias, const void* LeffT, const void* ReffT, const void* qL_codes,
                       const void* qL_alpha, const void* qR_codes, const void* qR_alpha, int qblock, int64_t r, float a_div,
                       void* y, int64_t ldy, char* ws, hipStream_t st) {
    const int64_t r8 = round_up(r, 8), r128 = round_up(r, 128);
    const bool q4 = qL_codes != nullptr;
    SLIM_REQUIRE(r == 0 || q4 || (LeffT && ReffT && ((uintptr_t)LeffT % 16) == 0 && ((uintptr_t)ReffT % 16) == 0),
                 "slim_linear24_forward: adapter operands must be 16-byte aligned");
    // every product goes through fp32 slices + one combine
    const int64_t t_bytes = round_up(M * r128 * 2, 256);
    float* slab = (float*)(ws + (r > 0 ? t_bytes : 0));
    bf16_t* t = nullptr;
    if (r > 0) {
        t = (bf16_t*)ws;
        DecArgs d{};
        d.x = (const bf16_t*)x;
        d.ldx = ldx;
        d.M = (int)M;
        d.N = (int)r128;
        d.Kd = (int)K;
        d.a_div = a_div;
        d.Wd = (const bf16_t*)LeffT;
        d.ldw = K;
        d.qcodes = (const uint8_t*)qL_codes;
        d.qalpha = (const bf16_t*)qL_alpha;
        d.qblock = qblock;
        d.slab = slab;
        d.slice0 = 0;
        const int s1 = decode_slices(M, r128, K, q4 ? 0.5 : 2.0);
        const int rc = q4 ? launch_decode<MODE_Q4>(d, s1, st) : launch_decode<MODE_DENSE>(d, s1, st);
        if (rc != SLIM_OK) return rc;
        const int64_t n1 = M * r128;
        linear24_reduce_kernel<<<(unsigned)ceil_div64(n1, 256), 256, 0, st>>>(slab, 0, s1, (int)M, (int)r128, nullptr, t, r128);
        SLIM_LAUNCH_CHECK();
    }
    DecArgs d{};
    d.x = (const bf16_t*)x;
    d.ldx = ldx;
    d.M = (int)M;
    d.N = (int)N;
    d.Kd = (int)K;
    d.a_div = 1.f;
    d.vals = (const uint8_t*)vals;
    d.meta = (const uint8_t*)meta;
    d.sf = sf;
    d.slab = slab;
    d.slice0 = 0;
    const int sm = decode_slices(M, N, K, 0.375);
    int rc = launch_decode<MODE_PACKED24>(d, sm, st);
    if (rc != SLIM_OK) return rc;
    int sa = 0;
    if (r > 0) {
        DecArgs a{};
        a.x = t;
        a.ldx = r128;
        a.M = (int)M;
        a.N = (int)N;
        a.Kd = (int)r8;
        a.a_div = a_div;
        a.Wd = (const bf16_t*)ReffT;
        a.ldw = r8;
        a.qcodes = (const uint8_t*)qR_codes;
        a.qalpha = (const bf16_t*)qR_alpha;
        a.qblock = qblock;
        a.slab = slab;
        a.slice0 = sm;
        sa = decode_slices(M, N, r8, q4 ? 0.5 : 2.0);
        rc = q4 ? launch_decode<MODE_Q4>(a, sa, st) : launch_decode<MODE_DENSE>(a, sa, st);
        if (rc != SLIM_OK) return rc;
    }
    const int64_t n2 = M * N;
    linear24_reduce_kernel<<<(unsigned)ceil_div64(n2, 256), 256, 0, st>>>(slab, sm, sa, (int)M, (int)N, (const bf16_t*)bias,
                                                                         (bf16_t*)y, ldy);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int64_t slim_linear24_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t r) {
    if (M <= 0 || N <= 0 || K <= 0 || r < 0) return -1;
    const int64_t r128 = round_up(r, 128);
    const int64_t t_bytes = round_up(M * r128 * 2, 256);
    int64_t slab = 0;
    if (M <= DECODE_MAX_M) {
        const int64_t r8 = round_up(r, 8);
        const int64_t s1 = r ? max(decode_slices(M, r128, K, 2.0), decode_slices(M, r128, K, 0.5)) : 0;
        const int64_t s2 = decode_slices(M, N, K, 0.375) +
                           (r ? max(decode_slices(M, N, r8, 2.0), decode_slices(M, N, r8, 0.5)) : 0);
        const int64_t a = s1 * M * r128 * 4, b = s2 * M * N * 4;
        slab = a > b ? a : b;       // the tiled path below may be taken instead: keep the larger of the two
    }
    for (int pass = 0; pass < 2; ++pass) {
        const int64_t n = pass ? N : r128;
        if (n == 0) continue;
        const int TBM = pick_tbm(M);
        const int tiles = (int)((n / FBN) * ((M + TBM - 1) / TBM));
        int sm, sa;
        plan_splits(tiles, M, n, pass ? (int)(K / FBK) : 0, pass ? (int)((round_up(r, 8) + FBK - 1) / FBK) : (int)(K / FBK), &sm, &sa);
        const int64_t b = (int64_t)(sm + sa) * M * n * 4;
        slab = b > slab ? b : slab;
    }
    return t_bytes + slab;
}

extern "C" int slim_linear24_forward(const void* x, int64_t ldx, int64_t M, int64_t K, const void* vals, const void* meta,
                                     const float* sf, int64_t N, const void* bias, const void* LeffT, const void* ReffT,
                                     int64_t r, float a_div, void* y, int64_t ldy, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(x && vals && meta && sf && y, "slim_linear24_forward: null pointer");
    SLIM_REQUIRE(M > 0 && M < (1 << 30) && N > 0 && K > 0 && N % 128 == 0 && K % 64 == 0 && N < (1 << 30) && K < (1 << 30),
                 "slim_linear24_forward: N must be a multiple of 128 and K of 64");
    SLIM_REQUIRE(ldx >= K && ldx % 8 == 0 && ldy >= N && ((uintptr_t)x % 16) == 0,
                 "slim_linear24_forward: x must be 16-byte aligned with a row stride that is a multiple of 8");
    SLIM_REQUIRE(r >= 0 && (r == 0 || (LeffT && ReffT)), "slim_linear24_forward: adapters need LeffT and ReffT");
    SLIM_REQUIRE(a_div > 0.f, "slim_linear24_forward: a_div must be positive");
    SLIM_REQUIRE(workspace_bytes >= slim_linear24_workspace_bytes(M, N, K, r) && (workspace || workspace_bytes == 0) &&
                     ((uintptr_t)workspace % 256) == 0,
                 "slim_linear24_forward: workspace too small or not 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t r8 = round_up(r, 8), r128 = round_up(r, 128);
    SLIM_PROF(M <= 128 ? "linear24_small_batch" : "linear24_large_batch", st,
              (double)N * K * 0.375 + (double)(K + N) * r * 2.0 + (double)M * (K + N) * 2.0,
              2.0 * M * ((double)N * K + (double)(K + N) * r));
    char* ws = (char*)workspace;
    int64_t ws_left = workspace_bytes;
    bf16_t* t = nullptr;
    // The register-streaming kernel pays five launches; it is ahead of the tiled one only where streaming the weight
    // outweighs them (measured: 28672 x 8192 yes, 4096 x 4096 no)
    if (M <= DECODE_MAX_M && N * K >= DECODE_MIN_WEIGHT)
        return decode_path(x, ldx, M, K, vals, meta, sf, N, bias, LeffT, ReffT, nullptr, nullptr, nullptr, nullptr, 0, r, a_div,
                           y, ldy, ws, st);
    if (r > 0) {
        SLIM_REQUIRE(((uintptr_t)LeffT % 16) == 0 && ((uintptr_t)ReffT % 16) == 0,
                     "slim_linear24_forward: adapter operands must be 16-byte aligned");
        const int64_t t_bytes = round_up(M * r128 * 2, 256);
        t = (bf16_t*)ws;
        ws += t_bytes;
        ws_left -= t_bytes;
        // t = bf16( bf16(x / a_div) L_eff ): L_eff^T is [r128, K], rows >= r zero, so t[:, r:] = 0
        FwdArgs g1{};
        g1.M = (int)M;
        g1.N = (int)r128;
        g1.a = (const bf16_t*)x;
        g1.lda = ldx;
        g1.Bt = (const bf16_t*)LeffT;
        g1.ldb = K;
        g1.Ka = (int)K;
        g1.a_div = a_div;
        g1.y = t;
        g1.ldy = r128;
        const int rc = run_linear24(g1, ws, ws_left, st);
        if (rc != SLIM_OK) return rc;
    }
    FwdArgs g{};
    g.x = (const bf16_t*)x;
    g.ldx = ldx;
    g.M = (int)M;
    g.K = (int)K;
    g.N = (int)N;
    g.vals = (const uint8_t*)vals;
    g.meta = (const uint8_t*)meta;
    g.sf = sf;
    g.bias = (const bf16_t*)bias;
    g.a_div = 1.f;
    if (r > 0) {
        // + bf16( bf16(t / a_div) R_eff ): R_eff^T is [N, r8], columns >= r zero
        g.a = t;
        g.lda = r128;
        g.Bt = (const bf16_t*)ReffT;
        g.ldb = r8;
        g.Ka = (int)r8;
        g.a_div = a_div;
    }
    g.y = (bf16_t*)y;
    g.ldy = ldy;
    return run_linear24(g, ws, ws_left, st);
}

// Small-batch forward (M <= 16) with the adapters in 4-bit group-quantised form, as slim.lora.quantize_lora
// defines it (quantization/utils.py:123-128, :188-193): value = bf16(code / 7 * alpha), one bf16 scale per
// `qblock` consecutive reduction channels.  Operands are laid out for the kernel:
//   qL_codes uint8 [r128, K / 2], qL_alpha bf16 [r128, K / qblock]   (L_eff^T, rows >= r zero)
//   qR_codes uint8 [N, r / 2],    qR_alpha bf16 [N, r / qblock]      (R_eff^T), r % qblock == 0, qblock % 32 == 0
// with the even channel of a byte in its low nibble.  Everything else as slim_linear24_forward.
extern "C" int slim_linear24_forward_q4(const void* x, int64_t ldx, int64_t M, int64_t K, const void* vals, const void* meta,
                                        const float* sf, int64_t N, const void* bias, const void* qL_codes,
                                        const void* qL_alpha, const void* qR_codes, const void* qR_alpha, int64_t r,
                                        int qblock, float a_div, void* y, int64_t ldy, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(x && vals && meta && sf && y && qL_codes && qL_alpha && qR_codes && qR_alpha, "slim_linear24_forward_q4: null pointer");
    if (M > DECODE_MAX_M) {
        slim_set_error("slim_linear24_forward_q4: batches above %d rows take dense adapter operands (slim_linear24_forward)", DECODE_MAX_M);
        return SLIM_ERR_UNSUPPORTED;
    }
    SLIM_REQUIRE(M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 64 == 0 && N < (1 << 30) && K < (1 << 30),
                 "slim_linear24_forward_q4: N must be a multiple of 128 and K of 64");
    SLIM_REQUIRE(ldx >= K && ldx % 8 == 0 && ldy >= N && ((uintptr_t)x % 16) == 0,
                 "slim_linear24_forward_q4: x must be 16-byte aligned with a row stride that is a multiple of 8");
    SLIM_REQUIRE(r > 0 && qblock >= 32 && qblock % 32 == 0 && r % qblock == 0 && K % qblock == 0,
                 "slim_linear24_forward_q4: r and K must be multiples of qblock, qblock of 32");
    SLIM_REQUIRE(((uintptr_t)qL_codes % 16) == 0 && ((uintptr_t)qR_codes % 16) == 0,
                 "slim_linear24_forward_q4: code arrays must be 16-byte aligned");
    SLIM_REQUIRE(a_div > 0.f, "slim_linear24_forward_q4: a_div must be positive");
    SLIM_REQUIRE(workspace && workspace_bytes >= slim_linear24_workspace_bytes(M, N, K, r) && ((uintptr_t)workspace % 256) == 0,
                 "slim_linear24_forward_q4: workspace too small or not 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SLIM_PROF("linear24_small_batch_q4", st,
              (double)N * K * 0.375 + (double)(K + N) * r * (0.5 + 2.0 / qblock) + (double)M * (K + N) * 2.0,
              2.0 * M * ((double)N * K + (double)(K + N) * r));
    return decode_path(x, ldx, M, K, vals, meta, sf, N, bias, nullptr, nullptr, qL_codes, qL_alpha, qR_codes, qR_alpha, qblock, r,
                       a_div, y, ldy, (char*)workspace, st);
}
