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
#include "gemm.hpp"
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;

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

// one thread per (row, 32-channel chunk), numbered in the order of the packed arrays: a 256-thread workgroup reads 128
// rows x 128 contiguous bytes (whole cache lines) and writes 2 KiB of values and 1 KiB of positions contiguously.
// status bit 0: a group with more than two non-zeros, bit 1: a value that is not bf16(c / sf) for an integer c in [-8, 7]
__global__ __launch_bounds__(256) void pack24_kernel(const bf16_t* __restrict__ W, const float* __restrict__ sf_ptr, int64_t N,
                                                     int64_t K, uint8_t* __restrict__ vals, uint8_t* __restrict__ meta,
                                                     int* __restrict__ status) {
    __shared__ bf16_t levels[16];
    const float sf = *sf_ptr;
    if (threadIdx.x < 16) levels[threadIdx.x] = f32_to_bf16((float)((int)threadIdx.x - 8) / sf);
    __syncthreads();
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= N * (K >> 5)) return;
    const int64_t tk = o >> 8, ktiles = K >> 6;
    const int64_t tn = tk / ktiles, kt = tk - tn * ktiles;
    const int64_t row = (tn << 7) + (o & 127), k32 = 2 * kt + ((o >> 7) & 1);
    const bf16_t* w = W + row * K + k32 * 32;
    u16x8 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *(const u16x8*)(w + 8 * i);
    uint32_t v[2] = {0u, 0u}, m = 0u;
    int bad = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        int pos[2] = {0, 1}, code[2] = {0, 0}, n = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t raw = x[g >> 1][(g & 1) * 4 + e];
            if ((raw & 0x7fffu) == 0) continue;
            const float c = rintf(bf16_to_f32(raw) * sf);
            const bool in_range = c >= -8.f && c <= 7.f;
            if (!in_range || levels[in_range ? (int)c + 8 : 0] != raw) bad |= 2;
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
                                            u16x8 (&r)[(TBM * 8 + 255) / 256]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < (TBM * 8 + 255) / 256; ++i) {
        const int c = t + 256 * i;
        if (TBM * 8 < 256 && c >= TBM * 8) break;
        int row = row0 + (c >> 3);
        row = row < rows ? row : rows - 1;
        // past the end of the reduction the address is clamped and the value zeroed on the way to LDS (store_a_tile):
        // a branch around the load would make the compiler wait for every load separately
        int k = k0 + (c & 7) * 8;
        k = k < Kd ? k : Kd - 8;
        r[i] = *(const u16x8*)(P + (int64_t)row * ld + k);
    }
}

// every element becomes bf16(a / div) when div != 1 (the hook's division by sqrt(rank)); chunks past Kd become zero
template <int TBM>
__device__ __forceinline__ void store_a_tile(bf16_t* lds, const u16x8 (&r)[(TBM * 8 + 255) / 256], int Kd, int k0, float div) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < (TBM * 8 + 255) / 256; ++i) {
        const int c = t + 256 * i;
        if (TBM * 8 < 256 && c >= TBM * 8) break;
        u16x8 v = r[i];
        if (div != 1.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f32_to_bf16(bf16_to_f32(v[j]) / div);
        }
        if (k0 + (c & 7) * 8 >= Kd) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0;
        }
        *(u16x8*)(lds + (c >> 3) * FSTRIDE + (c & 7) * 8) = v;
    }
}

__device__ __forceinline__ void load_b_tile(const bf16_t* __restrict__ P, int64_t ld, int rows, int Kd, int row0, int k0,
                                            u16x8 (&r)[4]) {
    const int t = threadIdx.x;
    int k = k0 + (t & 7) * 8;
    k = k < Kd ? k : Kd - 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = row0 + (t >> 3) + 32 * i;
        row = row < rows ? row : rows - 1;
        r[i] = *(const u16x8*)(P + (int64_t)row * ld + k);
    }
}

__device__ __forceinline__ void store_b_tile(bf16_t* lds, const u16x8 (&r)[4], int Kd, int k0) {
    const int t = threadIdx.x;
    const int kc = (t & 7) * 8;
    const bool past = k0 + kc >= Kd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u16x8 v = r[i];
        if (past) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0;
        }
        *(u16x8*)(lds + ((t >> 3) + 32 * i) * FSTRIDE + kc) = v;
    }
}

// Compressed form of the packed weight tile in LDS, as v_smfmac_f32_16x16x64_bf16 consumes it: per row 64 B of kept
// values (16 groups of 4 dense channels, 2 bf16 each, in channel order) and 8 B of positions (4 bits per group).
// A thread turns its unit (row, 32 channels: 8 groups) into 32 B of values through the code-byte -> two-levels table.
#define PSTRIDE 80      // bytes per row: 64 + 8, rounded up to a multiple of 16
__device__ __forceinline__ void store_packed_tile(bf16_t* lds, uint2 v, uint32_t m, const uint32_t* __restrict__ pair_lut) {
    const int t = threadIdx.x;
    const int row = t & 127, half = t >> 7;
    char* base = (char*)lds + row * PSTRIDE;
    uint4* dst = (uint4*)(base + half * 32);
    dst[0] = make_uint4(pair_lut[v.x & 0xffu], pair_lut[(v.x >> 8) & 0xffu], pair_lut[(v.x >> 16) & 0xffu], pair_lut[v.x >> 24]);
    dst[1] = make_uint4(pair_lut[v.y & 0xffu], pair_lut[(v.y >> 8) & 0xffu], pair_lut[(v.y >> 16) & 0xffu], pair_lut[v.y >> 24]);
    *(uint32_t*)(base + 64 + half * 4) = m;
}

template <int WAVES_M, int FR_M>
__global__ __launch_bounds__(256, (WAVES_M * FR_M <= 8 ? 2 : 1)) void linear24_kernel(FwdArgs g) {
    constexpr int TBM = WAVES_M * FR_M * 16;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int FR_N = FBN / (WAVES_N * 16);
    constexpr int NA = (TBM * 8 + 255) / 256;
    constexpr int BUF = (TBM + FBN) * FSTRIDE;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    auto As = [&](int b) { return smem + b * BUF; };
    auto Bs = [&](int b) { return smem + b * BUF + TBM * FSTRIDE; };
    uint32_t* pair_lut = (uint32_t*)(smem + 2 * BUF);

    // consecutive workgroups (one XCD's share of them, see below) walk down the x tiles of one weight
    // column panel, so the packed panel is read from HBM once and then served by that XCD's L2
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
    // ... and inside an XCD's share consecutive workgroups cycle over a group of 8 weight panels before they move to the
    // next activation tile: the ~64 workgroups an XCD runs at a time then work on 8 weight panels x 8 activation tiles,
    // every operand tile fetched into that L2 once serves eight of them (walking one panel down all activation tiles
    // fetched 32 different activation tiles for 2 panels: 3.4 x the traffic from beyond the L2)
    const int tiles_m = (g.M + TBM - 1) / TBM, tiles_n = g.N / FBN;
    const int grp = wgid / (8 * tiles_m), in_grp = wgid % (8 * tiles_m);
    const int gn = min(8, tiles_n - grp * 8);
    const int tn = grp * 8 + in_grp % gn, tm = in_grp / gn;
    const int row0 = tm * TBM, col0 = tn * FBN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;

    if (g.vals) {
        const float sf = *g.sf;
        const int t = threadIdx.x;
        const int c0 = ((int)((uint32_t)t << 28)) >> 28, c1 = ((int)((uint32_t)t << 24)) >> 28;
        pair_lut[t] = (uint32_t)f32_to_bf16((float)c0 / sf) | ((uint32_t)f32_to_bf16((float)c1 / sf) << 16);
        __syncthreads();
    }

    // The weight rows are the A operand and the activation rows the B operand of every matrix instruction, so the
    // accumulators hold y^T: acc[i][j] = rows n of weight row block j x activation rows m of block i, lane (m = fr,
    // q = fq) register r = y[m][4 q + r of the row block].
    f32x4 acc[FR_M][FR_N];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < FR_M; ++i)
#pragma unroll
            for (int j = 0; j < FR_N; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // All fragment reads of a K-step are issued before its MFMAs (and pinned there): left to itself the compiler
    // interleaves read-wait-MFMA rounds to save registers, and with one or two waves per SIMD each round's LDS
    // latency is exposed.
    // dense operands on both sides (the adapter product): two 32-channel instructions per K-step
    auto mma_dense = [&](int cur) {
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // packed weight: ONE sparse 64-channel instruction per K-step.  A lane (n = fr, kq = fq): the kept values of dense
    // channels 16 kq .. 16 kq + 15 of row n (16 B) and their sixteen position bits; B lane (m = fr, bq = fq): channels
    // 8 bq .. + 7 and 32 + 8 bq .. + 7 of activation row m (the operand layout the instruction pairs them in, see the
    // small-batch kernel below).
    auto mma_sparse = [&](int cur) {
        const bf16_t* a_base = As(cur) + (wr * FR_M * 16 + fr) * FSTRIDE + fq * 8;
        const char* w_base = (const char*)Bs(cur) + (wc * FR_N * 16 + fr) * PSTRIDE;
        bf16x16 xf[FR_M];
        bf16x8 wf[FR_N];
        int idx[FR_N];
#pragma unroll
        for (int i = 0; i < FR_M; ++i) {
            const bf16x8 lo = *(const bf16x8*)(a_base + i * 16 * FSTRIDE), hi = *(const bf16x8*)(a_base + i * 16 * FSTRIDE + 32);
            xf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
#pragma unroll
        for (int j = 0; j < FR_N; ++j) {
            wf[j] = *(const bf16x8*)(w_base + j * 16 * PSTRIDE + fq * 16);
            idx[j] = (int)*(const unsigned short*)(w_base + j * 16 * PSTRIDE + 64 + fq * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FR_M; ++i)
#pragma unroll
            for (int j = 0; j < FR_N; ++j)
                acc[i][j] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(wf[j], xf[i], acc[i][j], idx[j], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // Software pipeline of a product's K-steps [kb, ke): the operands of step k + 2 are requested from memory (into one of
    // two register sets) before step k is multiplied, and the set holding step k + 1 goes to the other LDS buffer after
    // it.  One K-step of the sparse product is ~300 cycles of matrix instructions per wave, a memory round trip several
    // times that: with a distance of one step the loop ran at the speed of the round trip.
    auto pipeline = [&](int kb, int ke, auto& regs, auto load, auto store, auto mma) {
        if (kb >= ke) return;
        // Every load and store below is issued unconditionally (past the end the last step is fetched again and never
        // multiplied): with a load inside a branch the compiler no longer knows how many are outstanding and waits for all
        // of them, the newest included, before the older set is stored.
        const int last = ke - 1;
        load(regs[0], kb);
        load(regs[1], min(kb + 1, last));
        store(regs[0], 0, kb);
        __syncthreads();
        for (int kt = kb; kt < ke; kt += 2) {
            load(regs[0], min(kt + 2, last));
            mma(0);
            store(regs[1], 1, min(kt + 1, last));
            __syncthreads();
            load(regs[1], min(kt + 3, last));
            if (kt + 1 < ke) mma(1);
            store(regs[0], 0, min(kt + 2, last));
            __syncthreads();
        }
    };
    struct MainRegs {
        u16x8 ra[NA];
        uint2 pv;
        uint32_t pm;
    };
    struct AdRegs {
        u16x8 ra[NA], rb[4];
    };
    // main product over K-steps [kb, ke)
    auto run_main = [&](int kb, int ke) {
        MainRegs regs[2];
        const int64_t ktiles = g.K / FBK;
        pipeline(kb, ke, regs,
                 [&](MainRegs& r, int kt) {
                     load_a_tile<TBM>(g.x, g.ldx, g.M, g.K, row0, kt * FBK, r.ra);
                     const int64_t unit = (((int64_t)tn * ktiles + kt) * 2 + (threadIdx.x >> 7)) * 128 + (threadIdx.x & 127);
                     r.pv = *(const uint2*)(g.vals + unit * 8);
                     r.pm = *(const uint32_t*)(g.meta + unit * 4);
                 },
                 [&](MainRegs& r, int buf, int kt) {
                     store_a_tile<TBM>(As(buf), r.ra, g.K, kt * FBK, 1.f);
                     store_packed_tile(Bs(buf), r.pv, r.pm, pair_lut);
                 },
                 mma_sparse);
    };
    // second product over its K-steps [kb, ke)
    auto run_ad = [&](int kb, int ke) {
        AdRegs regs[2];
        pipeline(kb, ke, regs,
                 [&](AdRegs& r, int kt) {
                     load_a_tile<TBM>(g.a, g.lda, g.M, g.Ka, row0, kt * FBK, r.ra);
                     load_b_tile(g.Bt, g.ldb, g.N, g.Ka, col0, kt * FBK, r.rb);
                 },
                 [&](AdRegs& r, int buf, int kt) {
                     store_a_tile<TBM>(As(buf), r.ra, g.Ka, kt * FBK, g.a_div);
                     store_b_tile(Bs(buf), r.rb, g.Ka, kt * FBK);
                 },
                 mma_dense);
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
                const int m = row0 + wr * FR_M * 16 + i * 16 + fr;
                const int n = col0 + wc * FR_N * 16 + j * 16 + fq * 4;
                if (m < g.M) *(f32x4*)(sl + (int64_t)m * g.N + n) = acc[i][j];
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
                const int n = col0 + wc * FR_N * 16 + j * 16 + fq * 4;
                bf16_t h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = f32_to_bf16(acc[i][j][r] + (g.bias ? bf16_to_f32(g.bias[n + r]) : 0.f));
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
    const bool wide = (g.ldy & 3) == 0 && ((uintptr_t)g.y & 7) == 0;
#pragma unroll
    for (int i = 0; i < FR_M; ++i)
#pragma unroll
        for (int j = 0; j < FR_N; ++j) {
            const int m = row0 + wr * FR_M * 16 + i * 16 + fr;
            const int n = col0 + wc * FR_N * 16 + j * 16 + fq * 4;
            if (m >= g.M) continue;
            bf16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (nk_main && nk_ad) {
                    const bf16_t mh = (bf16_t)((parked[i][j][r >> 1] >> (16 * (r & 1))) & 0xffffu);
                    o[r] = f32_to_bf16(bf16_to_f32(mh) + round_bf16(acc[i][j][r]));
                } else if (nk_main) {
                    o[r] = (bf16_t)((parked[i][j][r >> 1] >> (16 * (r & 1))) & 0xffffu);
                } else {
                    o[r] = f32_to_bf16(acc[i][j][r]);
                }
            }
            bf16_t* dst = g.y + (int64_t)m * g.ldy + n;
            if (wide) *(uint2*)dst = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = o[r];
            }
        }
}

// Both products on a tile for batches of hundreds of rows and more.  The 128 x 128 tile above is bound by the LDS, not by the
// matrix pipe: per K-step it expands the packed tile through randomly addressed table reads (bank conflicts), writes it to
// LDS and reads every fragment back, for only 16 sparse instructions per wave.  Here a workgroup of 4 waves owns 128
// activation rows x 256 weight rows, and wave w owns weight rows 64 w .. 64 w + 63 of the panel against ALL 128 activation
// rows:
//   * the weight never passes through LDS: a lane loads the packed bytes of its own operand slots straight from memory
//     (per instruction 4 code bytes + 16 position bits; the 16 lanes of a row group read 2 x 128 contiguous bytes) and turns
//     them into levels in registers;
//   * the code-byte -> two-levels table is replicated once per LDS bank (entry e of lane l at dword 32 e + (l & 31), 32 KiB):
//     every lane reads its own bank, so a lookup costs the conflict-free 2 cycles instead of ~7;
//   * only the activation tile is staged in LDS (two buffers, one barrier per K-step), read by every wave as 8 row blocks
//     for 8 x 4 sparse instructions, two row blocks at a time so that the reads of the next pair overlap the instructions
//     of the current one;
//   * two workgroups per CU (72 KiB of LDS and 256 registers per lane each): the waves of one workgroup reach the barrier
//     of a K-step together, so their matrix instructions and their loads / LDS traffic alternate instead of overlapping;
//     the second workgroup runs out of phase and fills the gaps (one workgroup of 8 waves on a 128 x 512 tile: 1.68 ms
//     against 1.58 ms at 4096 rows, 0.98 against 0.85 ms at 2048).
// No split reduction; the second product follows in the same launch (below).
#define WIDE_TBM 128
#define WIDE_TBN 256
#define WIDE_NT 256
// LDS row strides (bf16 elements) of the activation tiles: 64 + 16 and 32 + 16.  With these the 16 lanes that a ds_read_b128
// serves per cycle (rows r, r + 12 .. and the neighbouring 8-channel piece of rows r + 4 ..) fall on 16 different groups of four
// banks; 64 + 8 and 32 + 8 put two of them on the same banks (38 % of the LDS cycles were conflict cycles).
#define WSTRIDE 80
#define WSTRIDE2 48
__global__ __launch_bounds__(WIDE_NT, 2) void linear24_wide_kernel(FwdArgs g) {
    constexpr int NT = WIDE_NT, TBM = WIDE_TBM, TBN = WIDE_TBN, FR_M = TBM / 16, FR_N = 4;
    constexpr int NA = TBM * 8 / NT;                       // 16-byte pieces of the activation tile per thread
    constexpr int XBYTES = TBM * WSTRIDE * 2;
    extern __shared__ __attribute__((aligned(16))) char wsmem[];
    uint32_t* lut = (uint32_t*)(wsmem + 2 * XBYTES);       // [256][32]

    // tile order as in the kernel above: an XCD's share of the workgroups is contiguous, and inside it consecutive
    // workgroups cycle over a group of 8 weight panels before they move to the next activation tile
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
    const int tiles_m = (g.M + TBM - 1) / TBM, tiles_n = (g.N + TBN - 1) / TBN;
    const int grp = wgid / (8 * tiles_m), in_grp = wgid % (8 * tiles_m);
    const int gn = min(8, tiles_n - grp * 8);
    const int tn = grp * 8 + in_grp % gn, tm = in_grp / gn;
    const int row0 = tm * TBM, col0 = tn * TBN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    {
        const float sf = *g.sf;
#pragma unroll
        for (int i = 0; i < 256 * 32 / NT; ++i) {
            const int q = t + NT * i, e = q >> 5;
            const int c0 = ((int)((uint32_t)e << 28)) >> 28, c1 = ((int)((uint32_t)e << 24)) >> 28;
            lut[q] = (uint32_t)f32_to_bf16((float)c0 / sf) | ((uint32_t)f32_to_bf16((float)c1 / sf) << 16);
        }
    }
    const uint32_t* my_lut = lut + (lane & 31);

    f32x4 acc[FR_M][FR_N];
#pragma unroll
    for (int i = 0; i < FR_M; ++i)
#pragma unroll
        for (int j = 0; j < FR_N; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Per K-step a thread brings in NA pieces of the activation tile (to LDS) and, for each of its FR_N weight row blocks, the
    // 4 code bytes and 16 position bits of its operand slots: lane (n = fr, kq = fq) holds the kept values of dense channels
    // 16 kq .. 16 kq + 15 of row n, i.e. bytes 4 (kq & 1) .. + 3 of the codes and half-word kq & 1 of the positions of the packed
    // unit (row, 32-channel half kq >> 1).  The last panel of a weight whose row count is not a multiple of TBN reads the
    // last 128-row tile again (never stored).
    struct Regs {
        u16x8 ra[NA];
        uint32_t code[FR_N];
        uint32_t pos[FR_N];
    };
    const int64_t ktiles = g.K / FBK;
    const bf16_t* xp[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = t + NT * i;
        int row = row0 + (c >> 3);
        row = row < g.M ? row : g.M - 1;
        xp[i] = g.x + (int64_t)row * g.ldx + (c & 7) * 8;
    }
    const int tn128 = min(col0 / 128 + (wave >> 1), g.N / 128 - 1);
    const int64_t unit0 = ((int64_t)tn128 * ktiles * 2 + (fq >> 1)) * 128 + (wave & 1) * 64 + fr;
    const uint8_t* cp = g.vals + unit0 * 8 + (fq & 1) * 4;
    const uint8_t* mp = g.meta + unit0 * 4 + (fq & 1) * 2;
    // (the position bits are reloaded AFTER the products that use them, into the same registers: requested together with the
    // codes, before the products, they need registers of their own and a copy at the end of the loop body, and that copy
    // makes the loop wait for every load in flight once per two K-steps)
    auto load = [&](Regs& r, int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) r.ra[i] = *(const u16x8*)(xp[i] + (int64_t)kt * FBK);
#pragma unroll
        for (int j = 0; j < FR_N; ++j) r.code[j] = *(const uint32_t*)(cp + (int64_t)kt * 2048 + j * 128);
    };
    auto load_pos = [&](Regs& r, int kt) {
#pragma unroll
        for (int j = 0; j < FR_N; ++j) r.pos[j] = *(const unsigned short*)(mp + (int64_t)kt * 1024 + j * 64);
    };
    auto store = [&](const Regs& r, int buf) {
        char* xb = wsmem + buf * XBYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = t + NT * i;
            *(u16x8*)(xb + ((c >> 3) * WSTRIDE + (c & 7) * 8) * 2) = r.ra[i];
        }
    };
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    // One K-step: the codes of register set `r` become levels, the set is reloaded for step kt_next at once (its activation
    // pieces went to LDS a step ago), then the products from LDS buffer `buf`.  Operand layout of the sparse instruction: see
    // mma_sparse in the kernel above.
    auto step = [&](Regs& r, int buf, int kt_next) {
        const bf16_t* a_base = (const bf16_t*)(wsmem + buf * XBYTES) + fr * WSTRIDE + fq * 8;
        auto read_x = [&](int i) {
            const bf16x8 lo = *(const bf16x8*)(a_base + i * 16 * WSTRIDE), hi = *(const bf16x8*)(a_base + i * 16 * WSTRIDE + 32);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        };
        bf16x8 wf[FR_N];
        bf16x16 xa[2], xb[2];
#pragma unroll
        for (int j = 0; j < FR_N; ++j) {
            const uint32_t c = r.code[j];
            const u32x4 v = {my_lut[(c & 0xffu) * 32], my_lut[((c >> 8) & 0xffu) * 32], my_lut[((c >> 16) & 0xffu) * 32], my_lut[(c >> 24) * 32]};
            wf[j] = __builtin_bit_cast(bf16x8, v);
        }
        xa[0] = read_x(0);
        xa[1] = read_x(1);
        load(r, kt_next);
#pragma unroll
        for (int p = 0; p < FR_M / 2; p += 2) {
            xb[0] = read_x(2 * p + 2);
            xb[1] = read_x(2 * p + 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < FR_N; ++j)
                    acc[2 * p + ii][j] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(wf[j], xa[ii], acc[2 * p + ii][j], (int)r.pos[j], 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * p + 4 < FR_M) {
                xa[0] = read_x(2 * p + 4);
                xa[1] = read_x(2 * p + 5);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < FR_N; ++j)
                    acc[2 * p + 2 + ii][j] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(wf[j], xb[ii], acc[2 * p + 2 + ii][j], (int)r.pos[j], 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        load_pos(r, kt_next);
    };

    // K-steps two ahead in registers, the activation tile one ahead in LDS; every load and store unconditional (see
    // `pipeline` above: past the end the last step is fetched again and never multiplied)
    const int nk = (int)ktiles, last = nk - 1;
    Regs regs[2];
    load(regs[0], 0);
    load_pos(regs[0], 0);
    load(regs[1], min(1, last));
    load_pos(regs[1], min(1, last));
    store(regs[0], 0);
    __syncthreads();                  // also publishes the table
    for (int kt = 0; kt < nk; kt += 2) {
        step(regs[0], 0, min(kt + 2, last));
        store(regs[1], 1);
        __syncthreads();
        if (kt + 1 < nk) step(regs[1], 1, min(kt + 3, last));
        store(regs[0], 0);
        __syncthreads();
    }

    // y = bf16(acc + bias).  With a second product this is the first term of the sum: it is written now and read back by the
    // same lanes after the second product (from L2 or Infinity Cache), which keeps the accumulators free for it.  (Parking it
    // in LDS instead -- 128 KiB -- is possible on paper; the compiler then spills ~250 registers in the first product's loop.)
    constexpr int XBYTES2 = TBM * WSTRIDE2 * 2;            // second product: 32 channels per row and step
    const bool wide = (g.ldy & 3) == 0 && ((uintptr_t)g.y & 7) == 0;
    // (all loads of an epilogue are issued before the first value is used, from clamped addresses and without branches: a
    // load inside a guarded block is waited for on the spot, 32 memory round trips in a row per workgroup)
    //
    // 16-byte accesses: a lane holds 4 consecutive outputs (8 bytes) of a row per 16 x 16 block.  For a pair of neighbouring
    // blocks (j, j + 1) v_permlane16_swap exchanges the odd 16-lane rows of the first block's registers with the even rows of
    // the second's: afterwards the lanes with even fq hold 8 consecutive outputs of block j (their own 4 and those of lane
    // fq + 1), the lanes with odd fq 8 consecutive outputs of block j + 1 -- one 16-byte store per lane and pair instead of two
    // 8-byte ones (the epilogue is bound by the number of store instructions).  The swap is its own inverse, which the
    // read-back of the first term uses.
    const bool wide16 = (g.ldy & 7) == 0 && ((uintptr_t)g.y & 15) == 0;
    auto swap_pair = [&](uint2& x, uint2& y) {
        const auto r0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
        x = make_uint2(r0[0], r1[0]);
        y = make_uint2(r0[1], r1[1]);
    };
    auto pack4 = [&](const bf16_t (&o)[4]) { return make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)); };
    auto write_out = [&](bool add_to_y) {
        float bv[FR_N][4];
#pragma unroll
        for (int j = 0; j < FR_N; ++j) {
            const int n = min(col0 + wave * 64 + j * 16 + fq * 4, g.N - 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[j][r] = (!add_to_y && g.bias) ? bf16_to_f32(g.bias[n + r]) : 0.f;
        }
        if (wide16) {
            // lane's 16-byte piece of the pair (jp, jp + 1): block jp + (fq & 1), outputs 4 (fq & 2) .. + 7 of it
            const int n16 = col0 + wave * 64 + (fq & 1) * 16 + (fq & 2) * 4;
            uint4 prev[FR_M][FR_N / 2];
            if (add_to_y) {
#pragma unroll
                for (int i = 0; i < FR_M; ++i)
#pragma unroll
                    for (int jp = 0; jp < FR_N; jp += 2) {
                        const int m = min(row0 + i * 16 + fr, g.M - 1);
                        const int n = min(n16 + jp * 16, g.N - 8);
                        prev[i][jp / 2] = *(const uint4*)(g.y + (int64_t)m * g.ldy + n);
                    }
            }
#pragma unroll
            for (int i = 0; i < FR_M; ++i)
#pragma unroll
                for (int jp = 0; jp < FR_N; jp += 2) {
                    uint2 v[2];
                    if (add_to_y) {
                        uint2 h0 = make_uint2(prev[i][jp / 2].x, prev[i][jp / 2].y), h1 = make_uint2(prev[i][jp / 2].z, prev[i][jp / 2].w);
                        swap_pair(h0, h1);              // back to one block per register pair
                        const uint2 h[2] = {h0, h1};
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const bf16_t hh[4] = {(bf16_t)(h[q].x & 0xffffu), (bf16_t)(h[q].x >> 16), (bf16_t)(h[q].y & 0xffffu), (bf16_t)(h[q].y >> 16)};
                            bf16_t o[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(bf16_to_f32(hh[r]) + round_bf16(acc[i][jp + q][r]));
                            v[q] = pack4(o);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            bf16_t o[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(acc[i][jp + q][r] + bv[jp + q][r]);
                            v[q] = pack4(o);
                        }
                    }
                    swap_pair(v[0], v[1]);
                    const int m = row0 + i * 16 + fr;
                    const int n = n16 + jp * 16;
                    if (m >= g.M || n >= g.N) continue;
                    *(uint4*)(g.y + (int64_t)m * g.ldy + n) = make_uint4(v[0].x, v[0].y, v[1].x, v[1].y);
                }
            return;
        }
        uint2 prev[FR_M][FR_N];
        if (add_to_y) {
#pragma unroll
            for (int i = 0; i < FR_M; ++i)
#pragma unroll
                for (int j = 0; j < FR_N; ++j) {
                    const int m = min(row0 + i * 16 + fr, g.M - 1);
                    const int n = min(col0 + wave * 64 + j * 16 + fq * 4, g.N - 4);
                    const bf16_t* src = g.y + (int64_t)m * g.ldy + n;
                    if (wide) prev[i][j] = *(const uint2*)src;
                    else prev[i][j] = make_uint2((uint32_t)src[0] | ((uint32_t)src[1] << 16), (uint32_t)src[2] | ((uint32_t)src[3] << 16));
                }
        }
#pragma unroll
        for (int i = 0; i < FR_M; ++i)
#pragma unroll
            for (int j = 0; j < FR_N; ++j) {
                const int m = row0 + i * 16 + fr;
                const int n = col0 + wave * 64 + j * 16 + fq * 4;
                bf16_t o[4];
                if (add_to_y) {
                    const uint2 v = prev[i][j];
                    const bf16_t h[4] = {(bf16_t)(v.x & 0xffffu), (bf16_t)(v.x >> 16), (bf16_t)(v.y & 0xffffu), (bf16_t)(v.y >> 16)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(bf16_to_f32(h[r]) + round_bf16(acc[i][j][r]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(acc[i][j][r] + bv[j][r]);
                }
                if (m >= g.M || n >= g.N) continue;
                bf16_t* dst = g.y + (int64_t)m * g.ldy + n;
                if (wide) *(uint2*)dst = pack4(o);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r] = o[r];
                }
            }
    };
    write_out(false);
    if (!g.Bt) return;

    // Second product a [M, Ka] * Bt[N, Ka]^T (dense bf16 on both sides, the adapter term), K-steps of 32: the rows of Bt are
    // wave-private like the packed rows, so a lane loads its operand slots (8 consecutive reduction channels of one row)
    // straight from memory; the tile of `a` goes through LDS (one 16-byte piece per thread and step).  Pieces at or past Ka
    // (a multiple of 8) read as zero; `a` has zero columns up to the next multiple of 128.
#pragma unroll
    for (int i = 0; i < FR_M; ++i)
#pragma unroll
        for (int j = 0; j < FR_N; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NA2 = TBM * 4 / NT;                      // 16-byte pieces of the 32-channel tile of `a` per thread
    struct Regs2 {
        u16x8 ra[NA2];
        u16x8 b[FR_N];
    };
    // (the thread index passes through an empty asm statement so that the addresses of this phase are computed here and do not
    // occupy registers during the first product, whose loop has none to spare)
    int t2 = t;
    asm volatile("" : "+v"(t2));
    const int fr2 = t2 & 15, fq2 = (t2 >> 4) & 3, wave2 = t2 >> 6;
    const bf16_t* ap[NA2];
#pragma unroll
    for (int i = 0; i < NA2; ++i) {
        const int c = t2 + NT * i;
        int arow = row0 + (c >> 2);
        arow = arow < g.M ? arow : g.M - 1;
        ap[i] = g.a + (int64_t)arow * g.lda + (c & 3) * 8;
    }
    const bf16_t* bp[FR_N];
#pragma unroll
    for (int j = 0; j < FR_N; ++j) {
        int n = col0 + wave2 * 64 + j * 16 + fr2;
        n = n < g.N ? n : g.N - 1;
        bp[j] = g.Bt + (int64_t)n * g.ldb + fq2 * 8;
    }
    auto load2 = [&](Regs2& r, int kt) {
        // (a piece past Ka is fetched from the start of the row and zeroed where it is used: zeroing it here would make the
        // loop wait for the load at once)
#pragma unroll
        for (int i = 0; i < NA2; ++i) r.ra[i] = *(const u16x8*)(ap[i] + kt * 32);
        const int k = kt * 32 + fq2 * 8 < g.Ka ? kt * 32 : 0;
#pragma unroll
        for (int j = 0; j < FR_N; ++j) r.b[j] = *(const u16x8*)(bp[j] + k);
    };
    auto store2 = [&](const Regs2& r, int buf) {
#pragma unroll
        for (int i = 0; i < NA2; ++i) {
            const int c = t + NT * i;
            *(u16x8*)(wsmem + buf * XBYTES2 + ((c >> 2) * WSTRIDE2 + (c & 3) * 8) * 2) = r.ra[i];
        }
    };
    auto step2 = [&](Regs2& r, int buf, int kt, int kt_next) {
        const bf16_t* a_base = (const bf16_t*)(wsmem + buf * XBYTES2) + fr * WSTRIDE2 + fq * 8;
        bf16x8 wf[FR_N], xa[2], xb[2];
        const bool in = kt * 32 + fq2 * 8 < g.Ka;
#pragma unroll
        for (int j = 0; j < FR_N; ++j) wf[j] = __builtin_bit_cast(bf16x8, in ? r.b[j] : u16x8{0, 0, 0, 0, 0, 0, 0, 0});
        xa[0] = *(const bf16x8*)(a_base);
        xa[1] = *(const bf16x8*)(a_base + 16 * WSTRIDE2);
        load2(r, kt_next);
#pragma unroll
        for (int p = 0; p < FR_M / 2; p += 2) {
            xb[0] = *(const bf16x8*)(a_base + (2 * p + 2) * 16 * WSTRIDE2);
            xb[1] = *(const bf16x8*)(a_base + (2 * p + 3) * 16 * WSTRIDE2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < FR_N; ++j)
                    acc[2 * p + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xa[ii], acc[2 * p + ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * p + 4 < FR_M) {
                xa[0] = *(const bf16x8*)(a_base + (2 * p + 4) * 16 * WSTRIDE2);
                xa[1] = *(const bf16x8*)(a_base + (2 * p + 5) * 16 * WSTRIDE2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < FR_N; ++j)
                    acc[2 * p + 2 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xb[ii], acc[2 * p + 2 + ii][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nk2 = (g.Ka + 31) / 32, last2 = nk2 - 1;
    Regs2 regs2[2];
    load2(regs2[0], 0);
    load2(regs2[1], min(1, last2));
    store2(regs2[0], 0);              // (the loop above ended with a barrier: the first product's buffers are free)
    __syncthreads();
    for (int kt = 0; kt < nk2; kt += 2) {
        step2(regs2[0], 0, kt, min(kt + 2, last2));
        store2(regs2[1], 1);
        __syncthreads();
        if (kt + 1 < nk2) step2(regs2[1], 1, kt + 1, min(kt + 3, last2));
        store2(regs2[0], 0);
        __syncthreads();
    }
    write_out(true);
}

static int launch_linear24_wide(const FwdArgs& g, hipStream_t st) {
    const int lds = 2 * WIDE_TBM * WSTRIDE * 2 + 256 * 32 * 4;      // two activation buffers + the replicated table
    SLIM_HIP(slim_allow_dynamic_lds((const void*)linear24_wide_kernel, lds));
    const int tiles = ((g.M + WIDE_TBM - 1) / WIDE_TBM) * ((g.N + WIDE_TBN - 1) / WIDE_TBN);
    linear24_wide_kernel<<<dim3((unsigned)tiles), WIDE_NT, lds, st>>>(g);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
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
    const int lds = 2 * (TBM + FBN) * FSTRIDE * (int)sizeof(bf16_t) + 256 * 4;
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
// small-batch kernel (M <= 16): weights straight to registers, sparse MFMA
// ---------------------------------------------------------------------------------------------
// At M <= 16 the weight is streamed once and shared by nobody, so the round trip through LDS and the
// barrier per K-step of the tiled kernel are pure overhead.  Here a workgroup owns 16 or 32 output rows
// (one or two MFMA row blocks) and the WHOLE reduction, which its four waves share block by block (a block =
// 128 channels; wave w takes blocks w, w + 4, ...); the four partial sums meet in LDS and are added in wave
// order, so there are no partial sums in memory, no second launch to combine them and no atomics.  The
// weight is the A operand and the activations the B operand: C[n][m], a lane ends with four consecutive
// output rows of one activation row.
//
// Packed weight: v_smfmac_f32_16x16x64_bf16 takes the 2:4-compressed operand as it is stored -- per lane 8
// kept values (4 groups of 4 dense channels) and 16 bits of 2-bit positions -- so nothing is expanded to
// dense: lane (n = lane & 15, q = lane >> 4) reads the packed unit of row n, channels 128 kb + 32 q .. + 32
// (8 B of codes, 4 B of positions, 128 B-contiguous over the 16 lanes of a q group), turns each code byte
// into its two bf16 levels with one LDS table lookup, and feeds two instructions (groups 0-3 with the low
// half of the position word, groups 4-7 with the high half, selected by the instruction's ABID field).
// Measured operand layout of the instruction (tools/gpu/probe/smfmac_probe.hip): A lane (i, kq) slots
// 4 s' .. 4 s' + 3 (s' = 0, 1) meet B lane (j, bq = 2 (kq & 1) + s'), elements 8 (kq >> 1) .. + 8, slot s
// taking element 4 ((s >> 1) & 1) + position_s of them; position_s = bits [2 s + 1 : 2 s] of the 16-bit
// half-word ABID selects; C lane (j, q) register r = C[4 q + r][j].  So for instruction t of a block, B lane
// (m, bq) holds x[m][128 kb + 64 h + 32 (bq >> 1) + 16 t + 8 (bq & 1) .. + 8] in elements 8 h .. 8 h + 7.
//
// Dense bf16 and 4-bit group-quantised operands (the adapters) use v_mfma_f32_16x16x32_bf16 with the channel
// order of a block permuted the same way on both operands.
//
// A forward with adapters is two launches: [t = x L | the weight product, kept in fp32], then [t R, added to
// the weight product with the rounding points of the reference].
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
    // what becomes of the product p [M, N]:
    //   out32 != nullptr:  out32 = p                                     (fp32 [M, N], the first term of a later launch)
    //   else               y = bf16( first32 ? bf16(first32 + bias) + bf16(p) : p + bias )
    float* out32;
    const float* first32;
    const bf16_t* bias;
    bf16_t* y;
    int64_t ldy;
};

enum { MODE_NONE = -1, MODE_DENSE = 0, MODE_PACKED24 = 1, MODE_Q4 = 2 };

// lds: 256 words of lookup table, then 4 * RB * 256 floats for the partial sums of the four waves
template <int MODE, int RB>
__device__ __forceinline__ void decode_body(const DecArgs& g, const int row0, uint32_t* lds) {
    constexpr bool PACKED = MODE == MODE_PACKED24;
    uint32_t* pair_lut = lds;                   // PACKED: 256 entries
    float* q4_level = (float*)lds;              // Q4: code / 7 for the sixteen 4-bit codes (index = code & 15)
    float* red = (float*)(lds + 256);
    const int t = threadIdx.x;
    if (MODE == MODE_Q4) {
        if (t < 16) q4_level[t] = (float)((int)((uint32_t)t << 28) >> 28) / 7.0f;
        __syncthreads();
    }
    if (PACKED) {
        const float sf = *g.sf;
        const int c0 = ((int)((uint32_t)t << 28)) >> 28, c1 = ((int)((uint32_t)t << 24)) >> 28;
        pair_lut[t] = (uint32_t)f32_to_bf16((float)c0 / sf) | ((uint32_t)f32_to_bf16((float)c1 / sf) << 16);
        __syncthreads();
    }
    const int lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int nkb = (g.Kd + 127) / 128;
    const int m_row = fr < g.M ? fr : g.M - 1;
    const bf16_t* xrow = g.x + (int64_t)m_row * g.ldx;
    const int chunks = (g.Kd + 31) / 32;                // Kd is a multiple of 8 (of 64 when packed)
    const int64_t ktiles = g.Kd / 64;
    const int tn = row0 >> 7, r_tile0 = row0 & 127;     // packed tile row and the first row inside it

    f32x4 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Stage {              // operands of one block of 128 channels, as loaded
        uint2 pv[RB];
        uint32_t pm[RB];
        u16x8 wd[RB][4];
        u16x8 xb[4];
        uint4 qc[RB];           // MODE_Q4: 32 codes of the lane's chunk
        float qa[RB];           //          and their group scale
    };
    auto load_block = [&](Stage& sg, int kb) {
        const int chunk = 4 * kb + q;
        const bool ok = chunk < chunks;
        const int k = chunk * 32;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            // PACKED: piece (t = o >> 1, h = o & 1) of the sparse instruction's B operand, see above
            const int kx = PACKED ? 128 * kb + 64 * (o & 1) + 32 * (q >> 1) + 16 * (o >> 1) + 8 * (q & 1) : k + 8 * o;
            if (kx < g.Kd) sg.xb[o] = *(const u16x8*)(xrow + kx);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) sg.xb[o][j] = 0;
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r_in_tile = r_tile0 + rb * 16 + fr;
            if (PACKED) {
                if (ok) {
                    const int64_t unit = (((int64_t)tn * ktiles + (chunk >> 1)) * 2 + (chunk & 1)) * 128 + r_in_tile;
                    sg.pv[rb] = *(const uint2*)(g.vals + unit * 8);
                    sg.pm[rb] = *(const uint32_t*)(g.meta + unit * 4);
                } else {
                    sg.pv[rb] = make_uint2(0u, 0u);     // zero codes: the positions do not matter
                    sg.pm[rb] = 0u;
                }
            } else if (MODE == MODE_Q4) {
                const int64_t row = (int64_t)row0 + rb * 16 + fr;
                if (ok) {
                    sg.qc[rb] = *(const uint4*)(g.qcodes + row * (g.Kd / 2) + k / 2);
                    sg.qa[rb] = bf16_to_f32(g.qalpha[row * (g.Kd / g.qblock) + k / g.qblock]);
                } else {
                    sg.qc[rb] = make_uint4(0u, 0u, 0u, 0u);
                    sg.qa[rb] = 0.f;
                }
            } else {
                const bf16_t* wrow = g.Wd + ((int64_t)row0 + rb * 16 + fr) * g.ldw + k;
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
        u16x8 xs[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            xs[o] = sg.xb[o];
            if (g.a_div != 1.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xs[o][j] = f32_to_bf16(bf16_to_f32(xs[o][j]) / g.a_div);
            }
        }
        if (PACKED) {
            bf16x16 b[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const bf16x8 lo = __builtin_bit_cast(bf16x8, xs[2 * tt]), hi = __builtin_bit_cast(bf16x8, xs[2 * tt + 1]);
                b[tt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const uint32_t w0 = sg.pv[rb].x, w1 = sg.pv[rb].y;
                const uint4 a0 = make_uint4(pair_lut[w0 & 0xffu], pair_lut[(w0 >> 8) & 0xffu], pair_lut[(w0 >> 16) & 0xffu],
                                            pair_lut[w0 >> 24]);
                const uint4 a1 = make_uint4(pair_lut[w1 & 0xffu], pair_lut[(w1 >> 8) & 0xffu], pair_lut[(w1 >> 16) & 0xffu],
                                            pair_lut[w1 >> 24]);
                acc[rb] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf16x8, a0), b[0], acc[rb], (int)sg.pm[rb], 0, 0);
                acc[rb] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf16x8, a1), b[1], acc[rb], (int)sg.pm[rb], 0, 1);
            }
            return;
        }
        bf16x8 bx[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) bx[o] = __builtin_bit_cast(bf16x8, xs[o]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                bf16x8 a;
                if (MODE == MODE_Q4) {
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
    if (wave < nkb) load_block(s0, wave);
    for (int kb = wave; kb < nkb; kb += 8) {
        if (kb + 4 < nkb) load_block(s1, kb + 4);
        compute_block(s0);
        if (kb + 4 < nkb) {
            if (kb + 8 < nkb) load_block(s0, kb + 8);
            compute_block(s1);
        }
    }
    // the four waves' partial sums, added in wave order by the first RB waves
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) *(f32x4*)(red + ((wave * RB + rb) * 64 + lane) * 4) = acc[rb];
    __syncthreads();
    if (wave >= RB) return;
    const int rb = wave;
    f32x4 p = *(const f32x4*)(red + ((0 * RB + rb) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const f32x4 v = *(const f32x4*)(red + ((w * RB + rb) * 64 + lane) * 4);
        p[0] += v[0], p[1] += v[1], p[2] += v[2], p[3] += v[3];
    }
    // C[n][m]: the lane holds m = fr, rows n = 4 q .. 4 q + 3 of row block rb
    if (fr >= g.M) return;
    const int n0 = row0 + rb * 16 + 4 * q;
    if (g.out32) {
        *(f32x4*)(g.out32 + (int64_t)fr * g.N + n0) = p;
        return;
    }
    bf16_t h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float b = g.bias ? bf16_to_f32(g.bias[n0 + r]) : 0.f;
        const float out = g.first32 ? round_bf16(g.first32[(int64_t)fr * g.N + n0 + r] + b) + round_bf16(p[r]) : p[r] + b;
        h[r] = f32_to_bf16(out);
    }
    *(uint2*)(g.y + (int64_t)fr * g.ldy + n0) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
}

// blocks [0, blocks_a) work on product a (row blocks of RB_A * 16 rows), the rest on product b (MODE_B == MODE_NONE: none)
template <int MODE_A, int RB_A, int MODE_B, int RB_B>
__global__ __launch_bounds__(256) void linear24_decode_kernel(DecArgs a, DecArgs b, int blocks_a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 + 4 * (RB_A > RB_B ? RB_A : RB_B) * 256];
    const int bx = blockIdx.x;
    if (MODE_B == MODE_NONE || bx < blocks_a) decode_body<MODE_A, RB_A>(a, bx * RB_A * 16, lds);
    else decode_body<(MODE_B == MODE_NONE ? MODE_DENSE : MODE_B), RB_B>(b, (bx - blocks_a) * RB_B * 16, lds);
}

#define DECODE_MAX_M 16
#define DECODE_MIN_WEIGHT ((int64_t)1 << 26)
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

template <int MODE_A, int RB_A, int MODE_B, int RB_B>
static int launch_decode(const DecArgs& a, const DecArgs& b, hipStream_t st) {
    const int blocks_a = a.N / (RB_A * 16);
    const int blocks_b = MODE_B == MODE_NONE ? 0 : b.N / (RB_B * 16);
    linear24_decode_kernel<MODE_A, RB_A, MODE_B, RB_B><<<(unsigned)(blocks_a + blocks_b), 256, 0, st>>>(a, b, blocks_a);
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


// Workspace of the small-batch path: t [M, r128] bf16 (r128 = r rounded up to 128), then the weight product in fp32 [M, N].
static int64_t decode_ws_bytes(int64_t M, int64_t N, int64_t r) {
    return round_up(M * round_up(r, 128) * 2, 256) + round_up(M * N * 4, 256);
}

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
    SLIM_REQUIRE(ldy % 4 == 0 && ((uintptr_t)y % 8) == 0, "slim_linear24_forward: y must be 8-byte aligned with a row stride that is a multiple of 4");
    bf16_t* t = (bf16_t*)ws;
    float* first32 = (float*)(ws + round_up(M * r128 * 2, 256));
    // launch 1: t = x L and, beside it, the weight product
    DecArgs w{};
    w.x = (const bf16_t*)x;
    w.ldx = ldx;
    w.M = (int)M;
    w.N = (int)N;
    w.Kd = (int)K;
    w.a_div = 1.f;
    w.vals = (const uint8_t*)vals;
    w.meta = (const uint8_t*)meta;
    w.sf = sf;
    if (r == 0) {
        w.bias = (const bf16_t*)bias;
        w.y = (bf16_t*)y;
        w.ldy = ldy;
        return launch_decode<MODE_PACKED24, 2, MODE_NONE, 1>(w, w, st);
    }
    w.out32 = first32;
    DecArgs l{};
    l.x = (const bf16_t*)x;
    l.ldx = ldx;
    l.M = (int)M;
    l.N = (int)r128;
    l.Kd = (int)K;
    l.a_div = a_div;
    l.Wd = (const bf16_t*)LeffT;
    l.ldw = K;
    l.qcodes = (const uint8_t*)qL_codes;
    l.qalpha = (const bf16_t*)qL_alpha;
    l.qblock = qblock;
    l.y = t;
    l.ldy = r128;
    // (the t blocks come first in the grid: each streams a whole row block of L, several times the bytes of a weight block)
    int rc = q4 ? launch_decode<MODE_Q4, 1, MODE_PACKED24, 2>(l, w, st) : launch_decode<MODE_DENSE, 1, MODE_PACKED24, 2>(l, w, st);
    if (rc != SLIM_OK) return rc;
    // launch 2: t R, added to the weight product
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
    a.first32 = first32;
    a.bias = (const bf16_t*)bias;
    a.y = (bf16_t*)y;
    a.ldy = ldy;
    return q4 ? launch_decode<MODE_Q4, 2, MODE_NONE, 1>(a, a, st) : launch_decode<MODE_DENSE, 2, MODE_NONE, 1>(a, a, st);
}

#define PRESCALE_MIN_M 16
// out[row][0 .. cols) = bf16(in[row][..] / div), 8 elements per thread (cols % 8 == 0); in place allowed
__global__ __launch_bounds__(256) void div_rows_kernel(const bf16_t* in, int64_t ld_in, int64_t rows, int64_t cols, float div,
                                                       bf16_t* out, int64_t ld_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, per_row = cols >> 3;
    if (i >= rows * per_row) return;
    const int64_t row = i / per_row, c = (i - row * per_row) * 8;
    u16x8 v = *(const u16x8*)(in + row * ld_in + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f32_to_bf16(bf16_to_f32(v[j]) / div);
    *(u16x8*)(out + row * ld_out + c) = v;
}
static int launch_div_rows(const bf16_t* in, int64_t ld_in, int64_t rows, int64_t cols, float div, bf16_t* out, int64_t ld_out,
                           hipStream_t st) {
    div_rows_kernel<<<(unsigned)ceil_div64(rows * (cols >> 3), 256), 256, 0, st>>>(in, ld_in, rows, cols, div, out, ld_out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Whether a batch of M rows runs on the wide tile (linear24_wide_kernel, with the first adapter product as an NT GEMM) or on
// the 128 x 128 tile.  The wide tile wants a grid of a couple of hundred workgroups; below that the smaller tile's finer grid
// (and its split reduction) wins (28672 x 8192, tiled against wide: 0.147 / 0.177 ms at 128 rows, 0.203 / 0.174 at 192,
// 0.203 / 0.176 at 256, 0.316 / 0.228 at 384).
// slim_linear24_set_wide / SLIM_FWD_WIDE: -1 the rule, 0 never, 1 whenever the kernel can run (M > PRESCALE_MIN_M).
static int g_wide_mode = [] {
    const char* e = getenv("SLIM_FWD_WIDE");
    return e && *e ? atoi(e) : -1;
}();
extern "C" void slim_linear24_set_wide(int mode) { g_wide_mode = mode < 0 ? -1 : (mode ? 1 : 0); }
static int pick_wide_tile(int64_t M, int64_t N) {
    if (M <= PRESCALE_MIN_M || g_wide_mode == 0) return 0;
    if (g_wide_mode > 0) return 1;
    const int64_t tiles = ((M + WIDE_TBM - 1) / WIDE_TBM) * ((N + WIDE_TBN - 1) / WIDE_TBN);
    return M >= 160 && tiles >= 192;
}

// Slices of the reduction of t = x L_eff (an [M, r128] output with a long reduction) on the wide path: its 128 x 128 tiles
// alone leave most of the chip idle (1024 rows, rank 1024: 64 tiles), so the reduction is cut until about 512 workgroups
// run; the slab stays below 34 MB (slices x tiles <= 512).  Short reductions (K < 2048) are not cut.
static int wide_t_splits(int64_t M, int64_t r128, int64_t K) {
    const int64_t tiles = ((M + 127) / 128) * (r128 / 128), nk = K / FBK;
    if (tiles <= 0 || nk < 32) return 1;
    int64_t s = 512 / tiles;
    s = s < 1 ? 1 : (s > 8 ? 8 : s);
    return (int)s;
}

extern "C" int64_t slim_linear24_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t r) {
    if (M <= 0 || N <= 0 || K <= 0 || r < 0) return -1;
    const int64_t r128 = round_up(r, 128);
    const int64_t t_bytes = round_up(M * r128 * 2, 256);
    int64_t slab = 0;
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
    if (M > PRESCALE_MIN_M && r > 0) {
        const int64_t b = (int64_t)wide_t_splits(M, r128, K) * M * r128 * 4;  // slices of t = x L on the wide path
        slab = b > slab ? b : slab;
    }
    int64_t total = t_bytes + slab;
    if (M > PRESCALE_MIN_M && r > 0) total += round_up(M * K * 2, 256);      // bf16(x / a_div), written once per call
    if (M <= DECODE_MAX_M) {     // the small-batch path may be taken instead: keep the larger of the two
        const int64_t d = decode_ws_bytes(M, N, r);
        total = d > total ? d : total;
    }
    return total;
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
    const int wide = pick_wide_tile(M, N);
    // From PRESCALE_MIN_M rows on, the hook's two divisions by sqrt(rank) are made once per element by a streaming kernel
    // instead of once per element and column tile inside the products (an IEEE division is ~10 instructions; the second
    // product has N / 128 column tiles).
    const bool prescale = M > PRESCALE_MIN_M && r > 0 && a_div != 1.f;
    const float kernel_div = prescale ? 1.f : a_div;
    if (r > 0) {
        SLIM_REQUIRE(((uintptr_t)LeffT % 16) == 0 && ((uintptr_t)ReffT % 16) == 0,
                     "slim_linear24_forward: adapter operands must be 16-byte aligned");
        const int64_t t_bytes = round_up(M * r128 * 2, 256);
        t = (bf16_t*)ws;
        ws += t_bytes;
        ws_left -= t_bytes;
        const bf16_t* xa = (const bf16_t*)x;
        int64_t ldxa = ldx;
        if (prescale) {
            bf16_t* xs = (bf16_t*)ws;
            const int64_t xs_bytes = round_up(M * K * 2, 256);
            ws += xs_bytes;
            ws_left -= xs_bytes;
            const int rc = launch_div_rows((const bf16_t*)x, ldx, M, K, a_div, xs, K, st);
            if (rc != SLIM_OK) return rc;
            xa = xs;
            ldxa = K;
        }
        // t = bf16( bf16(x / a_div) L_eff ): L_eff^T is [r128, K], rows >= r zero, so t[:, r:] = 0
        if (wide) {
            GemmArgs gt;
            gt.A = xa;
            gt.lda = ldxa;
            gt.B = (const bf16_t*)LeffT;
            gt.ldb = K;
            gt.M = (int)M;
            gt.N = (int)r128;
            gt.Kd = (int)K;
            gt.C16 = t;
            gt.ldc16 = r128;
            const int splits = wide_t_splits(M, r128, K);
            if (splits > 1) {
                gt.slab = (float*)ws;
                gemm_plan_splits(gt, splits);
                if (gt.splits <= 1) gt.slab = nullptr;
            }
            const int rc = slim_gemm_launch(gt, st);
            if (rc != SLIM_OK) return rc;
        } else {
            FwdArgs g1{};
            g1.M = (int)M;
            g1.N = (int)r128;
            g1.a = xa;
            g1.lda = ldxa;
            g1.Bt = (const bf16_t*)LeffT;
            g1.ldb = K;
            g1.Ka = (int)K;
            g1.a_div = kernel_div;
            g1.y = t;
            g1.ldy = r128;
            const int rc = run_linear24(g1, ws, ws_left, st);
            if (rc != SLIM_OK) return rc;
        }
        if (prescale) {
            const int rc = launch_div_rows(t, r128, M, r128, a_div, t, r128, st);
            if (rc != SLIM_OK) return rc;
        }
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
        g.a_div = kernel_div;
    }
    g.y = (bf16_t*)y;
    g.ldy = ldy;
    if (wide) {
        // both products in one launch on the wide tile (the divisions by sqrt(rank) were made beforehand: a_div is 1 here)
        return launch_linear24_wide(g, st);
    }
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
