// scale_important_weights: the 1 % of input channels with the largest scaler_row * sum_rows |W| are quantised at twice
// their size and halved after dequantisation (reference slim/lora.py:132-138, slim/prune.py:331-335,
// slim/quantization/quantization.py:160-170 with its call sites :147, :156, :192, :215).
#include "common.hpp"
#include "../../include/slim_hip.h"

#define IC_SLAB_ROWS 256
#define IC_THREADS 1024

// grid (col_blocks, slabs): per-slab column sums of |W|, 8 consecutive columns per thread (a wave reads 1 KiB per row)
__global__ __launch_bounds__(256) void colabssum_kernel(const bf16_t* __restrict__ w, int64_t N, int64_t K,
                                                        float* __restrict__ partial) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c0 >= K) return;
    const int64_t r0 = (int64_t)blockIdx.y * IC_SLAB_ROWS;
    const int64_t r1 = r0 + IC_SLAB_ROWS < N ? r0 + IC_SLAB_ROWS : N;
    const bf16_t* p = w + r0 * K + c0;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {
        u16x8 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const u16x8*)(p + i * K);
        p += 4 * K;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf16_to_f32(v[i][j] & 0x7fffu);
    }
    for (; r < r1; ++r) {
        const u16x8 v0 = *(const u16x8*)(p);
        p += K;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_to_f32(v0[j] & 0x7fffu);
    }
    float* o = partial + (int64_t)blockIdx.y * K + c0;
    *(f32x4*)(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// One workgroup: metric[c] = scaler_row[c] * float(bf16(column sum)) (the reference's column sums are a bf16 tensor), then
// the `count` largest metrics are flagged; equal metrics at the boundary: lower index first.  Radix select on the float
// bit patterns of |metric| (non-negative, so pattern order is value order).
__global__ __launch_bounds__(IC_THREADS) void important_select_kernel(const float* __restrict__ partial, int slabs,
                                                                      const float* __restrict__ scaler_row, int K, int count,
                                                                      float* __restrict__ metric, uint8_t* __restrict__ flags) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_target, s_greater, s_fill;
    __shared__ unsigned wave_cnt[IC_THREADS / 64];
    const int t = threadIdx.x;
    for (int c = t; c < K; c += IC_THREADS) {
        double sum = 0.0;
        for (int s = 0; s < slabs; ++s) sum += (double)partial[(int64_t)s * K + c];
        metric[c] = scaler_row[c] * round_bf16((float)sum);
    }
    if (t == 0) s_greater = 0;
    __syncthreads();
    if (count <= 0) {
        for (int c = t; c < K; c += IC_THREADS) flags[c] = 0;
        return;
    }
    auto key_of = [&](int i) { return __float_as_uint(fabsf(metric[i])); };
    unsigned prefix = 0, mask = 0, target = (unsigned)(K - count);       // ascending rank of the count-th largest
    for (int pass = 3; pass >= 0; --pass) {
        if (t < 256) hist[t] = 0;
        __syncthreads();
        for (int i = t; i < K; i += IC_THREADS) {
            const unsigned k = key_of(i);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (t == 0) {
            unsigned acc = 0;
            int b = 0;
            while (b < 255 && acc + hist[b] <= target) acc += hist[b], ++b;
            s_target = target - acc;
            s_prefix = prefix | ((unsigned)b << (8 * pass));
        }
        __syncthreads();
        prefix = s_prefix;
        target = s_target;
        mask |= 0xffu << (8 * pass);
        __syncthreads();
    }
    const unsigned thr = prefix;
    unsigned mine = 0;
    for (int i = t; i < K; i += IC_THREADS) mine += key_of(i) > thr;
    mine = wave_sum_u32(mine);
    if ((t & 63) == 0 && mine) atomicAdd(&s_greater, mine);
    __syncthreads();
    if (t == 0) s_fill = s_greater;
    __syncthreads();
    // strictly larger metrics are in; metrics equal to the threshold fill what is left, lowest indices first
    for (int base = 0; base < K; base += IC_THREADS) {
        const int i = base + t;
        const unsigned k = i < K ? key_of(i) : 0u;
        const bool tie = i < K && k == thr;
        const unsigned long long b = __ballot(tie);
        if ((t & 63) == 0) wave_cnt[t >> 6] = (unsigned)__popcll(b);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int w = 0; w < IC_THREADS / 64; ++w) {
            if (w < (t >> 6)) before += wave_cnt[w];
            total += wave_cnt[w];
        }
        const unsigned pos = s_fill + before + (unsigned)__popcll(b & ((1ull << (t & 63)) - 1ull));
        if (i < K) flags[i] = (k > thr || (tie && pos < (unsigned)count)) ? 1 : 0;
        __syncthreads();
        if (t == 0) s_fill += total;
        __syncthreads();
    }
}

extern "C" int64_t slim_important_columns_workspace_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    return (ceil_div64(N, IC_SLAB_ROWS) + 1) * K * (int64_t)sizeof(float);
}

extern "C" int slim_important_columns(const void* W, const float* scaler_row, int64_t N, int64_t K, int64_t count,
                                      uint8_t* flags, void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(W && scaler_row && flags && workspace, "slim_important_columns: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && K < (1 << 30) && count >= 0 && count <= K, "slim_important_columns: bad sizes");
    SLIM_REQUIRE(K % 8 == 0 && ((uintptr_t)W % 16) == 0, "slim_important_columns: K must be a multiple of 8 and W 16-byte aligned");
    SLIM_REQUIRE(workspace_bytes >= slim_important_columns_workspace_bytes(N, K), "slim_important_columns: workspace too small");
    const int64_t slabs = ceil_div64(N, IC_SLAB_ROWS);
    SLIM_REQUIRE(slabs <= 65535, "slim_important_columns: too many rows for one launch");
    hipStream_t st = (hipStream_t)stream;
    SLIM_PROF("important_columns", stream, 2.0 * N * K);
    float* partial = (float*)workspace;
    float* metric = partial + slabs * K;
    colabssum_kernel<<<dim3((unsigned)ceil_div64(K, 256 * 8), (unsigned)slabs), 256, 0, st>>>((const bf16_t*)W, N, K, partial);
    SLIM_LAUNCH_CHECK();
    important_select_kernel<<<1, IC_THREADS, 0, st>>>(partial, (int)slabs, scaler_row, (int)K, (int)count, metric, flags);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// out = bf16(float(in) * factor) in the flagged columns, in elsewhere (`mat[:, cols] /= 1/2` and `*= 1/2` of
// quantization.py:166-168 are exact in bf16 short of overflow / underflow).  In place (out == in) allowed.
__global__ __launch_bounds__(256) void scale_columns_kernel(const bf16_t* in, const uint8_t* __restrict__ flags,
                                                            int64_t N, int64_t K, float factor, bf16_t* out) {
    const int64_t cpr = K >> 3, total = N * cpr;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / cpr, c0 = (t - row * cpr) << 3, off = row * K + c0;
        u16x8 x = *(const u16x8*)(in + off);
        const uint2 f = *(const uint2*)(flags + c0);
        if (f.x | f.y) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool on = ((j < 4 ? f.x >> (8 * j) : f.y >> (8 * (j - 4))) & 0xffu) != 0;
                if (on) x[j] = f32_to_bf16(bf16_to_f32(x[j]) * factor);
            }
        } else if (in == out) {
            continue;
        }
        *(u16x8*)(out + off) = x;
    }
}

extern "C" int slim_scale_columns_bf16(const void* in, const uint8_t* flags, int64_t N, int64_t K, float factor, void* out,
                                       void* stream) {
    SLIM_REQUIRE(in && flags && out && N > 0 && K > 0, "slim_scale_columns_bf16: bad arguments");
    SLIM_REQUIRE(K % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)flags % 8) == 0,
                 "slim_scale_columns_bf16: K must be a multiple of 8 and the pointers aligned");
    SLIM_PROF("scale_columns", stream, 4.0 * N * K);
    int64_t blocks = ceil_div64(N * (K >> 3), 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    scale_columns_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, flags, N, K, factor, (bf16_t*)out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// The same for an fp32 matrix (the per-tensor quantiser's dequantised values, quantization.py:212-215), in place.
__global__ __launch_bounds__(256) void scale_columns_f32_kernel(float* __restrict__ x, const uint8_t* __restrict__ flags,
                                                                int64_t N, int64_t K, float factor) {
    const int64_t total = N * K;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256)
        if (flags[t % K]) x[t] *= factor;
}

extern "C" int slim_scale_columns_f32(float* x, const uint8_t* flags, int64_t N, int64_t K, float factor, void* stream) {
    SLIM_REQUIRE(x && flags && N > 0 && K > 0, "slim_scale_columns_f32: bad arguments");
    int64_t blocks = ceil_div64(N * K, 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    scale_columns_f32_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, flags, N, K, factor);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}
