// Calibration statistics: per-column sum of squares of the activations and the running
// scaler_row recurrence of WrappedGPT.add_batch (reference slim/layerwrapper.py:25-41),
// shift_zeros (slim/utils.py:64-72) and the element-wise square root used for the saliency.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include <cstdlib>

#define SLAB_ROWS 256

// grid (col_blocks, slabs, chunks); each thread owns 8 consecutive columns and walks the
// rows of its slab, so a wave reads 1 KiB contiguous per row.
#define COLSUM_UNROLL 16      // (8, 16 and 32 rows in flight per lane measured the same 6.0 TB/s; the streaming loads made the difference: 5.5 before)
__global__ __launch_bounds__(256) void colsumsq_kernel(const bf16_t* __restrict__ x, int64_t rows_per_chunk,
                                                       int64_t K, int64_t ld, float* __restrict__ partial,
                                                       int slabs) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c0 >= K) return;
    const int slab = blockIdx.y;
    const int64_t chunk = blockIdx.z;
    const int64_t r0 = (int64_t)slab * SLAB_ROWS;
    int64_t r1 = r0 + SLAB_ROWS;
    if (r1 > rows_per_chunk) r1 = rows_per_chunk;
    const bf16_t* p = x + (chunk * rows_per_chunk + r0) * ld + c0;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int64_t r = r0;
    // COLSUM_UNROLL independent 16-byte loads in flight per lane (streamed: the activations are read once and should not
    // displace what the L2 holds); the squares are added in row order
    for (; r + COLSUM_UNROLL <= r1; r += COLSUM_UNROLL) {
        u16x8 v[COLSUM_UNROLL];
#pragma unroll
        for (int i = 0; i < COLSUM_UNROLL; ++i) v[i] = __builtin_nontemporal_load((const u16x8*)(p + i * ld));
        p += COLSUM_UNROLL * ld;
#pragma unroll
        for (int i = 0; i < COLSUM_UNROLL; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = bf16_to_f32(v[i][j]);
                acc[j] += a * a;
            }
        }
    }
    for (; r < r1; ++r) {
        u16x8 v0 = *(const u16x8*)(p);
        p += ld;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = bf16_to_f32(v0[j]);
            acc[j] += a * a;
        }
    }
    float* o = partial + (chunk * slabs + slab) * K + c0;
    *(f32x4*)(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// The same sums for TALL consecutive FULL slabs per workgroup: one uninterrupted walk over TALL * SLAB_ROWS rows, a set of
// accumulators per slab, every partial stored at the end.  The partials (one per slab of SLAB_ROWS rows, rows added in
// order) are the ones the kernel above writes, bit for bit; what changes is how the rows are fetched: no store between
// two slabs (loads and stores retire through one in-order counter, so a wait for the next slab's first loads would also
// wait for the previous slab's stores to be acknowledged) and a quarter of the workgroups, each with a walk four times as
// long: 6.2 -> 6.7 TB/s at K = 4096, 5.9 -> 6.3 at K = 11008 (eight slabs per workgroup, 2048-token samples).
template <int TALL>
__global__ __launch_bounds__(256) void colsumsq_tall_kernel(const bf16_t* __restrict__ x, int64_t rows_per_chunk, int64_t K,
                                                            int64_t ld, float* __restrict__ partial, int slabs) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c0 >= K) return;
    const int slab0 = blockIdx.y * TALL;
    const int64_t chunk = blockIdx.z;
    const bf16_t* p = x + (chunk * rows_per_chunk + (int64_t)slab0 * SLAB_ROWS) * ld + c0;
    float acc[TALL][8];
#pragma unroll
    for (int s = 0; s < TALL; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
#pragma unroll
    for (int s = 0; s < TALL; ++s) {
        for (int r = 0; r < SLAB_ROWS; r += COLSUM_UNROLL) {
            u16x8 v[COLSUM_UNROLL];
#pragma unroll
            for (int i = 0; i < COLSUM_UNROLL; ++i) v[i] = __builtin_nontemporal_load((const u16x8*)(p + i * ld));
            p += COLSUM_UNROLL * ld;
#pragma unroll
            for (int i = 0; i < COLSUM_UNROLL; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = bf16_to_f32(v[i][j]);
                    acc[s][j] += a * a;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < TALL; ++s) {
        float* o = partial + (chunk * slabs + slab0 + s) * K + c0;
        *(f32x4*)(o) = f32x4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
        *(f32x4*)(o + 4) = f32x4{acc[s][4], acc[s][5], acc[s][6], acc[s][7]};
    }
}
static_assert(SLAB_ROWS % COLSUM_UNROLL == 0, "a slab is a whole number of load batches");

// the launch of either kernel: eight or four slabs per workgroup when the chunks are whole groups of that many slabs and
// that still leaves a workgroup per CU (SLIM_COLSUM_TALL=0: never, an A/B aid)
static int launch_colsumsq(const bf16_t* x, int64_t n_chunks, int64_t rows_per_chunk, int64_t K, int64_t ld, float* partial,
                           int slabs, hipStream_t st) {
    static const bool allow_tall = [] {
        const char* v = getenv("SLIM_COLSUM_TALL");
        return !(v && v[0] == '0');
    }();
    const int64_t col_blocks = ceil_div64(K, 256 * 8);
    if (allow_tall && rows_per_chunk % (8 * SLAB_ROWS) == 0 && col_blocks * (slabs / 8) * n_chunks >= 256) {
        dim3 grid((unsigned)col_blocks, (unsigned)(slabs / 8), (unsigned)n_chunks);
        colsumsq_tall_kernel<8><<<grid, 256, 0, st>>>(x, rows_per_chunk, K, ld, partial, slabs);
    } else if (allow_tall && rows_per_chunk % (4 * SLAB_ROWS) == 0 && col_blocks * (slabs / 4) * n_chunks >= 256) {
        dim3 grid((unsigned)col_blocks, (unsigned)(slabs / 4), (unsigned)n_chunks);
        colsumsq_tall_kernel<4><<<grid, 256, 0, st>>>(x, rows_per_chunk, K, ld, partial, slabs);
    } else {
        dim3 grid((unsigned)col_blocks, (unsigned)slabs, (unsigned)n_chunks);
        colsumsq_kernel<<<grid, 256, 0, st>>>(x, rows_per_chunk, K, ld, partial, slabs);
    }
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// One thread per column: fold the slab partials of every chunk in order and apply
//   scaler_row *= n / (n + b); n += b; scaler_row += sqrt(sumsq)^2 / n
// once per chunk (b = batch_per_chunk samples per add_batch call).
__global__ void scaler_update_kernel(float* __restrict__ scaler_row, int64_t K, const float* __restrict__ partial,
                                     int64_t n_chunks, int slabs, int64_t batch_per_chunk, int64_t n_before) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    float sr = scaler_row[c];
    int64_t n = n_before;
    for (int64_t ch = 0; ch < n_chunks; ++ch) {
        float ss = 0.f;
        for (int s = 0; s < slabs; ++s) ss += partial[(ch * slabs + s) * K + c];
        const float f = (float)((double)n / (double)(n + batch_per_chunk));
        sr = sr * f;
        n += batch_per_chunk;
        const float nrm = sqrtf(ss);
        sr = sr + (nrm * nrm) / (float)n;
    }
    scaler_row[c] = sr;
}

extern "C" int64_t slim_colsumsq_workspace_bytes(int64_t n_chunks, int64_t rows_per_chunk, int64_t K) {
    if (n_chunks <= 0 || rows_per_chunk <= 0 || K <= 0) return 0;
    return n_chunks * ceil_div64(rows_per_chunk, SLAB_ROWS) * K * (int64_t)sizeof(float);
}

extern "C" int slim_scaler_row_accum(const void* x, int64_t n_chunks, int64_t rows_per_chunk, int64_t K, int64_t ld,
                                     int64_t batch_per_chunk, float* scaler_row, int64_t nsamples_before,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(x && scaler_row && workspace, "slim_scaler_row_accum: null pointer");
    SLIM_REQUIRE(n_chunks > 0 && rows_per_chunk > 0 && K > 0 && batch_per_chunk > 0 && nsamples_before >= 0,
                 "slim_scaler_row_accum: bad sizes");
    SLIM_REQUIRE(K % 8 == 0 && ld % 8 == 0 && ld >= K && ((uintptr_t)x % 16) == 0,
                 "slim_scaler_row_accum: K and ld must be multiples of 8 and x 16-byte aligned");
    SLIM_REQUIRE(workspace_bytes >= slim_colsumsq_workspace_bytes(n_chunks, rows_per_chunk, K),
                 "slim_scaler_row_accum: workspace too small");
    const int slabs = (int)ceil_div64(rows_per_chunk, SLAB_ROWS);
    SLIM_REQUIRE(n_chunks <= 65535 && slabs <= 65535, "slim_scaler_row_accum: too many chunks/slabs for one launch");
    hipStream_t st = (hipStream_t)stream;
    SLIM_PROF("colsumsq", stream, 2.0 * n_chunks * rows_per_chunk * K);
    { const int rc = launch_colsumsq((const bf16_t*)x, n_chunks, rows_per_chunk, K, ld, (float*)workspace, slabs, st); if (rc) return rc; }
    scaler_update_kernel<<<(unsigned)ceil_div64(K, 256), 256, 0, st>>>(scaler_row, K, (const float*)workspace, n_chunks,
                                                                      slabs, batch_per_chunk, nsamples_before);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- shift_zeros + sqrt ---------------------------------------------------------------
// Single workgroup: m = min over (x == 0 ? 1 : x); out = x + m.  NaN propagates like torch.min.
// sumsq[chunk][c] = slab partials of one chunk added in slab order (the order scaler_update_kernel uses)
__global__ void fold_slabs_kernel(const float* __restrict__ partial, int64_t K, int slabs, float* __restrict__ sumsq) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    const int64_t ch = blockIdx.y;
    float ss = 0.f;
    for (int s = 0; s < slabs; ++s) ss += partial[(ch * slabs + s) * K + c];
    sumsq[ch * K + c] = ss;
}

// The two halves of slim_scaler_row_accum as separate calls, for runs that shard the calibration samples over
// several GPUs: every rank reduces its samples to per-sample column sums, the sums are all-gathered into
// sample order, and every rank folds them with the same recurrence -- bit-identical to a single-GPU run.
extern "C" int slim_colsumsq(const void* x, int64_t n_chunks, int64_t rows_per_chunk, int64_t K, int64_t ld, float* sumsq,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(x && sumsq && workspace, "slim_colsumsq: null pointer");
    SLIM_REQUIRE(n_chunks > 0 && rows_per_chunk > 0 && K > 0, "slim_colsumsq: bad sizes");
    SLIM_REQUIRE(K % 8 == 0 && ld % 8 == 0 && ld >= K && ((uintptr_t)x % 16) == 0,
                 "slim_colsumsq: K and ld must be multiples of 8 and x 16-byte aligned");
    SLIM_REQUIRE(workspace_bytes >= slim_colsumsq_workspace_bytes(n_chunks, rows_per_chunk, K), "slim_colsumsq: workspace too small");
    const int slabs = (int)ceil_div64(rows_per_chunk, SLAB_ROWS);
    SLIM_REQUIRE(n_chunks <= 65535 && slabs <= 65535, "slim_colsumsq: too many chunks/slabs for one launch");
    hipStream_t st = (hipStream_t)stream;
    SLIM_PROF("colsumsq", stream, 2.0 * n_chunks * rows_per_chunk * K);
    { const int rc = launch_colsumsq((const bf16_t*)x, n_chunks, rows_per_chunk, K, ld, (float*)workspace, slabs, st); if (rc) return rc; }
    fold_slabs_kernel<<<dim3((unsigned)ceil_div64(K, 256), (unsigned)n_chunks), 256, 0, st>>>((const float*)workspace, K, slabs, sumsq);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_scaler_row_fold(float* scaler_row, int64_t K, const float* sumsq, int64_t n_chunks,
                                    int64_t batch_per_chunk, int64_t nsamples_before, void* stream) {
    SLIM_REQUIRE(scaler_row && sumsq && K > 0 && n_chunks > 0 && batch_per_chunk > 0 && nsamples_before >= 0,
                 "slim_scaler_row_fold: bad arguments");
    scaler_update_kernel<<<(unsigned)ceil_div64(K, 256), 256, 0, (hipStream_t)stream>>>(scaler_row, K, sumsq, n_chunks, 1,
                                                                                        batch_per_chunk, nsamples_before);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

__global__ __launch_bounds__(1024) void shift_zeros_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int64_t K) {
    __shared__ float red[16];
    __shared__ int has_nan[1];
    if (threadIdx.x == 0) has_nan[0] = 0;
    __syncthreads();
    float m = __builtin_inff();
    bool nan = false;
    for (int64_t i = threadIdx.x; i < K; i += 1024) {
        float v = in[i];
        if (v != v) nan = true;
        if (v == 0.f) v = 1.f;
        m = fminf(m, v);
    }
    if (nan) has_nan[0] = 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    float mm = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mm = fminf(mm, red[w]);
    if (has_nan[0]) mm = __builtin_nanf("");
    for (int64_t i = threadIdx.x; i < K; i += 1024) out[i] = in[i] + mm;
}

__global__ void sqrt_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) out[i] = sqrtf(in[i]);
}

__global__ void any_zero_kernel(const float* __restrict__ in, int64_t K, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K && in[i] == 0.f) atomicOr(flag, 1);
}

extern "C" int slim_shift_zeros(const float* in, float* out, int64_t K, void* stream) {
    SLIM_REQUIRE(in && out && K > 0, "slim_shift_zeros: bad arguments");
    shift_zeros_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(in, out, K);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_sqrt_f32(const float* in, float* out, int64_t K, void* stream) {
    SLIM_REQUIRE(in && out && K > 0, "slim_sqrt_f32: bad arguments");
    sqrt_kernel<<<(unsigned)ceil_div64(K, 256), 256, 0, (hipStream_t)stream>>>(in, out, K);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_any_zero_f32(const float* in, int64_t K, int32_t* flag, void* stream) {
    SLIM_REQUIRE(in && flag && K > 0, "slim_any_zero_f32: bad arguments");
    SLIM_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), (hipStream_t)stream));
    any_zero_kernel<<<(unsigned)ceil_div64(K, 256), 256, 0, (hipStream_t)stream>>>(in, K, flag);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}


// ---- per-linear scalars the driver needs before compressing a block --------------------------------------------------
// From one scaler_row vector: whether any entry is zero (the guard of slim/lora.py:139), the median of sqrt(scaler_row)
// (the lower median, as torch.median) and the `top` largest values of sqrt(scaler_row) in decreasing order with their
// indices (equal values: lower index first).  The driver decides from these which input channels are outliers that the
// low-rank fit represents exactly, and how wide a spectrum its orthonormalisations must cover.  One workgroup: radix
// select on the float bit patterns (non-negative, so pattern order is value order), then a rank sort of the few winners.
#define OS_THREADS 1024
#define OS_MAX_TOP 128
__global__ __launch_bounds__(OS_THREADS) void outlier_stats_kernel(const float* __restrict__ sr, int K, int top, int top_pad,
                                                                  float* __restrict__ out_vals, int32_t* __restrict__ out_idx) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_target, s_count, s_fill;
    __shared__ unsigned wave_cnt[OS_THREADS / 64];
    __shared__ unsigned keys[OS_MAX_TOP];
    __shared__ int idxs[OS_MAX_TOP];
    __shared__ int s_zero;
    const int t = threadIdx.x;
    auto key_of = [&](int i) { return __float_as_uint(fabsf(sr[i])); };
    if (t == 0) s_zero = 0, s_count = 0;
    __syncthreads();
    int z = 0;
    for (int i = t; i < K; i += OS_THREADS) z |= sr[i] == 0.f;
    if (z) s_zero = 1;
    // key of ascending rank `target`
    auto select = [&](unsigned target) {
        unsigned prefix = 0, mask = 0;
        for (int pass = 3; pass >= 0; --pass) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            for (int i = t; i < K; i += OS_THREADS) {
                const unsigned k = key_of(i);
                if ((k & mask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255u], 1u);
            }
            __syncthreads();
            if (t < 64) {
                // the bin that holds rank `target`: lane t sums bins 4 t .. 4 t + 3, a wave prefix sum places it
                const unsigned h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
                const unsigned mine = h0 + h1 + h2 + h3;
                unsigned incl = mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned up = __shfl_up(incl, d);
                    if (t >= d) incl += up;
                }
                const unsigned before = incl - mine;
                if (target >= before && target < incl) {
                    unsigned acc = before;
                    int b = 4 * t;
                    if (acc + h0 <= target) { acc += h0, ++b;
                        if (acc + h1 <= target) { acc += h1, ++b;
                            if (acc + h2 <= target) { acc += h2, ++b; } } }
                    s_target = target - acc;
                    s_prefix = prefix | ((unsigned)b << (8 * pass));
                }
            }
            __syncthreads();
            prefix = s_prefix;
            target = s_target;
            mask |= 0xffu << (8 * pass);
            __syncthreads();
        }
        return prefix;
    };
    const unsigned med = select((unsigned)(K - 1) / 2);
    const unsigned thr = select((unsigned)(K - top));          // the top-th largest key
    // strictly larger keys: at most top - 1 of them, in any order (they are sorted below)
    for (int i = t; i < K; i += OS_THREADS) {
        const unsigned k = key_of(i);
        if (k > thr) {
            const unsigned p = atomicAdd(&s_count, 1u);
            keys[p] = k;
            idxs[p] = i;
        }
    }
    __syncthreads();
    if (t == 0) s_fill = s_count;
    __syncthreads();
    // keys equal to the threshold fill the rest, lowest indices first
    for (int base = 0; base < K; base += OS_THREADS) {
        if (s_fill >= (unsigned)top) break;
        const int i = base + t;
        const bool tie = i < K && key_of(i) == thr;
        const unsigned long long b = __ballot(tie);
        if ((t & 63) == 0) wave_cnt[t >> 6] = (unsigned)__popcll(b);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int w = 0; w < OS_THREADS / 64; ++w) {
            if (w < (t >> 6)) before += wave_cnt[w];
            total += wave_cnt[w];
        }
        const unsigned pos = s_fill + before + (unsigned)__popcll(b & ((1ull << (t & 63)) - 1ull));
        if (tie && pos < (unsigned)top) {
            keys[pos] = thr;
            idxs[pos] = i;
        }
        __syncthreads();
        if (t == 0) s_fill += total;
        __syncthreads();
    }
    if (t == 0) {
        out_vals[0] = s_zero ? 1.f : 0.f;
        out_vals[1] = sqrtf(__uint_as_float(med));
    }
    if (t < top) {
        const unsigned k = keys[t];
        const int idx = idxs[t];
        int rank = 0;
        for (int j = 0; j < top; ++j) rank += (keys[j] > k) || (keys[j] == k && idxs[j] < idx);
        out_vals[2 + rank] = sqrtf(__uint_as_float(k));
        out_idx[rank] = idx;
    } else if (t < top_pad) {
        out_vals[2 + t] = 0.f;
        out_idx[t] = 0;
    }
}

extern "C" int slim_outlier_stats(const float* scaler_row, int64_t K, int top, float* out_vals, int32_t* out_idx, void* stream) {
    SLIM_REQUIRE(scaler_row && out_vals && out_idx, "slim_outlier_stats: null pointer");
    SLIM_REQUIRE(K > 0 && K < (1 << 30) && top > 0 && top <= OS_MAX_TOP, "slim_outlier_stats: bad sizes");
    const int have = top < K ? top : (int)K;
    outlier_stats_kernel<<<1, OS_THREADS, 0, (hipStream_t)stream>>>(scaler_row, (int)K, have, top, out_vals, out_idx);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}
