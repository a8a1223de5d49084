// SLiM-Quant: |W| histogram, clipping-threshold search, per-tensor symmetric quantisation
// (reference slim/quantization/quantization.py:6-20, 34-62, 133-217) and the element-wise
// stages of add_lora that surround the low-rank fit (slim/lora.py:146-156, 199-204).
//
// Histogram design.  The weights are bf16, so |W| takes at most 32768 distinct values.  One
// pass over W counts every bit pattern exactly in a 128 KiB LDS table per workgroup (one
// workgroup per CU); no min/max pre-pass is needed and the result does not depend on a bin
// range.  The reference's `torch.histogram(|W|, bins, density=True)` over [min|W|, max|W|]
// is then re-derived from the 32768 exact counts by a single small workgroup, using the same
// fp32 edge formula (step = (hi-lo)/bins; edge_i = fma(step, i, lo) in the lower half,
// fma(-step, bins-i, hi) in the upper half), the same "largest edge <= v" binning with the
// last bin closed, and the same density normalisation (count/total)/width.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include <mutex>

#define HIST_BINS 32768
#define MAX_SEARCH_BINS 20000

// ---- exact |bf16| histogram ------------------------------------------------------------
__global__ __launch_bounds__(1024) void hist_bf16_kernel(const bf16_t* __restrict__ w, int64_t numel,
                                                         uint32_t* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];
    for (int i = threadIdx.x; i < HIST_BINS; i += 1024) lh[i] = 0;
    __syncthreads();
    const int64_t nvec = numel >> 3;
    uint32_t zeros = 0;
    for (int64_t v = (int64_t)blockIdx.x * 1024 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 1024) {
        const u16x8 x = *(const u16x8*)(w + (v << 3));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t b = x[j] & 0x7fffu;
            // zeros (half of a masked matrix) would serialise on one LDS word: count them in a register
            zeros += (b == 0);
            if (b != 0) atomicAdd(&lh[b], 1u);
        }
    }
    // tail elements (numel not a multiple of 8): handled by block 0
    if (blockIdx.x == 0) {
        const int64_t t = (nvec << 3) + threadIdx.x;
        if (t < numel) atomicAdd(&lh[w[t] & 0x7fffu], 1u);
    }
    zeros = wave_sum_u32(zeros);
    if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&lh[0], zeros);
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 1024) {
        const uint32_t c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

extern "C" int slim_hist_abs_bf16(const void* W, int64_t numel, uint32_t* hist, int accumulate, void* stream) {
    SLIM_REQUIRE(W && hist && numel > 0, "slim_hist_abs_bf16: bad arguments");
    SLIM_REQUIRE(((uintptr_t)W % 16) == 0, "slim_hist_abs_bf16: W must be 16-byte aligned");
    SLIM_REQUIRE(numel < ((int64_t)1 << 32), "slim_hist_abs_bf16: more than 2^32 elements");
    hipStream_t st = (hipStream_t)stream;
    SLIM_HIP(slim_allow_dynamic_lds((const void*)hist_bf16_kernel, HIST_BINS * 4));
    if (!accumulate) SLIM_HIP(hipMemsetAsync(hist, 0, HIST_BINS * sizeof(uint32_t), st));
    SLIM_PROF("hist_bf16", stream, 2.0 * numel);
    int64_t blocks = ceil_div64(numel >> 3, 1024 * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;
    hist_bf16_kernel<<<(unsigned)blocks, 1024, HIST_BINS * 4, st>>>((const bf16_t*)W, numel, hist);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- search workspace layout (floats) ---------------------------------------------------
//  [0, B)            pdf
//  [B, 2B)           val (bin centres)
//  [2B, 3B)          total_loss (inf where not evaluated)
//  [3B, 3B+16)       coarse losses (11 used)
//  [3B+16, 3B+32)    coarse indices as int32 (11 used), then scalars
//  then, as int32: the bins with a non-zero count, listed per thread of a candidate's workgroup in the order that thread
//  visits them -- count256[256], count1024[1024], list256[ceil(B/256)][256], list1024[ceil(B/1024)][1024] (entry k of
//  thread t at [k][t]).  A candidate's loss is a sum over ALL bins, thread t of NT adding bins t, t + NT, ... in that
//  order; a bin with count zero adds +0.0 to a double, so leaving it out changes no bit of any partial sum.  The weights
//  are bf16: at 20 000 bins over [min |w|, max |w|] the upper binades hold one bf16 value per ~100 bins, and some 5 % of the
//  bins are occupied.
struct SearchWs {
    float* pdf;
    float* val;
    float* total_loss;
    float* coarse_loss;
    int32_t* coarse_idx;
    int32_t *count256, *count1024, *list256, *list1024;
};
__host__ __device__ static inline SearchWs search_ws(float* base, int B) {
    SearchWs w;
    w.pdf = base;
    w.val = base + B;
    w.total_loss = base + 2 * (int64_t)B;
    w.coarse_loss = base + 3 * (int64_t)B;
    w.coarse_idx = (int32_t*)(base + 3 * (int64_t)B + 16);
    w.count256 = (int32_t*)(base + 3 * (int64_t)B + 32);
    w.count1024 = w.count256 + 256;
    w.list256 = w.count1024 + 1024;
    w.list1024 = w.list256 + (int64_t)((B + 255) / 256) * 256;
    return w;
}

extern "C" int64_t slim_quant_search_workspace_bytes(int num_bins) {
    if (num_bins <= 0) return 0;
    const int64_t B = num_bins;
    return (3 * B + 32 + 256 + 1024 + (B + 255) / 256 * 256 + (B + 1023) / 1024 * 1024) * (int64_t)sizeof(float);
}

__device__ __forceinline__ float edge_at(int i, int bins, float lo, float hi, float step) {
    const int half = (bins + 1) / 2;
    return i < half ? __fmaf_rn(step, (float)i, lo) : __fmaf_rn(-step, (float)(bins - i), hi);
}

// Single workgroup: rebuild the reference's density histogram from the exact pattern counts.
__global__ __launch_bounds__(1024) void rebin_kernel(const uint32_t* __restrict__ hist, int bins, float* ws_base) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];  // bins counts
    __shared__ int s_lo, s_hi;
    __shared__ unsigned long long s_total;
    SearchWs ws = search_ws(ws_base, bins);
    if (threadIdx.x == 0) {
        s_lo = HIST_BINS;
        s_hi = -1;
        s_total = 0ull;
    }
    for (int i = threadIdx.x; i < bins; i += 1024) cnt[i] = 0;
    __syncthreads();
    // every thread's 32 pattern counts come from memory in one batch of independent loads and stay in registers for both
    // passes: the kernel is a single workgroup, so each dependent trip to memory would be paid in full
    uint32_t cnts[HIST_BINS / 1024];
#pragma unroll
    for (int u = 0; u < HIST_BINS / 1024; ++u) cnts[u] = hist[threadIdx.x + 1024 * u];
    int lo_p = HIST_BINS, hi_p = -1;
    unsigned long long tot = 0;
#pragma unroll
    for (int u = 0; u < HIST_BINS / 1024; ++u) {
        const int p = threadIdx.x + 1024 * u;
        if (cnts[u]) {
            lo_p = min(lo_p, p);
            hi_p = max(hi_p, p);
            tot += cnts[u];
        }
    }
    lo_p = wave_min_i32(lo_p);
    hi_p = wave_max_i32(hi_p);
    tot = wave_sum_u64(tot);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_lo, lo_p);
        atomicMax(&s_hi, hi_p);
        atomicAdd(&s_total, tot);
    }
    __syncthreads();
    float lo = bf16_to_f32((bf16_t)s_lo), hi = bf16_to_f32((bf16_t)s_hi);
    if (s_lo == s_hi) {  // constant |W|: the reference widens the range by half a unit each side
        lo -= 0.5f;
        hi += 0.5f;
    }
    const float step = (hi - lo) / (float)bins;
#pragma unroll
    for (int u = 0; u < HIST_BINS / 1024; ++u) {
        const uint32_t c = cnts[u];
        if (!c) continue;
        const float v = bf16_to_f32((bf16_t)(threadIdx.x + 1024 * u));
        int pos = (int)((v - lo) / (hi - lo) * (float)bins);
        pos = max(0, min(pos, bins - 1));
        while (pos > 0 && v < edge_at(pos, bins, lo, hi, step)) --pos;
        while (pos + 1 < bins && v >= edge_at(pos + 1, bins, lo, hi, step)) ++pos;
        atomicAdd(&cnt[pos], c);
    }
    __syncthreads();
    const float total = (float)s_total;
    for (int i = threadIdx.x; i < bins; i += 1024) {
        const float e0 = edge_at(i, bins, lo, hi, step), e1 = edge_at(i + 1, bins, lo, hi, step);
        const float width = e1 - e0;
        ws.pdf[i] = ((float)cnt[i] / total) / width;
        ws.val[i] = (e0 + e1) / 2.0f;
        ws.total_loss[i] = __builtin_inff();
    }
    if (threadIdx.x < 16) {
        ws.coarse_loss[threadIdx.x] = __builtin_inff();
        ws.coarse_idx[threadIdx.x] = 0;
    }
    // the occupied bins in the order the threads of a candidate's workgroup visit them (cnt[] is complete since the barrier
    // above and is only read from here on)
    {
        int k = 0;
        for (int i = threadIdx.x; i < bins; i += 1024)
            if (cnt[i]) ws.list1024[(k++) * 1024 + threadIdx.x] = i;
        ws.count1024[threadIdx.x] = k;
    }
    if (threadIdx.x < 256) {
        int k = 0;
        for (int i = threadIdx.x; i < bins; i += 256)
            if (cnt[i]) ws.list256[(k++) * 256 + threadIdx.x] = i;
        ws.count256[threadIdx.x] = k;
    }
}

// fmodf(a, b) for finite a >= 0, b > 0 whose quotient is below 2^23, without the library routine's loop: with n =
// trunc(a / b) the remainder a - n b is a multiple of the smaller operand's unit and below b, hence representable, so one
// fused multiply-add returns it exactly.  The division is rounded to nearest and can only overshoot an integer, never
// fall below one (integers of that size are representable), so n from floorf(a / b) is at most one too large, which shows as
// a negative remainder.  Quotients of 2^23 and more (never met in the scale search, whose quotients stay below 2^num_bits
// for bins under the candidate) take the library routine.
__device__ __forceinline__ float fmod_pos_f32(float a, float b) {
    const float qf = a / b;
    if (!(qf < 8388608.f)) return fmodf(a, b);
    float n = floorf(qf);
    float r = __builtin_fmaf(-n, b, a);
    if (r < 0.f) {
        n -= 1.f;
        r = __builtin_fmaf(-n, b, a);
    } else if (r >= b) {            // (cannot happen with a correctly rounded division; kept as a guard)
        n += 1.f;
        r = __builtin_fmaf(-n, b, a);
    }
    return r;
}

// floor division of non-negative fp32 values as the reference's `//` computes it
// (fmod-based, with the half-way correction), b > 0.
__device__ __forceinline__ float floor_div_f32(float a, float b) {
    const float mod = fmod_pos_f32(a, b);
    float div = (a - mod) / b;
    if (mod != 0.f && mod < 0.f) div -= 1.f;
    if (div == 0.f) return 0.f;
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.f;
    return fl;
}

// The same floor division without either of its two correctly rounded divisions, for finite a >= 0, b > 0 and rb any fp32
// approximation of 1 / b good to a few units in the last place.  With n = floor(a / b) (of the real quotient) and the exact
// remainder r = a - n b, the reference's value is n whatever the roundings of its own steps do: a - r is n b rounded once,
// the quotient by b is n (1 + d) with |d| < 2^-22, i.e. within a quarter of n for n < 2^20, and the final correction
// (`div - floor(div) > 0.5`) turns a value just below n back into n; n = 0 gives a - r = 0 and the early return.  And n
// follows from the estimate floor(a rb), which for quotients below 2^20 is off by at most one (absolute error below
// 2^-2), through one fused multiply-add: a remainder below zero or not below b says which way.  Quotients of 2^20 and
// more (the search's stay below 2^num_bits) take the literal form: from 2^22 on the reference's own roundings no longer
// return n.
// tests/test_hip_kernels.py compares the two on the search's operands and on adversarial ones (slim_floordiv_probe_f32).
__device__ __forceinline__ float floor_quot_f32(float a, float b, float rb) {
    const float qf = a * rb;
    if (!(qf < 1048576.f)) return floor_div_f32(a, b);
    float n = floorf(qf);
    const float r = __builtin_fmaf(-n, b, a);
    if (r < 0.f) n -= 1.f;
    else if (r >= b) n += 1.f;
    return n;
}

// loss of one candidate index, reduced over the workgroup (256 threads); element-wise terms
// are formed with the reference's fp32 operations, the sum is carried in fp64.
template <int NT>
__device__ float candidate_loss(const SearchWs& ws, int bins, int index, int q, double* red) {
    const float alpha = ws.val[index];
    const float dx = ws.val[1] - ws.val[0];
    const float step = alpha / (float)(1 << q);
    const float rstep = 1.0f / step;
    double sq = 0.0, sc = 0.0;
    static_assert(NT == 256 || NT == 1024, "the occupied bins are listed for workgroups of 256 and 1024 threads");
    // The regular case -- a positive finite step and no negative bin centre (min |w| >= 0 unless a constant |w| had its range
    // widened) -- visits the occupied bins only: every term of an empty bin is then +0.0 (finite d, p = 0).  Otherwise
    // (a candidate at or below zero in the widened range: 0 / 0 and inf * 0 appear, and 0 * NaN is not 0) every bin is
    // visited and the floor division takes its literal form.
    if (step > 0.f && step < __builtin_inff() && ws.val[0] >= 0.f) {
        const int32_t* list = NT == 256 ? ws.list256 : ws.list1024;
        const int occupied = (NT == 256 ? ws.count256 : ws.count1024)[threadIdx.x];
        for (int k = 0; k < occupied; ++k) {
            const int j = list[k * NT + threadIdx.x];
            const float v = ws.val[j], p = ws.pdf[j];
            if (j < index) {
                const float qv = floor_quot_f32(v, step, rstep) * step;
                const float d = v - qv;
                sq += (double)(p * (d * d));
            } else {
                const float d = v - alpha;
                sc += (double)(p * (d * d));
            }
        }
    } else {
        for (int j = threadIdx.x; j < bins; j += NT) {
            const float v = ws.val[j], p = ws.pdf[j];
            if (j < index) {
                const float qv = (step == 0.f ? (v / step) : floor_div_f32(v, step)) * step;
                const float d = v - qv;
                sq += (double)(p * (d * d));
            } else {
                const float d = v - alpha;
                sc += (double)(p * (d * d));
            }
        }
    }
    sq = wave_sum_d(sq);
    sc = wave_sum_d(sc);
    constexpr int NW = NT / 64;
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wv] = sq;
        red[NW + wv] = sc;
    }
    __syncthreads();
    double tq = 0.0, tc = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        tq += red[w];
        tc += red[NW + w];
    }
    __syncthreads();
    return (float)tq * dx + (float)tc * dx;
}

// coarse scan: block b evaluates index b * (bins / 10).  Eleven workgroups at most, so the launch lasts as long as one
// candidate: 1024 threads share its bins.
__global__ __launch_bounds__(1024) void search_coarse_kernel(float* ws_base, int bins, int q) {
    __shared__ double red[32];
    SearchWs ws = search_ws(ws_base, bins);
    const int i = blockIdx.x * (bins / 10);
    if (i >= bins) return;
    const float l = candidate_loss<1024>(ws, bins, i, q, red);
    if (threadIdx.x == 0) {
        ws.coarse_loss[blockIdx.x] = l;
        ws.coarse_idx[blockIdx.x] = i;
    }
}

// fine scan of [indices[max(tp-1,0)], indices[min(tp+1,10)]), block b evaluates start + b (thousands of workgroups:
// 256 threads each over the occupied bins -- a floor, a fused multiply-add, two comparisons and the squared difference per
// (candidate, bin) pair.  While every bin was visited the launch was bound by arithmetic: 64 us at 27 lane-cycles per pair
// with the floor division's two correctly rounded divisions, 44 us without them; four candidates per workgroup sharing the
// fetches of the bins, and the fetches batched eight deep, both measured slower then)
__global__ __launch_bounds__(256) void search_fine_kernel(float* ws_base, int bins, int q) {
    __shared__ double red[8];
    SearchWs ws = search_ws(ws_base, bins);
    int tp = 0;
    float best = ws.coarse_loss[0];
    for (int j = 1; j < 11; ++j) {
        const float l = ws.coarse_loss[j];
        if (l < best) {  // first minimum wins, as torch.min does
            best = l;
            tp = j;
        }
    }
    const int start = ws.coarse_idx[max(tp - 1, 0)];
    const int end = ws.coarse_idx[min(tp + 1, 10)];
    const int i = start + (int)blockIdx.x;
    if (i >= end) return;
    const float l = candidate_loss<256>(ws, bins, i, q, red);
    if (threadIdx.x == 0) ws.total_loss[i] = l;
}

// argmin (first occurrence) of total_loss; outputs alpha = val[idx], idx, scaling factor maxq / alpha.
__global__ __launch_bounds__(1024) void search_argmin_kernel(const float* ws_base, int bins, int num_bits,
                                                             float* alpha_out, int32_t* idx_out, float* sf_out) {
    __shared__ float bl[16];
    __shared__ int bi[16];
    SearchWs ws = search_ws((float*)ws_base, bins);
    float best = __builtin_inff();
    int besti = 0x7fffffff;
    for (int i = threadIdx.x; i < bins; i += 1024) {
        const float l = ws.total_loss[i];
        if (l < best) {
            best = l;
            besti = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ol = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ol < best || (ol == best && oi < besti)) {
            best = ol;
            besti = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        bl[threadIdx.x >> 6] = best;
        bi[threadIdx.x >> 6] = besti;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bl[w] < best || (bl[w] == best && bi[w] < besti)) {
                best = bl[w];
                besti = bi[w];
            }
        if (besti == 0x7fffffff) besti = 0;  // every loss inf: index 0, like torch.min on an all-inf vector
        const float alpha = ws.val[besti];
        alpha_out[0] = alpha;
        if (idx_out) idx_out[0] = besti;
        // `max_q / abs_max` with a tensor divisor evaluates as reciprocal(abs_max) * max_q (two roundings)
        if (sf_out) sf_out[0] = (1.0f / alpha) * (float)((1 << (num_bits - 1)) - 1);
    }
}

// Self-check of fmod_pos_f32 (the scale search's floor division stands on it): out_fast[i] = fmod_pos_f32(a[i], b[i]),
// out_ref[i] = fmodf(a[i], b[i]).  No counterpart in the reference.
__global__ void fmod_probe_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ out_fast,
                                  float* __restrict__ out_ref) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_fast[i] = fmod_pos_f32(a[i], b[i]);
    out_ref[i] = fmodf(a[i], b[i]);
}
extern "C" int slim_fmod_probe_f32(const float* a, const float* b, int64_t n, float* out_fast, float* out_ref, void* stream) {
    SLIM_REQUIRE(a && b && out_fast && out_ref && n > 0, "slim_fmod_probe_f32: bad arguments");
    fmod_probe_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, out_fast, out_ref);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Self-check of floor_quot_f32: out_fast[i] = floor_quot_f32(a[i], b[i], 1 / b[i]); out_ref[i] = the reference's floor
// division written out step by step with the library remainder (fmodf) and two IEEE divisions.
__global__ void floordiv_probe_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ out_fast,
                                      float* __restrict__ out_ref) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    out_fast[i] = floor_quot_f32(x, y, 1.0f / y);
    const float mod = fmodf(x, y);
    float div = (x - mod) / y;
    if (mod != 0.f && mod < 0.f) div -= 1.f;
    float fl = 0.f;
    if (div != 0.f) {
        fl = floorf(div);
        if (div - fl > 0.5f) fl += 1.f;
    }
    out_ref[i] = fl;
}
extern "C" int slim_floordiv_probe_f32(const float* a, const float* b, int64_t n, float* out_fast, float* out_ref, void* stream) {
    SLIM_REQUIRE(a && b && out_fast && out_ref && n > 0, "slim_floordiv_probe_f32: bad arguments");
    floordiv_probe_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, out_fast, out_ref);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// the four launches of the search on a workspace of slim_quant_search_workspace_bytes(num_bins) bytes
static int launch_search(const uint32_t* hist, int num_bins, int num_bits, float* alpha_out, int32_t* idx_out,
                         float* sf_out, float* ws, hipStream_t st) {
    SLIM_HIP(slim_allow_dynamic_lds((const void*)rebin_kernel, MAX_SEARCH_BINS * 4));
    rebin_kernel<<<1, 1024, (size_t)num_bins * 4, st>>>(hist, num_bins, ws);
    SLIM_LAUNCH_CHECK();
    const int stride = num_bins / 10;
    const int ncoarse = (num_bins + stride - 1) / stride;  // 10 or 11 points
    search_coarse_kernel<<<ncoarse, 1024, 0, st>>>(ws, num_bins, num_bits);
    SLIM_LAUNCH_CHECK();
    search_fine_kernel<<<2 * stride, 256, 0, st>>>(ws, num_bins, num_bits);
    SLIM_LAUNCH_CHECK();
    search_argmin_kernel<<<1, 1024, 0, st>>>(ws, num_bins, num_bits, alpha_out, idx_out, sf_out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_quant_search(const uint32_t* hist, int num_bins, int num_bits, float* alpha_out,
                                 int32_t* idx_out, float* sf_out, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    SLIM_REQUIRE(hist && alpha_out && workspace, "slim_quant_search: null pointer");
    SLIM_REQUIRE(num_bins >= 10 && num_bins <= MAX_SEARCH_BINS, "slim_quant_search: num_bins must be in [10, 20000]");
    SLIM_REQUIRE(num_bits >= 2 && num_bits <= 16, "slim_quant_search: num_bits out of range");
    SLIM_REQUIRE(workspace_bytes >= slim_quant_search_workspace_bytes(num_bins), "slim_quant_search: workspace too small");
    SLIM_PROF("quant_search", stream);
    return launch_search(hist, num_bins, num_bits, alpha_out, idx_out, sf_out, (float*)workspace, (hipStream_t)stream);
}

// ---- the search without a caller workspace and without any allocation ------------------------------------
// Search workspace for the largest histogram the search supports (102 016 words = 408 064 bytes for 20 000 bins), as a
// device array of this library's code object: every GPU the library runs on holds its own copy, at its own address,
// until the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress; the device
// address it returns (per device) is what the kernels get.
#define STATIC_SEARCH_FLOATS (3 * MAX_SEARCH_BINS + 32 + 256 + 1024 + (MAX_SEARCH_BINS + 255) / 256 * 256 + (MAX_SEARCH_BINS + 1023) / 1024 * 1024)
#define SLIM_MAX_DEVICES 64
static __device__ __align__(256) float scratch[STATIC_SEARCH_FLOATS];
static float* g_scratch_ptr[SLIM_MAX_DEVICES];  // NULL = not resolved on that device yet
static std::mutex g_scratch_mu;

extern "C" int slim_quant_search_noalloc(const uint32_t* hist, int num_bins, int num_bits, float* alpha_out,
                                         int32_t* idx_out, float* sf_out, void* stream) {
    SLIM_REQUIRE(hist && alpha_out, "slim_quant_search_noalloc: null pointer");
    SLIM_REQUIRE(num_bins <= MAX_SEARCH_BINS,
                 "slim_quant_search_noalloc: num_bins %d exceeds the %d bins the library's static scratch holds", num_bins,
                 MAX_SEARCH_BINS);
    SLIM_REQUIRE(num_bins >= 10, "slim_quant_search_noalloc: num_bins must be at least 10");
    SLIM_REQUIRE(num_bits >= 2 && num_bits <= 16, "slim_quant_search_noalloc: num_bits out of range");
    int dev = -1;
    SLIM_HIP(hipGetDevice(&dev));
    SLIM_REQUIRE(dev >= 0 && dev < SLIM_MAX_DEVICES, "slim_quant_search_noalloc: device index %d not supported", dev);
    float* ws = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[dev]) {  // first call on this device: resolve this device's copy of the array
            void* p = nullptr;
            SLIM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            SLIM_REQUIRE(p != nullptr, "slim_quant_search_noalloc: the scratch array has no address on device %d", dev);
            g_scratch_ptr[dev] = (float*)p;
        }
        ws = g_scratch_ptr[dev];
    }
    // rebin_kernel rewrites every word the later kernels read (pdf, val, total_loss below num_bins, the 32 coarse slots, the
    // counts of occupied bins and the list entries below those counts), so what an earlier, larger search left is never
    // looked at
    return launch_search(hist, num_bins, num_bits, alpha_out, idx_out, sf_out, ws, (hipStream_t)stream);
}

// abs-max of a bf16 tensor from its pattern histogram (the non-SLiM-Quant scale, quantization.py:190):
// scaling factor = bf16(maxq / bf16 absmax), returned as fp32.
__global__ __launch_bounds__(1024) void absmax_from_hist_kernel(const uint32_t* __restrict__ hist, int num_bits,
                                                                float* alpha_out, float* sf_out) {
    __shared__ int s_hi;
    if (threadIdx.x == 0) s_hi = 0;
    __syncthreads();
    int hi_p = 0;
    for (int p = threadIdx.x; p < HIST_BINS; p += 1024)
        if (hist[p]) hi_p = max(hi_p, p);
    atomicMax(&s_hi, hi_p);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = bf16_to_f32((bf16_t)s_hi);
        alpha_out[0] = a;
        // bf16 reciprocal, then bf16 product with max_q
        if (sf_out) sf_out[0] = round_bf16(round_bf16(1.0f / a) * (float)((1 << (num_bits - 1)) - 1));
    }
}

extern "C" int slim_absmax_scale(const uint32_t* hist, int num_bits, float* alpha_out, float* sf_out, void* stream) {
    SLIM_REQUIRE(hist && alpha_out, "slim_absmax_scale: null pointer");
    absmax_from_hist_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(hist, num_bits, alpha_out, sf_out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- per-tensor quantise / dequantise ---------------------------------------------------
// code = clamp(round_half_even(float(bf16(float(w) * sf))), -2^(b-1), 2^(b-1)-1)
__device__ __forceinline__ float quant_code(bf16_t w, float sf, float qmin, float qmax) {
    const float prod = round_bf16(bf16_to_f32(w) * sf);
    return fminf(fmaxf(rintf(prod), qmin), qmax);
}

__global__ __launch_bounds__(256) void quantize_pt_kernel(const bf16_t* __restrict__ w, int64_t numel,
                                                          const float* __restrict__ sf_p, int num_bits,
                                                          int8_t* __restrict__ codes) {
    const float sf = sf_p[0];
    const float qmax = (float)((1 << (num_bits - 1)) - 1), qmin = -qmax - 1.f;
    const int64_t nvec = numel >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const u16x8 x = *(const u16x8*)(w + (v << 3));
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo |= ((uint32_t)(uint8_t)(int8_t)(int)quant_code(x[j], sf, qmin, qmax)) << (8 * j);
            hi |= ((uint32_t)(uint8_t)(int8_t)(int)quant_code(x[4 + j], sf, qmin, qmax)) << (8 * j);
        }
        *(uint2*)(codes + (v << 3)) = uint2{lo, hi};
    }
    if (blockIdx.x == 0) {
        const int64_t t = (nvec << 3) + threadIdx.x;
        if (t < numel) codes[t] = (int8_t)(int)quant_code(w[t], sf, qmin, qmax);
    }
}

extern "C" int slim_quantize_pt(const void* W, int64_t numel, const float* sf, int num_bits, int8_t* codes,
                                void* stream) {
    SLIM_REQUIRE(W && sf && codes && numel > 0, "slim_quantize_pt: bad arguments");
    SLIM_REQUIRE(num_bits >= 2 && num_bits <= 8, "slim_quantize_pt: num_bits must be in [2, 8]");
    SLIM_REQUIRE(((uintptr_t)W % 16) == 0 && ((uintptr_t)codes % 8) == 0, "slim_quantize_pt: misaligned pointer");
    SLIM_PROF("quantize_pt", stream, 3.0 * numel);
    int64_t blocks = ceil_div64(numel >> 3, 256);
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    quantize_pt_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)W, numel, sf, num_bits, codes);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// out = float(code) / sf, stored as fp32 (out_bf16 = 0) or rounded to bf16 (out_bf16 = 1)
__global__ __launch_bounds__(256) void dequantize_pt_kernel(const int8_t* __restrict__ codes, int64_t numel,
                                                            const float* __restrict__ sf_p, void* __restrict__ out,
                                                            int out_bf16) {
    const float sf = sf_p[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float v = (float)codes[i] / sf;
        if (out_bf16) ((bf16_t*)out)[i] = f32_to_bf16(v);
        else ((float*)out)[i] = v;
    }
}

extern "C" int slim_dequantize_pt(const int8_t* codes, int64_t numel, const float* sf, void* out, int out_bf16,
                                  void* stream) {
    SLIM_REQUIRE(codes && sf && out && numel > 0, "slim_dequantize_pt: bad arguments");
    int64_t blocks = ceil_div64(numel, 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    dequantize_pt_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(codes, numel, sf, out, out_bf16);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Fused quantise -> dequantise -> bf16 (the final `weight.data` of add_lora, lora.py:203-209),
// optionally also emitting the int8 codes.
__global__ __launch_bounds__(256) void fake_quant_pt_kernel(const bf16_t* __restrict__ w, int64_t numel,
                                                            const float* __restrict__ sf_p, int num_bits,
                                                            bf16_t* __restrict__ out, int8_t* __restrict__ codes) {
    const float sf = sf_p[0];
    const float qmax = (float)((1 << (num_bits - 1)) - 1), qmin = -qmax - 1.f;
    const int64_t nvec = numel >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const u16x8 x = *(const u16x8*)(w + (v << 3));
        u16x8 o;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float c = quant_code(x[j], sf, qmin, qmax);
            o[j] = f32_to_bf16(c / sf);
            const uint32_t b = ((uint32_t)(uint8_t)(int8_t)(int)c) << (8 * (j & 3));
            if (j < 4) lo |= b;
            else hi |= b;
        }
        *(u16x8*)(out + (v << 3)) = o;
        if (codes) *(uint2*)(codes + (v << 3)) = uint2{lo, hi};
    }
    if (blockIdx.x == 0) {
        const int64_t t = (nvec << 3) + threadIdx.x;
        if (t < numel) {
            const float c = quant_code(w[t], sf, qmin, qmax);
            out[t] = f32_to_bf16(c / sf);
            if (codes) codes[t] = (int8_t)(int)c;
        }
    }
}

extern "C" int slim_fake_quant_pt(const void* W, int64_t numel, const float* sf, int num_bits, void* out_bf16,
                                  int8_t* codes, void* stream) {
    SLIM_REQUIRE(W && sf && out_bf16 && numel > 0, "slim_fake_quant_pt: bad arguments");
    SLIM_REQUIRE(num_bits >= 2 && num_bits <= 8, "slim_fake_quant_pt: num_bits must be in [2, 8]");
    SLIM_REQUIRE(((uintptr_t)W % 16) == 0 && ((uintptr_t)out_bf16 % 16) == 0 && (!codes || ((uintptr_t)codes % 8) == 0),
                 "slim_fake_quant_pt: misaligned pointer");
    SLIM_PROF("fake_quant_pt", stream, (codes ? 5.0 : 4.0) * numel);
    int64_t blocks = ceil_div64(numel >> 3, 256);
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    fake_quant_pt_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)W, numel, sf, num_bits,
                                                                           (bf16_t*)out_bf16, codes);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- error matrix of add_lora (lora.py:146-156) ------------------------------------------
// E = float(W) * s - mask0((code / sf) * s), s = sqrt(scaler_row) per column (s == NULL: ones,
// the "naive" branch lora.py:158-169).  sf == NULL: no quantisation (E = masked part of W * s).
// Written as fp32 and/or bf16.
__global__ __launch_bounds__(256) void error_matrix_kernel(const bf16_t* __restrict__ w, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ s, const float* __restrict__ sf_p,
                                                           int num_bits, int64_t N, int64_t K,
                                                           float* __restrict__ e32, bf16_t* __restrict__ e16,
                                                           const bf16_t* __restrict__ wdeq,
                                                           const uint8_t* __restrict__ colflags, int deq_bf16) {
    const float sf = sf_p ? sf_p[0] : 0.f;
    const float qmax = (float)((1 << (num_bits - 1)) - 1), qmin = -qmax - 1.f;
    const int64_t cpr = K >> 3, total = N * cpr;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / cpr, c0 = (t - row * cpr) << 3, off = row * K + c0;
        const u16x8 x = *(const u16x8*)(w + off);
        const uint2 mk = *(const uint2*)(mask + off);
        u16x8 xd = x;
        if (wdeq) xd = *(const u16x8*)(wdeq + off);
        uint2 cf = uint2{0u, 0u};
        if (colflags) cf = *(const uint2*)(colflags + c0);
        float sv[8];
        if (s) {
            const f32x4 a = *(const f32x4*)(s + c0), b = *(const f32x4*)(s + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sv[j] = a[j];
                sv[4 + j] = b[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = 1.f;
        }
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wf = bf16_to_f32(x[j]);
            const float metric = wf * sv[j];
            const bool pruned = ((j < 4 ? mk.x >> (8 * j) : mk.y >> (8 * (j - 4))) & 0xffu) != 0;
            float nw;
            if (pruned) nw = 0.f;
            else if (wdeq) nw = bf16_to_f32(xd[j]) * sv[j];
            else if (sf_p) {
                // flagged columns (quantization.py:160-170): the code of the doubled weight, the dequantised value halved
                // a bf16 scale (abs-max, quantization.py:190-194) makes `codes / scale` a bf16 tensor (:212)
                const bool scaled = ((j < 4 ? cf.x >> (8 * j) : cf.y >> (8 * (j - 4))) & 0xffu) != 0;
                float dq = quant_code(scaled ? f32_to_bf16(wf * 2.f) : x[j], sf, qmin, qmax) / sf;
                if (deq_bf16) dq = round_bf16(dq);
                if (scaled) dq *= 0.5f;
                nw = dq * sv[j];
            } else nw = metric;
            e[j] = metric - nw;
        }
        if (e32) {
            *(f32x4*)(e32 + off) = f32x4{e[0], e[1], e[2], e[3]};
            *(f32x4*)(e32 + off + 4) = f32x4{e[4], e[5], e[6], e[7]};
        }
        if (e16) {
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16(e[j]);
            *(u16x8*)(e16 + off) = o;
        }
    }
}

extern "C" int slim_error_matrix(const void* W, const uint8_t* mask, const float* s, const float* sf, int num_bits,
                                 int64_t N, int64_t K, float* E_f32, void* E_bf16, void* stream) {
    SLIM_REQUIRE(W && mask && (E_f32 || E_bf16) && N > 0 && K > 0, "slim_error_matrix: bad arguments");
    SLIM_REQUIRE(K % 8 == 0, "slim_error_matrix: K must be a multiple of 8");
    SLIM_REQUIRE(!sf || (num_bits >= 2 && num_bits <= 8), "slim_error_matrix: num_bits must be in [2, 8]");
    SLIM_PROF("error_matrix", stream, (3.0 + (E_f32 ? 4.0 : 0.0) + (E_bf16 ? 2.0 : 0.0)) * N * K);
    int64_t blocks = ceil_div64(N * (K >> 3), 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    error_matrix_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)W, mask, s, sf, num_bits, N, K,
                                                                          E_f32, (bf16_t*)E_bf16, nullptr, nullptr, 0);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// slim_error_matrix in its general form.  col_flags (may be NULL): col_flags[c] != 0 marks the input channels whose
// weights are quantised at twice their size and halved after dequantisation (scale_important_weights).  deq_bf16: the
// dequantised value code / sf is rounded to bf16 before it meets s (what the reference's abs-max quantiser, whose scale is
// a bf16 tensor, leaves; SLiM-Quant scales are fp32 and leave fp32 values).
extern "C" int slim_error_matrix_cols(const void* W, const uint8_t* mask, const float* s, const float* sf, int num_bits,
                                      const uint8_t* col_flags, int deq_bf16, int64_t N, int64_t K, float* E_f32, void* E_bf16,
                                      void* stream) {
    SLIM_REQUIRE(W && mask && sf && (E_f32 || E_bf16) && N > 0 && K > 0, "slim_error_matrix_cols: bad arguments");
    SLIM_REQUIRE(K % 8 == 0 && ((uintptr_t)col_flags % 8) == 0, "slim_error_matrix_cols: K must be a multiple of 8");
    SLIM_REQUIRE(num_bits >= 2 && num_bits <= 8, "slim_error_matrix_cols: num_bits must be in [2, 8]");
    SLIM_PROF("error_matrix", stream, (3.0 + (E_f32 ? 4.0 : 0.0) + (E_bf16 ? 2.0 : 0.0)) * N * K);
    int64_t blocks = ceil_div64(N * (K >> 3), 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    error_matrix_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)W, mask, s, sf, num_bits, N, K,
                                                                          E_f32, (bf16_t*)E_bf16, nullptr, col_flags, deq_bf16);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// The same matrix (per-tensor quantiser or none) written as E [N, K] and E^T [K, N] in one pass: 64 x 64 tiles, rows of E
// straight from the registers, the transposed tile through LDS (the separate transpose read E a second time).  The element
// arithmetic is error_matrix_kernel's.
__global__ __launch_bounds__(256) void error_matrix_t_kernel(const bf16_t* __restrict__ w, const uint8_t* __restrict__ mask,
                                                             const float* __restrict__ s, const float* __restrict__ sf_p,
                                                             int num_bits, int64_t N, int64_t K, bf16_t* __restrict__ e16,
                                                             bf16_t* __restrict__ et16, int deq_bf16) {
    __shared__ bf16_t tile[64][72];      // row stride 144 bytes: 16-byte rows for the vector stores, odd multiple of 16 bytes
    const float sf = sf_p ? sf_p[0] : 0.f;
    const float qmax = (float)((1 << (num_bits - 1)) - 1), qmin = -qmax - 1.f;
    const int64_t n0 = (int64_t)blockIdx.y * 64, k0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x, c = (t & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = (t >> 3) + 32 * pass;
        const int64_t row = n0 + r, col = k0 + c;
        u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < N && col < K) {
            const int64_t off = row * K + col;
            const u16x8 x = *(const u16x8*)(w + off);
            const uint2 mk = *(const uint2*)(mask + off);
            float sv[8];
            if (s) {
                const f32x4 a = *(const f32x4*)(s + col), b = *(const f32x4*)(s + col + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sv[j] = a[j];
                    sv[4 + j] = b[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) sv[j] = 1.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wf = bf16_to_f32(x[j]);
                const float metric = wf * sv[j];
                const bool pruned = ((j < 4 ? mk.x >> (8 * j) : mk.y >> (8 * (j - 4))) & 0xffu) != 0;
                float nw;
                if (pruned) nw = 0.f;
                else if (sf_p) {
                    float dq = quant_code(x[j], sf, qmin, qmax) / sf;
                    if (deq_bf16) dq = round_bf16(dq);
                    nw = dq * sv[j];
                } else nw = metric;
                o[j] = f32_to_bf16(metric - nw);
            }
            *(u16x8*)(e16 + off) = o;
        }
        *(u16x8*)&tile[r][c] = o;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int kc = (t >> 3) + 32 * pass;          // column of the tile = row of E^T
        const int64_t col = k0 + kc, row = n0 + c;
        if (col >= K || row >= N) continue;
        u16x8 o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = tile[c + u][kc];
        *(u16x8*)(et16 + col * N + row) = o;
    }
}

extern "C" int slim_error_matrix_t(const void* W, const uint8_t* mask, const float* s, const float* sf, int num_bits, int64_t N,
                                   int64_t K, void* E_bf16, void* Et_bf16, void* stream) {
    return slim_error_matrix_t_rounded(W, mask, s, sf, num_bits, 0, N, K, E_bf16, Et_bf16, stream);
}

extern "C" int slim_error_matrix_t_rounded(const void* W, const uint8_t* mask, const float* s, const float* sf, int num_bits,
                                           int deq_bf16, int64_t N, int64_t K, void* E_bf16, void* Et_bf16, void* stream) {
    SLIM_REQUIRE(W && mask && E_bf16 && Et_bf16 && N > 0 && K > 0, "slim_error_matrix_t: bad arguments");
    SLIM_REQUIRE(K % 8 == 0 && N % 8 == 0, "slim_error_matrix_t: N and K must be multiples of 8");
    SLIM_REQUIRE(!sf || (num_bits >= 2 && num_bits <= 8), "slim_error_matrix_t: num_bits must be in [2, 8]");
    SLIM_REQUIRE(ceil_div64(N, 64) <= 65535, "slim_error_matrix_t: too many rows");
    SLIM_PROF("error_matrix", stream, 7.0 * N * K);
    dim3 grid((unsigned)ceil_div64(K, 64), (unsigned)ceil_div64(N, 64));
    error_matrix_t_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)W, mask, s, sf, num_bits, N, K, (bf16_t*)E_bf16,
                                                                 (bf16_t*)Et_bf16, deq_bf16);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Same with the dequantised weight given as a tensor (any quantiser, e.g. the tiled group quantiser whose
// dequantised values are bf16): E = W * s - mask0(Wdeq * s).
extern "C" int slim_error_matrix_deq(const void* W, const void* Wdeq, const uint8_t* mask, const float* s, int64_t N,
                                     int64_t K, float* E_f32, void* E_bf16, void* stream) {
    SLIM_REQUIRE(W && Wdeq && mask && (E_f32 || E_bf16) && N > 0 && K > 0, "slim_error_matrix_deq: bad arguments");
    SLIM_REQUIRE(K % 8 == 0, "slim_error_matrix_deq: K must be a multiple of 8");
    SLIM_PROF("error_matrix", stream, (5.0 + (E_f32 ? 4.0 : 0.0) + (E_bf16 ? 2.0 : 0.0)) * N * K);
    int64_t blocks = ceil_div64(N * (K >> 3), 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    error_matrix_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)W, mask, s, nullptr, 4, N, K, E_f32,
                                                                          (bf16_t*)E_bf16, (const bf16_t*)Wdeq, nullptr, 0);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- residual weight (lora.py:199-201): Wn = bf16(float(W) - float(LR)), pruned entries 0 -----
// Optionally accumulates the |bf16| pattern histogram of Wn for the second scale search through
// a per-workgroup LDS table, saving one full read of Wn.
template <bool WITH_HIST>
__global__ __launch_bounds__(1024) void residual_weight_kernel(const bf16_t* __restrict__ w,
                                                               const bf16_t* __restrict__ lr,
                                                               const uint8_t* __restrict__ mask, int64_t numel,
                                                               bf16_t* __restrict__ out, uint32_t* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];
    if (WITH_HIST) {
        for (int i = threadIdx.x; i < HIST_BINS; i += 1024) lh[i] = 0;
        __syncthreads();
    }
    uint32_t zeros = 0;
    const int64_t nvec = numel >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 1024 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 1024) {
        const int64_t off = v << 3;
        const u16x8 x = *(const u16x8*)(w + off);
        const u16x8 l = *(const u16x8*)(lr + off);
        const uint2 mk = *(const uint2*)(mask + off);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool pruned = ((j < 4 ? mk.x >> (8 * j) : mk.y >> (8 * (j - 4))) & 0xffu) != 0;
            o[j] = pruned ? (bf16_t)0 : f32_to_bf16(bf16_to_f32(x[j]) - bf16_to_f32(l[j]));
            if (WITH_HIST) {
                const uint32_t b = o[j] & 0x7fffu;
                zeros += (b == 0);
                if (b != 0) atomicAdd(&lh[b], 1u);
            }
        }
        *(u16x8*)(out + off) = o;
    }
    if (WITH_HIST) {
        zeros = wave_sum_u32(zeros);
        if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&lh[0], zeros);
        __syncthreads();
        for (int i = threadIdx.x; i < HIST_BINS; i += 1024) {
            const uint32_t c = lh[i];
            if (c) atomicAdd(&hist[i], c);
        }
    }
}

extern "C" int slim_residual_weight(const void* W, const void* LR, const uint8_t* mask, int64_t numel, void* out,
                                    uint32_t* hist, void* stream) {
    SLIM_REQUIRE(W && LR && mask && out && numel > 0, "slim_residual_weight: bad arguments");
    SLIM_REQUIRE(numel % 8 == 0, "slim_residual_weight: numel must be a multiple of 8");
    hipStream_t st = (hipStream_t)stream;
    SLIM_PROF("residual_weight", stream, 7.0 * numel);
    int64_t blocks = ceil_div64(numel >> 3, 1024 * 2);
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    if (hist) {
        SLIM_HIP(slim_allow_dynamic_lds((const void*)residual_weight_kernel<true>, HIST_BINS * 4));
        SLIM_HIP(hipMemsetAsync(hist, 0, HIST_BINS * sizeof(uint32_t), st));
        residual_weight_kernel<true><<<(unsigned)blocks, 1024, HIST_BINS * 4, st>>>(
            (const bf16_t*)W, (const bf16_t*)LR, mask, numel, (bf16_t*)out, hist);
    } else {
        blocks = ceil_div64(numel >> 3, 1024);
        blocks = blocks > 2048 ? 2048 : blocks;
        residual_weight_kernel<false><<<(unsigned)blocks, 1024, 0, st>>>((const bf16_t*)W, (const bf16_t*)LR, mask, numel,
                                                                        (bf16_t*)out, nullptr);
    }
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}
