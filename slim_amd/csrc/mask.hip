// N:M mask selection (reference slim/utils.py:75-88, saliency slim/prune.py:264-265).
//
// The reference marks, in every aligned group of M columns, the N entries that
// topk(largest=False) returns.  On CPU that routine is std::nth_element (libstdc++
// introselect) over (value, index) pairs with a comparator that looks at values only and
// orders NaN after everything else, so the outcome on ties is a fixed function of the
// pairwise order of the M values.  For M = 4 that function is tabulated once on the host by
// running the selection below over every consistent combination of the six pairwise
// comparisons (3^6 signatures, 75 of them reachable); the kernel computes a group's
// signature with integer compares and looks the 4-bit mask up in LDS.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include <mutex>

namespace {

struct Item {
    int key;  // rank of the value (smaller = smaller value); equal keys = tie
    int idx;
};

inline bool item_less(const Item& a, const Item& b) { return a.key < b.key; }
inline void item_swap(Item& a, Item& b) { Item t = a; a = b; b = t; }

// libstdc++ std::nth_element on a small array: introselect with median-of-three pivoting
// while more than three elements remain, then insertion sort.
void nth_select(Item* a, int n_items, int nth) {
    int first = 0, last = n_items;
    int depth = 0;
    for (int s = n_items; s > 1; s >>= 1) ++depth;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            // heap-select fallback: unreachable for n_items <= 8 (depth budget is never exhausted
            // before the range shrinks to three), kept as a plain selection for completeness
            for (int i = first; i <= nth; ++i)
                for (int j = i + 1; j < last; ++j)
                    if (item_less(a[j], a[i])) item_swap(a[i], a[j]);
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        // move median of (first+1, mid, last-1) to first
        {
            Item &r = a[first], &x = a[first + 1], &y = a[mid], &z = a[last - 1];
            if (item_less(x, y)) {
                if (item_less(y, z)) item_swap(r, y);
                else if (item_less(x, z)) item_swap(r, z);
                else item_swap(r, x);
            } else if (item_less(x, z)) item_swap(r, x);
            else if (item_less(y, z)) item_swap(r, z);
            else item_swap(r, y);
        }
        // unguarded partition of [first+1, last) around the pivot now at first
        int lo = first + 1, hi = last;
        for (;;) {
            while (item_less(a[lo], a[first])) ++lo;
            --hi;
            while (item_less(a[first], a[hi])) --hi;
            if (!(lo < hi)) break;
            item_swap(a[lo], a[hi]);
            ++lo;
        }
        const int cut = lo;
        if (cut <= nth) first = cut;
        else last = cut;
    }
    // insertion sort of [first, last)
    for (int i = first + 1; i < last; ++i) {
        Item v = a[i];
        if (item_less(v, a[first])) {
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = v;
        } else {
            int j = i;
            while (item_less(v, a[j - 1])) {
                a[j] = a[j - 1];
                --j;
            }
            a[j] = v;
        }
    }
}

// signature of 4 keys: for each pair (i<j) in order (01,02,03,12,13,23): 0 if k_i<k_j, 1 if ==, 2 if >
inline int signature4(const int* k) {
    static const int P[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    int sig = 0, mul = 1;
    for (int p = 0; p < 6; ++p) {
        const int a = k[P[p][0]], b = k[P[p][1]];
        sig += mul * (a < b ? 0 : (a == b ? 1 : 2));
        mul *= 3;
    }
    return sig;
}

uint8_t g_lut[3][729];
std::once_flag g_lut_once;

void build_luts() {
    for (int n = 1; n <= 3; ++n)
        for (int s = 0; s < 729; ++s) g_lut[n - 1][s] = 0;
    // every weak ordering of four values appears among the tuples over {0,1,2,3}
    for (int t = 0; t < 256; ++t) {
        int k[4] = {t & 3, (t >> 2) & 3, (t >> 4) & 3, (t >> 6) & 3};
        const int sig = signature4(k);
        for (int n = 1; n <= 3; ++n) {
            Item it[4];
            for (int i = 0; i < 4; ++i) it[i] = Item{k[i], i};
            nth_select(it, 4, n - 1);
            uint8_t m = 0;
            for (int i = 0; i < n; ++i) m |= (uint8_t)(1u << it[i].idx);
            g_lut[n - 1][sig] = m;
        }
    }
}

}  // namespace

extern "C" int slim_nm_lut4(int n, uint8_t* out729) {
    SLIM_REQUIRE(n >= 1 && n <= 3 && out729, "slim_nm_lut4: n must be 1..3");
    std::call_once(g_lut_once, build_luts);
    for (int s = 0; s < 729; ++s) out729[s] = g_lut[n - 1][s];
    return SLIM_OK;
}

// Monotone integer key of an fp32 value under the reference comparator: ascending by value,
// -0 == +0, every NaN equal to every other NaN and greater than +inf.
__device__ __forceinline__ int32_t order_key(float v) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fffffff;  // NaN
    if (u == 0x80000000u) u = 0;                             // -0
    return (u & 0x80000000u) ? -(int32_t)(u & 0x7fffffffu) : (int32_t)u;
}

__device__ __forceinline__ int sig4(int32_t a, int32_t b, int32_t c, int32_t d) {
    auto cmp = [](int32_t x, int32_t y) { return x < y ? 0 : (x == y ? 1 : 2); };
    return cmp(a, b) + 3 * cmp(a, c) + 9 * cmp(a, d) + 27 * cmp(b, c) + 81 * cmp(b, d) + 243 * cmp(c, d);
}

struct LutArg {
    uint8_t v[736];
};

// One thread = 8 consecutive columns (two groups of four) of one row.
// WITH_SCALE: metric = |W| * s[col] (fp32 product, as the reference forms it); otherwise the
// metric is the fp32 input itself (generic prune_nm).
template <bool FROM_BF16>
__global__ __launch_bounds__(256) void mask_nm4_kernel(const void* __restrict__ src, const float* __restrict__ s,
                                                       int64_t N, int64_t K, uint8_t* __restrict__ mask,
                                                       LutArg lut_arg) {
    __shared__ uint8_t lut[736];
    for (int i = threadIdx.x; i < 736; i += 256) lut[i] = lut_arg.v[i];
    __syncthreads();
    const int64_t chunks_per_row = K >> 3;
    const int64_t total = N * chunks_per_row;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / chunks_per_row;
        const int64_t c0 = (t - row * chunks_per_row) << 3;
        float m[8];
        if (FROM_BF16) {
            const u16x8 w = *(const u16x8*)((const bf16_t*)src + row * K + c0);
            if (s) {
                const f32x4 s0 = *(const f32x4*)(s + c0);
                const f32x4 s1 = *(const f32x4*)(s + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    m[j] = bf16_to_f32(w[j] & 0x7fff) * s0[j];
                    m[4 + j] = bf16_to_f32(w[4 + j] & 0x7fff) * s1[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = bf16_to_f32(w[j] & 0x7fff);
            }
        } else {
            const f32x4 a = *(const f32x4*)((const float*)src + row * K + c0);
            const f32x4 b = *(const f32x4*)((const float*)src + row * K + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[j] = a[j];
                m[4 + j] = b[j];
            }
        }
        int32_t k[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = order_key(m[j]);
        const uint32_t b0 = lut[sig4(k[0], k[1], k[2], k[3])];
        const uint32_t b1 = lut[sig4(k[4], k[5], k[6], k[7])];
        uint2 o;
        o.x = (b0 & 1u) | ((b0 & 2u) << 7) | ((b0 & 4u) << 14) | ((b0 & 8u) << 21);
        o.y = (b1 & 1u) | ((b1 & 2u) << 7) | ((b1 & 4u) << 14) | ((b1 & 8u) << 21);
        *(uint2*)(mask + row * K + c0) = o;
    }
}

static int launch_mask(const void* src, bool from_bf16, const float* s, int64_t N, int64_t K, int n, int m,
                       uint8_t* mask, void* stream) {
    SLIM_REQUIRE(src && mask && N > 0 && K > 0, "slim_mask_nm: bad arguments");
    if (m != 4 || n < 1 || n > 3) {
        slim_set_error("slim_mask_nm: only M = 4 with N in 1..3 is implemented (got %d:%d)", n, m);
        return SLIM_ERR_UNSUPPORTED;
    }
    SLIM_REQUIRE(K % 8 == 0, "slim_mask_nm: K must be a multiple of 8");
    SLIM_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)mask % 8) == 0 && (!s || ((uintptr_t)s % 16) == 0),
                 "slim_mask_nm: pointers must be 16-byte aligned");
    SLIM_PROF("mask_nm", stream, (from_bf16 ? 3.0 : 5.0) * N * K + (s ? 4.0 * K : 0.0));
    std::call_once(g_lut_once, build_luts);
    LutArg la;
    for (int i = 0; i < 736; ++i) la.v[i] = i < 729 ? g_lut[n - 1][i] : 0;
    const int64_t total = N * (K / 8);
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 8192) blocks = 8192;
    if (from_bf16)
        mask_nm4_kernel<true><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(src, s, N, K, mask, la);
    else
        mask_nm4_kernel<false><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(src, nullptr, N, K, mask, la);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_mask_nm(const void* W_bf16, const float* s, int64_t N, int64_t K, int n, int m, uint8_t* mask,
                            void* stream) {
    return launch_mask(W_bf16, true, s, N, K, n, m, mask, stream);
}

extern "C" int slim_prune_nm_f32(const float* metric, int64_t N, int64_t K, int n, int m, uint8_t* mask,
                                 void* stream) {
    return launch_mask(metric, false, nullptr, N, K, n, m, mask, stream);
}

// ---- unstructured (per-row) pruning -------------------------------------------------------------
// slim/prune.py:275-279: per output row, a stable ascending sort of the saliency and the first
// int(K * sparsity_ratio) indices pruned, i.e. the k smallest with ties going to the lower index; NaN
// sorts last and -0 equals +0, as in torch.sort.  One workgroup per row: the row's keys sit in LDS, a
// 4-pass radix select finds the k-th smallest key, and an ordered count settles the ties at that key.
__device__ __forceinline__ uint32_t sort_key(float v) {
    if (v != v) return 0xffffffffu;
    uint32_t b = __float_as_uint(v);
    if (b == 0x80000000u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <bool FROM_BF16>
__global__ __launch_bounds__(256) void mask_unstructured_kernel(const void* __restrict__ src, const float* __restrict__ s,
                                                                int64_t K, int64_t k_prune, uint8_t* __restrict__ mask) {
    extern __shared__ uint32_t lds_u32[];
    uint32_t* keys = lds_u32;
    uint32_t* hist = lds_u32 + K;      // 256 bins
    uint32_t* scan = hist + 256;       // 256 per-thread counts
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    for (int64_t i = tid; i < K; i += 256) {
        float v;
        if (FROM_BF16) {
            const bf16_t w = ((const bf16_t*)src)[row * K + i];
            v = bf16_to_f32((bf16_t)(w & 0x7fffu)) * s[i];
        } else {
            v = ((const float*)src)[row * K + i];
        }
        keys[i] = sort_key(v);
    }
    uint32_t prefix = 0u, pmask = 0u;
    uint32_t remaining = (uint32_t)k_prune;   // how many of the keys matching the prefix are still to be taken
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0u;
        __syncthreads();
        for (int64_t i = tid; i < K; i += 256) {
            const uint32_t key = keys[i];
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        uint32_t cum = 0u, bin = 255u;
        for (uint32_t b = 0; b < 256u; ++b) {    // every thread scans the same 256 counters (LDS broadcast)
            const uint32_t h = hist[b];
            if (cum + h >= remaining) {
                bin = b;
                break;
            }
            cum += h;
        }
        remaining -= cum;
        prefix |= bin << (8 * pass);
        pmask |= 0xffu << (8 * pass);
        __syncthreads();
    }
    // keys < prefix are pruned; of the keys == prefix the first `remaining` in index order
    const int64_t chunk = (K + 255) / 256;
    const int64_t i0 = tid * chunk, i1 = (i0 + chunk < K) ? i0 + chunk : K;
    uint32_t eq = 0u;
    for (int64_t i = i0; i < i1; ++i) eq += keys[i] == prefix;
    scan[tid] = eq;
    __syncthreads();
    uint32_t before = 0u;
    for (int t = 0; t < tid; ++t) before += scan[t];
    for (int64_t i = i0; i < i1; ++i) {
        const uint32_t key = keys[i];
        uint32_t m = key < prefix;
        if (key == prefix) m = before++ < remaining;
        keys[i] = m;
    }
    __syncthreads();
    for (int64_t i = tid; i < K; i += 256) mask[row * K + i] = (uint8_t)keys[i];
}

static int launch_unstructured(const void* src, bool from_bf16, const float* s, int64_t N, int64_t K, int64_t k_prune,
                               uint8_t* mask, void* stream) {
    SLIM_REQUIRE(src && mask && N > 0 && K > 0 && (!from_bf16 || s), "slim_mask_unstructured: bad arguments");
    SLIM_REQUIRE(k_prune >= 0 && k_prune <= K, "slim_mask_unstructured: the number of pruned weights per row must be in [0, K]");
    const int64_t lds = (K + 512) * 4;
    if (lds > 160 * 1024) {
        slim_set_error("slim_mask_unstructured: rows longer than %d elements are not supported", (160 * 1024 - 2048) / 4);
        return SLIM_ERR_UNSUPPORTED;
    }
    SLIM_REQUIRE(N < ((int64_t)1 << 31), "slim_mask_unstructured: too many rows");
    hipStream_t st = (hipStream_t)stream;
    SLIM_PROF("mask_unstructured", stream, (double)N * K * (from_bf16 ? 3.0 : 5.0));
    if (k_prune == 0) {
        SLIM_HIP(hipMemsetAsync(mask, 0, (size_t)(N * K), st));
        return SLIM_OK;
    }
    if (from_bf16) {
        SLIM_HIP(slim_allow_dynamic_lds((const void*)mask_unstructured_kernel<true>, (int)lds));
        mask_unstructured_kernel<true><<<(unsigned)N, 256, (size_t)lds, st>>>(src, s, K, k_prune, mask);
    } else {
        SLIM_HIP(slim_allow_dynamic_lds((const void*)mask_unstructured_kernel<false>, (int)lds));
        mask_unstructured_kernel<false><<<(unsigned)N, 256, (size_t)lds, st>>>(src, s, K, k_prune, mask);
    }
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_mask_unstructured(const void* W_bf16, const float* s, int64_t N, int64_t K, int64_t k_prune,
                                      uint8_t* mask, void* stream) {
    return launch_unstructured(W_bf16, true, s, N, K, k_prune, mask, stream);
}

extern "C" int slim_prune_unstructured_f32(const float* metric, int64_t N, int64_t K, int64_t k_prune, uint8_t* mask,
                                           void* stream) {
    return launch_unstructured(metric, false, nullptr, N, K, k_prune, mask, stream);
}
