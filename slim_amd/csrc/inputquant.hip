// Activation fake-quantisation applied by the input hooks of the emulated compressed forward
// (reference slim/quantization/quantization.py:219-263 and :280-292).  Both variants reproduce
// the reference's bf16 arithmetic step by step (every intermediate is rounded to bf16).
#include "common.hpp"
#include "../../include/slim_hip.h"

// ---- abs-max of a bf16 tensor into a device word (|bf16| patterns order like integers) ----------
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const bf16_t* __restrict__ x, int64_t numel,
                                                          uint32_t* __restrict__ out) {
    uint32_t m = 0;
    const int64_t nvec = numel >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const u16x8 a = *(const u16x8*)(x + (v << 3));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t b = a[j] & 0x7fffu;
            m = b > m ? b : m;
        }
    }
    if (blockIdx.x == 0) {
        const int64_t t = (nvec << 3) + threadIdx.x;
        if (t < numel) {
            const uint32_t b = x[t] & 0x7fffu;
            m = b > m ? b : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// round a non-negative fp32 value to a float format with `mant` mantissa bits, minimum normal
// exponent 2^emin, round-to-nearest-even, subnormals kept; values above `fmax` are not produced by
// the callers below.
__device__ __forceinline__ float round_to_minifloat(float a, int mant, int emin) {
    const float min_normal = __builtin_ldexpf(1.f, emin);
    if (a < min_normal) {
        const float quantum = __builtin_ldexpf(1.f, emin - mant);
        return rintf(a / quantum) * quantum;
    }
    uint32_t u = __float_as_uint(a);
    const int drop = 23 - mant;
    u += ((1u << (drop - 1)) - 1u) + ((u >> drop) & 1u);
    u &= ~((1u << drop) - 1u);
    return __uint_as_float(u);
}

// 8-bit branch (quantization.py:286-290): m = max|x| (bf16); e4m3 if m < 488 else e5m2;
//   q = fp8(bf16(bf16(x / m) * (fmax - 1)));  out = bf16(bf16(q / (fmax - 1)) * m)
__global__ __launch_bounds__(256) void input_fp8_kernel(const bf16_t* __restrict__ x, int64_t numel,
                                                        const uint32_t* __restrict__ absmax, bf16_t* __restrict__ out) {
    const float m = bf16_to_f32((bf16_t)absmax[0]);
    const bool e4m3 = m < 488.f;
    const float scale = e4m3 ? 447.f : 57343.f;
    const int mant = e4m3 ? 3 : 2, emin = e4m3 ? -6 : -14;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float xv = bf16_to_f32(x[i]);
        const float t1 = round_bf16(xv / m);
        const float t2 = round_bf16(t1 * scale);
        float q = round_to_minifloat(fabsf(t2), mant, emin);
        q = t2 < 0.f ? -q : q;
        if (t2 != t2) q = t2;
        const float d1 = round_bf16(q / scale);
        out[i] = f32_to_bf16(d1 * m);
    }
}

extern "C" int slim_input_fake_quant_fp8(const void* x, int64_t numel, void* out, uint32_t* absmax_scratch,
                                         void* stream) {
    SLIM_REQUIRE(x && out && absmax_scratch && numel > 0, "slim_input_fake_quant_fp8: bad arguments");
    SLIM_REQUIRE(((uintptr_t)x % 16) == 0, "slim_input_fake_quant_fp8: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SLIM_HIP(hipMemsetAsync(absmax_scratch, 0, sizeof(uint32_t), st));
    int64_t blocks = ceil_div64(numel >> 3, 256);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    absmax_bf16_kernel<<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)x, numel, absmax_scratch);
    SLIM_LAUNCH_CHECK();
    blocks = ceil_div64(numel, 256);
    blocks = blocks > 8192 ? 8192 : blocks;
    input_fp8_kernel<<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)x, numel, absmax_scratch, (bf16_t*)out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Integer group branch (Quantizer.quantize_input / dequantize_input, symmetric): one wave per group.
//   range = max|x| over the group (bf16); range < eps * maxq -> 1
//   scale = bf16(bf16(1 / range) * maxq); q = clamp(round(bf16(x * scale)), -maxq-1, maxq); out = bf16(q / scale)
__global__ __launch_bounds__(256) void input_int_kernel(const bf16_t* __restrict__ x, int64_t n_groups, int group,
                                                        float maxq, float thr, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t gidx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gidx >= n_groups) return;
    const bf16_t* p = x + gidx * group;
    uint32_t m = 0;
    for (int i = lane; i < group; i += 64) {
        const uint32_t b = p[i] & 0x7fffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    float range = bf16_to_f32((bf16_t)m);
    if (range < thr) range = 1.f;
    const float scale = round_bf16(round_bf16(1.0f / range) * maxq);   // `max_q / range` = reciprocal(range) * max_q in bf16
    bf16_t* o = out + gidx * group;
    for (int i = lane; i < group; i += 64) {
        const float prod = round_bf16(bf16_to_f32(p[i]) * scale);
        const float q = fminf(fmaxf(rintf(prod), -maxq - 1.f), maxq);
        o[i] = f32_to_bf16(q / scale);
    }
}

extern "C" int slim_input_fake_quant_int(const void* x, int64_t n_groups, int group, int num_bits, float eps,
                                         void* out, void* stream) {
    SLIM_REQUIRE(x && out && n_groups > 0 && group > 0, "slim_input_fake_quant_int: bad arguments");
    SLIM_REQUIRE(num_bits >= 2 && num_bits <= 16, "slim_input_fake_quant_int: num_bits out of range");
    const float maxq = (float)((1 << (num_bits - 1)) - 1);
    const float thr = (float)((double)eps * (double)maxq);
    input_int_kernel<<<(unsigned)ceil_div64(n_groups, 4), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, n_groups, group,
                                                                                     maxq, thr, (bf16_t*)out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}
