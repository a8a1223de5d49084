// Group quantisation of the low-rank adapters (reference Triton kernels
// slim/quantization/utils.py:66-120, 131-185, 216-251 and their launchers :123-128, :188-193,
// :254-271, called with one scale per (block rows x 1 column), quantization.py:146-148).
//
// The reference launches one program per (block x 1) column strip, which walks memory with a
// row stride.  Here a workgroup owns `block` rows x 128 columns: 16 column-threads of 8
// adjacent columns each (256-byte row segments) times 16 row-threads, so every access is a
// 16-byte vector on a contiguous segment and the per-column maximum is folded through LDS.
#include "common.hpp"
#include "../../include/slim_hip.h"

__device__ __forceinline__ float f16_round(float v) { return (float)(_Float16)v; }

// code = round-half-toward-zero(clamp(x / fp16(alpha) * maxq, -(maxq+1), maxq)); NaN (0/0 in an
// all-zero group) gives code 0.
__device__ __forceinline__ int gq_code(float x, float a16, float maxq) {
    float v = x / a16 * maxq;
    if (v != v) return 0;
    v = fminf(fmaxf(v, -(maxq + 1.f)), maxq);
    const float vi = truncf(v);
    const float fr = v - vi;
    return (int)vi + (fr > 0.5f ? 1 : 0) - (fr < -0.5f ? 1 : 0);
}
__device__ __forceinline__ bf16_t gq_deq(int code, float alpha, float maxq) {
    return f32_to_bf16((float)(_Float16)(float)code / maxq * alpha);
}

// mode bits: 1 = write alphas, 2 = write codes, 4 = write dequantised bf16
// alphas_in != NULL: use the given scales instead of computing them.
__global__ __launch_bounds__(256) void group_quant_kernel(const bf16_t* __restrict__ x, const int8_t* __restrict__ codes_in,
                                                          const bf16_t* __restrict__ alphas_in, int64_t M, int64_t N,
                                                          int block, float maxq, bf16_t* __restrict__ alphas_out,
                                                          int8_t* __restrict__ codes_out, bf16_t* __restrict__ deq_out) {
    __shared__ uint32_t red[16][16][4];  // [row-thread][col-thread][4 x packed pairs of bf16 maxima]
    const int ct = threadIdx.x & 15, rt = threadIdx.x >> 4;
    const int64_t c0 = ((int64_t)blockIdx.x * 16 + ct) * 8;
    const int64_t g = blockIdx.y;
    const int64_t r0 = g * block;
    const bool active = c0 < N;
    u16x8 amax;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax[j] = 0;
    if (!alphas_in) {
        if (active)
            for (int r = rt; r < block; r += 16) {
                const u16x8 v = *(const u16x8*)(x + (r0 + r) * N + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bf16_t a = v[j] & 0x7fff;  // |bf16| patterns order like unsigned integers
                    amax[j] = a > amax[j] ? a : amax[j];
                }
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rt][ct][j] = (uint32_t)amax[2 * j] | ((uint32_t)amax[2 * j + 1] << 16);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) amax[j] = 0;
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t p = red[t][ct][j];
                const bf16_t lo = (bf16_t)(p & 0xffff), hi = (bf16_t)(p >> 16);
                amax[2 * j] = lo > amax[2 * j] ? lo : amax[2 * j];
                amax[2 * j + 1] = hi > amax[2 * j + 1] ? hi : amax[2 * j + 1];
            }
        // NaN inputs: |NaN| patterns exceed every finite pattern, so a NaN wins the maximum like torch.amax
    } else if (active) {
        amax = *(const u16x8*)(alphas_in + g * N + c0);
    }
    if (!active) return;
    if (alphas_out && rt == 0) *(u16x8*)(alphas_out + g * N + c0) = amax;
    if (!codes_out && !deq_out) return;
    float a32[8], a16[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a32[j] = bf16_to_f32(amax[j]);
        a16[j] = f16_round(a32[j]);
    }
    for (int r = rt; r < block; r += 16) {
        const int64_t off = (r0 + r) * N + c0;
        int c[8];
        if (codes_in) {
            const uint2 p = *(const uint2*)(codes_in + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = (int)(int8_t)((p.x >> (8 * j)) & 0xff);
                c[4 + j] = (int)(int8_t)((p.y >> (8 * j)) & 0xff);
            }
        } else {
            const u16x8 v = *(const u16x8*)(x + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = gq_code(bf16_to_f32(v[j]), a16[j], maxq);
        }
        if (codes_out) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo |= ((uint32_t)(uint8_t)(int8_t)c[j]) << (8 * j);
                hi |= ((uint32_t)(uint8_t)(int8_t)c[4 + j]) << (8 * j);
            }
            *(uint2*)(codes_out + off) = uint2{lo, hi};
        }
        if (deq_out) {
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gq_deq(c[j], a32[j], maxq);
            *(u16x8*)(deq_out + off) = o;
        }
    }
}

static int gq_launch(const char* who, const void* x, const int8_t* codes_in, const void* alphas_in, int64_t M, int64_t N,
                     int block, int bits, void* alphas_out, int8_t* codes_out, void* deq_out, void* stream) {
    SLIM_REQUIRE(M > 0 && N > 0 && block > 0, "%s: bad sizes", who);
    // same divisibility contract as the reference (quantization.py:140) on the grouped dimension
    SLIM_REQUIRE(M % block == 0, "%s: rows (%lld) not divisible by block size (%d)", who, (long long)M, block);
    SLIM_REQUIRE(N % 8 == 0, "%s: columns must be a multiple of 8", who);
    SLIM_REQUIRE(bits >= 2 && bits <= 8, "%s: bits must be in [2, 8]", who);
    SLIM_REQUIRE(M / block <= 65535, "%s: too many row groups", who);
    const float maxq = (float)((1 << (bits - 1)) - 1);
    SLIM_PROF("group_quant", stream, ((x ? 2.0 : 1.0) + (codes_out ? 1.0 : 0.0) + (deq_out ? 2.0 : 0.0)) * M * N);
    dim3 grid((unsigned)ceil_div64(N, 128), (unsigned)(M / block));
    group_quant_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, codes_in, (const bf16_t*)alphas_in, M, N,
                                                             block, maxq, (bf16_t*)alphas_out, codes_out,
                                                             (bf16_t*)deq_out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_group_absmax(const void* x, int64_t M, int64_t N, int block, void* alphas, void* stream) {
    SLIM_REQUIRE(x && alphas, "slim_group_absmax: null pointer");
    return gq_launch("slim_group_absmax", x, nullptr, nullptr, M, N, block, 8, alphas, nullptr, nullptr, stream);
}

extern "C" int slim_group_quantize(const void* x, const void* alphas, int64_t M, int64_t N, int block, int bits,
                                   int8_t* codes, void* stream) {
    SLIM_REQUIRE(x && alphas && codes, "slim_group_quantize: null pointer");
    return gq_launch("slim_group_quantize", x, nullptr, alphas, M, N, block, bits, nullptr, codes, nullptr, stream);
}

extern "C" int slim_group_dequantize(const int8_t* codes, const void* alphas, int64_t M, int64_t N, int block, int bits,
                                     void* out, void* stream) {
    SLIM_REQUIRE(codes && alphas && out, "slim_group_dequantize: null pointer");
    return gq_launch("slim_group_dequantize", nullptr, codes, alphas, M, N, block, bits, nullptr, nullptr, out, stream);
}

extern "C" int slim_group_fake_quant(const void* x, int64_t M, int64_t N, int block, int bits, void* out, void* alphas,
                                     int8_t* codes, void* stream) {
    SLIM_REQUIRE(x && out, "slim_group_fake_quant: null pointer");
    return gq_launch("slim_group_fake_quant", x, nullptr, nullptr, M, N, block, bits, alphas, codes, out, stream);
}
