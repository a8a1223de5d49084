// Shared device/host helpers for the slim_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SLIM_OK 0
#define SLIM_ERR_INVALID (-1)
#define SLIM_ERR_LAUNCH (-2)
#define SLIM_ERR_UNSUPPORTED (-3)
#define SLIM_ERR_SOLVER (-4)

void slim_set_error(const char* fmt, ...);
// Raise a kernel's dynamic-LDS limit above 64 KiB once per (kernel, device).
hipError_t slim_allow_dynamic_lds(const void* func, int bytes);

// Optional per-launch timing (slim_profile_enable): an RAII scope records a HIP event pair on the
// launch stream around the kernels it encloses.
// launches: how many kernel launches the scope encloses (a chain of identical launches is reported per launch)
int slim_prof_begin(const char* name, hipStream_t st, double bytes = 0.0, double flops = 0.0, int launches = 1);
void slim_prof_end(int token, hipStream_t st);
extern bool g_slim_prof_on;
struct SlimProfScope {
    int token;
    hipStream_t st;
    SlimProfScope(const char* name, hipStream_t s, double bytes = 0.0, double flops = 0.0, int launches = 1)
        : token(g_slim_prof_on ? slim_prof_begin(name, s, bytes, flops, launches) : -1), st(s) {}
    ~SlimProfScope() {
        if (token >= 0) slim_prof_end(token, st);
    }
};
// bytes / flops: the ALGORITHMIC traffic or arithmetic of the enclosed launches (inputs read once,
// outputs written once), accumulated per scope name for roofline reporting
#define SLIM_PROF(name, stream, ...) SlimProfScope prof_scope__(name, (hipStream_t)(stream), ##__VA_ARGS__)

#define SLIM_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            slim_set_error(__VA_ARGS__);     \
            return SLIM_ERR_INVALID;         \
        }                                    \
    } while (0)

#define SLIM_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            slim_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return SLIM_ERR_LAUNCH;                                                      \
        }                                                                                \
    } while (0)

#define SLIM_LAUNCH_CHECK() SLIM_HIP(hipGetLastError())

typedef unsigned short bf16_t;  // raw bf16 bit pattern

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}

// fp32 -> bf16, round to nearest even, NaN stays NaN (quiet).  On the device this is the hardware conversion
// (v_cvt_pk_bf16_f32), which returns the same 16 bits as the integer arithmetic below for every one of the 2^32 inputs
// (tools/probes/bf16_cvt_probe.hip compares them exhaustively) at a sixth of the instructions.
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
    union { uint32_t u; float f; } c;
    c.f = f;
    // branch-free (a select): inside store loops a branch per element costs far more than the arithmetic
    const uint32_t nan = (c.u >> 16) | 0x0040u;
    const uint32_t r = (c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16;
    return (bf16_t)(((c.u & 0x7fffffffu) > 0x7f800000u) ? nan : r);
}

__host__ __device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ uint32_t f32_bits(float f) { return __float_as_uint(f); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
