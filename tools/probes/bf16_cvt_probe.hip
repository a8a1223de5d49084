// Exhaustive comparison of the library's integer fp32 -> bf16 rounding (common.hpp) with the compiler's conversion
// ((__bf16)x: v_cvt_pk_bf16_f32 on gfx950) over all 2^32 bit patterns.
//   hipcc -O3 --offload-arch=gfx950 -I slim_amd/csrc tools/probes/bf16_cvt_probe.hip -o /tmp/bf16_cvt_probe && /tmp/bf16_cvt_probe
#include "common.hpp"
#include <cstdio>
__global__ void probe(unsigned long long* counts) {
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long bad_finite = 0, bad_nan = 0, bad_denorm = 0;
    for (uint32_t hi = 0; hi < 4096; ++hi) {
        const uint32_t u = (hi << 20) | base;               // base < 2^20
        const float f = __uint_as_float(u);
        const bf16_t a = f32_to_bf16(f);
        const __bf16 h = (__bf16)f;
        const bf16_t b = __builtin_bit_cast(unsigned short, h);
        if (a != b) {
            const uint32_t e = (u >> 23) & 0xff;
            if (e == 0xff && (u & 0x7fffff)) ++bad_nan;
            else if (e == 0) ++bad_denorm;
            else ++bad_finite;
        }
    }
    atomicAdd(&counts[0], bad_finite);
    atomicAdd(&counts[1], bad_nan);
    atomicAdd(&counts[2], bad_denorm);
}
int main() {
    unsigned long long* c;
    (void)hipMalloc(&c, 3 * sizeof(unsigned long long));
    (void)hipMemset(c, 0, 3 * sizeof(unsigned long long));
    probe<<<(1 << 20) / 256, 256>>>(c);
    unsigned long long h[3];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("mismatches: normal/inf %llu, NaN %llu, fp32 denormal or zero %llu\n", h[0], h[1], h[2]);
    return 0;
}
