// Issue rate of the sparse and dense bf16 matrix instructions: one or two waves per SIMD, 16 independent accumulators each.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/smfmac_rate.hip -o /tmp/smfmac_rate && /tmp/smfmac_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int SPARSE>
__global__ void rate_kernel(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a;
    bf16x16 b;
    for (int q = 0; q < 8; ++q) a[q] = (__bf16)(float)(threadIdx.x + q);
    for (int q = 0; q < 16; ++q) b[q] = (__bf16)(float)(threadIdx.x * 3 + q);
    bf16x8 b8 = __builtin_shufflevector(b, b, 0, 1, 2, 3, 4, 5, 6, 7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (SPARSE) acc[i] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, acc[i], (int)threadIdx.x, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b8, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    const int iters = 20000;
    for (int sparse = 0; sparse < 2; ++sparse)
        for (int threads : {256, 512}) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (sparse) rate_kernel<1><<<256, threads>>>(out, iters);
                else rate_kernel<0><<<256, threads>>>(out, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)iters * 16;                      // instructions per wave
            const double per_simd = n * (threads / 256);             // per SIMD
            const double flops = 256.0 * 4 * per_simd * (sparse ? 32768.0 : 16384.0);
            printf("%s waves/SIMD %d: %.3f ms, %.1f ns per instruction per SIMD, %.0f TFLOP/s (dense-equivalent)\n", sparse ? "smfmac 16x16x64" : "mfma 16x16x32",
                   threads / 256, ms, ms * 1e6 / per_simd, flops / ms / 1e9);
        }
    return 0;
}
