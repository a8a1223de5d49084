// Issue rate of v_smfmac_f32_16x16x64_bf16 / v_smfmac_f32_32x32x32_bf16 against v_mfma_f32_16x16x32_bf16 on gfx950:
// one wave per SIMD, long chains of independent instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8};
    bf16x16 b = {1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 6, 7, 8};
    bf16x8 b8 = {1, 2, 3, 4, 5, 6, 7, 8};
    f32x4 c[8];
    f32x16 d[4];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) d[i][j] = 0;
    int idx = threadIdx.x * 0x1111;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b8, c[i], 0, 0, 0);
            if (KIND == 1) c[i] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c[i], idx, 0, 0);
            if (KIND == 2) d[i & 3] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, d[i & 3], idx, 0, 0);
            if (KIND == 3) d[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b8, d[i & 3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0];
    for (int i = 0; i < 4; ++i) s += d[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, double flops_per_instr) {
    float* out;
    hipMalloc(&out, 256 * 256 * 4 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256 * 2, 256>>>(out, 100);
    hipEventRecord(e0);
    k<KIND><<<256 * 2, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = 256.0 * 2 * 4 * iters * 8;   // waves * instructions
    printf("%-32s %8.3f ms  %.1f instr/us/chip  %.0f TFLOP/s (dense-equivalent)  ~%.1f clk per instr per SIMD at 2.4 GHz\n", name, ms,
           instr / ms / 1e3, instr * flops_per_instr / ms / 1e9, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
    run<1>("smfmac_f32_16x16x64_bf16", 2.0 * 16 * 16 * 64);
    run<2>("smfmac_f32_32x32x32_bf16", 2.0 * 32 * 32 * 32);
    run<3>("mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
    return 0;
}
