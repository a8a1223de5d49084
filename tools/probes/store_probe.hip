// Micro-probe: how long does a workgroup take to write a 128 x 128 bf16 tile (32 KiB) as sixteen 8-byte stores per
// thread, as a function of the tile's row stride in memory?  Timestamps (100 MHz) around the store loop of every workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned short* out, long ld, int tiles_n, long* stamps, int wait) {
    const int t = threadIdx.x;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    unsigned short* base = out + (long)tm * 128 * ld + tn * 128;
    const int c = (t % 32) * 4, rt = t / 32;
    long t0 = wall_clock64();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        ushort4 v = make_ushort4(t, it, c, 1);
        *(ushort4*)(base + (long)(rt + it * 8) * ld + c) = v;
    }
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long t1 = wall_clock64();
    if (t == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = t1;
    }
}

int main() {
    const int M = 4096;
    long* stamps;
    hipMalloc(&stamps, 4096 * 16);
    for (long N : {128L, 640L, 4096L, 11008L}) {
        unsigned short* out;
        hipMalloc(&out, (size_t)M * N * 2);
        const int tiles_n = N / 128, tiles = (M / 128) * tiles_n;
        const int grid = std::min(tiles, 160);
        for (int wait = 0; wait < 2; ++wait) {
            for (int rep = 0; rep < 3; ++rep) {
                probe<<<grid, 256>>>(out, N, tiles_n, stamps, wait);
                hipDeviceSynchronize();
            }
            std::vector<long> h(2 * grid);
            hipMemcpy(h.data(), stamps, sizeof(long) * 2 * grid, hipMemcpyDeviceToHost);
            double mean = 0, mx = 0;
            for (int i = 0; i < grid; ++i) {
                double d = (h[2 * i + 1] - h[2 * i]) * 0.01;
                mean += d / grid;
                mx = std::max(mx, d);
            }
            printf("row stride %6ld elements, %3d workgroups, wait=%d: store loop mean %.2f us, max %.2f us\n", N, grid, wait, mean, mx);
        }
        hipFree(out);
    }
    return 0;
}
