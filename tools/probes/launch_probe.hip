// Micro-probe: how many short kernel launches per second does one process sustain, from 1..8 host threads each
// feeding its own stream?   hipcc --offload-arch=gfx950 -O3 -o launch_probe launch_probe.hip -lpthread && ./launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void spin(long* out, int cycles) {
    long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = t0;
}

int main() {
    const int N = 4000;
    for (int work : {0, 4000}) {          // kernel body: ~0 or ~4000 cycles (about 2 us)
        for (int blocks : {1, 160}) {
            for (int T : {1, 2, 4, 7, 8}) {
                std::vector<hipStream_t> st(T);
                for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                auto run = [&](int i) {
                    for (int k = 0; k < N; ++k) spin<<<blocks, 256, 0, st[i]>>>(nullptr, work);
                    hipStreamSynchronize(st[i]);
                };
                for (int i = 0; i < T; ++i) run(i);   // warm
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int i = 0; i < T; ++i) th.emplace_back(run, i);
                for (auto& t : th) t.join();
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                printf("work %4d cycles, %3d blocks, %d threads/streams: %.2f us per launch per stream, %.0f launches/ms total\n",
                       work, blocks, T, dt / N * 1e6, N * T / dt / 1e3);
                for (auto& s : st) hipStreamDestroy(s);
            }
        }
    }
    return 0;
}
