// Adapter pruning (reference slim/lora.py:7-53, prune_and_optimize_lora): the left adapter L [K, r] is pruned 2:4
// and both adapters are re-fitted to the product they had before, by num_iters Adam steps on
//     loss = || L R - T ||_F / || T ||_F .
// The reference runs this as a torch autograd loop; here one call enqueues the whole loop on a stream with no host
// synchronisation: per iteration three fp32 library GEMMs (L R, D R^T, L^T D: plain products, rocBLAS), one fused
// residual + sum-of-squares pass, and one fused Adam + mask pass per adapter that takes the gradient's scale
// 1 / (||D|| ||T||) from device memory.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include "solver.hpp"
#include "f32gemm.hpp"

#define CHK(call)                   \
    do {                            \
        const int rc_ = (call);     \
        if (rc_ != SLIM_OK) return rc_; \
    } while (0)

#define LP_BLOCKS 256

// D = O - T in place over O; per-block partial sums of D^2 in fp64 (fixed block -> element map: deterministic)
__global__ __launch_bounds__(256) void lp_residual_kernel(float* __restrict__ O, const float* __restrict__ T, int64_t n,
                                                          double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)LP_BLOCKS * 256 * 4) {
        if (i + 3 < n) {
            f32x4 o = *(const f32x4*)(O + i);
            const f32x4 t = *(const f32x4*)(T + i);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[c] -= t[c];
                acc += (double)o[c] * (double)o[c];
            }
            *(f32x4*)(O + i) = o;
        } else {
            for (int64_t k = i; k < n; ++k) {
                const float d = O[k] - T[k];
                O[k] = d;
                acc += (double)d * (double)d;
            }
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// sum of squares of T alone (the target's norm, once)
__global__ __launch_bounds__(256) void lp_sumsq_kernel(const float* __restrict__ T, int64_t n, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)LP_BLOCKS * 256) acc += (double)T[i] * (double)T[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// scal[0] = ||T|| (when set_target) or: scal[1] = 1 / (||D|| ||T||), loss_out = ||D|| / ||T||
__global__ __launch_bounds__(LP_BLOCKS) void lp_finish_kernel(const double* __restrict__ part, float* __restrict__ scal,
                                                              int set_target, float* __restrict__ loss_out) {
    __shared__ double red[LP_BLOCKS / 64];
    double v = wave_sum_d(part[threadIdx.x]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < LP_BLOCKS / 64; ++w) s += red[w];
        const float nrm = (float)sqrt(s);
        if (set_target) {
            scal[0] = nrm;
        } else {
            const float tn = scal[0];
            scal[1] = 1.0f / (nrm * tn);
            if (loss_out) *loss_out = nrm / tn;
        }
    }
}

// One Adam step (torch.optim.Adam defaults: no weight decay, no amsgrad) on p with gradient g * scal[1]; entries with
// mask != 0 are reset to zero after the step (slim/lora.py:41).
__global__ __launch_bounds__(256) void lp_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, const uint8_t* __restrict__ mask, int64_t n,
                                                      const float* __restrict__ scal, float lr_over_bc1, float inv_sqrt_bc2,
                                                      float b1, float b2, float eps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * scal[1];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    const float pi = p[i] - lr_over_bc1 * (mi / denom);
    p[i] = (mask && mask[i]) ? 0.f : pi;
}

static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }
struct LpWs {
    float *D, *gL, *gR, *mL, *vL, *mR, *vR, *scal;
    double* part;
    int64_t total;
};
static LpWs lp_ws(char* base, int64_t K, int64_t r, int64_t N) {
    LpWs w;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + o : nullptr;
        o += al256(bytes);
        return p;
    };
    w.D = (float*)take(K * N * 4);
    w.gL = (float*)take(K * r * 4);
    w.gR = (float*)take(r * N * 4);
    w.mL = (float*)take(K * r * 4);
    w.vL = (float*)take(K * r * 4);
    w.mR = (float*)take(r * N * 4);
    w.vR = (float*)take(r * N * 4);
    w.scal = (float*)take(16);
    w.part = (double*)take(LP_BLOCKS * 8);
    w.total = o;
    return w;
}

extern "C" int64_t slim_lora_prune_workspace_bytes(int64_t K, int64_t r, int64_t N) {
    if (K <= 0 || r <= 0 || N <= 0) return 0;
    return lp_ws(nullptr, K, r, N).total;
}

// L [K, r], R [r, N] fp32 in/out (L already zero where L_mask is set); T [K, N] fp32 target; L_mask [K, r] bytes.
// losses (device, num_iters floats, or NULL) receives the loss seen by every iteration.
extern "C" int slim_lora_prune_optimize(float* L, float* R, const uint8_t* L_mask, const float* T, int64_t K, int64_t r,
                                        int64_t N, int num_iters, float lr, float lr_end_factor, float* losses,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(L && R && L_mask && T && workspace, "slim_lora_prune_optimize: null pointer");
    SLIM_REQUIRE(K > 0 && r > 0 && N > 0 && K < (1 << 30) && N < (1 << 30) && r < (1 << 30) && num_iters >= 0,
                 "slim_lora_prune_optimize: bad sizes");
    SLIM_REQUIRE(((uintptr_t)T % 16) == 0 && ((uintptr_t)workspace % 16) == 0, "slim_lora_prune_optimize: buffers must be 16-byte aligned");
    SLIM_REQUIRE(workspace_bytes >= slim_lora_prune_workspace_bytes(K, r, N), "slim_lora_prune_optimize: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // the three fp32 products of an iteration run on the library's own fp32 MFMA GEMM (f32gemm.hip); rocBLAS sgemm only
    // behind SLIM_PRUNE_LORA_LIBRARY_GEMM=1 (about twice as fast on these shapes, and a dlopen'd dependency)
    static const bool library = getenv("SLIM_PRUNE_LORA_LIBRARY_GEMM") && atoi(getenv("SLIM_PRUNE_LORA_LIBRARY_GEMM")) != 0;
    rb_handle h = nullptr;
    if (library) {
        int rc = solver_handle(st, &h);
        if (rc) return rc;
        SLIM_REQUIRE(g_solver.sgemm, "slim_lora_prune_optimize: rocblas_sgemm not available");
    } else {
        SLIM_REQUIRE(K % 4 == 0 && r % 4 == 0 && N % 4 == 0 && ((uintptr_t)L % 16) == 0 && ((uintptr_t)R % 16) == 0,
                     "slim_lora_prune_optimize: K, r, N must be multiples of 4 and L, R 16-byte aligned");
    }
    // C [m, n] = A B with A, B row-major; a_t / b_t: the operand is stored transposed (A as [kd, m], B as [n, kd])
    auto product = [&](const float* A, int lda, bool a_t, const float* B, int ldb, bool b_t, int m, int n, int kd, float* C) {
        F32GemmArgs g;
        g.A = A;
        g.B = B;
        g.C = C;
        g.M = m;
        g.N = n;
        g.Kd = kd;
        g.lda = lda;
        g.ldb = ldb;
        g.ldc = n;
        g.a_kmajor = a_t ? 1 : 0;
        g.b_kmajor = b_t ? 0 : 1;
        return slim_f32gemm_launch(g, stream);
    };
    LpWs w = lp_ws((char*)workspace, K, r, N);
    const int64_t nT = K * N, nL = K * r, nR = r * N;
    SLIM_HIP(hipMemsetAsync(w.mL, 0, (size_t)((char*)w.scal - (char*)w.mL), st));     // both moment pairs
    lp_sumsq_kernel<<<LP_BLOCKS, 256, 0, st>>>(T, nT, w.part);
    SLIM_LAUNCH_CHECK();
    lp_finish_kernel<<<1, LP_BLOCKS, 0, st>>>(w.part, w.scal, 1, nullptr);
    SLIM_LAUNCH_CHECK();
    const float one = 1.f, zero = 0.f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    double p1 = 1.0, p2 = 1.0;
    SLIM_PROF("lora_prune_optimize", st, (double)num_iters * (8.0 * nT + 28.0 * (nL + nR)),
              (double)num_iters * 6.0 * K * r * N, num_iters > 0 ? num_iters : 1);
    for (int it = 0; it < num_iters; ++it) {
        // D = L R
        if (!library) {
            CHK(product(L, (int)r, false, R, (int)N, false, (int)K, (int)N, (int)r, w.D));
        } else if (g_solver.sgemm(h, RB_OP_NONE, RB_OP_NONE, (int)N, (int)K, (int)r, &one, R, (int)N, L, (int)r, &zero, w.D, (int)N) != 0) {
            // (row-major X [a, b] is the column-major b x a matrix X^T with leading dimension b: D^T = R^T L^T)
            slim_set_error("rocblas_sgemm failed");
            return SLIM_ERR_SOLVER;
        }
        lp_residual_kernel<<<LP_BLOCKS, 256, 0, st>>>(w.D, T, nT, w.part);
        SLIM_LAUNCH_CHECK();
        lp_finish_kernel<<<1, LP_BLOCKS, 0, st>>>(w.part, w.scal, 0, losses ? losses + it : nullptr);
        SLIM_LAUNCH_CHECK();
        // gL = D R^T [K, r],  gR = L^T D [r, N]
        if (!library) {
            CHK(product(w.D, (int)N, false, R, (int)N, true, (int)K, (int)r, (int)N, w.gL));
            CHK(product(L, (int)r, true, w.D, (int)N, false, (int)r, (int)N, (int)K, w.gR));
        } else if (g_solver.sgemm(h, RB_OP_TRANSPOSE, RB_OP_NONE, (int)r, (int)K, (int)N, &one, R, (int)N, w.D, (int)N, &zero, w.gL, (int)r) != 0 ||
                   g_solver.sgemm(h, RB_OP_NONE, RB_OP_TRANSPOSE, (int)N, (int)r, (int)K, &one, w.D, (int)N, L, (int)r, &zero, w.gR, (int)N) != 0) {
            slim_set_error("rocblas_sgemm failed");
            return SLIM_ERR_SOLVER;
        }
        p1 *= b1;
        p2 *= b2;
        const double lr_t = (double)lr * (1.0 + ((double)lr_end_factor - 1.0) * (double)it / (double)num_iters);
        const float lr_over_bc1 = (float)(lr_t / (1.0 - p1)), inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - p2));
        lp_adam_kernel<<<(unsigned)ceil_div64(nL, 256), 256, 0, st>>>(L, w.gL, w.mL, w.vL, L_mask, nL, w.scal, lr_over_bc1,
                                                                      inv_sqrt_bc2, b1, b2, eps);
        SLIM_LAUNCH_CHECK();
        lp_adam_kernel<<<(unsigned)ceil_div64(nR, 256), 256, 0, st>>>(R, w.gR, w.mR, w.vR, nullptr, nR, w.scal, lr_over_bc1,
                                                                      inv_sqrt_bc2, b1, b2, eps);
        SLIM_LAUNCH_CHECK();
    }
    return SLIM_OK;
}
