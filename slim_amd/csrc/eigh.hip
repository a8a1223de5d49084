// Eigen-decomposition of a small dense symmetric fp32 matrix (the l x l Rayleigh-Ritz matrix of the low-rank fit,
// l of a few hundred to two thousand) with a short launch chain.
//
// The library routine for this (rocSOLVER syevd) spends most of its time in a blocked Householder
// tridiagonalisation that issues four to five tiny kernels per column; at these sizes the matrix lives in L2 /
// Infinity Cache and the chain is bound by the number of dependent launches, not by arithmetic.  Here the reduction
// to tridiagonal form takes ONE launch per column: the kernel of column j
//   (1) finishes reflector j-1: w = p - (tau/2)(p.v) v from the matrix-vector product p = tau A v that the
//       previous launch left in memory (every workgroup does this redundantly, it is O(n)),
//   (2) forms row j of the updated matrix A - v w^T - w v^T on the fly and derives reflector j from it,
//   (3) applies the rank-2 update of reflector j-1 to its share of the trailing rows while it accumulates their
//       dot products with the new v (the next p), and
//   (4) multiplies its share of the rows of the accumulated orthogonal matrix Q by reflector j (Q <- Q H_j is
//       row-local, so no back-transformation pass over n reflectors is needed afterwards).
// The tridiagonal problem then goes to rocSOLVER's divide-and-conquer solver (stedc) and the wanted eigenvectors
// are Q times its eigenvectors (one fp32 GEMM).
#include "common.hpp"
#include "../../include/slim_hip.h"
#include "eigh.hpp"
#include "solver.hpp"

#define TD_THREADS 256
#define TD_ROWS_PER_WAVE 2

struct TridiagArgs {
    float* A;      // [batch][n, n] symmetric, both triangles valid; destroyed
    float* Q;      // [batch][n, n] accumulated reflectors (row-major), written from column 0 on
    float* vbuf;   // [batch][2, n] reflector vectors, double-buffered by column parity
    float* pbuf;   // [batch][2, n] tau * A v, double-buffered
    float* taus;   // [batch][n]
    float* d;      // [batch][n] diagonal of the tridiagonal matrix
    float* e;      // [batch][n] off-diagonal (n - 1 used)
    int n, j;
};

// Wave-wide sums on the data-parallel-primitive path (cross-lane moves inside the VALU, a few cycles each) instead of
// ds_bpermute shuffles (an LDS round trip each): the kernel below is a chain of four dependent reductions and is
// latency-bound, so the shuffles alone were a third of its run time.  After the four steps every lane of a 16-lane
// row holds the row's sum; the four row sums are combined through scalar lane reads.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_fast(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_d(double x, int lane) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_fast_d(double v) {
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return lane_d(v, 0) + lane_d(v, 16) + lane_d(v, 32) + lane_d(v, 48);
}

// sum over the 256 threads; `red` must be a slot of four doubles that no earlier call in the kernel used
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_fast_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// MAXV: float4 chunks per lane per row, n <= 256 * MAXV
template <int MAXV>
__global__ __launch_bounds__(TD_THREADS) void tridiag_step_kernel(TridiagArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[8];
    const int n = g.n, j = g.j, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    {
        const int64_t bz = blockIdx.y;
        g.A += bz * n * n;
        g.Q += bz * n * n;
        g.vbuf += bz * 2 * n;
        g.pbuf += bz * 2 * n;
        g.taus += bz * n;
        g.d += bz * n;
        g.e += bz * n;
    }
    float* vp = sm;            // previous reflector vector (entries >= j, vp[j] == 1)
    float* wp = sm + n;        // its w
    float* v = sm + 2 * n;     // this column's reflector (zero up to and including j)

    // Everything this workgroup reads from memory was written by the previous launch on other CUs, so each read is a
    // trip to the memory side of the chip (a microsecond).  All of them are issued here, before the first barrier:
    // the previous reflector's vectors, row j, and the rows this wave will update.
    const int nA = n - j - 1;                              // trailing rows j+1 .. n-1
    const int mstart = (j + 1) / 256;
    const int msq = j == 0 ? 0 : mstart;
    int task_row[TD_ROWS_PER_WAVE];                        // >= 0: row of A; < 0: ~row of Q; INT_MIN: none
    f32x4 rowv[TD_ROWS_PER_WAVE][MAXV];
#pragma unroll
    for (int rr = 0; rr < TD_ROWS_PER_WAVE; ++rr) {
        const int task = ((int)blockIdx.x * (TD_THREADS / 64) + wave) * TD_ROWS_PER_WAVE + rr;
        const bool isA = task < nA, isQ = !isA && task < nA + n;
        const int i = isA ? j + 1 + task : task - nA;
        task_row[rr] = isA ? i : (isQ ? ~i : (int)0x80000000);
        const float* row = (isA ? g.A : g.Q) + (int64_t)i * n;
#pragma unroll
        for (int m = 0; m < MAXV; ++m) {
            const int k = 256 * m + 4 * lane;
            rowv[rr][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((isA && m >= mstart && k < n) || (isQ && j > 0 && m >= msq && k < n)) rowv[rr][m] = *(const f32x4*)(row + k);
            if (isQ && j == 0 && k < n) {
#pragma unroll
                for (int c = 0; c < 4; ++c) rowv[rr][m][c] = (k + c == i) ? 1.f : 0.f;
            }
        }
    }
    float pv[MAXV], pp[MAXV], px[MAXV];                    // strided shares of v', p' and row j (n <= 256 * MAXV)
    const float tau_p = j > 0 ? g.taus[j - 1] : 0.f;
    {
        const float* vg = g.vbuf + ((j - 1) & 1) * n;
        const float* pg = g.pbuf + ((j - 1) & 1) * n;
        const float* rowj = g.A + (int64_t)j * n;
#pragma unroll
        for (int s = 0; s < MAXV; ++s) {
            const int k = j + t + TD_THREADS * s;
            pv[s] = pp[s] = px[s] = 0.f;
            if (k < n) {
                px[s] = rowj[k];
                if (j > 0) {
                    pv[s] = vg[k];
                    pp[s] = pg[k];
                }
            }
        }
    }

    // (1) finish reflector j-1
    if (j > 0) {
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < MAXV; ++s) dot += (double)pv[s] * (double)pp[s];
        dot = block_sum_d(dot, red);
        const float alpha = -0.5f * tau_p * (float)dot;
#pragma unroll
        for (int s = 0; s < MAXV; ++s) {
            const int k = j + t + TD_THREADS * s;
            if (k < n) {
                vp[k] = pv[s];
                wp[k] = pp[s] + alpha * pv[s];
            }
        }
        __syncthreads();
    }

    // (2) row j of the updated matrix -> d[j] and reflector j from its entries right of the diagonal
    const float wj = j > 0 ? wp[j] : 0.f;
    double nrm = 0.0;
#pragma unroll
    for (int s = 0; s < MAXV; ++s) {
        const int k = j + t + TD_THREADS * s;
        if (k < n) {
            float x = px[s];
            if (j > 0) x -= wp[k] + wj * vp[k];      // v'[j] w'[k] + w'[j] v'[k] with v'[j] == 1
            v[k] = x;
            if (k >= j + 2) nrm += (double)x * (double)x;
        }
    }
    nrm = block_sum_d(nrm, red + 4);                 // its barrier also publishes v[]
    const float dj = v[j], a1 = v[j + 1];
    const float xnorm = (float)sqrt(nrm);
    float beta, tau, scale;
    if (xnorm == 0.f) {
        tau = 0.f;
        beta = a1;
        scale = 0.f;
    } else {
        beta = -copysignf(hypotf(a1, xnorm), a1);
        tau = (beta - a1) / beta;
        scale = 1.f / (a1 - beta);
    }
    __syncthreads();                                 // every thread has read v[j], v[j + 1]
    for (int k = j + t; k < n; k += TD_THREADS) v[k] = k <= j ? 0.f : (k == j + 1 ? 1.f : v[k] * scale);
    for (int k = t; k < j; k += TD_THREADS) v[k] = 0.f;
    __syncthreads();
    if (blockIdx.x == 0) {
        float* vg = g.vbuf + (j & 1) * n;
        for (int k = j + 1 + t; k < n; k += TD_THREADS) vg[k] = v[k];
        if (t == 0) {
            g.taus[j] = tau;
            g.d[j] = dj;
            g.e[j] = beta;
        }
    }

    // (3) + (4): one row per wave at a time, from the registers filled above
    float* pg = g.pbuf + (j & 1) * n;
#pragma unroll
    for (int rr = 0; rr < TD_ROWS_PER_WAVE; ++rr) {
        const int tr = task_row[rr];
        if (tr == (int)0x80000000) continue;
        if (tr >= 0) {
            const int i = tr;
            float* row = g.A + (int64_t)i * n;
            const float vi = j > 0 ? vp[i] : 0.f, wi = j > 0 ? wp[i] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < MAXV; ++m) {
                const int k = 256 * m + 4 * lane;
                if (m < mstart || k >= n) continue;
                f32x4 a = rowv[rr][m];
                const f32x4 vv = *(const f32x4*)(v + k);
                if (j > 0) {
                    const f32x4 w4 = *(const f32x4*)(wp + k), v4 = *(const f32x4*)(vp + k);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (k + c >= j + 1) a[c] -= vi * w4[c] + wi * v4[c];
                    *(f32x4*)(row + k) = a;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc += a[c] * vv[c];     // v is zero left of column j + 1
                if (i == n - 1 && k <= n - 1 && n - 1 < k + 4) g.d[n - 1] = a[(n - 1) - k];   // final once j == n - 2
            }
            acc = wave_sum_fast(acc);
            if (lane == 0) pg[i] = tau * acc;
        } else if (j == 0 || tau != 0.f) {                 // Q is created at j == 0
            const int i = ~tr;
            float* row = g.Q + (int64_t)i * n;
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < MAXV; ++m) {
                const int k = 256 * m + 4 * lane;
                if (m < msq || k >= n) continue;
                const f32x4 vv = *(const f32x4*)(v + k);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc += rowv[rr][m][c] * vv[c];
            }
            const float s = tau * wave_sum_fast(acc);
#pragma unroll
            for (int m = 0; m < MAXV; ++m) {
                const int k = 256 * m + 4 * lane;
                if (m < msq || k >= n) continue;
                const f32x4 vv = *(const f32x4*)(v + k);
                f32x4 q = rowv[rr][m];
#pragma unroll
                for (int c = 0; c < 4; ++c) q[c] -= s * vv[c];
                *(f32x4*)(row + k) = q;
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

struct EighWs {
    float *Q, *C, *e, *vbuf, *pbuf, *taus;
    int64_t total;
};
static EighWs eigh_ws(char* base, int n, int batch) {
    EighWs w;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + o : nullptr;
        o += al256(bytes * batch);
        return (float*)p;
    };
    w.Q = take((int64_t)n * n * 4);
    w.C = take((int64_t)n * n * 4);
    w.e = take((int64_t)n * 4);
    w.vbuf = take((int64_t)2 * n * 4);
    w.pbuf = take((int64_t)2 * n * 4);
    w.taus = take((int64_t)n * 4);
    w.total = o;
    return w;
}

bool slim_eigh_supported(int n) {
    rb_handle h = nullptr;
    if (n < 3 || n > SLIM_EIGH_MAX_N || n % 4 != 0) return false;
    if (solver_handle(nullptr, &h) != SLIM_OK) return false;
    return g_solver.stedc && g_solver.sgemm;
}

int64_t slim_eigh_ws_bytes(int n, int batch) { return eigh_ws(nullptr, n, batch).total; }

template <int MAXV>
static void launch_step(const TridiagArgs& a, int batch, hipStream_t st) {
    const int n = a.n, j = a.j;
    const int tasks = (n - j - 1) + (j == n - 2 ? 0 : n);
    const int per_wg = (TD_THREADS / 64) * TD_ROWS_PER_WAVE;
    dim3 grid((unsigned)((tasks + per_wg - 1) / per_wg), (unsigned)batch);
    hipLaunchKernelGGL(tridiag_step_kernel<MAXV>, grid, dim3(TD_THREADS), (size_t)3 * n * sizeof(float), st, a);
}

// Householder reduction Q^T A Q = T of `batch` symmetric matrices, one launch per column: d / e receive the diagonal and
// off-diagonal of T, Q the accumulated reflectors (row-major); A is destroyed.
static int tridiagonalize(float* A, float* Q, float* vbuf, float* pbuf, float* taus, float* d, float* e, int n, int batch,
                          hipStream_t st) {
    // algorithmic traffic: launch j reads and writes the trailing (n-j-1)^2 block of A and n x (n-j-1) of Q, fp32
    double bytes = 0.0;
    for (int j = 0; j <= n - 2; ++j) {
        const double w = n - j - 1;
        bytes += 8.0 * (w * w + (j == n - 2 ? 0.0 : (double)n * w));
    }
    SLIM_PROF("eigh_tridiag", st, bytes * batch, (4.0 / 3.0 + 2.0) * n * (double)n * n * batch, n - 1);
    TridiagArgs a{A, Q, vbuf, pbuf, taus, d, e, n, 0};
    for (int j = 0; j <= n - 2; ++j) {
        a.j = j;
        if (n <= 768) launch_step<3>(a, batch, st);
        else if (n <= 1280) launch_step<5>(a, batch, st);
        else launch_step<8>(a, batch, st);
    }
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

int slim_eigh_launch(float* G, int n, int batch, int top, float* evals, int* info, int info_stride, void* ws,
                     int64_t ws_bytes, hipStream_t st) {
    SLIM_REQUIRE(G && evals && info && ws && batch >= 1 && batch <= 65535, "slim_eigh: bad arguments");
    SLIM_REQUIRE(n >= 3 && n <= SLIM_EIGH_MAX_N && n % 4 == 0, "slim_eigh: n must be a multiple of 4 in [4, 2048]");
    SLIM_REQUIRE(top >= 1 && top <= n, "slim_eigh: bad number of wanted eigenvectors");
    SLIM_REQUIRE(((uintptr_t)G % 16) == 0 && ((uintptr_t)ws % 16) == 0, "slim_eigh: buffers must be 16-byte aligned");
    SLIM_REQUIRE(ws_bytes >= slim_eigh_ws_bytes(n, batch), "slim_eigh: workspace too small");
    rb_handle h = nullptr;
    int rc = solver_handle(st, &h);
    if (rc) return rc;
    SLIM_REQUIRE(g_solver.stedc && g_solver.sgemm, "slim_eigh: rocsolver_sstedc / rocblas_sgemm not available");
    EighWs w = eigh_ws((char*)ws, n, batch);
    rc = tridiagonalize(G, w.Q, w.vbuf, w.pbuf, w.taus, evals, w.e, n, batch, st);
    if (rc) return rc;
    {
        SLIM_PROF("solver_stedc", st, 0.0, 4.0 * n * (double)n * n * batch);
        for (int b = 0; b < batch; ++b)
            if (g_solver.stedc(h, RB_EVECT_TRIDIAGONAL, n, evals + (int64_t)b * n, w.e + (int64_t)b * n,
                               w.C + (int64_t)b * n * n, n, info + (int64_t)b * info_stride) != 0) {
                slim_set_error("rocsolver_sstedc failed");
                return SLIM_ERR_SOLVER;
            }
    }
    {
        // eigenvector k of G = Q c_k.  Column-major view: the buffer holding row-major Q is Q^T, the solver's C holds c_k
        // as column k, and column k of the product lands where the row-major result keeps "row k = eigenvector k".
        SLIM_PROF("eigh_backproduct", st, 0.0, 2.0 * n * (double)n * top * batch);
        const float one = 1.f, zero = 0.f;
        const int64_t off = (int64_t)(n - top) * n;
        for (int b = 0; b < batch; ++b)
            if (g_solver.sgemm(h, RB_OP_TRANSPOSE, RB_OP_NONE, n, top, n, &one, w.Q + (int64_t)b * n * n, n,
                               w.C + (int64_t)b * n * n + off, n, &zero, G + (int64_t)b * n * n + off, n) != 0) {
                slim_set_error("rocblas_sgemm failed");
                return SLIM_ERR_SOLVER;
            }
    }
    return SLIM_OK;
}

// The reduction alone (no library call): A [n, n] symmetric fp32 (destroyed) -> d [n], e [n] (n - 1 used), Q [n, n]
// row-major with Q^T A Q = tridiag(d, e).  workspace: 5 n floats.
extern "C" int slim_tridiagonalize_f32(float* A, int n, float* d, float* e, float* Q, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
    SLIM_REQUIRE(A && d && e && Q && workspace, "slim_tridiagonalize_f32: null pointer");
    SLIM_REQUIRE(n >= 3 && n <= SLIM_EIGH_MAX_N && n % 4 == 0, "slim_tridiagonalize_f32: n must be a multiple of 4 in [4, 2048]");
    SLIM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)Q % 16) == 0, "slim_tridiagonalize_f32: buffers must be 16-byte aligned");
    SLIM_REQUIRE(workspace_bytes >= (int64_t)5 * n * 4, "slim_tridiagonalize_f32: workspace too small");
    float* ws = (float*)workspace;
    return tridiagonalize(A, Q, ws, ws + 2 * n, ws + 4 * n, d, e, n, 1, (hipStream_t)stream);
}

// `batch` reductions in one launch chain: A [batch][n, n] (destroyed), d / e [batch][n], Q [batch][n, n]; workspace of
// 5 n floats per problem.
extern "C" int slim_tridiagonalize_f32_batched(float* A, int n, int batch, float* d, float* e, float* Q, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(A && d && e && Q && workspace && batch >= 1 && batch <= 65535, "slim_tridiagonalize_f32_batched: bad arguments");
    SLIM_REQUIRE(n >= 3 && n <= SLIM_EIGH_MAX_N && n % 4 == 0, "slim_tridiagonalize_f32_batched: n must be a multiple of 4 in [4, 2048]");
    SLIM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)Q % 16) == 0, "slim_tridiagonalize_f32_batched: buffers must be 16-byte aligned");
    SLIM_REQUIRE(workspace_bytes >= (int64_t)5 * n * 4 * batch, "slim_tridiagonalize_f32_batched: workspace too small");
    float* ws = (float*)workspace;
    return tridiagonalize(A, Q, ws, ws + (int64_t)2 * n * batch, ws + (int64_t)4 * n * batch, d, e, n, batch, (hipStream_t)stream);
}

// Second stage for one problem: eigen-decomposition of tridiag(d, e) by rocSOLVER stedc (d becomes the eigenvalues,
// ascending; e is destroyed) and rows n-top .. n-1 of vecs [n, n] = the eigenvectors of the `top` largest eigenvalues of
// the original matrix (Q times the tridiagonal ones).  workspace: n * n floats.
extern "C" int slim_eigh_tail_f32(float* d, float* e, const float* Q, int n, int top, float* vecs, int32_t* info, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(d && e && Q && vecs && info && workspace, "slim_eigh_tail_f32: null pointer");
    SLIM_REQUIRE(n >= 3 && n <= SLIM_EIGH_MAX_N && top >= 1 && top <= n, "slim_eigh_tail_f32: bad sizes");
    SLIM_REQUIRE(workspace_bytes >= (int64_t)n * n * 4, "slim_eigh_tail_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    rb_handle h = nullptr;
    int rc = solver_handle(st, &h);
    if (rc) return rc;
    SLIM_REQUIRE(g_solver.stedc && g_solver.sgemm, "slim_eigh_tail_f32: rocsolver_sstedc / rocblas_sgemm not available");
    float* C = (float*)workspace;
    {
        SLIM_PROF("solver_stedc", st, 0.0, 4.0 * n * (double)n * n);
        if (g_solver.stedc(h, RB_EVECT_TRIDIAGONAL, n, d, e, C, n, info) != 0) {
            slim_set_error("rocsolver_sstedc failed");
            return SLIM_ERR_SOLVER;
        }
    }
    SLIM_PROF("eigh_backproduct", st, 0.0, 2.0 * n * (double)n * top);
    const float one = 1.f, zero = 0.f;
    const int64_t off = (int64_t)(n - top) * n;
    if (g_solver.sgemm(h, RB_OP_TRANSPOSE, RB_OP_NONE, n, top, n, &one, Q, n, C + off, n, &zero, vecs + off, n) != 0) {
        slim_set_error("rocblas_sgemm failed");
        return SLIM_ERR_SOLVER;
    }
    return SLIM_OK;
}

extern "C" int64_t slim_eigh_workspace_bytes(int n) { return n > 0 ? slim_eigh_ws_bytes(n, 1) : 0; }

// Eigenvalues (ascending) and the `top` leading eigenvectors of the symmetric matrix G [n, n] (fp32, both triangles
// given).  On return row n-1-i of G is the eigenvector of the i-th largest eigenvalue, for i < top; the other rows
// hold no result.  info [1] receives the tridiagonal solver's status word (0 = converged).
extern "C" int slim_eigh_f32(float* G, int n, int top, float* evals, int32_t* info, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    return slim_eigh_launch(G, n, 1, top, evals, info, 1, workspace, workspace_bytes, (hipStream_t)stream);
}
