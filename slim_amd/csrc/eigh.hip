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
// The tridiagonal problem is then solved in double precision, whole batch at once, in two launches: every eigenvalue by
// multisection on Sturm counts (a wavefront per eigenvalue, 64 shifts per sweep), every wanted eigenvector by a twisted
// factorisation at its eigenvalue (a thread per eigenvector).  The matrix is only known to fp32 accuracy, so eigenvalues
// that coincide to fp32 rounding are still ~1e7 units of double rounding apart: each vector is resolved on its own and
// no re-orthogonalisation inside clusters is needed.  The wanted eigenvectors of the dense matrix are Q times the
// tridiagonal ones (one fp32 MFMA GEMM, f32gemm.hip).  SLIM_EIGH_LIBRARY_TAIL=1 sends the tridiagonal problem to
// rocSOLVER's divide-and-conquer solver (stedc) and rocBLAS instead.
#include "common.hpp"
#include "../../include/slim_hip.h"
#include "eigh.hpp"
#include "solver.hpp"
#include "f32gemm.hpp"
#include <cfloat>
#include <cstdlib>

#define TD_THREADS 256
#define TD_ROWS_PER_WAVE 2
#define TW_CHUNK 16          // stored pivots / multipliers the twisted-factorisation kernel requests at a time

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


// ---- tridiagonal eigen-solver in double precision -------------------------------------------------------------------
// Number of eigenvalues of tridiag(d, e) that are <= x (LAPACK dstebz's count: negative pivots of the LDL^T factorisation
// of T - x I, a pivot smaller than pivmin in magnitude counted as negative).  e2[i] = e[i]^2.
// 1 / x for the pivot recurrences below: the hardware's reciprocal estimate and two Newton steps (full double precision
// for the normal, non-zero pivots these recurrences guard with pivmin / tiny), half the dependent instructions of an IEEE
// division -- the recurrences are chains of n dependent divisions and nothing else
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Number of eigenvalues of tridiag(d, e) that are <= x (LAPACK dstebz's count: negative pivots of the LDL^T factorisation
// of T - x I, a pivot smaller than pivmin in magnitude counted as negative).  e2[i] = e[i]^2.
__device__ __forceinline__ int sturm_count(const double* __restrict__ d, const double* __restrict__ e2, int n, double x,
                                           double pivmin) {
    double q = d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int cnt = q <= 0.0;
    for (int i = 1; i < n; ++i) {
        q = d[i] - x - e2[i - 1] * fast_rcp(q);
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += q <= 0.0;
    }
    return cnt;
}

struct TriEigArgs {
    const float* d;     // [batch][n]
    const float* e;     // [batch][n] (n - 1 used)
    double* lam;        // [batch][n] eigenvalues, ascending (workspace)
    float* evals;       // [batch][n] the same rounded to fp32
    double* lplus;      // [batch][n][ldz]
    double* uminus;     // [batch][n][ldz]
    double* zbuf;       // [batch][n][ldz]
    float* Z;           // [batch][n][ldz]: column m holds the eigenvector of eigenvalue n - ldz + m
    int* info;          // status word per problem (stride info_stride): bit 0 set = a vector overflowed
    int n, ldz, info_stride;
};

// one wavefront per eigenvalue: the interval [lo, hi] that holds eigenvalue k shrinks 65-fold per sweep
__global__ __launch_bounds__(256) void tri_multisect_kernel(TriEigArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sd[];      // d [n], e2 [n]
    __shared__ double red[8];
    const int n = g.n, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t bz = blockIdx.y;
    const float* d32 = g.d + bz * n;
    const float* e32 = g.e + bz * n;
    double* d = sd;
    double* e2 = sd + n;
    double gl = DBL_MAX, gu = -DBL_MAX, emax = 0.0;
    for (int i = t; i < n; i += 256) {
        const double di = (double)d32[i];
        const double el = i > 0 ? fabs((double)e32[i - 1]) : 0.0, er = i < n - 1 ? fabs((double)e32[i]) : 0.0;
        d[i] = di;
        e2[i] = er * er;
        gl = fmin(gl, di - el - er);
        gu = fmax(gu, di + el + er);
        emax = fmax(emax, er * er);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gl = fmin(gl, __shfl_xor(gl, o, 64));
        gu = fmax(gu, __shfl_xor(gu, o, 64));
        emax = fmax(emax, __shfl_xor(emax, o, 64));
    }
    if (lane == 0) {
        red[wave] = gl;
        red[4 + wave] = gu;
    }
    __shared__ double red_e[4];
    if (lane == 0) red_e[wave] = emax;
    __syncthreads();
    gl = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    gu = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    emax = fmax(fmax(red_e[0], red_e[1]), fmax(red_e[2], red_e[3]));
    const double norm = fmax(fabs(gl), fabs(gu));
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    const double slack = 2.0 * DBL_EPSILON * norm * n + 2.0 * pivmin;
    const int k = (int)blockIdx.x * 4 + wave;        // eigenvalue index, ascending
    if (k >= n) return;
    double lo = gl - slack, hi = gu + slack;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double w = hi - lo;
        if (w <= 2.0 * DBL_EPSILON * norm + 2.0 * pivmin) break;     // accuracy relative to the matrix, as the vectors need it
        // (two shifts per lane, 129-fold per sweep, measured slower: 753 against 590 us -- a sweep is bound by fp64 issue)
        const double x = lo + w * ((double)(lane + 1) / 65.0);
        const int cnt = sturm_count(d, e2, n, x, pivmin);
        // eigenvalue k is the smallest x with count(x) >= k + 1: the first lane that reaches the count bounds it from above
        const unsigned long long ok = __ballot(cnt >= k + 1);
        const int f = ok ? __ffsll((long long)ok) - 1 : 64;
        const double xf = __shfl(x, f < 64 ? f : 63, 64), xp = __shfl(x, f > 0 ? f - 1 : 0, 64);
        const double nhi = f < 64 ? xf : hi, nlo = f > 0 ? xp : lo;
        hi = nhi;
        lo = nlo;
    }
    if (lane == 0) {
        const double lam = 0.5 * (lo + hi);
        g.lam[bz * n + k] = lam;
        g.evals[bz * n + k] = (float)lam;
    }
}

// one thread per wanted eigenvector: twisted factorisation of T - lam I (forward L D L^T and backward U D U^T pivots, joined
// at the index where they fit best), which gives the eigenvector of an isolated eigenvalue to high relative accuracy in
// one pass, without iteration or starting vector
__global__ __launch_bounds__(64) void tri_twisted_kernel(TriEigArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sd[];      // d [n], e [n]
    const int n = g.n, ldz = g.ldz, t = threadIdx.x;
    const int64_t bz = blockIdx.y;
    double* d = sd;
    double* e = sd + n;
    double nrm = 0.0;
    for (int i = t; i < n; i += 64) {
        d[i] = (double)g.d[bz * n + i];
        e[i] = i < n - 1 ? (double)g.e[bz * n + i] : 0.0;
        nrm = fmax(nrm, fabs(d[i]) + 2.0 * fabs(e[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o, 64));
    __syncthreads();
    const int col = (int)blockIdx.x * 64 + t;        // column of Z
    if (col >= ldz) return;
    const int m = n - ldz + col;                     // eigenvalue index
    const double* lamv = g.lam + bz * n;
    const double lam = lamv[m];
    double* lp = g.lplus + bz * n * ldz + col;
    double* um = g.uminus + bz * n * ldz + col;
    double* zb = g.zbuf + bz * n * ldz + col;
    const double tiny = fmax(nrm, DBL_MIN * 1e30) * 1e-30;       // a zero pivot is a shift of the eigenvalue by this much
    // forward pivots D+ (kept in zb) and multipliers
    double dp = d[0] - lam;
    for (int i = 0; i < n - 1; ++i) {
        if (fabs(dp) < tiny) dp = dp < 0.0 ? -tiny : tiny;
        zb[(int64_t)i * ldz] = dp;
        const double l = e[i] * fast_rcp(dp);
        lp[(int64_t)i * ldz] = l;
        dp = (d[i + 1] - lam) - l * e[i];
    }
    if (fabs(dp) < tiny) dp = dp < 0.0 ? -tiny : tiny;
    zb[(int64_t)(n - 1) * ldz] = dp;
    // how many eigenvalues below this one coincide with it to rounding: copies of an eigenvalue of a matrix that splits
    // into blocks each take another of the equally good twist indices, so that their vectors sit in different blocks
    int dup = 0;
    for (int j = m - 1; j >= 0 && fabs(lamv[j] - lam) <= 16.0 * DBL_EPSILON * nrm; --j) ++dup;
    // backward pivots D- and multipliers; gamma_i = D+_i + D-_i - (d_i - lam), twist at the smallest |gamma|
    int kbest = n - 1;
    double gbest = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double dm = d[n - 1] - lam;
        int seen = 0;
        double best = fabs(zb[(int64_t)(n - 1) * ldz]);
        if (pass == 1 && best <= gbest * (1.0 + 1e-6) + tiny) {
            if (seen == dup) kbest = n - 1;
            ++seen;
        }
        if (pass == 0) kbest = n - 1;
        // the forward pivots come back from memory TW_CHUNK at a time, all requested before the first is used: one trip to
        // memory per chunk instead of one per step of the recurrence
        for (int i0 = n - 2; i0 >= 0; i0 -= TW_CHUNK) {
            double zf[TW_CHUNK];
#pragma unroll
            for (int c = 0; c < TW_CHUNK; ++c) zf[c] = i0 - c >= 0 ? zb[(int64_t)(i0 - c) * ldz] : 0.0;
#pragma unroll
            for (int c = 0; c < TW_CHUNK; ++c) {
                const int i = i0 - c;
                if (i >= 0) {
                    if (fabs(dm) < tiny) dm = dm < 0.0 ? -tiny : tiny;
                    const double u = e[i] * fast_rcp(dm);
                    if (pass == 0) um[(int64_t)i * ldz] = u;
                    dm = (d[i] - lam) - u * e[i];
                    const double gam = fabs(zf[c] + dm - (d[i] - lam));
                    if (pass == 0) {
                        if (gam < best) {
                            best = gam;
                            kbest = i;
                        }
                    } else if (gam <= gbest * (1.0 + 1e-6) + tiny) {
                        if (seen == dup) kbest = i;
                        ++seen;
                    }
                }
            }
        }
        if (pass == 0) gbest = best;
        if (dup == 0) break;
    }
    // the vector: 1 at the twist, multipliers outwards
    double z = 1.0, ss = 1.0;
    zb[(int64_t)kbest * ldz] = 1.0;
    for (int i0 = kbest - 1; i0 >= 0; i0 -= TW_CHUNK) {
        double m[TW_CHUNK];
#pragma unroll
        for (int c = 0; c < TW_CHUNK; ++c) m[c] = i0 - c >= 0 ? lp[(int64_t)(i0 - c) * ldz] : 0.0;
#pragma unroll
        for (int c = 0; c < TW_CHUNK; ++c) {
            const int i = i0 - c;
            if (i >= 0) {
                z = -m[c] * z;
                zb[(int64_t)i * ldz] = z;
                ss += z * z;
            }
        }
    }
    z = 1.0;
    for (int i0 = kbest; i0 < n - 1; i0 += TW_CHUNK) {
        double m[TW_CHUNK];
#pragma unroll
        for (int c = 0; c < TW_CHUNK; ++c) m[c] = i0 + c < n - 1 ? um[(int64_t)(i0 + c) * ldz] : 0.0;
#pragma unroll
        for (int c = 0; c < TW_CHUNK; ++c) {
            const int i = i0 + c;
            if (i < n - 1) {
                z = -m[c] * z;
                zb[(int64_t)(i + 1) * ldz] = z;
                ss += z * z;
            }
        }
    }
    const bool finite = ss < DBL_MAX && ss == ss;
    if (!finite) atomicOr(g.info + bz * g.info_stride, 1);
    const double inv = finite ? 1.0 / sqrt(ss) : 0.0;
    float* Zc = g.Z + bz * n * ldz + col;
    for (int i0 = 0; i0 < n; i0 += TW_CHUNK) {
        double m[TW_CHUNK];
#pragma unroll
        for (int c = 0; c < TW_CHUNK; ++c) m[c] = i0 + c < n ? zb[(int64_t)(i0 + c) * ldz] : 0.0;
#pragma unroll
        for (int c = 0; c < TW_CHUNK; ++c)
            if (i0 + c < n) Zc[(int64_t)(i0 + c) * ldz] = (float)(m[c] * inv);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

// workspace of the tridiagonal stage for `batch` problems with ldz wanted vectors each
struct TailWs {
    double *lam, *lplus, *uminus, *zbuf;
    float *dcopy, *Z, *C;      // C: n x n scratch of the library path
    int64_t total;
};
static TailWs tail_ws(char* base, int n, int ldz, int batch) {
    TailWs w;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + o : nullptr;
        o += al256(bytes * batch);
        return p;
    };
    w.lam = (double*)take((int64_t)n * 8);
    w.lplus = (double*)take((int64_t)n * ldz * 8);
    w.uminus = (double*)take((int64_t)n * ldz * 8);
    w.zbuf = (double*)take((int64_t)n * ldz * 8);
    w.dcopy = (float*)take((int64_t)n * 4);
    w.Z = (float*)take((int64_t)n * ldz * 4);
    w.C = w.Z;                 // the library path needs n x n floats: callers size ldz = n for it
    w.total = o;
    return w;
}
static inline int round_up4(int v) { return (v + 3) & ~3; }
static bool library_tail() {
    static const bool on = [] {
        const char* v = getenv("SLIM_EIGH_LIBRARY_TAIL");
        return v && v[0] == '1';
    }();
    return on;
}

struct EighWs {
    float *Q, *e, *vbuf, *pbuf, *taus;
    char* tail;
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
    w.e = take((int64_t)n * 4);
    w.vbuf = take((int64_t)2 * n * 4);
    w.pbuf = take((int64_t)2 * n * 4);
    w.taus = take((int64_t)n * 4);
    w.tail = base ? base + o : nullptr;
    o += tail_ws(nullptr, n, n, batch).total;
    w.total = o;
    return w;
}

bool slim_eigh_supported(int n) { return n >= 4 && n <= SLIM_EIGH_MAX_N && n % 4 == 0; }

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

// Second stage for `batch` problems: tridiag(d, e) [batch][n] with the reduction's Q [batch][n, n] -> eigenvalues ascending in
// `evals` (which may be the buffer d), rows n - top .. n - 1 of vecs [batch][n, n] = eigenvectors of the `top` largest
// eigenvalues of the dense matrix; info[b * info_stride] = status of problem b.
static int eigh_tail(float* d, float* e, const float* Q, int n, int batch, int top, float* evals, float* vecs, int* info,
                     int info_stride, char* ws, int64_t ws_bytes, hipStream_t st) {
    const int ldz = library_tail() ? n : round_up4(top);
    TailWs w = tail_ws(ws, n, ldz, batch);
    SLIM_REQUIRE(ws_bytes >= w.total, "slim_eigh: workspace too small");
    if (library_tail()) {
        rb_handle h = nullptr;
        int rc = solver_handle(st, &h);
        if (rc) return rc;
        SLIM_REQUIRE(g_solver.stedc && g_solver.sgemm, "slim_eigh: rocsolver_sstedc / rocblas_sgemm not available");
        if (evals != d) SLIM_HIP(hipMemcpyAsync(evals, d, (size_t)batch * n * sizeof(float), hipMemcpyDeviceToDevice, st));
        const float one = 1.f, zero = 0.f;
        const int64_t off = (int64_t)(n - top) * n;
        for (int b = 0; b < batch; ++b) {
            float* C = w.C + (int64_t)b * n * n;
            {
                SLIM_PROF("solver_stedc", st, 0.0, 4.0 * n * (double)n * n);
                if (g_solver.stedc(h, RB_EVECT_TRIDIAGONAL, n, evals + (int64_t)b * n, e + (int64_t)b * n, C, n,
                                   info + (int64_t)b * info_stride) != 0) {
                    slim_set_error("rocsolver_sstedc failed");
                    return SLIM_ERR_SOLVER;
                }
            }
            // eigenvector k of G = Q c_k.  Column-major view: the buffer holding row-major Q is Q^T, the solver's C holds c_k
            // as column k, and column k of the product lands where the row-major result keeps "row k = eigenvector k".
            SLIM_PROF("eigh_backproduct", st, 0.0, 2.0 * n * (double)n * top);
            if (g_solver.sgemm(h, RB_OP_TRANSPOSE, RB_OP_NONE, n, top, n, &one, Q + (int64_t)b * n * n, n, C + off, n, &zero,
                               vecs + (int64_t)b * n * n + off, n) != 0) {
                slim_set_error("rocblas_sgemm failed");
                return SLIM_ERR_SOLVER;
            }
        }
        return SLIM_OK;
    }
    SLIM_HIP(hipMemcpyAsync(w.dcopy, d, (size_t)batch * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemset2DAsync(info, (size_t)info_stride * sizeof(int), 0, sizeof(int), (size_t)batch, st));
    TriEigArgs a{w.dcopy, e, w.lam, evals, w.lplus, w.uminus, w.zbuf, w.Z, info, n, ldz, info_stride};
    {
        // a sweep evaluates 64 shifts of n pivots each per eigenvalue; about eight sweeps reach double precision
        SLIM_PROF("eigh_multisect", st, 0.0, 8.0 * 64.0 * 3.0 * n * (double)n * batch);
        SLIM_HIP(slim_allow_dynamic_lds((const void*)tri_multisect_kernel, 2 * SLIM_EIGH_MAX_N * (int)sizeof(double)));
        tri_multisect_kernel<<<dim3((unsigned)((n + 3) / 4), (unsigned)batch), 256, (size_t)2 * n * sizeof(double), st>>>(a);
        SLIM_LAUNCH_CHECK();
    }
    {
        SLIM_PROF("eigh_twisted", st, 44.0 * n * (double)ldz * batch, 12.0 * n * (double)ldz * batch);
        SLIM_HIP(slim_allow_dynamic_lds((const void*)tri_twisted_kernel, 2 * SLIM_EIGH_MAX_N * (int)sizeof(double)));
        tri_twisted_kernel<<<dim3((unsigned)((ldz + 63) / 64), (unsigned)batch), 64, (size_t)2 * n * sizeof(double), st>>>(a);
        SLIM_LAUNCH_CHECK();
    }
    // vecs[n - ldz + m][i] = sum_k Z[k][m] Q[i][k]
    F32GemmArgs g;
    g.A = w.Z;
    g.lda = ldz;
    g.a_kmajor = 1;
    g.B = Q;
    g.ldb = n;
    g.b_kmajor = 0;
    g.M = ldz;
    g.N = n;
    g.Kd = n;
    g.C = vecs + (int64_t)(n - ldz) * n;
    g.ldc = n;
    g.batch = batch;
    g.sA = (int64_t)n * ldz;
    g.sB = g.sC = (int64_t)n * n;
    return slim_f32gemm_launch(g, st);
}

int slim_eigh_launch(float* G, int n, int batch, int top, float* evals, int* info, int info_stride, void* ws,
                     int64_t ws_bytes, hipStream_t st) {
    SLIM_REQUIRE(G && evals && info && ws && batch >= 1 && batch <= 65535, "slim_eigh: bad arguments");
    SLIM_REQUIRE(n >= 4 && n <= SLIM_EIGH_MAX_N && n % 4 == 0, "slim_eigh: n must be a multiple of 4 in [4, 2048]");
    SLIM_REQUIRE(top >= 1 && top <= n, "slim_eigh: bad number of wanted eigenvectors");
    SLIM_REQUIRE(((uintptr_t)G % 16) == 0 && ((uintptr_t)ws % 16) == 0, "slim_eigh: buffers must be 16-byte aligned");
    SLIM_REQUIRE(ws_bytes >= slim_eigh_ws_bytes(n, batch), "slim_eigh: workspace too small");
    EighWs w = eigh_ws((char*)ws, n, batch);
    int rc = tridiagonalize(G, w.Q, w.vbuf, w.pbuf, w.taus, evals, w.e, n, batch, st);
    if (rc) return rc;
    return eigh_tail(evals, w.e, w.Q, n, batch, top, evals, G, info, info_stride, w.tail, tail_ws(nullptr, n, n, batch).total, st);
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

// Second stage for `batch` problems of one size: eigen-decomposition of tridiag(d, e) (d becomes the eigenvalues, ascending;
// e is read) and rows n-top .. n-1 of vecs [n, n] = the eigenvectors of the `top` largest eigenvalues of the original
// matrix (Q times the tridiagonal ones).  workspace: slim_eigh_tail_workspace_bytes(n, top, batch).
extern "C" int64_t slim_eigh_tail_workspace_bytes(int n, int top, int batch) {
    if (n <= 0 || top <= 0 || batch <= 0) return 0;
    return tail_ws(nullptr, n, library_tail() ? n : round_up4(top), batch).total;
}
extern "C" int slim_eigh_tail_f32_batched(float* d, float* e, const float* Q, int n, int batch, int top, float* vecs,
                                          int32_t* info, void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(d && e && Q && vecs && info && workspace, "slim_eigh_tail_f32: null pointer");
    SLIM_REQUIRE(n >= 4 && n <= SLIM_EIGH_MAX_N && n % 4 == 0 && top >= 1 && top <= n && batch >= 1 && batch <= 65535,
                 "slim_eigh_tail_f32: bad sizes");
    SLIM_REQUIRE(((uintptr_t)Q % 16) == 0 && ((uintptr_t)vecs % 16) == 0 && ((uintptr_t)workspace % 16) == 0,
                 "slim_eigh_tail_f32: buffers must be 16-byte aligned");
    return eigh_tail(d, e, Q, n, batch, top, d, vecs, info, 1, (char*)workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int slim_eigh_tail_f32(float* d, float* e, const float* Q, int n, int top, float* vecs, int32_t* info, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    return slim_eigh_tail_f32_batched(d, e, Q, n, 1, top, vecs, info, workspace, workspace_bytes, stream);
}

extern "C" int64_t slim_eigh_workspace_bytes(int n) { return n > 0 ? slim_eigh_ws_bytes(n, 1) : 0; }

// Eigenvalues (ascending) and the `top` leading eigenvectors of the symmetric matrix G [n, n] (fp32, both triangles
// given).  On return row n-1-i of G is the eigenvector of the i-th largest eigenvalue, for i < top; the other rows
// hold no result.  info [1] receives the tridiagonal solver's status word (0 = converged).
extern "C" int slim_eigh_f32(float* G, int n, int top, float* evals, int32_t* info, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    return slim_eigh_launch(G, n, 1, top, evals, info, 1, workspace, workspace_bytes, (hipStream_t)stream);
}
