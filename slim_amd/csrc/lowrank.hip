// Saliency-weighted low-rank fit of the compression error (reference slim/lora.py:172-199,
// which takes a full torch.svd of the N x K error matrix and keeps the leading r triplets).
//
// Here the leading r singular triplets are found by randomized subspace iteration with
// oversampling: every product with the N x K matrix is a bf16 MFMA GEMM (gemm.hip, NT form,
// the tall factors are carried both as X and X^T so no product needs a strided operand), the
// iterate is re-orthonormalised after every product by a Newton-Schulz polynomial iteration that
// is itself only GEMMs (or, selectably, by Cholesky-QR through rocSOLVER potrf + trtri), and the
// final Rayleigh-Ritz step diagonalises the l x l matrix (Q^T E)(Q^T E)^T with rocSOLVER syevd.
//
//   E   [N, K]  error matrix (bf16), Et [K, N] its transpose
//   l = r + oversampling,  q = number of subspace iterations
//   outputs: U [N, r], Ut [r, N] (= lora_right), L [K, r] (= lora_left = V_r S_r), S [r]
#include "common.hpp"
#include "../../include/slim_hip.h"
#include <dlfcn.h>
#include <mutex>
#include <map>
#include <utility>

// ---- rocSOLVER / rocBLAS entry points, resolved at first use ---------------------------------
// (values of the rocBLAS enums are part of its stable C ABI)
namespace {
typedef void* rb_handle;
enum { RB_FILL_UPPER = 121, RB_FILL_LOWER = 122, RB_DIAG_NON_UNIT = 131, RB_EVECT_ORIGINAL = 211 };
typedef int (*fn_create)(rb_handle*);
typedef int (*fn_set_stream)(rb_handle, hipStream_t);
typedef int (*fn_potrf)(rb_handle, int, int, float*, int, int*);
typedef int (*fn_trtri)(rb_handle, int, int, int, float*, int, int*);
typedef int (*fn_syevd)(rb_handle, int, int, int, float*, int, float*, float*, int*);

struct Solver {
    bool ok = false;
    fn_create create = nullptr;
    fn_set_stream set_stream = nullptr;
    fn_potrf potrf = nullptr;
    fn_trtri trtri = nullptr;
    fn_syevd syevd = nullptr;
    std::map<std::pair<int, hipStream_t>, rb_handle> handles;  // per (device, stream): rocSOLVER keeps device workspace in the handle
    std::mutex mu;
};
Solver g_solver;

int solver_init() {
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librocsolver.so.0", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librocsolver.so.0", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        g_solver.create = (fn_create)dlsym(h, "rocblas_create_handle");
        g_solver.set_stream = (fn_set_stream)dlsym(h, "rocblas_set_stream");
        g_solver.potrf = (fn_potrf)dlsym(h, "rocsolver_spotrf");
        g_solver.trtri = (fn_trtri)dlsym(h, "rocsolver_strtri");
        g_solver.syevd = (fn_syevd)dlsym(h, "rocsolver_ssyevd");
        g_solver.ok = g_solver.create && g_solver.set_stream && g_solver.potrf && g_solver.trtri && g_solver.syevd;
    });
    if (!g_solver.ok) {
        slim_set_error("rocSOLVER (librocsolver.so.0) could not be loaded: %s", dlerror() ? dlerror() : "missing symbol");
        return SLIM_ERR_SOLVER;
    }
    return SLIM_OK;
}

int solver_handle(hipStream_t st, rb_handle* out) {
    int rc = solver_init();
    if (rc) return rc;
    int dev = 0;
    SLIM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_solver.mu);
    const auto key = std::make_pair(dev, st);
    auto it = g_solver.handles.find(key);
    if (it == g_solver.handles.end()) {
        rb_handle h = nullptr;
        if (g_solver.create(&h) != 0 || !h) {
            slim_set_error("rocblas_create_handle failed");
            return SLIM_ERR_SOLVER;
        }
        it = g_solver.handles.emplace(key, h).first;
    }
    if (g_solver.set_stream(it->second, st) != 0) {
        slim_set_error("rocblas_set_stream failed");
        return SLIM_ERR_SOLVER;
    }
    *out = it->second;
    return SLIM_OK;
}
}  // namespace

// ---- small kernels -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// standard-normal bf16 test matrix from a counter-based hash (Box-Muller on two 24-bit uniforms)
__global__ void randn_bf16_kernel(bf16_t* __restrict__ out, int64_t n, uint32_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = mix32((uint32_t)i * 2u + 0x9e3779b9u * (seed + 1u));
    const uint32_t b = mix32((uint32_t)i * 2u + 1u + 0x85ebca6bu * (seed + 7u) + (uint32_t)(i >> 31));
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
    out[i] = f32_to_bf16(z);
}

// rocSOLVER works column-major; an l x l buffer holding inv(R) column-major (R upper) reads
// row-major as inv(R)^T, lower triangular.  Emit that as bf16 with the other triangle zeroed.
__global__ void tri_to_bf16_kernel(const float* __restrict__ a, int l, bf16_t* __restrict__ out) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;  // row-major (i, j)
    if (j >= l) return;
    out[(int64_t)i * l + j] = j <= i ? f32_to_bf16(a[(int64_t)i * l + j]) : (bf16_t)0;
}

// add a small multiple of the mean diagonal to the Gram matrix so Cholesky cannot break down on a
// numerically rank-deficient iterate (the perturbed basis is re-orthonormalised again later)
__global__ void ridge_kernel(float* __restrict__ g, int l, float rel) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < l; i += 256) s += g[(int64_t)i * l + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tr = red[0] + red[1] + red[2] + red[3];
    const float add = rel * tr / (float)l;
    for (int i = threadIdx.x; i < l; i += 256) g[(int64_t)i * l + i] += add;
}

// syevd returns eigenvalues ascending and eigenvectors as columns (column-major), i.e. row i of the
// row-major view is eigenvector i.  Emit the top r, largest first, as bf16 [r, l], and S = sqrt(max(w, 0)).
__global__ void top_eigvecs_kernel(const float* __restrict__ vecs, const float* __restrict__ w, int l, int r,
                                   bf16_t* __restrict__ out, float* __restrict__ S) {
    const int j = blockIdx.y;  // output row
    const int src = l - 1 - j;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < l) out[(int64_t)j * l + c] = f32_to_bf16(vecs[(int64_t)src * l + c]);
    if (c == 0 && S) S[j] = sqrtf(fmaxf(w[src], 0.f));
}

// lora_left = bf16(bf16(L) / bf16(s[k])) row-wise (slim/lora.py:193-195)
__global__ void scale_rows_kernel(bf16_t* __restrict__ L, const float* __restrict__ s, int64_t K, int64_t r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * r) return;
    const int64_t k = i / r;
    const float d = round_bf16(s[k]);
    L[i] = f32_to_bf16(bf16_to_f32(L[i]) / d);
}

extern "C" int slim_randn_bf16(void* out, int64_t n, uint32_t seed, void* stream) {
    SLIM_REQUIRE(out && n > 0, "slim_randn_bf16: bad arguments");
    randn_bf16_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>((bf16_t*)out, n, seed);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_scale_rows_bf16(void* L, const float* s, int64_t K, int64_t r, void* stream) {
    SLIM_REQUIRE(L && s && K > 0 && r > 0, "slim_scale_rows_bf16: bad arguments");
    scale_rows_kernel<<<(unsigned)ceil_div64(K * r, 256), 256, 0, (hipStream_t)stream>>>((bf16_t*)L, s, K, r);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- Newton-Schulz helper kernels ------------------------------------------------------------------
// scal[0] = 1 / sqrt(||G||_F): after X <- scal * X every singular value of X is <= 1 (and >= l^(-1/4)
// of the largest), the range in which the polynomial iteration below is stable.
#define FRO_BLOCKS 64
__global__ __launch_bounds__(256) void fro_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)FRO_BLOCKS * 256) {
        const double v = (double)g[i];
        acc += v * v;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void fro_rsqrt_kernel(const double* __restrict__ part, float* __restrict__ scal) {
    const double t = wave_sum_d(part[threadIdx.x]);
    if (threadIdx.x == 0) {
        const double fro = sqrt(t);
        scal[0] = fro > 0.0 ? (float)(1.0 / sqrt(fro)) : 1.0f;
    }
}

// P = c * (a I + b c^2 G + q c^4 G2), bf16 [l, l]; c = scal[0] or 1 when scal == NULL.
__global__ void ns_poly_kernel(const float* __restrict__ G, const float* __restrict__ G2, const float* __restrict__ scal,
                               float a, float b, float q, int l, bf16_t* __restrict__ P) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= l) return;
    const float c = scal ? scal[0] : 1.f;
    const float c2 = c * c;
    const int64_t o = (int64_t)i * l + j;
    float v = b * c2 * G[o] + (q != 0.f ? q * c2 * c2 * G2[o] : 0.f);
    if (i == j) v += a;
    P[o] = f32_to_bf16(c * v);
}

// ---- workspace ----------------------------------------------------------------------------------
#define GRAM_MAX_SPLITS 16
static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

struct Pair {
    bf16_t *X, *Xt;  // [rows, l] and its transpose [l, rows]
};
struct FitWs {
    Pair A[3];      // row-side iterates (rows = N): Y / Q, rotating
    Pair B[3];      // column-side iterates (rows = K): Z / P, rotating
    float* G;       // [l, l] Gram matrix / Cholesky factor / eigenvectors
    float* G2;      // [l, l] square of the Gram matrix
    bf16_t* G16;    // [l, l] bf16 copy of G
    bf16_t* T16;    // [l, l] polynomial / inv(R)^T, later [r, l] leading eigenvectors
    float* evals;   // [l]
    float* ework;   // [l]
    float* scal;    // [4]
    double* fro_part;  // [FRO_BLOCKS] partial sums of squares
    float* slab;    // [GRAM_MAX_SPLITS, l, l] split-K partials of the Gram products
    int* info;      // [SLIM_FIT_INFO_WORDS]
    int64_t total;
};
static FitWs fit_ws(char* base, int64_t N, int64_t K, int64_t l) {
    FitWs w;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + o : nullptr;
        o += al256(bytes);
        return p;
    };
    for (int i = 0; i < 3; ++i) {
        w.A[i].X = (bf16_t*)take(N * l * 2);
        w.A[i].Xt = (bf16_t*)take(N * l * 2);
    }
    for (int i = 0; i < 3; ++i) {
        w.B[i].X = (bf16_t*)take(K * l * 2);
        w.B[i].Xt = (bf16_t*)take(K * l * 2);
    }
    w.G = (float*)take(l * l * 4);
    w.G2 = (float*)take(l * l * 4);
    w.G16 = (bf16_t*)take(l * l * 2);
    w.T16 = (bf16_t*)take(l * l * 2);
    w.evals = (float*)take(l * 4);
    w.ework = (float*)take(l * 4);
    w.scal = (float*)take(16);
    w.fro_part = (double*)take(FRO_BLOCKS * 8);
    w.slab = (float*)take((int64_t)GRAM_MAX_SPLITS * l * l * 4);
    w.info = (int*)take(SLIM_FIT_INFO_WORDS * 4);
    w.total = o;
    return w;
}

extern "C" int64_t slim_lowrank_fit_workspace_bytes(int64_t N, int64_t K, int64_t l) {
    if (N <= 0 || K <= 0 || l <= 0) return 0;
    return fit_ws(nullptr, N, K, l).total;
}

// G (fp32) and optionally G16 (bf16) <- Xt Xt^T, reduction over `rows` split so that tiles * splits covers the chip
static int gram(const bf16_t* Xt, int64_t rows, int l, FitWs& w, bf16_t* G16, void* stream) {
    const int tiles = ((l + 127) / 128) * ((l + 127) / 128);
    int splits = (512 + tiles - 1) / tiles;
    const int nk = (int)((rows + 63) / 64);
    if (splits > nk / 4) splits = nk / 4;      // at least four K-steps per slice
    if (splits > GRAM_MAX_SPLITS) splits = GRAM_MAX_SPLITS;
    if (splits < 1) splits = 1;
    return slim_gemm_nt_bf16_splitk(Xt, rows, Xt, rows, l, l, (int)rows, 1.f, w.G, l, G16, l, splits, w.slab, stream);
}

#define CHK(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

// Cholesky-QR: dst <- orthonormal basis of the column space of src (G = X^T X = R^T R, Q = X inv(R)),
// rocSOLVER potrf + trtri on the l x l factor.
static int cholesky_qr(const Pair& src, const Pair& dst, int64_t rows, int l, FitWs& w, int* info, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    rb_handle h = nullptr;
    CHK(solver_handle(st, &h));
    CHK(gram(src.Xt, rows, l, w, nullptr, stream));
    ridge_kernel<<<1, 256, 0, st>>>(w.G, l, 1e-6f);
    SLIM_LAUNCH_CHECK();
    int solver_tok = g_slim_prof_on ? slim_prof_begin("solver_potrf_trtri", st, 0.0, (double)l * l * l) : -1;
    if (g_solver.potrf(h, RB_FILL_UPPER, l, w.G, l, info) != 0) {
        slim_set_error("rocsolver_spotrf failed");
        return SLIM_ERR_SOLVER;
    }
    if (g_solver.trtri(h, RB_FILL_UPPER, RB_DIAG_NON_UNIT, l, w.G, l, info + 1) != 0) {
        slim_set_error("rocsolver_strtri failed");
        return SLIM_ERR_SOLVER;
    }
    if (solver_tok >= 0) slim_prof_end(solver_tok, st);
    dim3 grid((unsigned)((l + 255) / 256), (unsigned)l);
    tri_to_bf16_kernel<<<grid, 256, 0, st>>>(w.G, l, w.T16);
    SLIM_LAUNCH_CHECK();
    return slim_gemm_nt_bf16(src.X, l, w.T16, l, (int)rows, l, l, 1.f, 0.f, nullptr, 0, dst.X, l, dst.Xt, rows, stream);
}

// Newton-Schulz orthonormalisation on MFMA: X <- X p(X^T X) keeps the column space and drives every
// singular value towards 1.  `nq` quintic steps (3.4445, -4.7750, 2.0315: fast, settles in [0.68, 1.13]),
// then `nc` cubic steps (1.5, -0.5: converges to 1 quadratically).  Each step is the l x l Gram matrix
// (one tall GEMM), its square (one small GEMM), the polynomial (element-wise) and X P (one tall GEMM).
// bufs[cur] holds the input; returns the index of the buffer holding the result.
static int ns_orth(Pair* bufs, int cur, int64_t rows, int l, int nq, int nc, FitWs& w, void* stream, int* out_idx) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 pgrid((unsigned)((l + 255) / 256), (unsigned)l);
    for (int sidx = 0; sidx < nq + nc; ++sidx) {
        const bool quintic = sidx < nq;
        const Pair& x = bufs[cur];
        const int nxt = (cur + 1) % 3;
        CHK(gram(x.Xt, rows, l, w, w.G16, stream));
        if (sidx == 0) {
            fro_partial_kernel<<<FRO_BLOCKS, 256, 0, st>>>(w.G, (int64_t)l * l, w.fro_part);
            SLIM_LAUNCH_CHECK();
            fro_rsqrt_kernel<<<1, 64, 0, st>>>(w.fro_part, w.scal);
            SLIM_LAUNCH_CHECK();
        }
        if (quintic)
            CHK(slim_gemm_nt_bf16(w.G16, l, w.G16, l, l, l, l, 1.f, 0.f, w.G2, l, nullptr, 0, nullptr, 0, stream));
        ns_poly_kernel<<<pgrid, 256, 0, st>>>(w.G, w.G2, sidx == 0 ? w.scal : nullptr, quintic ? 3.4445f : 1.5f,
                                              quintic ? -4.7750f : -0.5f, quintic ? 2.0315f : 0.f, l, w.T16);
        SLIM_LAUNCH_CHECK();
        // P is symmetric, so it serves as its own transpose in the NT product X P
        CHK(slim_gemm_nt_bf16(x.X, l, w.T16, l, (int)rows, l, l, 1.f, 0.f, nullptr, 0, bufs[nxt].X, l, bufs[nxt].Xt, rows,
                              stream));
        cur = nxt;
    }
    *out_idx = cur;
    return SLIM_OK;
}

// one orthonormalisation in the selected mode; cur/out index the rotating buffers
static int orth(Pair* bufs, int cur, int64_t rows, int l, bool final_pass, int flags, int nq, FitWs& w, int* info,
                void* stream, int* out_idx) {
    if (flags & SLIM_FIT_CHOLESKY_QR) {
        const int nxt = (cur + 1) % 3;
        CHK(cholesky_qr(bufs[cur], bufs[nxt], rows, l, w, info, stream));
        *out_idx = nxt;
        return SLIM_OK;
    }
    if (!final_pass) return ns_orth(bufs, cur, rows, l, nq, 0, w, stream, out_idx);
    // last basis before the Rayleigh-Ritz step: it must be orthonormal in every direction, including ones the
    // polynomial iteration has not lifted yet (a numerically rank-deficient iterate), so finish with an exact
    // Cholesky-QR on the Newton-Schulz-conditioned matrix
    int mid = cur;
    CHK(ns_orth(bufs, cur, rows, l, nq, 0, w, stream, &mid));
    const int nxt = (mid + 1) % 3;
    CHK(cholesky_qr(bufs[mid], bufs[nxt], rows, l, w, info, stream));
    *out_idx = nxt;
    return SLIM_OK;
}

// Orthonormal basis (Q [rows, l], Qt [l, rows]) of the column space of X [rows, l] (Xt its transpose).
extern "C" int64_t slim_orthonormalize_workspace_bytes(int64_t rows, int64_t l) {
    if (rows <= 0 || l <= 0) return 0;
    return fit_ws(nullptr, rows, 8, l).total;
}

extern "C" int slim_orthonormalize_bf16(const void* X, const void* Xt, int64_t rows, int l, int flags, int ns_quintic,
                                        int ns_cubic, void* Q, void* Qt, int32_t* info_out, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(X && Xt && Q && Qt && workspace, "slim_orthonormalize_bf16: null pointer");
    SLIM_REQUIRE(rows > 0 && l > 0 && l <= rows && rows % 8 == 0 && l % 8 == 0,
                 "slim_orthonormalize_bf16: rows and l must be multiples of 8 with l <= rows");
    SLIM_REQUIRE(workspace_bytes >= slim_orthonormalize_workspace_bytes(rows, l), "slim_orthonormalize_bf16: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, rows, 8, l);
    const size_t bytes = (size_t)rows * l * sizeof(bf16_t);
    SLIM_HIP(hipMemsetAsync(w.info, 0, SLIM_FIT_INFO_WORDS * sizeof(int), st));
    SLIM_HIP(hipMemcpyAsync(w.A[0].X, X, bytes, hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemcpyAsync(w.A[0].Xt, Xt, bytes, hipMemcpyDeviceToDevice, st));
    int a = 0;
    if (flags & SLIM_FIT_CHOLESKY_QR) {
        CHK(cholesky_qr(w.A[0], w.A[1], rows, l, w, w.info, stream));
        a = 1;
    } else {
        CHK(ns_orth(w.A, 0, rows, l, ns_quintic, ns_cubic, w, stream, &a));
    }
    SLIM_HIP(hipMemcpyAsync(Q, w.A[a].X, bytes, hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemcpyAsync(Qt, w.A[a].Xt, bytes, hipMemcpyDeviceToDevice, st));
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}

extern "C" int slim_lowrank_fit(const void* E, const void* Et, int64_t N, int64_t K, int r, int l, int n_iter,
                                uint32_t seed, int flags, int ns_steps, const float* s, void* U, void* Ut, void* L, float* S,
                                int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(E && Et && U && L && workspace, "slim_lowrank_fit: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && r > 0 && l >= r && n_iter >= 0, "slim_lowrank_fit: bad sizes");
    SLIM_REQUIRE(l <= N && l <= K, "slim_lowrank_fit: subspace dimension l exceeds min(N, K)");
    SLIM_REQUIRE(N % 8 == 0 && K % 8 == 0 && l % 8 == 0 && r % 8 == 0,
                 "slim_lowrank_fit: N, K, l, r must be multiples of 8");
    SLIM_REQUIRE(2 * (n_iter + 1) + 1 <= SLIM_FIT_INFO_WORDS / 2, "slim_lowrank_fit: too many iterations");
    SLIM_REQUIRE(workspace_bytes >= slim_lowrank_fit_workspace_bytes(N, K, l), "slim_lowrank_fit: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    rb_handle h = nullptr;
    CHK(solver_handle(st, &h));
    FitWs w = fit_ws((char*)workspace, N, K, l);
    SLIM_HIP(hipMemsetAsync(w.info, 0, SLIM_FIT_INFO_WORDS * sizeof(int), st));
    const bf16_t* Eb = (const bf16_t*)E;
    const bf16_t* Etb = (const bf16_t*)Et;
    int ninfo = 0, a = 0, b = 0;
    const int nq = ns_steps > 0 ? (ns_steps > 16 ? 16 : ns_steps) : 4;
    // when the subspace is the whole space the basis must be orthonormal to working precision in every
    // direction, however weak: take the exact factorisation
    if (l == N || l == K) flags |= SLIM_FIT_CHOLESKY_QR;

    // P0^T [l, K]: Gaussian test matrix;  Y = E P0;  Q = orth(Y)
    CHK(slim_randn_bf16(w.B[0].Xt, (int64_t)l * K, seed, stream));
    CHK(slim_gemm_nt_bf16(w.B[0].Xt, K, Eb, K, l, (int)N, (int)K, 1.f, 0.f, nullptr, 0, w.A[0].Xt, N, w.A[0].X, l, stream));
    CHK(orth(w.A, 0, N, l, n_iter == 0, flags, nq, w, w.info + ninfo, stream, &a));
    ninfo += 2;
    for (int it = 0; it < n_iter; ++it) {
        // Z = E^T Q ; P = orth(Z)
        CHK(slim_gemm_nt_bf16(w.A[a].Xt, N, Etb, N, l, (int)K, (int)N, 1.f, 0.f, nullptr, 0, w.B[0].Xt, K, w.B[0].X, l, stream));
        CHK(orth(w.B, 0, K, l, false, flags, nq, w, w.info + ninfo, stream, &b));
        ninfo += 2;
        // Y = E P ; Q = orth(Y)
        CHK(slim_gemm_nt_bf16(w.B[b].Xt, K, Eb, K, l, (int)N, (int)K, 1.f, 0.f, nullptr, 0, w.A[0].Xt, N, w.A[0].X, l, stream));
        CHK(orth(w.A, 0, N, l, it == n_iter - 1, flags, nq, w, w.info + ninfo, stream, &a));
        ninfo += 2;
    }
    // Rayleigh-Ritz: B = Q^T E (kept as B^T = Z [K, l]); M = B B^T = W diag(S^2) W^T
    CHK(slim_gemm_nt_bf16(w.A[a].Xt, N, Etb, N, l, (int)K, (int)N, 1.f, 0.f, nullptr, 0, w.B[0].Xt, K, w.B[0].X, l, stream));
    CHK(gram(w.B[0].Xt, K, l, w, nullptr, stream));
    int syevd_tok = g_slim_prof_on ? slim_prof_begin("solver_syevd", st, 0.0, 9.0 * l * l * l) : -1;
    if (g_solver.syevd(h, RB_EVECT_ORIGINAL, RB_FILL_UPPER, l, w.G, l, w.evals, w.ework, w.info + ninfo) != 0) {
        slim_set_error("rocsolver_ssyevd failed");
        return SLIM_ERR_SOLVER;
    }
    ninfo += 1;
    if (syevd_tok >= 0) slim_prof_end(syevd_tok, st);
    {
        dim3 grid((unsigned)((l + 255) / 256), (unsigned)r);
        top_eigvecs_kernel<<<grid, 256, 0, st>>>(w.G, w.evals, l, r, w.T16, S);
        SLIM_LAUNCH_CHECK();
    }
    // L = V_r S_r = B^T W_r  [K, r];  U = Q W_r  [N, r] (+ U^T = lora_right)
    CHK(slim_gemm_nt_bf16(w.B[0].X, l, w.T16, l, (int)K, r, l, 1.f, 0.f, nullptr, 0, L, r, nullptr, 0, stream));
    CHK(slim_gemm_nt_bf16(w.A[a].X, l, w.T16, l, (int)N, r, l, 1.f, 0.f, nullptr, 0, U, r, Ut, N, stream));
    if (s) CHK(slim_scale_rows_bf16(L, s, K, r, stream));
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}
