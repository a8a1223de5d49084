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
#include "gemm.hpp"
#include <cstdlib>
#include "solver.hpp"
#include "eigh.hpp"
#include <dlfcn.h>

#define NS_SCHED_MAX_E 12      // widest Newton-Schulz schedule: singular values within a factor 2^12

// ---- rocSOLVER / rocBLAS entry points, resolved at first use ---------------------------------
Solver g_solver;

static int solver_init() {
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
        g_solver.potrf_sb = (fn_potrf_sb)dlsym(h, "rocsolver_spotrf_strided_batched");
        g_solver.trtri_sb = (fn_trtri_sb)dlsym(h, "rocsolver_strtri_strided_batched");
        g_solver.syevd_sb = (fn_syevd_sb)dlsym(h, "rocsolver_ssyevd_strided_batched");
        g_solver.stedc = (fn_stedc)dlsym(h, "rocsolver_sstedc");
        g_solver.sgemm = (fn_sgemm)dlsym(h, "rocblas_sgemm");
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

#define SCAL_STRIDE 4        // floats of per-problem scalars (FitWs::scal): 0 Newton-Schulz scale, 1-3 see cheb_prepare_kernel
// ---- small kernels -----------------------------------------------------------------------------
// Every kernel below works on a strided batch of identical problems: the last grid dimension in use selects
// the problem, `stride` arguments are in elements.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// standard-normal bf16 test matrix from a counter-based hash (Box-Muller on two 24-bit uniforms); every
// problem of a batch (blockIdx.y) receives the same matrix, so a batched fit equals the single fits
__global__ void randn_bf16_kernel(bf16_t* __restrict__ out, int64_t n, uint32_t seed, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = mix32((uint32_t)i * 2u + 0x9e3779b9u * (seed + 1u));
    const uint32_t b = mix32((uint32_t)i * 2u + 1u + 0x85ebca6bu * (seed + 7u) + (uint32_t)(i >> 31));
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
    out[(int64_t)blockIdx.y * stride + i] = f32_to_bf16(z);
}

// rocSOLVER works column-major; an l x l buffer holding inv(R) column-major (R upper) reads
// row-major as inv(R)^T, lower triangular.  Emit that as bf16 with the other triangle zeroed.
__global__ void tri_to_bf16_kernel(const float* __restrict__ a, int l, bf16_t* __restrict__ out) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;  // row-major (i, j)
    if (j >= l) return;
    const int64_t o = (int64_t)blockIdx.z * l * l + (int64_t)i * l + j;
    out[o] = j <= i ? f32_to_bf16(a[o]) : (bf16_t)0;
}

// add a small multiple of the mean diagonal to the Gram matrix so Cholesky cannot break down on a
// numerically rank-deficient iterate (the perturbed basis is re-orthonormalised again later)
__global__ void ridge_kernel(float* __restrict__ g, int l, float rel) {
    __shared__ float red[4];
    g += (int64_t)blockIdx.x * l * l;
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
// out_div / out_mul (optional): the same rows divided / multiplied by their singular value (a zero singular value gives a
// zero row in out_div).
__global__ void top_eigvecs_kernel(const float* __restrict__ vecs, const float* __restrict__ w, int l, int r,
                                   bf16_t* __restrict__ out, float* __restrict__ S, bf16_t* __restrict__ out_div,
                                   bf16_t* __restrict__ out_mul) {
    const int64_t bz = blockIdx.z;
    const int j = blockIdx.y;  // output row
    const int src = l - 1 - j;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const float sv = sqrtf(fmaxf(w[bz * l + src], 0.f));
    if (c < l) {
        const float v = vecs[bz * l * l + (int64_t)src * l + c];
        const int64_t o = bz * l * l + (int64_t)j * l + c;
        if (out) out[o] = f32_to_bf16(v);
        if (out_div) out_div[o] = f32_to_bf16(sv > 0.f ? v / sv : 0.f);
        if (out_mul) out_mul[o] = f32_to_bf16(v * sv);
    }
    if (c == 0 && S) S[bz * r + j] = sv;
}

// After the fit: the eigenvalues of the Rayleigh-Ritz matrix are the squared singular values of E inside the subspace, so
// their extreme ratio is the conditioning the orthonormalisations had to cope with.  When it exceeds what they were sized
// for, the last status word of the problem receives the level (log2 of the ratio of singular values, rounded up) that would
// have covered it and the caller fits that layer again; otherwise 0.  levels: the encoded conditioning levels of the fit
// (ns_levels_*): the iteration's orthonormalisations after the first were sized for eigenvalue ratios up to 2^later when
// later > 0 (half a level of slack), else for singular value ratios up to 2^cond_e (an application of the Gram matrix
// squares the conditioning, which the fit allows for by itself).  Word SLIM_FIT_INFO_LEVEL_WORD receives the ratio found,
// ceil(4 log2(largest / weakest eigenvalue that matters)): what a caller needs to size the next, similar fit.
// scal (cheb_prepare_kernel's, optional): word SLIM_FIT_INFO_RATIO_WORD receives round(2048 * cut / mean), cut =
// evals[l - r - 1] the largest eigenvalue the fit drops (the eigen-solver's value, or the split's estimate of it), mean the
// mean eigenvalue of the Gram matrix: where the cut sits in the spectrum -- the ratio a Chebyshev-filtered fit of a similar
// matrix is given; and a fit that was filtered with an interval reaching to within 3 % of its cut or above it (wanted
// directions were not amplified) asks to be repeated.
__global__ void cond_check_kernel(const float* __restrict__ evals, int l, int r, int batch, int levels, int* __restrict__ info,
                                  const float* __restrict__ scal = nullptr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int cond_e = levels & 0xff, later = (levels >> 8) & 0xff;
    // ascending eigenvalues; the weakest direction that still matters sits halfway into the oversampling part.  A filtered
    // iteration damps that part of the subspace (its Ritz values there say nothing about the matrix): the largest dropped
    // eigenvalue, just below the rank, stands in -- on the flat spectra of these matrices the two lie within a few per cent
    const float hi = evals[(int64_t)b * l + l - 1];
    float lo = evals[(int64_t)b * l + (l - r) / 2];
    if (scal && scal[b * SCAL_STRIDE + 2] > 0.f && l - r - 1 >= 0) lo = evals[(int64_t)b * l + l - r - 1];
    int need = 0, found = 0;
    if (hi > 0.f) {
        const float c = lo > 0.f ? log2f(hi / lo) : 200.f, lg = 0.5f * c;
        found = (int)ceilf(4.f * c);
        found = found < 0 ? 0 : (found > 255 ? 255 : found);
        if (later > 0 ? c > (float)later + 0.5f : lg > (float)cond_e + 0.5f) {
            need = (int)ceilf(lg);
            if (need > NS_SCHED_MAX_E / 2) need = NS_SCHED_MAX_E / 2;
            if (later > 0) need = need < 1 ? 1 : need;      // the repeat runs without the caller's hint
            else if (need <= cond_e) need = 0;              // already at the widest schedule
        }
    }
    int ratio_q = 0;
    if (scal) {
        const float mean = scal[b * SCAL_STRIDE + 1], beta = scal[b * SCAL_STRIDE + 2];
        const float cut = l - r - 1 >= 0 ? evals[(int64_t)b * l + l - r - 1] : 0.f;
        if (mean > 0.f && cut > 0.f) {
            const float q = 2048.f * cut / mean;
            ratio_q = q > 65535.f ? 65535 : (int)(q + 0.5f);
        }
        if (beta > 0.f && !(beta <= 0.97f * cut) && need < 1) need = 1;
    }
    info[(int64_t)b * SLIM_FIT_INFO_WORDS + SLIM_FIT_INFO_COND_WORD] = need;
    info[(int64_t)b * SLIM_FIT_INFO_WORDS + SLIM_FIT_INFO_LEVEL_WORD] = found;
    info[(int64_t)b * SLIM_FIT_INFO_WORDS + SLIM_FIT_INFO_RATIO_WORD] = ratio_q;
}

// Chebyshev-filtered subspace iteration: set-up per problem (one workgroup each).  mean = trace(C) / Dm (the mean eigenvalue
// of the Gram matrix C: the scale a caller's `ratio` refers to) -> scal[1]; ratio > 0: beta = ratio * mean -> scal[2], the
// upper end of the interval [0, beta] the filter damps; C <- C - (beta / 2) I in place (bf16); scal[3] = -beta^2 / 8.
// With C' = C - (beta / 2) I the degree-2 Chebyshev polynomial of the first kind on [0, beta] applied to a block Y is,
// up to a constant factor, C' (C' Y) - (beta^2 / 8) Y: two products with C' and one addend.
__global__ __launch_bounds__(256) void cheb_prepare_kernel(bf16_t* __restrict__ C16, int64_t sC, int Dm, const float* __restrict__ ratios,
                                                           float* __restrict__ scal) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    bf16_t* C = C16 + (int64_t)b * sC;
    double s = 0.0;
    for (int i = threadIdx.x; i < Dm; i += 256) s += (double)bf16_to_f32(C[(int64_t)i * Dm + i]);
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)Dm);
    const float ratio = ratios ? ratios[b] : 0.f;
    const float beta = ratio > 0.f ? ratio * mean : 0.f;
    if (threadIdx.x == 0) {
        scal[b * SCAL_STRIDE + 1] = mean;
        scal[b * SCAL_STRIDE + 2] = beta;
        scal[b * SCAL_STRIDE + 3] = -0.125f * beta * beta;
    }
    if (beta > 0.f)
        for (int i = threadIdx.x; i < Dm; i += 256) C[(int64_t)i * Dm + i] = f32_to_bf16(bf16_to_f32(C[(int64_t)i * Dm + i]) - 0.5f * beta);
}

// bf16 identity [batch][rows, l] and its transpose [batch][l, rows] (rows == l)
__global__ void identity_pair_kernel(bf16_t* __restrict__ X, bf16_t* __restrict__ Xt, int l) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= l) return;
    const int64_t o = (int64_t)blockIdx.z * l * l + (int64_t)i * l + j;
    X[o] = Xt[o] = i == j ? (bf16_t)0x3f80 : (bf16_t)0;
}

// lora_left = bf16(bf16(L) / bf16(s[k])) row-wise (slim/lora.py:193-195)
__global__ void scale_rows_kernel(bf16_t* __restrict__ L, const float* __restrict__ s, int64_t K, int64_t r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * r) return;
    L += (int64_t)blockIdx.y * K * r;
    s += (int64_t)blockIdx.y * K;
    const int64_t k = i / r;
    const float d = round_bf16(s[k]);
    L[i] = f32_to_bf16(bf16_to_f32(L[i]) / d);
}

// status words of a batched solver call (one per problem) into word `idx` of each problem's info block
__global__ void info_scatter_kernel(const int* __restrict__ tmp, int batch, int* __restrict__ info, int idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) info[b * SLIM_FIT_INFO_WORDS + idx] = tmp[b];
}

extern "C" int slim_randn_bf16(void* out, int64_t n, uint32_t seed, void* stream) {
    SLIM_REQUIRE(out && n > 0, "slim_randn_bf16: bad arguments");
    randn_bf16_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>((bf16_t*)out, n, seed, 0);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

static int scale_rows_batched(void* L, const float* s, int64_t K, int64_t r, int batch, void* stream) {
    dim3 grid((unsigned)ceil_div64(K * r, 256), (unsigned)batch);
    scale_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((bf16_t*)L, s, K, r);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int slim_scale_rows_bf16(void* L, const float* s, int64_t K, int64_t r, void* stream) {
    SLIM_REQUIRE(L && s && K > 0 && r > 0, "slim_scale_rows_bf16: bad arguments");
    return scale_rows_batched(L, s, K, r, 1, stream);
}

// ---- Newton-Schulz helper kernels ------------------------------------------------------------------
// scal[0] = 1 / sqrt(||G||_F): after X <- scal * X every singular value of X is <= 1 (and >= l^(-1/4)
// of the largest), the range in which the polynomial iteration below is stable.
#define FRO_BLOCKS 64
__global__ __launch_bounds__(256) void fro_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    __shared__ double red[4];
    g += (int64_t)blockIdx.y * n;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)FRO_BLOCKS * 256) {
        const double v = (double)g[i];
        acc += v * v;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.y * FRO_BLOCKS + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void fro_rsqrt_kernel(const double* __restrict__ part, float* __restrict__ scal) {
    const double t = wave_sum_d(part[blockIdx.x * FRO_BLOCKS + threadIdx.x]);
    if (threadIdx.x == 0) {
        const double fro = sqrt(t);
        scal[blockIdx.x * SCAL_STRIDE] = fro > 0.0 ? (float)(1.0 / sqrt(fro)) : 1.0f;
    }
}

// P = c * (a I + b c^2 G + q c^4 G2), bf16 [l, l]; c = scal[0] or 1 when scal == NULL.
__global__ void ns_poly_kernel(const float* __restrict__ G, const float* __restrict__ G2, const float* __restrict__ scal,
                               float a, float b, float q, int l, bf16_t* __restrict__ P) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= l) return;
    const float c = scal ? scal[blockIdx.z * SCAL_STRIDE] : 1.f;
    const float c2 = c * c;
    const int64_t o = (int64_t)blockIdx.z * l * l + (int64_t)i * l + j;
    float v = b * c2 * G[o] + (q != 0.f ? q * c2 * c2 * G2[o] : 0.f);
    if (i == j) v += a;
    P[o] = f32_to_bf16(c * v);
}

// A = bf16(c^2 G - shift I), c = scal[0] or 1 when scal == NULL: the operand of a Newton-Schulz step's square
__global__ void shifted_gram_bf16_kernel(const float* __restrict__ G, const float* __restrict__ scal, float shift, int l,
                                         bf16_t* __restrict__ A) {
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= l) return;
    const float c = scal ? scal[blockIdx.z * SCAL_STRIDE] : 1.f;
    const int64_t o = (int64_t)blockIdx.z * l * l + (int64_t)i * l + j;
    const float v = c * c * G[o];
    A[o] = f32_to_bf16(i == j ? v - shift : v);
}

// ---- Newton-Schulz coefficient schedules ---------------------------------------------------------------------------
// A step maps every singular value x of the iterate to a x + b x^3 + c x^5.  For singular values known to lie in
// [2^-e, 1] the odd quintic minimising max |1 - p(x)| on the interval is optimal for the step, and its image interval is
// the next step's domain; composing these greedy steps shrinks the conditioning about fourfold per step at first and
// quadratically at the end (a fixed polynomial gains 3.44x per step and stalls at [0.68, 1.13]).  Row e of the table
// (tools/ns_schedule.py: linear programme on a grid, each interval widened by 5 % at the top for the rounding of bf16
// iterates and polynomials, positivity checked on [0, 1.05]): `enough` steps leave a conditioning of at most 2.35 -- all an intermediate
// basis of the subspace iteration needs -- and all `total` steps leave the singular values within 0.5 % of 1.
#define NS_SCHED_MAX_STEPS 7
struct NsStep {
    float a, b, c;
};
struct NsSchedule {
    int enough, total;
    NsStep step[NS_SCHED_MAX_STEPS];
};
static const NsSchedule kNsSchedule[NS_SCHED_MAX_E + 1] = {
    {0, 0, {}},
    {1, 2, {{2.608928f, -3.018470f, 1.397845f}, {1.830920f, -1.161389f, 0.330565f}}},   // 2^-1: all steps -> [0.9999, 1.0001]
    {1, 2, {{3.715880f, -6.445240f, 3.704211f}, {1.860599f, -1.186662f, 0.328539f}}},   // 2^-2: all steps -> [0.9955, 1.0045]
    {1, 3, {{5.018020f, -10.880343f, 6.832525f}, {2.012485f, -1.322178f, 0.334595f}, {1.832439f, -1.162303f, 0.330073f}}},   // 2^-3: all steps -> [0.9998, 1.0002]
    {2, 3, {{6.187269f, -14.988916f, 9.772026f}, {2.358624f, -1.601237f, 0.355535f}, {1.866572f, -1.192093f, 0.328569f}}},   // 2^-4: all steps -> [0.9944, 1.0056]
    {2, 4, {{7.019487f, -17.945850f, 11.898074f}, {2.835361f, -1.945254f, 0.385560f}, {2.018725f, -1.327552f, 0.334937f}, {1.832662f, -1.162457f, 0.330019f}}},   // 2^-5: all steps -> [0.9998, 1.0002]
    {3, 4, {{7.525912f, -19.753300f, 13.200213f}, {3.287672f, -2.247777f, 0.413545f}, {2.359361f, -1.601798f, 0.355582f}, {1.866700f, -1.192210f, 0.328570f}}},   // 2^-6: all steps -> [0.9943, 1.0057]
    {3, 5, {{7.806895f, -20.758127f, 13.924754f}, {3.617011f, -2.458654f, 0.433512f}, {2.830861f, -1.942149f, 0.385278f}, {2.016502f, -1.325638f, 0.334815f}, {1.832581f, -1.162401f, 0.330038f}}},   // 2^-7: all steps -> [0.9998, 1.0002]
    {4, 5, {{7.955126f, -21.288716f, 14.307496f}, {3.819579f, -2.585373f, 0.445627f}, {3.281919f, -2.244032f, 0.413194f}, {2.353202f, -1.597107f, 0.355194f}, {1.865636f, -1.191241f, 0.328561f}}},   // 2^-8: all steps -> [0.9946, 1.0054]
    {4, 6, {{8.031288f, -21.561458f, 14.504279f}, {3.932568f, -2.655211f, 0.452333f}, {3.612559f, -2.455845f, 0.433245f}, {2.822601f, -1.936443f, 0.384761f}, {2.012468f, -1.322163f, 0.334594f}, {1.832438f, -1.162303f, 0.330073f}}},   // 2^-9: all steps -> [0.9998, 1.0002]
    {5, 6, {{8.069896f, -21.699746f, 14.604063f}, {3.992338f, -2.691930f, 0.455866f}, {3.816823f, -2.583663f, 0.445464f}, {3.274655f, -2.239300f, 0.412749f}, {2.345510f, -1.591237f, 0.354711f}, {1.864334f, -1.190056f, 0.328552f}}},   // 2^-10: all steps -> [0.9948, 1.0052]
    {5, 7, {{8.089333f, -21.769373f, 14.654306f}, {4.023083f, -2.710758f, 0.457679f}, {3.931025f, -2.654260f, 0.452241f}, {3.607575f, -2.452699f, 0.432945f}, {2.813427f, -1.930097f, 0.384187f}, {2.008060f, -1.318357f, 0.334355f}, {1.832287f, -1.162201f, 0.330111f}}},   // 2^-11: all steps -> [0.9998, 1.0002]
    {6, 7, {{8.099085f, -21.804308f, 14.679517f}, {4.038681f, -2.720296f, 0.458598f}, {3.991523f, -2.691430f, 0.455818f}, {3.813886f, -2.581841f, 0.445289f}, {3.266954f, -2.234280f, 0.412278f}, {2.337455f, -1.585078f, 0.354205f}, {1.863001f, -1.188844f, 0.328545f}}},   // 2^-12: all steps -> [0.9951, 1.0049]
};

// ---- workspace ----------------------------------------------------------------------------------
#define GRAM_MAX_SPLITS 16
#define SKETCH_MAX_SPLITS 3
#define GRAM_SPLITS_SHARED_CHIP 2
static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

struct Pair {
    bf16_t *X, *Xt;  // [batch][rows, l] and its transpose [batch][l, rows]
};
struct FitWs {
    int batch;
    int gram_cap;   // upper bound on the split-K slices of the Gram products
    Pair A[3];      // row-side iterates (rows = N): Y / Q, rotating
    Pair B[3];      // column-side iterates (rows = K): Z / P, rotating
    float* G;       // [batch][l, l] Gram matrix / Cholesky factor / eigenvectors
    float* G2;      // [batch][l, l] square of the Gram matrix
    bf16_t* G16;    // [batch][l, l] bf16 copy of G
    bf16_t* T16;    // [batch][l, l] polynomial / inv(R)^T, later [r, l] leading eigenvectors
    float* evals;   // [batch][l]
    float* ework;   // [batch][l]
    float* scal;    // [batch][SCAL_STRIDE]
    double* fro_part;  // [batch][FRO_BLOCKS] partial sums of squares
    float* slab;    // [batch][GRAM_MAX_SPLITS, l, l] split-K partials of the Gram products
    float* sketch_slab;  // [batch][SKETCH_MAX_SPLITS, l, max(N, K)] split-K partials of the sketch products
    bf16_t* C16;    // [batch][Dm, Dm], Dm = min(N, K): the Gram matrix E^T E or E E^T of the Gram-side iteration
    bf16_t* T16b;   // [batch][l, l] second set of scaled eigenvectors
    int* info;      // [batch][SLIM_FIT_INFO_WORDS]
    int* info_tmp;  // [batch] status words of one batched solver call
    int64_t total;
};
static FitWs fit_ws(char* base, int batch, int64_t N, int64_t K, int64_t l) {
    FitWs w;
    w.gram_cap = GRAM_MAX_SPLITS;
    w.batch = batch;
    int64_t o = 0;
    auto take = [&](int64_t bytes) {
        char* p = base ? base + o : nullptr;
        o += al256(bytes * batch);
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
    w.scal = (float*)take(SCAL_STRIDE * 4);
    w.fro_part = (double*)take(FRO_BLOCKS * 8);
    w.slab = (float*)take((int64_t)GRAM_MAX_SPLITS * l * l * 4);
    w.sketch_slab = (float*)take((int64_t)SKETCH_MAX_SPLITS * l * (N > K ? N : K) * 4);
    w.info = (int*)take(SLIM_FIT_INFO_WORDS * 4);
    w.info_tmp = (int*)take(4);
    w.C16 = (bf16_t*)take((N < K ? N : K) * (N < K ? N : K) * 2);
    w.T16b = (bf16_t*)take(l * l * 2);
    w.total = o;
    return w;
}

extern "C" int64_t slim_lowrank_fit_batched_workspace_bytes(int batch, int64_t N, int64_t K, int64_t l) {
    if (batch <= 0 || N <= 0 || K <= 0 || l <= 0) return 0;
    return fit_ws(nullptr, batch, N, K, l).total;
}
extern "C" int64_t slim_lowrank_fit_workspace_bytes(int64_t N, int64_t K, int64_t l) {
    return slim_lowrank_fit_batched_workspace_bytes(1, N, K, l);
}

#define CHK(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

// Strided-batched NT product: C[b] = A[b] (M x Kd, row stride lda) * Bt[b]^T (N x Kd, row stride ldb); a
// stride of 0 shares the operand between the problems.  Outputs as in slim_gemm_nt_bf16.
static int nt(int batch, const bf16_t* A, int64_t lda, int64_t sA, const bf16_t* Bt, int64_t ldb, int64_t sB, int M, int N,
              int Kd, float* C32, bf16_t* C16, int64_t ldc, bf16_t* Ct16, int64_t ldct, void* stream, float* split_slab = nullptr,
              int64_t split_slab_stride = 0) {
    GemmArgs g;
    g.A = A;
    g.lda = lda;
    g.B = Bt;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.Kd = Kd;
    g.C32 = C32;
    g.ldc32 = ldc;
    g.C16 = C16;
    g.ldc16 = ldc;
    g.Ct16 = Ct16;
    g.ldct = ldct;
    g.batch = batch;
    g.sA = sA;
    g.sB = sB;
    g.sC32 = g.sC16 = g.sCt = (int64_t)M * N;
    if (split_slab && !C32) {
        // a product whose tile grid leaves most of the chip idle (the l x 4096 sketches: 160 tiles on 256 CUs, one
        // latency-bound workgroup each) is cut along the reduction; the split depends on the shape only
        const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
        const int nk = (Kd + 63) / 64;
        int splits = tiles >= 256 ? 1 : (int)((512 + tiles - 1) / tiles);
        if (splits > SKETCH_MAX_SPLITS) splits = SKETCH_MAX_SPLITS;
        if (splits > nk / 8) splits = nk / 8;              // at least eight K-steps per slice
        if (splits > 1) {
            g.slab = split_slab;
            g.sSlab = split_slab_stride;
            gemm_plan_splits(g, splits);
        }
    }
    return slim_gemm_launch(g, stream);
}

// G (fp32) and optionally G16 (bf16) <- Xt Xt^T per problem, reduction over `rows` split so that tiles * splits
// covers the chip.  The split count depends on the shape only (not on the batch size), so a batched fit adds its
// partial sums in the same order as a single one.
// diag16 != 0: G16 receives bf16(G - diag16 I); *shifted tells whether that happened (it needs the split-K reduction, which
// short reductions do not have -- the caller then forms the shifted copy itself)
static int gram(const bf16_t* Xt, int64_t rows, int l, FitWs& w, bf16_t* G16, void* stream, float diag16 = 0.f,
                bool* shifted = nullptr) {
    const int tiles = ((l + 127) / 128) * ((l + 127) / 128);
    int splits = (512 + tiles - 1) / tiles;
    const int nk = (int)((rows + 63) / 64);
    if (splits > nk / 4) splits = nk / 4;      // at least four K-steps per slice
    if (splits > GRAM_MAX_SPLITS) splits = GRAM_MAX_SPLITS;
    // (side by side with other fits: slices of about 16 K-steps, at least gram_cap of them)
    // (16 K-steps per slice: 19.75 ms per block against 20.0 with 32 and 20.05 with 8)
    const int cap = w.gram_cap == GRAM_MAX_SPLITS ? GRAM_MAX_SPLITS : (nk / 16 > w.gram_cap ? nk / 16 : w.gram_cap);
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    GemmArgs g;
    g.A = Xt;
    g.lda = rows;
    g.B = Xt;
    g.ldb = rows;
    g.M = l;
    g.N = l;
    g.Kd = (int)rows;
    g.C32 = w.G;
    g.ldc32 = l;
    g.C16 = G16;
    g.ldc16 = l;
    g.batch = w.batch;
    g.sA = g.sB = rows * l;
    g.sC32 = g.sC16 = (int64_t)l * l;
    g.slab = w.slab;
    g.sSlab = (int64_t)GRAM_MAX_SPLITS * l * l;
    g.sym = 1;          // honoured when the product is cut along the reduction (the reduction mirrors the lower tiles)
    gemm_plan_splits(g, splits);
    if (shifted) *shifted = false;
    if (diag16 != 0.f) {
        if (g.splits > 1) {
            g.c16_diag = diag16;
            if (shifted) *shifted = true;
        } else {
            g.C16 = nullptr;
        }
    }
    return slim_gemm_launch(g, stream);
}

static int scatter_info(FitWs& w, int idx, hipStream_t st) {
    info_scatter_kernel<<<(w.batch + 63) / 64, 64, 0, st>>>(w.info_tmp, w.batch, w.info, idx);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Cholesky-QR: dst <- orthonormal basis of the column space of src (G = X^T X = R^T R, Q = X inv(R)),
// rocSOLVER potrf + trtri on the l x l factors (one strided-batched call chain for the whole batch).
static int cholesky_qr(const Pair& src, const Pair& dst, int64_t rows, int l, FitWs& w, int info_idx, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    rb_handle h = nullptr;
    CHK(solver_handle(st, &h));
    CHK(gram(src.Xt, rows, l, w, nullptr, stream));
    ridge_kernel<<<w.batch, 256, 0, st>>>(w.G, l, 1e-6f);
    SLIM_LAUNCH_CHECK();
    int solver_tok = g_slim_prof_on ? slim_prof_begin("solver_potrf_trtri", st, 0.0, (double)l * l * l * w.batch) : -1;
    const int64_t sG = (int64_t)l * l;
    if (w.batch > 1 && g_solver.potrf_sb && g_solver.trtri_sb) {
        if (g_solver.potrf_sb(h, RB_FILL_UPPER, l, w.G, l, sG, w.info_tmp, w.batch) != 0) {
            slim_set_error("rocsolver_spotrf_strided_batched failed");
            return SLIM_ERR_SOLVER;
        }
        CHK(scatter_info(w, info_idx, st));
        if (g_solver.trtri_sb(h, RB_FILL_UPPER, RB_DIAG_NON_UNIT, l, w.G, l, sG, w.info_tmp, w.batch) != 0) {
            slim_set_error("rocsolver_strtri_strided_batched failed");
            return SLIM_ERR_SOLVER;
        }
        CHK(scatter_info(w, info_idx + 1, st));
    } else {
        for (int b = 0; b < w.batch; ++b) {
            int* info = w.info + b * SLIM_FIT_INFO_WORDS + info_idx;
            if (g_solver.potrf(h, RB_FILL_UPPER, l, w.G + b * sG, l, info) != 0) {
                slim_set_error("rocsolver_spotrf failed");
                return SLIM_ERR_SOLVER;
            }
            if (g_solver.trtri(h, RB_FILL_UPPER, RB_DIAG_NON_UNIT, l, w.G + b * sG, l, info + 1) != 0) {
                slim_set_error("rocsolver_strtri failed");
                return SLIM_ERR_SOLVER;
            }
        }
    }
    if (solver_tok >= 0) slim_prof_end(solver_tok, st);
    dim3 grid((unsigned)((l + 255) / 256), (unsigned)l, (unsigned)w.batch);
    tri_to_bf16_kernel<<<grid, 256, 0, st>>>(w.G, l, w.T16);
    SLIM_LAUNCH_CHECK();
    return nt(w.batch, src.X, l, rows * l, w.T16, l, sG, (int)rows, l, l, nullptr, dst.X, l, dst.Xt, rows, stream);
}

// Newton-Schulz orthonormalisation on MFMA: X <- X (a I + b G + c G^2), G = X^T X, keeps the column space and drives
// every singular value towards 1 (coefficients per step: a schedule of kNsSchedule, or fixed ones).  The first step
// works on X / sqrt(||G||_F), whose singular values are at most 1.  Each step is the l x l Gram matrix (one tall GEMM),
// for c != 0 its square with the polynomial evaluated in the product's epilogue (one small GEMM; c == 0: an element-wise
// kernel) and X P (one tall GEMM).  bufs[cur] holds the input; returns the index of the buffer holding the result.
static int ns_orth(Pair* bufs, int cur, int64_t rows, int l, const NsStep* steps, int nsteps, FitWs& w, void* stream,
                   int* out_idx) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 pgrid((unsigned)((l + 255) / 256), (unsigned)l, (unsigned)w.batch);
    const int64_t sG = (int64_t)l * l;
    for (int sidx = 0; sidx < nsteps; ++sidx) {
        const NsStep& c = steps[sidx];
        const Pair& x = bufs[cur];
        const int nxt = (cur + 1) % 3;
        // a + b x + c x^2 = c (x - alpha)^2 + gamma, alpha = -b / (2 c), gamma = a - b^2 / (4 c): the square of the SHIFTED
        // Gram matrix.  Its bf16 rounding error enters the polynomial multiplied by 2 c |x - alpha| (about 1 % for the
        // steepest first steps), where the expanded form has 2 c x (six times as much -- more than the margin the
        // schedules allow, which made them diverge).
        const bool quintic = c.c != 0.f;
        const float alpha = quintic ? -c.b / (2.f * c.c) : 0.f, gamma = quintic ? c.a - c.b * c.b / (4.f * c.c) : 0.f;
        bool shifted = false;
        if (sidx == 0 || !quintic) {
            CHK(gram(x.Xt, rows, l, w, nullptr, stream));
        } else {
            CHK(gram(x.Xt, rows, l, w, w.G16, stream, alpha, &shifted));   // the reduction writes bf16(G - alpha I) directly
        }
        if (sidx == 0) {
            fro_partial_kernel<<<dim3(FRO_BLOCKS, w.batch), 256, 0, st>>>(w.G, sG, w.fro_part);
            SLIM_LAUNCH_CHECK();
            fro_rsqrt_kernel<<<w.batch, 64, 0, st>>>(w.fro_part, w.scal);
            SLIM_LAUNCH_CHECK();
        }
        if (quintic && !shifted) {
            shifted_gram_bf16_kernel<<<pgrid, 256, 0, st>>>(w.G, sidx == 0 ? w.scal : nullptr, alpha, l, w.G16);
            SLIM_LAUNCH_CHECK();
        }
        if (quintic) {
            // P = s (gamma I + c A A), A = bf16(s^2 G - alpha I), evaluated in the product's epilogue
            GemmArgs pg;
            pg.A = w.G16;
            pg.lda = l;
            pg.B = w.G16;
            pg.ldb = l;
            pg.M = pg.N = pg.Kd = l;
            pg.C16 = w.T16;
            pg.ldc16 = l;
            pg.batch = w.batch;
            pg.sA = pg.sB = pg.sC16 = sG;
            pg.poly_G = w.G;            // marks the polynomial epilogue; not read (poly_b == 0)
            pg.poly_sG = sG;
            pg.poly_scal = sidx == 0 ? w.scal : nullptr;
            pg.poly_sScal = SCAL_STRIDE;
            pg.poly_a = gamma;
            pg.poly_b = 0.f;
            pg.poly_q = c.c;
            pg.poly_prescaled = 1;
            CHK(slim_gemm_launch(pg, stream));
        } else {
            ns_poly_kernel<<<pgrid, 256, 0, st>>>(w.G, w.G2, sidx == 0 ? w.scal : nullptr, c.a, c.b, 0.f, l, w.T16);
            SLIM_LAUNCH_CHECK();
        }
        // P is symmetric, so it serves as its own transpose in the NT product X P
        CHK(nt(w.batch, x.X, l, rows * l, w.T16, l, sG, (int)rows, l, l, nullptr, bufs[nxt].X, l, bufs[nxt].Xt, rows, stream));
        cur = nxt;
    }
    *out_idx = cur;
    return SLIM_OK;
}

// Steps of one Newton-Schulz orthonormalisation of the fit.  cond_e: log2 of the conditioning of the iterate (largest /
// smallest singular value inside the subspace) the caller expects.
//   1. A step with the fixed coefficients (3.4445, -4.7750, 2.0315): it takes any singular value of the Frobenius-scaled
//      iterate (largest anywhere in [l^(-1/4), 1]) into [0.65, 1.2] or lifts it 3.44x, i.e. it pins the top of the
//      spectrum next to 1 whatever the scaling lost.
//   2. More such steps while the conditioning that is left exceeds 16 (each gains 3.44 / 1.2 = 2.87x).  The minimax
//      polynomials are not used there on purpose: on a wide interval they take some mid-range singular values down to the
//      bottom of their image interval (equi-oscillation), and a direction held at 1/k of the largest one in bf16 picks up
//      k / 512 of noise -- harmless for k <= 4, ruinous for k in the hundreds.
//   3. The minimax schedule for what is left (kNsSchedule, rescaled for singular values up to 1.25): `enough` steps for an
//      intermediate basis, all of them for the last basis before the Rayleigh-Ritz step.
// The conditioning levels of a fit travel in one int (the ns_steps / cond_e arguments): bits 0-7 = log2 of the ratio of
// singular values of E inside the subspace that the orthonormalisations must cover, as the caller estimates it beforehand
// (0: the default, 9); bits 8-15 (optional) = log2 of the ratio of EIGENVALUES of the Gram matrix inside the subspace as a
// similar fit measured it (status word SLIM_FIT_INFO_LEVEL_WORD): the Gram-side iteration then sizes its
// orthonormalisations after the first for that, the first (a random start) stays on the estimate.
static inline int ns_levels_clamped(int ns_steps) {
    int cond = ns_steps & 0xff, later = (ns_steps >> 8) & 0xff;
    cond = cond > 0 ? (cond > NS_SCHED_MAX_E ? NS_SCHED_MAX_E : cond) : 9;
    later = later > NS_SCHED_MAX_E ? NS_SCHED_MAX_E : later;
    return cond | (later << 8);
}
#define NS_MAX_STEPS 24
static int build_ns_steps(int cond_e, bool final_pass, NsStep* out) {
    const NsStep fixed{3.4445f, -4.7750f, 2.0315f};
    int n = 0;
    out[n++] = fixed;
    double kap = exp2((double)cond_e);
    while (kap > 16.0 && n < NS_MAX_STEPS - NS_SCHED_MAX_STEPS) {
        out[n++] = fixed;
        kap /= 2.87;
    }
    int es = (int)ceil(log2(kap));
    es = es < 1 ? 1 : (es > 4 ? 4 : es);
    const NsSchedule& sc = kNsSchedule[es];
    const int cnt = final_pass ? sc.total : sc.enough;
    // the schedule's first polynomial expects singular values up to 1: it is applied to x / 1.25 (the fixed steps leave
    // them below 1.2); the later ones take the image interval of their predecessor as it is
    const float h = 1.25f, h3 = h * h * h, h5 = h3 * h * h;
    for (int i = 0; i < cnt; ++i)
        out[n++] = i == 0 ? NsStep{sc.step[i].a / h, sc.step[i].b / h3, sc.step[i].c / h5} : sc.step[i];
    return n;
}

// one orthonormalisation in the selected mode; cur/out index the rotating buffers
static int orth(Pair* bufs, int cur, int64_t rows, int l, bool final_pass, int flags, int cond_e, FitWs& w, int info_idx,
                void* stream, int* out_idx) {
    NsStep steps[NS_MAX_STEPS];
    if (flags & SLIM_FIT_CHOLESKY_QR) {
        const int nxt = (cur + 1) % 3;
        CHK(cholesky_qr(bufs[cur], bufs[nxt], rows, l, w, info_idx, stream));
        *out_idx = nxt;
        return SLIM_OK;
    }
    // An intermediate basis only has to be well conditioned (the next product with the matrix spreads its singular
    // values again); the last one before the Rayleigh-Ritz step runs the schedule to its end, which leaves the singular
    // values within 0.5 % of 1 -- the accuracy a bf16 basis can hold.  On small layers, where l is a large part of
    // min(N, K), the weakest directions of the last iterate are far below the conditioning the caller's estimate covers:
    // the last pass allows for a factor 32 more there, and for a factor 4 when l is at most a quarter of the rows.
    const int extra = 4 * (int64_t)l > rows ? 5 : 2;
    const int e = final_pass ? (cond_e + extra > NS_SCHED_MAX_E ? NS_SCHED_MAX_E : cond_e + extra) : cond_e;
    return ns_orth(bufs, cur, rows, l, steps, build_ns_steps(e, final_pass, steps), w, stream, out_idx);
}

extern "C" int64_t slim_orthonormalize_workspace_bytes(int64_t rows, int64_t l) {
    if (rows <= 0 || l <= 0) return 0;
    return fit_ws(nullptr, 1, rows, 8, l).total;
}

extern "C" int slim_orthonormalize_bf16(const void* X, const void* Xt, int64_t rows, int l, int flags, int ns_quintic,
                                        int ns_cubic, void* Q, void* Qt, int32_t* info_out, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(X && Xt && Q && Qt && workspace, "slim_orthonormalize_bf16: null pointer");
    SLIM_REQUIRE(rows > 0 && l > 0 && l <= rows && rows % 8 == 0 && l % 8 == 0,
                 "slim_orthonormalize_bf16: rows and l must be multiples of 8 with l <= rows");
    SLIM_REQUIRE(workspace_bytes >= slim_orthonormalize_workspace_bytes(rows, l), "slim_orthonormalize_bf16: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, 1, rows, 8, l);
    const size_t bytes = (size_t)rows * l * sizeof(bf16_t);
    SLIM_HIP(hipMemsetAsync(w.info, 0, SLIM_FIT_INFO_WORDS * sizeof(int), st));
    SLIM_HIP(hipMemcpyAsync(w.A[0].X, X, bytes, hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemcpyAsync(w.A[0].Xt, Xt, bytes, hipMemcpyDeviceToDevice, st));
    int a = 0;
    if (flags & SLIM_FIT_CHOLESKY_QR) {
        CHK(cholesky_qr(w.A[0], w.A[1], rows, l, w, 0, stream));
        a = 1;
    } else {
        // fixed coefficients: quintic steps (fast, settle in [0.68, 1.13]), then cubic ones (quadratic convergence to 1)
        SLIM_REQUIRE(ns_quintic >= 0 && ns_cubic >= 0 && ns_quintic + ns_cubic <= 32, "slim_orthonormalize_bf16: bad step counts");
        NsStep steps[32];
        for (int i = 0; i < ns_quintic + ns_cubic; ++i)
            steps[i] = i < ns_quintic ? NsStep{3.4445f, -4.7750f, 2.0315f} : NsStep{1.5f, -0.5f, 0.f};
        CHK(ns_orth(w.A, 0, rows, l, steps, ns_quintic + ns_cubic, w, stream, &a));
    }
    SLIM_HIP(hipMemcpyAsync(Q, w.A[a].X, bytes, hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemcpyAsync(Qt, w.A[a].Xt, bytes, hipMemcpyDeviceToDevice, st));
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}

// `batch` fits of identical shape in lockstep: E [batch][N, K], Et [batch][K, N], s [batch][K] (or NULL),
// outputs U [batch][N, r], Ut [batch][r, N], L [batch][K, r], S [batch][r], info_out [batch][SLIM_FIT_INFO_WORDS].
// Every launch covers the whole batch (gridDim.z / strided-batched solver calls), so the launch count and the
// length of the dependent chain are those of ONE fit.  Each problem's result is bit-identical to a single fit
// wherever the solver's batched and single routines agree.
// phase FIT_ALL: the whole fit.  FIT_FRONT: up to the Rayleigh-Ritz matrix M (copied to M_ext), *state receives what
// FIT_BACK needs to resume.  FIT_BACK: from the eigenvectors (rows of ext_vecs, ascending, as slim_eigh_f32 leaves them),
// eigenvalues ext_evals and solver status ext_info of M to the factors.  Front and back share the workspace.
enum { FIT_ALL = 0, FIT_FRONT = 1, FIT_BACK = 2 };
#define FIT_STATE_GRAM (1 << 30)      // marks a state word written by the Gram-side front half
static int gram_fit_impl(int phase, const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l, int n_iter,
                         uint32_t seed, int flags, int cond_e, const float* s, void* U, void* Ut, void* L, float* S,
                         int32_t* info_out, FitWs& w, void* stream, float* M_ext, int* state, const float* ext_vecs,
                         const float* ext_evals, const int* ext_info);
static int fit_impl(int phase, const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l, int n_iter,
                    uint32_t seed, int flags, int ns_steps, const float* s, void* U, void* Ut, void* L, float* S,
                    int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream, float* M_ext, int* state,
                    const float* ext_vecs, const float* ext_evals, const int* ext_info) {
    SLIM_REQUIRE(workspace && (phase == FIT_BACK || (E && Et)) && (phase == FIT_FRONT || (U && L)), "slim_lowrank_fit: null pointer");
    SLIM_REQUIRE(batch >= 1 && batch <= 1024, "slim_lowrank_fit: batch must be in [1, 1024]");
    SLIM_REQUIRE(N > 0 && K > 0 && r > 0 && l >= r && n_iter >= 0, "slim_lowrank_fit: bad sizes");
    SLIM_REQUIRE(l <= N && l <= K, "slim_lowrank_fit: subspace dimension l exceeds min(N, K)");
    SLIM_REQUIRE(N % 8 == 0 && K % 8 == 0 && l % 8 == 0 && r % 8 == 0,
                 "slim_lowrank_fit: N, K, l, r must be multiples of 8");
    SLIM_REQUIRE(2 * (n_iter + 1) + 1 <= SLIM_FIT_INFO_WORDS / 2, "slim_lowrank_fit: too many iterations");
    SLIM_REQUIRE(workspace_bytes >= slim_lowrank_fit_batched_workspace_bytes(batch, N, K, l), "slim_lowrank_fit: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, batch, N, K, l);
    if (phase != FIT_BACK) SLIM_HIP(hipMemsetAsync(w.info, 0, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), st));
    if (phase == FIT_BACK ? (*state & FIT_STATE_GRAM) != 0 : !(flags & SLIM_FIT_CHOLESKY_QR)) {
        int inner = state ? (*state & ~FIT_STATE_GRAM) : 0;
        const int cond = ns_levels_clamped(ns_steps);
        int rc = gram_fit_impl(phase, E, Et, batch, N, K, r, l, n_iter, seed, flags, cond, s, U, Ut, L, S, info_out, w, stream, M_ext,
                               &inner, ext_vecs, ext_evals, ext_info);
        if (rc == SLIM_OK && phase == FIT_FRONT) *state = inner | FIT_STATE_GRAM;
        return rc;
    }
    const bf16_t* Eb = (const bf16_t*)E;
    const bf16_t* Etb = (const bf16_t*)Et;
    const int64_t sE = N * K, sYn = N * l, sZk = K * l, sG = (int64_t)l * l;
    const int64_t sSk = (int64_t)SKETCH_MAX_SPLITS * l * (N > K ? N : K);
    // A caller that runs several fits side by side fills the chip with them: cutting products along the reduction then
    // only adds slab traffic and workgroups (measured on a decoder block's seven fits: 16 -> 2 Gram slices, 47.8 -> 42.5 ms)
    float* sk_slab = (flags & SLIM_FIT_NO_SKETCH_SPLIT) ? nullptr : w.sketch_slab;
    if (flags & SLIM_FIT_NO_SKETCH_SPLIT) w.gram_cap = GRAM_SPLITS_SHARED_CHIP;
    int ninfo = 0, a = 0, b = 0;
    // ns_steps: log2 of the conditioning the orthonormalisations must cover (0: 2^9)
    const int nq = ns_levels_clamped(ns_steps) & 0xff;
    // when the subspace is the whole space the basis must be orthonormal to working precision in every
    // direction, however weak: take the exact factorisation
    if (l == N || l == K) flags |= SLIM_FIT_CHOLESKY_QR;

    // Y^T = P0^T E^T  (A = P0^T [l, K], Bt = E [N, K]) -> Y^T [l, N] and Y [N, l]
    auto sketch_rows = [&](const bf16_t* Pt) {
        return nt(batch, Pt, K, sZk, Eb, K, sE, l, (int)N, (int)K, nullptr, w.A[0].Xt, N, w.A[0].X, l, stream, sk_slab, sSk);
    };
    // Z^T = Q^T E  (A = Q^T [l, N], Bt = E^T [K, N]) -> Z^T [l, K] and Z [K, l]
    auto sketch_cols = [&](const bf16_t* Qt) {
        return nt(batch, Qt, N, sYn, Etb, N, sE, l, (int)K, (int)N, nullptr, w.B[0].Xt, K, w.B[0].X, l, stream, sk_slab, sSk);
    };

    if (phase != FIT_BACK) {
    // P0^T [l, K]: Gaussian test matrix;  Y = E P0;  Q = orth(Y)
    {
        const int64_t n = (int64_t)l * K;
        randn_bf16_kernel<<<dim3((unsigned)ceil_div64(n, 256), (unsigned)batch), 256, 0, st>>>(w.B[0].Xt, n, seed, sZk);
        SLIM_LAUNCH_CHECK();
    }
    CHK(sketch_rows(w.B[0].Xt));
    CHK(orth(w.A, 0, N, l, n_iter == 0, flags, nq, w, ninfo, stream, &a));
    ninfo += 2;
    for (int it = 0; it < n_iter; ++it) {
        // Z = E^T Q ; P = orth(Z)
        CHK(sketch_cols(w.A[a].Xt));
        CHK(orth(w.B, 0, K, l, false, flags, nq, w, ninfo, stream, &b));
        ninfo += 2;
        // Y = E P ; Q = orth(Y)
        CHK(sketch_rows(w.B[b].Xt));
        CHK(orth(w.A, 0, N, l, it == n_iter - 1, flags, nq, w, ninfo, stream, &a));
        ninfo += 2;
    }
    // Rayleigh-Ritz: B = Q^T E (kept as B^T = Z [K, l]); M = B B^T = W diag(S^2) W^T
    CHK(sketch_cols(w.A[a].Xt));
    CHK(gram(w.B[0].Xt, K, l, w, nullptr, stream));
    if (phase == FIT_FRONT) {
        SLIM_REQUIRE(M_ext && state, "slim_lowrank_fit_front: bad arguments");
        SLIM_HIP(hipMemcpyAsync(M_ext, w.G, (size_t)batch * sG * sizeof(float), hipMemcpyDeviceToDevice, st));
        *state = a | (ninfo << 8);
        return SLIM_OK;
    }
    } else {
        SLIM_REQUIRE(state && ext_vecs && ext_evals, "slim_lowrank_fit_back: bad arguments");
        a = *state & 0xff;
        ninfo = *state >> 8;
        SLIM_REQUIRE(a >= 0 && a < 3 && ninfo >= 0 && ninfo < SLIM_FIT_INFO_WORDS, "slim_lowrank_fit_back: bad state");
        if (ext_info) {
            info_scatter_kernel<<<(batch + 63) / 64, 64, 0, st>>>(ext_info, batch, w.info, ninfo);
            SLIM_LAUNCH_CHECK();
        }
    }
    if (phase == FIT_BACK) {
        // the eigenproblem was solved outside (shared with other fits)
    } else if (!(flags & SLIM_FIT_LIBRARY_EIGH) && slim_eigh_supported(l) &&
        slim_eigh_ws_bytes(l, batch) <= (int64_t)GRAM_MAX_SPLITS * l * l * 4 * batch) {
        // in-library solver (one launch per column); the split-K slab is free at this point and serves as its workspace
        CHK(slim_eigh_launch(w.G, l, batch, r, w.evals, w.info + ninfo, SLIM_FIT_INFO_WORDS, w.slab,
                             (int64_t)GRAM_MAX_SPLITS * l * l * 4 * batch, st));
    } else {
        rb_handle h = nullptr;
        CHK(solver_handle(st, &h));
        int syevd_tok = g_slim_prof_on ? slim_prof_begin("solver_syevd", st, 0.0, 9.0 * l * l * l * batch) : -1;
        if (batch > 1 && g_solver.syevd_sb) {
            if (g_solver.syevd_sb(h, RB_EVECT_ORIGINAL, RB_FILL_UPPER, l, w.G, l, sG, w.evals, l, w.ework, l, w.info_tmp, batch) != 0) {
                slim_set_error("rocsolver_ssyevd_strided_batched failed");
                return SLIM_ERR_SOLVER;
            }
            CHK(scatter_info(w, ninfo, st));
        } else {
            for (int bi = 0; bi < batch; ++bi)
                if (g_solver.syevd(h, RB_EVECT_ORIGINAL, RB_FILL_UPPER, l, w.G + bi * sG, l, w.evals + (int64_t)bi * l,
                                   w.ework + (int64_t)bi * l, w.info + bi * SLIM_FIT_INFO_WORDS + ninfo) != 0) {
                    slim_set_error("rocsolver_ssyevd failed");
                    return SLIM_ERR_SOLVER;
                }
        }
        if (syevd_tok >= 0) slim_prof_end(syevd_tok, st);
    }
    ninfo += 1;
    {
        dim3 grid((unsigned)((l + 255) / 256), (unsigned)r, (unsigned)batch);
        top_eigvecs_kernel<<<grid, 256, 0, st>>>(phase == FIT_BACK ? ext_vecs : w.G, phase == FIT_BACK ? ext_evals : w.evals, l, r,
                                                 w.T16, S, nullptr, nullptr);
        SLIM_LAUNCH_CHECK();
    }
    // L = V_r S_r = B^T W_r  [K, r];  U = Q W_r  [N, r] (+ U^T = lora_right)
    CHK(nt(batch, w.B[0].X, l, sZk, w.T16, l, sG, (int)K, r, l, nullptr, (bf16_t*)L, r, nullptr, 0, stream));
    CHK(nt(batch, w.A[a].X, l, sYn, w.T16, l, sG, (int)N, r, l, nullptr, (bf16_t*)U, r, (bf16_t*)Ut, N, stream));
    if (s) CHK(scale_rows_batched(L, s, K, r, batch, stream));
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}


// ---- Chebyshev filter request ---------------------------------------------------------------------------------------------
// slim_lowrank_fit_set_filter leaves, for the NEXT front-half call on the calling thread (it is consumed there), one ratio
// per problem and the conditioning (log2) the orthonormalisations of the filtered rounds are sized for.
#define FILTER_MAX 1024
static thread_local float tl_filter_ratio[FILTER_MAX];
static thread_local int tl_filter_count = 0, tl_filter_bits = 0;
extern "C" int slim_lowrank_fit_set_filter(const float* ratios, int count, int ns_bits) {
    SLIM_REQUIRE(count >= 0 && count <= FILTER_MAX && (count == 0 || ratios) && ns_bits >= 0 && ns_bits <= NS_SCHED_MAX_E,
                 "slim_lowrank_fit_set_filter: bad arguments");
    for (int i = 0; i < count; ++i) {
        SLIM_REQUIRE(ratios[i] >= 0.f && ratios[i] < 64.f, "slim_lowrank_fit_set_filter: ratios must lie in [0, 64)");
        tl_filter_ratio[i] = ratios[i];
    }
    tl_filter_count = count;
    tl_filter_bits = ns_bits;
    return SLIM_OK;
}
struct FilterReq {
    int count = 0, bits = 0;
    float ratio[FILTER_MAX];
};
// (takes the request of this thread and clears it; `on` only when every one of `batch` problems has a positive ratio)
static bool take_filter(int batch, float* ratios_out, int* bits_out) {
    const int count = tl_filter_count;
    tl_filter_count = 0;
    if (count != batch) return false;
    for (int i = 0; i < batch; ++i) {
        if (!(tl_filter_ratio[i] > 0.f)) return false;
        ratios_out[i] = tl_filter_ratio[i];
    }
    *bits_out = tl_filter_bits;
    return true;
}

// ---- the fit by subspace iteration on the Gram matrix ------------------------------------------------------------------
// Subspace iteration on the symmetric matrix C [batch][Dm, Dm] (bf16): n_iter + 1 applications of C to an l-column
// iterate that starts as a Gaussian matrix, each followed by an orthonormalisation (the last one to the accuracy of a
// final basis).  Pb: three rotating buffer pairs of Dm rows; *a_out = index of the pair that holds the basis P.
//
// filter_ratios (host array of `batch` positive ratios, or NULL): the Chebyshev-filtered iteration instead.  With beta =
// ratio * trace(C) / Dm per problem (cheb_prepare_kernel), every round applies the degree-2 Chebyshev polynomial on
// [0, beta] -- eigenvalues below beta, the unwanted end of the spectrum, stay bounded while the wanted ones grow like
// cosh(2 acosh(2 lambda / beta - 1)) -- as two products with C - (beta / 2) I and one addend, then orthonormalises:
// (n_iter + 1) / 2 rounds, orthonormalisations sized for 2^filter_bits (two levels more for the first, whose start is random).
// Per application of C the gap between a wanted eigenvalue and beta closes several times faster than under the plain
// iteration (flat spectra: lambda_r / lambda_(l+1) = 1.07 per application against 1.6 per degree), and a round has ONE
// orthonormalisation for two applications.  C is modified in place (its diagonal is shifted).
static int gram_iterate(Pair* Pb, bf16_t* C16, int64_t sC, int batch, int64_t Dm, int l, int n_iter, uint32_t seed,
                        int flags, int cond_e, FitWs& w, float* sk_slab, int64_t sSk, void* stream, int* a_out, int* ninfo,
                        const float* filter_ratios = nullptr, int filter_bits = 0) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t sP = Dm * l;
    {
        // the mean eigenvalue of C (reported with the fit's status words), and the filter's set-up
        float* dev_ratios = nullptr;
        if (filter_ratios) {
            dev_ratios = w.evals;                                  // [batch][l] floats, free until the eigen-stage
            SLIM_HIP(hipMemcpyAsync(dev_ratios, filter_ratios, (size_t)batch * sizeof(float), hipMemcpyHostToDevice, st));
        }
        cheb_prepare_kernel<<<batch, 256, 0, st>>>(C16, sC, (int)Dm, dev_ratios, w.scal);
        SLIM_LAUNCH_CHECK();
    }
    if (filter_ratios) {
        const int64_t n = (int64_t)l * Dm;
        randn_bf16_kernel<<<dim3((unsigned)ceil_div64(n, 256), (unsigned)batch), 256, 0, st>>>(Pb[1].Xt, n, seed, sP);
        SLIM_LAUNCH_CHECK();
        const int rounds = (n_iter + 1) / 2 > 0 ? (n_iter + 1) / 2 : 1;
        int src = 1, a = 1;
        for (int rd = 0; rd < rounds; ++rd) {
            const int d1 = (src + 1) % 3, d2 = (src + 2) % 3;
            // W1^T = P^T C' ;  Y^T = W1^T C' - (beta^2 / 8) P^T  (both orientations of Y for the orthonormalisation)
            CHK(nt(batch, Pb[src].Xt, Dm, sP, C16, Dm, sC, l, (int)Dm, (int)Dm, nullptr, Pb[d1].Xt, Dm, nullptr, 0, stream));
            {
                GemmArgs g;
                g.A = Pb[d1].Xt, g.lda = Dm, g.B = C16, g.ldb = Dm;
                g.M = l, g.N = (int)Dm, g.Kd = (int)Dm;
                g.C16 = Pb[d2].Xt, g.ldc16 = Dm, g.Ct16 = Pb[d2].X, g.ldct = l;
                g.batch = batch;
                g.sA = sP, g.sB = sC, g.sC16 = g.sCt = sP;
                g.add16 = Pb[src].Xt, g.ldadd = Dm, g.sAdd = sP;
                g.add_coef = w.scal + 3, g.sAddCoef = SCAL_STRIDE;
                CHK(slim_gemm_launch(g, stream));
            }
            int bits = filter_bits + (rd == 0 ? 2 : 0);
            bits = bits > NS_SCHED_MAX_E ? NS_SCHED_MAX_E : bits;
            CHK(orth(Pb, d2, Dm, l, rd + 1 == rounds, flags, bits, w, *ninfo, stream, &a));
            *ninfo += 2;
            src = a;
        }
        *a_out = a;
        return SLIM_OK;
    }
    // cond_e: encoded levels (ns_levels_clamped).  An application of C squares the conditioning inside the subspace
    const int est = 2 * (cond_e & 0xff) > NS_SCHED_MAX_E ? NS_SCHED_MAX_E : 2 * (cond_e & 0xff);
    const int later = (cond_e >> 8) & 0xff ? (cond_e >> 8) & 0xff : est;
    const int64_t n = (int64_t)l * Dm;
    randn_bf16_kernel<<<dim3((unsigned)ceil_div64(n, 256), (unsigned)batch), 256, 0, st>>>(Pb[1].Xt, n, seed, sP);
    SLIM_LAUNCH_CHECK();
    int src = 1, a = 1;
    for (int it = 0; it <= n_iter; ++it) {
        // Y^T = P^T C (C symmetric) -> Y^T [l, Dm] and Y [Dm, l] in the buffer after the source, then Q = orth(Y)
        const int dst = (src + 1) % 3;
        CHK(nt(batch, Pb[src].Xt, Dm, sP, C16, Dm, sC, l, (int)Dm, (int)Dm, nullptr, Pb[dst].Xt, Dm, Pb[dst].X, l, stream, sk_slab,
               sSk));
        CHK(orth(Pb, dst, Dm, l, it == n_iter, flags, it == 0 ? est : later, w, *ninfo, stream, &a));
        *ninfo += 2;
        src = a;
    }
    *a_out = a;
    return SLIM_OK;
}


// With Dm = min(N, K) the small side of E, C = E^T E (K <= N) or E E^T (N < K) is formed once (a symmetric product: only
// the tiles on and above the diagonal are computed) and the subspace iteration runs on C: one Dm x Dm x l product and one
// orthonormalisation of a Dm x l iterate per iteration, against two products with the N x K matrix and two
// orthonormalisations (one of them on the long side) for the alternating iteration above.  An application of C squares
// the conditioning inside the subspace, so its orthonormalisation is sized for 2^(2 cond_e).  C is kept in bf16: its
// rounding perturbs the eigenvectors by angles of the order 2^-9 x (entry size / eigenvalue gap), far below what the
// 1e-3 budget of the reconstruction error notices (measured: identical excess with an fp32 C).  The Rayleigh-Ritz step
// uses the matrix itself: Z = E P (or E^T P) for the orthonormal basis P of the right (left) subspace, M = Z^T Z.
//   K <= N:  P [K, l] right basis;  Z = E P [N, l];  U = Z W S^-1,  L = P W S
//   N <  K:  P [N, l] left basis;   Z = E^T P [K, l];  U = P W,     L = Z W
static int gram_fit_impl(int phase, const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l, int n_iter,
                         uint32_t seed, int flags, int cond_e, const float* s, void* U, void* Ut, void* L, float* S,
                         int32_t* info_out, FitWs& w, void* stream, float* M_ext, int* state, const float* ext_vecs,
                         const float* ext_evals, const int* ext_info) {
    hipStream_t st = (hipStream_t)stream;
    const bool right = K <= N;
    const int64_t Dm = right ? K : N, Dx = right ? N : K;
    const bf16_t* Eb = (const bf16_t*)E;
    const bf16_t* Etb = (const bf16_t*)Et;
    const int64_t sE = N * K, sP = Dm * l, sZ = Dx * l, sG = (int64_t)l * l, sC = Dm * Dm;
    const int64_t sSk = (int64_t)SKETCH_MAX_SPLITS * l * (N > K ? N : K);
    float* sk_slab = (flags & SLIM_FIT_NO_SKETCH_SPLIT) ? nullptr : w.sketch_slab;
    if (flags & SLIM_FIT_NO_SKETCH_SPLIT) w.gram_cap = GRAM_SPLITS_SHARED_CHIP;
    Pair* Pb = right ? w.B : w.A;        // rotating buffers of the iterate (Dm rows)
    Pair& Zb = right ? w.A[0] : w.B[0];  // the projection (Dx rows)
    int ninfo = 0, a = 0;
    if (phase != FIT_BACK) {
        if (l == Dm) {
            // the subspace is the whole space: any orthonormal basis will do, and the identity is exact
            identity_pair_kernel<<<dim3((unsigned)((l + 255) / 256), (unsigned)l, (unsigned)batch), 256, 0, st>>>(Pb[0].X, Pb[0].Xt, l);
            SLIM_LAUNCH_CHECK();
            a = 0;
        } else {
            {   // C = E^T E (operand E^T [K, N]) or E E^T (operand E [N, K]), bf16, upper tiles mirrored
                GemmArgs g;
                g.A = g.B = right ? Etb : Eb;
                g.lda = g.ldb = Dx;
                g.M = g.N = (int)Dm;
                g.Kd = (int)Dx;
                g.C16 = g.Ct16 = w.C16;
                g.ldc16 = g.ldct = Dm;
                g.batch = batch;
                g.sA = g.sB = sE;
                g.sC16 = g.sCt = sC;
                g.sym = 1;
                CHK(slim_gemm_launch(g, stream));
            }
            float fr[FILTER_MAX];
            int fbits = 0;
            const bool filt = phase == FIT_FRONT && take_filter(batch, fr, &fbits);
            CHK(gram_iterate(Pb, w.C16, sC, batch, Dm, l, n_iter, seed, flags, cond_e, w, sk_slab, sSk, stream, &a, &ninfo,
                             filt ? fr : nullptr, fbits));
        }
        // Z^T = P^T E^T (operand E [N, K]: Z = E P) or P^T E (operand E^T: Z = E^T P) -> Z^T [l, Dx] and Z [Dx, l]
        CHK(nt(batch, Pb[a].Xt, Dm, sP, right ? Eb : Etb, Dm, sE, l, (int)Dx, (int)Dm, nullptr, Zb.Xt, Dx, Zb.X, l, stream, sk_slab, sSk));
        CHK(gram(Zb.Xt, Dx, l, w, nullptr, stream));
        if (phase == FIT_FRONT) {
            SLIM_REQUIRE(M_ext && state, "slim_lowrank_fit_front: bad arguments");
            SLIM_HIP(hipMemcpyAsync(M_ext, w.G, (size_t)batch * sG * sizeof(float), hipMemcpyDeviceToDevice, st));
            *state = a | (ninfo << 8) | ((cond_e & 0xff) << 16) | (((cond_e >> 8) & 0x3f) << 24);
            return SLIM_OK;
        }
    } else {
        SLIM_REQUIRE(state && ext_vecs && ext_evals, "slim_lowrank_fit_back: bad arguments");
        a = *state & 0xff;
        ninfo = (*state >> 8) & 0xff;
        cond_e = ((*state >> 16) & 0xff) | (((*state >> 24) & 0x3f) << 8);
        SLIM_REQUIRE(a >= 0 && a < 3 && ninfo >= 0 && ninfo < SLIM_FIT_INFO_WORDS - 2, "slim_lowrank_fit_back: bad state");
        if (ext_info) {
            info_scatter_kernel<<<(batch + 63) / 64, 64, 0, st>>>(ext_info, batch, w.info, ninfo);
            SLIM_LAUNCH_CHECK();
        }
    }
    if (phase == FIT_ALL) {
        if (!(flags & SLIM_FIT_LIBRARY_EIGH) && slim_eigh_supported(l) &&
            slim_eigh_ws_bytes(l, batch) <= (int64_t)GRAM_MAX_SPLITS * l * l * 4 * batch) {
            CHK(slim_eigh_launch(w.G, l, batch, r, w.evals, w.info + ninfo, SLIM_FIT_INFO_WORDS, w.slab,
                                 (int64_t)GRAM_MAX_SPLITS * l * l * 4 * batch, st));
        } else {
            rb_handle h = nullptr;
            CHK(solver_handle(st, &h));
            for (int bi = 0; bi < batch; ++bi)
                if (g_solver.syevd(h, RB_EVECT_ORIGINAL, RB_FILL_UPPER, l, w.G + bi * sG, l, w.evals + (int64_t)bi * l,
                                   w.ework + (int64_t)bi * l, w.info + bi * SLIM_FIT_INFO_WORDS + ninfo) != 0) {
                    slim_set_error("rocsolver_ssyevd failed");
                    return SLIM_ERR_SOLVER;
                }
        }
    }
    ninfo += 1;
    const float* vecs = phase == FIT_BACK ? ext_vecs : w.G;
    const float* evals = phase == FIT_BACK ? ext_evals : w.evals;
    if (l < Dm) {
        cond_check_kernel<<<(batch + 63) / 64, 64, 0, st>>>(evals, l, r, batch, cond_e, w.info, w.scal);
        SLIM_LAUNCH_CHECK();
    }
    const dim3 grid((unsigned)((l + 255) / 256), (unsigned)r, (unsigned)batch);
    if (right) {
        // U = Z (W S^-1), L = P (W S)
        top_eigvecs_kernel<<<grid, 256, 0, st>>>(vecs, evals, l, r, nullptr, S, w.T16, w.T16b);
        SLIM_LAUNCH_CHECK();
        CHK(nt(batch, Pb[a].X, l, sP, w.T16b, l, sG, (int)K, r, l, nullptr, (bf16_t*)L, r, nullptr, 0, stream));
        CHK(nt(batch, Zb.X, l, sZ, w.T16, l, sG, (int)N, r, l, nullptr, (bf16_t*)U, r, (bf16_t*)Ut, N, stream));
    } else {
        // U = P W, L = Z W
        top_eigvecs_kernel<<<grid, 256, 0, st>>>(vecs, evals, l, r, w.T16, S, nullptr, nullptr);
        SLIM_LAUNCH_CHECK();
        CHK(nt(batch, Zb.X, l, sZ, w.T16, l, sG, (int)K, r, l, nullptr, (bf16_t*)L, r, nullptr, 0, stream));
        CHK(nt(batch, Pb[a].X, l, sP, w.T16, l, sG, (int)N, r, l, nullptr, (bf16_t*)U, r, (bf16_t*)Ut, N, stream));
    }
    if (s) CHK(scale_rows_batched(L, s, K, r, batch, stream));
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}


// ---- the tail of add_lora in one call ------------------------------------------------------------------------------------
// out[k][j] = bf16(float(in[k][j]) * scale) for j < cols_out: a [rows, ld_in] factor cut to its first cols_out columns
__global__ void scale_cut_kernel(const bf16_t* __restrict__ in, int64_t rows, int64_t ld_in, int64_t cols_out, float scale,
                                 bf16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols_out) return;
    const int64_t k = i / cols_out, j = i - k * cols_out;
    out[i] = f32_to_bf16(bf16_to_f32(in[k * ld_in + j]) * scale);
}

// out[j][n] = bf16(float(in[n][j]) * scale) for j < rows_out: the transpose of a [N, ld_in] factor cut to rows_out rows
__global__ __launch_bounds__(256) void scale_transpose_cut_kernel(const bf16_t* __restrict__ in, int64_t N, int64_t ld_in,
                                                                  int64_t rows_out, float scale, bf16_t* __restrict__ out) {
    __shared__ bf16_t tile[64][66];
    const int64_t n0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int64_t n = n0 + r, j = j0 + tx;
        tile[r][tx] = (n < N && j < rows_out) ? in[n * ld_in + j] : (bf16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t j = j0 + r, n = n0 + tx;
        if (j < rows_out && n < N) out[j * N + n] = f32_to_bf16(bf16_to_f32(tile[tx][r]) * scale);
    }
}

extern "C" int64_t slim_lora_finish_workspace_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return -1;
    const int64_t numel = N * K;
    const int64_t bins = numel / 1000 < 512 ? 512 : (numel / 1000 > 20000 ? 20000 : numel / 1000);
    return al256(32768 * 4) + al256(slim_quant_search_workspace_bytes((int)bins)) + 256;
}

// slim/lora.py:199-209 + slim/prune.py:296-301 for separate adapters and a per-tensor quantiser (or none), in one call:
//   Wn = mask ? 0 : bf16(W - bf16(U L^T));  sf = scale of Wn (SLiM-Quant search, quant_mode 1; abs-max, 2; none, 0);
//   W_out = deq(quant(Wn, sf));  lora_left_out [K, rank] = L[:, :rank] * c;  lora_right_out [rank, N] = (U[:, :rank])^T * c
//   with c = sqrt(rank) (prescale != 0: the pre-scaling of prune.py:298-300) or 1.
extern "C" int slim_lora_finish(const void* U, const void* L, int64_t N, int64_t K, int r8, int rank, const void* W,
                                const uint8_t* mask, int quant_mode, int num_bits, int prescale, void* W_out, float* sf_out,
                                void* lora_left_out, void* lora_right_out, void* workspace, int64_t workspace_bytes,
                                void* stream) {
    SLIM_REQUIRE(U && L && W && mask && W_out && lora_left_out && lora_right_out && workspace, "slim_lora_finish: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && r8 > 0 && r8 % 8 == 0 && rank > 0 && rank <= r8 && N < (1 << 30) && K < (1 << 30),
                 "slim_lora_finish: bad sizes");
    SLIM_REQUIRE(quant_mode >= 0 && quant_mode <= 2 && (quant_mode == 0 || sf_out), "slim_lora_finish: bad quantiser mode");
    SLIM_REQUIRE(workspace_bytes >= slim_lora_finish_workspace_bytes(N, K), "slim_lora_finish: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    CHK(slim_lowrank_residual(U, L, (int)N, (int)K, r8, W, mask, W_out, nullptr, stream));
    if (quant_mode) {
        uint32_t* hist = (uint32_t*)workspace;
        char* search_ws = (char*)workspace + al256(32768 * 4);
        const int64_t numel = N * K;
        const int64_t bins = numel / 1000 < 512 ? 512 : (numel / 1000 > 20000 ? 20000 : numel / 1000);
        const int64_t search_bytes = al256(slim_quant_search_workspace_bytes((int)bins));
        float* alpha = (float*)(search_ws + search_bytes);
        CHK(slim_hist_abs_bf16(W_out, numel, hist, 0, stream));
        if (quant_mode == 1) CHK(slim_quant_search(hist, (int)bins, num_bits, alpha, nullptr, sf_out, search_ws, search_bytes, stream));
        else CHK(slim_absmax_scale(hist, num_bits, alpha, sf_out, stream));
        CHK(slim_fake_quant_pt(W_out, numel, sf_out, num_bits, W_out, nullptr, stream));
    }
    const float sq = prescale ? sqrtf((float)rank) : 1.f;
    scale_cut_kernel<<<(unsigned)ceil_div64(K * rank, 256), 256, 0, st>>>((const bf16_t*)L, K, r8, rank, sq, (bf16_t*)lora_left_out);
    SLIM_LAUNCH_CHECK();
    const dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)ceil_div64(rank, 64));
    scale_transpose_cut_kernel<<<grid, 256, 0, st>>>((const bf16_t*)U, N, r8, rank, sq, (bf16_t*)lora_right_out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// The last step of slim_lora_finish alone (slim/prune.py:296-301 on the factors of slim/lora.py:206-209): lora_left_out
// [K, rank] = L[:, :rank] * c, lora_right_out [rank, N] = (U[:, :rank])^T * c with c = sqrt(rank) (prescale != 0) or 1, for
// callers that produce the residual weight themselves (a linear held in shards by several GPUs).
extern "C" int slim_lora_adapters(const void* U, const void* L, int64_t N, int64_t K, int r8, int rank, int prescale,
                                  void* lora_left_out, void* lora_right_out, void* stream) {
    SLIM_REQUIRE(U && L && lora_left_out && lora_right_out, "slim_lora_adapters: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && r8 > 0 && r8 % 8 == 0 && rank > 0 && rank <= r8 && N < (1 << 30) && K < (1 << 30),
                 "slim_lora_adapters: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const float sq = prescale ? sqrtf((float)rank) : 1.f;
    scale_cut_kernel<<<(unsigned)ceil_div64(K * rank, 256), 256, 0, st>>>((const bf16_t*)L, K, r8, rank, sq, (bf16_t*)lora_left_out);
    SLIM_LAUNCH_CHECK();
    const dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)ceil_div64(rank, 64));
    scale_transpose_cut_kernel<<<grid, 256, 0, st>>>((const bf16_t*)U, N, r8, rank, sq, (bf16_t*)lora_right_out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- factors of a fit with split-off outlier columns, assembled -----------------------------------------------------------
// out [rows, r8]: columns < t from `exact` [rows, tp] (each row divided by bf16(s[row]) when s is given, the arithmetic of
// scale_rows_kernel), columns t .. rank - 1 from the first rank - t columns of `fit` [rows, rf], the rest zero
__global__ void assemble_factor_kernel(const bf16_t* __restrict__ exact, int tp, int t, const float* __restrict__ s,
                                       const bf16_t* __restrict__ fit, int rf, int rank, int r8, int64_t rows,
                                       bf16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * r8) return;
    const int64_t row = i / r8;
    const int j = (int)(i - row * r8);
    bf16_t v = 0;
    if (j < t) {
        v = exact[row * tp + j];
        if (s) v = f32_to_bf16(bf16_to_f32(v) / round_bf16(s[row]));
    } else if (j < rank) {
        v = fit[row * rf + (j - t)];
    }
    out[i] = v;
}

// U [N, r8], Ut [r8, N], L [K, r8] of a layer whose t outlier columns were split off before the randomized fit: the exact
// part (UO [N, tp], Ct [K, tp], Ct's rows divided by bf16(s) as the fit divides its own factor) in the first t columns, the
// fit's first rank - t columns (fitU [N, rf], fitL [K, rf]) behind them, columns >= rank zero
extern "C" int slim_lora_assemble(const void* UO, const void* Ct, const float* s, const void* fitU, const void* fitL, int64_t N,
                                  int64_t K, int tp, int t, int rf, int rank, int r8, void* U, void* Ut, void* L, void* stream) {
    SLIM_REQUIRE(UO && Ct && fitU && fitL && U && Ut && L, "slim_lora_assemble: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && t > 0 && t <= tp && rank > t && rank - t <= rf && rank <= r8 && r8 % 8 == 0,
                 "slim_lora_assemble: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    assemble_factor_kernel<<<(unsigned)ceil_div64(N * r8, 256), 256, 0, st>>>((const bf16_t*)UO, tp, t, nullptr, (const bf16_t*)fitU, rf,
                                                                             rank, r8, N, (bf16_t*)U);
    SLIM_LAUNCH_CHECK();
    assemble_factor_kernel<<<(unsigned)ceil_div64(K * r8, 256), 256, 0, st>>>((const bf16_t*)Ct, tp, t, s, (const bf16_t*)fitL, rf, rank,
                                                                             r8, K, (bf16_t*)L);
    SLIM_LAUNCH_CHECK();
    return slim_transpose_bf16(U, N, r8, Ut, stream);
}

// ---- the head of add_lora in one call ------------------------------------------------------------------------------------
// Outlier columns of the error matrix (input channels whose saliency scale is far above the median) are split off before
// the randomized fit: with Q_O an orthonormal basis of those t columns, E = Q_O (Q_O^T E) + E'.  The columns are rows of
// E^T, so everything below reads contiguous rows.  The basis comes from the Cholesky factor of the columns' normalised
// Gram matrix (Q_O = E_O D^-1 R^-1 with D the column norms): after the normalisation the columns are close to orthogonal
// (error columns of different channels are nearly uncorrelated), so the factorisation is well conditioned, and it is
// carried in double precision.

// G[i][j] (double, [tp][tp]) = <column c_i, column c_j> for i <= j < t; one workgroup per pair
__global__ __launch_bounds__(256) void outlier_gram_kernel(const bf16_t* __restrict__ Et, int64_t N, const int32_t* __restrict__ cols,
                                                           int t, int tp, double* __restrict__ G) {
    __shared__ double red[4];
    const int i = blockIdx.y, j = blockIdx.x;
    if (i > j || j >= t) {
        if (threadIdx.x == 0) G[i * tp + j] = 0.0;
        return;
    }
    const bf16_t* a = Et + (int64_t)cols[i] * N;
    const bf16_t* b = Et + (int64_t)cols[j] * N;
    double acc = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += 256) acc += (double)(bf16_to_f32(a[n]) * bf16_to_f32(b[n]));
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) G[i * tp + j] = red[0] + red[1] + red[2] + red[3];
}

// T [tp][tp] (fp32, upper triangular) with Q_O = E_O T: one workgroup of 64 threads, thread j owns column j.  A column whose
// pivot vanishes (a zero column, or one inside the span of the columns before it) gets a zero column of T: the basis then
// spans the other columns only.
__global__ __launch_bounds__(64) void outlier_basis_kernel(const double* __restrict__ G, int t, int tp, float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double* R = sh;                 // [t][t] Cholesky factor of the normalised Gram matrix, upper triangular, row-major
    double* Ri = sh + t * t;        // its inverse
    double* dn = sh + 2 * t * t;    // column norms; dead[j] behind them
    double* dead = dn + t;
    const int j = threadIdx.x;
    if (j < t) {
        dn[j] = sqrt(G[j * tp + j]);
        dead[j] = 0.0;
    }
    __syncthreads();
    if (j < t)
        for (int i = 0; i < t; ++i) {
            const double den = dn[i] * dn[j];
            R[i * t + j] = (i <= j && den > 0.0) ? G[i * tp + j] / den : 0.0;
            Ri[i * t + j] = 0.0;
        }
    __syncthreads();
    // right-looking Cholesky, row k at a time: R[k][k] = sqrt(pivot), R[k][j] /= R[k][k], trailing update by thread j
    for (int k = 0; k < t; ++k) {
        const double piv = R[k * t + k];
        const bool ok = piv > 1e-10;
        __syncthreads();
        if (j == k) {
            R[k * t + k] = ok ? sqrt(piv) : 1.0;
            if (!ok) dead[k] = 1.0;
        }
        __syncthreads();
        if (j > k && j < t) R[k * t + j] = ok ? R[k * t + j] / R[k * t + k] : 0.0;
        __syncthreads();
        if (j > k && j < t) {
            const double rkj = R[k * t + j];
            for (int i = k + 1; i <= j; ++i) R[i * t + j] -= R[k * t + i] * rkj;
        }
        __syncthreads();
    }
    // inverse of the upper triangular factor, column j by thread j (back substitution on R Ri = I)
    if (j < t) {
        for (int i = j; i >= 0; --i) {
            double v = i == j ? 1.0 : 0.0;
            for (int k = i + 1; k <= j; ++k) v -= R[i * t + k] * Ri[k * t + j];
            Ri[i * t + j] = v / R[i * t + i];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tp * tp; e += 64) {
        const int i = e / tp, c = e - i * tp;
        float v = 0.f;
        if (i <= c && c < t && dead[c] == 0.0 && dead[i] == 0.0 && dn[i] > 0.0) v = (float)(Ri[i * t + c] / dn[i]);
        T[e] = v;
    }
}

// UOt [tp][N] and UO [N][tp] (bf16) = the basis Q_O = E_O T and its transpose; columns >= t zero.  One thread per row n.
__global__ __launch_bounds__(256) void outlier_q_kernel(const bf16_t* __restrict__ Et, int64_t N, const int32_t* __restrict__ cols,
                                                        int t, int tp, const float* __restrict__ T, bf16_t* __restrict__ UO,
                                                        bf16_t* __restrict__ UOt) {
    extern __shared__ __attribute__((aligned(16))) float ev[];      // [t][256] this workgroup's entries of the columns, then T
    float* Ts = ev + t * 256;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < t; ++i) ev[i * 256 + threadIdx.x] = n < N ? bf16_to_f32(Et[(int64_t)cols[i] * N + n]) : 0.f;
    for (int e = threadIdx.x; e < tp * tp; e += 256) Ts[e] = T[e];
    __syncthreads();
    if (n >= N) return;
    for (int c = 0; c < tp; ++c) {
        float q = 0.f;
        if (c < t)
            for (int i = 0; i <= c; ++i) q += Ts[i * tp + c] * ev[i * 256 + threadIdx.x];
        const bf16_t h = f32_to_bf16(q);
        UOt[(int64_t)c * N + n] = h;
        UO[n * tp + c] = h;
    }
}

// Ct [K, tp] = bf16(Et UOt^T): the projection of every column of E (a row of Et [K, N]) on the tp basis vectors (rows of
// UOt [tp, N]).  A product with tp <= 64 output columns: a wave per row of Et streams the row once (16 bytes per lane and
// step) against the basis, which stays in cache; the tile GEMM would run one 128-wide tile column for it.
template <int TP>
__global__ __launch_bounds__(256) void outlier_project_kernel(const bf16_t* __restrict__ Et, const bf16_t* __restrict__ UOt, int64_t K,
                                                              int64_t N, bf16_t* __restrict__ Ct) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    const bf16_t* row = Et + k * N;
    float acc[TP];
#pragma unroll
    for (int c = 0; c < TP; ++c) acc[c] = 0.f;
    for (int64_t n = (int64_t)lane * 8; n < N; n += 512) {
        const u16x8 e = *(const u16x8*)(row + n);
        float ef[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ef[u] = bf16_to_f32(e[u]);
#pragma unroll
        for (int c = 0; c < TP; ++c) {
            const u16x8 b = *(const u16x8*)(UOt + (int64_t)c * N + n);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[c] += ef[u] * bf16_to_f32(b[u]);
        }
    }
#pragma unroll
    for (int c = 0; c < TP; ++c) acc[c] = wave_sum(acc[c]);
    if (lane < TP) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < TP; ++c)
            if (lane == c) v = acc[c];
        Ct[k * TP + lane] = f32_to_bf16(v);
    }
}

extern "C" int slim_outlier_project(const void* Et, const void* UOt, int64_t K, int64_t N, int tp, void* Ct, void* stream) {
    SLIM_REQUIRE(Et && UOt && Ct && K > 0 && N > 0 && N % 8 == 0, "slim_outlier_project: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(K, 4);
    if (tp == 8) outlier_project_kernel<8><<<grid, 256, 0, st>>>((const bf16_t*)Et, (const bf16_t*)UOt, K, N, (bf16_t*)Ct);
    else if (tp == 16) outlier_project_kernel<16><<<grid, 256, 0, st>>>((const bf16_t*)Et, (const bf16_t*)UOt, K, N, (bf16_t*)Ct);
    else {
        // wider bases (rare: more than 16 outlier channels): the tile GEMM, transposed output only
        SLIM_REQUIRE(tp > 0 && tp % 8 == 0 && tp <= 64, "slim_outlier_project: tp must be a multiple of 8, at most 64");
        return slim_gemm_nt_bf16(UOt, N, Et, N, tp, (int)K, (int)N, 1.f, 0.f, nullptr, 0, nullptr, 0, Ct, tp, stream);
    }
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// E2 [N, K] = bf16(E - bf16(UO Ct^T)) and its transpose in one pass (UO [N, tp], Ct [K, tp], tp <= 64): the subtraction of
// the exact part of the split-off columns.  With so short a reduction the product is element-wise work: 64 x 64 tiles, the
// two operand tiles in LDS, fp32 sums in column order, the rounding points of slim_lowrank_residual.
__global__ __launch_bounds__(256) void deflate_apply_kernel(const bf16_t* __restrict__ E, const bf16_t* __restrict__ UO,
                                                            const bf16_t* __restrict__ Ct, int64_t N, int64_t K, int tp,
                                                            bf16_t* __restrict__ E2, bf16_t* __restrict__ E2t) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* us = dsm;                      // [64][tp + 1]
    float* cs = dsm + 64 * (tp + 1);      // [64][tp + 1]
    bf16_t* tile = (bf16_t*)(dsm + 2 * 64 * (tp + 1));      // [64][72]
    const int64_t n0 = (int64_t)blockIdx.y * 64, k0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x, c = (t & 7) * 8;
    for (int idx = t; idx < 64 * tp; idx += 256) {
        const int r = idx / tp, j = idx - r * tp;
        us[r * (tp + 1) + j] = n0 + r < N ? bf16_to_f32(UO[(n0 + r) * tp + j]) : 0.f;
        cs[r * (tp + 1) + j] = k0 + r < K ? bf16_to_f32(Ct[(k0 + r) * tp + j]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = (t >> 3) + 32 * pass;
        const int64_t row = n0 + r, col = k0 + c;
        u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < N && col < K) {
            const u16x8 e = *(const u16x8*)(E + row * K + col);
            float lr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) lr[u] = 0.f;
            for (int j = 0; j < tp; ++j) {
                const float uj = us[r * (tp + 1) + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) lr[u] += uj * cs[(c + u) * (tp + 1) + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = f32_to_bf16(bf16_to_f32(e[u]) - round_bf16(lr[u]));
            *(u16x8*)(E2 + row * K + col) = o;
        }
        *(u16x8*)(tile + r * 72 + c) = o;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int kc = (t >> 3) + 32 * pass;
        const int64_t col = k0 + kc, row = n0 + c;
        if (col >= K || row >= N) continue;
        u16x8 o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = tile[(c + u) * 72 + kc];
        *(u16x8*)(E2t + col * N + row) = o;
    }
}

extern "C" int slim_deflate_apply(const void* E, const void* UO, const void* Ct, int64_t N, int64_t K, int tp, void* E2, void* E2t,
                                  void* stream) {
    SLIM_REQUIRE(E && UO && Ct && E2 && E2t && N > 0 && K > 0, "slim_deflate_apply: bad arguments");
    SLIM_REQUIRE(N % 8 == 0 && K % 8 == 0 && tp > 0 && tp <= 64, "slim_deflate_apply: N, K multiples of 8 and tp in [1, 64]");
    SLIM_REQUIRE(ceil_div64(N, 64) <= 65535, "slim_deflate_apply: too many rows");
    const size_t lds = (size_t)2 * 64 * (tp + 1) * sizeof(float) + (size_t)64 * 72 * sizeof(bf16_t);
    SLIM_PROF("deflate_apply", stream, 6.0 * N * K);
    dim3 grid((unsigned)ceil_div64(K, 64), (unsigned)ceil_div64(N, 64));
    deflate_apply_kernel<<<grid, 256, lds, (hipStream_t)stream>>>((const bf16_t*)E, (const bf16_t*)UO, (const bf16_t*)Ct, N, K, tp,
                                                                   (bf16_t*)E2, (bf16_t*)E2t);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

static inline int64_t prepare_search_bins(int64_t numel) { return numel / 1000 < 512 ? 512 : (numel / 1000 > 20000 ? 20000 : numel / 1000); }

extern "C" int64_t slim_outlier_basis_workspace_bytes(int tp) {
    return tp < 0 ? -1 : al256((int64_t)tp * tp * 8) + al256((int64_t)tp * tp * 4);
}

// UO [N, tp] / UOt [tp, N] = an orthonormal basis of the columns cols[0 .. t) of E, given as rows of Et [K, N]; columns >= t zero
extern "C" int slim_outlier_basis(const void* Et, int64_t N, const int32_t* cols, int t, int tp, void* UO, void* UOt, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(Et && cols && UO && UOt && workspace, "slim_outlier_basis: null pointer");
    SLIM_REQUIRE(N > 0 && t > 0 && tp >= t && tp % 8 == 0 && tp <= 64, "slim_outlier_basis: bad sizes");
    SLIM_REQUIRE(workspace_bytes >= slim_outlier_basis_workspace_bytes(tp), "slim_outlier_basis: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* G = (double*)workspace;
    float* T = (float*)((char*)workspace + al256((int64_t)tp * tp * 8));
    outlier_gram_kernel<<<dim3((unsigned)tp, (unsigned)tp), 256, 0, st>>>((const bf16_t*)Et, N, cols, t, tp, G);
    SLIM_LAUNCH_CHECK();
    const size_t basis_lds = (size_t)(2 * t * t + 2 * t) * sizeof(double);       // 65 KiB at t = 64: above the default limit
    SLIM_HIP(slim_allow_dynamic_lds((const void*)outlier_basis_kernel, (int)basis_lds));
    outlier_basis_kernel<<<1, 64, basis_lds, st>>>(G, t, tp, T);
    SLIM_LAUNCH_CHECK();
    const size_t q_lds = ((size_t)t * 256 + (size_t)tp * tp) * sizeof(float);
    SLIM_HIP(slim_allow_dynamic_lds((const void*)outlier_q_kernel, (int)q_lds));
    outlier_q_kernel<<<(unsigned)ceil_div64(N, 256), 256, q_lds, st>>>((const bf16_t*)Et, N, cols, t, tp, T, (bf16_t*)UO, (bf16_t*)UOt);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

extern "C" int64_t slim_lora_prepare_workspace_bytes(int64_t N, int64_t K, int tp) {
    if (N <= 0 || K <= 0 || tp < 0) return -1;
    return al256(32768 * 4) + al256(slim_quant_search_workspace_bytes((int)prepare_search_bins(N * K))) + 256 +
           slim_outlier_basis_workspace_bytes(tp);
}

// slim/prune.py:264-272 + slim/lora.py:139-156 for a per-tensor quantiser (or none) with quantize_first, in one call:
//   s = sqrt(scaler_row);  mask = prune_nm(|W| * s, prune_n, prune_m) when prune_n > 0 (else `mask` is an input);
//   sf = scale of W (quant_mode 1: SLiM-Quant search; 2: abs-max; 0: no quantiser);
//   E = W * s' - mask0(deq(quant(W, sf)) * s') in bf16 and its transpose, s' = s when scale_error != 0, else 1;
// and, when t > 0 outlier columns are listed (outlier_idx, padded to tp entries, tp a multiple of 8): UO [N, tp], UOt [tp, N]
// = an orthonormal basis of those columns of E (columns >= t zero), Ct [K, tp] = (UO^T E)^T, E2 = E - UO Ct^T and E2t.
extern "C" int slim_lora_prepare(const void* W, const float* scaler_row, int64_t N, int64_t K, int prune_n, int prune_m, uint8_t* mask,
                                 int quant_mode, int num_bits, int scale_error, float* s_out, float* sf_out, void* E, void* Et,
                                 const int32_t* outlier_idx, int t, int tp, void* UO, void* UOt, void* Ct, void* E2, void* E2t,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(W && scaler_row && mask && s_out && E && Et && workspace, "slim_lora_prepare: null pointer");
    SLIM_REQUIRE(N > 0 && K > 0 && N < (1 << 30) && K < (1 << 30) && N % 8 == 0 && K % 8 == 0, "slim_lora_prepare: bad sizes");
    SLIM_REQUIRE(quant_mode >= 0 && quant_mode <= 2 && (quant_mode == 0 || sf_out), "slim_lora_prepare: bad quantiser mode");
    SLIM_REQUIRE(t >= 0 && tp >= t && tp % 8 == 0 && tp <= 64, "slim_lora_prepare: bad outlier counts");
    SLIM_REQUIRE(t == 0 || (outlier_idx && UO && UOt && Ct && E2 && E2t), "slim_lora_prepare: null pointer (outlier outputs)");
    SLIM_REQUIRE(workspace_bytes >= slim_lora_prepare_workspace_bytes(N, K, tp), "slim_lora_prepare: workspace too small");
    const int64_t numel = N * K;
    CHK(slim_sqrt_f32(scaler_row, s_out, K, stream));
    if (prune_n > 0) CHK(slim_mask_nm(W, s_out, N, K, prune_n, prune_m, mask, stream));
    uint32_t* hist = (uint32_t*)workspace;
    char* search_ws = (char*)workspace + al256(32768 * 4);
    const int64_t bins = prepare_search_bins(numel);
    const int64_t search_bytes = al256(slim_quant_search_workspace_bytes((int)bins));
    float* alpha = (float*)(search_ws + search_bytes);
    char* basis_ws = search_ws + search_bytes + 256;
    if (quant_mode) {
        CHK(slim_hist_abs_bf16(W, numel, hist, 0, stream));
        if (quant_mode == 1) CHK(slim_quant_search(hist, (int)bins, num_bits, alpha, nullptr, sf_out, search_ws, search_bytes, stream));
        else CHK(slim_absmax_scale(hist, num_bits, alpha, sf_out, stream));
    }
    // (the abs-max quantiser's scale is a bf16 tensor in the reference, so its dequantised values are bf16)
    CHK(slim_error_matrix_t_rounded(W, mask, scale_error ? s_out : nullptr, quant_mode ? sf_out : nullptr, num_bits, quant_mode == 2, N, K,
                                    E, Et, stream));
    if (t == 0) return SLIM_OK;
    CHK(slim_outlier_basis(Et, N, outlier_idx, t, tp, UO, UOt, basis_ws, slim_outlier_basis_workspace_bytes(tp), stream));
    CHK(slim_outlier_project(Et, UOt, K, N, tp, Ct, stream));
    CHK(slim_deflate_apply(E, UO, Ct, N, K, tp, E2, E2t, stream));
    return SLIM_OK;
}

// ---- the pieces of the Gram-side fit as separate calls -----------------------------------------------------------------
// For a layer sharded over several GPUs along its long dimension: every GPU forms the Gram matrix of its shard of E
// (slim_gemm_nt_bf16 with fp32 output), the sum over the shards goes to one GPU, which runs the subspace iteration
// (slim_gramfit_iterate) and sends the basis P to all; every GPU projects its shard (slim_gramfit_project: Z = shard x P and
// the shard's share of the Rayleigh-Ritz matrix Z^T Z, summed over the GPUs again), and after the l x l eigenproblem turns
// its rows of the factors out (slim_gramfit_factors).  No counterpart in the reference, which is single-GPU.
extern "C" int64_t slim_gramfit_workspace_bytes(int64_t rows, int64_t l) {
    if (rows <= 0 || l <= 0) return 0;
    return fit_ws(nullptr, 1, rows, 8, l).total;
}

extern "C" int slim_gramfit_iterate(const void* C16, int64_t Dm, int l, int n_iter, uint32_t seed, int flags, int cond_e, void* P,
                                    void* Pt, int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(C16 && P && Pt && workspace, "slim_gramfit_iterate: null pointer");
    SLIM_REQUIRE(Dm > 0 && l > 0 && l < Dm && Dm % 8 == 0 && l % 8 == 0 && n_iter >= 0, "slim_gramfit_iterate: bad sizes");
    SLIM_REQUIRE(2 * (n_iter + 1) + 1 <= SLIM_FIT_INFO_WORDS / 2, "slim_gramfit_iterate: too many iterations");
    SLIM_REQUIRE(workspace_bytes >= slim_gramfit_workspace_bytes(Dm, l), "slim_gramfit_iterate: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, 1, Dm, 8, l);
    SLIM_HIP(hipMemsetAsync(w.info, 0, SLIM_FIT_INFO_WORDS * sizeof(int), st));
    const int cond = ns_levels_clamped(cond_e);
    int a = 0, ninfo = 0;
    // (no filter on this path: the caller's matrix is only read)
    CHK(gram_iterate(w.A, (bf16_t*)C16, Dm * Dm, 1, Dm, l, n_iter, seed, flags & ~SLIM_FIT_CHOLESKY_QR,
                     cond, w, w.sketch_slab, (int64_t)SKETCH_MAX_SPLITS * l * Dm, stream, &a, &ninfo));
    const size_t bytes = (size_t)Dm * l * sizeof(bf16_t);
    SLIM_HIP(hipMemcpyAsync(P, w.A[a].X, bytes, hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemcpyAsync(Pt, w.A[a].Xt, bytes, hipMemcpyDeviceToDevice, st));
    if (info_out) SLIM_HIP(hipMemcpyAsync(info_out, w.info, SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}

// slim_gramfit_iterate for `batch` Gram matrices of one size in lockstep (C16 [batch][Dm, Dm]; P [batch][Dm, l], Pt [batch][l, Dm],
// info_out [batch][SLIM_FIT_INFO_WORDS]): every launch carries the whole batch, as in slim_lowrank_fit_front_batched.  A
// request left by slim_lowrank_fit_set_filter on the calling thread (one ratio per matrix) is honoured: the Chebyshev-
// filtered iteration, which shifts the diagonals of C16 IN PLACE.  scal_out (optional, [batch][4] floats): per matrix
// {-, mean eigenvalue trace / Dm, upper end beta of the damped interval or 0, -} for slim_gramfit_check.
extern "C" int64_t slim_gramfit_batched_workspace_bytes(int batch, int64_t rows, int64_t l) {
    if (batch <= 0 || rows <= 0 || l <= 0) return 0;
    return fit_ws(nullptr, batch, rows, 8, l).total;
}

extern "C" int slim_gramfit_iterate_batched(void* C16, int batch, int64_t Dm, int l, int n_iter, uint32_t seed, int flags, int cond_e,
                                            void* P, void* Pt, int32_t* info_out, float* scal_out, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(C16 && P && Pt && workspace, "slim_gramfit_iterate_batched: null pointer");
    SLIM_REQUIRE(batch >= 1 && batch <= FILTER_MAX && Dm > 0 && l > 0 && l < Dm && Dm % 8 == 0 && l % 8 == 0 && n_iter >= 0,
                 "slim_gramfit_iterate_batched: bad sizes");
    SLIM_REQUIRE(2 * (n_iter + 1) + 1 <= SLIM_FIT_INFO_WORDS / 2, "slim_gramfit_iterate_batched: too many iterations");
    SLIM_REQUIRE(workspace_bytes >= slim_gramfit_batched_workspace_bytes(batch, Dm, l), "slim_gramfit_iterate_batched: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, batch, Dm, 8, l);
    SLIM_HIP(hipMemsetAsync(w.info, 0, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), st));
    SLIM_HIP(hipMemsetAsync(w.scal, 0, (size_t)batch * SCAL_STRIDE * sizeof(float), st));
    const int cond = ns_levels_clamped(cond_e);
    float fr[FILTER_MAX];
    int fbits = 0, a = 0, ninfo = 0;
    const bool filt = take_filter(batch, fr, &fbits);
    // several problems side by side fill the chip: no split of the products along their reduction
    const bool alone = batch == 1 && !(flags & SLIM_FIT_NO_SKETCH_SPLIT);
    if (!alone) w.gram_cap = GRAM_SPLITS_SHARED_CHIP;
    CHK(gram_iterate(w.A, (bf16_t*)C16, Dm * Dm, batch, Dm, l, n_iter, seed, flags & ~SLIM_FIT_CHOLESKY_QR, cond, w,
                     alone ? w.sketch_slab : nullptr, (int64_t)SKETCH_MAX_SPLITS * l * Dm, stream, &a, &ninfo, filt ? fr : nullptr, fbits));
    const size_t bytes = (size_t)batch * Dm * l * sizeof(bf16_t);
    SLIM_HIP(hipMemcpyAsync(P, w.A[a].X, bytes, hipMemcpyDeviceToDevice, st));
    SLIM_HIP(hipMemcpyAsync(Pt, w.A[a].Xt, bytes, hipMemcpyDeviceToDevice, st));
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (scal_out)
        SLIM_HIP(hipMemcpyAsync(scal_out, w.scal, (size_t)batch * SCAL_STRIDE * sizeof(float), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}

// The reports of a finished fit for `batch` problems (words SLIM_FIT_INFO_COND_WORD, _LEVEL_WORD, _RATIO_WORD of info
// [batch][SLIM_FIT_INFO_WORDS]) from the eigenvalues of their Rayleigh-Ritz matrices (evals [batch][l] ascending, or a
// Ritz split's estimates in that layout), for an iteration sized for cond_e; scal (optional): what
// slim_gramfit_iterate_batched left in scal_out, for the ratio word and the check of the filter's interval.
extern "C" int slim_gramfit_check(const float* evals, int l, int r, int batch, int cond_e, const float* scal, int32_t* info, void* stream) {
    SLIM_REQUIRE(evals && info && l > 0 && r > 0 && r <= l && batch >= 1, "slim_gramfit_check: bad arguments");
    cond_check_kernel<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(evals, l, r, batch, ns_levels_clamped(cond_e), info, scal);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Z [rows, l] = X P and Z^T, M32 [l, l] = Z^T Z (fp32): X [rows, Dm] is the shard of E (or of E^T) whose columns the basis P
// [Dm, l] (given as Pt [l, Dm]) spans
extern "C" int slim_gramfit_project(const void* X, int64_t rows, int64_t Dm, int l, const void* Pt, void* Z, void* Zt, float* M32,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(X && Pt && Z && Zt && M32 && workspace, "slim_gramfit_project: null pointer");
    SLIM_REQUIRE(rows > 0 && Dm > 0 && l > 0 && rows % 8 == 0 && Dm % 8 == 0 && l % 8 == 0, "slim_gramfit_project: bad sizes");
    SLIM_REQUIRE(workspace_bytes >= slim_gramfit_workspace_bytes(rows, l), "slim_gramfit_project: workspace too small");
    FitWs w = fit_ws((char*)workspace, 1, rows, 8, l);
    CHK(nt(1, (const bf16_t*)Pt, Dm, 0, (const bf16_t*)X, Dm, 0, l, (int)rows, (int)Dm, nullptr, (bf16_t*)Zt, rows, (bf16_t*)Z, l, stream,
           w.sketch_slab, (int64_t)SKETCH_MAX_SPLITS * l * rows));
    w.G = M32;
    return gram((const bf16_t*)Zt, rows, l, w, nullptr, stream);
}

// The factors from the eigenvectors of the Rayleigh-Ritz matrix (vecs [l, l]: row l-1-i = eigenvector of the i-th largest
// eigenvalue, evals [l] ascending, as slim_eigh_tail_f32 leaves them):
//   right != 0 (P spans right singular vectors, Dm = K):  dm_out = L = P W S [K, r],  rows_out = U rows = Z W S^-1
//   right == 0 (P spans left singular vectors, Dm = N):   dm_out = U = P W [N, r],    rows_out = L rows = Z W
// dm_out_t / rows_out_t (optional): the transposes.  S [r] (optional): singular values.  s (optional): the saliency
// scales of L's rows (length Dm when right, else rows); L is divided by bf16(s) as slim_lowrank_fit does.  info (optional,
// SLIM_FIT_INFO_WORDS words): its last word receives the conditioning check of slim_lowrank_fit for an iteration sized for
// cond_e (0: the default).
extern "C" int slim_gramfit_factors(int right, const void* P, const void* Z, int64_t Dm, int64_t rows, int l, int r,
                                    const float* vecs, const float* evals, const float* s, void* dm_out, void* dm_out_t,
                                    void* rows_out, void* rows_out_t, float* S, int cond_e, int32_t* info, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(P && Z && vecs && evals && dm_out && rows_out && workspace, "slim_gramfit_factors: null pointer");
    SLIM_REQUIRE(!s || !(right ? dm_out_t : rows_out_t), "slim_gramfit_factors: the transpose of a scaled factor is not produced");
    SLIM_REQUIRE(Dm > 0 && rows > 0 && l > 0 && r > 0 && r <= l && Dm % 8 == 0 && rows % 8 == 0 && l % 8 == 0 && r % 8 == 0,
                 "slim_gramfit_factors: bad sizes");
    SLIM_REQUIRE(workspace_bytes >= 2 * al256((int64_t)l * l * 2), "slim_gramfit_factors: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    bf16_t* T16 = (bf16_t*)workspace;
    bf16_t* T16b = (bf16_t*)((char*)workspace + al256((int64_t)l * l * 2));
    const dim3 grid((unsigned)((l + 255) / 256), (unsigned)r, 1);
    if (info) {     // the conditioning the spectrum shows against the level the iteration was sized for -> last status word
        cond_check_kernel<<<1, 64, 0, st>>>(evals, l, r, 1, ns_levels_clamped(cond_e), info);
        SLIM_LAUNCH_CHECK();
    }
    if (right) {
        top_eigvecs_kernel<<<grid, 256, 0, st>>>(vecs, evals, l, r, nullptr, S, T16, T16b);
        SLIM_LAUNCH_CHECK();
        CHK(nt(1, (const bf16_t*)P, l, 0, T16b, l, 0, (int)Dm, r, l, nullptr, (bf16_t*)dm_out, r, (bf16_t*)dm_out_t, Dm, stream));
        CHK(nt(1, (const bf16_t*)Z, l, 0, T16, l, 0, (int)rows, r, l, nullptr, (bf16_t*)rows_out, r, (bf16_t*)rows_out_t, rows, stream));
        if (s) CHK(scale_rows_batched(dm_out, s, Dm, r, 1, stream));
    } else {
        top_eigvecs_kernel<<<grid, 256, 0, st>>>(vecs, evals, l, r, T16, S, nullptr, nullptr);
        SLIM_LAUNCH_CHECK();
        CHK(nt(1, (const bf16_t*)P, l, 0, T16, l, 0, (int)Dm, r, l, nullptr, (bf16_t*)dm_out, r, (bf16_t*)dm_out_t, Dm, stream));
        CHK(nt(1, (const bf16_t*)Z, l, 0, T16, l, 0, (int)rows, r, l, nullptr, (bf16_t*)rows_out, r, (bf16_t*)rows_out_t, rows, stream));
        if (s) CHK(scale_rows_batched(rows_out, s, rows, r, 1, stream));
    }
    return SLIM_OK;
}

extern "C" int slim_lowrank_fit_batched(const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l,
                                        int n_iter, uint32_t seed, int flags, int ns_steps, const float* s, void* U,
                                        void* Ut, void* L, float* S, int32_t* info_out, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    return fit_impl(FIT_ALL, E, Et, batch, N, K, r, l, n_iter, seed, flags, ns_steps, s, U, Ut, L, S, info_out, workspace,
                    workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// The fit in two halves around its l x l eigenproblem, so that a caller compressing several layers at once can solve all
// their eigenproblems' launch-bound first stage in ONE batched chain (slim_tridiagonalize_f32_batched) instead of one
// 639-launch chain per layer.  front: everything up to the Rayleigh-Ritz matrix, written to M_out [l, l] fp32; *state
// (host) is handed to back unchanged, with the same workspace and sizes.  back: vecs [l, l] (row l-1-i = eigenvector of
// the i-th largest eigenvalue, i < r), evals [l] ascending, eig_info (device, solver status or NULL) -> the factors.
// The _batched forms run `batch` same-shape fits in lockstep (operands strided as in slim_lowrank_fit_batched,
// M_out / vecs [batch][l, l], evals [batch][l], eig_info [batch]): each launch then carries the whole group.
extern "C" int slim_lowrank_fit_front_batched(const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l,
                                              int n_iter, uint32_t seed, int flags, int ns_steps, float* M_out,
                                              int32_t* state, void* workspace, int64_t workspace_bytes, void* stream) {
    return fit_impl(FIT_FRONT, E, Et, batch, N, K, r, l, n_iter, seed, flags, ns_steps, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr, workspace, workspace_bytes, stream, M_out, state, nullptr, nullptr, nullptr);
}
extern "C" int slim_lowrank_fit_back_batched(int batch, int64_t N, int64_t K, int r, int l, int n_iter, const int32_t* state,
                                             const float* vecs, const float* evals, const int32_t* eig_info, const float* s,
                                             void* U, void* Ut, void* L, float* S, int32_t* info_out, void* workspace,
                                             int64_t workspace_bytes, void* stream) {
    int st_copy = state ? *state : -1;
    return fit_impl(FIT_BACK, nullptr, nullptr, batch, N, K, r, l, n_iter, 0, 0, 0, s, U, Ut, L, S, info_out, workspace,
                    workspace_bytes, stream, nullptr, &st_copy, vecs, evals, eig_info);
}
// ---- lockstep group of fits with different shapes -----------------------------------------------------------------------
// The subspace iteration of the Gram-side fit sees only the Dm x Dm matrix C, Dm = min(N, K): fits of layers with different
// shapes but the same Dm, l and r (all seven linears of a LLaMA block: 4096 x 4096, 11008 x 4096, 4096 x 11008) run it as ONE
// batched chain.  What touches the N x K matrices (C itself, the projection Z, the factors) runs member by member between the
// batched parts; every member's arithmetic is that of slim_lowrank_fit_front / _back on it alone, so the results are
// bit-identical to single fits.  E / Et / U / Ut / L / s: host arrays of `batch` device pointers; Ns / Ks: host arrays of
// the shapes.  The workspace is that of a batch of Dx_max x Dm problems, Dx_max the largest max(N, K).
static int group_dims(const int64_t* Ns, const int64_t* Ks, int batch, int64_t* Dm_out, int64_t* Dx_max_out) {
    SLIM_REQUIRE(Ns && Ks && batch >= 1 && batch <= 1024, "slim_lowrank_fit_group: bad batch arguments");
    int64_t Dm = 0, Dx_max = 0;
    for (int b = 0; b < batch; ++b) {
        const int64_t mn = Ns[b] < Ks[b] ? Ns[b] : Ks[b], mx = Ns[b] < Ks[b] ? Ks[b] : Ns[b];
        SLIM_REQUIRE(mn > 0 && Ns[b] % 8 == 0 && Ks[b] % 8 == 0, "slim_lowrank_fit_group: N, K must be positive multiples of 8");
        SLIM_REQUIRE(b == 0 || mn == Dm, "slim_lowrank_fit_group: the members must share min(N, K)");
        Dm = mn;
        Dx_max = mx > Dx_max ? mx : Dx_max;
    }
    *Dm_out = Dm;
    *Dx_max_out = Dx_max;
    return SLIM_OK;
}

extern "C" int64_t slim_lowrank_fit_group_workspace_bytes(const int64_t* Ns, const int64_t* Ks, int batch, int64_t l) {
    int64_t Dm = 0, Dx = 0;
    if (l <= 0 || group_dims(Ns, Ks, batch, &Dm, &Dx) != SLIM_OK) return 0;
    return fit_ws(nullptr, batch, Dx, Dm, l).total;
}

// the batched workspace as member b alone sees it (the buffers the per-member products use)
static FitWs member_ws(const FitWs& w, int b, int64_t l) {
    FitWs m = w;
    const int64_t sG = l * l;
    m.batch = 1;
    m.G = w.G + b * sG;
    m.G2 = w.G2 + b * sG;
    m.G16 = w.G16 + b * sG;
    m.T16 = w.T16 + b * sG;
    m.T16b = w.T16b + b * sG;
    m.slab = w.slab + (int64_t)b * GRAM_MAX_SPLITS * sG;
    m.info = w.info + b * SLIM_FIT_INFO_WORDS;
    return m;
}

extern "C" int slim_lowrank_fit_front_group(const void* const* E, const void* const* Et, const int64_t* Ns, const int64_t* Ks,
                                            int batch, int r, int l, int n_iter, uint32_t seed, int flags, int ns_steps,
                                            float* M_out, int32_t* state, void* workspace, int64_t workspace_bytes,
                                            void* stream) {
    int64_t Dm = 0, Dxm = 0;
    CHK(group_dims(Ns, Ks, batch, &Dm, &Dxm));
    SLIM_REQUIRE(E && Et && M_out && state && workspace, "slim_lowrank_fit_front_group: null pointer");
    SLIM_REQUIRE(r > 0 && l >= r && l < Dm && l % 8 == 0 && r % 8 == 0 && n_iter >= 0, "slim_lowrank_fit_front_group: bad sizes");
    SLIM_REQUIRE(2 * (n_iter + 1) + 1 <= SLIM_FIT_INFO_WORDS / 2, "slim_lowrank_fit_front_group: too many iterations");
    SLIM_REQUIRE(!(flags & SLIM_FIT_CHOLESKY_QR),
                 "slim_lowrank_fit_front_group: only the Gram-side iteration runs as a mixed-shape group");
    SLIM_REQUIRE(workspace_bytes >= fit_ws(nullptr, batch, Dxm, Dm, l).total, "slim_lowrank_fit_front_group: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, batch, Dxm, Dm, l);      // w.B: the iterates (Dm rows), w.A[0]: the projections
    SLIM_HIP(hipMemsetAsync(w.info, 0, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), st));
    const int cond = ns_levels_clamped(ns_steps);
    const int64_t sP = Dm * l, sZ = Dxm * l, sG = (int64_t)l * l, sC = Dm * Dm, sSk = (int64_t)SKETCH_MAX_SPLITS * l * Dxm;
    float* sk_slab = (flags & SLIM_FIT_NO_SKETCH_SPLIT) ? nullptr : w.sketch_slab;
    if (flags & SLIM_FIT_NO_SKETCH_SPLIT) w.gram_cap = GRAM_SPLITS_SHARED_CHIP;
    bool same_dx = true;
    for (int b = 0; b < batch; ++b) {
        SLIM_REQUIRE(E[b] && Et[b], "slim_lowrank_fit_front_group: null operand");
        same_dx = same_dx && (Ks[b] <= Ns[b] ? Ns[b] : Ks[b]) == (Ks[0] <= Ns[0] ? Ns[0] : Ks[0]);
    }
    // C = E^T E (operand E^T [K, N]) or E E^T (operand E [N, K]) per member.  The members' matrices are separate
    // allocations: up to GEMM_MAX_PTRS of them go into one launch (their reduction lengths may differ), so that the tiles of
    // all members share the waves over the chip instead of every member ending with a partly filled one.
    for (int b0 = 0; b0 < batch; b0 += GEMM_MAX_PTRS) {
        const int nb = batch - b0 < GEMM_MAX_PTRS ? batch - b0 : GEMM_MAX_PTRS;
        GemmArgs g;
        for (int z = 0; z < nb; ++z) {
            const int b = b0 + z;
            const bool right = Ks[b] <= Ns[b];
            const bf16_t* op = (const bf16_t*)(right ? Et[b] : E[b]);
            const int64_t dx = right ? Ns[b] : Ks[b];
            g.set_problem(z, op, op, dx, (int)dx);
        }
        g.A = g.B = g.Ap0;
        g.lda = g.ldb = g.ldp0;
        g.Kd = g.Kdp0;
        g.M = g.N = (int)Dm;
        g.C16 = g.Ct16 = w.C16 + b0 * sC;
        g.ldc16 = g.ldct = Dm;
        g.sC16 = g.sCt = sC;
        g.sym = 1;
        if (nb > 1) {
            g.batch = g.nptr = nb;
        }
        CHK(slim_gemm_launch(g, stream));
    }
    int a = 0, ninfo = 0;
    float fr[FILTER_MAX];
    int fbits = 0;
    const bool filt = take_filter(batch, fr, &fbits);
    CHK(gram_iterate(w.B, w.C16, sC, batch, Dm, l, n_iter, seed, flags, cond, w, sk_slab, sSk, stream, &a, &ninfo, filt ? fr : nullptr,
                     fbits));
    // the projections Z^T = P^T E^T (or P^T E): members with one long dimension, un-split, as one launch (operands of the
    // members in separate allocations, outputs strided in the workspace); otherwise member by member
    const bool proj_batched = same_dx && !sk_slab && batch > 1 && batch <= GEMM_MAX_PTRS;
    if (proj_batched) {
        const int64_t Dx = Ks[0] <= Ns[0] ? Ns[0] : Ks[0];
        GemmArgs g;
        for (int z = 0; z < batch; ++z)
            g.set_problem(z, w.B[a].Xt + z * sP, (const bf16_t*)(Ks[z] <= Ns[z] ? E[z] : Et[z]), Dm, (int)Dm);
        g.A = g.Ap0;
        g.B = g.Bp0;
        g.lda = g.ldb = Dm;
        g.M = l;
        g.N = (int)Dx;
        g.Kd = (int)Dm;
        g.C16 = w.A[0].Xt;
        g.ldc16 = Dx;
        g.Ct16 = w.A[0].X;
        g.ldct = l;
        g.sC16 = g.sCt = sZ;
        g.batch = g.nptr = batch;
        CHK(slim_gemm_launch(g, stream));
    }
    for (int b = 0; b < batch; ++b) {
        const bool right = Ks[b] <= Ns[b];
        const int64_t Dx = right ? Ns[b] : Ks[b];
        FitWs m = member_ws(w, b, l);
        bf16_t* Z = w.A[0].X + b * sZ;
        bf16_t* Zt = w.A[0].Xt + b * sZ;
        if (!proj_batched)
            CHK(nt(1, w.B[a].Xt + b * sP, Dm, 0, (const bf16_t*)(right ? E[b] : Et[b]), Dm, 0, l, (int)Dx, (int)Dm, nullptr, Zt, Dx, Z, l,
                   stream, sk_slab ? sk_slab + b * sSk : nullptr, sSk));
        CHK(gram(Zt, Dx, l, m, nullptr, stream));
    }
    SLIM_HIP(hipMemcpyAsync(M_out, w.G, (size_t)batch * sG * sizeof(float), hipMemcpyDeviceToDevice, st));
    *state = a | (ninfo << 8) | ((cond & 0xff) << 16) | (((cond >> 8) & 0x3f) << 24) | FIT_STATE_GRAM;
    return SLIM_OK;
}

extern "C" int slim_lowrank_fit_back_group(const int64_t* Ns, const int64_t* Ks, int batch, int r, int l, const int32_t* state,
                                           const float* vecs, const float* evals, const int32_t* eig_info,
                                           const float* const* s, void* const* U, void* const* Ut, void* const* L, float* S,
                                           int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream) {
    int64_t Dm = 0, Dxm = 0;
    CHK(group_dims(Ns, Ks, batch, &Dm, &Dxm));
    SLIM_REQUIRE(state && vecs && evals && U && Ut && L && workspace, "slim_lowrank_fit_back_group: null pointer");
    SLIM_REQUIRE(r > 0 && l >= r && l < Dm && l % 8 == 0 && r % 8 == 0, "slim_lowrank_fit_back_group: bad sizes");
    SLIM_REQUIRE(workspace_bytes >= fit_ws(nullptr, batch, Dxm, Dm, l).total, "slim_lowrank_fit_back_group: workspace too small");
    SLIM_REQUIRE(*state & FIT_STATE_GRAM, "slim_lowrank_fit_back_group: bad state");
    hipStream_t st = (hipStream_t)stream;
    FitWs w = fit_ws((char*)workspace, batch, Dxm, Dm, l);
    const int inner = *state & ~FIT_STATE_GRAM;
    const int a = inner & 0xff, cond_e = ((inner >> 16) & 0xff) | (((inner >> 24) & 0x3f) << 8);
    int ninfo = (inner >> 8) & 0xff;
    SLIM_REQUIRE(a >= 0 && a < 3 && ninfo >= 0 && ninfo < SLIM_FIT_INFO_WORDS - 2, "slim_lowrank_fit_back_group: bad state");
    if (eig_info) {
        info_scatter_kernel<<<(batch + 63) / 64, 64, 0, st>>>(eig_info, batch, w.info, ninfo);
        SLIM_LAUNCH_CHECK();
    }
    ninfo += 1;
    cond_check_kernel<<<(batch + 63) / 64, 64, 0, st>>>(evals, l, r, batch, cond_e, w.info, w.scal);
    SLIM_LAUNCH_CHECK();
    const int64_t sP = Dm * l, sZ = Dxm * l, sG = (int64_t)l * l;
    const dim3 grid((unsigned)((l + 255) / 256), (unsigned)r, 1);
    for (int b = 0; b < batch; ++b) {
        SLIM_REQUIRE(U[b] && Ut[b] && L[b], "slim_lowrank_fit_back_group: null output");
        const int64_t N = Ns[b], K = Ks[b];
        const bool right = K <= N;
        const bf16_t* P = w.B[a].X + b * sP;
        const bf16_t* Z = w.A[0].X + b * sZ;
        bf16_t* T16 = w.T16 + b * sG;
        bf16_t* T16b = w.T16b + b * sG;
        float* Sb = S ? S + (int64_t)b * r : nullptr;
        if (right) {        // U = Z (W S^-1), L = P (W S)
            top_eigvecs_kernel<<<grid, 256, 0, st>>>(vecs + b * sG, evals + (int64_t)b * l, l, r, nullptr, Sb, T16, T16b);
            SLIM_LAUNCH_CHECK();
            CHK(nt(1, P, l, 0, T16b, l, 0, (int)K, r, l, nullptr, (bf16_t*)L[b], r, nullptr, 0, stream));
            CHK(nt(1, Z, l, 0, T16, l, 0, (int)N, r, l, nullptr, (bf16_t*)U[b], r, (bf16_t*)Ut[b], N, stream));
        } else {            // U = P W, L = Z W
            top_eigvecs_kernel<<<grid, 256, 0, st>>>(vecs + b * sG, evals + (int64_t)b * l, l, r, T16, Sb, nullptr, nullptr);
            SLIM_LAUNCH_CHECK();
            CHK(nt(1, Z, l, 0, T16, l, 0, (int)K, r, l, nullptr, (bf16_t*)L[b], r, nullptr, 0, stream));
            CHK(nt(1, P, l, 0, T16, l, 0, (int)N, r, l, nullptr, (bf16_t*)U[b], r, (bf16_t*)Ut[b], N, stream));
        }
        if (s && s[b]) CHK(scale_rows_batched(L[b], s[b], K, r, 1, stream));
    }
    if (info_out)
        SLIM_HIP(hipMemcpyAsync(info_out, w.info, (size_t)batch * SLIM_FIT_INFO_WORDS * sizeof(int), hipMemcpyDeviceToDevice, st));
    return SLIM_OK;
}

extern "C" int slim_lowrank_fit_front(const void* E, const void* Et, int64_t N, int64_t K, int r, int l, int n_iter,
                                      uint32_t seed, int flags, int ns_steps, float* M_out, int32_t* state, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
    return slim_lowrank_fit_front_batched(E, Et, 1, N, K, r, l, n_iter, seed, flags, ns_steps, M_out, state, workspace,
                                          workspace_bytes, stream);
}
extern "C" int slim_lowrank_fit_back(int64_t N, int64_t K, int r, int l, int n_iter, const int32_t* state, const float* vecs,
                                     const float* evals, const int32_t* eig_info, const float* s, void* U, void* Ut, void* L,
                                     float* S, int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream) {
    return slim_lowrank_fit_back_batched(1, N, K, r, l, n_iter, state, vecs, evals, eig_info, s, U, Ut, L, S, info_out,
                                         workspace, workspace_bytes, stream);
}

extern "C" int slim_lowrank_fit(const void* E, const void* Et, int64_t N, int64_t K, int r, int l, int n_iter,
                                uint32_t seed, int flags, int ns_steps, const float* s, void* U, void* Ut, void* L, float* S,
                                int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream) {
    return slim_lowrank_fit_batched(E, Et, 1, N, K, r, l, n_iter, seed, flags, ns_steps, s, U, Ut, L, S, info_out, workspace,
                                    workspace_bytes, stream);
}
