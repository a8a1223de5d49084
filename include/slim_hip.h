/*
 * slim_hip.h - C ABI of libslim_hip.so: the SLiM per-layer compression hot path as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless a comment says "host".  Buffers are
 *     caller-allocated, row-major, and never freed or retained by the library.
 *   - bf16 tensors are passed as `void*` (raw 16-bit patterns); weights are [N, K]
 *     (out_features x in_features), the layout of nn.Linear.weight.
 *   - `stream` is a hipStream_t (NULL = the default stream).  All work is enqueued on it;
 *     nothing synchronises the host.
 *   - Return value: 0 on success, negative on error (SLIM_ERR_*); slim_last_error() gives
 *     the message for the calling thread.  Nothing throws.
 *   - Reference citations are file:line in the sempfi/slim checkout this library replaces.
 */
#ifndef SLIM_HIP_H
#define SLIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIM_HIP_ABI_VERSION 1
#define SLIM_FIT_INFO_WORDS 64
/* info_out of the fits: words 0 .. 61 are status words of solver / orthonormalisation steps (0 = fine).  Word 63 is not
 * an error: 0, or -- when the singular values of E inside the fitted subspace turned out to be spread over more than the
 * orthonormalisations were sized for -- the ns_steps value (bits 0-7, no bits 8-15) that covers them; the fit should then
 * be repeated with it (its result may have lost accuracy in the weakest directions). */
#define SLIM_FIT_INFO_COND_WORD 63
/* Word 62 is not an error either: ceil(4 log2(largest / weakest eigenvalue that matters)) of the Gram matrix inside the
 * fitted subspace, as the fit found it.  A caller that fits a similar matrix next (the same linear of the next decoder
 * block) may pass ceil(word / 4 + 1) in bits 8-15 of ns_steps: the orthonormalisations after the first are then sized for
 * that measured spread instead of the a-priori estimate in bits 0-7, which usually saves one or two Newton-Schulz steps
 * each; word 63 reports when the hint was too low. */
#define SLIM_FIT_INFO_LEVEL_WORD 62
/* Word 61 is not an error either: round(2048 * cut / mean), cut the largest eigenvalue of the fitted subspace that the fit
 * drops, mean = trace / dimension of the Gram matrix of E: where the cut sits in the spectrum.  A caller that fits a similar
 * matrix next may hand about 0.9 x word / 2048 to slim_lowrank_fit_set_filter. */
#define SLIM_FIT_INFO_RATIO_WORD 61
#define SLIM_FIT_CHOLESKY_QR 1 /* slim_lowrank_fit flags: a comparison mode -- the alternating subspace iteration on E and E^T
                                * (two products with the N x K matrix and two orthonormalisations per iteration) with every
                                * basis orthonormalised by Cholesky-QR (rocSOLVER) instead of the iteration on the Gram
                                * matrix of E with Newton-Schulz orthonormalisations */
#define SLIM_FIT_NO_SKETCH_SPLIT 16 /* the caller runs several fits side by side: do not cut the sketch products along the
                                     * reduction (by default the ones that under-fill the chip are) and cut the Gram products
                                     * into at most two slices */
#define SLIM_FIT_LIBRARY_EIGH 4 /* Rayleigh-Ritz eigenproblem through rocSOLVER syevd instead of slim_eigh_f32's method */

int slim_abi_version(void);
const char* slim_last_error(void); /* host string, valid until the next failing call on this thread */

/* Optional timing of the library's own launches: slim_profile_enable(1) starts recording a HIP event
 * pair on the launch stream around each kernel family, slim_profile_summary writes
 * "name count total_ms" lines into a HOST buffer (it waits for the recorded events), and
 * slim_profile_enable(0) stops and clears.  No counterpart in the reference. */
int slim_profile_enable(int on);
int slim_profile_summary(char* buf_host, int64_t buflen);

/* ---- (1) calibration statistics -------------------------------------------------------- */

/* WrappedGPT.add_batch, slim/layerwrapper.py:25-41, for `n_chunks` consecutive add_batch calls
 * in one launch.  x is [n_chunks * rows_per_chunk, K] bf16 with row stride ld; chunk c is the
 * input of one call carrying `batch_per_chunk` samples.  Applies, in order, per chunk:
 *   scaler_row *= n / (n + b);  n += b;  scaler_row += ||x_chunk[:, col]||_2^2 / n
 * scaler_row: fp32 [K], updated in place.  nsamples_before: n before the first chunk (host int).
 * workspace: slim_colsumsq_workspace_bytes(...) bytes. */
int64_t slim_colsumsq_workspace_bytes(int64_t n_chunks, int64_t rows_per_chunk, int64_t K);
int slim_scaler_row_accum(const void* x, int64_t n_chunks, int64_t rows_per_chunk, int64_t K, int64_t ld,
                          int64_t batch_per_chunk, float* scaler_row, int64_t nsamples_before, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* The two halves of slim_scaler_row_accum as separate calls (sample-sharded multi-GPU runs; no counterpart in
 * the reference, which is single-GPU): sumsq fp32 [n_chunks, K] = per-chunk column sums of squares;
 * slim_scaler_row_fold applies the recurrence of layerwrapper.py:33-41 to such sums, chunk by chunk.
 * colsumsq + (all-gather into sample order) + fold is bit-identical to slim_scaler_row_accum on all samples. */
int slim_colsumsq(const void* x_bf16, int64_t n_chunks, int64_t rows_per_chunk, int64_t K, int64_t ld, float* sumsq,
                  void* workspace, int64_t workspace_bytes, void* stream);
int slim_scaler_row_fold(float* scaler_row, int64_t K, const float* sumsq, int64_t n_chunks, int64_t batch_per_chunk,
                         int64_t nsamples_before, void* stream);

/* SparseGPT.add_batch, slim/sparsegpt.py:28-39: H = H * n/(n+b) + (2/(n+b)) * X^T X, fp32 [K, K],
 * X = [T, K] bf16 (T = b * tokens).  MFMA bf16 products, fp32 accumulation. */
int64_t slim_gram_workspace_bytes(int64_t T, int64_t K);
int slim_gram_accum(const void* x, int64_t T, int64_t K, int64_t ld, float* H, int64_t nsamples_before,
                    int64_t batch, void* workspace, int64_t workspace_bytes, void* stream);

/* shift_zeros, slim/utils.py:64-72: out = in + min(in with zeros replaced by 1).  May alias. */
int slim_shift_zeros(const float* in, float* out, int64_t K, void* stream);
/* element-wise sqrt (torch.sqrt(scaler_row), slim/prune.py:264-265, slim/lora.py:146) */
int slim_sqrt_f32(const float* in, float* out, int64_t K, void* stream);
/* flag[0] = 1 if any element is zero (the `any(scaler_row == 0)` guard of slim/lora.py:139) */
int slim_any_zero_f32(const float* in, int64_t K, int32_t* flag, void* stream);
/* The scalars the driver needs from one scaler_row vector before it compresses the layer, without host round trips per
 * quantity: out_vals[0] = 1 if any entry is zero (the guard of slim/lora.py:139) else 0; out_vals[1] = the (lower) median
 * of sqrt(scaler_row); out_vals[2 .. 2 + top) = the `top` largest values of sqrt(scaler_row) in decreasing order (equal
 * values: lower index first; zero-filled past K) and out_idx[0 .. top) their indices.  top <= 128.  No counterpart in the
 * reference: it serves the outlier deflation and the conditioning estimate of the low-rank fit (slim_lowrank_fit). */
int slim_outlier_stats(const float* scaler_row, int64_t K, int top, float* out_vals, int32_t* out_idx, void* stream);

/* ---- (2) N:M mask ------------------------------------------------------------------------ */

/* Saliency + prune_nm fused: metric = |W| * s[col] (slim/prune.py:264-265, s = sqrt(scaler_row);
 * s = NULL: metric = |W|), then in every aligned group of m columns the n smallest metrics are
 * marked 1 = pruned (slim/utils.py:75-88), with the reference's tie behaviour.  mask: uint8 [N, K].
 * Implemented for m = 4, n in 1..3; other patterns return SLIM_ERR_UNSUPPORTED. */
int slim_mask_nm(const void* W_bf16, const float* s, int64_t N, int64_t K, int n, int m, uint8_t* mask,
                 void* stream);
/* prune_nm on an arbitrary fp32 metric matrix, slim/utils.py:75-88. */
int slim_prune_nm_f32(const float* metric, int64_t N, int64_t K, int n, int m, uint8_t* mask, void* stream);

/* Unstructured Wanda pruning, slim/prune.py:264-265 + :275-279: per row the k_prune = int(K * sparsity_ratio)
 * smallest saliencies |W| * s are marked (1 = pruned), ties to the lower index (stable sort), NaN last.
 * Rows of up to 40448 elements (the row's keys are held in LDS).  The _f32 form takes a ready metric. */
int slim_mask_unstructured(const void* W_bf16, const float* s, int64_t N, int64_t K, int64_t k_prune, uint8_t* mask,
                           void* stream);
int slim_prune_unstructured_f32(const float* metric, int64_t N, int64_t K, int64_t k_prune, uint8_t* mask, void* stream);
/* HOST: the 729-entry tie table of the 4-wide selection (index = base-3 digits of the six pairwise
 * comparisons 01,02,03,12,13,23 with 0: <, 1: ==, 2: >; value = bit i set if element i is selected). */
int slim_nm_lut4(int n, uint8_t* out729_host);

/* ---- (3) SLiM-Quant scale search and per-tensor quantisation ------------------------------- */

/* Exact histogram of |w| over the 32768 non-negative bf16 bit patterns; hist: uint32 [32768].
 * accumulate = 0 zeroes hist first.  Replaces the torch.histogram pass of quantization.py:36. */
int slim_hist_abs_bf16(const void* W_bf16, int64_t numel, uint32_t* hist, int accumulate, void* stream);

/* find_optimal_quantiztion_cap, slim/quantization/quantization.py:34-62 with compute_average_error
 * :6-20, from the pattern histogram: rebuilds the reference's density histogram over
 * [min|w|, max|w|] with num_bins bins, runs the coarse (every num_bins/10) and fine scans and
 * returns alpha = val[argmin] (device float), its bin index, and the scaling factor
 * sf = reciprocal(alpha) * (2^(bits-1)-1), which is how `max_q / abs_max` evaluates at :194.
 * idx_out, sf_out may be NULL. */
int64_t slim_quant_search_workspace_bytes(int num_bins);
int slim_quant_search(const uint32_t* hist, int num_bins, int num_bits, float* alpha_out, int32_t* idx_out,
                      float* sf_out, void* workspace, int64_t workspace_bytes, void* stream);
/* Self-check of the exact remainder the scale search's floor division (quantization.py:25, `//` on fp32 tensors) is built
 * on: out_fast[i] = the library's own remainder of a[i] / b[i] (a >= 0, b > 0), out_ref[i] = fmodf(a[i], b[i]); the two
 * must be equal bit for bit (tests/test_hip_kernels.py). */
int slim_fmod_probe_f32(const float* a, const float* b, int64_t n, float* out_fast, float* out_ref, void* stream);
/* Self-check of the floor division itself as the search evaluates it (no division, one fused multiply-add): out_fast[i] =
 * the library's floor(a[i] / b[i]), out_ref[i] = the reference's `//` written out step by step (remainder, subtraction,
 * division, half-way correction: quantization.py:25 through torch's floor_divide); equal bit for bit. */
int slim_floordiv_probe_f32(const float* a, const float* b, int64_t n, float* out_fast, float* out_ref, void* stream);
/* slim_quant_search for callers that cannot allocate or bring a workspace (HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The workspace is a 408 064-byte array inside the
 * library's code object (one copy per GPU, held until the process ends), enough for num_bins <= 20000; more bins return
 * SLIM_ERR_INVALID.  Calls no hipMalloc / hipFree.  The first call on a device resolves that device's copy of the array
 * (hipGetSymbolAddress) and sets a kernel attribute: make it outside a capture.  All calls on one GPU share the array:
 * calls that overlap on one GPU (different streams or threads) race; calls on one stream are ordered.  Results are
 * identical to slim_quant_search. */
int slim_quant_search_noalloc(const uint32_t* hist, int num_bins, int num_bits, float* alpha_out, int32_t* idx_out,
                              float* sf_out, void* stream);
/* abs-max scale (slim_quant = False, quantization.py:190-194): alpha = max|w| (bf16 value),
 * sf = bf16(bf16(1 / alpha) * maxq) widened to fp32. */
int slim_absmax_scale(const uint32_t* hist, int num_bits, float* alpha_out, float* sf_out, void* stream);

/* Quantizer.quantize_block, quantization.py:194-200:
 *   codes = clamp(round_half_even(float(bf16(float(w) * sf))), -2^(b-1), 2^(b-1)-1) as int8. */
int slim_quantize_pt(const void* W_bf16, int64_t numel, const float* sf, int num_bits, int8_t* codes, void* stream);
/* Quantizer.dequantize_absmax (1,1) branch, quantization.py:211-212: float(code) / sf, as fp32
 * (out_bf16 = 0) or rounded to bf16 (out_bf16 = 1). */
int slim_dequantize_pt(const int8_t* codes, int64_t numel, const float* sf, void* out, int out_bf16, void* stream);
/* quantise + dequantise + bf16 cast in one pass (the weight written back at slim/lora.py:203-209);
 * codes may be NULL. */
int slim_fake_quant_pt(const void* W_bf16, int64_t numel, const float* sf, int num_bits, void* out_bf16,
                       int8_t* codes, void* stream);

/* ---- (4) low-rank fit of the compression error ----------------------------------------------- */

/* Error matrix of add_lora, slim/lora.py:146-156 (s != NULL) or :158-169 (s = NULL):
 *   E = float(W) * s - mask0((code / sf) * s);   sf = NULL means "no quantiser" (:141-144).
 * Written as fp32 and/or bf16 (either pointer may be NULL). */
int slim_error_matrix(const void* W_bf16, const uint8_t* mask, const float* s, const float* sf, int num_bits,
                      int64_t N, int64_t K, float* E_f32, void* E_bf16, void* stream);
/* Same with the dequantised weight given as a bf16 tensor (tiled group quantiser, lora.py:149-150 with
 * block_quantization): E = W * s - mask0(Wdeq * s). */
int slim_error_matrix_deq(const void* W_bf16, const void* Wdeq_bf16, const uint8_t* mask, const float* s, int64_t N,
                          int64_t K, float* E_f32, void* E_bf16, void* stream);

/* scale_important_weights, slim/lora.py:132-136 (and slim/prune.py:331-335): flags[c] = 1 for the `count` input channels
 * with the largest scaler_row[c] * bf16(sum_rows |W[:, c]|) (the reference's column sums are a bf16 tensor), 0 elsewhere;
 * equal metrics at the boundary: lower index first.  K a multiple of 8; workspace from the query. */
int64_t slim_important_columns_workspace_bytes(int64_t N, int64_t K);
int slim_important_columns(const void* W_bf16, const float* scaler_row, int64_t N, int64_t K, int64_t count, uint8_t* flags,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* Quantizer.scale_important_columns, quantization.py:160-170: out = bf16(float(in) * factor) in the flagged columns of
 * in [N, K], a copy elsewhere (factor 2 before the codes are taken, 0.5 after dequantisation); out == in allowed. */
int slim_scale_columns_bf16(const void* in, const uint8_t* flags, int64_t N, int64_t K, float factor, void* out, void* stream);
/* The same on an fp32 matrix in place (the per-tensor quantiser's dequantised values, quantization.py:212-215). */
int slim_scale_columns_f32(float* x, const uint8_t* flags, int64_t N, int64_t K, float factor, void* stream);
/* slim_error_matrix in its general form (slim/lora.py:149-156 with quantization.py:156, :192, :212-215).
 * col_flags (NULL: none): scale_important_weights -- in the flagged columns the code is that of the doubled weight and the
 * dequantised value is halved, E = W * s - mask0(((code / sf) * 0.5) * s).
 * deq_bf16 != 0: code / sf is rounded to bf16 before it is multiplied by s -- the abs-max quantiser's scale is a bf16
 * (1, 1) tensor in the reference (quantization.py:190-194), which makes `quantized_mat / scaling_factor` a bf16 tensor;
 * SLiM-Quant scales are fp32 and leave fp32 values (deq_bf16 = 0, what slim_error_matrix computes). */
int slim_error_matrix_cols(const void* W_bf16, const uint8_t* mask, const float* s, const float* sf, int num_bits,
                           const uint8_t* col_flags, int deq_bf16, int64_t N, int64_t K, float* E_f32, void* E_bf16,
                           void* stream);

/* slim_error_matrix (per-tensor quantiser or none, bf16 output) and the transpose of its result in one pass: E [N, K] and
 * Et [K, N]; N and K multiples of 8. */
int slim_error_matrix_t(const void* W_bf16, const uint8_t* mask, const float* s, const float* sf, int num_bits, int64_t N,
                        int64_t K, void* E_bf16, void* Et_bf16, void* stream);
/* The same with the dequantised values rounded to bf16 when deq_bf16 != 0 (abs-max quantiser, see slim_error_matrix_cols). */
int slim_error_matrix_t_rounded(const void* W_bf16, const uint8_t* mask, const float* s, const float* sf, int num_bits,
                                int deq_bf16, int64_t N, int64_t K, void* E_bf16, void* Et_bf16, void* stream);
int slim_transpose_bf16(const void* in, int64_t R, int64_t C, void* out, void* stream);

/* C[M, N] = alpha * A[M, Kd] * Bt[N, Kd]^T + beta * C_f32, bf16 operands (reduction dimension
 * contiguous), fp32 accumulation on MFMA.  Any of C_f32 [M, N], C_bf16 [M, N], Ct_bf16 [N, M] may be
 * NULL; beta != 0 requires C_f32. */
int slim_gemm_nt_bf16(const void* A, int64_t lda, const void* Bt, int64_t ldb, int M, int N, int Kd, float alpha,
                      float beta, float* C_f32, int64_t ldc32, void* C_bf16, int64_t ldc16, void* Ct_bf16,
                      int64_t ldct, void* stream);

/* Same product with the reduction split into `splits` slices whose fp32 partial tiles go to
 * slab [splits, M, N] and are summed in slice order (deterministic).  For small outputs with a long
 * reduction (the l x l Gram matrices of the fit). */
int slim_gemm_nt_bf16_splitk(const void* A, int64_t lda, const void* Bt, int64_t ldb, int M, int N, int Kd,
                             float alpha, float* C_f32, int64_t ldc32, void* C_bf16, int64_t ldc16, int splits,
                             float* slab, void* stream);
/* the same with C_bf16 (optional) and its transpose Ct_bf16 [N, M] as outputs; slab holds splits * M * N floats */
int slim_gemm_nt_bf16_splitk_t(const void* A, int64_t lda, const void* Bt, int64_t ldb, int M, int N, int Kd,
                               float alpha, void* C_bf16, int64_t ldc16, void* Ct_bf16, int64_t ldct, int splits,
                               float* slab, void* stream);

/* Adapter pruning, the optimisation loop of prune_and_optimize_lora (slim/lora.py:28-50): num_iters Adam steps
 * (torch defaults; lr decaying linearly to lr * lr_end_factor) on fp32 L [K, r] (zero where L_mask is set and kept so)
 * and R [r, N] for || L R - T ||_F / || T ||_F, T [K, N] fp32.  Everything is enqueued on `stream`, nothing
 * synchronises; losses (num_iters floats on the device, or NULL) receives the loss each iteration saw. */
int64_t slim_lora_prune_workspace_bytes(int64_t K, int64_t r, int64_t N);
int slim_lora_prune_optimize(float* L, float* R, const uint8_t* L_mask, const float* T, int64_t K, int64_t r, int64_t N,
                             int num_iters, float lr, float lr_end_factor, float* losses, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Tuning aid: force the workgroup tile of every following GEMM launch in this process (0: 128x128, 1: 128x64,
 * 2: 64x128, 3: 64x64; anything else: the library's choice, 128x128, the default).  Results do not depend on the tile. */
void slim_gemm_set_tile(int tile);
/* The same for one shape family (0: small output with a long reduction, 1: l x l x l, 2: tall x (l x l),
 * 3: products with the error matrix, 4: large square, 5: residual); -1 restores the library's choice. */
void slim_gemm_set_family_tile(int family, int tile);
/* Tuning aid: while buf (device memory, 8 * capacity_workgroups int64) is set, every GEMM launch records, for each
 * workgroup below the capacity, 100 MHz timestamps of kernel entry, first tile staged, K-loop done and epilogue done.
 * NULL switches it off (the default). */
void slim_gemm_set_trace(void* buf, int capacity_workgroups);

/* Eigenvalues (ascending) and the `top` leading eigenvectors of a symmetric fp32 matrix G [n, n] (both triangles
 * given, destroyed), n a multiple of 4 in [4, 2048].  On return row n-1-i of G is the eigenvector of the i-th
 * largest eigenvalue for i < top; info[0] = status of the tridiagonal solver (0 = converged).  This is the
 * Rayleigh-Ritz step of slim_lowrank_fit (the part of torch.svd at slim/lora.py:172 that remains after the
 * subspace iteration): Householder tridiagonalisation with one launch per column, then the tridiagonal problem in double
 * precision (see slim_eigh_tail_f32), one fp32 GEMM. */
int64_t slim_eigh_workspace_bytes(int n);
/* The first stage alone (no library call): Householder reduction Q^T A Q = tridiag(d, e) of a symmetric fp32 matrix
 * A [n, n] (destroyed), one launch per column; Q [n, n] row-major, e holds n - 1 values, workspace of 5 n floats. */
int slim_tridiagonalize_f32(float* A, int n, float* d, float* e, float* Q, void* workspace, int64_t workspace_bytes,
                            void* stream);
/* `batch` reductions of equal size in ONE launch chain: A [batch][n, n], d / e [batch][n], Q [batch][n, n], workspace of
 * 5 n floats per problem.  Several layers' fits share it (see slim_lowrank_fit_front / _back). */
int slim_tridiagonalize_f32_batched(float* A, int n, int batch, float* d, float* e, float* Q, void* workspace,
                                    int64_t workspace_bytes, void* stream);
/* second stage: the eigenvalues of tridiag(d, e) by multisection on Sturm counts and the wanted eigenvectors by twisted
 * factorisations, both in double precision, then Q times them on the fp32 matrix cores.  d becomes the eigenvalues,
 * ascending; rows n-top .. n-1 of vecs [n, n] = eigenvectors of the `top` largest eigenvalues of the original matrix.
 * info: device status word per problem (0 = fine).  The batched form takes d / e [batch][n], Q / vecs [batch][n, n],
 * info [batch] and solves all problems in three launches. */
int64_t slim_eigh_tail_workspace_bytes(int n, int top, int batch);
/* The same job for the fit without an eigen-decomposition (no reference counterpart beyond the truncation of
 * slim/lora.py:174-185): from M [batch][l, l] (symmetric positive semi-definite fp32, left intact) an orthonormal basis of
 * the invariant subspace of the keep[b] largest eigenvalues, found as the orthogonal complement of the l - keep[b]
 * smallest (Chebyshev-filtered subspace iteration in `n_rounds` rounds of degrees[] steps, Householder QR between; about
 * sixty launches for three rounds).  Outputs in the layout of slim_eigh_tail_f32_batched: row l-1-j of vecs [batch][l, l]
 * = j-th basis vector for j < keep[b], zero rows for keep[b] <= j < top; evals [batch][l]: evals[l-1-j] = its Rayleigh
 * quotient, except evals[l-1] >= an estimate of the largest eigenvalue, evals[(l-top)/2] = mean Rayleigh quotient of
 * the dropped block and evals[l-top-1] = estimate of the largest dropped eigenvalue (the values the fit's conditioning
 * check reads).  keep / degrees: host arrays.  info [batch]:
 * 1 = a non-finite quotient.  Limits (slim_ritz_split_supported): l <= 576, a multiple of 8; at most 80 directions
 * dropped; 1 <= keep[b] <= top, top a multiple of 4. */
/* Chebyshev-filtered subspace iteration for the NEXT slim_lowrank_fit_front / _front_batched / _front_group call on the
 * calling thread (the request is consumed by it; other threads are not affected): ratios[b] > 0 per problem of that call,
 * beta_b = ratios[b] * trace(C_b) / Dm the upper end of the interval [0, beta_b] of eigenvalues of C_b = E^T E (or E E^T)
 * that the filter damps -- it must lie below the eigenvalues the fit is to keep (status word SLIM_FIT_INFO_RATIO_WORD of a
 * similar fit says where they end; word SLIM_FIT_INFO_COND_WORD asks for a repeat when beta turned out too high).  The
 * call then runs (n_iter + 1) / 2 rounds of [degree-2 Chebyshev polynomial of C, orthonormalisation sized for 2^ns_bits]
 * instead of n_iter + 1 rounds of [C, orthonormalisation].  count = 0 withdraws a request. */
int slim_lowrank_fit_set_filter(const float* ratios, int count, int ns_bits);
int slim_ritz_split_supported(int l, int drop_max);
int64_t slim_ritz_split_workspace_bytes(int l, int drop_max, int batch);
int slim_ritz_split_f32_batched(const float* M, int l, int batch, int top, const int32_t* keep, const int32_t* degrees,
                                int n_rounds, float* vecs, float* evals, int32_t* info, void* workspace,
                                int64_t workspace_bytes, void* stream);
int slim_eigh_tail_f32(float* d, float* e, const float* Q, int n, int top, float* vecs, int32_t* info, void* workspace,
                       int64_t workspace_bytes, void* stream);
int slim_eigh_tail_f32_batched(float* d, float* e, const float* Q, int n, int batch, int top, float* vecs, int32_t* info,
                               void* workspace, int64_t workspace_bytes, void* stream);
/* C [M, N] (fp32, row stride ldc) = alpha * a b on the fp32 matrix cores, a [M, Kd], b [Kd, N]; a_kmajor: A is stored
 * [Kd, M] (row stride lda), else [M, Kd]; b_kmajor: B is stored [Kd, N] (row stride ldb), else [N, Kd].  Sizes and strides
 * multiples of 4.  The eigen-solver's back-product; no counterpart in the reference (inside torch.svd, slim/lora.py:172). */
int slim_gemm_f32(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, int M, int N, int Kd,
                  float alpha, float* C, int64_t ldc, void* stream);
int slim_eigh_f32(float* G, int n, int top, float* evals, int32_t* info, void* workspace, int64_t workspace_bytes,
                  void* stream);

/* Orthonormal basis Q [rows, l] (and Q^T) of the column space of X [rows, l] (Xt = X^T), either by
 * ns_quintic + ns_cubic Newton-Schulz steps on MFMA or (flags & SLIM_FIT_CHOLESKY_QR) by Cholesky-QR.
 * Building block of slim_lowrank_fit; also used to deflate outlier columns before the fit. */
int64_t slim_orthonormalize_workspace_bytes(int64_t rows, int64_t l);
int slim_orthonormalize_bf16(const void* X, const void* Xt, int64_t rows, int l, int flags, int ns_quintic,
                             int ns_cubic, void* Q, void* Qt, int32_t* info_out, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Replaces torch.svd + truncation, slim/lora.py:172-185 (+ the bf16 division by sqrt(scaler_row)
 * of :193-195 when s != NULL).  Randomized subspace iteration: l = r + oversampling columns,
 * n_iter power iterations.
 *   E [N, K], Et [K, N] bf16;  outputs U [N, r], Ut [r, N] (= lora_right, may be NULL),
 *   L [K, r] (= lora_left), S fp32 [r] (may be NULL), info_out int32 [SLIM_FIT_INFO_WORDS]
 *   (rocSOLVER status words, all zero on success; may be NULL).
 *   flags: SLIM_FIT_*; ns_steps bits 0-7: log2 of the conditioning (largest / smallest singular value of E inside the
 *   subspace) the Newton-Schulz orthonormalisations are sized for, 1 .. 12 (0 = 9): it selects the coefficient schedule
 *   and the number of steps; bits 8-15 (optional): see SLIM_FIT_INFO_LEVEL_WORD. */
int64_t slim_lowrank_fit_workspace_bytes(int64_t N, int64_t K, int64_t l);
int slim_lowrank_fit(const void* E, const void* Et, int64_t N, int64_t K, int r, int l, int n_iter, uint32_t seed,
                     int flags, int ns_steps, const float* s, void* U, void* Ut, void* L, float* S, int32_t* info_out,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* `batch` fits of identical shape in lockstep (the q/k/v/o or gate/up projections of one decoder block):
 * E [batch][N, K], Et [batch][K, N], s [batch][K] or NULL, U [batch][N, r], Ut [batch][r, N], L [batch][K, r],
 * S [batch][r], info_out [batch][SLIM_FIT_INFO_WORDS].  Every launch covers the whole batch, so the launch
 * count and the dependent chain (in particular the l x l eigen-solve) are those of one fit. */
int64_t slim_lowrank_fit_batched_workspace_bytes(int batch, int64_t N, int64_t K, int64_t l);
int slim_lowrank_fit_batched(const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l, int n_iter,
                             uint32_t seed, int flags, int ns_steps, const float* s, void* U, void* Ut, void* L,
                             float* S, int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream);
/* The fit in two halves around its l x l eigenproblem (no reference counterpart: the reference solves one torch.svd per
 * layer, slim/lora.py:172).  front: up to the Rayleigh-Ritz matrix, written to M_out [l, l] fp32; *state (host int) goes
 * to back unchanged, with the same workspace (slim_lowrank_fit_workspace_bytes) and sizes.  back: vecs [l, l] (row
 * l-1-i = eigenvector of the i-th largest eigenvalue, i < r), evals [l] ascending, eig_info (device status word or
 * NULL) -> U, Ut, L, S as slim_lowrank_fit returns them. */
int slim_lowrank_fit_front(const void* E, const void* Et, int64_t N, int64_t K, int r, int l, int n_iter, uint32_t seed,
                           int flags, int ns_steps, float* M_out, int32_t* state, void* workspace, int64_t workspace_bytes,
                           void* stream);
int slim_lowrank_fit_back(int64_t N, int64_t K, int r, int l, int n_iter, const int32_t* state, const float* vecs,
                          const float* evals, const int32_t* eig_info, const float* s, void* U, void* Ut, void* L, float* S,
                          int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream);
/* The two halves for `batch` same-shape fits in lockstep (operands strided as in slim_lowrank_fit_batched, workspace
 * slim_lowrank_fit_batched_workspace_bytes; M_out and vecs [batch][l, l], evals [batch][l], eig_info [batch]). */
int slim_lowrank_fit_front_batched(const void* E, const void* Et, int batch, int64_t N, int64_t K, int r, int l, int n_iter,
                                   uint32_t seed, int flags, int ns_steps, float* M_out, int32_t* state, void* workspace,
                                   int64_t workspace_bytes, void* stream);
int slim_lowrank_fit_back_batched(int batch, int64_t N, int64_t K, int r, int l, int n_iter, const int32_t* state,
                                  const float* vecs, const float* evals, const int32_t* eig_info, const float* s, void* U,
                                  void* Ut, void* L, float* S, int32_t* info_out, void* workspace, int64_t workspace_bytes,
                                  void* stream);

/* The two halves for a lockstep group of fits whose shapes differ but whose short dimension Dm = min(N, K), l and r agree
 * (all seven linears of a LLaMA-2-7B block: 4096 x 4096, 11008 x 4096, 4096 x 11008).  No reference counterpart (one
 * torch.svd per layer, slim/lora.py:172).  The Gram-side subspace iteration sees only the Dm x Dm matrix, so it runs as ONE
 * batched chain; what touches the N x K matrices runs member by member; every member's result is bit-identical to
 * slim_lowrank_fit_front / _back on it alone.  E, Et, s, U, Ut, L: HOST arrays of `batch` device pointers (s, or single
 * entries of it, may be NULL); Ns, Ks: host arrays of the shapes; M_out / vecs [batch][l, l], evals [batch][l], eig_info
 * [batch] or NULL, S [batch][r] or NULL, info_out [batch][SLIM_FIT_INFO_WORDS].  flags: SLIM_FIT_NO_SKETCH_SPLIT only
 * (the other modes do not run as a mixed-shape group). */
int64_t slim_lowrank_fit_group_workspace_bytes(const int64_t* Ns, const int64_t* Ks, int batch, int64_t l);
int slim_lowrank_fit_front_group(const void* const* E, const void* const* Et, const int64_t* Ns, const int64_t* Ks, int batch,
                                 int r, int l, int n_iter, uint32_t seed, int flags, int ns_steps, float* M_out,
                                 int32_t* state, void* workspace, int64_t workspace_bytes, void* stream);
int slim_lowrank_fit_back_group(const int64_t* Ns, const int64_t* Ks, int batch, int r, int l, const int32_t* state,
                                const float* vecs, const float* evals, const int32_t* eig_info, const float* const* s,
                                void* const* U, void* const* Ut, void* const* L, float* S, int32_t* info_out, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* The pieces of the fit for a layer that several GPUs hold in shards along its long dimension (no counterpart in the
 * reference, which is single-GPU; what they compute is part of the torch.svd of slim/lora.py:172).  With Dm the short
 * dimension of E: (1) every GPU forms the Dm x Dm Gram matrix of its shard (slim_gemm_nt_bf16, fp32 output) and the
 * shards' sum goes to one GPU; (2) slim_gramfit_iterate: subspace iteration on that matrix (bf16 [Dm, Dm]) -> orthonormal
 * basis P [Dm, l] and P^T; (3) slim_gramfit_project: Z = X P [rows, l] (+ Z^T) for the shard X [rows, Dm] of E or E^T, and
 * the shard's share M32 [l, l] = Z^T Z of the Rayleigh-Ritz matrix; (4) after the eigenproblem of the summed M
 * (slim_tridiagonalize_f32 + slim_eigh_tail_f32), slim_gramfit_factors: right != 0 (Dm = K): dm_out = lora_left = P W S
 * [K, r], rows_out = this shard's rows of U = Z W S^-1; right == 0 (Dm = N): dm_out = U = P W [N, r], rows_out = this
 * shard's rows of lora_left = Z W.  s: saliency scales of lora_left's rows (it is divided by bf16(s)); the transposed
 * outputs are optional.  info (optional, the 64 status words of slim_gramfit_iterate): word SLIM_FIT_INFO_COND_WORD receives
 * the conditioning check for an iteration sized for cond_e, as in slim_lowrank_fit.  Workspaces:
 * slim_gramfit_workspace_bytes(Dm or rows, l); factors: 2 * (l * l * 2 rounded up to 256). */
int64_t slim_gramfit_workspace_bytes(int64_t rows, int64_t l);
int slim_gramfit_iterate(const void* C16, int64_t Dm, int l, int n_iter, uint32_t seed, int flags, int cond_e, void* P, void* Pt,
                         int32_t* info_out, void* workspace, int64_t workspace_bytes, void* stream);
/* slim_gramfit_iterate for `batch` Gram matrices of one size in lockstep (C16 [batch][Dm, Dm], P [batch][Dm, l], Pt [batch][l, Dm],
 * info_out [batch][SLIM_FIT_INFO_WORDS]).  Honours a request left by slim_lowrank_fit_set_filter on the calling thread (one
 * ratio per matrix): the filtered iteration shifts the diagonals of C16 in place.  scal_out (optional, [batch][4] floats) is
 * what slim_gramfit_check takes.  No counterpart in the reference (single-GPU, full SVD: slim/lora.py:172). */
int64_t slim_gramfit_batched_workspace_bytes(int batch, int64_t rows, int64_t l);
int slim_gramfit_iterate_batched(void* C16, int batch, int64_t Dm, int l, int n_iter, uint32_t seed, int flags, int cond_e, void* P,
                                 void* Pt, int32_t* info_out, float* scal_out, void* workspace, int64_t workspace_bytes,
                                 void* stream);
/* The three report words of a finished fit (SLIM_FIT_INFO_COND_WORD / _LEVEL_WORD / _RATIO_WORD of info
 * [batch][SLIM_FIT_INFO_WORDS]) from the eigenvalues of the Rayleigh-Ritz matrices (evals [batch][l], ascending); scal: the
 * scal_out of slim_gramfit_iterate_batched, or NULL. */
int slim_gramfit_check(const float* evals, int l, int r, int batch, int cond_e, const float* scal, int32_t* info, void* stream);

int slim_gramfit_project(const void* X, int64_t rows, int64_t Dm, int l, const void* Pt, void* Z, void* Zt, float* M32,
                         void* workspace, int64_t workspace_bytes, void* stream);
int slim_gramfit_factors(int right, const void* P, const void* Z, int64_t Dm, int64_t rows, int l, int r, const float* vecs,
                         const float* evals, const float* s, void* dm_out, void* dm_out_t, void* rows_out, void* rows_out_t,
                         float* S, int cond_e, int32_t* info, void* workspace, int64_t workspace_bytes, void* stream);

/* slim/lora.py:199-201: Wn = mask ? 0 : bf16(W - bf16(U L^T)); U [N, r], L [K, r].  LR_bf16 (the
 * N x K low-rank product itself) is written too when non-NULL; mask may be NULL (nothing zeroed). */
int slim_lowrank_residual(const void* U, const void* L, int N, int K, int r, const void* W_bf16,
                          const uint8_t* mask, void* Wn_bf16, void* LR_bf16, void* stream);
/* The tail of add_lora, slim/lora.py:199-209, with the pre-scaling of the adapters of slim/prune.py:296-301, for separate
 * adapters and a per-tensor quantiser, in one call: Wn = mask ? 0 : bf16(W - bf16(U L^T)) (U [N, r8], L [K, r8], columns
 * >= rank zero); sf = the scale of Wn (quant_mode 1: SLiM-Quant search with max(512, min(N K / 1000, 20000)) bins,
 * quantization.py:187; 2: abs-max; 0: no quantiser, sf_out untouched); W_out = deq(quant(Wn, sf)) in bf16;
 * lora_left_out [K, rank] = L[:, :rank] * c, lora_right_out [rank, N] = U[:, :rank]^T * c, c = sqrt(rank) when prescale != 0
 * (what the driver stores, prune.py:298-300), else 1 (what add_lora itself leaves). */
int64_t slim_lora_finish_workspace_bytes(int64_t N, int64_t K);
int slim_lora_finish(const void* U, const void* L, int64_t N, int64_t K, int r8, int rank, const void* W_bf16,
                     const uint8_t* mask, int quant_mode, int num_bits, int prescale, void* W_out_bf16, float* sf_out,
                     void* lora_left_out, void* lora_right_out, void* workspace, int64_t workspace_bytes, void* stream);
/* The last step of slim_lora_finish alone (slim/prune.py:296-301 applied to the factors of slim/lora.py:206-209): lora_left_out
 * [K, rank] = L[:, :rank] * c, lora_right_out [rank, N] = (U[:, :rank])^T * c, c = sqrt(rank) when prescale != 0, else 1. */
int slim_lora_adapters(const void* U, const void* L, int64_t N, int64_t K, int r8, int rank, int prescale, void* lora_left_out,
                       void* lora_right_out, void* stream);
/* The head of add_lora with the Wanda mask before it, slim/prune.py:264-272 + slim/lora.py:139-156, for a per-tensor
 * quantiser (or none) and quantize_first, in one call: s_out [K] = sqrt(scaler_row); mask [N, K] = prune_nm(|W| * s, prune_n,
 * prune_m) when prune_n > 0 (else `mask` is read: a precomputed mask, prune.py:269-271); sf_out = the scale of W (quant_mode
 * as in slim_lora_finish); E, Et = the bf16 error matrix W s' - mask0(deq(quant(W)) s') and its transpose (s' = s when
 * scale_error != 0: the `any(scaler_row == 0)` guard of lora.py:139 is the caller's).  No counterpart in the reference for
 * the rest: when t > 0 outlier columns of E are listed (outlier_idx, tp entries, tp a multiple of 8 and at most 64, the
 * first t real), UO [N, tp] / UOt [tp, N] receive an orthonormal basis of those columns (columns >= t zero), Ct [K, tp] =
 * (UO^T E)^T, E2 = E - UO Ct^T and E2t its transpose: what the randomized fit of slim_lowrank_fit* then runs on.
 * slim_outlier_basis is the basis step alone: the columns cols[0 .. t) of E are rows of Et [K, N]; the basis is E_O D^-1 R^-1
 * with D the column norms and R the Cholesky factor of the normalised columns' Gram matrix (double precision). */
/* Factors of a layer whose t outlier columns were split off: U [N, r8] = [UO[:, :t] | fitU[:, :rank - t] | 0], Ut its
 * transpose, L [K, r8] = [Ct[:, :t] / bf16(s) row-wise (s may be NULL) | fitL[:, :rank - t] | 0].  No counterpart in the
 * reference. */
int slim_lora_assemble(const void* UO, const void* Ct, const float* s, const void* fitU, const void* fitL, int64_t N, int64_t K,
                       int tp, int t, int rf, int rank, int r8, void* U, void* Ut, void* L, void* stream);
int64_t slim_outlier_basis_workspace_bytes(int tp);
int slim_outlier_basis(const void* Et_bf16, int64_t N, const int32_t* cols, int t, int tp, void* UO, void* UOt, void* workspace,
                       int64_t workspace_bytes, void* stream);
/* Ct [K, tp] = bf16(Et UOt^T) (fp32 accumulation): every column of E projected on the basis of slim_outlier_basis. */
int slim_outlier_project(const void* Et_bf16, const void* UOt, int64_t K, int64_t N, int tp, void* Ct, void* stream);
/* E2 [N, K] = bf16(E - bf16(UO Ct^T)) and E2t = its transpose, in one pass (the rounding points of slim_lowrank_residual; the
 * fp32 sums over the tp <= 64 columns run in column order). */
int slim_deflate_apply(const void* E_bf16, const void* UO, const void* Ct, int64_t N, int64_t K, int tp, void* E2_bf16,
                       void* E2t_bf16, void* stream);
int64_t slim_lora_prepare_workspace_bytes(int64_t N, int64_t K, int tp);
int slim_lora_prepare(const void* W_bf16, const float* scaler_row, int64_t N, int64_t K, int prune_n, int prune_m, uint8_t* mask,
                      int quant_mode, int num_bits, int scale_error, float* s_out, float* sf_out, void* E_bf16, void* Et_bf16,
                      const int32_t* outlier_idx, int t, int tp, void* UO, void* UOt, void* Ct, void* E2_bf16, void* E2t_bf16,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* Same subtraction from a materialised LR, optionally accumulating the |bf16| histogram of Wn. */
int slim_residual_weight(const void* W_bf16, const void* LR_bf16, const uint8_t* mask, int64_t numel, void* Wn_bf16,
                         uint32_t* hist, void* stream);

int slim_randn_bf16(void* out, int64_t n, uint32_t seed, void* stream);
int slim_scale_rows_bf16(void* L, const float* s, int64_t K, int64_t r, void* stream);

/* ---- (5) group quantisation of the adapters ---------------------------------------------------- */

/* compute_quantization_params(x, block, 1, symmetric=True), slim/quantization/utils.py:254-271:
 * alphas bf16 [M / block, N] = max |x| over each block of rows, per column. */
int slim_group_absmax(const void* x_bf16, int64_t M, int64_t N, int block, void* alphas_bf16, void* stream);
/* quantize_tensor, utils.py:123-128 (+ :12-31): int8 codes, round half toward zero. */
int slim_group_quantize(const void* x_bf16, const void* alphas_bf16, int64_t M, int64_t N, int block, int bits,
                        int8_t* codes, void* stream);
/* dequantize_tensor, utils.py:188-193 (+ :34-47): bf16 output. */
int slim_group_dequantize(const int8_t* codes, const void* alphas_bf16, int64_t M, int64_t N, int block, int bits,
                          void* out_bf16, void* stream);
/* The three above in one pass (quantize_lora, slim/lora.py:91-97; QuantizedMatmul.forward,
 * quantization.py:313-317); alphas_bf16 and codes may be NULL. */
int slim_group_fake_quant(const void* x_bf16, int64_t M, int64_t N, int block, int bits, void* out_bf16,
                          void* alphas_bf16, int8_t* codes, void* stream);

/* ---- (5) activation fake-quantisation of the input hooks ------------------------------------------ */

/* 8-bit branch of input_quantization_pre_hook, slim/quantization/quantization.py:286-290: per-tensor
 * abs-max, e4m3fn if max < 488 else e5m2, every intermediate rounded to bf16 as the reference does.
 * absmax_scratch: one uint32 of device scratch.  out may alias x. */
int slim_input_fake_quant_fp8(const void* x_bf16, int64_t numel, void* out_bf16, uint32_t* absmax_scratch,
                              void* stream);
/* Quantizer.quantize_input + dequantize_input (symmetric), quantization.py:219-263: x viewed as
 * [n_groups, group]; per-group abs-max scale, round-half-even, clamp, dequantise; bf16 in, bf16 out. */
int slim_input_fake_quant_int(const void* x_bf16, int64_t n_groups, int group, int num_bits, float eps,
                              void* out_bf16, void* stream);

/* ---- (6) forward pass of a compressed Linear from packed storage ----------------------------------- */

/* Packed form of the weight the compression leaves in `weight.data` (2:4 sparse, 4-bit, per-tensor
 * scale; slim/prune.py:281-301 + slim/lora.py:199-204, which the reference keeps as a dense bf16 tensor):
 * per group of 4 input channels 8 bits of codes and 4 bits of positions, 0.375 byte per dense element,
 * stored in tiles of 128 rows x 64 channels.  N % 128 == 0, K % 64 == 0. */
int slim_packed24_bytes(int64_t N, int64_t K, int64_t* vals_bytes, int64_t* meta_bytes);
/* W bf16 [N, K] with W == bf16(c / sf), c integer in [-8, 7], at most two non-zeros per group of 4.
 * status (device int32): 0, or bit 0 = a group with more than two non-zeros, bit 1 = a value that is not
 * a quantisation level of sf.  sf: device fp32 scalar (Quantizer.scaling_factor). */
int slim_pack24_int4(const void* W_bf16, const float* sf, int64_t N, int64_t K, void* vals, void* meta,
                     int32_t* status, void* stream);
/* Exact inverse (a dequantised -0 comes back as +0). */
int slim_unpack24_int4(const void* vals, const void* meta, const float* sf, int64_t N, int64_t K, void* W_bf16,
                       void* stream);
/* nn.Linear forward + add_lora_hook (slim/prune.py:303-321) from the packed weight (two launches; from 17 rows on two
 * more that divide by a_div once per element):
 *   t = bf16( bf16(x / a_div) L_eff ),   y = bf16( bf16(x W^T + bias) + bf16( bf16(t / a_div) R_eff ) ).
 * x bf16 [M, K] (row stride ldx), y bf16 [M, N] (row stride ldy), bias bf16 [N] or NULL.
 * LeffT bf16 [r128, K] = L_eff^T with rows >= r zero (r128 = r rounded up to 128); ReffT bf16 [N, r8] =
 * R_eff^T with columns >= r zero (r8 = r rounded up to 8); r = 0: no adapters.
 * Unquantised adapters (prune.py:317-321): L_eff = bf16(lora_left / sqrt(rank)), same for R_eff, a_div = 1.
 * Quantised adapters (prune.py:304-315): L_eff / R_eff = the group fake-quantised adapters, a_div = sqrt(rank). */
int64_t slim_linear24_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t r);
/* Which kernel runs batches of more than 16 rows: -1 the library's rule (the 128 x 256 tile with wave-private weight rows
 * when its grid covers most of the chip, else the 128 x 128 tile), 0 always the 128 x 128 tile, 1 always the 128 x 256 tile.  Results do not depend on the
 * choice (both accumulate the reduction in the same order).  A test and tuning aid; no counterpart in the reference. */
void slim_linear24_set_wide(int mode);
int slim_linear24_forward(const void* x_bf16, int64_t ldx, int64_t M, int64_t K, const void* vals, const void* meta,
                          const float* sf, int64_t N, const void* bias_bf16, const void* LeffT, const void* ReffT,
                          int64_t r, float a_div, void* y_bf16, int64_t ldy, void* workspace, int64_t workspace_bytes,
                          void* stream);

/* The same for batches of at most 16 rows with the adapters kept in the 4-bit group-quantised form of
 * slim.lora.quantize_lora (quantization/utils.py:123-128, :188-193: value = bf16(code / 7 * alpha), one bf16 scale per
 * qblock consecutive reduction channels), which is what the reference's QuantizedMatmul recomputes from the
 * dequantised tensors on every call (quantization.py:313-317):
 *   qL_codes uint8 [r128, K / 2], qL_alpha bf16 [r128, K / qblock]  (L_eff^T, rows >= r zero),
 *   qR_codes uint8 [N, r / 2],    qR_alpha bf16 [N, r / qblock]     (R_eff^T);  even channel in the low nibble;
 *   r % qblock == 0, K % qblock == 0, qblock % 32 == 0.  Larger batches: -3 (use the dense-operand entry point). */
int slim_linear24_forward_q4(const void* x_bf16, int64_t ldx, int64_t M, int64_t K, const void* vals, const void* meta,
                             const float* sf, int64_t N, const void* bias_bf16, const void* qL_codes, const void* qL_alpha,
                             const void* qR_codes, const void* qR_alpha, int64_t r, int qblock, float a_div, void* y_bf16,
                             int64_t ldy, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLIM_HIP_H */
