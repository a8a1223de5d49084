// Argument block of the NT bf16 MFMA GEMM (gemm.hip), shared with the low-rank fit, which issues strided
// batches of it.
#pragma once
#include "common.hpp"

#define GEMM_MAX_PTRS 4

struct GemmArgs {
    const bf16_t* A = nullptr;
    int64_t lda = 0;
    const bf16_t* B = nullptr;
    int64_t ldb = 0;
    int M = 0, N = 0, Kd = 0;
    float alpha = 1.f, beta = 0.f;  // C32 = alpha * acc + beta * C32 (beta = 0: C32 is not read)
    float* C32 = nullptr;
    int64_t ldc32 = 0;
    bf16_t* C16 = nullptr;
    int64_t ldc16 = 0;
    bf16_t* Ct16 = nullptr;
    int64_t ldct = 0;
    // optional fused epilogue of the adapter subtraction (slim/lora.py:199-201):
    // Wn = mask ? 0 : bf16(float(W) - float(bf16(acc)))
    const bf16_t* W = nullptr;
    const uint8_t* mask = nullptr;
    bf16_t* Wn = nullptr;
    int64_t ldw = 0;
    // split-K: gridDim.y slices of the reduction; slice z writes its fp32 partial tile to slab[z][M][N]
    // (no other output), summed afterwards in slice order by splitk_reduce_kernel
    int splits = 0, ksteps_per_split = 0;
    float* slab = nullptr;
    // symmetric product (A == B, M == N, split-K only): workgroups cover the tiles on and above the diagonal, and the
    // reduction reads the mirrored element for the tiles below it (X^T X computed once per pair of tiles; the mirrored
    // value is what the lower tile would have produced, since every product of the sum commutes)
    // (without split-K: set Ct16 = C16 and ldct = ldc16; the upper tiles then write themselves and their mirror image)
    int sym = 0;
    int sym_block = 0;  // set by the launcher: tile order of a symmetric product, see gemm_tile_origin
    int mirror32 = 0;   // set by the launcher: a symmetric product with an fp32 output -- the workgroups of the tiles above the
                        // diagonal also write their tile's transpose into C32 (which must be symmetric on entry when beta != 0)
    // strided batch (gridDim.z problems of identical shape); element strides between consecutive problems
    int batch = 1;
    int64_t sA = 0, sB = 0, sC32 = 0, sC16 = 0, sCt = 0, sSlab = 0;
    // batch of problems whose operands are separate allocations (nptr = batch, at most GEMM_MAX_PTRS): problem z reads
    // Ap<z> / Bp<z> with row stride ldp<z> (both operands) and reduction length Kdp<z>; M, N and the outputs (strided as
    // above) are common.  A == Ap0 etc. for the host-side checks.
    // (separate scalar members, not arrays: an array in the argument block that a kernel indexes by blockIdx moves the whole
    // block into scratch memory)
    int nptr = 0;
    const bf16_t *Ap0 = nullptr, *Ap1 = nullptr, *Ap2 = nullptr, *Ap3 = nullptr;
    const bf16_t *Bp0 = nullptr, *Bp1 = nullptr, *Bp2 = nullptr, *Bp3 = nullptr;
    int64_t ldp0 = 0, ldp1 = 0, ldp2 = 0, ldp3 = 0;
    int Kdp0 = 0, Kdp1 = 0, Kdp2 = 0, Kdp3 = 0;
    void set_problem(int z, const bf16_t* a, const bf16_t* b, int64_t ld, int kd) {     // host side
        switch (z) {
            case 0: Ap0 = a, Bp0 = b, ldp0 = ld, Kdp0 = kd; break;
            case 1: Ap1 = a, Bp1 = b, ldp1 = ld, Kdp1 = kd; break;
            case 2: Ap2 = a, Bp2 = b, ldp2 = ld, Kdp2 = kd; break;
            default: Ap3 = a, Bp3 = b, ldp3 = ld, Kdp3 = kd; break;
        }
    }
    // optional fused epilogue of a Newton-Schulz step: with acc = G G, C16 = bf16(c * (a I + b c^2 G + q c^4 acc)),
    // c = *poly_scal (1 when NULL); needs M == N, no other output form
    const float* poly_G = nullptr;
    const float* poly_scal = nullptr;
    float poly_a = 0.f, poly_b = 0.f, poly_q = 0.f;
    int poly_prescaled = 0;   // the operands already carry the scaling c^2 (acc is used as it is): C16 = bf16(c (a I + q acc))
    // split-K only: the bf16 output is C - c16_diag I (the shifted Gram matrix of a Newton-Schulz step); C32 is unshifted
    float c16_diag = 0.f;
    int64_t poly_sG = 0, poly_sScal = 0;      // batch strides of poly_G / poly_scal
    // optional addend (no split-K, alpha == 1, beta == 0): every output form receives acc + (*add_coef) * add16, add16 a
    // bf16 [M, N] matrix (row stride ldadd), add_coef a float in device memory (a recurrence whose constant an earlier
    // kernel computed); batch strides sAdd / sAddCoef
    const bf16_t* add16 = nullptr;
    int64_t ldadd = 0, sAdd = 0;
    const float* add_coef = nullptr;
    int64_t sAddCoef = 0;
    // tuning aid (slim_gemm_set_trace): when set, workgroup w < trace_cap stores 100 MHz timestamps at
    // trace[8 w ..]: kernel entry, first tile staged, K-loop done, epilogue done
    int64_t* trace = nullptr;
    int trace_cap = 0;
};

// Launch (and, for splits > 1, reduce).  Split-K callers fill splits / ksteps_per_split through
// gemm_plan_splits.
int slim_gemm_launch(const GemmArgs& g, void* stream);
inline void gemm_plan_splits(GemmArgs& g, int splits) {
    const int nk = (g.Kd + 63) / 64;
    if (splits > nk) splits = nk;
    if (splits < 1) splits = 1;
    g.ksteps_per_split = (nk + splits - 1) / splits;
    g.splits = (nk + g.ksteps_per_split - 1) / g.ksteps_per_split;
}
