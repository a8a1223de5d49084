// bf16 MFMA GEMM, "NT" form:  C[M,N] = alpha * A[M,Kd] * Bt[N,Kd]^T, fp32 accumulation.
// Both operands are row-major with the reduction dimension contiguous, which is the layout
// every product of the low-rank fit is arranged to have (the tall matrices are kept both
// as X and X^T; this kernel can emit C, bf16(C) and bf16(C)^T in one pass for that purpose).
//
// Tiling: 128x128 (or, for small problems, 128x64 / 64x128 / 64x64) output tile per 256-thread workgroup (2x2 waves,
// up to 64x64 per wave as 4x4 v_mfma_f32_16x16x32_bf16 fragments), K-step 64, register-staged global->LDS copies with
// the next tile's loads issued before the current tile's MFMAs, two LDS buffers, one barrier
// per K-step.  LDS rows are padded by 8 elements (144-byte stride) so the 16 rows read by a
// ds_read_b128 lane group fall on distinct banks.
#include "common.hpp"
#include "../../include/slim_hip.h"

#include "gemm.hpp"
#include <cstdlib>

#define BK 64
#define LDS_PAD 16          // bf16 of padding per LDS row (SLIM_LDS_PAD at build time for experiments)
#define LDS_STRIDE (BK + LDS_PAD)

// Tile shapes: a 256-thread workgroup is 2 x 2 waves, a wave owns FM x FN fragments of 16 x 16, so the workgroup tile
// is (32 FM) x (32 FN).  The big tile has the best reuse; the small ones exist because the products of the fit are
// small (a few hundred tiles at most) and each workgroup's K-loop is bound by load latency: more, smaller workgroups
// keep more loads in flight per CU and fill the 256 CUs.
// Variants 0-4 stage tiles through registers (one K-step ahead, padded LDS rows); the DMA variants deliver them by LDS-DMA,
// several K-steps ahead (gemm_nt_dma_kernel).
// TILE_288x128 (512 threads, 2 x 4 waves of 144 x 32): for products whose M is the subspace dimension of the fit (576 = 2 x 288,
// 1152 = 4 x 288): no padded rows, and the four (l x Dm) products of a chain are 256 tiles -- one per CU, two waves per SIMD --
// instead of 640 tiles of 128 x 128 on 512 workgroup slots
enum { TILE_128x128, TILE_128x64, TILE_64x128, TILE_64x64, TILE_288x128, TILE_DMA_288x128, TILE_DMA_128x128, TILE_DMA_256x128, TILE_DMA_128x128_S2, TILE_COUNT };
static const int kTileM[TILE_COUNT] = {128, 128, 64, 64, 288, 288, 128, 256, 128};
static const int kTileN[TILE_COUNT] = {128, 64, 128, 64, 128, 128, 128, 128, 128};
static const int kTileBK[TILE_COUNT] = {64, 64, 64, 64, 64, 64, 64, 64, 64};
static const int kTileThreads[TILE_COUNT] = {256, 256, 256, 256, 512, 512, 256, 512, 256};
static const int kTileStages[TILE_COUNT] = {0, 0, 0, 0, 0, 3, 4, 3, 2};       // LDS-DMA stages (0: register staging, two buffers)

// BKT elements of the reduction per tile row: BKT / 8 threads cover a row, NT / (BKT / 8) rows per pass of the NT threads; the
// last pass may cover fewer rows than that (288-row tiles with 512 threads)
template <int ROWS, int BKT, int NT>
struct TileRegs {
    static constexpr int TPR = BKT / 8, RPP = NT / TPR, NP = (ROWS + RPP - 1) / RPP;
    u16x8 r[NP];
};

template <int ROWS, int BKT, int NT>
__device__ __forceinline__ void load_tile(const bf16_t* __restrict__ P, int64_t ld, int rows, int Kd, int row0, int k0,
                                          TileRegs<ROWS, BKT, NT>& tr) {
    constexpr int TPR = BKT / 8, RPP = NT / TPR, NP = (ROWS + RPP - 1) / RPP;
    const int t = threadIdx.x;
    const int kc = (t % TPR) * 8;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int lr = (t / TPR) + RPP * i;
        if (ROWS % RPP != 0 && lr >= ROWS) continue;
        int row = row0 + lr;
        row = row < rows ? row : rows - 1;
        const int k = k0 + kc;
        if (k < Kd) tr.r[i] = *(const u16x8*)(P + (int64_t)row * ld + k);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) tr.r[i][j] = 0;
        }
    }
}

template <int ROWS, int BKT, int NT>
__device__ __forceinline__ void store_tile(bf16_t* lds, const TileRegs<ROWS, BKT, NT>& tr) {
    constexpr int TPR = BKT / 8, RPP = NT / TPR, NP = (ROWS + RPP - 1) / RPP;
    const int t = threadIdx.x;
    const int kc = (t % TPR) * 8;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = (t / TPR) + RPP * i;
        if (ROWS % RPP != 0 && row >= ROWS) continue;
        *(u16x8*)(lds + row * (BKT + LDS_PAD) + kc) = tr.r[i];
    }
}

// Origin of this workgroup's output tile, and the batch offsets applied to the argument block.
template <int BM, int BN>
__device__ __forceinline__ void gemm_tile_origin(GemmArgs& g, const GemmArgs& gin, int& row0, int& col0) {
    // XCD-aware tile order: consecutive hardware workgroup ids round-robin over the 8 XCDs, so give
    // each XCD a contiguous run of tiles (neighbouring tiles share an A or B panel in that XCD's L2).
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = orig % 8;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + orig / 8;
    // consecutive tiles share the panel of the operand with MORE tiles and walk over the few tiles of the other one,
    // so the large operand (the N x K error matrix in the sketch products) is streamed once and the small one is
    // served by L2
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    int tm, tn;
    if (g.sym && BM == 2 * BN) {
        // symmetric product on tiles twice as tall as wide: row tm (256 rows) holds the tiles from column 2 tm on, i.e.
        // every tile with an element on or above the diagonal; the two leftmost of a row straddle the diagonal and compute
        // a few elements below it as well (the same values their mirror images receive: the products of a sum commute)
        tm = 0;
        int rest = wgid;
        while (rest >= tiles_n - 2 * tm) {
            rest -= tiles_n - 2 * tm;
            ++tm;
        }
        tn = 2 * tm + rest;
    } else if (g.sym && g.sym_block > 1) {
        // tiles on and above the diagonal in square blocks of sym_block x sym_block tiles (block rows, then the tiles of a
        // block): the run of consecutive tiles that one XCD works on at a time then shares sym_block row panels and
        // sym_block column panels in its L2, where a row-by-row order shares one row panel and as many column panels as
        // there are tiles
        const int S = g.sym_block, nb = (tiles_n + S - 1) / S;
        int rest = wgid, br = 0, bc = 0;
        bool found = false;
        for (br = 0; br < nb && !found; ++br) {
            const int nr = min(S, tiles_n - br * S);
            for (bc = br; bc < nb; ++bc) {
                const int nc = min(S, tiles_n - bc * S);
                const int cnt = bc == br ? nr * (nr + 1) / 2 : nr * nc;
                if (rest < cnt) {
                    found = true;
                    break;
                }
                rest -= cnt;
            }
            if (found) break;
        }
        const int nc = min(S, tiles_n - bc * S);
        if (bc == br) {
            int r = 0;
            while (rest >= nc - r) {
                rest -= nc - r;
                ++r;
            }
            tm = br * S + r;
            tn = bc * S + r + rest;
        } else {
            tm = br * S + rest / nc;
            tn = bc * S + rest % nc;
        }
    } else if (g.sym) {
        // tiles on and above the diagonal, row by row: row tm holds tiles_n - tm of them
        tm = 0;
        int rest = wgid;
        while (rest >= tiles_n - tm) {
            rest -= tiles_n - tm;
            ++tm;
        }
        tn = tm + rest;
    } else if (tiles_m < tiles_n) {
        tn = wgid / tiles_m;
        tm = wgid % tiles_m;
    } else {
        tm = wgid / tiles_n;
        tn = wgid % tiles_n;
    }
    row0 = tm * BM;
    col0 = tn * BN;

    if (g.batch > 1) {     // strided batch: blockIdx.z selects the problem
        const int64_t bz = blockIdx.z;
        if (g.nptr) {
            // operands in separate allocations.  They are read from the kernel's argument block itself (gin), not from the
            // private copy the kernel modifies: the compiler turns the selection into an index, and an indexed private copy
            // would live in scratch memory
            const int z = (int)bz;
            g.A = z == 0 ? gin.Ap0 : z == 1 ? gin.Ap1 : z == 2 ? gin.Ap2 : gin.Ap3;
            g.B = z == 0 ? gin.Bp0 : z == 1 ? gin.Bp1 : z == 2 ? gin.Bp2 : gin.Bp3;
            g.lda = g.ldb = z == 0 ? gin.ldp0 : z == 1 ? gin.ldp1 : z == 2 ? gin.ldp2 : gin.ldp3;
            g.Kd = z == 0 ? gin.Kdp0 : z == 1 ? gin.Kdp1 : z == 2 ? gin.Kdp2 : gin.Kdp3;
        } else {
            g.A += bz * g.sA;
            g.B += bz * g.sB;
        }
        if (g.C32) g.C32 += bz * g.sC32;
        if (g.C16) g.C16 += bz * g.sC16;
        if (g.Ct16) g.Ct16 += bz * g.sCt;
        if (g.slab) g.slab += bz * g.sSlab;
        if (g.poly_G) g.poly_G += bz * g.poly_sG;
        if (g.poly_scal) g.poly_scal += bz * g.poly_sScal;
        if (g.add16) g.add16 += bz * g.sAdd;
        if (g.add_coef) g.add_coef += bz * g.sAddCoef;
    }

}

__device__ __forceinline__ void gemm_stamp(const GemmArgs& g, int slot) {
    if (g.trace && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (int)blockIdx.x < g.trace_cap)
        g.trace[8 * blockIdx.x + slot] = wall_clock64();
}

// Epilogue shared by the GEMM kernels.  Fragment (i, j) of a wave holds rows wr*16*FM + i*16 + fq*4 + reg, column
// wc*16*FN + j*16 + fr of the tile: written straight from the fragments every store would be a 2- or 4-byte piece of
// sixteen different cache lines (and the transposed output a scatter), which costs more than the whole K-loop of the
// fit's short products.  So the tile goes through LDS (free once the K-loop is done) as fp32 [BM][BN + 1] and leaves
// in whole rows: 16-byte stores along N for C / the split-K slab / the residual, 16-byte stores along M for C^T.
// The odd row stride keeps both the row-wise and the column-wise LDS reads conflict-free.
template <int FM, int FN, int WM, int WN, bool ALL_FORMS = WM * WN == 4>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[FM][FN], float* T, int row0, int col0, int wr,
                                              int wc, int fr, int fq) {
    constexpr int BM = 16 * FM * WM, BN = 16 * FN * WN, TS = BN + 1, NT = 64 * WM * WN;
    // (ALL_FORMS false: the eight-wave tall tiles are launched for plain bf16 outputs only, the other output forms are
    // compiled out of them)
    __syncthreads();                 // every wave is done with the operand tiles
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                T[(wr * 16 * FM + i * 16 + fq * 4 + r) * TS + wc * 16 * FN + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    gemm_stamp(g, 4);
    const int t = threadIdx.x;
    if (g.add16) {
        // the addend joins the tile in LDS, so that every output form (and the transposed copy) carries it
        const float cf = g.add_coef ? *g.add_coef : 1.f;
        constexpr int TPRA = BN / 8;
        const int c = (t % TPRA) * 8, col = col0 + c;
        if (col < g.N) {
            const bool v8 = g.N - col >= 8 && (g.ldadd & 7) == 0 && ((uintptr_t)g.add16 & 15) == 0;
            for (int r = t / TPRA; r < BM && row0 + r < g.M; r += NT / TPRA) {
                const bf16_t* p = g.add16 + (int64_t)(row0 + r) * g.ldadd + col;
                if (v8) {
                    const u16x8 d = *(const u16x8*)p;
#pragma unroll
                    for (int u = 0; u < 8; ++u) T[r * TS + c + u] += cf * bf16_to_f32(d[u]);
                } else {
                    for (int u = 0; u < 8 && col + u < g.N; ++u) T[r * TS + c + u] += cf * bf16_to_f32(p[u]);
                }
            }
        }
        __syncthreads();
    }
    const bool split = ALL_FORMS && g.splits > 1;
    // Row-wise outputs.  On this ISA loads and stores retire through one in-order counter, so a loop that waits for a
    // load also waits for every earlier store's acknowledgement (a round trip to L2 per iteration): the common case
    // (no beta, no residual) therefore runs in a loop of its own that contains no global load at all, and the residual
    // form fetches all its W / mask pieces before its first store.
    const bool wide16 = !split && g.C16 && !g.C32 && !g.Wn && !g.poly_G && g.beta == 0.f && (g.ldc16 & 7) == 0 &&
                        ((uintptr_t)g.C16 & 15) == 0;
    if (wide16) {
        // bf16 C alone (the tall products of the fit): 16-byte stores, eight columns per lane -- the store tail of a tile is
        // bound by the number of store instructions, not by bytes
        constexpr int TPR8 = BN / 8, RSTEP8 = NT / TPR8, ITERS8 = BM / RSTEP8;
        const int c = (t % TPR8) * 8, col = col0 + c, rt = t / TPR8;
        const int nc = g.N - col;
        if (nc > 0) {
            float v[ITERS8][8];
#pragma unroll
            for (int it = 0; it < ITERS8; ++it)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[it][u] = T[(rt + it * RSTEP8) * TS + c + u] * g.alpha;
#pragma unroll
            for (int it = 0; it < ITERS8; ++it) {
                const int row = row0 + rt + it * RSTEP8;
                if (row >= g.M) continue;
                bf16_t* p = g.C16 + (int64_t)row * g.ldc16 + col;
                if (nc >= 8)
                    *(u16x8*)p = u16x8{f32_to_bf16(v[it][0]), f32_to_bf16(v[it][1]), f32_to_bf16(v[it][2]), f32_to_bf16(v[it][3]),
                                       f32_to_bf16(v[it][4]), f32_to_bf16(v[it][5]), f32_to_bf16(v[it][6]), f32_to_bf16(v[it][7])};
                else
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < nc) p[u] = f32_to_bf16(v[it][u]);
            }
        }
    } else if (split || g.C32 || g.C16 || g.Wn) {
        constexpr int TPR = BN / 4;                    // threads per tile row, four columns each
        constexpr int RSTEP = NT / TPR, ITERS = BM / RSTEP;
        const int c = (t % TPR) * 4, col = col0 + c;
        const int nc = g.N - col;                      // valid columns from `col` on
        const int rt = t / TPR;
        if (nc > 0) {
            if (split) {
                float* slab = g.slab + (int64_t)blockIdx.y * g.M * g.N;
                const bool v32 = (g.N & 3) == 0 && ((uintptr_t)slab & 15) == 0 && nc >= 4;
                float v[ITERS][4];
#pragma unroll
                for (int it = 0; it < ITERS; ++it)
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[it][u] = T[(rt + it * RSTEP) * TS + c + u];
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int row = row0 + rt + it * RSTEP;
                    if (row >= g.M) continue;
                    float* p = slab + (int64_t)row * g.N + col;
                    if (v32) *(f32x4*)p = f32x4{v[it][0], v[it][1], v[it][2], v[it][3]};
                    else
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nc) p[u] = v[it][u];
                }
            } else if (ALL_FORMS && g.Wn) {
                u16x4 w[ITERS];
                uint32_t mk[ITERS];
                const bool vw = nc >= 4 && (g.ldw & 3) == 0 && ((uintptr_t)g.W & 7) == 0 && ((uintptr_t)g.Wn & 7) == 0 &&
                                (!g.mask || ((uintptr_t)g.mask & 3) == 0);
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int row = row0 + rt + it * RSTEP;
                    w[it] = u16x4{0, 0, 0, 0};
                    mk[it] = 0;
                    if (row >= g.M) continue;
                    const int64_t o = (int64_t)row * g.ldw + col;
                    if (vw) {
                        w[it] = *(const u16x4*)(g.W + o);
                        if (g.mask) mk[it] = *(const uint32_t*)(g.mask + o);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nc) {
                                w[it][u] = g.W[o + u];
                                if (g.mask) mk[it] |= (uint32_t)(g.mask[o + u] != 0) << (8 * u);
                            }
                    }
                }
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int r = rt + it * RSTEP, row = row0 + r;
                    if (row >= g.M) continue;
                    const int64_t o = (int64_t)row * g.ldw + col;
                    bf16_t lr16[4], wn[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float v = T[r * TS + c + u] * g.alpha;
                        lr16[u] = f32_to_bf16(v);
                        wn[u] = ((mk[it] >> (8 * u)) & 0xffu) ? (bf16_t)0 : f32_to_bf16(bf16_to_f32(w[it][u]) - bf16_to_f32(lr16[u]));
                    }
                    if (vw) *(u16x4*)(g.Wn + o) = u16x4{wn[0], wn[1], wn[2], wn[3]};
                    else
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nc) g.Wn[o + u] = wn[u];
                    if (g.C16) {
                        bf16_t* p = g.C16 + (int64_t)row * g.ldc16 + col;
                        if (nc >= 4 && (g.ldc16 & 3) == 0 && ((uintptr_t)g.C16 & 7) == 0) *(u16x4*)p = u16x4{lr16[0], lr16[1], lr16[2], lr16[3]};
                        else
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (u < nc) p[u] = lr16[u];
                    }
                }
            } else if (ALL_FORMS && g.poly_G) {
                // the polynomial of a Newton-Schulz step, evaluated where the tile of G G is still on chip; every value of
                // G is fetched before the first store (see above), the arithmetic is the stand-alone polynomial kernel's
                const float sc = g.poly_scal ? *g.poly_scal : 1.f;
                const float c2 = sc * sc;
                const float c4 = g.poly_prescaled ? 1.f : c2 * c2;
                const bool vg = nc >= 4 && (g.N & 3) == 0 && ((uintptr_t)g.poly_G & 15) == 0;
                const bool v16 = nc >= 4 && (g.ldc16 & 3) == 0 && ((uintptr_t)g.C16 & 7) == 0;
                float gv[ITERS][4];
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int row = row0 + rt + it * RSTEP;
#pragma unroll
                    for (int u = 0; u < 4; ++u) gv[it][u] = 0.f;
                    if (row >= g.M || g.poly_b == 0.f) continue;
                    const float* p = g.poly_G + (int64_t)row * g.N + col;
                    if (vg) {
                        const f32x4 t4 = *(const f32x4*)p;
#pragma unroll
                        for (int u = 0; u < 4; ++u) gv[it][u] = t4[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nc) gv[it][u] = p[u];
                    }
                }
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int r = rt + it * RSTEP, row = row0 + r;
                    if (row >= g.M) continue;
                    bf16_t o[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float v = g.poly_b * c2 * gv[it][u] + g.poly_q * c4 * T[r * TS + c + u];
                        if (row == col + u) v += g.poly_a;
                        o[u] = f32_to_bf16(sc * v);
                    }
                    bf16_t* p = g.C16 + (int64_t)row * g.ldc16 + col;
                    if (v16) *(u16x4*)p = u16x4{o[0], o[1], o[2], o[3]};
                    else
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u < nc) p[u] = o[u];
                }
            } else {
                const bool v32 = nc >= 4 && (g.ldc32 & 3) == 0 && ((uintptr_t)g.C32 & 15) == 0;
                const bool v16 = nc >= 4 && (g.ldc16 & 3) == 0 && ((uintptr_t)g.C16 & 7) == 0;
                auto store_row = [&](int row, const float (&v)[4]) {
                    if (g.C32) {
                        float* p = g.C32 + (int64_t)row * g.ldc32 + col;
                        if (v32) *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
                        else
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (u < nc) p[u] = v[u];
                    }
                    if (g.C16) {
                        bf16_t* p = g.C16 + (int64_t)row * g.ldc16 + col;
                        if (v16) *(u16x4*)p = u16x4{f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
                        else
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (u < nc) p[u] = f32_to_bf16(v[u]);
                    }
                };
                if (ALL_FORMS && g.beta != 0.f) {
                    for (int r = rt; r < BM && row0 + r < g.M; r += RSTEP) {
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            v[u] = T[r * TS + c + u] * g.alpha;
                            if (u < nc) v[u] += g.beta * g.C32[(int64_t)(row0 + r) * g.ldc32 + col + u];
                        }
                        store_row(row0 + r, v);
                        if (g.mirror32) {          // the finished values, for the transposed copy below
#pragma unroll
                            for (int u = 0; u < 4; ++u) T[r * TS + c + u] = v[u];
                        }
                    }
                } else if (ALL_FORMS && g.mirror32) {
                    for (int r = rt; r < BM && row0 + r < g.M; r += RSTEP) {
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = T[r * TS + c + u] * g.alpha;
                        store_row(row0 + r, v);
#pragma unroll
                        for (int u = 0; u < 4; ++u) T[r * TS + c + u] = v[u];
                    }
                } else {
                    // fully unrolled: all LDS reads are in flight before the first store is built
                    float v[ITERS][4];
#pragma unroll
                    for (int it = 0; it < ITERS; ++it)
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[it][u] = T[(rt + it * RSTEP) * TS + c + u] * g.alpha;
#pragma unroll
                    for (int it = 0; it < ITERS; ++it)
                        if (row0 + rt + it * RSTEP < g.M) store_row(row0 + rt + it * RSTEP, v[it]);
                }
            }
        }
    }
    gemm_stamp(g, 5);
    // (tiles twice as tall as wide: only the tiles right of the diagonal's 2 x 2 group are mirrored -- the positions the two
    // straddling tiles of a row would mirror into are computed directly, by themselves or by their neighbour)
    const bool mirrors = BM == 2 * BN ? col0 >= row0 + BM : row0 != col0;
    if (ALL_FORMS && g.mirror32 && !split && mirrors) {
        // symmetric fp32 product: this tile's transpose goes to the tile below the diagonal, which no workgroup computes.
        // Lanes walk along the tile's rows (the output's columns): 256 contiguous bytes per wave and output row, and the odd
        // LDS row stride keeps the column-wise reads conflict-free.
        __syncthreads();
        for (int idx = t; idx < BM * BN; idx += NT) {
            const int c = idx / BM, r = idx % BM;
            if (row0 + r < g.M && col0 + c < g.N) g.C32[(int64_t)(col0 + c) * g.ldc32 + row0 + r] = T[r * TS + c];
        }
    }
    if (g.Ct16 && !split && !(g.sym && BM == 2 * BN && !mirrors)) {
        // a wave writes eight rows of C^T (tile columns) at a time, 8 lanes x 16 bytes = 128 contiguous bytes each
        const int lane = t & 63, wave = t >> 6;
        const bool vt = (g.ldct & 7) == 0 && ((uintptr_t)g.Ct16 & 15) == 0;
        constexpr int CW = 8 * (NT / 64);                    // tile columns the waves cover per round
        constexpr int HALVES = (BM + 63) / 64;               // 64-row groups; the last one may be partial (BM a multiple of 8)
#pragma unroll
        for (int cbi = 0; cbi < BN / CW; ++cbi) {
            const int c = cbi * CW + wave * 8 + (lane & 7), col = col0 + c;
            bf16_t o[HALVES][8];
#pragma unroll
            for (int half = 0; half < HALVES; ++half)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = half * 64 + (lane >> 3) * 8 + u;
                    o[half][u] = f32_to_bf16(T[(r < BM ? r : BM - 1) * TS + c] * g.alpha);
                }
            if (col >= g.N) continue;
#pragma unroll
            for (int half = 0; half < HALVES; ++half) {
                const int row = row0 + half * 64 + (lane >> 3) * 8;
                if (row >= g.M || half * 64 + (lane >> 3) * 8 >= BM) continue;
                bf16_t* p = g.Ct16 + (int64_t)col * g.ldct + row;
                if (row + 7 < g.M && vt)
                    *(u16x8*)p = u16x8{o[half][0], o[half][1], o[half][2], o[half][3], o[half][4], o[half][5], o[half][6], o[half][7]};
                else
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (row + u < g.M) p[u] = o[half][u];
            }
        }
    }
}

// FAMILY only names the instantiation (one symbol per shape class of the fit), so that a profiler's per-kernel
// summary separates them the way the in-library timer does; the code is identical.
enum { FAM_GRAM, FAM_SMALL, FAM_APPLY, FAM_SKETCH, FAM_BIG, FAM_RESIDUAL, FAM_COUNT };
template <int FAMILY, int FM, int FN, int BKT = 64, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(const GemmArgs gin) {
    constexpr int BM = 16 * FM * WM, BN = 16 * FN * WN, STRIDE = BKT + LDS_PAD, NT = 64 * WM * WN;
    GemmArgs g = gin;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    // buffer b: A tile at smem + b * (BM + BN) * STRIDE, B tile right after it
    auto As = [&](int b) { return smem + b * (BM + BN) * STRIDE; };
    auto Bs = [&](int b) { return smem + b * (BM + BN) * STRIDE + BM * STRIDE; };

    int row0, col0;
    gemm_tile_origin<BM, BN>(g, gin, row0, col0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gemm_stamp(g, 0);
    // (A register pipeline two K-steps deep -- a second set of staging registers -- was measured slower on every family, SYRK
    // 592 -> 838 us: at 198 VGPRs only one workgroup fits a CU, and two resident workgroups hide more latency than the deeper
    // prefetch of one.)
    TileRegs<BM, BKT, NT> ra;
    TileRegs<BN, BKT, NT> rb;
    const int nk_total = (g.Kd + BKT - 1) / BKT;
    const int per_split = g.ksteps_per_split * (BK / BKT);               // the plan counts 64-deep steps
    const int kt0 = g.splits > 1 ? (int)blockIdx.y * per_split : 0;
    const int nk = g.splits > 1 ? min(per_split, nk_total - kt0) : nk_total;
    load_tile<BM, BKT, NT>(g.A, g.lda, g.M, g.Kd, row0, kt0 * BKT, ra);
    load_tile<BN, BKT, NT>(g.B, g.ldb, g.N, g.Kd, col0, kt0 * BKT, rb);
    store_tile<BM, BKT, NT>(As(0), ra);
    store_tile<BN, BKT, NT>(Bs(0), rb);
    __syncthreads();
    gemm_stamp(g, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<BM, BKT, NT>(g.A, g.lda, g.M, g.Kd, row0, (kt0 + kt + 1) * BKT, ra);
            load_tile<BN, BKT, NT>(g.B, g.ldb, g.N, g.Kd, col0, (kt0 + kt + 1) * BKT, rb);
        }
        const bf16_t* a_base = As(cur) + (wr * 16 * FM + fr) * STRIDE + fq * 8;
        const bf16_t* b_base = Bs(cur) + (wc * 16 * FN + fr) * STRIDE + fq * 8;
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(a_base + i * 16 * STRIDE + ks * 32);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = *(const bf16x8*)(b_base + j * 16 * STRIDE + ks * 32);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile<BM, BKT, NT>(As(cur ^ 1), ra);
            store_tile<BN, BKT, NT>(Bs(cur ^ 1), rb);
        }
        __syncthreads();
    }
    gemm_stamp(g, 2);
    gemm_epilogue<FM, FN, WM, WN>(g, acc, (float*)smem, row0, col0, wr, wc, fr, fq);
    gemm_stamp(g, 3);
}

// The same product with the operand tiles delivered by LDS-DMA (global_load_lds, 16 bytes per lane), STAGES - 1 K-steps
// ahead of the matrix instructions.  The products of the fit are a few hundred tiles at most -- one or two workgroups
// per CU, nothing else resident to hide a K-step's trip to memory -- and with register staging (one step ahead) a K-step
// costs that trip, 2.5-3.5 times what its matrix instructions take.  The DMA needs no staging registers, so the depth
// is bounded by LDS alone.
// LDS image of a stage: the BM rows of A then the BN rows of B, 128 bytes (64 reduction elements) per row, no padding (a
// wave-instruction of the DMA writes 64 x 16 contiguous bytes: 8 rows); the 16-byte piece c of row r is stored at
// position c ^ (r & 7), applied on the SOURCE side (lane (r, p) of a piece fetches column p ^ r), which makes the
// fragment reads (16 rows x one 16-byte column per quarter wave) conflict-free.
// Ordering: a wave waits for its own DMAs of a stage by count (vmcnt), then the workgroup barrier makes every wave's
// visible; the stage freed by the previous K-step is refilled after that same barrier.  Needs Kd % 64 == 0.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int FAMILY, int FM, int FN, int WM, int WN, int STAGES, bool ALL_FORMS>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_dma_kernel(const GemmArgs gin) {
    constexpr int BM = 16 * FM * WM, BN = 16 * FN * WN, NW = WM * WN;
    constexpr int PIECES = (BM + BN) / 8;               // 1 KiB pieces (8 rows) of a stage
    constexpr int PPW = (PIECES + NW - 1) / NW;         // per wave; the last wave(s) repeat the final piece (same bytes, same place)
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(STAGES >= 2 && (STAGES - 1) * PPW < 64, "the load counter has six bits");
    GemmArgs g = gin;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    char* const lds = (char*)smem;

    int row0, col0;
    gemm_tile_origin<BM, BN>(g, gin, row0, col0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gemm_stamp(g, 0);
    const int nk_total = g.Kd / BK;
    const int kt0 = g.splits > 1 ? (int)blockIdx.y * g.ksteps_per_split : 0;
    const int nk = g.splits > 1 ? min(g.ksteps_per_split, nk_total - kt0) : nk_total;

    // this lane's sources: row (lane / 8) of each of the wave's pieces, 16-byte column (lane % 8) ^ (lane / 8)
    const bf16_t* src[PPW];
    int dst[PPW];
    {
        const int pr = lane >> 3, pc = (lane & 7) ^ pr;
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const int q = min(s * NW + wave, PIECES - 1);
            const int R = q * 8 + pr;
            if (R < BM) src[s] = g.A + (int64_t)min(row0 + R, g.M - 1) * g.lda + (int64_t)kt0 * BK + pc * 8;
            else src[s] = g.B + (int64_t)min(col0 + R - BM, g.N - 1) * g.ldb + (int64_t)kt0 * BK + pc * 8;
            dst[s] = q * 1024;
        }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int s = 0; s < PPW; ++s)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[s] + (int64_t)kt * BK), (lds_ptr_t)(lds + buf * STAGE_BYTES + dst[s]), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nk) issue(st, st);
    gemm_stamp(g, 1);

    // fragment addresses: row (.. + fr), 16-byte column (ks * 4 + fq) ^ (fr & 7); the second 32-deep half flips bit 2
    const int x0 = (fq ^ (fr & 7)) * 16;
    const int a_off = (wr * 16 * FM + fr) * 128, b_off = (BM + wc * 16 * FN + fr) * 128;
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most the STAGES - 2 younger ones are outstanding
        if (kt + STAGES - 2 < nk) wait_vmcnt<(STAGES - 2) * PPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* a_base = lds + buf * STAGE_BYTES + a_off;
        const char* b_base = lds + buf * STAGE_BYTES + b_off;
        // the first 32-deep half's fragments are requested together, the second half's while the first half's products run
        // (left alone the compiler sinks every read to its first use: an LDS round trip per pair of products)
        bf16x8 af[2][FM], bfr[2][FN];
        auto fetch = [&](int ks) {
            const int x = ks == 0 ? x0 : (x0 ^ 64);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[ks][j] = *(const bf16x8*)(b_base + j * 16 * 128 + x);
#pragma unroll
            for (int i = 0; i < FM; ++i) af[ks][i] = *(const bf16x8*)(a_base + i * 16 * 128 + x);
        };
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        // every wave is past the previous K-step's fragment reads: its buffer takes stage kt + STAGES - 1.  Requested behind
        // this step's first fragment reads, not ahead of them: the reads are what the matrix instructions wait for (5 % of the
        // loop; a raised wave priority around the matrix instructions measured nothing)
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, buf == 0 ? STAGES - 1 : buf - 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
        // the second half's reads go between the first half's products, one per row of fragments
        __builtin_amdgcn_sched_group_barrier(0x100, FN, 0);          // (B's fragments first)
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bfr[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
    gemm_stamp(g, 2);
    gemm_epilogue<FM, FN, WM, WN, ALL_FORMS>(g, acc, (float*)smem, row0, col0, wr, wc, fr, fq);
    gemm_stamp(g, 3);
}

// C32 / C16 <- alpha * sum over slices of slab[z] (+ beta * C32), slices added in index order.  One 64 x 64 tile per
// workgroup, read as 16-byte pieces of every slice.  Symmetric products fill only the 128 x 128 tiles on and above the
// diagonal: the workgroups of the tiles below it leave at once, those above it write their tile and, through LDS, its
// transpose into the mirrored position (bit-identical to summing the mirrored slab entries: the same numbers in the
// same order).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N, float alpha,
                                                            float beta, float* __restrict__ C32, int64_t ldc32,
                                                            bf16_t* __restrict__ C16, int64_t ldc16, int64_t sSlab, int64_t sC32,
                                                            int64_t sC16, int sym_tile, float c16_diag) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const int64_t bz = blockIdx.z;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const bool mirror = sym_tile && row0 / sym_tile < col0 / sym_tile;
    if (sym_tile && row0 / sym_tile > col0 / sym_tile) return;
    slab += bz * sSlab;
    if (C32) C32 += bz * sC32;
    if (C16) C16 += bz * sC16;
    const int64_t n = (int64_t)M * N;
    const int tx = (t & 15) * 4, ty = t >> 4;
    const bool vec = (N & 3) == 0 && ((uintptr_t)slab & 15) == 0;
    // (the four rows of a thread are fetched slice by slice together -- four independent 16-byte loads in flight per slice --
    // and leave as 16- and 8-byte stores)
    const bool vec_out = (!C32 || ((ldc32 & 3) == 0 && ((uintptr_t)C32 & 15) == 0)) && (!C16 || ((ldc16 & 3) == 0 && ((uintptr_t)C16 & 7) == 0));
    const int col = col0 + tx;
    if (vec && vec_out && col + 3 < N && row0 + 63 < M) {
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[i][u] = 0.f;
        const float* p = slab + (int64_t)(row0 + ty) * N + col;
        for (int z = 0; z < splits; ++z) {
            f32x4 q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = *(const f32x4*)(p + (int64_t)z * n + (int64_t)16 * i * N);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u) v[i][u] += q[i][u];
        }
        f32x4 old[4];
        if (C32 && beta != 0.f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) old[i] = *(const f32x4*)(C32 + (int64_t)(row0 + ty + 16 * i) * ldc32 + col);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ty + 16 * i, row = row0 + r;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[i][u] *= alpha;
                if (C32 && beta != 0.f) v[i][u] += beta * old[i][u];
                tile[r][tx + u] = v[i][u];
            }
            if (C32) *(f32x4*)(C32 + (int64_t)row * ldc32 + col) = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
            if (C16) {
                bf16_t o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = f32_to_bf16(row == col + u ? v[i][u] - c16_diag : v[i][u]);
                *(u16x4*)(C16 + (int64_t)row * ldc16 + col) = u16x4{o[0], o[1], o[2], o[3]};
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 16 * i, row = row0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < M && col < N) {
            const float* p = slab + (int64_t)row * N + col;
            if (vec && col + 3 < N) {
                for (int z = 0; z < splits; ++z) {
                    const f32x4 q = *(const f32x4*)(p + (int64_t)z * n);
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] += q[u];
                }
            } else {
                for (int z = 0; z < splits; ++z)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (col + u < N) v[u] += p[(int64_t)z * n + u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] *= alpha;
            if (row < M && col + u < N) {
                if (C32) {
                    if (beta != 0.f) v[u] += beta * C32[(int64_t)row * ldc32 + col + u];
                    C32[(int64_t)row * ldc32 + col + u] = v[u];
                }
                if (C16) C16[(int64_t)row * ldc16 + col + u] = f32_to_bf16(row == col + u ? v[u] - c16_diag : v[u]);
            }
            tile[r][tx + u] = v[u];
        }
    }
    if (!mirror) return;
    __syncthreads();
    // the transposed tile: element (col0 + c, row0 + r) = tile[r][c]; strictly off the diagonal, so no diagonal shift
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = ty + 16 * i, mrow = col0 + c;          // row of the mirrored position
        if (mrow >= N) continue;
        if (vec_out && row0 + tx + 3 < M && beta == 0.f) {
            const float v0 = tile[tx][c], v1 = tile[tx + 1][c], v2 = tile[tx + 2][c], v3 = tile[tx + 3][c];
            if (C32) *(f32x4*)(C32 + (int64_t)mrow * ldc32 + row0 + tx) = f32x4{v0, v1, v2, v3};
            if (C16) *(u16x4*)(C16 + (int64_t)mrow * ldc16 + row0 + tx) = u16x4{f32_to_bf16(v0), f32_to_bf16(v1), f32_to_bf16(v2), f32_to_bf16(v3)};
            continue;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int mcol = row0 + tx + u;
            if (mcol >= M) continue;
            float v = tile[tx + u][c];
            if (C32) {
                if (beta != 0.f) v += beta * C32[(int64_t)mrow * ldc32 + mcol];
                C32[(int64_t)mrow * ldc32 + mcol] = v;
            }
            if (C16) C16[(int64_t)mrow * ldc16 + mcol] = f32_to_bf16(v);
        }
    }
}

// The same reduction for products that also want the transposed bf16 output (the sketch products of the fit): a
// 64 x 64 tile per workgroup, summed slice by slice into LDS, written as rows of C and as rows of C^T.
__global__ __launch_bounds__(256) void splitk_reduce_t_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                                              float alpha, bf16_t* __restrict__ C16, int64_t ldc16,
                                                              bf16_t* __restrict__ Ct16, int64_t ldct, int64_t sSlab,
                                                              int64_t sC16, int64_t sCt) {
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const int64_t bz = blockIdx.z;
    slab += bz * sSlab;
    if (C16) C16 += bz * sC16;
    if (Ct16) Ct16 += bz * sCt;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const int64_t n = (int64_t)M * N;
    const int tx = (t & 15) * 4, ty = t >> 4;
    const bool vec = (N & 3) == 0 && ((uintptr_t)slab & 15) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 16 * i, row = row0 + r, col = col0 + tx;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < M && col < N) {
            const float* p = slab + (int64_t)row * N + col;
            if (vec && col + 3 < N) {
                for (int z = 0; z < splits; ++z) {
                    const f32x4 q = *(const f32x4*)(p + (int64_t)z * n);
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] += q[u];
                }
            } else {
                for (int z = 0; z < splits; ++z)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (col + u < N) v[u] += p[(int64_t)z * n + u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] *= alpha;
            tile[r][tx + u] = v[u];
        }
        if (C16 && row < M && col < N) {
            bf16_t* o = C16 + (int64_t)row * ldc16 + col;
            if (col + 3 < N && (ldc16 & 3) == 0 && ((uintptr_t)C16 & 7) == 0)
                *(u16x4*)o = u16x4{f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
            else
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (col + u < N) o[u] = f32_to_bf16(v[u]);
        }
    }
    __syncthreads();
    if (Ct16) {
        const int c = t >> 2, r0 = (t & 3) * 16, col = col0 + c;     // one C^T row per four lanes, 16 values each
        if (col < N) {
            bf16_t* o = Ct16 + (int64_t)col * ldct + row0 + r0;
            const bool v8 = (ldct & 7) == 0 && ((uintptr_t)Ct16 & 15) == 0 && row0 + r0 + 15 < M;
            bf16_t h[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) h[u] = f32_to_bf16(tile[r0 + u][c]);
            if (v8) {
                *(u16x8*)o = u16x8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
                *(u16x8*)(o + 8) = u16x8{h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]};
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (row0 + r0 + u < M) o[u] = h[u];
            }
        }
    }
}

typedef void (*gemm_kernel_t)(GemmArgs);
template <int FAMILY>
static gemm_kernel_t gemm_kernel_of(int tile) {
    switch (tile) {
        case TILE_128x64: return gemm_nt_kernel<FAMILY, 4, 2>;
        case TILE_64x128: return gemm_nt_kernel<FAMILY, 2, 4>;
        case TILE_64x64: return gemm_nt_kernel<FAMILY, 2, 2>;
        case TILE_288x128: return gemm_nt_kernel<FAMILY, 9, 2, 64, 2, 4>;
        case TILE_DMA_288x128: return gemm_nt_dma_kernel<FAMILY, 9, 2, 2, 4, 3, false>;
        case TILE_DMA_128x128: return gemm_nt_dma_kernel<FAMILY, 4, 4, 2, 2, 4, true>;
        case TILE_DMA_256x128: return gemm_nt_dma_kernel<FAMILY, 4, 4, 4, 2, 3, true>;
        case TILE_DMA_128x128_S2: return gemm_nt_dma_kernel<FAMILY, 4, 4, 2, 2, 2, true>;
        default: return gemm_nt_kernel<FAMILY, 4, 4>;
    }
}

// tile shape forced for every launch (-1: chosen per problem); a tuning aid, see slim_gemm_set_tile
static int g_tile_override = -1;
extern "C" void slim_gemm_set_tile(int tile) { g_tile_override = (tile >= 0 && tile < TILE_COUNT) ? tile : -1; }

static int64_t* g_trace_buf = nullptr;
static int g_trace_cap = 0;
extern "C" void slim_gemm_set_trace(void* buf, int capacity_workgroups) {
    g_trace_buf = (int64_t*)buf;
    g_trace_cap = buf ? capacity_workgroups : 0;
}

// Tile choice.  Alone on the chip a small product finishes sooner with small tiles (a workgroup's K-loop is
// latency-bound, so more workgroups mean more loads in flight), but the fit runs several linears' chains on
// concurrent streams, where what counts is how long a product holds CU slots: there the 128 x 128 register-staged
// tile is the cheapest for every shape class of the fit (measured on the whole compression step), so it is the default
// and the other variants are reached through slim_gemm_set_tile.
static int g_family_tile[FAM_COUNT] = {-1, -1, -1, -1, -1, -1};
extern "C" void slim_gemm_set_family_tile(int family, int tile) {
    if (family >= 0 && family < FAM_COUNT) g_family_tile[family] = (tile >= 0 && tile < TILE_COUNT) ? tile : -1;
}

// SLIM_GEMM_FAMILY_TILES="family:tile,..." (a tuning aid): slim_gemm_set_family_tile from the environment
static const bool g_family_tiles_env = [] {
    const char* v = getenv("SLIM_GEMM_FAMILY_TILES");
    while (v && *v) {
        char* end = nullptr;
        const long fam = strtol(v, &end, 10);
        if (end == v || *end != ':') break;
        v = end + 1;
        const long tile = strtol(v, &end, 10);
        if (end == v) break;
        slim_gemm_set_family_tile((int)fam, (int)tile);
        v = *end == ',' ? end + 1 : end;
    }
    return true;
}();

// SLIM_GEMM_TALL_TILE=0: the 288 x 128 tile is never chosen (A/B aid)
static const bool g_tall_tile = [] {
    const char* v = getenv("SLIM_GEMM_TALL_TILE");
    return !(v && v[0] == '0');
}();

// SLIM_GEMM_SYM_BLOCK=n (A/B aid): block size of the symmetric products' tile order (0 or 1: row by row)
static const int g_sym_block = [] {
    const char* v = getenv("SLIM_GEMM_SYM_BLOCK");
    return v ? atoi(v) : 8;
}();
static const bool g_dma_tiles = [] {
    const char* v = getenv("SLIM_GEMM_DMA");
    return !(v && v[0] == '0');
}();
// SLIM_GEMM_SYM_RECT=0: the big symmetric products stay on the square register-staged tile (A/B aid)
static const bool g_sym_rect = [] {
    const char* v = getenv("SLIM_GEMM_SYM_RECT");
    return !(v && v[0] == '0');
}();
static int gemm_pick_tile(int family) {
    if (g_tile_override >= 0) return g_tile_override;
    return g_family_tile[family] >= 0 ? g_family_tile[family] : TILE_128x128;
}

int slim_gemm_launch(const GemmArgs& g_in, void* stream) {
    GemmArgs g = g_in;
    g.trace = g_trace_buf;
    g.trace_cap = g_trace_cap;
    SLIM_REQUIRE(g.A && g.B && g.M > 0 && g.N > 0 && g.Kd > 0, "slim_gemm_nt_bf16: bad arguments");
    SLIM_REQUIRE(g.Kd % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.lda >= g.Kd && g.ldb >= g.Kd,
                 "slim_gemm_nt_bf16: Kd, lda, ldb must be multiples of 8 and ld >= Kd");
    SLIM_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "slim_gemm_nt_bf16: operands must be 16-byte aligned");
    SLIM_REQUIRE(g.C32 || g.C16 || g.Ct16 || g.Wn, "slim_gemm_nt_bf16: no output requested");
    SLIM_REQUIRE(!g.Ct16 || (((uintptr_t)g.Ct16 % 8) == 0 && g.beta == 0.f),
                 "slim_gemm_nt_bf16: Ct must be 8-byte aligned and cannot be combined with beta");
    SLIM_REQUIRE(!g.add16 || (g.splits <= 1 && g.alpha == 1.f && g.beta == 0.f && !g.poly_G && !g.Wn && !g.sym),
                 "slim_gemm_nt_bf16: the addend needs an unsplit plain product");
    SLIM_REQUIRE(!g.poly_G || (g.C16 && !g.C32 && !g.Ct16 && !g.Wn && g.splits <= 1 && g.beta == 0.f && g.M == g.N),
                 "slim_gemm_nt_bf16: the polynomial epilogue writes C_bf16 only, of a square product");
    SLIM_REQUIRE(g.nptr == 0 || (g.nptr == g.batch && g.nptr <= GEMM_MAX_PTRS && g.splits <= 1), "slim_gemm_nt_bf16: bad pointer batch");
    {
        const bf16_t* ap[4] = {g.Ap0, g.Ap1, g.Ap2, g.Ap3};
        const bf16_t* bp[4] = {g.Bp0, g.Bp1, g.Bp2, g.Bp3};
        const int64_t ld[4] = {g.ldp0, g.ldp1, g.ldp2, g.ldp3};
        const int kd[4] = {g.Kdp0, g.Kdp1, g.Kdp2, g.Kdp3};
        for (int z = 0; z < g.nptr; ++z)
            SLIM_REQUIRE(ap[z] && bp[z] && kd[z] > 0 && kd[z] % 8 == 0 && ld[z] % 8 == 0 && ld[z] >= kd[z] &&
                             ((uintptr_t)ap[z] % 16) == 0 && ((uintptr_t)bp[z] % 16) == 0,
                         "slim_gemm_nt_bf16: bad operand in a pointer batch");
    }
    const int batch = g.batch > 1 ? g.batch : 1;
    SLIM_REQUIRE(batch <= 65535 && (batch == 1 || (!g.Wn && g.beta == 0.f)), "slim_gemm_nt_bf16: bad batch arguments");
    const int mx = g.M > g.N ? g.M : g.N, mn = g.M < g.N ? g.M : g.N;
    const int fam = g.Wn ? FAM_RESIDUAL
                    : (mx <= 1536 && g.Kd > 1536) ? FAM_GRAM       // small output, long reduction
                    : (mx <= 1536) ? FAM_SMALL                      // l x l x l
                    : (g.Kd <= 1536) ? FAM_APPLY                    // tall x (l x l)
                    : (mn <= 1536) ? FAM_SKETCH                     // (l x K) x (N x K)^T products with E
                    : FAM_BIG;
    int tile = gemm_pick_tile(fam);
    // the tall tile where it wastes no more rows than the square one: the (l x Dm) products of the subspace iteration and
    // the projection, l = 576 or 1152 (bf16 outputs, no split-K, no residual / polynomial epilogue)
    const bool tall_ok = g_tile_override < 0 && g_family_tile[fam] < 0 && g_tall_tile && fam == FAM_SKETCH && g.M > 128 && g.M <= 1536 &&
                         g.splits <= 1 && !g.C32 && !g.Wn && !g.poly_G && !g.sym && g.beta == 0.f;
    // ... fed by LDS-DMA when the reduction is whole 64-deep steps (SLIM_GEMM_DMA=0: register staging everywhere, an A/B aid)
    if (tall_ok && (288 - g.M % 288) % 288 <= (128 - g.M % 128) % 128) tile = g_dma_tiles && g.Kd % 64 == 0 ? TILE_DMA_288x128 : TILE_288x128;
    // the big symmetric products with mirrored bf16 output (C = E^T E of the fit) on the 256 x 128 tile fed by LDS-DMA
    {
        const bool form16 = g.C16 && g.Ct16 == g.C16 && g.ldct == g.ldc16 && !g.C32 && g.beta == 0.f;
        const bool form32 = g.C32 && !g.C16 && !g.Ct16 && batch == 1;          // (the calibration Gram matrix, accumulated into)
        if (g_tile_override < 0 && g_family_tile[fam] < 0 && g_dma_tiles && g_sym_rect && fam == FAM_BIG && g.sym && g.M == g.N && g.A == g.B &&
            g.lda == g.ldb && g.splits <= 1 && (form16 || form32) && !g.Wn && !g.poly_G && g.Kd % 64 == 0 && g.M >= 1024 &&
            (g.nptr < 1 || g.Kdp0 % 64 == 0) && (g.nptr < 2 || g.Kdp1 % 64 == 0) && (g.nptr < 3 || g.Kdp2 % 64 == 0) && (g.nptr < 4 || g.Kdp3 % 64 == 0))
            // (the fp32 form on the two-stage square tile: as fast at K = 11008 and, with 528 instead of 272 tiles, better
            // balanced at K = 4096 -- 0.61 against 0.68 ms)
            tile = form32 ? TILE_DMA_128x128_S2 : TILE_DMA_256x128;
    }
    // the short products of the fit (Gram slices, l x l x l, tall x (l x l), residual: a few K-steps, latency-bound) on the
    // two-stage 128 x 128 tile fed by LDS-DMA: the same prefetch distance as register staging without its staging
    // registers and LDS stores, two workgroups per CU as before (Gram slices 23 -> 20 us, l x l x l 18 -> 16, tall 27 -> 25)
    if (g_tile_override < 0 && g_family_tile[fam] < 0 && g_dma_tiles && tile == TILE_128x128 &&
        (fam == FAM_GRAM || fam == FAM_SMALL || fam == FAM_APPLY || fam == FAM_RESIDUAL) && g.Kd % 64 == 0 &&
        (g.nptr < 1 || g.Kdp0 % 64 == 0) && (g.nptr < 2 || g.Kdp1 % 64 == 0) && (g.nptr < 3 || g.Kdp2 % 64 == 0) && (g.nptr < 4 || g.Kdp3 % 64 == 0))
        tile = TILE_DMA_128x128_S2;
    // (a forced DMA tile falls back where its conditions do not hold: whole 64-deep K-steps; plain bf16 outputs for 288 x 128)
    if (kTileStages[tile] > 0 && g.Kd % 64 != 0) tile = TILE_128x128;
    if (tile == TILE_DMA_288x128 && (g.splits > 1 || g.C32 || g.Wn || g.poly_G || g.sym || g.beta != 0.f)) tile = TILE_DMA_128x128;
    const int BM = kTileM[tile], BN = kTileN[tile], NT = kTileThreads[tile];
    // symmetric products: with split-K the reduction mirrors the lower tiles; without it the workgroups of the upper
    // tiles write their tile and its transpose, which the caller asks for by aiming C^T at C itself (bf16 outputs only)
    const bool sym_shape = g.M == g.N && (BM == BN || tile == TILE_DMA_256x128) && g.A == g.B && g.lda == g.ldb;
    const int tiles_m_early = (g.M + BM - 1) / BM;
    const bool sym_split = g.splits > 1 && !g.Ct16;
    const bool sym_mirror = g.splits <= 1 && g.C16 && g.Ct16 == g.C16 && g.ldct == g.ldc16 && !g.C32 && !g.Wn && !g.poly_G && g.beta == 0.f;
    // ... or, for an fp32 output alone (possibly accumulated into: beta), its fp32 transpose
    const bool sym_mirror32 = g.splits <= 1 && g.C32 && !g.C16 && !g.Ct16 && !g.Wn && !g.poly_G && batch == 1 && (NT == 256 || tile == TILE_DMA_256x128);
    if (g.sym && !(sym_shape && (sym_split || sym_mirror || sym_mirror32))) g.sym = 0;
    // (the rectangular tile serves the mirrored forms only, bf16 or fp32, not the split-K one)
    if (g.sym && tile == TILE_DMA_256x128 && !(sym_shape && (sym_mirror || sym_mirror32))) g.sym = 0;
    g.mirror32 = g.sym && sym_mirror32 && !sym_split ? 1 : 0;
    g.sym_block = g.sym && tiles_m_early >= 16 ? g_sym_block : 0;
    const int64_t tiles_m = (g.M + BM - 1) / BM;
    int64_t tiles = g.sym ? tiles_m * (tiles_m + 1) / 2 : tiles_m * ((g.N + BN - 1) / BN);
    if (g.sym && BM == 2 * BN) {
        const int64_t tiles_n = (g.N + BN - 1) / BN;
        tiles = 0;
        for (int64_t tm = 0; tm < tiles_m; ++tm) tiles += tiles_n - 2 * tm;
    }
    SLIM_REQUIRE(tiles < ((int64_t)1 << 31), "slim_gemm_nt_bf16: too many tiles");
    const int loop_bytes = kTileStages[tile] > 0 ? kTileStages[tile] * (BM + BN) * 128
                                                 : 2 * (BM + BN) * (kTileBK[tile] + LDS_PAD) * (int)sizeof(bf16_t);
    const int epilogue_bytes = BM * (BN + 1) * (int)sizeof(float);       // the output tile staged as fp32
    const int lds_bytes = loop_bytes > epilogue_bytes ? loop_bytes : epilogue_bytes;
    static const char* const names[FAM_COUNT] = {"gemm_nt_gram", "gemm_nt_small", "gemm_nt_apply", "gemm_nt_sketch",
                                                 "gemm_nt_big", "gemm_nt_residual"};
    gemm_kernel_t kernel = nullptr;
    switch (fam) {
        case FAM_GRAM: kernel = gemm_kernel_of<FAM_GRAM>(tile); break;
        case FAM_SMALL: kernel = gemm_kernel_of<FAM_SMALL>(tile); break;
        case FAM_APPLY: kernel = gemm_kernel_of<FAM_APPLY>(tile); break;
        case FAM_SKETCH: kernel = gemm_kernel_of<FAM_SKETCH>(tile); break;
        case FAM_BIG: kernel = gemm_kernel_of<FAM_BIG>(tile); break;
        default: kernel = gemm_kernel_of<FAM_RESIDUAL>(tile); break;
    }
    SLIM_HIP(slim_allow_dynamic_lds((const void*)kernel, lds_bytes));
    const char* scope = names[fam];
    const double out_bytes = (double)g.M * g.N * ((g.C32 ? 4.0 : 0.0) + (g.C16 ? 2.0 : 0.0) + (g.Ct16 ? 2.0 : 0.0) + (g.Wn ? 5.0 : 0.0));
    if (g.splits > 1) {
        SLIM_REQUIRE(g.slab && !g.Wn && !(g.Ct16 && (g.C32 || g.beta != 0.f)),
                     "slim_gemm_nt_bf16: split-K supports C_f32 / C_bf16 outputs, or C_bf16 / C^T_bf16");
        {
            // the product proper (its own scope, so that the scope's time per launch is the kernel's)
            SLIM_PROF(scope, stream, (2.0 * ((double)g.M + g.N) * g.Kd + 4.0 * g.splits * (double)g.M * g.N) * batch,
                      2.0 * g.M * g.N * g.Kd * batch);
            dim3 grid((unsigned)tiles, (unsigned)g.splits, (unsigned)batch);
            hipLaunchKernelGGL(kernel, grid, dim3(NT), lds_bytes, (hipStream_t)stream, g);
            SLIM_LAUNCH_CHECK();
        }
        SLIM_PROF("splitk_reduce", stream, (4.0 * g.splits * (double)g.M * g.N + out_bytes) * batch);
        if (g.Ct16) {
            dim3 tgrid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)batch);
            SLIM_REQUIRE(tgrid.y <= 65535, "slim_gemm_nt_bf16: too many rows for the transposing reduction");
            splitk_reduce_t_kernel<<<tgrid, 256, 0, (hipStream_t)stream>>>(g.slab, g.splits, g.M, g.N, g.alpha, g.C16, g.ldc16,
                                                                           g.Ct16, g.ldct, g.sSlab, g.sC16, g.sCt);
            SLIM_LAUNCH_CHECK();
            return SLIM_OK;
        }
        dim3 rgrid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)batch);
        SLIM_REQUIRE(rgrid.y <= 65535, "slim_gemm_nt_bf16: too many rows for the split-K reduction");
        splitk_reduce_kernel<<<rgrid, 256, 0, (hipStream_t)stream>>>(g.slab, g.splits, g.M, g.N, g.alpha, g.beta, g.C32, g.ldc32,
                                                                     g.C16, g.ldc16, g.sSlab, g.sC32, g.sC16, g.sym ? BM : 0, g.c16_diag);
        SLIM_LAUNCH_CHECK();
        return SLIM_OK;
    }
    SLIM_PROF(scope, stream, (2.0 * ((double)g.M + g.N) * g.Kd + out_bytes) * batch, 2.0 * g.M * g.N * g.Kd * batch);
    hipLaunchKernelGGL(kernel, dim3((unsigned)tiles, 1, (unsigned)batch), dim3(NT), lds_bytes, (hipStream_t)stream, g);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// Split-K variant for products with a small output and a long reduction (the l x l Gram matrices of
// the fit: 25 output tiles would leave most of the chip idle).  `slab` holds splits * M * N floats.
extern "C" int slim_gemm_nt_bf16_splitk(const void* A, int64_t lda, const void* Bt, int64_t ldb, int M, int N, int Kd,
                                        float alpha, float* C_f32, int64_t ldc32, void* C_bf16, int64_t ldc16,
                                        int splits, float* slab, void* stream) {
    SLIM_REQUIRE(splits >= 1 && (splits == 1 || slab), "slim_gemm_nt_bf16_splitk: bad split arguments");
    GemmArgs g{};
    g.A = (const bf16_t*)A;
    g.lda = lda;
    g.B = (const bf16_t*)Bt;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.Kd = Kd;
    g.alpha = alpha;
    g.C32 = C_f32;
    g.ldc32 = ldc32;
    g.C16 = (bf16_t*)C_bf16;
    g.ldc16 = ldc16;
    const int nk = (Kd + BK - 1) / BK;
    if (splits > nk) splits = nk;
    g.ksteps_per_split = (nk + splits - 1) / splits;
    g.splits = (nk + g.ksteps_per_split - 1) / g.ksteps_per_split;
    g.slab = slab;
    g.sym = (A == Bt && lda == ldb && M == N);      // X X^T: the tiles below the diagonal are mirrored, not recomputed
    return slim_gemm_launch(g, stream);
}

// Split-K with the bf16 result and its transpose as outputs (the form the sketch products of the fit use).
extern "C" int slim_gemm_nt_bf16_splitk_t(const void* A, int64_t lda, const void* Bt, int64_t ldb, int M, int N, int Kd,
                                          float alpha, void* C_bf16, int64_t ldc16, void* Ct_bf16, int64_t ldct, int splits,
                                          float* slab, void* stream) {
    SLIM_REQUIRE(splits >= 1 && (splits == 1 || slab) && Ct_bf16, "slim_gemm_nt_bf16_splitk_t: bad arguments");
    GemmArgs g{};
    g.A = (const bf16_t*)A;
    g.lda = lda;
    g.B = (const bf16_t*)Bt;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.Kd = Kd;
    g.alpha = alpha;
    g.C16 = (bf16_t*)C_bf16;
    g.ldc16 = ldc16;
    g.Ct16 = (bf16_t*)Ct_bf16;
    g.ldct = ldct;
    g.slab = slab;
    gemm_plan_splits(g, splits);
    if (g.splits <= 1) g.slab = nullptr;
    return slim_gemm_launch(g, stream);
}

extern "C" int slim_gemm_nt_bf16(const void* A, int64_t lda, const void* Bt, int64_t ldb, int M, int N, int Kd,
                                 float alpha, float beta, float* C_f32, int64_t ldc32, void* C_bf16, int64_t ldc16,
                                 void* Ct_bf16, int64_t ldct, void* stream) {
    SLIM_REQUIRE(beta == 0.f || C_f32, "slim_gemm_nt_bf16: beta needs C_f32");
    GemmArgs g{};
    g.beta = beta;
    g.A = (const bf16_t*)A;
    g.lda = lda;
    g.B = (const bf16_t*)Bt;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.Kd = Kd;
    g.alpha = alpha;
    g.C32 = C_f32;
    g.ldc32 = ldc32;
    g.C16 = (bf16_t*)C_bf16;
    g.ldc16 = ldc16;
    g.Ct16 = (bf16_t*)Ct_bf16;
    g.ldct = ldct;
    // X X^T with the same buffer given for C and C^T (bf16 only): the tiles on and above the diagonal are computed and
    // mirrored, as the fit does for its Gram matrices
    g.sym = A == Bt && lda == ldb && M == N && C_bf16 && Ct_bf16 == C_bf16 && ldct == ldc16 && !C_f32 ? 1 : 0;
    return slim_gemm_launch(g, stream);
}

// Wn[N,K] = mask ? 0 : bf16(W - bf16(U[N,r] * L[K,r]^T)) : the adapter product of slim/lora.py:199
// fused with the subtraction and masking of :200-201, so the N x K low-rank matrix is never stored.
extern "C" int slim_lowrank_residual(const void* U, const void* L, int N, int K, int r, const void* W,
                                     const uint8_t* mask, void* Wn, void* LR_bf16, void* stream) {
    SLIM_REQUIRE(W && Wn, "slim_lowrank_residual: null pointer");
    GemmArgs g{};
    g.A = (const bf16_t*)U;
    g.lda = r;
    g.B = (const bf16_t*)L;
    g.ldb = r;
    g.M = N;
    g.N = K;
    g.Kd = r;
    g.alpha = 1.f;
    g.C16 = (bf16_t*)LR_bf16;
    g.ldc16 = K;
    g.W = (const bf16_t*)W;
    g.mask = mask;
    g.Wn = (bf16_t*)Wn;
    g.ldw = K;
    return slim_gemm_launch(g, stream);
}

// ---- bf16 transpose through a padded LDS tile ----------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t R, int64_t C,
                                                             bf16_t* __restrict__ out) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[r * C + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[c * R + r] = tile[tx][i];
    }
}

extern "C" int slim_transpose_bf16(const void* in, int64_t R, int64_t C, void* out, void* stream) {
    SLIM_REQUIRE(in && out && R > 0 && C > 0, "slim_transpose_bf16: bad arguments");
    SLIM_REQUIRE(ceil_div64(R, 64) <= 65535, "slim_transpose_bf16: too many rows");
    SLIM_PROF("transpose", stream, 4.0 * R * C);
    dim3 grid((unsigned)ceil_div64(C, 64), (unsigned)ceil_div64(R, 64));
    transpose_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, R, C, (bf16_t*)out);
    SLIM_LAUNCH_CHECK();
    return SLIM_OK;
}

// ---- Gram statistic (reference slim/sparsegpt.py:28-39) -------------------------------------------
// H = H * n/(n+b) + (2/(n+b)) X^T X.  X [T, K] is transposed chunk-wise into the workspace so the
// product runs in the NT form with the token dimension contiguous.
#define GRAM_CHUNK 4096
extern "C" int64_t slim_gram_workspace_bytes(int64_t T, int64_t K) {
    if (T <= 0 || K <= 0) return 0;
    const int64_t c = T < GRAM_CHUNK ? ((T + 7) / 8) * 8 : GRAM_CHUNK;
    return K * c * 2;
}

__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16_t* __restrict__ in, int64_t R, int64_t C,
                                                            int64_t ld, bf16_t* __restrict__ out, int64_t Rpad) {
    __shared__ bf16_t tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[r * ld + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < Rpad) out[c * Rpad + r] = tile[tx][i];
    }
}

extern "C" int slim_gram_accum(const void* x, int64_t T, int64_t K, int64_t ld, float* H, int64_t nsamples_before,
                               int64_t batch, void* workspace, int64_t workspace_bytes, void* stream) {
    SLIM_REQUIRE(x && H && workspace && T > 0 && K > 0 && batch > 0 && nsamples_before >= 0, "slim_gram_accum: bad arguments");
    SLIM_REQUIRE(K % 8 == 0 && ld >= K, "slim_gram_accum: K must be a multiple of 8");
    SLIM_REQUIRE(workspace_bytes >= slim_gram_workspace_bytes(T, K), "slim_gram_accum: workspace too small");
    const double n1 = (double)(nsamples_before + batch);
    const float alpha = (float)(2.0 / n1);
    const float beta = (float)((double)nsamples_before / n1);
    const bf16_t* xb = (const bf16_t*)x;
    bf16_t* xt = (bf16_t*)workspace;
    bool first = true;
    for (int64_t t0 = 0; t0 < T; t0 += GRAM_CHUNK) {
        const int64_t rows = T - t0 < GRAM_CHUNK ? T - t0 : GRAM_CHUNK;
        const int64_t rpad = ((rows + 7) / 8) * 8;
        dim3 grid((unsigned)ceil_div64(K, 64), (unsigned)ceil_div64(rpad, 64));
        transpose_pad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(xb + t0 * ld, rows, K, ld, xt, rpad);
        SLIM_LAUNCH_CHECK();
        GemmArgs g{};
        g.A = xt;
        g.lda = rpad;
        g.B = xt;
        g.ldb = rpad;
        g.M = (int)K;
        g.N = (int)K;
        g.Kd = (int)rpad;
        g.alpha = alpha;
        g.beta = first ? beta : 1.f;
        g.C32 = H;
        g.ldc32 = K;
        g.sym = 1;            // X^T X: the tiles on and above the diagonal, mirrored (H is symmetric on entry: zeros or a previous H)
        int rc = slim_gemm_launch(g, stream);
        if (rc) return rc;
        first = false;
    }
    return SLIM_OK;
}
