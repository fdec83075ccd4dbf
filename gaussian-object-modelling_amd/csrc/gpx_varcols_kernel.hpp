// gpx_varcols_kernel.hpp -- the variance contraction of SMALL models (the reference's own sizes: N = 166 .. 724 training points,
// SURVEY section 0; anything up to VARCOLS_MAX_N rows) as one wave per workgroup that keeps EVERY ROW of the product
// resident:
//
//   v[q] = k(0) - sum_m (w[m][q])^2 / D_m ,   w = X K'^T (+ the low-rank fit, fp64),   X = L^-1 lower triangular
//
// (the reference's `cholesker.solve(Kqp^T)` + `Kqp * V` + `diagonal()`, gp_regressor.hpp:316-319).  The 128 x 128 tiles of
// gpx_vargemm.hip are built for N = 16384, where a row tile sees thousands of k; on a model of 277 points they
//   * exploit the triangle of X only per 128-row tile (1.5 x the MFMAs of the triangle at N = 277),
//   * pay one fp64 epilogue, one partial-sum store and the first-load latency per 128 x 128 tile,
//   * read the operand K' from HBM once per row tile, after a separate kernel has written it there.
// Here a wave owns CF column fragments (16 CF queries) and ALL row fragments of the model: acc[FS][CF] fragments of
// v_mfma_f32_16x16x4_f32 in the 256 AGPRs (CF = 2: 32 row fragments = 512 rows).  The k loop runs over 16-deep chunks and
// is unrolled over the triangle: chunk c multiplies only the row fragments i >= c -- the skipping is per 16-row fragment
// (MFMA work = F (F + 1) / 2 fragment-chunks for F = ceil(N / 16) against the algorithmic F^2 / 2: 1.06 x at N = 277).
// The triangle is laid out for FS fragments and a model with fewer enters it at chunk FS - F (its fragments sit at the END
// of the static triangle): one uniform branch per chunk, none per MFMA group.  Models with more than FS row fragments
// take several passes over row blocks, the partial block first, every later block with all FS fragments and a rolled
// rectangular part in front of its triangle.
// A fragments (16 rows x 16 k of X, one 16-byte buffer load per lane) live in one register slot per row fragment that is
// refilled for the next chunk right after its MFMAs: a prefetch distance of one chunk with FS x 4 VGPRs; X is <= 4 MB and
// stays in L2.
// B fragments: with GEN the wave forms k(|q - p|) - fit for its own queries in registers, chunk by chunk -- the training
// points sit in LDS (12 bytes each), a lane evaluates 4 CF pairs per chunk, and the evaluation of chunk c + 1 is cut into
// stages of ~6 VALU instructions that are placed one behind each MFMA of chunk c: the operand never exists in memory
// (no kqp launch, no HBM round trip of N x 4 bytes per query).  Without GEN (operands formed in fp64, e.g. the thin plate)
// the fragments are read from the operand buffer as before.
// The fp64 epilogue (fit added back on the fp64 matrix pipe, w^2 / D, column sums) of a row fragment is issued INSIDE the
// triangle, right after the fragment's diagonal chunk has made it final: its fp64 MFMAs take their turn in the matrix
// pipe, everything else runs beside the fp32 MFMAs that follow.  Column sums stay in registers across passes and v is
// written directly: no partial sums, no var_finish launch.
// Every loaded value is consumed inside the chunk that loads it (values that cross a chunk boundary meet hipcc's merge of
// the run-time entry path with the fall-through path: copies behind a vmcnt(0)).
// Asm MFMAs are invisible to hipcc's hazard recogniser: tests/test_codeobj.py disassembles the loops of this file.
// (Kernel template only; gpx_varcols.hip instantiates and launches it.)
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpx_cov.hpp"
#include "gpx_internal.hpp"

namespace gpx {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

struct VarColsDev {
    const float *X;  // inverse factor, fp32, [np][ldx]
    long ldx;
    int x_rows;      // rows of X that may be addressed (np)
    const float *Kq;  // !GEN: operand K' of this batch, [nq_tile][ldk]
    long ldk;
    const double *rowcorr;  // [VAR_NCORR][ldrc]
    long ldrc;
    const double *colcoef;  // [VAR_NCOEF][ldcc], query index relative to the batch
    long ldcc;
    const double *dinv64;
    double *v;  // v[q], q relative to the batch
    double k0;
    long nq_valid;
    int nfrag;  // F = ceil(n / 16): row fragments (= k chunks) that hold data
    // GEN: the centred fp32 points, the queries and the covariance function
    const float *px, *py, *pz;
    const double *qx, *qy, *qz;
    double cen[3];
    int n;
    Cov<float> cov;
    int compact_coef;  // GEN: colcoef = [3][ldcc] rows a_q, b_q, c_q; the 14 coefficients are derived here (var_fit_coefs)
#ifdef VC_TIMING
    long long *dbg;  // diagnostic build (make EXTRA=-DVC_TIMING): in-kernel time stamps
#endif
};

// Shape: FS row-fragment slots x CF column fragments of accumulators.  Shipped: 12 x 2 (96 AGPRs) in 256 registers, i.e. TWO
// waves per SIMD.  Measured against one wave per SIMD with 32 x 2, 21 x 3 and 16 x 4 fragments (2^19 queries, Matern-5/2,
// profiles/r04_var_cols_shapes.txt): N = 277 0.610 vs 0.669 / 0.643 / 0.643 ms, N = 512 1.49 vs 1.48 / 1.54 / 1.51,
// N = 1024 4.91 vs 4.71 / 5.10 / 5.11 -- within 4 % of the best everywhere, and a kernel of 12 chunks compiles in 7 s
// where the 32-chunk triangle takes 3 minutes.
constexpr int VC_FS = 12, VC_CF = 2;
constexpr int VC_MAX_FRAG = VARCOLS_MAX_N / 16;

// one MFMA on fragment (row il, column j): inline asm with the accumulator tied in place
#define VC_MFMA(ACC_, A_, B_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_))
// MFMA number M_ (0 .. 4 CF - 1) of row fragment IL_ on the current chunk: k step outermost, so that an accumulator is
// touched again CF MFMAs later
#define VC_MFMA1(IL_, M_) VC_MFMA(acc[IL_][(M_) % CF], a[IL_][(M_) / CF], bq[(M_) % CF][(M_) / CF])
// the first MFMA of a chunk: its B operand may have been written by a VALU move a moment ago (bq = bn at the end of the
// previous chunk), and hipcc pads no hazard whose consumer sits inside an asm string -- the two wait states between a VALU
// write and the MFMA that reads it go inside the string, behind every instruction hipcc can place in front of the statement
#define VC_MFMA1_FIRST(IL_) \
    asm volatile("s_nop 1\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[IL_][0]) : "v"(a[IL_][0]), "v"(bq[0][0]))
// ties: nothing that reads the named registers may be scheduled above the statement (asm statements keep their order)
#define VC_TIE_ACC(IL_, NOPS_)                                                                                \
    {                                                                                                         \
        if constexpr (CF == 1)                                                                                \
            asm volatile(NOPS_ : "+a"(acc[IL_][0]));                                                          \
        else if constexpr (CF == 2)                                                                           \
            asm volatile(NOPS_ : "+a"(acc[IL_][0]), "+a"(acc[IL_][1]));                                       \
        else if constexpr (CF == 3)                                                                           \
            asm volatile(NOPS_ : "+a"(acc[IL_][0]), "+a"(acc[IL_][1]), "+a"(acc[IL_][2]));                    \
        else                                                                                                  \
            asm volatile(NOPS_ : "+a"(acc[IL_][0]), "+a"(acc[IL_][1]), "+a"(acc[IL_][2]), "+a"(acc[IL_][3])); \
    }
#define VC_TIE_D(NOPS_)                                                           \
    {                                                                             \
        if constexpr (CF == 1)                                                    \
            asm volatile(NOPS_ : "+v"(d[0]));                                     \
        else if constexpr (CF == 2)                                               \
            asm volatile(NOPS_ : "+v"(d[0]), "+v"(d[1]));                         \
        else if constexpr (CF == 3)                                               \
            asm volatile(NOPS_ : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]));             \
        else                                                                      \
            asm volatile(NOPS_ : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])); \
    }

template <int CF, int FS, bool GEN, int KID>
__global__ __attribute__((aligned(256))) __launch_bounds__(64, (FS * CF <= 24 ? 2 : 1)) void var_cols_kernel(VarColsDev g)
{
    constexpr int NM = 4 * CF;                // MFMAs of one row fragment on one chunk
    constexpr int NSTAGE = GEN ? 3 * NM : 0;  // evaluation stages per chunk: 2 CF pairs of points per lane, six stages each
    const int lane = threadIdx.x;
    const int r16 = lane & 15, lg = lane >> 4;
    const long q0 = (long)blockIdx.x * (16 * CF);
    const int F = g.nfrag;

    char *abase = const_cast<char *>(reinterpret_cast<const char *>(g.X));
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(abase, 0, (int)((long)g.x_rows * g.ldx * 4), 0x00020000);
    const int aoff = (int)((r16 * g.ldx + 4 * lg) * 4);
    const int astep = (int)(16 * g.ldx * 4);
#define VC_LOAD_A(RF_, KC_) __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff, (RF_) * astep + (KC_) * 64, 0))
    // (!GEN) the operand rows of this wave's queries
    char *bbase = const_cast<char *>(reinterpret_cast<const char *>(GEN ? g.X : g.Kq + (size_t)q0 * g.ldk));
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(bbase, 0, (int)(16L * CF * g.ldk * 4), 0x00020000);
    const int boff = (int)((r16 * g.ldk + 4 * lg) * 4);
    const int bstep = (int)(16 * g.ldk * 4);
#define VC_LOAD_B(J_, KC_) __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(brsrc, boff, (J_) * bstep + (KC_) * 64, 0))

    // (GEN) the model's centred fp32 points in LDS, four per f4v: chunk kc of a lane group lg is entry 4 kc + lg
    __shared__ f4v lpx[GEN ? 4 * VC_MAX_FRAG : 1], lpy[GEN ? 4 * VC_MAX_FRAG : 1], lpz[GEN ? 4 * VC_MAX_FRAG : 1];
    // the lane's queries (centred coordinates, a_q b_q c_q of the fit), 6 CF floats per lane, also in LDS: read back by the
    // stages that use them
    __shared__ float lq[GEN ? 6 * CF * 64 : 1];
    if constexpr (GEN) {
        for (int i = lane; i < 4 * F; i += 64) {
            lpx[i] = *reinterpret_cast<const f4v *>(g.px + 4 * i);
            lpy[i] = *reinterpret_cast<const f4v *>(g.py + 4 * i);
            lpz[i] = *reinterpret_cast<const f4v *>(g.pz + 4 * i);
        }
#pragma unroll
        for (int j = 0; j < CF; ++j) {
            // (columns past the last query are computed on the last query's data and never written)
            const long q = min(q0 + 16 * j + r16, g.nq_valid - 1);
            lq[(6 * j + 0) * 64 + lane] = (float)(g.qx[q] - g.cen[0]);
            lq[(6 * j + 1) * 64 + lane] = (float)(g.qy[q] - g.cen[1]);
            lq[(6 * j + 2) * 64 + lane] = (float)(g.qz[q] - g.cen[2]);
            const size_t frow = g.compact_coef ? 0 : VAR_NCORR;  // row of a_q in the coefficient array
            lq[(6 * j + 3) * 64 + lane] = (float)g.colcoef[(size_t)frow * g.ldcc + q];
            lq[(6 * j + 4) * 64 + lane] = (float)g.colcoef[(size_t)(frow + 1) * g.ldcc + q];
            lq[(6 * j + 5) * 64 + lane] = (float)g.colcoef[(size_t)(frow + 2) * g.ldcc + q];
        }
    }

    // column side of the fp64 epilogue: step s of a fragment's rank-14 product uses vector c = 4 s + lg (zero from 14 on)
    // (kept in LDS, 4 CF doubles per lane, and read back one column fragment at a time in front of its fp64 MFMAs: 8 CF
    // registers that would otherwise be held for the whole kernel)
    __shared__ double lcb[4 * CF * 64];
    if (GEN && g.compact_coef) {
        // the 14 coefficients from (a_q, b_q, c_q) and the centred query, exactly as var_fit_kernel forms them (fp32 operand:
        // q' and a, b, c are the float-rounded values) -- 136 bytes per query that never travel through memory
#pragma unroll
        for (int j = 0; j < CF; ++j) {
            const long q = min(q0 + 16 * j + r16, g.nq_valid - 1);
            const double ax = (double)(float)(g.qx[q] - g.cen[0]), ay = (double)(float)(g.qy[q] - g.cen[1]),
                         az = (double)(float)(g.qz[q] - g.cen[2]);
            double cf[VAR_NCORR];
            var_fit_coefs(g.colcoef[q], g.colcoef[g.ldcc + q], g.colcoef[2 * g.ldcc + q], ax, ay, az, cf);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < VAR_NCORR; ++c)
                    v = (4 * s + lg == c) ? cf[c] : v;
                lcb[(j * 4 + s) * 64 + lane] = v;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < CF; ++j) {
                const int c = 4 * s + lg;
                const long q = GEN ? min(q0 + 16 * j + r16, g.nq_valid - 1) : q0 + 16 * j + r16;
                lcb[(j * 4 + s) * 64 + lane] = c < VAR_NCORR ? g.colcoef[(size_t)c * g.ldcc + q] : 0.0;
            }
    }
    double sj[CF];
#pragma unroll
    for (int j = 0; j < CF; ++j)
        sj[j] = 0.0;
    // v_mfma_f64_16x16x4_f64 hands rows lg, lg + 4, lg + 8, lg + 12 of a fragment to a lane where the fp32 form hands rows
    // 4 lg .. 4 lg + 3: the A operand of the epilogue is fed with its rows permuted (operand row rho = tile row
    // 4 (rho & 3) + (rho >> 2)), so that result register r of a lane is row 4 lg + r -- the accumulators' layout
    const int prow = 4 * (r16 & 3) + (r16 >> 2);

    // (GEN) The evaluation of a chunk: a lane forms 4 CF values k(|q_j - p_e|) - fit, as 2 CF PAIRS of points (packed fp32
    // arithmetic: v_pk_*), each pair in six stages of at most ~28 issue cycles -- what fits behind one 32-cycle MFMA
    // without the matrix pipe running dry (a transcendental is 16 cycles, a packed or plain fp32 operation 4):
    //   0: differences and squared distance (6 packed)       1, 2: the two square roots
    //   3: decay argument, first exponential                  4: second exponential
    //   5: amplitude, polynomial, fit, difference (7 packed), mask, store into the fragment
    // Same arithmetic as cov_k<float, KID, MathAcc> / kqp_kernel (gpx_cov.hpp, gpx_pairwise.hip).
    int p4_idx = 0;  // LDS entry (4 kc + lg) of the lane's four points of the chunk being formed
    f2v e_d2 = {0.f, 0.f}, e_d = {0.f, 0.f}, e_t = {0.f, 0.f}, e_arg = {0.f, 0.f}, e_ex = {0.f, 0.f};
    f4v bq[CF], bn[CF];  // B fragments of the current / the next chunk
#pragma unroll
    for (int j = 0; j < CF; ++j)
        bq[j] = bn[j] = f4v{0.f, 0.f, 0.f, 0.f};
    // stage k of the chunk whose first point is gj0 (mask: the chunk holds points past n)
    auto eval_stage = [&](int k, int gj0, bool mask) {
        const int pp = k / 6, sub = k % 6, j = pp / 2, h = pp % 2;
        if (sub == 0) {
            const f2v px2 = h ? lpx[p4_idx].hi : lpx[p4_idx].lo, py2 = h ? lpy[p4_idx].hi : lpy[p4_idx].lo,
                      pz2 = h ? lpz[p4_idx].hi : lpz[p4_idx].lo;
            const f2v dx = lq[(6 * j + 0) * 64 + lane] - px2, dy = lq[(6 * j + 1) * 64 + lane] - py2, dz = lq[(6 * j + 2) * 64 + lane] - pz2;
            e_d2 = dx * dx + dy * dy + dz * dz;
        } else if (sub == 1) {
            e_d.x = __builtin_amdgcn_sqrtf(e_d2.x);
        } else if (sub == 2) {
            e_d.y = __builtin_amdgcn_sqrtf(e_d2.y);
        } else if (sub == 3) {
            if constexpr (KID != GPX_KERNEL_THINPLATE) {
                e_t = g.cov.s * e_d;
                e_arg = -e_t * 1.44269504088896340736f;
                e_ex.x = __builtin_amdgcn_exp2f(e_arg.x);
            }
        } else if (sub == 4) {
            if constexpr (KID != GPX_KERNEL_THINPLATE)
                e_ex.y = __builtin_amdgcn_exp2f(e_arg.y);
        } else {
            f2v kv;
            if constexpr (KID == GPX_KERNEL_THINPLATE) {
                const f2v e = e_d - g.cov.R;
                kv = e * e * (2.0f * e_d + g.cov.R);
            } else {
                const f2v e = g.cov.a * e_ex;
                if constexpr (KID == GPX_KERNEL_MATERN32)
                    kv = e * (1.0f + e_t);
                else if constexpr (KID == GPX_KERNEL_MATERN52)
                    kv = e * (1.0f + e_t + e_t * e_t * (float)(1.0 / 3.0));
                else
                    kv = e;
            }
            kv -= lq[(6 * j + 3) * 64 + lane] + e_d2 * (lq[(6 * j + 4) * 64 + lane] + lq[(6 * j + 5) * 64 + lane] * e_d2);
            if (mask) {
                kv.x = (gj0 + 4 * lg + 2 * h < g.n) ? kv.x : 0.f;
                kv.y = (gj0 + 4 * lg + 2 * h + 1 < g.n) ? kv.y : 0.f;
            }
            if (h)
                bn[j].hi = kv;
            else
                bn[j].lo = kv;
        }
    };
    auto load_points = [&](int kc) { p4_idx = 4 * kc + lg; };

#ifdef VC_TIMING
    long long ts0 = __builtin_readcyclecounter(), ts1 = 0, ts2 = 0;
    __shared__ long long tchunk[80];
    const long long trt0 = __builtin_amdgcn_s_memrealtime();
#define VC_STAMP(I_)                                 \
    if (lane == 0)                                   \
        tchunk[I_] = __builtin_readcyclecounter();
#else
#define VC_STAMP(I_)
#endif
    const int npass = (F + FS - 1) / FS;
    const int first = F - (npass - 1) * FS;  // row fragments of pass 0 (1 .. FS); every later pass has FS
#pragma nounroll
    for (int p = 0; p < npass; ++p) {
        const int nfr = p == 0 ? first : FS;
        const int f_lo = p == 0 ? 0 : first + (p - 1) * FS;  // first row fragment of the pass
        const int sh = FS - nfr;                             // the pass's fragments sit in slots sh .. FS - 1

        // Prologue: B of chunk 0 first -- the triangle is entered at a run-time chunk, and hipcc's wait for the first MFMA
        // of EVERY chunk is the stricter of the entry path and the fall-through path; with B oldest the two agree.
        f4v a[FS];
        if constexpr (GEN) {
            load_points(0);
#pragma unroll
            for (int k = 0; k < NSTAGE; ++k)
                eval_stage(k, 0, F == 1);
#pragma unroll
            for (int j = 0; j < CF; ++j)
                bq[j] = bn[j];
        } else {
#pragma unroll
            for (int j = 0; j < CF; ++j)
                bq[j] = VC_LOAD_B(j, 0);
        }
        // chunk 0 of every slot (slots below sh are never multiplied: they load row fragment 0, harmlessly)
#pragma unroll
        for (int il = 0; il < FS; ++il)
            a[il] = VC_LOAD_A(max(f_lo + il - sh, 0), 0);
        f4v acc[FS][CF];  // (zeroed while the first loads are in flight)
#pragma unroll
        for (int i = 0; i < FS; ++i)
#pragma unroll
            for (int j = 0; j < CF; ++j)
                acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

#ifdef VC_TIMING
        ts1 = __builtin_readcyclecounter();
#endif
        // Everything that is not an MFMA is a FILLER: a piece of at most ~28 issue cycles placed behind one MFMA and pinned
        // there (sched_barrier: hipcc would otherwise merge neighbouring pieces -- it packs them, correctly, but two MFMAs back
        // to back followed by two pieces leave the matrix pipe idle for the length of a piece).
#define VC_PIN
        // rectangular part (passes after the first: sh = 0): chunks in front of the block's own columns, all FS fragments
#pragma nounroll
        for (int c = 0; c < f_lo; ++c) {
#pragma unroll
            for (int il = 0; il < FS; ++il) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int mm = il * NM + m;
                    if (mm == 0)
                        VC_MFMA1_FIRST(il);
                    else
                        VC_MFMA1(il, m);
                    if (mm == 0) {  // the next chunk's points / operand fragments, behind the chunk's first MFMA
                        if constexpr (GEN) {
                            load_points(c + 1);
                        } else {
#pragma unroll
                            for (int j = 0; j < CF; ++j)
                                bn[j] = VC_LOAD_B(j, c + 1);
                        }
                        VC_PIN;
                    }
                    if (GEN && mm >= NM && mm - NM < NSTAGE) {
                        eval_stage(mm - NM, 16 * (c + 1), false);  // (c + 1 < F - 1: never the chunk with the padding)
                        VC_PIN;
                    }
                }
                a[il] = VC_LOAD_A(f_lo + il, c + 1);
            }
#pragma unroll
            for (int j = 0; j < CF; ++j)
                bq[j] = bn[j];
        }
        // The triangle: static chunk cs is k chunk f_lo + cs - sh and meets the slots il >= cs.  Slot cs is FINAL after
        // its group of this chunk, and its epilogue -- w = acc + sum_c rowcorr[c][m] colcoef[c][q] on the fp64 matrix pipe,
        // then w^2 / D into the column sums -- is issued a few groups later in the same chunk: the fp64 MFMAs take their
        // turn in the matrix pipe, everything else (accumulator reads, conversions, fp64 FMAs) goes behind the fp32 MFMAs
        // that follow, two values at a time.  Fillers of a chunk, by MFMA number mm (group gi = mm / NM):
        //   mm = 0, 1         next chunk's points (LDS) or operand fragments | row vectors of slot cs
        //   mm = NM ..        the 3 NM evaluation stages of the next chunk (GEN)
        //   group GA          (its last MFMAs) accumulator reads of slot cs; behind the group the 4 CF fp64 MFMAs, the slot's 1 / D
        //   group GB          2 CF pieces of the fp64 square-and-sum
        // A chunk with too few groups runs what is left behind its last MFMA.
#pragma unroll
        for (int cs = 0; cs < FS; ++cs) {
            if (cs >= sh) {
                VC_STAMP(cs)
                const int NG = FS - cs;  // groups of this chunk (a constant after unrolling)
                const int GA = NG - 1 < 4 ? NG - 1 : 4, GB = NG - 1 < 6 ? NG - 1 : 6;  // the groups the epilogue's two halves follow
                const int kc = f_lo + cs - sh;
                const bool has_next = cs + 1 < FS;
                double ra[4], rw[4];
                float t[CF][4];
                d4v d[CF];
                // piece u (0 .. 2 CF - 1) of the square-and-sum: two rows of column fragment u / 2
                auto fin_b = [&](int u) {
                    const int j = u / 2, r0 = 2 * (u % 2);
#pragma unroll
                    for (int r = r0; r < r0 + 2; ++r) {
                        const double w = (double)t[j][r] + d[j][r];
                        sj[j] = fma(w * w, rw[r], sj[j]);
                    }
                };
#pragma unroll
                for (int il = cs; il < FS; ++il) {
                    const int gi = il - cs;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const int mm = gi * NM + m;
                        if (mm == 0)
                            VC_MFMA1_FIRST(il);
                        else
                            VC_MFMA1(il, m);
                        if (mm == 0 && has_next) {
                            if constexpr (GEN) {
                                load_points(kc + 1);
                            } else {
#pragma unroll
                                for (int j = 0; j < CF; ++j)
                                    bn[j] = VC_LOAD_B(j, kc + 1);
                            }
                            VC_PIN;
                        }
                        if (mm == 1) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)  // (vectors 14, 15 do not exist: the column side is zero there)
                                ra[s4] = g.rowcorr[(size_t)min(4 * s4 + lg, VAR_NCORR - 1) * g.ldrc + 16 * kc + prow];
                            VC_PIN;
                        }
                        if (GEN && has_next && mm >= NM && mm - NM < NSTAGE) {
                            eval_stage(mm - NM, 16 * (kc + 1), cs + 1 == FS - 1);
                            VC_PIN;
                        }
                        if (GA >= 1 && gi == GA && m >= NM - CF) {
                            // accumulator reads of slot cs, one column fragment per MFMA (its last MFMAs are >= NM MFMAs back)
                            const int j = m - (NM - CF);
                            if (j == 0)
                                VC_TIE_ACC(cs, "")
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                t[j][r] = acc[cs][j][r];
                            VC_PIN;
                        }
                        if (gi == GB && GB > GA && m < 2 * CF) {
                            if (m == 0)
                                VC_TIE_D("")  // fp32 MFMAs were issued behind the fp64 ones -- NM of them when GB = GA + 2, a single one
                                              // for the 6-group chunk (GA = 4, GB = 5); what makes the read safe is that the matrix
                                              // pipe completes in order (the fp32 MFMA behind the chain cannot finish before it) and
                                              // the VALU read of d[] interlocks on the fp64 result like any VALU-after-MFMA read
                            fin_b(m);
                            VC_PIN;
                        }
                    }
                    if (il >= cs + 1)
                        a[il] = VC_LOAD_A(f_lo + il - sh, kc + 1);
                    if (gi == GA) {
                        if (GA == 0) {  // the last chunk: no MFMA behind the slot's own
                            VC_TIE_ACC(cs, "s_nop 15\n s_nop 15")
#pragma unroll
                            for (int j = 0; j < CF; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    t[j][r] = acc[cs][j][r];
                        }
#pragma unroll
                        for (int j = 0; j < CF; ++j) {
                            double cbj[4];
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                cbj[s4] = lcb[(j * 4 + s4) * 64 + lane];
                            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(d[j]) : "v"(ra[0]), "v"(cbj[0]));
#pragma unroll
                            for (int s4 = 1; s4 < 4; ++s4)
                                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(ra[s4]), "v"(cbj[s4]));
                        }
                        // (1 / D of the slot's rows: not needed before the square-and-sum)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            rw[r] = g.dinv64[16 * kc + 4 * lg + r];
                        VC_PIN;
                    }
                }
                if (GB == GA) {  // no group behind the fp64 MFMAs: wait for them
                    VC_TIE_D("s_nop 15\n s_nop 15")
#pragma unroll
                    for (int u = 0; u < 2 * CF; ++u)
                        fin_b(u);
                }
                if (has_next) {
                    // evaluation stages that found no MFMA to sit behind (the last chunks of the triangle)
                    if constexpr (GEN) {
#pragma unroll
                        for (int k = 0; k < NSTAGE; ++k)
                            if (k >= (NG - 1) * NM)
                                eval_stage(k, 16 * (kc + 1), cs + 1 == FS - 1);
                    }
#pragma unroll
                    for (int j = 0; j < CF; ++j)
                        bq[j] = bn[j];
                }
            }
        }
#undef VC_PIN
#ifdef VC_TIMING
        ts2 = __builtin_readcyclecounter();
#endif
        VC_STAMP(FS)
    }  // pass
#undef VC_LOAD_A
#undef VC_LOAD_B

#pragma unroll
    for (int j = 0; j < CF; ++j) {
        double t = sj[j];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        const long q = q0 + 16 * j + r16;
        if (lg == 0 && q < g.nq_valid)
            g.v[q] = g.k0 - t;
    }
#ifdef VC_TIMING
    if (g.dbg && lane == 0 && blockIdx.x < 8192) {
        long long ts3 = __builtin_readcyclecounter();
        long long *o = g.dbg + 8 * blockIdx.x;
        o[0] = ts0, o[1] = ts1, o[2] = ts2, o[3] = ts3;
        o[4] = __builtin_amdgcn_s_memrealtime() - trt0;
        if (blockIdx.x == 1500)
            for (int i = 0; i <= FS; ++i)
                g.dbg[8 * 8192 + i] = tchunk[i];
    }
#endif
#undef VC_STAMP
}
#undef VC_MFMA
#undef VC_MFMA1
#undef VC_MFMA1_FIRST
#undef VC_TIE_ACC
#undef VC_TIE_D


}  // namespace gpx
