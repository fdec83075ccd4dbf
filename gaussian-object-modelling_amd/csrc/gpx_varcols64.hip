// gpx_varcols64.hip -- the variance of SMALL fp64 models (the reference's own arithmetic at the reference's own sizes: the header
// shim creates GPX_PREC_F64 models by default, N = 166 .. 724 in the node) as ONE kernel per call:
//
//   v[q] = k(0) - sum_m (w[m][q])^2 / D_m ,   w = X k_q ,   X = L^-1 lower triangular          (gp_regressor.hpp:307-319)
//
// The general fp64 path (kqp_kernel -> 128 x 64 one-wave tiles -> var_finish) writes the operand k_q to HBM (8 N bytes per
// query) and reads it back per row tile, exploits the triangle of X per 128 rows only and pays an epilogue per tile: at N = 277
// its contraction kernel reaches 40 % of the fp64 MFMA peak on the flop it executes and the stage around it costs another third
// (scripts/var64_sweep.py: 6.8 ms per 2^21 queries = 34 % on the algorithmic flop).  Here, as in the fp32 small-model kernel
// (gpx_varcols_kernel.hpp) but without its fit:
//   * a wave owns 32 queries (two column fragments) and up to FS = 22 row fragments (16 rows each) of the product in
//     accumulators of v_mfma_f64_16x16x4_f64 -- 224 AGPRs + 128 VGPRs, one wave per SIMD (16 slots: 2-4 % slower at every
//     size, 24 slots: VGPRs spill into the AGPRs); models of more than 352 points take
//     passes over row blocks, the short pass first (the operand is formed once per pass up to the pass's last row);
//   * the k loop runs over 16-deep chunks; chunk c multiplies only the row fragments >= c.  That range always ENDS at the
//     pass's last fragment, so the slots count from the end and the code of a chunk is a straight line with one exit;
//   * the operand is formed in the wave: the model's points sit in LDS, a lane evaluates its eight values of the chunk (the
//     summation index of MFMA step kk in lane group g is k = 4 g + kk) with the fast fp64 sqrt of gpx_cov.hpp and the
//     table exponential of the mean kernel (ExpMean: exact to 1e-18);
//   * X is first copied into FRAGMENT ORDER (pack64_kernel, microseconds, redone per call: no state): a lane's MFMA row is
//     its lane number mod 16, so from X itself -- or from its transpose, eight-byte requests -- the L1's tag rate sets the time;
//     fragments are requested four ahead of their MFMAs into a ring of five register sets, the first four of a chunk before its
//     operand is formed;
//   * a workgroup (four waves side by side on the same fragments) is alone on its CU, so the grid is one workgroup per CU
//     and each walks over its share of the query blocks: the prologue (points, table) runs once;
//   * w^2 / D is summed per lane straight from the accumulators, the four lane groups are combined at the end, v is written
//     directly: no operand buffer, no partial sums, no finish launch;
//   * without a gradient request the mean f = sum_p alpha_p k(q, p) rides on the operand values of the last pass (which forms
//     every chunk of the model): alpha in LDS, 8 FMAs per chunk, no launch of the mean kernel (GPX_VAR_COLS64_MEAN=0: that launch).
// Measured (profiles/r05_var64_sweep.txt, 2^21 queries): N = 277 6.8 -> 3.65 ms (64 % of the fp64 MFMA peak on the algorithmic
// triangle, general path 34 %), N = 512 11.1 -> 10.1 ms (71 %), N = 724 22.0 -> 19.9 ms (74 %), N = 900 36.1 -> 30.0 ms; the
// general path, whose time is flat per 128 rows, is ahead again from ~1000 points.  Where the rest goes
// (profiles/r05_var64_parts.txt, N = 277): MFMAs + loop + epilogue 2.9 ms (the MFMAs alone: 2.4), operand 0.5, requests for X
// 0.25 -- one after the other, since nothing but scalar work hides behind an MFMA of the same SIMD (a second wave per SIMD
// with half the slots was built and measured: no gain).
// Models of up to VARCOLS64_DEFAULT_N points are routed here (the kernel's LDS arrays hold up to VARCOLS64_MAX_N points);
// GPX_VAR_COLS64=0: the general path.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gpx_cov.hpp"
#include "gpx_internal.hpp"

namespace gpx {

namespace {
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#ifndef VC64_FS
#define VC64_FS 22
#endif
#ifndef VC64_AHEAD
#define VC64_AHEAD 4
#endif
#ifndef VC64_AGPR
#define VC64_AGPR 14
#endif
// The shape of a wave's share.  ONE wave per SIMD (round 5): 32 queries (two column fragments: a fragment of X feeds 8 MFMAs) and
// 22 row-fragment slots, all 512 registers of a SIMD lane.  TWO waves per SIMD (round 6): 16 queries (one column fragment, 4 MFMAs
// per fragment of X) and VC64_FS2 slots in 256 registers each -- the same rows per pass, hence the same number of operand
// evaluations per query as the one-wave form (round 5's two-wave build halved the SLOTS and so evaluated the operand 1.7 times as
// often) and, with the same 22 slots, the same order of every sum (bit-identical results), at twice the requests for X.  What the
// second wave buys is latency: the vector work itself cannot hide (on gfx950 the fp64 MFMA runs at the fp64 VECTOR rate -- it
// occupies the lanes the vector ALU would use: profiles/r06_pmc_small_kernels.txt, SQ_VALU_MFMA_COEXEC_CYCLES), but one wave's
// waits -- LDS, the requests for X, the MFMA's own result latency in front of the epilogue -- are filled by the other.  Measured
// (profiles/r06_var64_two.txt, 2^21 queries, Matern-5/2): N = 277 3.81 -> 3.49 ms, 512 10.5 -> 9.97, 724 20.5 -> 19.65 (+4.5 .. 8.5 %).
// Accumulators: 16 slots in AGPRs + 6 in VGPRs is the ONLY split hipcc compiles without copies (it gives a kernel of 512 threads
// 128 registers of each class): builds with 20, 8 or 0 AGPR slots moved accumulators through the other class right behind asm
// MFMAs and returned wrong variances (dv up to 100 x max v) -- codeobj.py's guard fails them.
// Tried on top and dropped (ledger, round 6): the chunk's fragments of X staged once per workgroup in LDS by LDS-DMA, a chunk
// ahead (L2 requests 119 -> 36 GB per launch): 19.65 -> 21.5 ms at N = 724 -- the barrier per chunk and the DMA's own cost exceed
// what the requests cost (1.8 ms; a build whose requests always hit the L1 shows that this cost is the L2's, not their issue).
struct VC64One {
    static constexpr int CF = 2, FS = VC64_FS, WAVES = 4, AGPR = VC64_AGPR, AHEAD = VC64_AHEAD;
};
#ifndef VC64_FS2
#define VC64_FS2 22
#endif
#ifndef VC64_AHEAD2
#define VC64_AHEAD2 2
#endif
#ifndef VC64_AGPR2
#define VC64_AGPR2 16
#endif
struct VC64Two {
    static constexpr int CF = 1, FS = VC64_FS2, WAVES = 8, AGPR = VC64_AGPR2, AHEAD = VC64_AHEAD2;
};

struct VarCols64Dev {
    const double *Xp;  // X in fragment order (pack64_kernel): a wave's request for a 16 x 16 fragment is two contiguous KB
    int fp;            // fragments per side of the packed copy
    int n, nfrag;  // F = ceil(n / 16)
    const double *px, *py, *pz, *dinv;
    const double *alpha;  // with f: the GP weights -- the mean f[q] = sum_p alpha_p k(q, p) rides on the operand values of the
    double *f;            // last pass (every chunk of the model is formed there): 8 FMAs per chunk instead of a launch
    const double *qx, *qy, *qz;
    double *v;
    long nq;
    double k0;
    Cov<double> cov;
};

// f(0), f(1), ... while the slot number is below nact: nested branches, i.e. straight-line code with one way out
template <int R, int N, class Fn>
__device__ __forceinline__ void slot_chain(int nact, Fn &&f)
{
    if constexpr (R < N) {
        if (R < nact) {
            f(std::integral_constant<int, R>{});
            slot_chain<R + 1, N>(nact, f);
        }
    }
}

// DBG (diagnostic builds only, make EXTRA=-DVC64_DBG=n OUTDIR=../lib_t OBJDIR=../build_t; results are wrong by construction):
// 1 operand values not evaluated, 2 no MFMAs, 3 no requests for X, 4 neither operand nor requests, 5 every request answered by
// the L1 -- how the kernel's time splits (profiles/r05_var64_parts.txt, r06_var64_parts.txt)
#ifndef VC64_DBG
#define VC64_DBG 0
#endif
template <int KID, class CFG, int DBG = VC64_DBG>
__global__ __launch_bounds__(64 * CFG::WAVES, 1) void var_cols64_kernel(VarCols64Dev g)
{
    constexpr int FS64 = CFG::FS;        // row-fragment slots per pass
    constexpr int CF64 = CFG::CF;        // column fragments (16 queries each) per wave
    constexpr int WAVES64 = CFG::WAVES;  // waves per workgroup, side by side on the same fragments of X (L1 hits)
    constexpr int AGPR_SLOTS = CFG::AGPR;  // slots whose accumulators live in AGPRs (the others: VGPRs)
    constexpr int AHEAD64 = CFG::AHEAD, RING64 = AHEAD64 + 1;  // fragments of X requested ahead of their MFMAs; register sets
    __shared__ double lp[3][VARCOLS64_MAX_N];
    __shared__ double ld[VARCOLS64_MAX_N];
    __shared__ double la[VARCOLS64_MAX_N];
    const int lane = threadIdx.x & 63, r16 = lane & 15, lg = lane >> 4;
    const int F = g.nfrag, n = g.n;
    // The exponential kernels take e^(-s d) from the 512-entry table of the mean kernel (gpx_cov.hpp, ExpMean: 11 instructions,
    // exact to 1e-18) without their amplitude a, which goes into the row weights as a^2.  Points of the padding sit at the origin
    // with weight 0: their operand values meet zero columns of X in the rows of the model and identity rows of weight 0 below.
    constexpr bool EXPK = KID != GPX_KERNEL_THINPLATE;
    const Cov<double> cov = g.cov;
    const double w2 = EXPK ? cov.a * cov.a : 1.0;
    for (int k = threadIdx.x; k < 16 * F; k += 64 * WAVES64) {
        const bool in = k < n;
        lp[0][k] = in ? g.px[k] : 0.0, lp[1][k] = in ? g.py[k] : 0.0, lp[2][k] = in ? g.pz[k] : 0.0;
        ld[k] = in ? g.dinv[k] * w2 : 0.0;
        la[k] = (in && g.f) ? g.alpha[k] * (EXPK ? cov.a : 1.0) : 0.0;
    }
    ExpMean<KID> em;
    if constexpr (EXPK) {
        ExpTab::init(threadIdx.x, 64 * WAVES64);
        em.prep(cov);
    }
    __syncthreads();  // (the kernel's only barrier)
    // A workgroup is resident alone on its CU (512 registers per lane), so nothing would cover the prologue above if it ran once
    // per 128 queries: the grid is one workgroup per CU and each walks over its share of the query blocks.
    const long per_wg = 16L * CF64 * WAVES64, nblk = (g.nq + per_wg - 1) / per_wg;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long q0 = (blk * WAVES64 + (threadIdx.x >> 6)) * (16 * CF64);
        if (q0 >= g.nq)
            break;  // (a wave past the last query)
        double ax[CF64], ay[CF64], az[CF64];
#pragma unroll
        for (int j = 0; j < CF64; ++j) {  // (columns past the last query work on the last query's data and are not written)
            const long q = q0 + 16 * j + r16, qc = q < g.nq ? q : g.nq - 1;
            ax[j] = g.qx[qc], ay[j] = g.qy[qc], az[j] = g.qz[qc];
        }
        // passes over row blocks of at most FS64 fragments; the operand is formed once per pass up to the pass's last row, so the
        // SHORT pass comes first (F = 18: 2 + 18 chunks of operand instead of 9 + 18)
        const int npass = (F + FS64 - 1) / FS64;
        double colsum[CF64], fsum[CF64];
#pragma unroll
        for (int j = 0; j < CF64; ++j)
            colsum[j] = fsum[j] = 0.0;
        const unsigned lane_off = (unsigned)(lane * 2 * sizeof(double));  // (lg, r16) -> 16 bytes at (16 lg + r16) * 16
        int f_lo = 0;
        for (int p = 0; p < npass; ++p) {
            const int nfr = p == 0 ? F - FS64 * (npass - 1) : FS64, f_hi = f_lo + nfr;
            f64x4 acc[FS64][CF64];
            // (the AGPR accumulators are zeroed by an MFMA of zeros: every definition of them is then tied to an AGPR and the
            // register allocator gives them no second home in VGPRs -- with a plain assignment it copies eight slots in and out
            // around every MFMA statement)
            const double zero = 0.0;
#pragma unroll
            for (int il = 0; il < FS64; ++il)
#pragma unroll
                for (int j = 0; j < CF64; ++j)
                    if (il < AGPR_SLOTS)
                        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %1, 0" : "=a"(acc[il][j]) : "v"(zero));
                    else
                        acc[il][j] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma nounroll
            for (int c = 0; c < f_hi; ++c) {
                // Row fragments max(c, f_lo) .. f_hi - 1 take part: a range that always ENDS at the pass's last fragment, so
                // the slots count from the end (slot r = fragment f_hi - 1 - r) and the active ones are r < nact -- a straight
                // line of code with one exit, in which the compiler counts the requests in flight exactly.  Slices of X are
                // requested four fragments ahead of the MFMAs that use them, the first four before the operand is formed, into
                // a ring of five register sets (the set a request lands in is never the one the current MFMAs read: no copy,
                // no wait); past the last active fragment the request repeats it (an L1 hit, never used).
                const int nact = f_hi - max(c, f_lo);
                // fragment (c, f) of the packed copy: 2 KB at ((c fp + f) * 256) doubles, the lane's two halves 1 KB apart
                const int xc = (c * g.fp + f_hi - 1) * 256;
                f64x2 alo[RING64], ahi[RING64];
                auto load_a = [&](int r, f64x2 &lo, f64x2 &hi) {
                    if constexpr (DBG >= 3) {
                        lo = f64x2{ax[0], ay[0]}, hi = f64x2{az[0], ax[CF64 - 1]};
                        return;
                    }
                    // (DBG 5: every request reads the first fragment -- always an L1 hit: what the requests cost beyond the L1)
                    const char *src = reinterpret_cast<const char *>(g.Xp + (DBG == 5 ? (xc & 1) : (xc - 256 * min(r, nact - 1)))) + lane_off;
                    lo = *reinterpret_cast<const f64x2 *>(src);
                    hi = *reinterpret_cast<const f64x2 *>(src + 1024);
                };
#pragma unroll
                for (int u = 0; u < AHEAD64; ++u)
                    load_a(u, alo[u], ahi[u]);
                // the lane's operand values of the chunk: k(|q - p|), p = 16 c + 4 lg + kk, for its query of either column fragment
                double b[CF64][4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int pi = 16 * c + 4 * lg + kk;
                    const double x = lp[0][pi], y = lp[1][pi], z = lp[2][pi];
#pragma unroll
                    for (int j = 0; j < CF64; ++j) {
                        const double dx = ax[j] - x, dy = ay[j] - y, dz = az[j] - z;
                        const double d2 = dx * dx + dy * dy + dz * dz + 1e-300;
                        if constexpr (DBG == 1 || DBG == 4)
                            b[j][kk] = d2;
                        else if constexpr (EXPK)
                            b[j][kk] = em.k(MathFast::sqrt_(d2));
                        else
                            b[j][kk] = cov_k<double, KID, MathFast>(cov, d2);
                    }
                }
                if (g.f && p == npass - 1) {  // (uniform; the last pass forms the operand of every chunk)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double al = la[16 * c + 4 * lg + kk];
#pragma unroll
                        for (int j = 0; j < CF64; ++j)
                            fsum[j] = fma(b[j][kk], al, fsum[j]);
                    }
                }
                slot_chain<0, FS64>(nact, [&](auto slot) {
                    constexpr int il = decltype(slot)::value;
                    if constexpr (CF64 == 2) {
                        // (asm operands inside a lambda must be the lambda's own variables)
                        f64x4 &c0 = acc[il][0], &c1 = acc[il][CF64 - 1];
                        const double b00 = b[0][0], b01 = b[0][1], b02 = b[0][2], b03 = b[0][3];
                        const double b10 = b[CF64 - 1][0], b11 = b[CF64 - 1][1], b12 = b[CF64 - 1][2], b13 = b[CF64 - 1][3];
                        const f64x2 a0 = alo[il % RING64], a1 = ahi[il % RING64];
                        load_a(il + AHEAD64, alo[(il + AHEAD64) % RING64], ahi[(il + AHEAD64) % RING64]);
                        // (one statement: the two column fragments alternate, so consecutive MFMAs are independent, and no VALU
                        // write can be scheduled into the two wait states in front of an MFMA that reads it)
#define VC64_MFMAS(CLS_)                                                                                                 \
        asm volatile("s_nop 1\n"                                                                                             \
                     "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %1, %2, %10, %1\n"                                                              \
                     "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %1, %3, %11, %1\n"                                                              \
                     "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %1, %4, %12, %1\n"                                                              \
                     "v_mfma_f64_16x16x4_f64 %0, %5, %9, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %1, %5, %13, %1"                                                                 \
                     : CLS_(c0), CLS_(c1)                                                                                    \
                     : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(b00), "v"(b01), "v"(b02), "v"(b03), "v"(b10),     \
                       "v"(b11), "v"(b12), "v"(b13))
                        if constexpr (DBG == 2 && il < AGPR_SLOTS)
                            asm volatile("" : "+a"(c0), "+a"(c1) : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(b00), "v"(b01),
                                         "v"(b02), "v"(b03), "v"(b10), "v"(b11), "v"(b12), "v"(b13));
                        else if constexpr (DBG == 2)
                            asm volatile("" : "+v"(c0), "+v"(c1) : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(b00), "v"(b01),
                                         "v"(b02), "v"(b03), "v"(b10), "v"(b11), "v"(b12), "v"(b13));
                        else if constexpr (il < AGPR_SLOTS)
                            VC64_MFMAS("+a");
                        else
                            VC64_MFMAS("+v");
#undef VC64_MFMAS
                    } else {
                        // one column fragment: the four MFMAs of a slot form an accumulate chain (which issues at the full rate,
                        // scripts/mfma_f64_chain_probe.hip); the other wave of the SIMD fills the pipe between the statements
                        f64x4 &c0 = acc[il][0];
                        const double b00 = b[0][0], b01 = b[0][1], b02 = b[0][2], b03 = b[0][3];
                        const f64x2 a0 = alo[il % RING64], a1 = ahi[il % RING64];
                        load_a(il + AHEAD64, alo[(il + AHEAD64) % RING64], ahi[(il + AHEAD64) % RING64]);
#define VC64_MFMAS1(CLS_)                                                                                                \
        asm volatile("s_nop 1\n"                                                                                             \
                     "v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n"                                                               \
                     "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0"                                                                  \
                     : CLS_(c0)                                                                                              \
                     : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(b00), "v"(b01), "v"(b02), "v"(b03))
                        if constexpr (DBG == 2 && il < AGPR_SLOTS)
                            asm volatile("" : "+a"(c0) : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(b00), "v"(b01), "v"(b02), "v"(b03));
                        else if constexpr (DBG == 2)
                            asm volatile("" : "+v"(c0) : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(b00), "v"(b01), "v"(b02), "v"(b03));
                        else if constexpr (il < AGPR_SLOTS)
                            VC64_MFMAS1("+a");
                        else
                            VC64_MFMAS1("+v");
#undef VC64_MFMAS1
                    }
                });
            }
            // the accumulators are read by the VALU from here on: the MFMA's wait states first (hipcc pads no hazard whose producer
            // sits inside an asm string), every accumulator tied behind them (asm statements keep their order)
            asm volatile("s_nop 15\n s_nop 15" : "+a"(acc[0][0]));
#pragma unroll
            for (int il = 0; il < FS64; ++il)
#pragma unroll
                for (int j = 0; j < CF64; ++j)
                    if (il < AGPR_SLOTS)
                        asm volatile("" : "+a"(acc[il][j]));
                    else
                        asm volatile("" : "+v"(acc[il][j]));
            // w^2 / D of the pass's rows: register r of lane (lg, query) in slot il is row 16 (f_hi - 1 - il) + 4 r + lg
#pragma unroll
            for (int il = 0; il < FS64; ++il)
                if (il < nfr) {
                    const int r0 = 16 * (f_hi - 1 - il) + lg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double di = ld[r0 + 4 * r];
#pragma unroll
                        for (int j = 0; j < CF64; ++j)
                            colsum[j] = fma(acc[il][j][r] * acc[il][j][r], di, colsum[j]);
                    }
                }
            f_lo = f_hi;
        }
#pragma unroll
        for (int j = 0; j < CF64; ++j) {
            double cs = colsum[j];
            cs += __shfl_xor(cs, 16);
            cs += __shfl_xor(cs, 32);
            const long q = q0 + 16 * j + r16;
            double fs = fsum[j];
            if (g.f) {
                fs += __shfl_xor(fs, 16);
                fs += __shfl_xor(fs, 32);
            }
            if (lg == 0 && q < g.nq) {
                g.v[q] = g.k0 - cs;
                if (g.f)
                    g.f[q] = fs;
            }
        }
    }
}
// X (lower triangular, leading rows x rows part, rows a multiple of 32) in the order the variance kernel reads it:
// Xp[c][f][h][lg][r16][e] = X[16 f + r16][16 c + 4 lg + 2 h + e] for the fragments f >= c (the others are never read)
__global__ __launch_bounds__(128) void pack64_kernel(const double *X, long ldx, double *Xp, int fp)
{
    const int f = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
    if (f < c)
        return;
    const int h = t >> 6, lg = (t >> 4) & 3, r16 = t & 15;  // (128 threads: t = 64 h + 16 lg + r16)
    const f64x2 val = *reinterpret_cast<const f64x2 *>(X + (size_t)(16 * f + r16) * ldx + 16 * c + 4 * lg + 2 * h);
    *reinterpret_cast<f64x2 *>(Xp + ((size_t)(c * fp + f) * 256 + t * 2)) = val;
}
}  // namespace

size_t var_cols64_ws_bytes(int n)
{
    const size_t rows = ((size_t)n + 31) / 32 * 32;
    return rows * rows * sizeof(double);
}

bool var_cols64_fits(int n, int np, long ldx)
{
    const bool on = gpxh::switches().var_cols64 != 0;  // GPX_VAR_COLS64=0: the general path (tests compare the routes)
    return on && n > 0 && n <= VARCOLS64_DEFAULT_N && ldx % 2 == 0 && np % 32 == 0;
}

void launch_var_cols64(const CovHost &h, int n, int np, const double *X, long ldx, const double *px, const double *py,
                       const double *pz, const double *dinv, long nq, const double *qx, const double *qy, const double *qz,
                       double *v, double *xp_ws, hipStream_t st, const double *alpha, double *f)
{
    // The kernel streams 16 x 16 fragments of X, and a lane's MFMA row is its lane number mod 16: read from X itself, every
    // quarter of a wave touches sixteen cache lines for sixteen bytes each, and the L1's tag rate -- not the MFMAs -- sets the
    // time (measured: 3.4 ms of 4.0 at N = 277 with the MFMAs taken out).  So X is first copied into fragment order (a wave's
    // request = two contiguous KB); for a model of <= 1024 points that costs microseconds and is redone per call: no state.
    const int rows = (n + 31) / 32 * 32;
    hipLaunchKernelGGL(pack64_kernel, dim3(rows / 16, rows / 16), dim3(128), 0, st, X, ldx, xp_ws, rows / 16);
    VarCols64Dev g;
    g.Xp = xp_ws, g.fp = rows / 16;
    g.n = n, g.nfrag = (n + 15) / 16;
    g.px = px, g.py = py, g.pz = pz, g.dinv = dinv;
    g.alpha = alpha, g.f = alpha ? f : nullptr;
    g.qx = qx, g.qy = qy, g.qz = qz, g.v = v, g.nq = nq;
    g.k0 = h.k0;
    g.cov = lower_cov<double>(h);
    (void)np;
    const bool two = gpxh::switches().var_cols64 != 1;  // GPX_VAR_COLS64=1: the one-wave-per-SIMD form of round 5 (the tested twin)
    const long per_wg = two ? 16L * VC64Two::CF * VC64Two::WAVES : 16L * VC64One::CF * VC64One::WAVES;
    const long nblk = (nq + per_wg - 1) / per_wg;
    int devid = 0, ncu = 0;
    (void)hipGetDevice(&devid);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || ncu <= 0)
        ncu = 256;
    const unsigned nwg = (unsigned)std::min<long>(nblk, (long)ncu);
    if (two) {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_cols64_kernel<KID, VC64Two>), dim3(nwg), dim3(64 * VC64Two::WAVES), 0, st, g));
    } else {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_cols64_kernel<KID, VC64One>), dim3(nwg), dim3(64 * VC64One::WAVES), 0, st, g));
    }
}

}  // namespace gpx
