// gpx_vargemm.hip -- the long-k products as ONE WAVE PER WORKGROUP (round 3): the fp32 variance contraction
// (GPX_VAR_TILE = 6, var_w1_kernel), its fp64 form (var_w1_f64_kernel) and the large fp64 [k][n] products of the
// inverse-factor assembly (w1_f64_nn_kernel).  The variance contraction
//
//   partial[mt][n] = sum_{m in row tile mt} w[m][n]^2 / D_m ,   w = X[m][:] . K'[n][:]  (+ the low-rank fit, fp64)
//
// replaces the LDS-staged 4-wave tiles of gpx_gemm.hip on this one product (the reference's
// `cholesker.solve(Kqp^T)` + `Kqp * V` + `diagonal()`, gp_regressor.hpp:316-319, restated as a column sum of squares of
// L^-1 K'^T).  Why a different structure: the 4-wave tiles hold the MFMA pipe 88-91 % busy whatever their shape
// (DESIGN.md section 4); what they lose is the per-k-tile barrier that couples four SIMDs each shared by three
// workgroups.  Here a workgroup is a single wave that owns one SIMD and its whole 512-entry register file:
//   * 128 x 128 tile per wave: 8 x 8 accumulator fragments of v_mfma_f32_16x16x4_f32 = 256 AGPRs;
//   * NO LDS and NO barrier: a lane's MFMA operands are 16 contiguous bytes of one row of X or K' (k = 4 g .. 4 g + 3
//     of a 16-deep chunk for lane group g -- the same k-to-lane assignment on both operands, so any assignment is
//     valid), fetched straight from global memory into the fragment registers, one chunk ahead (2 x 64 VGPRs);
//   * buffer loads (one descriptor per operand, the row-block step in the scalar offset, one shared per-lane offset):
//     no per-load address arithmetic.  With flat loads the 64-bit VALU adds in front of each load cost ~15 cycles of
//     MFMA issue apiece (8441 shader cycles per 256-MFMA chunk against 8226 here and 8199 with no loads at all --
//     in-kernel s_memtime stamps, scripts/w1_gemm.hip, profiles/r03_w1_gemm.txt);
//   * the triangle of X is exploited per 128-row tile (k < m0 + 128) and, inside the tile's diagonal block, per 16-row
//     fragment (round 4: W1_COMPUTE_DIAG; +0.6 % at N = 16384, +3.8 % at N = 1536);
//   * the MFMAs are inline asm with the accumulators tied in place ("+a"): left to the register allocator the loop
//     carried ~1100 v_accvgpr moves.  Loads stay ordinary builtins, so hipcc still counts vmcnt for them.
// Traffic per flop is that of the 128 x 128 LDS tile (each wave reads a 128-row slice of both operands once per
// chunk); with one wave per SIMD nothing overlaps the epilogue, so the fp64 add-back of the fit runs on the fp64
// MATRIX pipe: a 128 x 128 x 16 product of the row vectors with the coefficient vectors (256 v_mfma_f64_16x16x4_f64,
// two chunks' worth of time) instead of 3584 fp64 FMAs per lane.
// Measured faster than the LDS tiles at every model size (scripts/var_tile_sweep.py), although for a few hundred rows
// the epilogue and the first loads are 20-30 % of a tile; it does NOT pay for read-modify-write products with K = 256
// (the trailing updates of the LDL^T: DESIGN.md section 4).
// tests/test_codeobj.py disassembles the main loops of this file: nothing but MFMAs, buffer loads, waits and scalar
// arithmetic may appear in them (asm MFMAs are invisible to hipcc's hazard recogniser).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "gpx_internal.hpp"

namespace gpx {

// the one-wave tiles leave the zero fragments of their diagonal block out (round 4; bit-identical to multiplying them, which
// the deleted GPX_VAR_DIAG_SKIP=0 selected: profiles/r04_diag_skip.txt)
static int var_diag_skip() { return 1; }

typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d4v __attribute__((ext_vector_type(4)));

struct VarW1Dev {
    const float *X;
    long ldx;
    const float *Kq;
    long ldk;
    const float *dinv;       // plain epilogue: 1/D in fp32
    float *partial;          // plain epilogue
    double *partial64;       // fp64 epilogue
    long ldp;
    const double *rowcorr, *colcoef, *dinv64;
    long ldrc, ldcc;
    int paired;  // 1: gridDim.y = row tiles / 2 and a workgroup does tile MT - 1 - y, then tile y walked in descending k
    int mt_base;  // first row tile of this launch (0 unless the last row tile runs on its own)
    int k_limit;  // columns of K' that can be non-zero, rounded up to 32
    int diag_skip;  // leave the zero fragments of the diagonal block out (GPX_VAR_DIAG_SKIP=0: multiply them)
};

// NI = row fragments that hold data: 8 for every full tile; 2, 4 or 6 for the LAST row tile of a model whose row count
// leaves most of that tile to the identity padding (N = 266 pads to 384 rows: the last tile holds 10 rows) -- rows of the
// padding give w = 0 exactly (X is the identity there and K', the row vectors of the fit too are zero), so their MFMAs,
// loads and epilogue blocks are simply left out.  launch_var_w1 runs such a tile as a launch of its own.
template <bool CORR, int NI>
__global__ __attribute__((aligned(256))) __launch_bounds__(64, 1) void var_w1_kernel(VarW1Dev g)
{
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, n0 = nt * 128;
    const int r16 = lane & 15, lg = lane >> 4;
    // Which row tile(s).  Plain launch: one tile per workgroup, heavy row tiles first (a tile of row block mt has mt + 1
    // units of k).  Paired launch (launch_var_w1 picks it when the pairs fill the chip's SIMDs in whole rounds): tile
    // MT - 1 - y and then tile y, so that EVERY workgroup has MT + 1 units and the workgroups of a round run in step --
    // the heavy tiles at k = t, the light ones walked in DESCENDING k at k = T - 1 - t whatever their length -- and the
    // row slices of X and K' they share are fetched from HBM once per round instead of once per workgroup.
    const int nphase = g.paired ? 2 : 1;
    const int MT = g.paired ? 2 * (int)gridDim.y : (int)gridDim.y;
#pragma nounroll
    for (int ph = 0; ph < nphase; ++ph) {
    const int mt = g.mt_base + ((ph == 0) ? MT - 1 - (int)blockIdx.y : (int)blockIdx.y);
    const int m0 = mt * 128;
    // 16-deep k chunks: X is lower triangular (k < m0 + 128) and K' is zero from the first padding column on (k_limit, a
    // multiple of 32); an even number of chunks
    const int nch = min(m0 + 128, g.k_limit) / 16;
    const int cfirst = (ph == 0) ? 0 : nch - 1, cdir = (ph == 0) ? 1 : -1;  // chunk j of the walk is cfirst + cdir * j
    char *abase = const_cast<char *>(reinterpret_cast<const char *>(g.X + (size_t)m0 * g.ldx));
    char *bbase = const_cast<char *>(reinterpret_cast<const char *>(g.Kq + (size_t)n0 * g.ldk));
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(abase, 0, (int)(128 * g.ldx * 4), 0x00020000);
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(bbase, 0, (int)(128 * g.ldk * 4), 0x00020000);
    const unsigned aoff = (unsigned)(r16 * g.ldx * 4 + lg * 16);
    const unsigned boff = (unsigned)(r16 * g.ldk * 4 + lg * 16);
    const int astep = (int)(16 * g.ldx * 4), bstep = (int)(16 * g.ldk * 4);

    // fp64 epilogue: this tile's slices of the 14 row vectors (+ a zero row), the 14 coefficient vectors and 1/D go to
    // LDS now, so that the epilogue -- which nothing overlaps with one wave per SIMD -- reads them at LDS latency
    __shared__ double epi_rowc[CORR ? (VAR_NCORR + 1) * 128 : 1];  // row VAR_NCORR: zeros (steps past the last vector)
    __shared__ double epi_colc[CORR ? VAR_NCORR * 128 : 1];
    __shared__ double epi_roww[CORR ? 128 : 1];
    __shared__ float4 epi_acc[CORR ? 8 * 64 : 1];                  // one row block of accumulator fragments
    if constexpr (CORR) {
#pragma unroll
        for (int c = 0; c < VAR_NCORR; ++c) {
            const double2 rv = *reinterpret_cast<const double2 *>(g.rowcorr + (size_t)c * g.ldrc + m0 + 2 * lane);
            const double2 cv = *reinterpret_cast<const double2 *>(g.colcoef + (size_t)c * g.ldcc + n0 + 2 * lane);
            *reinterpret_cast<double2 *>(epi_rowc + c * 128 + 2 * lane) = rv;
            *reinterpret_cast<double2 *>(epi_colc + c * 128 + 2 * lane) = cv;
        }
        *reinterpret_cast<double2 *>(epi_rowc + VAR_NCORR * 128 + 2 * lane) = double2{0.0, 0.0};
        *reinterpret_cast<double2 *>(epi_roww + 2 * lane) = *reinterpret_cast<const double2 *>(g.dinv64 + m0 + 2 * lane);
    }

    f4v acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

    // one piece = one 16-byte load per lane of the next chunk: pieces 0 .. NI - 1 are the A fragments, NI .. NI + 7 the B
    // fragments (the builtin's result is bit-cast as a whole: indexing its elements mis-compiles to a one-dword load on ROCm 7.2)
    float4 a0[NI], b0[8], a1[NI], b1[8];
#define W1_PIECE(A_, B_, KB_, P_)                                                                                      \
    {                                                                                                                  \
        if ((P_) < NI)                                                                                                 \
            A_[(P_) < NI ? (P_) : 0] = __builtin_bit_cast(                                                             \
                float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)(aoff + (KB_)), (P_) * astep, 0));           \
        else                                                                                                           \
            B_[((P_) - NI) & 7] = __builtin_bit_cast(                                                                  \
                float4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)(boff + (KB_)), (((P_) - NI) & 7) * bstep, 0)); \
    }
#define W1_ROW(A_, B_, S_, I_)                       \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) \
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[I_][j_]) : "v"(A_[I_].S_), "v"(B_[j_].S_));
    // 256 MFMAs on (A_, B_); the 16 pieces of (AN_, BN_) go out one per 8 MFMAs over the first half, so the last one has
    // 128 MFMAs (~4000 cycles) to land
    // (partial tiles, NI < 8: the NI + 8 pieces at the top of the chunk -- where the loads sit was measured not to matter)
#define W1_COMPUTE_LD(A_, B_, AN_, BN_, KB_)                                                                       \
    {                                                                                                              \
        if constexpr (NI == 8) {                                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_PIECE(AN_, BN_, KB_, i_) W1_ROW(A_, B_, x, i_) }     \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_PIECE(AN_, BN_, KB_, 8 + i_) W1_ROW(A_, B_, y, i_) } \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, z, i_) }                             \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, w, i_) }                             \
        } else {                                                                                                   \
            _Pragma("unroll") for (int p_ = 0; p_ < NI + 8; ++p_) W1_PIECE(AN_, BN_, KB_, p_)                      \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) { W1_ROW(A_, B_, x, i_) }                            \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) { W1_ROW(A_, B_, y, i_) }                            \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) { W1_ROW(A_, B_, z, i_) }                            \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) { W1_ROW(A_, B_, w, i_) }                            \
        }                                                                                                          \
    }
    // A chunk of the tile's DIAGONAL 128-block (k = m0 + 16 cd .. + 15) meets zeros of the lower-triangular X in every row
    // fragment i < cd: those MFMAs are left out (36 of the 64 fragment-chunks of the block remain; 0.7 % of a tile at
    // N = 16384, 5 % at N = 2048).  cd is wave-uniform, so a row of 8 MFMAs sits behind one scalar branch; the loads of the
    // next chunk all go out at the top.  Full tiles with the fp64 epilogue only.
#define W1_COMPUTE_DIAG(A_, B_, AN_, BN_, KB_, CD_)                                                                \
    {                                                                                                              \
        _Pragma("unroll") for (int p_ = 0; p_ < 16; ++p_) W1_PIECE(AN_, BN_, KB_, p_)                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { if (i_ >= (CD_)) { W1_ROW(A_, B_, x, i_) } }            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { if (i_ >= (CD_)) { W1_ROW(A_, B_, y, i_) } }            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { if (i_ >= (CD_)) { W1_ROW(A_, B_, z, i_) } }            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { if (i_ >= (CD_)) { W1_ROW(A_, B_, w, i_) } }            \
    }
    {
        const unsigned kb0 = (unsigned)cfirst * 64u;
#pragma unroll
        for (int p = 0; p < NI + 8; ++p)
            W1_PIECE(a0, b0, kb0, p)
    }
    // walk positions 0 .. nch - 1 stand for chunk cfirst + cdir * position; the nd chunks of the diagonal block are the last
    // positions of an ascending walk and the first of a descending one (nd is even: m0 and k_limit are multiples of 32)
    constexpr bool DIAG = NI == 8 && CORR;  // (the plain-epilogue variant has no registers to spare: it multiplies the zeros)
    const int nd = (DIAG && g.diag_skip) ? max(0, nch - m0 / 16) : 0, ng = nch - nd;
    int pos = 0;
#define W1_KB(POS_) ((unsigned)(cfirst + cdir * min((POS_), nch - 1)) * 64u)
    if constexpr (DIAG) {
        if (ph != 0) {
#pragma nounroll
            for (int d = 0; d < nd; d += 2, pos += 2) {
                const int cd = __builtin_amdgcn_readfirstlane(nd - 1 - d);
                W1_COMPUTE_DIAG(a0, b0, a1, b1, W1_KB(pos + 1), cd);
                W1_COMPUTE_DIAG(a1, b1, a0, b0, W1_KB(pos + 2), cd - 1);
            }
        }
    }
    asm volatile(".p2align 6");
    for (int c = 0; c < ng; c += 2, pos += 2) {
        // (the last trip re-loads its own second chunk: nothing in the loop is conditional)
        const unsigned kb1 = W1_KB(pos + 1);
        const unsigned kb2 = W1_KB(pos + 2);
        W1_COMPUTE_LD(a0, b0, a1, b1, kb1);
        W1_COMPUTE_LD(a1, b1, a0, b0, kb2);
    }
    if constexpr (DIAG) {
        if (ph == 0) {
#pragma nounroll
            for (int d = 0; d < nd; d += 2, pos += 2) {
                const int cd = __builtin_amdgcn_readfirstlane(d);
                W1_COMPUTE_DIAG(a0, b0, a1, b1, W1_KB(pos + 1), cd);
                W1_COMPUTE_DIAG(a1, b1, a0, b0, W1_KB(pos + 2), cd + 1);
            }
        }
    }
#undef W1_KB
#undef W1_PIECE
#undef W1_ROW
#undef W1_COMPUTE_LD
#undef W1_COMPUTE_DIAG
    // the asm MFMAs are opaque to hipcc's hazard recogniser: let the last ones retire before the accumulators are read
    // (tied to the last row of fragments, so that no read of them can be scheduled above the wait states; every other
    // fragment's last MFMA is at least 8 MFMAs = 256 cycles older)
    asm volatile("s_nop 15\n s_nop 15"
                 : "+a"(acc[NI - 1][0]), "+a"(acc[NI - 1][1]), "+a"(acc[NI - 1][2]), "+a"(acc[NI - 1][3]),
                   "+a"(acc[NI - 1][4]), "+a"(acc[NI - 1][5]), "+a"(acc[NI - 1][6]), "+a"(acc[NI - 1][7])
                 :
                 : "memory");

    // acc[i][j][r] is row 16 i + 4 lg + r, column 16 j + r16 of the tile
    if constexpr (!CORR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float4 w = *reinterpret_cast<const float4 *>(g.dinv + m0 + 16 * i + 4 * lg);
                s = fmaf(acc[i][j][0] * acc[i][j][0], w.x, s);
                s = fmaf(acc[i][j][1] * acc[i][j][1], w.y, s);
                s = fmaf(acc[i][j][2] * acc[i][j][2], w.z, s);
                s = fmaf(acc[i][j][3] * acc[i][j][3], w.w, s);
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lg == 0)
                g.partial[(size_t)mt * g.ldp + n0 + 16 * j + r16] = s;
        }
    } else {
        // Low-rank fit added back in fp64 (gpx_eval.hip, "centred kernel operand"): w = acc + sum_c rowcorr[c][m] colcoef[c][n],
        // then w^2 / D and the column sums, all fp64.  The rank-14 sum is a 128 x 128 x 16 fp64 product on the matrix
        // pipe.  v_mfma_f64_16x16x4_f64 returns rows lg, lg + 4, lg + 8, lg + 12 of a fragment to a lane where the fp32
        // form returned rows 4 lg .. 4 lg + 3, so the A operand is fed with its rows permuted: operand row rho carries
        // tile row 4 (rho & 3) + (rho >> 2), and result register r of a lane is then row 4 lg + r -- acc's layout.
        // Step s of a fragment uses vector c = 4 s + lg; c >= VAR_NCORR reads the zero row (and any finite coefficient).
        //
        // Shape of the code.  A ROLLED loop over the eight row blocks whose body hipcc compiles once, at the register
        // pressure of one block; the accumulators reach it through LDS, written from their AGPRs by the asm of the switch
        // (8 KiB per block).  Any C++ read of acc[][] -- or the MFMA builtin, whose result wants AGPRs -- makes hipcc copy
        // all 256 accumulators to VGPRs at the top of the epilogue and spill most of them.  The fp64 MFMAs are inline asm
        // with VGPR results for the same reason; asm is opaque to the hazard recogniser, hence the explicit wait states
        // (tied to the results, so that no consumer is scheduled above them) and eight independent chains (dependent
        // MFMAs 8 x 32 cycles apart).  One wave is the whole workgroup and a wave's LDS operations execute in order:
        // no barrier anywhere, only lgkmcnt waits.
        const unsigned epi_lane = (unsigned)(size_t)(epi_acc + lane);  // low half of the flat address = the LDS offset
        const int prow = 4 * (r16 & 3) + (r16 >> 2);
        int crow[4], ccol[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            crow[s] = min(4 * s + lg, VAR_NCORR) * 128 + prow;
            ccol[s] = min(4 * s + lg, VAR_NCORR - 1) * 128 + r16;
        }
        double sj[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            sj[j] = 0.0;
#define W1_DUMP(I_)                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                                                   \
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(epi_lane), "a"(acc[I_][j_]), "n"(1024 * j_) : "memory");
#pragma nounroll
        for (int i = 0; i < NI; ++i) {  // (row blocks past NI hold zeros and rows of the padding: nothing to add)
            switch (i) {
            case 0: W1_DUMP(0) break;
            case 1: W1_DUMP(1) break;
            case 2: W1_DUMP(2) break;
            case 3: W1_DUMP(3) break;
            case 4: W1_DUMP(4) break;
            case 5: W1_DUMP(5) break;
            case 6: W1_DUMP(6) break;
            default: W1_DUMP(7) break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double ra[4], rw[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                ra[s] = epi_rowc[crow[s] + 16 * i];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rw[r] = epi_roww[16 * i + 4 * lg + r];
            d4v d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double cb = epi_colc[ccol[0] + 16 * j];
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(d[j]) : "v"(ra[0]), "v"(cb));
            }
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double cb = epi_colc[ccol[s] + 16 * j];
                    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(ra[s]), "v"(cb));
                }
            asm volatile("s_nop 15\n s_nop 15"
                         : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 av = epi_acc[64 * j + lane];
                double w = (double)av.x + d[j][0];
                sj[j] = fma(w * w, rw[0], sj[j]);
                w = (double)av.y + d[j][1];
                sj[j] = fma(w * w, rw[1], sj[j]);
                w = (double)av.z + d[j][2];
                sj[j] = fma(w * w, rw[2], sj[j]);
                w = (double)av.w + d[j][3];
                sj[j] = fma(w * w, rw[3], sj[j]);
            }
        }
#undef W1_DUMP
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double t = sj[j];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (lg == 0)
                g.partial64[(size_t)mt * g.ldp + n0 + 16 * j + r16] = t;
        }
    }
    }  // phase (row tile of the pair)
}

// ---- the same structure in fp64 (GPX_PREC_F64 models; the derivative-observation GP) ------------------------------------
// v_mfma_f64_16x16x4_f64 accumulates a 16 x 16 fragment in 8 registers, so 256 AGPRs hold 8 x 4 fragments: a 128 x 64 tile
// per wave.  A lane's operand for an 8-deep chunk is again 16 contiguous bytes of one row -- k = 2 g, 2 g + 1 for lane
// group g -- and the two MFMA steps of the chunk take its two halves; 64 MFMAs of 64 cycles and 12 loads per chunk.
// Plain epilogue only (fp64 models carry no fit).  Result rows of a lane: g, g + 4, g + 8, g + 12 of each fragment.
struct VarW1F64Dev {
    const double *X;
    long ldx;
    const double *Kq;
    long ldk;
    const double *dinv;
    double *partial;
    long ldp;
    int paired;  // as in VarW1Dev
    int diag_skip;
};

__global__ __attribute__((aligned(256))) __launch_bounds__(64, 1) void var_w1_f64_kernel(VarW1F64Dev g)
{
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, n0 = nt * 64;
    const int r16 = lane & 15, lg = lane >> 4;
    // plain launch: heavy row tiles first; paired launch: tile MT - 1 - y, then tile y in descending k (see var_w1_kernel)
    const int nphase = g.paired ? 2 : 1;
    const int MT = g.paired ? 2 * (int)gridDim.y : (int)gridDim.y;
#pragma nounroll
    for (int ph = 0; ph < nphase; ++ph) {
    const int mt = (ph == 0) ? MT - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const int m0 = mt * 128;
    const int nch = (m0 + 128) / 8;  // 8-deep k chunks; a multiple of 16
    const int cfirst = (ph == 0) ? 0 : nch - 1, cdir = (ph == 0) ? 1 : -1;
    char *abase = const_cast<char *>(reinterpret_cast<const char *>(g.X + (size_t)m0 * g.ldx));
    char *bbase = const_cast<char *>(reinterpret_cast<const char *>(g.Kq + (size_t)n0 * g.ldk));
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(abase, 0, (int)(128 * g.ldx * 8), 0x00020000);
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(bbase, 0, (int)(64 * g.ldk * 8), 0x00020000);
    const unsigned aoff = (unsigned)(r16 * g.ldx * 8 + lg * 16);
    const unsigned boff = (unsigned)(r16 * g.ldk * 8 + lg * 16);
    const int astep = (int)(16 * g.ldx * 8), bstep = (int)(16 * g.ldk * 8);

    d4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = d4v{0.0, 0.0, 0.0, 0.0};

    double2 a0[8], b0[4], a1[8], b1[4];
    // pieces 0-7: the A fragments, 8-11: the B fragments
#define W1D_PIECE(A_, B_, KB_, P_)                                                                                  \
    {                                                                                                               \
        if ((P_) < 8)                                                                                               \
            A_[(P_) & 7] = __builtin_bit_cast(                                                                      \
                double2, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)(aoff + (KB_)), ((P_) & 7) * astep, 0)); \
        else                                                                                                        \
            B_[(P_) & 3] = __builtin_bit_cast(                                                                      \
                double2, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)(boff + (KB_)), ((P_) & 3) * bstep, 0)); \
    }
#define W1D_ROW(A_, B_, S_, I_)                      \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) \
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[I_][j_]) : "v"(A_[I_].S_), "v"(B_[j_].S_));
    // 64 MFMAs on (A_, B_); the 12 pieces of the next chunk go out one per 4 MFMAs (first 48 MFMAs = 3000 cycles)
#define W1D_COMPUTE_LD(A_, B_, AN_, BN_, KB_)                                                                        \
    {                                                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1D_PIECE(AN_, BN_, KB_, i_) W1D_ROW(A_, B_, x, i_) }     \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { W1D_PIECE(AN_, BN_, KB_, 8 + i_) W1D_ROW(A_, B_, y, i_) } \
        _Pragma("unroll") for (int i_ = 4; i_ < 8; ++i_) { W1D_ROW(A_, B_, y, i_) }                                  \
    }
    // the diagonal 128-block (the last 16 chunks of an ascending walk): chunks 2 cd and 2 cd + 1 meet zeros of X in the row
    // fragments i < cd -- left out, as in var_w1_kernel.  Here with cd a compile-time constant and the eight pairs written
    // out: behind per-row branches hipcc renamed accumulators at the merges (v_accvgpr_read / _write next to the asm MFMAs,
    // which its hazard recogniser cannot see: results read while still in flight, 1e-6 errors at N = 4096).
#define W1D_COMPUTE_DIAG(A_, B_, AN_, BN_, KB_, CD_)                                           \
    {                                                                                          \
        _Pragma("unroll") for (int p_ = 0; p_ < 12; ++p_) W1D_PIECE(AN_, BN_, KB_, p_)         \
        _Pragma("unroll") for (int i_ = (CD_); i_ < 8; ++i_) { W1D_ROW(A_, B_, x, i_) }        \
        _Pragma("unroll") for (int i_ = (CD_); i_ < 8; ++i_) { W1D_ROW(A_, B_, y, i_) }        \
    }
#define W1D_DIAG_PAIR(CD_)                                          \
    {                                                               \
        W1D_COMPUTE_DIAG(a0, b0, a1, b1, W1D_KB(pos + 1), CD_);     \
        W1D_COMPUTE_DIAG(a1, b1, a0, b0, W1D_KB(pos + 2), CD_);     \
        pos += 2;                                                   \
    }
    {
        const unsigned kb0 = (unsigned)cfirst * 64u;
#pragma unroll
        for (int p = 0; p < 12; ++p)
            W1D_PIECE(a0, b0, kb0, p)
    }
    const int nd = g.diag_skip ? 16 : 0, ng = nch - nd;
    int pos = 0;
#define W1D_KB(POS_) ((unsigned)(cfirst + cdir * min((POS_), nch - 1)) * 64u)
    if (ph != 0 && nd) {
        W1D_DIAG_PAIR(7) W1D_DIAG_PAIR(6) W1D_DIAG_PAIR(5) W1D_DIAG_PAIR(4)
        W1D_DIAG_PAIR(3) W1D_DIAG_PAIR(2) W1D_DIAG_PAIR(1) W1D_DIAG_PAIR(0)
    }
    asm volatile(".p2align 6");
    for (int c = 0; c < ng; c += 2, pos += 2) {
        const unsigned kb1 = W1D_KB(pos + 1);
        const unsigned kb2 = W1D_KB(pos + 2);
        W1D_COMPUTE_LD(a0, b0, a1, b1, kb1);
        W1D_COMPUTE_LD(a1, b1, a0, b0, kb2);
    }
    if (ph == 0 && nd) {
        W1D_DIAG_PAIR(0) W1D_DIAG_PAIR(1) W1D_DIAG_PAIR(2) W1D_DIAG_PAIR(3)
        W1D_DIAG_PAIR(4) W1D_DIAG_PAIR(5) W1D_DIAG_PAIR(6) W1D_DIAG_PAIR(7)
    }
#undef W1D_DIAG_PAIR
#undef W1D_KB
#undef W1D_PIECE
#undef W1D_ROW
#undef W1D_COMPUTE_LD
#undef W1D_COMPUTE_DIAG
    // (as in the fp32 kernel: wait states tied to the last MFMAs' fragments; the others are >= 4 x 64 cycles older)
    asm volatile("s_nop 15\n s_nop 15"
                 : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3])
                 :
                 : "memory");

    // acc[i][j][r] is row 16 i + lg + 4 r, column 16 j + r16 of the tile.  A ROLLED loop over the row blocks with the block's
    // fragments picked by a switch: unrolled, hipcc copies all 256 accumulator registers to VGPRs at once and spills.
    double sj[4] = {0.0, 0.0, 0.0, 0.0};
#define W1D_PICK(I_) t[0] = acc[I_][0], t[1] = acc[I_][1], t[2] = acc[I_][2], t[3] = acc[I_][3]
#pragma nounroll
    for (int i = 0; i < 8; ++i) {
        d4v t[4];
        switch (i) {
        case 0: W1D_PICK(0); break;
        case 1: W1D_PICK(1); break;
        case 2: W1D_PICK(2); break;
        case 3: W1D_PICK(3); break;
        case 4: W1D_PICK(4); break;
        case 5: W1D_PICK(5); break;
        case 6: W1D_PICK(6); break;
        default: W1D_PICK(7); break;
        }
        double w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            w[r] = g.dinv[m0 + 16 * i + lg + 4 * r];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sj[j] = fma(t[j][r] * t[j][r], w[r], sj[j]);
    }
#undef W1D_PICK
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double s = sj[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lg == 0)
            g.partial[(size_t)mt * g.ldp + n0 + 16 * j + r16] = s;
    }
    }  // phase (row tile of the pair)
}

bool var_w1_f64_fits(const GemmArgs &a)
{
    return a.epi == EPI_COLSQ && !a.nn && a.a_lower && !a.b_lower && !a.lower_only && a.batch == 1 && a.M_last < 0 &&
           a.M > 0 && a.N > 0 && a.M % 128 == 0 && a.N % 64 == 0 && a.K >= a.M && a.lda % 2 == 0 && a.ldb % 2 == 0 &&
           a.lda >= a.M && a.ldb >= a.M && 128 * a.lda * 8 < (1L << 31) && 64 * a.ldb * 8 < (1L << 31) && !a.colcoef;
}

// Paired launch (see var_w1_kernel) when the row tiles pair up and the pairs fill the SIMDs in whole rounds -- equal-length
// workgroups would otherwise leave a partly filled last round that nothing balances.
static bool var_w1_paired(int MT, int NT)
{
    if (MT < 2 || MT % 2)
        return false;
    static std::atomic<int> cu_count[MAX_DEVICES];  // per device, 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
        return false;
    int cus = cu_count[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return false;
        cu_count[dev].store(cus, std::memory_order_relaxed);
    }
    const long slots = 4L * cus;  // one wave per SIMD
    return slots > 0 && ((long)(MT / 2) * NT) % slots == 0;
}

void launch_var_w1_f64(const GemmArgs &a, hipStream_t st)
{
    VarW1F64Dev g;
    g.X = (const double *)a.A, g.ldx = a.lda;
    g.Kq = (const double *)a.B, g.ldk = a.ldb;
    g.dinv = (const double *)a.rowweight;
    g.partial = (double *)a.partial, g.ldp = a.ldp;
    const int MT = a.M / 128, NT = a.N / 64;
    const bool paired = !a.no_pair && var_w1_paired(MT, NT);
    g.paired = paired ? 1 : 0;
    g.diag_skip = var_diag_skip();
    hipLaunchKernelGGL(var_w1_f64_kernel, dim3(NT, paired ? MT / 2 : MT), dim3(64), 0, st, g);
}

// ---- C = alpha A B with B in [k][n] form, fp64: the large products of the inverse-factor assembly -----------------------
// (T = L21 X11 with X11 lower triangular, X21 = -X22 T with X22 lower triangular; gpx_build.hip trtri_levels / _combine.)
// Same one-wave structure, 128 x 64 tile.  A's fragments are fetched as in the kernels above (16 bytes = k = 2 g, 2 g + 1 of
// one row).  B's rows run along n, so a lane's 16 bytes are TWO COLUMNS of one k: the lane of column index c fetches columns
// n0 + 32 jj + 2 c and + 1 of row k = 2 g + s and hands them to fragments 2 jj and 2 jj + 1 -- fragment j = 2 jj + p
// therefore holds the columns n0 + 32 jj + 2 c + p, a permutation the store undoes for free (a lane's results for fragments
// 2 jj, 2 jj + 1 are two adjacent columns of C: one 16-byte store).  B is walked by moving the descriptor's base (a row
// block of k is ld * 64 bytes: a 32-bit offset from the matrix corner would not reach the far rows of a large model).
struct W1NNDev {
    const double *A, *B;
    double *C;
    long lda, ldb, ldc;
    int M, K;
    long sA, sB, sC;
    int batch, M_last, k_eq_m;
    double alpha;
    int a_lower, b_lower;
};

__global__ __attribute__((aligned(256))) __launch_bounds__(64, 1) void w1_f64_nn_kernel(W1NNDev g)
{
    const int lane = threadIdx.x;
    const int r16 = lane & 15, lg = lane >> 4;
    // heavy tiles first: k < m0 + 128 for a lower-triangular A (large mt first), k >= n0 for a lower-triangular B (small nt
    // first, the column index slowest so that neighbours in launch order share the k range)
    int mt, nt;
    if (g.b_lower)
        mt = blockIdx.x, nt = blockIdx.y;
    else
        nt = blockIdx.x, mt = g.a_lower ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const int z = blockIdx.z;
    int Mz = g.M, Kz = g.K;
    if (g.M_last >= 0 && z == g.batch - 1) {
        Mz = g.M_last;
        if (g.k_eq_m)
            Kz = g.M_last;
    }
    const int m0 = mt * 128, n0 = nt * 64;
    if (m0 >= Mz)
        return;
    const int klo = g.b_lower ? n0 : 0;
    const int khi = g.a_lower ? min(Kz, m0 + 128) : Kz;
    const int nch = (khi - klo) / 8;  // 8-deep chunks; klo is a multiple of 64 and khi of 128: a multiple of 8
    double *C = g.C + (size_t)z * g.sC;  // (an empty k range -- not a shape of the assembly -- stores zeros: the loop does not run)
    char *abase = const_cast<char *>(reinterpret_cast<const char *>(g.A + (size_t)z * g.sA + (size_t)m0 * g.lda + klo));
    const char *bcorner = reinterpret_cast<const char *>(g.B + (size_t)z * g.sB + (size_t)klo * g.ldb + n0);
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(abase, 0, (int)(128 * g.lda * 8), 0x00020000);
    const unsigned aoff = (unsigned)(r16 * g.lda * 8 + lg * 16);
    const int astep = (int)(16 * g.lda * 8);
    const size_t bchunk = (size_t)8 * g.ldb * 8;          // bytes from one chunk of B to the next
    const int brange = (int)(8 * g.ldb * 8);              // one chunk of rows
    const unsigned boff0 = (unsigned)((2 * lg) * g.ldb * 8 + r16 * 16), boff1 = boff0 + (unsigned)(g.ldb * 8);

    d4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = d4v{0.0, 0.0, 0.0, 0.0};

    // operands of a chunk: a[i] = {A[row][k], A[row][k + 1]}; b[2 s + jj] = {B[k + s][col pair jj]} for step s
    double2 a0[8], b0[4], a1[8], b1[4];
#define NN_APIECE(A_, KB_, P_) \
    A_[P_] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)(aoff + (KB_)), (P_) * astep, 0));
#define NN_BPIECE(B_, RS_, P_) \
    B_[P_] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(RS_, (int)(((P_) >> 1) ? boff1 : boff0), ((P_) & 1) * 256, 0));
#define NN_MFMA(ACC_, A_, B_) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_));
    // step s = 0 (k = 2 g): A_.x with b[0], b[1]; step 1 (k = 2 g + 1): A_.y with b[2], b[3]
#define NN_ROW0(A_, B_, I_) \
    NN_MFMA(acc[I_][0], A_[I_].x, B_[0].x) NN_MFMA(acc[I_][1], A_[I_].x, B_[0].y) NN_MFMA(acc[I_][2], A_[I_].x, B_[1].x) NN_MFMA(acc[I_][3], A_[I_].x, B_[1].y)
#define NN_ROW1(A_, B_, I_) \
    NN_MFMA(acc[I_][0], A_[I_].y, B_[2].x) NN_MFMA(acc[I_][1], A_[I_].y, B_[2].y) NN_MFMA(acc[I_][2], A_[I_].y, B_[3].x) NN_MFMA(acc[I_][3], A_[I_].y, B_[3].y)
#define NN_COMPUTE_LD(A_, B_, AN_, BN_, KB_, RS_)                                                              \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { NN_APIECE(AN_, KB_, i_) NN_ROW0(A_, B_, i_) }       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { NN_BPIECE(BN_, RS_, i_) NN_ROW1(A_, B_, i_) }       \
        _Pragma("unroll") for (int i_ = 4; i_ < 8; ++i_) { NN_ROW1(A_, B_, i_) }                               \
    }
    {
        const auto brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bcorner), 0, brange, 0x00020000);
#pragma unroll
        for (int p = 0; p < 8; ++p)
            NN_APIECE(a0, 0u, p)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            NN_BPIECE(b0, brs, p)
    }
    asm volatile(".p2align 6");
    for (int c = 0; c < nch; c += 2) {
        const int c1 = c + 1, c2 = min(c + 2, nch - 1);
        const auto brs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bcorner + c1 * bchunk), 0, brange, 0x00020000);
        const auto brs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bcorner + c2 * bchunk), 0, brange, 0x00020000);
        NN_COMPUTE_LD(a0, b0, a1, b1, (unsigned)c1 * 64u, brs1);
        NN_COMPUTE_LD(a1, b1, a0, b0, (unsigned)c2 * 64u, brs2);
    }
#undef NN_APIECE
#undef NN_BPIECE
#undef NN_MFMA
#undef NN_ROW0
#undef NN_ROW1
#undef NN_COMPUTE_LD
    asm volatile("s_nop 15\n s_nop 15"
                 : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3])
                 :
                 : "memory");

    // acc[i][j][r] is row 16 i + lg + 4 r, column 32 (j >> 1) + 2 r16 + (j & 1): rolled over the row blocks (see above)
    const double alpha = g.alpha;
#define NN_PICK(I_) t[0] = acc[I_][0], t[1] = acc[I_][1], t[2] = acc[I_][2], t[3] = acc[I_][3]
#pragma nounroll
    for (int i = 0; i < 8; ++i) {
        d4v t[4];
        switch (i) {
        case 0: NN_PICK(0); break;
        case 1: NN_PICK(1); break;
        case 2: NN_PICK(2); break;
        case 3: NN_PICK(3); break;
        case 4: NN_PICK(4); break;
        case 5: NN_PICK(5); break;
        case 6: NN_PICK(6); break;
        default: NN_PICK(7); break;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *row = C + (size_t)(m0 + 16 * i + lg + 4 * r) * g.ldc + n0 + 2 * r16;
            *reinterpret_cast<double2 *>(row) = double2{alpha * t[0][r], alpha * t[1][r]};
            *reinterpret_cast<double2 *>(row + 32) = double2{alpha * t[2][r], alpha * t[3][r]};
        }
    }
#undef NN_PICK
}

bool w1_f64_nn_fits(const GemmArgs &a)
{
    return a.epi == EPI_STORE && a.nn && a.beta == 0 && !a.lower_only && !(a.a_lower && a.b_lower) && a.batch >= 1 &&
           a.M > 0 && a.N > 0 && a.M % 128 == 0 && a.N % 64 == 0 && a.K % 128 == 0 && (a.M_last < 0 || a.M_last % 128 == 0) &&
           a.K >= W1_NN_MIN_K && a.lda % 2 == 0 && a.ldb % 2 == 0 && a.ldc % 2 == 0 && a.sA % 2 == 0 && a.sB % 2 == 0 &&
           a.sC % 2 == 0 && 128 * a.lda * 8 < (1L << 31) && 8 * a.ldb * 8 < (1L << 31);
}

void launch_w1_f64_nn(const GemmArgs &a, hipStream_t st)
{
    W1NNDev g;
    g.A = (const double *)a.A, g.B = (const double *)a.B, g.C = (double *)a.C;
    g.lda = a.lda, g.ldb = a.ldb, g.ldc = a.ldc;
    g.M = a.M, g.K = a.K;
    g.sA = a.sA, g.sB = a.sB, g.sC = a.sC;
    g.batch = a.batch, g.M_last = a.M_last, g.k_eq_m = a.k_eq_m;
    g.alpha = a.alpha;
    g.a_lower = a.a_lower, g.b_lower = a.b_lower;
    const int mt = a.M / 128, nt = a.N / 64;
    const dim3 grid = a.b_lower ? dim3(mt, nt, a.batch) : dim3(nt, mt, a.batch);
    hipLaunchKernelGGL(w1_f64_nn_kernel, grid, dim3(64), 0, st, g);
}

bool var_w1_fits(const GemmArgs &a)
{
    // 128-row slices of both operands must fit a buffer descriptor's 32-bit range; 16-byte aligned rows
    return a.epi == EPI_COLSQ && !a.nn && a.a_lower && !a.b_lower && !a.lower_only && a.batch == 1 && a.M_last < 0 &&
           a.M > 0 && a.N > 0 && a.M % 128 == 0 && a.N % 128 == 0 && a.K >= a.M && a.lda % 4 == 0 && a.ldb % 4 == 0 &&
           a.lda >= a.M && a.ldb >= a.M && 128 * a.lda * 4 < (1L << 31) && 128 * a.ldb * 4 < (1L << 31);
}

template <int NI>
static void var_w1_launch_ni(const VarW1Dev &g, bool corr, dim3 grid, hipStream_t st)
{
    if (corr)
        hipLaunchKernelGGL((var_w1_kernel<true, NI>), grid, dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL((var_w1_kernel<false, NI>), grid, dim3(64), 0, st, g);
}

void launch_var_w1(const GemmArgs &a, hipStream_t st)
{
    VarW1Dev g;
    g.X = (const float *)a.A, g.ldx = a.lda;
    g.Kq = (const float *)a.B, g.ldk = a.ldb;
    g.dinv = (const float *)a.rowweight;
    g.partial = (float *)a.partial, g.partial64 = (double *)a.partial, g.ldp = a.ldp;
    g.rowcorr = a.rowcorr, g.colcoef = a.colcoef, g.dinv64 = a.rowweight64;
    g.ldrc = a.ldrc, g.ldcc = a.ldcc;
    const bool corr = a.colcoef != nullptr;
    const int MT = a.M / 128, NT = a.N / 128;
    // rows that hold data (the rest of the last tile is the identity padding): columns of K' past them are zero
    const int mv = (a.m_valid > 0 && a.m_valid <= a.M) ? a.m_valid : a.M;
    g.k_limit = std::min(a.M, (mv + 31) / 32 * 32);
    g.diag_skip = var_diag_skip();
    const int r_last = mv - (MT - 1) * 128;  // data rows of the last row tile
    const int ni_last = r_last <= 0 ? 8 : (r_last <= 32 ? 2 : (r_last <= 64 ? 4 : (r_last <= 96 ? 6 : 8)));
    int mt_main = MT;
    if (ni_last < 8) {  // the last row tile (the longest k range) as a launch of its own, first
        g.paired = 0, g.mt_base = MT - 1;
        if (ni_last == 2)
            var_w1_launch_ni<2>(g, corr, dim3(NT, 1), st);
        else if (ni_last == 4)
            var_w1_launch_ni<4>(g, corr, dim3(NT, 1), st);
        else
            var_w1_launch_ni<6>(g, corr, dim3(NT, 1), st);
        mt_main = MT - 1;
    }
    if (mt_main <= 0)
        return;
    const bool paired = !a.no_pair && var_w1_paired(mt_main, NT);
    g.paired = paired ? 1 : 0, g.mt_base = 0;
    var_w1_launch_ni<8>(g, corr, dim3(NT, paired ? mt_main / 2 : mt_main), st);
}

}  // namespace gpx
