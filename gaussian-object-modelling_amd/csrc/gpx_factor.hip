// gpx_factor.hip -- the latency-bound pieces of the blocked LDL^T factorisation and of the
// triangular solves (gfx950).  Replaces Eigen::LDLT<MatrixXd>::compute / ::solve as used by the
// reference (gp_regressor.hpp:161-163); the O(N^3) work is in gpx_gemm.hip.
//
//   diag_ldl   : one 128 x 128 diagonal block by one workgroup of 8 (4) waves: right-looking LDL^T blocked by 32, no
//                pivoting inside the block (the Eigen rule picks pivots from the ORIGINAL diagonal, so the permutation
//                is applied to the points before kbuild), and the unit-lower inverse of its L.  The inverse blocks turn
//                every panel solve and every block substitution into matrix products.  Round 2 (diag_ldlm_kernel):
//                the 32 x 32 sub-block is factorised AND inverted by rank-1 MFMA updates of accumulators that hold it,
//                the rows below by one product with that inverse, the trailing blocks stay in registers, the 128 x 128
//                inverse is assembled left-looking in the shadow of the panel steps.  identity_blocks: the same result
//                for blocks that lie in the padding.
//   tri_solve  : L y = b and L^T x = D^-1 y in one launch each (a workgroup per block row, results handed on
//                through self-validating entries); fwd / bwd step kernels: the same, one launch per block step.
#include "gpx_internal.hpp"
#include "gpx_blk.hpp"

// phase timing of diag_ldlm_kernel for scripts/diag_bench.hip (which defines GPX_STAMP); nothing in the library build
#ifndef GPX_STAMP
#define GPX_STAMP(i)
#endif

namespace gpx {

// threads of the diagonal-block kernel: 8 waves -- one factorises a sub-block, one inverts the previous one, six
// carry the trailing update (one 32 x 32 block each); at most 256 VGPRs per lane (fp64 spills 24 of them).
// The 4-wave instantiation (fp32) is the one that fits on a CU BESIDE a workgroup of the trailing-update GEMM
// (<= 256 VGPRs per SIMD, 68 of the 160 KB of LDS): the 8-wave one needs a whole CU and starves behind a running GEMM.
constexpr int DIAG_THREADS = 512, DIAG_THREADS_NARROW = 256;


// ---- diagonal block on the matrix cores (round 2): diag_ldlm_kernel<float | double> ---------------------------------
// LDL^T of a 128 x 128 block without pivoting + the unit-lower inverse of its L (the round-1 kernel it replaced -- sub-block by
// v_readlane + FMA, inverse on a second wave: fp32 38 us, fp64 55 -- was deleted in round 5).  In fp32 -- fp64: subblock_ldl in
// gpx_blk.hpp --
// the 32 x 32 sub-block is no longer factorised by 1000 v_readlane + FMA pairs in one wave (5.1 us): it sits in the
// accumulator of v_mfma_f32_32x32x2_f32 (column on the lane, rows in the 16 registers) and every elimination step is
// ONE rank-1 MFMA:  M <- M - (u_j / d_j) u_j^T  with u_j = row j of M, which is one accumulator register of one wave
// half -- and that register in that half IS the B operand (k = half), and scaled by -1/d_j the A operand as well (the
// block is symmetric), so no lane movement at all; the other half supplies a zero A.  A second accumulator that starts
// as the identity takes the same A operand against ITS row j:  X <- X - l_j X[j,:]; after 32 steps X = L11^-1.  That
// removes the separate inverse of each sub-block (4.1 us on a second wave, the last one on the critical path) and turns
// the rows below from a forward substitution into one product W = A21 X11^T.  The f32 MFMA is an exact fmaf chain
// (/opt/skills/guides/MI355X_MICROARCH.md), so the arithmetic is that of the VALU version with the roundings of an LU
// sweep over the full symmetric block.  Cost per step (scripts/mfma_lat_probe.hip): 213 cycles -- 64 per MFMA, and a
// VALU read of an MFMA result waits for the whole matrix pipe to drain, so the inverse's MFMA is NOT hidden behind the
// pivot arithmetic (144 cycles without it).
// The trailing 32 x 32 blocks stay in the registers of fixed owner waves for the whole kernel and travel to the next
// panel through LDS (next diagonal sub-block, next panel column), so nothing the kernel computes is read back from
// global memory, and the barriers wait for LDS traffic only: the stores of L, D and X drain in the background (with
// __syncthreads every step waited for its global stores to be acknowledged, ~1.5 us per barrier).
// The inverse of the 128 x 128 L is assembled LEFT-looking, block row by block row, X_i: = -Xd_i (sum_k<i L_ik X_k:),
// by waves that are idle during the panel steps; only the last block row needs Xd_3 and costs one product per block
// after the last sub-block (was: last inverse 4.2 us + three product stages 3.4 us).

// trailing block (bi, bj), 1 <= bj <= bi <= 3, held in slot e of the k-th trailing wave:
//   6 waves: one block each, (1,1) (2,1) (3,1) (2,2) (3,2) (3,3);  2 waves: the first the diagonal blocks, the second the others
template <int NCW>
__device__ __forceinline__ void own_block(int k, int e, int &bi, int &bj)
{
    if (NCW == 6) {
        bi = k < 3 ? k + 1 : (k < 5 ? k - 1 : 3);
        bj = k < 3 ? 1 : (k < 5 ? 2 : 3);
    } else if (k == 0) {
        bi = bj = e + 1;
    } else {
        bi = e == 0 ? 2 : 3;
        bj = e == 2 ? 2 : 1;
    }
}


template <typename T, int DT>
__global__ __launch_bounds__(DT, 2) void diag_ldlm_kernel(T *__restrict__ A, long lda, T *__restrict__ linv,
                                                       T *__restrict__ d, T *__restrict__ dinv, int *__restrict__ info,
                                                       int blk)
{
    constexpr int BLK = NB * PLD;                  // one 32 x 32 LDS block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *Di = reinterpret_cast<T *>(smem_raw);       // [TILE]       1 / D
    T *Ls = Di + TILE;                             // [6][NB][PLD] L blocks (1,0) (2,0) (3,0) | (2,1) (3,1) | (3,2)
    T *Xd = Ls + 6 * BLK;                          // [4][NB][PLD] inverses of the diagonal sub-blocks
    T *Dn = Xd + 4 * BLK;                          // [NB][PLD]    next diagonal sub-block, handed over by the trailing update
    T *Lx = Dn + BLK;                              // [NB][PLD]    L11 of the current sub-block, for its row-wise write-out
    T *P = Lx + BLK;                               // [96][PLD]    panel rows 32 .. 127: A entries, then W = L D
    T *Pv = P - NB * PLD;                          //              (row r at Pv + r * PLD)
    // fp32: 63.8 KB in all -- no more than a workgroup of the GEMM, so that the kernel finds room on a CU wherever one of
    // those does (LDS is allocated contiguously); fp64: 127.7 KB, a CU of its own (as the round-1 kernel: 132 KB).  The assembly of the inverse therefore lives in blocks that are dead by then:
    //   panel rows 32-63 : T0 (A2: L10 Xd0; B2: L20 Xd0 + L21 X10) -> X20 in place (C2)
    //   panel rows 64-95 : X10 (A2)                panel rows 96-127: S31 (A3; W of panel 2 until then)
    //   L10 block        : L21 Xd1 -> X21 in place (B2)
    //   L20 block        : S32 (C2)                L21 block: S30 (A3)
    T *T0 = P, *X10 = P + BLK, *S31 = P + 2 * BLK;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    T *Xg = linv + (size_t)blk * TILE * TILE;
    constexpr int NW = DT / 64;
    constexpr int NCW = NW - 2;                    // waves that carry the trailing update
    constexpr int MAXB = 6 / NCW;
    constexpr int HC = NW == 4 ? 3 : NW - 2;       // a wave without an active trailing block in panel 2
    BlkAcc<T> cacc[MAXB];
    const T *L10 = Ls, *L20 = Ls + BLK, *L30 = Ls + 2 * BLK, *L21 = Ls + 3 * BLK, *L31 = Ls + 4 * BLK, *L32 = Ls + 5 * BLK;
    const T *Xd0 = Xd, *Xd1 = Xd + BLK, *Xd2 = Xd + 2 * BLK, *Xd3 = Xd + 3 * BLK;
    T *X20 = T0, *X21 = Ls, *S32 = Ls + BLK, *S30 = Ls + 3 * BLK;
    // a product chain inside one wave goes through LDS: the LDS queue of a wave is in order, the fence only keeps the
    // compiler from moving the reads of the next product above the stores of this one
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    T dvec = T(1);                                 // wave 0: D of the current sub-block (lane = row), kept for the write-out
    unsigned long long mneg = 0, mbad = 0;
    // L11 (strictly lower, from Lx) and D of sub-block jb to global: wave 0, off the critical path (steps B and C).
    // (Tried: the inverse's recurrence on wave 1, fed column by column through Lx with self-validating entries -- the LDS
    // polls make it the slower of the two waves: step A 5.0 us instead of 4.4.)
    auto write_out = [&](int jb, int lane) {
        const int c0 = NB * jb, half = lane >> 5, col = lane & 31;
#pragma unroll
        for (int c = 0; c < NB / 2; ++c) {
            const int cc = half * (NB / 2) + c;
            if (cc < col)
                A[(size_t)(c0 + col) * lda + c0 + cc] = Lx[col * PLD + cc];
        }
        if (lane < NB) {
            A[(size_t)(c0 + lane) * lda + c0 + lane] = dvec;
            d[blk * TILE + c0 + lane] = dvec;
            dinv[blk * TILE + c0 + lane] = Di[c0 + lane];
            if (lane == 0) {
                if (mbad)
                    atomicCAS(&info[0], 0, blk * TILE + c0 + 1);
                if (mneg)
                    atomicAdd(&info[1], __builtin_popcountll(mneg));
            }
        }
    };

    GPX_STAMP(0);
    if (wave >= 2) {
        // the trailing blocks this wave owns, in accumulator layout (a diagonal block whole: its upper half is never read)
#pragma unroll
        for (int e = 0; e < MAXB; ++e) {
            int bi, bj;
            own_block<NCW>(wave - 2, e, bi, bj);
            cacc[e].load(A + (size_t)(NB * bi) * lda + NB * bj, lda, lane0);
        }
    }
    for (int jb = 0; jb < 4; ++jb) {
        const int c0 = NB * jb;
        const int nrows = TILE - c0;
        // the lane index is made opaque once per panel: left loop-invariant, hipcc computes the per-lane addresses of every
        // block load / store of all four panels (16 64-bit pointers each) at kernel entry and spills >100 VGPRs to scratch
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        GPX_STAMP(1 + 4 * jb);
        // ---- A: wave 0 factorises sub-block jb and inverts its L on the matrix core ----
        if (wave == 0) {
            if (jb == 0) {  // the first sub-block comes from global; the later ones were left in Dn by the trailing update
#pragma unroll
                for (int k = 0; k < NB * NB / 64; ++k) {
                    const int idx = k * 64 + lane;
                    Dn[(idx >> 5) * PLD + (idx & 31)] = A[(size_t)(idx >> 5) * lda + (idx & 31)];
                }
                wave_sync();
            }
            subblock_ldl(Dn, Lx, Xd + jb * BLK, lane, dvec);
            const bool neg = lane < NB && dvec < T(0);
            const bool bad = lane < NB && (!(fabs(dvec) > T(0)) || !(fabs(dvec) < pivot_huge(T(0))));
            mneg = __ballot(neg), mbad = __ballot(bad);
            if (lane < NB)
                Di[c0 + lane] = T(1) / dvec;
        } else if (jb == 0) {
            // the rows below the first sub-block -> panel buffer (the later panels are left there by the trailing update);
            // all loads first: as one load-store loop the 7 trips were 7 serial round trips to L2 (fp64: 7 us)
            constexpr int NST = ((TILE - NB) * NB + DT - 64 - 1) / (DT - 64);
            T stage[NST];
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int idx = tid - 64 + k * (DT - 64);
                stage[k] = idx < (TILE - NB) * NB ? A[(size_t)(NB + (idx >> 5)) * lda + (idx & 31)] : T(0);
            }
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int idx = tid - 64 + k * (DT - 64);
                if (idx < (TILE - NB) * NB)
                    Pv[(NB + (idx >> 5)) * PLD + (idx & 31)] = stage[k];
            }
        }
        if (wave >= 2) {
            // the column of this panel below the sub-block, final since the last trailing update, from the registers of its
            // owners to the panel buffer (W of the previous panel is dead since the barrier)
            if (jb >= 1) {
#pragma unroll
                for (int e = 0; e < MAXB; ++e) {
                    int bi, bj;
                    own_block<NCW>(wave - 2, e, bi, bj);
                    if (bj == jb && bi > bj)
                        cacc[e].store(T(1), Pv + NB * bi * PLD, (T *)nullptr, 0, lane);
                }
            }
            // what is known of X by now: zeros right of the diagonal block in block row jb, diagonal block jb - 1
            for (int idx = tid - 128; idx < NB * (TILE - c0 - NB); idx += DT - 128) {
                const int w_ = TILE - c0 - NB, r_ = idx / w_, c_ = idx - r_ * w_;
                Xg[(size_t)(c0 + r_) * TILE + c0 + NB + c_] = T(0);
            }
            if (jb >= 1)
                for (int idx = tid - 128; idx < NB * NB; idx += DT - 128) {
                    const int b = jb - 1, r_ = idx >> 5, c_ = idx & 31;
                    Xg[(size_t)(b * NB + r_) * TILE + b * NB + c_] = Xd[(b * NB + r_) * PLD + c_];
                }
        }
        // left-looking assembly, in the shadow of step A
        if (jb == 2 && wave == NW - 1) {         // block row 1: X10 = -Xd1 (L10 Xd0)
            BlkAcc<T> acc;
            acc.zero();
            acc.mac(L10, Xd0, lane);
            acc.store(T(1), T0, (T *)nullptr, 0, lane);
            wave_sync();
            acc.zero();
            acc.mac(Xd1, T0, lane);
            acc.store(T(-1), X10, Xg + (size_t)NB * TILE, TILE, lane);
        }
        if (jb == 3 && wave == NW - 2) {         // S30 = L30 Xd0 + L31 X10 + L32 X20
            BlkAcc<T> acc;
            acc.zero();
            acc.mac(L30, Xd0, lane);
            acc.mac(L31, X10, lane);
            acc.mac(L32, X20, lane);
            acc.store(T(1), S30, (T *)nullptr, 0, lane);
        }
        if (jb == 3 && wave == NW - 1) {         // S31 = L31 Xd1 + L32 X21
            BlkAcc<T> acc;
            acc.zero();
            acc.mac(L31, Xd1, lane);
            acc.mac(L32, X21, lane);
            acc.store(T(1), S31, (T *)nullptr, 0, lane);
        }
        lds_barrier();
        GPX_STAMP(2 + 4 * jb);
        if (wave == 0 && jb < 3)
            write_out(jb, lane);         // (wave 0 has nothing else to do in B and C; Lx is not touched again before the next A)
        const int nb_rows = nrows - NB;  // rows below the diagonal sub-block
        if (nb_rows > 0) {
            T *Lsp = Ls + (jb == 0 ? 0 : (jb == 1 ? 3 : 5)) * BLK;  // L blocks (jb+1 .., jb)
            // ---- B: W = A21 X11^T on MFMA, one 32-row block per wave; L21 = W D^-1 to LDS and to global ----
            if (wave >= 1 && wave <= nb_rows / NB) {
                const int t = wave - 1;
                T *Wb = Pv + (c0 + NB + NB * t) * PLD;
                BlkAcc<T> wacc;
                wacc.zero();
                wacc.template mac_nt<false>(Wb, Xd + jb * BLK, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * i + BlkMma<T>::crow(lane, r), cl = 16 * j + (lane & 15);
                            const T w = wacc.t[i][j][r], l = w * Di[c0 + cl];
                            Wb[row * PLD + cl] = w;
                            Lsp[(NB * t + row) * PLD + cl] = l;
                            A[(size_t)(c0 + NB + NB * t + row) * lda + c0 + cl] = l;
                        }
            }
            if (jb == 2 && wave == NW - 1) {     // block row 2, column 1, first half: L21 Xd1 into the dead L10 block
                BlkAcc<T> acc;
                acc.zero();
                acc.mac(L21, Xd1, lane);
                acc.store(T(1), X21, (T *)nullptr, 0, lane);
            }
            if (jb == 2 && wave == NW - 2) {     // column 0, first half: T0 = L20 Xd0 + L21 X10
                BlkAcc<T> acc;
                acc.zero();
                acc.mac(L20, Xd0, lane);
                acc.mac(L21, X10, lane);
                acc.store(T(1), T0, (T *)nullptr, 0, lane);
            }
            lds_barrier();
            GPX_STAMP(3 + 4 * jb);
            // ---- C: trailing update A22 -= W L21^T of the blocks this wave owns (registers); the next diagonal sub-block
            //         goes to wave 0 through Dn, the rows below it to the panel buffer at the start of the next step ----
            if (wave >= 2) {
#pragma unroll
                for (int e = 0; e < MAXB; ++e) {
                    int bi, bj;
                    own_block<NCW>(wave - 2, e, bi, bj);
                    if (bj > jb) {
                        cacc[e].template mac_nt<true>(Pv + NB * bi * PLD, Lsp + (bj - jb - 1) * BLK, lane);
                        if (bj == jb + 1 && bi == bj)
                            cacc[e].store(T(1), Dn, (T *)nullptr, 0, lane);
                    }
                }
            }
            if (jb == 2) {
                if (wave == HC) {                // second half: X20 = -Xd2 T0, in place
                    BlkAcc<T> acc;
                    acc.zero();
                    acc.mac(Xd2, T0, lane);
                    wave_sync();
                    acc.store(T(-1), X20, Xg + (size_t)2 * NB * TILE, TILE, lane);
                }
                if (wave == 0) {                 // second half: X21 = -Xd2 (L21 Xd1), in place
                    BlkAcc<T> acc;
                    acc.zero();
                    acc.mac(Xd2, X21, lane);
                    wave_sync();
                    acc.store(T(-1), X21, Xg + (size_t)2 * NB * TILE + NB, TILE, lane);
                }
                if (wave == 1) {                 // S32 = L32 Xd2, in the dead L20 block
                    BlkAcc<T> acc;
                    acc.zero();
                    acc.mac(L32, Xd2, lane);
                    acc.store(T(1), S32, (T *)nullptr, 0, lane);
                }
            }
        }
        lds_barrier();
        GPX_STAMP(4 + 4 * jb);
    }
    GPX_STAMP(20);
    GPX_STAMP(21);
    GPX_STAMP(22);
    const int lane = lane0;
    // ---- last block row of X: X3c = -Xd3 S3c, diagonal block 3 ----
    if (wave >= 1 && wave <= 3) {
        BlkAcc<T> acc;
        acc.zero();
        acc.mac(Xd3, wave == 1 ? S30 : (wave == 2 ? S31 : S32), lane);
        acc.store(T(-1), (T *)nullptr, Xg + (size_t)3 * NB * TILE + (wave - 1) * NB, TILE, lane);
    } else {
        if (wave == 0)
            write_out(3, lane);
        const int ft = wave == 0 ? lane : 64 + (tid - 256);
        constexpr int NF = 64 + (DT > 256 ? DT - 256 : 0);
        for (int idx = ft; idx < NB * NB; idx += NF)
            Xg[(size_t)(3 * NB + (idx >> 5)) * TILE + 3 * NB + (idx & 31)] = Xd3[(idx >> 5) * PLD + (idx & 31)];
    }
    GPX_STAMP(23);
    GPX_STAMP(24);
}

static size_t diagm_shmem_bytes(size_t esz)
{
    return esz * (size_t)(TILE + 12 * NB * PLD + 96 * PLD);  // Di + Ls[6] Xd[4] Dn Lx + panel rows 32 .. 127
}
template <typename T, int DT>
static void diagm_t(void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk, hipStream_t st)
{
    hipLaunchKernelGGL((diag_ldlm_kernel<T, DT>), dim3(1), dim3(DT), diagm_shmem_bytes(sizeof(T)), st, (T *)Ablk, lda, (T *)linv,
                       (T *)d, (T *)dinv, info, blk);
}
void factor_init(int prec)
{
    static PerDeviceOnce once64, once32;  // per device, see gpx_internal.hpp
    if (prec == GPX_PREC_F64) {
        once64.run([] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldlm_kernel<double, DIAG_THREADS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)diagm_shmem_bytes(sizeof(double)));
        });
    } else {
        once32.run([] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldlm_kernel<float, DIAG_THREADS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)diagm_shmem_bytes(sizeof(float)));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldlm_kernel<float, DIAG_THREADS_NARROW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)diagm_shmem_bytes(sizeof(float)));
        });
    }
}

// narrow: the 4-wave instantiation (fp32 only), for launches that run beside a GEMM on another stream
void launch_diag_ldl(int prec, void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk,
                     hipStream_t st, bool narrow)
{
    if (prec == GPX_PREC_F64)
        diagm_t<double, DIAG_THREADS>(Ablk, lda, linv, d, dinv, info, blk, st);
    else if (narrow)
        diagm_t<float, DIAG_THREADS_NARROW>(Ablk, lda, linv, d, dinv, info, blk, st);
    else
        diagm_t<float, DIAG_THREADS>(Ablk, lda, linv, d, dinv, info, blk, st);
}

// Diagonal blocks that lie entirely in the padding (the kernel matrix is the identity there): L = I, D = 1, inverse = I
template <typename T>
__global__ __launch_bounds__(256) void identity_blocks_kernel(int blk0, T *__restrict__ linv, T *__restrict__ d,
                                                              T *__restrict__ dinv)
{
    const int blk = blk0 + blockIdx.x;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256)
        linv[(size_t)blk * TILE * TILE + idx] = (idx >> 7) == (idx & 127) ? T(1) : T(0);
    if (threadIdx.x < TILE) {
        d[blk * TILE + threadIdx.x] = T(1);
        dinv[blk * TILE + threadIdx.x] = T(1);
    }
}

void launch_identity_blocks(int prec, int blk0, int nblk, void *linv, void *d, void *dinv, hipStream_t st)
{
    if (blk0 >= nblk)
        return;
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(identity_blocks_kernel<double>, dim3(nblk - blk0), dim3(256), 0, st, blk0, (double *)linv,
                           (double *)d, (double *)dinv);
    else
        hipLaunchKernelGGL(identity_blocks_kernel<float>, dim3(nblk - blk0), dim3(256), 0, st, blk0, (float *)linv,
                           (float *)d, (float *)dinv);
}

template <typename T>
__global__ __launch_bounds__(256) void place_diag_kernel(const T *__restrict__ linv, T *__restrict__ X, long ldx)
{
    const int blk = blockIdx.x;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256) {
        const int i = idx >> 7, j = idx & 127;
        X[(size_t)(blk * TILE + i) * ldx + blk * TILE + j] = linv[(size_t)blk * TILE * TILE + idx];
    }
}

void launch_place_diag(int prec, int nblk, const void *linv, void *X, long ldx, hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(place_diag_kernel<double>, dim3(nblk), dim3(256), 0, st, (const double *)linv,
                           (double *)X, ldx);
    else
        hipLaunchKernelGGL(place_diag_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float *)linv, (float *)X,
                           ldx);
}

// y[r] = sum_c M[r][c] v[c] for a 128 x 128 row-major block; one wave per 32 rows.
template <typename T>
__device__ __forceinline__ void block_matvec(const T *__restrict__ M, long ldm, const T *v_lds, T *out_lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T v0 = v_lds[lane], v1 = v_lds[lane + 64];
    T m0[32], m1[32];  // issue all 64 loads of the wave's 32 rows before any reduction (latency-bound otherwise)
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const T *row = M + (size_t)(wave * 32 + rr) * ldm;
        m0[rr] = row[lane];
        m1[rr] = row[lane + 64];
    }
    // 32 row sums over 64 lanes by recursive halving: a lane gives away the half of the rows its partner keeps, so
    // the wave needs 16 + 8 + 4 + 2 + 1 + 1 = 32 shuffles instead of 32 x 6 for one butterfly per row; afterwards
    // lane l holds row l >> 1
    T s[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr)
        s[rr] = m0[rr] * v0 + m1[rr] * v1;
#define GPX_HALVE(N, OFF)                                       \
    _Pragma("unroll") for (int i = 0; i < (N); ++i)             \
    {                                                           \
        const bool up = (lane & (OFF)) != 0;                    \
        const T send = up ? s[i] : s[i + (N)];                  \
        const T keep = up ? s[i + (N)] : s[i];                  \
        s[i] = keep + __shfl_xor(send, (OFF));                  \
    }
    GPX_HALVE(16, 32)
    GPX_HALVE(8, 16)
    GPX_HALVE(4, 8)
    GPX_HALVE(2, 4)
    GPX_HALVE(1, 2)
#undef GPX_HALVE
    s[0] += __shfl_xor(s[0], 1);
    if ((lane & 1) == 0)
        out_lds[wave * 32 + (lane >> 1)] = s[0];
}

// y[c] = sum_r M[r][c] v[r]; threads 0..127 own a column, two halves of r combined through LDS.
template <typename T>
__device__ __forceinline__ void block_matvec_t(const T *__restrict__ M, long ldm, const T *v_lds, T *out_lds,
                                               T *scratch_lds)
{
    const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
    T mv[64];
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        mv[rr] = M[(size_t)(hh * 64 + rr) * ldm + c];
    T s = T(0);
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        s += mv[rr] * v_lds[hh * 64 + rr];
    if (hh == 1)
        scratch_lds[c] = s;
    __syncthreads();
    if (hh == 0)
        out_lds[c] = s + scratch_lds[c];
}

// Step kb of L y = b (unit lower L, 128-blocks): y_kb = Linv_kb b_kb ; b_rb -= L[rb,kb] y_kb (rb > kb).
// b is the working right-hand side (updated), y the output.  grid = nblk - kb.
template <typename T>
__global__ __launch_bounds__(256) void fwd_step_kernel(int kb, const T *__restrict__ L, long ld,
                                                       const T *__restrict__ linv, T *__restrict__ b,
                                                       T *__restrict__ y)
{
    __shared__ T bk[TILE], yk[TILE], upd[TILE];
    const int tid = threadIdx.x;
    if (tid < TILE)
        bk[tid] = b[kb * TILE + tid];
    __syncthreads();
    block_matvec<T>(linv + (size_t)kb * TILE * TILE, TILE, bk, yk);
    __syncthreads();
    const int rb = kb + blockIdx.x;
    if (blockIdx.x == 0) {
        if (tid < TILE)
            y[kb * TILE + tid] = yk[tid];
        return;
    }
    block_matvec<T>(L + (size_t)rb * TILE * ld + (size_t)kb * TILE, ld, yk, upd);
    __syncthreads();
    if (tid < TILE)
        b[rb * TILE + tid] -= upd[tid];
}

// Step kb of L^T x = y: x_kb = Linv_kb^T y_kb ; y_cb -= L[kb,cb]^T x_kb (cb < kb).  grid = kb + 1.
template <typename T>
__global__ __launch_bounds__(256) void bwd_step_kernel(int kb, const T *__restrict__ L, long ld,
                                                       const T *__restrict__ linv, T *__restrict__ y,
                                                       T *__restrict__ x)
{
    __shared__ T yk[TILE], xk[TILE], upd[TILE], scratch[TILE];
    const int tid = threadIdx.x;
    if (tid < TILE)
        yk[tid] = y[kb * TILE + tid];
    __syncthreads();
    block_matvec_t<T>(linv + (size_t)kb * TILE * TILE, TILE, yk, xk, scratch);
    __syncthreads();
    const int cb = blockIdx.x;
    if (cb == kb) {
        if (tid < TILE)
            x[kb * TILE + tid] = xk[tid];
        return;
    }
    block_matvec_t<T>(L + (size_t)kb * TILE * ld + (size_t)cb * TILE, ld, xk, upd, scratch);
    __syncthreads();
    if (tid < TILE)
        y[cb * TILE + tid] -= upd[tid];
}

// ---- the whole substitution in one launch per direction ------------------------------------------------------
// Workgroup i owns block row i of L y = b: it subtracts L(i,k) y_k for k = 0 .. i-1 as the y_k appear and then
// publishes y_i = Linv_i b_i.  The output vector is filled with a sentinel (all-ones NaN, which arithmetic never
// produces) before the launch and every consumer polls the entries it needs with cache-bypassing loads until they
// are no sentinel any more: the data validates itself, no flag, no fence.  Workgroups are dispatched in blockIdx
// order and wait only for lower ones, so the scheme cannot deadlock even when not all of them are resident.  The
// 128 x 128 block of the NEXT step (the last one: Linv_i) is loaded into a second register set before the poll of
// the current one, so a block costs a poll round trip and a register matvec (~2.5 us) -- with the loads behind the
// poll it was 8 us and the 127 blocks of the last row took as long as 256 step launches.  In-order dispatch is what
// the hardware does, not a guarantee of the programming model (other streams, host threads and processes share the
// GPU): a poll that exceeds the spin limit gives up with 0 and raises info[5], every workgroup still reaches its
// end, and the host then REDOES the solve with the launch-per-step kernels below (build_model, gpx_build.hip;
// gpx_stats.solve_fallbacks) -- neither a hung GPU nor a failed call.
// The backward direction runs the same scheme bottom-up on L^T, with D^-1 folded into its start.
constexpr int SOLVE_SPIN_LIMIT = 1 << 21;

template <typename T>
__device__ __forceinline__ void block_load(const T *__restrict__ M, long ldm, T (&m0)[32], T (&m1)[32])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const T *row = M + (size_t)(wave * 32 + rr) * ldm;
        m0[rr] = row[lane];
        m1[rr] = row[lane + 64];
    }
}
// out = M v with M in the registers filled by block_load
template <typename T>
__device__ __forceinline__ void block_mv_regs(const T (&m0)[32], const T (&m1)[32], const T *v_lds, T *out_lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T v0 = v_lds[lane], v1 = v_lds[lane + 64];
    T s[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr)
        s[rr] = m0[rr] * v0 + m1[rr] * v1;
#define GPX_HALVE(N, OFF)                                       \
    _Pragma("unroll") for (int i = 0; i < (N); ++i)             \
    {                                                           \
        const bool up = (lane & (OFF)) != 0;                    \
        const T send = up ? s[i] : s[i + (N)];                  \
        const T keep = up ? s[i + (N)] : s[i];                  \
        s[i] = keep + __shfl_xor(send, (OFF));                  \
    }
    GPX_HALVE(16, 32)
    GPX_HALVE(8, 16)
    GPX_HALVE(4, 8)
    GPX_HALVE(2, 4)
    GPX_HALVE(1, 2)
#undef GPX_HALVE
    s[0] += __shfl_xor(s[0], 1);
    if ((lane & 1) == 0)
        out_lds[wave * 32 + (lane >> 1)] = s[0];
}
// transposed: thread (c = tid & 127, hh = tid >> 7) holds rows hh*64 .. +63 of column c
template <typename T>
__device__ __forceinline__ void block_load_t(const T *__restrict__ M, long ldm, T (&mv)[64])
{
    const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        mv[rr] = M[(size_t)(hh * 64 + rr) * ldm + c];
}
template <typename T>
__device__ __forceinline__ void block_mv_regs_t(const T (&mv)[64], const T *v_lds, T *out_lds, T *scratch_lds)
{
    const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
    T s = T(0);
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        s += mv[rr] * v_lds[hh * 64 + rr];
    if (hh == 1)
        scratch_lds[c] = s;
    __syncthreads();
    if (hh == 0)
        out_lds[c] = s + scratch_lds[c];
}

// entry *p of the shared vector once it is no sentinel any more (0 and info[5] = 1 after spin_limit polls)
__device__ __forceinline__ float poll_entry(const float *p, int *info, int spin_limit)
{
    const unsigned *u = reinterpret_cast<const unsigned *>(p);
    for (int spins = 0; spins < spin_limit; ++spins) {
        const unsigned b = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != 0xffffffffu)
            return __uint_as_float(b);
        __builtin_amdgcn_s_sleep(1);
    }
    atomicExch(&info[5], 1);
    return 0.0f;
}
__device__ __forceinline__ double poll_entry(const double *p, int *info, int spin_limit)
{
    const unsigned long long *u = reinterpret_cast<const unsigned long long *>(p);
    for (int spins = 0; spins < spin_limit; ++spins) {
        const unsigned long long b = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != 0xffffffffffffffffull)
            return __longlong_as_double((long long)b);
        __builtin_amdgcn_s_sleep(1);
    }
    atomicExch(&info[5], 1);
    return 0.0;
}
__device__ __forceinline__ void publish_entry(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_entry(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// TRANS = false: L y = b, rows top-down.  TRANS = true: L^T x = D^-1 y, rows bottom-up (scale = 1/D).
// `out` must hold the sentinel in every entry at launch.
template <typename T, bool TRANS>
__global__ __launch_bounds__(256) void tri_solve_kernel(int nblk, const T *__restrict__ L, long ld,
                                                        const T *__restrict__ linv, const T *__restrict__ scale,
                                                        const T *__restrict__ rhs, T *out, int *info, int spin_limit)
{
    __shared__ T vk[TILE], upd[TILE], scratch[TILE];
    constexpr int RS = TRANS ? 64 : 32;  // registers of one half block set
    const int tid = threadIdx.x;
    const int i = TRANS ? nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int nprev = TRANS ? nblk - 1 - i : i;  // blocks this row depends on, in the order they get ready
    T acc = T(0);                                // entry tid of the working right-hand side (tid < TILE)
    if (tid < TILE)
        acc = TRANS ? rhs[i * TILE + tid] * scale[i * TILE + tid] : rhs[i * TILE + tid];
    // block s of the sequence: L(i, s) (TRANS: L(nblk-1-s, i)) for s < nprev, then the inverse diagonal block
    auto src = [&](int s) -> const T * {
        if (s >= nprev)
            return linv + (size_t)i * TILE * TILE;
        return TRANS ? L + (size_t)(nblk - 1 - s) * TILE * ld + (size_t)i * TILE : L + (size_t)i * TILE * ld + (size_t)s * TILE;
    };
    auto ldm = [&](int s) -> long { return s >= nprev ? (long)TILE : ld; };
    T a0[RS], a1[TRANS ? 1 : 32], b0[RS], b1[TRANS ? 1 : 32];
    auto load = [&](int s, T(&r0)[RS], T(&r1)[TRANS ? 1 : 32]) {
        if constexpr (TRANS)
            block_load_t<T>(src(s), ldm(s), r0);
        else
            block_load<T>(src(s), ldm(s), r0, r1);
    };
    auto mv = [&](const T(&r0)[RS], const T(&r1)[TRANS ? 1 : 32]) {
        if constexpr (TRANS)
            block_mv_regs_t<T>(r0, vk, upd, scratch);
        else
            block_mv_regs<T>(r0, r1, vk, upd);
    };
    // one step: the block in (r0, r1) times the published vector block of step s
    auto step = [&](int s, const T(&r0)[RS], const T(&r1)[TRANS ? 1 : 32]) {
        const int k = TRANS ? nblk - 1 - s : s;
        if (tid < TILE)
            vk[tid] = poll_entry(out + k * TILE + tid, info, spin_limit);
        __syncthreads();
        mv(r0, r1);
        __syncthreads();
        if (tid < TILE)
            acc -= upd[tid];
    };
    load(0, a0, a1);
    bool last_in_b = false;
    for (int s = 0; s < nprev; s += 2) {
        load(s + 1, b0, b1);
        step(s, a0, a1);
        if (s + 1 < nprev) {
            load(s + 2, a0, a1);
            step(s + 1, b0, b1);
        } else {
            last_in_b = true;
        }
    }
    if (tid < TILE)
        vk[tid] = acc;
    __syncthreads();
    if (last_in_b)
        mv(b0, b1);
    else
        mv(a0, a1);
    __syncthreads();
    if (tid < TILE)
        publish_entry(out + i * TILE + tid, upd[tid]);
}

template <typename T>
static void tri_solve_t(int nblk, const void *L, long ld, const void *linv, const void *dinv, const void *b, void *y,
                        void *x, int *info, hipStream_t st, int spin_limit)
{
    if (spin_limit <= 0)
        spin_limit = SOLVE_SPIN_LIMIT;
    const size_t words = (size_t)nblk * TILE * (sizeof(T) / 4);
    if ((char *)x == (char *)y + words * 4) {  // adjacent (the model's vectors are): one fill
        (void)hipMemsetD32Async((hipDeviceptr_t)y, (int)0xffffffff, 2 * words, st);
    } else {
        (void)hipMemsetD32Async((hipDeviceptr_t)y, (int)0xffffffff, words, st);
        (void)hipMemsetD32Async((hipDeviceptr_t)x, (int)0xffffffff, words, st);
    }
    hipLaunchKernelGGL((tri_solve_kernel<T, false>), dim3(nblk), dim3(256), 0, st, nblk, (const T *)L, ld, (const T *)linv,
                       (const T *)nullptr, (const T *)b, (T *)y, info, spin_limit);
    hipLaunchKernelGGL((tri_solve_kernel<T, true>), dim3(nblk), dim3(256), 0, st, nblk, (const T *)L, ld, (const T *)linv,
                       (const T *)dinv, (const T *)y, (T *)x, info, spin_limit);
}

// x = (L D L^T)^-1 b (y: scratch of the same length), two launches
void launch_tri_solve(int prec, int nblk, const void *L, long ld, const void *linv, const void *dinv, const void *b,
                      void *y, void *x, int *info, hipStream_t st, int spin_limit)
{
    if (prec == GPX_PREC_F64)
        tri_solve_t<double>(nblk, L, ld, linv, dinv, b, y, x, info, st, spin_limit);
    else
        tri_solve_t<float>(nblk, L, ld, linv, dinv, b, y, x, info, st, spin_limit);
}

void launch_fwd_step(int prec, int kb, int nblk, const void *L, long ld, const void *linv, void *b, void *y,
                     hipStream_t st)
{
    dim3 grid(nblk - kb);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(fwd_step_kernel<double>, grid, dim3(256), 0, st, kb, (const double *)L, ld,
                           (const double *)linv, (double *)b, (double *)y);
    else
        hipLaunchKernelGGL(fwd_step_kernel<float>, grid, dim3(256), 0, st, kb, (const float *)L, ld,
                           (const float *)linv, (float *)b, (float *)y);
}

void launch_bwd_step(int prec, int kb, const void *L, long ld, const void *linv, void *y, void *x, hipStream_t st)
{
    dim3 grid(kb + 1);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(bwd_step_kernel<double>, grid, dim3(256), 0, st, kb, (const double *)L, ld,
                           (const double *)linv, (double *)y, (double *)x);
    else
        hipLaunchKernelGGL(bwd_step_kernel<float>, grid, dim3(256), 0, st, kb, (const float *)L, ld,
                           (const float *)linv, (float *)y, (float *)x);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_vec_kernel(int n, T *__restrict__ b, const T *__restrict__ s)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        b[i] *= s[i];
}

void launch_scale_vec(int prec, int npad, void *b, const void *dinv, hipStream_t st)
{
    dim3 grid((npad + 255) / 256);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(scale_vec_kernel<double>, grid, dim3(256), 0, st, npad, (double *)b,
                           (const double *)dinv);
    else
        hipLaunchKernelGGL(scale_vec_kernel<float>, grid, dim3(256), 0, st, npad, (float *)b, (const float *)dinv);
}

}  // namespace gpx
