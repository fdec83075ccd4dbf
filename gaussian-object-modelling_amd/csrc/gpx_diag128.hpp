// gpx_diag128.hpp -- LDL^T of one 128 x 128 diagonal block + the inverse of its L on the matrix cores (one workgroup of 8 / 4
// waves): the body shared by the launch chain (gpx_factor.hip: diag_ldlm_kernel) and the 128 x 128 dataflow factorisation
// (gpx_dataflow_wide.hpp).  gfx950 only.
#pragma once
#include "gpx_blk.hpp"

// phase timing for scripts/diag_bench.hip (which defines GPX_STAMP); nothing in the library build
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


// (device body: the launch chain's kernel below and the 128 x 128 dataflow factorisation of gpx_dataflow_wide.hpp call it)
template <typename T, int DT>
__device__ __forceinline__ void diag_ldlm_body(T *__restrict__ A, long lda, T *__restrict__ linv, T *__restrict__ d,
                                               T *__restrict__ dinv, int *__restrict__ info, int blk,
                                               unsigned char *smem_raw)
{
    constexpr int BLK = NB * PLD;                  // one 32 x 32 LDS block
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

template <typename T, int DT>
__global__ __launch_bounds__(DT, 2) void diag_ldlm_kernel(T *__restrict__ A, long lda, T *__restrict__ linv,
                                                       T *__restrict__ d, T *__restrict__ dinv, int *__restrict__ info,
                                                       int blk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    diag_ldlm_body<T, DT>(A, lda, linv, d, dinv, info, blk, smem_raw);
}

inline size_t diagm_shmem_bytes(size_t esz)
{
    return esz * (size_t)(TILE + 12 * NB * PLD + 96 * PLD);  // Di + Ls[6] Xd[4] Dn Lx + panel rows 32 .. 127
}

}  // namespace gpx
