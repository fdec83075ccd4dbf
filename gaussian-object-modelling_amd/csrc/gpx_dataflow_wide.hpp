// gpx_dataflow_wide.hpp -- the dataflow factorisation of gpx_dataflow.hpp with 128 x 128 tiles, for LARGE models.
//
// With 64 x 64 tiles every workgroup streams its own row panel L_i,0..j-1 from memory: 45 GB of tile reads at N = 16384, and the
// PMC counters show the fp32 launch moving 61.7 GB at 3.6 TB/s (profiles/r05_pmc_traffic_64tiles.json) -- the kernel is HBM-bound
// there.  A 128 x 128 tile does four times the flops of a 64 x 64 one on twice the operand bytes: half the traffic per flop
// (measured: 36.4 GB per launch, profiles/r05_pmc_traffic_w1.json, at 15.5 ms).
//
//   * one workgroup of EIGHT waves per lower 128 x 128 tile (I, J), column by column; wave w owns rows 32 (w >> 1) .. +32 and
//     columns 64 (w & 1) .. +64 of the tile (two 32 x 32 accumulator blocks);
//   * the update loop walks 64-wide k slices (two per tile column): A = (L_I D) slice, B = L_J slice, 128 x 64 each, in the wide
//     LDS layout of gpx_dataflow.hpp; operands of the next slice in flight while the matrix cores work; fp32 sums in 256-column
//     chunks as there;
//   * a DIAGONAL tile stores its sums to its place in K and runs the launch chain's 128 x 128 diagonal-block routine on them
//     (gpx_diag128.hpp: LDL^T + the inverse of L, 27 us fp32 / 49 us fp64) -- L, D, 1/D and the inverse block `linv` come out of it;
//   * a tile BELOW the diagonal stores its sums the same way and forms L_IJ = (A_IJ Linv_J^T) D_J^-1 with the same slice loop
//     (A from its own tile, B = the inverse block);
//   * flags, write-through publication, ordinary loads on the consumer side, time budgets of the waits and the give-up path: as gpx_dataflow.hpp.
// No inverse-factor jobs: every wait is for a workgroup with a lower index.
// Where the time goes (round 6, make EXTRA=-DWIDE_TIMING; profiles/r06_wide_timing.txt; N = 16384 fp32, 15.6 ms, every CU busy throughout): waits for
// tile flags 2.5 %, operands into LDS + first barrier 10 %, MFMAs + second barrier 78 % (the matrix pipe 83 % busy inside it), kernel-matrix
// entries / stores / the rest 9.5 % -- neither the bytes (2.3 TB/s) nor the waits set the time.  Two forms that overlap the staging with the
// MFMAs were built on two sets of LDS buffers (147 KB) and are SLOWER: slice s + 1 written behind the MFMAs of slice s, one barrier per slice
// (22.1 ms), and the present order without the second barrier (21.3 ms) -- once the waves of a workgroup drift apart, the ds_write bursts of
// one wave sit in the LDS queue in front of the operand reads the other waves' MFMAs wait for; the two barriers keep the phases apart.
#pragma once
#include "gpx_dataflow.hpp"
#include "gpx_diag128.hpp"

namespace gpx {
namespace dataflow {

constexpr int WT = TILE;  // 128
constexpr int WIDE_TIMING_MAX_TILES = 16384;
constexpr int WIDE_THREADS = 512;
// LDS: two operands of 128 x 64 in the wide layout (8 blocks of 32 x 36 each) -- or the diagonal-block routine's own layout
constexpr int WIDE_OPERAND_ELEMS = 16 * WBLK;
template <typename T>
constexpr size_t wide_lds_bytes()
{
    constexpr size_t op = (size_t)WIDE_OPERAND_ELEMS * sizeof(T);
    return op > (sizeof(T) * (size_t)(TILE + 12 * NB * PLD + 96 * PLD)) ? op : sizeof(T) * (size_t)(TILE + 12 * NB * PLD + 96 * PLD);
}

// 128 x 64 slice at g (leading dimension ld): element e of thread t (512 threads) is (row 8 e + (t >> 6), column t & 63)
template <typename T>
__device__ __forceinline__ void slice_load(T (&v)[16], const T *g, long ld)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        v[e] = g[(size_t)(8 * e + (tid >> 6)) * ld + (tid & 63)];
}
// ... into eight 32 x 32 blocks [(r >> 5) * 2 + (c >> 5)] of the wide layout, the thread's column scaled by s
template <typename T>
__device__ __forceinline__ void slice_to_lds(T *buf, const T (&v)[16], T s)
{
    const int tid = threadIdx.x, c = tid & 63;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = 8 * e + (tid >> 6);
        buf[((r >> 5) * 2 + (c >> 5)) * WBLK + (r & 31) * WLD + (c & 31)] = v[e] * s;
    }
}

// make EXTRA=-DWIDE_TIMING OUTDIR=../lib_t OBJDIR=../build_t: shader-clock split of every workgroup of the wide launch -- [0] whole
// workgroup, [1] waits for tile flags, [2] operands into LDS + first barrier of a slice, [3] MFMAs + second barrier, [4] the
// diagonal-block routine; mid_factor_t (gpx_small.hip) sums them over the launch and prints one line (scripts/ldlt_sweep.py)
#ifdef WIDE_TIMING
__device__ unsigned long long wide_timing[WIDE_TIMING_MAX_TILES * 8];
#define WT_NOW() clock64()
#define WT_ADD(k, t0) wt_acc[k] += clock64() - (t0)
#else
#define WT_NOW() 0ull
#define WT_ADD(k, t0) (void)(t0)
#endif

template <typename T, int KID>
__device__ __forceinline__ void wide_factor_tile(const FactorArgs<T> &f, int *info, T *sm)
{
#ifdef WIDE_TIMING
    unsigned long long wt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long wt_start = clock64();
    struct WtFlush {
        unsigned long long *acc, start;
        __device__ ~WtFlush()
        {
            acc[0] = clock64() - start;
            if (threadIdx.x == 0 && blockIdx.x < WIDE_TIMING_MAX_TILES)
                for (int k = 0; k < 8; ++k)
                    wide_timing[(size_t)blockIdx.x * 8 + k] = acc[k];
        }
    } wt_flush{wt_acc, wt_start};
#endif
    __shared__ int s_ok, s_next;
    __shared__ double s_best[8];
    __shared__ int s_bi[8], s_bj[8];
    const unsigned long long wt_start0 = WT_NOW();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int np = f.np, n = f.n;
    int J = 0, rem = (int)blockIdx.x;
    while (rem >= f.nbt - J) {
        rem -= f.nbt - J;
        ++J;
    }
    const int I = J + rem;
    u64 *Ff = f.flags;
    T *Ktile = f.K + (size_t)(WT * I) * np + WT * J;
    T *lblk = f.linv + (size_t)I * WT * WT;
    if (I >= f.nb) {  // entirely in the padding: identity
        for (int e = 0; e < 32; ++e) {
            const int idx = e * WIDE_THREADS + tid, r = idx >> 7, c = idx & 127;
            const T v = (I == J && r == c) ? T(1) : T(0);
            Ktile[(size_t)r * np + c] = v;
            if (I == J)
                lblk[(size_t)r * WT + c] = v;
        }
        if (I == J && tid < WT) {
            f.d[WT * I + tid] = T(1);
            f.dinv[WT * I + tid] = T(1);
        }
        return;
    }
    // ---- the wave's 32 x 64 part of the kernel matrix, straight into its two accumulator blocks ----
    T *rowp = sm;            // [4][128] x y z s2 of the tile's rows
    T *colp = sm + 4 * WT;   // [3][128] x y z of its columns
    if (tid < WT) {
        const int r = WT * I + tid;
        rowp[tid] = f.px[r], rowp[WT + tid] = f.py[r], rowp[2 * WT + tid] = f.pz[r], rowp[3 * WT + tid] = f.ps2[r];
    } else if (tid < 2 * WT) {
        const int t = tid - WT, c = WT * J + t;
        colp[t] = f.px[c], colp[WT + t] = f.py[c], colp[2 * WT + t] = f.pz[c];
    }
    __syncthreads();
    BlkAcc<T> acc[2];
    {
        const Cov<T> cov = f.cov;
        double best = -1.0;
        int bi = 0, bj = 0;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 32 * wi + 16 * i2 + BlkMma<T>::crow(lane, r), col = 64 * wj + 32 * cb + 16 * j2 + (lane & 15);
                        const int gi = WT * I + row, gj = WT * J + col;
                        const T dx = rowp[row] - colp[col], dy = rowp[WT + row] - colp[WT + col],
                                dz = rowp[2 * WT + row] - colp[2 * WT + col];
                        const T d2 = dx * dx + dy * dy + dz * dz;
                        T kv = cov_k<T, KID>(cov, d2);
                        if (gi == gj)
                            kv += rowp[3 * WT + row];
                        if (gi < n && gj < n) {
                            if ((double)d2 > best)
                                best = (double)d2, bi = gi, bj = gj;
                        } else {
                            kv = gi == gj ? T(1) : T(0);  // identity on the padding
                        }
                        acc[cb].t[i2][j2][r] = kv;
                    }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
            if (ob > best)
                best = ob, bi = oi, bj = oj;
        }
        if (lane == 0)
            s_best[wave] = best, s_bi[wave] = bi, s_bj[wave] = bj;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 8; ++w)
                if (s_best[w] > best)
                    best = s_best[w], bi = s_bi[w], bj = s_bj[w];
            f.tmax[blockIdx.x] = best;
            f.tij[2 * blockIdx.x] = bi;
            f.tij[2 * blockIdx.x + 1] = bj;
        }
        __syncthreads();  // (rowp / colp are about to be overwritten by the operands)
    }
    WT_ADD(5, wt_start0);  // [5] prologue: points, the tile's kernel-matrix entries, arg-max
    T *bufA = sm, *bufB = sm + 8 * WBLK;
    // one product loop over `nsl` 64-wide k slices: acc += A_slice B_slice^T with A rows from gA (scaled per column by dsc, or 1),
    // B rows from gB; slice s of both starts s * 64 elements further along their rows.  ready(s): the producers' flags of slice s.
    constexpr bool CHUNKED = sizeof(T) == 4;
    T master[CHUNKED ? 32 : 1];  // fp32: the running value of the update, the accumulators hold one 256-column chunk (gpx_dataflow.hpp)
    auto product = [&](const T *gA, long ldA, const T *gB, long ldB, const T *dsc, T sign, int nsl, auto &&flag_a, auto &&flag_b,
                       bool chunked) -> bool {
        T va[16], vb[16];
        T dk = T(1);
        int sready = 0;  // slices [0, sready) are known to be published
        // A (the workgroup's own row panel: from memory) is requested before the products of the slice in flight, B (shared
        // by the whole tile column: L2-resident) half way through them -- both in flight behind MFMAs, and the 32 registers of
        // B are not held across the first half (the fp64 form runs on 256 registers per wave)
        auto issue_a = [&](int s) {
            slice_load(va, gA + (size_t)ST * s, ldA);
            dk = dsc ? dsc[ST * s + (tid & 63)] : T(1);
        };
        auto issue_b = [&](int s) { slice_load(vb, gB + (size_t)ST * s, ldB); };
        auto issue = [&](int s) {
            issue_a(s);
            issue_b(s);
        };
        // wait (blocking) for the tile column of slice s; afterwards every slice of that column is ready
        auto block_for = [&](int s) -> bool {
            const u64 *fa = flag_a(s >> 1), *fb = flag_b(s >> 1);
            if (fa || fb) {
                const unsigned long long tw0 = WT_NOW();
                const bool okw = wait_tiles(fa ? fa : fb, fa ? fb : nullptr, f, &s_ok);
                WT_ADD(1, tw0);
                if (!okw)
                    return false;
            }
            sready = max(sready, (s | 1) + 1);
            return true;
        };
        if (!block_for(0))
            return false;
        issue(0);
        for (int s = 0; s < nsl; ++s) {
            const unsigned long long tl0 = WT_NOW();
            slice_to_lds(bufA, va, sign * dk);
            slice_to_lds(bufB, vb, T(1));
            // a cheap look one tile column ahead (thread 0, one or two flag loads issued here, used after the products)
            u64 pa = f.epoch, pb = f.epoch;
            const bool look = s + 1 < nsl && s + 1 >= sready;
            if (look && tid == 0) {
                const u64 *fa = flag_a((s + 1) >> 1), *fb = flag_b((s + 1) >> 1);
                pa = fa ? ld_flag(fa) : f.epoch;
                pb = fb ? ld_flag(fb) : f.epoch;
            }
            __syncthreads();
            WT_ADD(2, tl0);
            const unsigned long long tm0 = WT_NOW();
            const bool ready = s + 1 < nsl && s + 1 < sready;
            if (ready)
                issue_a(s + 1);  // in flight while the matrix cores work on slice s
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    mac_nt_wide<T>(acc[cb], bufA + (wi * 2 + h) * WBLK, bufB + ((2 * wj + cb) * 2 + h) * WBLK, lane);
                if (h == 0 && ready)
                    issue_b(s + 1);
            }
            if constexpr (CHUNKED) {
                if (chunked && ((s & 3) == 3 || s + 1 == nsl)) {
#pragma unroll
                    for (int q = 0; q < 32; ++q) {
                        master[q] += acc[q >> 4].t[(q >> 3) & 1][(q >> 2) & 1][q & 3];
                        acc[q >> 4].t[(q >> 3) & 1][(q >> 2) & 1][q & 3] = s + 1 < nsl ? T(0) : master[q];
                    }
                }
            }
            if (look && tid == 0)
                s_next = (pa == f.epoch && pb == f.epoch) ? 1 : 0;
            __syncthreads();
            WT_ADD(3, tm0);
            if (!ready && s + 1 < nsl) {
                if (look && s_next)
                    sready = max(sready, ((s + 1) | 1) + 1);
                else if (!block_for(s + 1))
                    return false;
                issue(s + 1);
            }
        }
        return true;
    };
    // ---- A_IJ -= sum over the earlier tile columns of (L_I D) L_J^T ----
    if (J > 0) {
        if constexpr (CHUNKED) {
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                master[q] = acc[q >> 4].t[(q >> 3) & 1][(q >> 2) & 1][q & 3];
                acc[q >> 4].t[(q >> 3) & 1][(q >> 2) & 1][q & 3] = T(0);
            }
        }
        const bool ok = product(
            f.K + (size_t)(WT * I) * np, np, f.K + (size_t)(WT * J) * np, np, f.d, T(-1), 2 * J,
            [&](int kk) -> const u64 * { return Ff + tidx(I, kk); },
            [&](int kk) -> const u64 * { return I != J ? Ff + tidx(J, kk) : nullptr; }, CHUNKED);
        if (!ok)
            return;
    }
    // ---- the finished sums go to the tile's place in K (read back by this workgroup only: same L2) ----
    const unsigned long long wt_mid0 = WT_NOW();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
        acc[cb].store(T(1), (T *)nullptr, Ktile + (size_t)(32 * wi) * np + 64 * wj + 32 * cb, np, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    WT_ADD(6, wt_mid0);  // [6] sums stored and drained
    if (I == J) {
        // ---- diagonal tile: LDL^T + inverse of the 128 x 128 block by the launch chain's routine, in place ----
        const unsigned long long td0 = WT_NOW();
        diag_ldlm_body<T, WIDE_THREADS>(Ktile, np, f.linv, f.d, f.dinv, info, I, reinterpret_cast<unsigned char *>(sm));
        WT_ADD(4, td0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the routine stores with ordinary stores: write the L2 back once per tile
            st_flag(Ff + tidx(I, I), f.epoch);
        }
        return;
    }
    // ---- tile below the diagonal: L_IJ = (A_IJ Linv_J^T) D_J^-1, the same slice loop on (own tile, inverse block) ----
    acc[0].zero();
    acc[1].zero();
    {
        const bool ok = product(
            Ktile, np, f.linv + (size_t)J * WT * WT, WT, nullptr, T(1), 2,
            [&](int) -> const u64 * { return Ff + tidx(J, J); }, [&](int) -> const u64 * { return nullptr; }, false);
        if (!ok)
            return;
    }
    const unsigned long long wt_epi0 = WT_NOW();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[cb].t[i2][j2][r] *= f.dinv[WT * J + 64 * wj + 32 * cb + 16 * j2 + (lane & 15)];
        store_blk_cg<T>(acc[cb], T(1), nullptr, Ktile + (size_t)(32 * wi) * np + 64 * wj + 32 * cb, np, lane);
    }
    publish_tile(Ff + tidx(I, J), f.epoch);
    WT_ADD(7, wt_epi0);  // [7] scaled, stored write-through, acknowledged, flag up
}

}  // namespace dataflow
}  // namespace gpx
