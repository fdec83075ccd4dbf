// gpx_small.hip -- GPRegressor::create (reference gp_regressor.hpp:110-182) for the reference's OWN model sizes
// (N = 166 .. 724 points, rebuilt after every touch: src/gp_node.cpp:696, :750-751, :922) in THREE launches instead of
// the ~40 dependent launches of the general chain (gpx_build.hip), for models of up to 1024 padded rows trained in fp64:
//
//   small_factor_kernel  kernel matrix -> LDL^T -> inverse factor X = L^-1, as ONE dataflow over 64 x 64 tiles: a
//                        workgroup per lower tile (i, j) builds its entries of K from the points straight into MFMA
//                        accumulators, subtracts W_ik L_jk^T for k < j as those tiles are published, then either
//                        factorises (diagonal: two 32 x 32 sub-blocks by rank-1 MFMA updates, gpx_blk.hpp) or solves
//                        against the diagonal tile's inverse; afterwards the same workgroup forms its tile of X
//                        (X_ij = -Xd_i sum_k L_ik X_kj), which tracks the factorisation one product behind.  Tiles
//                        travel through global memory (L2 / MALL) with one flag per tile: release fence + flag store,
//                        flag poll + acquire fence (agent scope: the XCDs' L2s are not coherent with each other).
//   small_alpha_kernel   alpha = X^T D^-1 X y with fp64 matrix-free residual refinement (adaptive, as the chain), the 14
//                        row-correction vectors of the variance fit, the fit's weight offset and the result block;
//                        a grid of workgroups with an atomic-counter barrier between the phases.
//   small_demote_kernel  (fp32-mode models, which train in fp64 at this size) the fp32 state: blob part 0 and X rounded once.
//
// Every wait has a spin limit: a workgroup that runs out of patience raises the abort flag, all others see it in their
// polls and leave, and the host redoes the create with the general chain (gpx_stats.solve_fallbacks = 1) -- as
// tri_solve_kernel does.  Workgroups are enumerated column by column, so a factorisation job only ever waits for
// workgroups with a lower index; the inverse jobs also wait for later ones, which is why all (<= 136) must be resident.
#include <algorithm>
#include <atomic>

#include "gpx_blk.hpp"
#include "gpx_cov.hpp"
#include "gpx_small.hpp"

namespace gpx {

namespace {
constexpr int ST = SMALL_TILE;       // 64
constexpr int SBLK = NB * PLD;       // one 32 x 32 block in LDS (doubles)
constexpr int SM_THREADS = 256;
constexpr int SM_LDS_DOUBLES = 12 * SBLK + 2 * ST + 4 * ST + 3 * ST;
typedef unsigned long long u64;
#ifdef SM_TIMING
#define SM_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0)                                                                         \
            a.dbg[(size_t)blockIdx.x * SMALL_DBG_STAMPS + (k)] = wall_clock64();                      \
    } while (0)
#else
#define SM_STAMP(k)
#endif

__device__ __forceinline__ u64 ld_flag(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Data that travels between workgroups INSIDE a launch is stored and loaded at agent scope (sc1: written through to /
// read from the memory side, the point where the XCDs' L2s meet): the tiles and vectors validate themselves behind a flag
// or a barrier without any L2 write-back or invalidation -- so everything else (the points, X and X^T in the second
// launch) stays cached.  Same idea as tri_solve_kernel's entries (gpx_factor.hip).
__device__ __forceinline__ double ld_cg(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_cg(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 of the workgroup: wait until *f == want (false: the abort flag went up, or the spin limit ran out and this
// call raised it)
__device__ bool poll_flag(const u64 *f, u64 want, u64 *abortf, int limit)
{
    for (int s = 0; s < limit; ++s) {
        if (ld_flag(f) == want)
            return true;
        if ((s & 31) == 31 && ld_flag(abortf) == want)
            return false;
        __builtin_amdgcn_s_sleep(1);
    }
    st_flag(abortf, want);
    return false;
}

// all threads: wait for one or two tile flags (the tiles behind them are then read with ld_cg)
__device__ __forceinline__ bool wait_tiles(const u64 *f1, const u64 *f2, const SmallArgs &a, int *s_ok)
{
    if (threadIdx.x == 0) {
        bool ok = poll_flag(f1, a.epoch, a.flags + a.abort_idx, a.spin_limit);
        if (ok && f2)
            ok = poll_flag(f2, a.epoch, a.flags + a.abort_idx, a.spin_limit);
        *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();  // (s_ok is reused by the next wait)
    return ok;
}

// all threads: the tile this workgroup has just stored (st_cg) is complete, then the flag goes up
__device__ __forceinline__ void publish_tile(u64 *f, u64 v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0)
        st_flag(f, v);
}

__device__ __forceinline__ int tidx(int i, int j) { return i * (i + 1) / 2 + j; }

// 64 x 64 tile at g (leading dimension ld) -> four 32 x 32 LDS blocks [(r >> 5) * 2 + (c >> 5)], optionally with its
// columns scaled by colscale[c]
__device__ __forceinline__ void stage_tile(double *buf, const double *g, long ld, const double *colscale)
{
    const int tid = threadIdx.x;
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int idx = e * SM_THREADS + tid, r = idx >> 6, c = idx & 63;
        v[e] = ld_cg(g + (size_t)r * ld + c);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int idx = e * SM_THREADS + tid, r = idx >> 6, c = idx & 63;
        const double s = colscale ? colscale[c] : 1.0;
        buf[((r >> 5) * 2 + (c >> 5)) * SBLK + (r & 31) * PLD + (c & 31)] = v[e] * s;
    }
}

// sign * (32 x 32 accumulator block) -> LDS block (may be null) and / or global with write-through stores (may be null)
__device__ __forceinline__ void store_blk_cg(const BlkAcc<double> &b, double sign, double *lds, double *g, long ldg, int lane)
{
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i2 + BlkMma<double>::crow(lane, r), col = 16 * j2 + (lane & 15);
                const double v = sign * b.t[i2][j2][r];
                if (lds)
                    lds[row * PLD + col] = v;
                if (g)
                    st_cg(g + (size_t)row * ldg + col, v);
            }
}

// one 16 x 16 tile of a product of two 32 x 32 LDS blocks (K = 32) on one wave: rows 16 i2.. of A, columns 16 j2.. of the
// result; NT: B is [n][k] (A B^T), else [k][n].  The four waves of the workgroup share a 32 x 32 product this way.
typedef BlkMma<double>::acc_t acc16_t;
template <bool NT, bool NEG>
__device__ __forceinline__ acc16_t mma16(const double *Ab, const double *Bb, int i2, int j2, int lane, acc16_t t)
{
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const int k = 4 * kk + kq;
        const double av = Ab[(16 * i2 + i) * PLD + k];
        const double bv = NT ? Bb[(16 * j2 + i) * PLD + k] : Bb[k * PLD + 16 * j2 + i];
        t = BlkMma<double>::mma(NEG ? -av : av, bv, t);
    }
    return t;
}

__device__ __forceinline__ double lds_tile(const double *buf, int r, int c)
{
    return buf[((r >> 5) * 2 + (c >> 5)) * SBLK + (r & 31) * PLD + (c & 31)];
}

// a tile that lies entirely in the padding: K and X are the identity there
__device__ void trivial_tile(const SmallArgs &a, int i, int j)
{
    const int tid = threadIdx.x, np = a.np;
    const bool same128 = (i >> 1) == (j >> 1);
    for (int e = 0; e < 16; ++e) {
        const int idx = e * SM_THREADS + tid, r = idx >> 6, c = idx & 63;
        const double v = (i == j && r == c) ? 1.0 : 0.0;
        const size_t lo = (size_t)(ST * i + r) * np + ST * j + c, up = (size_t)(ST * j + r) * np + ST * i + c;
        a.K[lo] = v;
        a.X[lo] = v;
        if (i != j)
            a.X[up] = 0.0;
        a.XT[up] = (i == j && r == c) ? 1.0 : 0.0;  // (the transposed copy is only read on and above its diagonal)
        if (same128) {
            double *lb = a.linv + (size_t)(i >> 1) * TILE * TILE;
            lb[(size_t)(ST * (i & 1) + r) * TILE + ST * (j & 1) + c] = v;
            if (i == j && !(i & 1))
                lb[(size_t)r * TILE + ST + c] = 0.0;
        }
    }
    if (i == j && tid < ST) {
        a.d[ST * i + tid] = 1.0;
        a.dinv[ST * i + tid] = 1.0;
    }
}

// the tile's rows of the model's vectors, from the staging block (tiles of column 0 only)
__device__ void scatter_rows(const SmallArgs &a, int i)
{
    const int tid = threadIdx.x, np = a.np;
    if (tid < ST) {
        const int r = ST * i + tid;
        const double x = a.stage[r], y = a.stage[np + r], z = a.stage[2 * (size_t)np + r];
        a.d_x[r] = x, a.d_y[r] = y, a.d_z[r] = z;
        const bool in = r < a.n;
        a.t_x[r] = in ? x - a.cen[0] : 0.0;
        a.t_y[r] = in ? y - a.cen[1] : 0.0;
        a.t_z[r] = in ? z - a.cen[2] : 0.0;
        a.d_lab[r] = a.stage[3 * (size_t)np + r];
        const double s2 = a.stage[4 * (size_t)np + r];
        a.d_s2[r] = s2;
        a.t_s2[r] = s2;
    }
}
}  // namespace

template <int KID>
__global__ __launch_bounds__(SM_THREADS, 1) void small_factor_kernel(const SmallArgs *__restrict__ ap)
{
    const SmallArgs &a = *ap;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *bufA = sm, *bufB = sm + 4 * SBLK, *bufC = sm + 8 * SBLK;
    double *dvec = sm + 12 * SBLK;   // [64] D of the diagonal tile | scale vector of a staged operand
    double *dinvv = dvec + ST;       // [64] 1 / D
    double *rowp = dinvv + ST;       // [4][64] x y z s2 of the tile's rows (centred coordinates)
    double *colp = rowp + 4 * ST;    // [3][64] x y z of its columns
    __shared__ int s_ok;
    __shared__ double s_best[4];
    __shared__ int s_bi[4], s_bj[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1;
    const int np = a.np, n = a.n;
    int j = 0, rem = (int)blockIdx.x;
    while (rem >= a.nbt - j) {
        rem -= a.nbt - j;
        ++j;
    }
    const int i = j + rem;
    u64 *Ff = a.flags, *Xf = a.flags + a.ntiles, *Pf = a.flags + a.pre_idx;  // factor tiles, inverse tiles, handed-over sums (per row)
    if (blockIdx.x == 0 && tid < 16) {
        // state of the second launch: barrier counter, residual maxima (stream order: it starts after this grid has ended)
        if (tid == 0)
            st_flag(a.flags + a.bar_idx, 0);
        if (tid < 8)
            a.rmaxv[tid] = 0.0;
        if (tid < BLOB_META)
            a.d_meta[tid] = tid < 3 ? a.cen[tid] : (tid == 3 ? 1.0 : 0.0);
    }
    if (j == 0)
        scatter_rows(a, i);
    if (i >= a.nb) {
        trivial_tile(a, i, j);
        return;
    }
    // ---- the tile's entries of the kernel matrix, straight into the accumulator layout (gp_regressor.hpp:132-159) ----
    if (tid < ST) {
        const int r = ST * i + tid;
        const bool in = r < n;
        rowp[tid] = in ? a.stage[r] - a.cen[0] : 0.0;
        rowp[ST + tid] = in ? a.stage[np + r] - a.cen[1] : 0.0;
        rowp[2 * ST + tid] = in ? a.stage[2 * (size_t)np + r] - a.cen[2] : 0.0;
        rowp[3 * ST + tid] = a.stage[4 * (size_t)np + r];
    } else if (tid < 2 * ST) {
        const int t = tid - ST, c = ST * j + t;
        const bool in = c < n;
        colp[t] = in ? a.stage[c] - a.cen[0] : 0.0;
        colp[ST + t] = in ? a.stage[np + c] - a.cen[1] : 0.0;
        colp[2 * ST + t] = in ? a.stage[2 * (size_t)np + c] - a.cen[2] : 0.0;
    }
    __syncthreads();
    BlkAcc<double> acc;
    {
        const Cov<double> cov = a.cov;
        double best = -1.0;
        int bi = 0, bj = 0;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 32 * qi + 16 * i2 + BlkMma<double>::crow(lane, r), col = 32 * qj + 16 * j2 + (lane & 15);
                    const int gi = ST * i + row, gj = ST * j + col;
                    const double dx = rowp[row] - colp[col], dy = rowp[ST + row] - colp[ST + col],
                                 dz = rowp[2 * ST + row] - colp[2 * ST + col];
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    double kv = cov_k<double, KID>(cov, d2);
                    if (gi == gj)
                        kv += rowp[3 * ST + row];
                    if (gi < n && gj < n) {
                        if (d2 > best)
                            best = d2, bi = gi, bj = gj;
                    } else {
                        kv = gi == gj ? 1.0 : 0.0;  // identity on the padding
                    }
                    acc.t[i2][j2][r] = kv;
                }
        // the tile's largest squared distance (Model::R = Kpp.maxCoeff(), :135); the host takes the maximum over the tiles
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
            for (int w = 1; w < 4; ++w)
                if (s_best[w] > best)
                    best = s_best[w], bi = s_bi[w], bj = s_bj[w];
            a.tmax[blockIdx.x] = best;
            a.tij[2 * blockIdx.x] = bi;
            a.tij[2 * blockIdx.x + 1] = bj;
        }
    }
    // ---- A_ij -= sum_k (L_ik D_k) L_jk^T as the tiles of the earlier columns appear ----
    // The chain of the factorisation is diagonal tile j -> panel tile (j+1, j) -> diagonal tile j+1.  The diagonal tile
    // takes the middle link itself: tile (j+1, j) hands over its finished sums A_{j+1,j} BEFORE diagonal tile j is done
    // (through the unused tile (j, j+1) above the diagonal), and diagonal tile j+1, which has staged them while it waited,
    // forms W = A Xd_j^T, L = W D_j^-1 and its own last update as soon as Xd_j appears -- one publish, one poll and one
    // round trip to memory less per 64 columns on the critical path; the panel tile does the same product again for
    // everybody else.
    const int kend = i == j ? j - 1 : j;  // (the diagonal tile's last step is the one described above)
    SM_STAMP(0);
    for (int k = 0; k < kend; ++k) {
        if (!wait_tiles(Ff + tidx(i, k), i != j ? Ff + tidx(j, k) : nullptr, a, &s_ok))
            return;
        if (tid < ST)
            dvec[tid] = ld_cg(a.d + ST * k + tid);
        __syncthreads();
        stage_tile(bufA, a.K + (size_t)(ST * i) * np + ST * k, np, dvec);
        stage_tile(bufB, a.K + (size_t)(ST * j) * np + ST * k, np, nullptr);
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            acc.template mac_nt<true>(bufA + (qi * 2 + h) * SBLK, bufB + (qj * 2 + h) * SBLK, lane);
        __syncthreads();
    }
    SM_STAMP(1);
    if (i == j && j >= 1) {
        const int k = j - 1;
        if (!wait_tiles(Pf + i, nullptr, a, &s_ok))
            return;
        stage_tile(bufA, a.K + (size_t)(ST * k) * np + ST * i, np, nullptr);  // A_{i,k}, parked above the diagonal
        SM_STAMP(2);
        if (!wait_tiles(Ff + tidx(k, k), nullptr, a, &s_ok))
            return;
        SM_STAMP(3);
        if (tid < ST)
            dvec[tid] = ld_cg(a.dinv + ST * k + tid);
        stage_tile(bufB, a.X + (size_t)(ST * k) * np + ST * k, np, nullptr);
        __syncthreads();
        SM_STAMP(4);
        BlkAcc<double> w;
        w.zero();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            w.template mac_nt<false>(bufA + (qi * 2 + h) * SBLK, bufB + (qj * 2 + h) * SBLK, lane);
        __syncthreads();  // every wave has read A before L takes its place
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i2 + BlkMma<double>::crow(lane, r), col = 16 * j2 + (lane & 15);
                    const double wv = w.t[i2][j2][r];
                    bufC[(qi * 2 + qj) * SBLK + row * PLD + col] = wv;
                    bufA[(qi * 2 + qj) * SBLK + row * PLD + col] = wv * dvec[32 * qj + col];
                }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            acc.template mac_nt<true>(bufC + (qi * 2 + h) * SBLK, bufA + (qj * 2 + h) * SBLK, lane);
        __syncthreads();
        SM_STAMP(5);
    }
    if (i == j + 1) {
        // hand the finished sums to the diagonal tile of this row (see above)
        store_blk_cg(acc, 1.0, nullptr, a.K + (size_t)(ST * j + 32 * qi) * np + ST * i + 32 * qj, np, lane);
        publish_tile(Pf + i, a.epoch);
    }
    double *Ktile = a.K + (size_t)(ST * i) * np + ST * j;
    double *Xtile = a.X + (size_t)(ST * i) * np + ST * j;
    const bool same128 = (i >> 1) == (j >> 1);
    double *lb = a.linv + (size_t)(i >> 1) * TILE * TILE + (size_t)(ST * (i & 1)) * TILE + ST * (j & 1);
    if (i == j) {
        // ---- diagonal tile: LDL^T of the 64 x 64 block and the inverse of its L (gp_regressor.hpp:161-162) ----
        acc.store(1.0, bufC + (qi * 2 + qj) * SBLK, (double *)nullptr, 0, lane);
        __syncthreads();
        SM_STAMP(6);
        double *Lx0 = bufA, *W21 = bufA + SBLK, *L21 = bufA + 2 * SBLK, *Lx1 = bufA + 3 * SBLK;
        double *Xd0 = bufB, *T0 = bufB + SBLK, *X10 = bufB + 2 * SBLK, *Xd1 = bufB + 3 * SBLK;
        double *A21 = bufC + 2 * SBLK, *A22 = bufC + 3 * SBLK;
        // The two 32 x 32 sub-blocks are factorised (and their L inverted) by wave 0, one rank-1 MFMA update per column
        // (gpx_blk.hpp: 6.9 us each in fp64); the products between them are shared by the four waves, a 16 x 16 tile each.
        const int i2 = wave >> 1, j2 = wave & 1;
        const int trow = 16 * i2 + (lane >> 4), tcol = 16 * j2 + (lane & 15);  // element r of the lane's tile: row trow + 4 r
        double dv = 1.0;
        unsigned long long mneg = 0, mbad = 0;
        if (wave == 0) {
            subblock_ldl(bufC, Lx0, Xd0, lane, dv);
            if (lane < NB) {
                dvec[lane] = dv;
                dinvv[lane] = 1.0 / dv;
            }
            mneg = __ballot(lane < NB && dv < 0.0);
            mbad = __ballot(lane < NB && (!(fabs(dv) > 0.0) || !(fabs(dv) < pivot_huge(0.0))));
        }
        __syncthreads();
        SM_STAMP(16);
        {   // W21 = A21 X11^T, L21 = W21 D^-1
            acc16_t t = {0.0, 0.0, 0.0, 0.0};
            t = mma16<true, false>(A21, Xd0, i2, j2, lane, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                W21[(trow + 4 * r) * PLD + tcol] = t[r];
                L21[(trow + 4 * r) * PLD + tcol] = t[r] * dinvv[tcol];
            }
        }
        __syncthreads();
        SM_STAMP(17);
        {   // A22 -= W21 L21^T ; T0 = L21 Xd0 (for X10, off the chain of the second sub-block)
            acc16_t c, t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                c[r] = A22[(trow + 4 * r) * PLD + tcol];
            c = mma16<true, true>(W21, L21, i2, j2, lane, c);
            t = mma16<false, false>(L21, Xd0, i2, j2, lane, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                A22[(trow + 4 * r) * PLD + tcol] = c[r];
                T0[(trow + 4 * r) * PLD + tcol] = t[r];
            }
        }
        __syncthreads();
        SM_STAMP(18);
        if (wave == 0) {
            subblock_ldl(A22, Lx1, Xd1, lane, dv);
            if (lane < NB) {
                dvec[NB + lane] = dv;
                dinvv[NB + lane] = 1.0 / dv;
            }
            const unsigned long long mneg1 = __ballot(lane < NB && dv < 0.0);
            const unsigned long long mbad1 = __ballot(lane < NB && (!(fabs(dv) > 0.0) || !(fabs(dv) < pivot_huge(0.0))));
            if (lane == 0) {
                a.negcnt[i] = __builtin_popcountll(mneg) + __builtin_popcountll(mneg1);
                int bad = 0;
                if (mbad)
                    bad = ST * i + __builtin_ctzll(mbad) + 1;
                else if (mbad1)
                    bad = ST * i + NB + __builtin_ctzll(mbad1) + 1;
                a.badrow[i] = bad;
            }
        }
        __syncthreads();
        SM_STAMP(19);
        {   // X10 = -Xd1 (L21 Xd0)
            acc16_t t = {0.0, 0.0, 0.0, 0.0};
            t = mma16<false, true>(Xd1, T0, i2, j2, lane, t);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                X10[(trow + 4 * r) * PLD + tcol] = t[r];
        }
        __syncthreads();
        SM_STAMP(7);
        // what the tiles below wait for first: Xd (to the tile of X), L, D, 1/D; then the flag; the rest afterwards
        for (int e = 0; e < 16; ++e) {
            const int idx = e * SM_THREADS + tid, r = idx >> 6, c = idx & 63;
            double lv, xv;
            if (r == c)
                lv = dvec[r];
            else if (r < c)
                lv = 0.0;
            else if (r < NB)
                lv = Lx0[r * PLD + c];
            else if (c < NB)
                lv = L21[(r - NB) * PLD + c];
            else
                lv = Lx1[(r - NB) * PLD + (c - NB)];
            if (r < NB)
                xv = c < NB ? Xd0[r * PLD + c] : 0.0;
            else
                xv = c < NB ? X10[(r - NB) * PLD + c] : Xd1[(r - NB) * PLD + (c - NB)];
            st_cg(Ktile + (size_t)r * np + c, lv);
            st_cg(Xtile + (size_t)r * np + c, xv);
        }
        if (tid < ST) {
            st_cg(a.d + ST * i + tid, dvec[tid]);
            st_cg(a.dinv + ST * i + tid, dinvv[tid]);
        }
        publish_tile(Ff + tidx(i, i), a.epoch);
        SM_STAMP(8);
        for (int e = 0; e < 16; ++e) {
            const int idx = e * SM_THREADS + tid, r = idx >> 6, c = idx & 63;
            double xv;
            if (r < NB)
                xv = c < NB ? Xd0[r * PLD + c] : 0.0;
            else
                xv = c < NB ? X10[(r - NB) * PLD + c] : Xd1[(r - NB) * PLD + (c - NB)];
            lb[(size_t)r * TILE + c] = xv;
            if (!(i & 1))
                lb[(size_t)r * TILE + ST + c] = 0.0;  // upper-right quadrant of the 128 x 128 inverse block
        }
        for (int e = 0; e < 16; ++e) {  // transposed copy, coalesced along its rows
            const int idx = e * SM_THREADS + tid, c = idx >> 6, r = idx & 63;
            double xv;
            if (r < NB)
                xv = c < NB ? Xd0[r * PLD + c] : 0.0;
            else
                xv = c < NB ? X10[(r - NB) * PLD + c] : Xd1[(r - NB) * PLD + (c - NB)];
            a.XT[(size_t)(ST * i + c) * np + ST * i + r] = xv;
        }
        SM_STAMP(9);
        return;
    }
    // ---- tile below the diagonal: L_ij = (A_ij Xd_j^T) D_j^-1 (the panel solve as a product with the inverse block) ----
    SM_STAMP(10);
    if (!wait_tiles(Ff + tidx(j, j), nullptr, a, &s_ok))
        return;
    SM_STAMP(11);
    if (tid < ST)
        dvec[tid] = ld_cg(a.dinv + ST * j + tid);
    acc.store(1.0, bufA + (qi * 2 + qj) * SBLK, (double *)nullptr, 0, lane);
    stage_tile(bufB, a.X + (size_t)(ST * j) * np + ST * j, np, nullptr);
    __syncthreads();
    {
        BlkAcc<double> w;
        w.zero();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            w.template mac_nt<false>(bufA + (qi * 2 + h) * SBLK, bufB + (qj * 2 + h) * SBLK, lane);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    w.t[i2][j2][r] *= dvec[32 * qj + 16 * j2 + (lane & 15)];
        store_blk_cg(w, 1.0, bufC + (qi * 2 + qj) * SBLK, Ktile + (size_t)(32 * qi) * np + 32 * qj, np, lane);
    }
    SM_STAMP(12);
    publish_tile(Ff + tidx(i, j), a.epoch);  // (its barrier also orders the LDS stores of L before the products below)
    // ---- the tile of the inverse factor: X_ij = -Xd_i sum_{k = j}^{i-1} L_ik X_kj ----
    acc.zero();
#pragma unroll
    for (int h = 0; h < 2; ++h)
        acc.mac(bufC + (qi * 2 + h) * SBLK, bufB + (h * 2 + qj) * SBLK, lane);  // k = j: L_ij Xd_j
    for (int k = j + 1; k < i; ++k) {
        if (!wait_tiles(Ff + tidx(i, k), Xf + tidx(k, j), a, &s_ok))
            return;
        stage_tile(bufA, a.K + (size_t)(ST * i) * np + ST * k, np, nullptr);
        stage_tile(bufB, a.X + (size_t)(ST * k) * np + ST * j, np, nullptr);
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            acc.mac(bufA + (qi * 2 + h) * SBLK, bufB + (h * 2 + qj) * SBLK, lane);
    }
    SM_STAMP(13);
    if (!wait_tiles(Ff + tidx(i, i), nullptr, a, &s_ok))
        return;
    SM_STAMP(14);
    stage_tile(bufA, a.X + (size_t)(ST * i) * np + ST * i, np, nullptr);
    acc.store(1.0, bufC + (qi * 2 + qj) * SBLK, (double *)nullptr, 0, lane);
    __syncthreads();
    {
        BlkAcc<double> x;
        x.zero();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            x.mac(bufA + (qi * 2 + h) * SBLK, bufC + (h * 2 + qj) * SBLK, lane);
        store_blk_cg(x, -1.0, bufB + (qi * 2 + qj) * SBLK, Xtile + (size_t)(32 * qi) * np + 32 * qj, np, lane);
        if (same128)
            x.store(-1.0, (double *)nullptr, lb + (size_t)(32 * qi) * TILE + 32 * qj, TILE, lane);
    }
    __syncthreads();
    for (int e = 0; e < 16; ++e) {
        const int idx = e * SM_THREADS + tid, c = idx >> 6, r = idx & 63;
        a.XT[(size_t)(ST * j + c) * np + ST * i + r] = lds_tile(bufB, r, c);
        a.X[(size_t)(ST * j + c) * np + ST * i + r] = 0.0;  // the tile above the diagonal: structural zeros
        a.K[(size_t)(ST * j + c) * np + ST * i + r] = 0.0;
    }
    publish_tile(Xf + tidx(i, j), a.epoch);
    SM_STAMP(15);
}

// ---- alpha, refinement, row corrections ----------------------------------------------------------------------------
namespace {
// grid barrier number `index` (1, 2, ...) of a launch of gridDim.x workgroups.  The vectors that cross it are written
// with st_cg and read with ld_cg, so it needs no cache maintenance: X and X^T stay in the L2 from phase to phase.
__device__ __forceinline__ bool grid_barrier(const SmallArgs &a, int index, int *s_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *cnt = a.flags + a.bar_idx, *abortf = a.flags + a.abort_idx;
        __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 target = (u64)gridDim.x * (u64)index;
        bool ok = false;
        for (int s = 0; s < a.spin_limit; ++s) {
            if (ld_flag(cnt) >= target) {
                ok = true;
                break;
            }
            if ((s & 31) == 31 && ld_flag(abortf) == a.epoch)
                break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok)
            st_flag(abortf, a.epoch);
        *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();
    return ok;
}

__device__ __forceinline__ double wave_sum(double s)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off);
    return s;
}
}  // namespace

template <int KID>
__global__ __launch_bounds__(SM_THREADS) void small_alpha_kernel(const SmallArgs *__restrict__ ap)
{
    const SmallArgs &a = *ap;
    __shared__ int s_ok;
    __shared__ double vec[SMALL_CREATE_MAX_NP];  // the vector the current phase multiplies with
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = (int)blockIdx.x * 4 + (tid >> 6), nw = (int)gridDim.x * 4;
    const int np = a.np, n = a.n, nrows = a.nb * ST;
    const bool aborted = ld_flag(a.flags + a.abort_idx) == a.epoch;  // the factorisation gave up: nothing here is valid
    int bar = 0, ir = 0;
    double rmax_last = 0.0;
    auto stage_vec = [&](const double *src, int len) {
        for (int k = tid; k < len; k += SM_THREADS)
            vec[k] = ld_cg(src + k);
        __syncthreads();
    };
    if (!aborted) {
        for (int it = 0;; ++it) {
            // ---- t = X rhs, u = D^-1 t (L y = b and the scaling of LDLT::solve, gp_regressor.hpp:163) ----
            stage_vec(it == 0 ? a.d_lab : a.d_r, np);
            for (int r = gw; r < np; r += nw) {
                double s = 0.0;
                if (r < nrows) {
                    const double *row = a.X + (size_t)r * np;
                    for (int c = lane; c <= r; c += 64)
                        s = fma(row[c], vec[c], s);
                    s = wave_sum(s);
                } else {
                    s = vec[r];
                }
                if (lane == 0)
                    st_cg(a.u + r, s * a.dinv[r]);
                if (it == 0 && a.want_corr) {
                    // row-correction vectors of the variance fit: out[c][r] = sum_{l < n} X[r][l] b_c(p'_l) (gpx_internal.hpp)
                    double sc[VAR_NCORR];
#pragma unroll
                    for (int c = 0; c < VAR_NCORR; ++c)
                        sc[c] = 0.0;
                    if (r < nrows) {
                        const double *row = a.X + (size_t)r * np;
                        const int lend = min(n, r + 1);
                        for (int l = lane; l < lend; l += 64) {
                            const double xv = row[l];
                            double x = a.d_x[l] - a.cen[0], y = a.d_y[l] - a.cen[1], z = a.d_z[l] - a.cen[2];
                            if (!a.op64)
                                x = (double)(float)x, y = (double)(float)y, z = (double)(float)z;
                            const double r2 = x * x + y * y + z * z;
                            const double xx = xv * x, xy = xv * y, xz = xv * z, xr = xv * r2;
                            sc[0] += xv;
                            sc[1] += xx;
                            sc[2] += xy;
                            sc[3] += xz;
                            sc[4] = fma(xx, x, sc[4]);
                            sc[5] = fma(xy, y, sc[5]);
                            sc[6] = fma(xz, z, sc[6]);
                            sc[7] = fma(xx, y, sc[7]);
                            sc[8] = fma(xx, z, sc[8]);
                            sc[9] = fma(xy, z, sc[9]);
                            sc[10] = fma(xr, x, sc[10]);
                            sc[11] = fma(xr, y, sc[11]);
                            sc[12] = fma(xr, z, sc[12]);
                            sc[13] = fma(xr, r2, sc[13]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < VAR_NCORR; ++c) {
                        const double v = wave_sum(sc[c]);
                        if (lane == 0)
                            a.d_corr[(size_t)c * np + r] = v;
                    }
                }
            }
            if (!grid_barrier(a, ++bar, &s_ok))
                break;
            // ---- alpha += X^T u (L^T x = y) ----
            stage_vec(a.u, np);
            for (int c = gw; c < np; c += nw) {
                double s = 0.0;
                if (c < nrows) {
                    const double *row = a.XT + (size_t)c * np;
                    for (int r = c + lane; r < nrows; r += 64)
                        s = fma(row[r], vec[r], s);
                    s = wave_sum(s);
                } else {
                    s = vec[c];
                }
                if (lane == 0)
                    st_cg(a.d_alpha + c, (it == 0 ? 0.0 : ld_cg(a.d_alpha + c)) + s);
            }
            if (!grid_barrier(a, ++bar, &s_ok))
                break;
            // ---- r = y - K alpha in fp64, matrix-free from the fp64 points ----
            stage_vec(a.d_alpha, np);
            {
                const Cov<double> cov = a.cov;
                double wmax = 0.0;
                for (int r = gw; r < np; r += nw) {
                    double res = 0.0;
                    if (r < n) {
                        const double px = a.d_x[r], py = a.d_y[r], pz = a.d_z[r];
                        double s = 0.0;
                        for (int c = lane; c < n; c += 64) {
                            const double dx = px - a.d_x[c], dy = py - a.d_y[c], dz = pz - a.d_z[c];
                            const double d2 = dx * dx + dy * dy + dz * dz;
                            s = fma(cov_k<double, KID, MathFast>(cov, d2 + 1e-300), vec[c], s);
                        }
                        s = wave_sum(s);
                        res = a.d_lab[r] - s - a.d_s2[r] * vec[r];
                    }
                    if (lane == 0)
                        st_cg(a.d_r + r, res);
                    wmax = fmax(wmax, fabs(res));
                }
                if (lane == 0 && wmax > 0.0)
                    atomicMax((u64 *)&a.rmaxv[it], (u64)__double_as_longlong(wmax));
            }
            if (!grid_barrier(a, ++bar, &s_ok))
                break;
            ir = it;
            rmax_last = __longlong_as_double((long long)ld_flag((const u64 *)&a.rmaxv[it]));
            if (it >= a.ir_max)
                break;
            if (a.ir_adaptive && it >= 1 && !(rmax_last > a.ir_tol))
                break;
        }
    }
    // ---- the rest of the state and the result block (every workgroup has passed the same barriers) ----
    for (int r = (int)blockIdx.x * SM_THREADS + tid; r < np; r += (int)gridDim.x * SM_THREADS) {
        a.d_dinv64[r] = a.dinv[r];
        a.t_alpha[r] = ld_cg(a.d_alpha + r);
        a.res_d[r] = a.d[r];
    }
    if (blockIdx.x == 0 && tid == 0) {
        int bad = 0, neg = 0;
        for (int t = 0; t < a.nb; ++t) {
            if (!bad && a.badrow[t])
                bad = a.badrow[t];
            neg += a.negcnt[t];
        }
        double best = -2.0;
        int bt = 0;
        for (int t = 0; t < a.ntiles; ++t) {
            // (tiles in the padding wrote nothing: skip them by their row index)
            int jj = 0, rem = t;
            while (rem >= a.nbt - jj) {
                rem -= a.nbt - jj;
                ++jj;
            }
            if (jj + rem >= a.nb)
                continue;
            if (a.tmax[t] > best)
                best = a.tmax[t], bt = t;
        }
        const bool gave_up = ld_flag(a.flags + a.abort_idx) == a.epoch;
        SmallResult *res = a.res;
        res->info[0] = bad;
        res->info[1] = neg;
        res->info[2] = a.tij[2 * bt];
        res->info[3] = a.tij[2 * bt + 1];
        res->info[4] = 0;
        res->info[5] = gave_up ? 1 : 0;
        res->info[6] = res->info[7] = 0;
        res->rmax = rmax_last;
        res->ir_done = ir;
        res->d2max = best;
        for (int k = 0; k < 8; ++k)
            a.info[k] = k == 5 ? 0 : res->info[k];  // (info[5] is the substitution's give-up flag of the general chain)
        a.d_meta[4] = fmax(best, 0.0) / VAR_FIT_WDELTA_DIV;
    }
}

// ---- the fp32 state of a model that trained in fp64 (MIXED-style demotion), one launch ------------------------------
__global__ __launch_bounds__(SM_THREADS) void small_demote_kernel(const SmallArgs *__restrict__ ap)
{
    const SmallArgs &a = *ap;
    const SmallResult *res = a.res;
    // an indefinite kernel matrix keeps its fp64 state (build_model): nothing to round
    if (res->info[0] != 0 || res->info[1] != 0 || res->info[5] != 0)
        return;
    const size_t np = (size_t)a.np;
    const size_t tid = (size_t)blockIdx.x * SM_THREADS + threadIdx.x, nt = (size_t)gridDim.x * SM_THREADS;
    const size_t n64 = np * (5 + VAR_NCORR);
    const double *b0 = a.blob0;
    double *nb64 = (double *)a.nblob;
    for (size_t k = tid; k < n64; k += nt)
        nb64[k] = b0[k];
    float *tf = (float *)(nb64 + n64);
    const double *t4 = a.t_x;  // x' y' z' 1/D, contiguous
    for (size_t k = tid; k < 4 * np; k += nt)
        tf[k] = (float)t4[k];
    double *nmeta = (double *)(tf + 4 * np);
    for (size_t k = tid; k < BLOB_META; k += nt)
        nmeta[k] = a.d_meta[k];
    for (size_t k = tid; k < np * np; k += nt)
        a.nX[k] = (float)a.X[k];
}

void small_create_init()
{
    static PerDeviceOnce once;
    once.run([] {
#define GPX_SM_ATTR(KID)                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&small_factor_kernel<KID>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SM_LDS_DOUBLES * sizeof(double)))
        GPX_SM_ATTR(GPX_KERNEL_GAUSSIAN);
        GPX_SM_ATTR(GPX_KERNEL_THINPLATE);
        GPX_SM_ATTR(GPX_KERNEL_MATERN32);
        GPX_SM_ATTR(GPX_KERNEL_MATERN52);
#undef GPX_SM_ATTR
    });
}

u64 small_create_epoch()
{
    static std::atomic<u64> ctr{1};
    return 0x5a17c0de00000000ull | (ctr.fetch_add(1) & 0xffffffffull);
}

void launch_small_create(int kernel_id, const SmallArgs &a, const SmallArgs *d_args, bool demote, hipStream_t st,
                         hipEvent_t ev_factor, hipEvent_t ev_solve)
{
    const size_t lds = SM_LDS_DOUBLES * sizeof(double);
    GPX_DISPATCH_KID(kernel_id, hipLaunchKernelGGL((small_factor_kernel<KID>), dim3(a.ntiles), dim3(SM_THREADS), lds, st, d_args));
    if (ev_factor)
        (void)hipEventRecord(ev_factor, st);
    const int g = std::max(16, std::min(128, a.np / 8));
    GPX_DISPATCH_KID(kernel_id, hipLaunchKernelGGL((small_alpha_kernel<KID>), dim3(g), dim3(SM_THREADS), 0, st, d_args));
    if (ev_solve)
        (void)hipEventRecord(ev_solve, st);
    if (demote)
        hipLaunchKernelGGL(small_demote_kernel, dim3(std::max(8, a.np * a.np / (SM_THREADS * 16))), dim3(SM_THREADS), 0, st, d_args);
}

}  // namespace gpx
