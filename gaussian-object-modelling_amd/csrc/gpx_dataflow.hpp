// gpx_dataflow.hpp -- kernel matrix -> LDL^T [-> inverse factor] as ONE launch: a dataflow over 64 x 64 tiles
// (replaces buildEuclideanDistanceMatrix + the kernel loop + Eigen::LDLT::compute, reference gp_regressor.hpp:132-162).
//
// A workgroup of four waves per lower tile (i, j), enumerated column by column:
//   * it forms its entries of K from the points straight into MFMA accumulators (one 32 x 32 quadrant per wave),
//   * subtracts (L_ik D_k) L_jk^T for k < j as those tiles are published -- the operands of step k + 1 are in flight while the
//     matrix cores work on step k,
//   * a DIAGONAL tile then factorises: two 32 x 32 sub-blocks by rank-1 MFMA updates on wave 0 (gpx_blk.hpp: L, D and L^-1 in one
//     sweep), the products between them shared by the four waves; it publishes L, D, 1/D and the inverse Xd of its L;
//   * a tile BELOW the diagonal solves against that inverse, L_ij = (A_ij Xd_j^T) D_j^-1, and publishes L_ij;
//   * the chain diagonal j -> tile (j+1, j) -> diagonal j+1 is shortened by one publish / poll / memory round trip: tile
//     (j+1, j) hands its finished sums to diagonal tile j+1 BEFORE diagonal j is done (through the unused tile above the
//     diagonal), and diagonal tile j+1 forms that panel solve and its own last update itself as soon as Xd_j appears;
//   * FULL (small models, gpx_small.hip): the same workgroup afterwards forms its tile of X = L^-1,
//     X_ij = -Xd_i sum_{k=j}^{i-1} L_ik X_kj, which tracks the factorisation one product behind, plus the transposed copy;
//     otherwise (mid-size models, the general chain continues with the substitution and the recursive-doubling inverse) only
//     the 128 x 128 inverse diagonal blocks `linv` the rest of the library works with are completed.
//
// Tiles travel through global memory behind one flag per tile.  The producer stores them at agent scope (sc1: written
// through to the memory side, where the XCDs' L2s meet), waits for the acknowledgements and raises the flag; a consumer polls
// the flag at agent scope and then reads the tile with ORDINARY loads: a tile's lines are never touched before its flag is up
// (no speculative reads on this hardware, tiles are whole cache lines: 64 elements = 256 / 512 bytes, rows aligned), and the
// L2s start every kernel clean, so no stale copy can exist -- no L2 write-back, no invalidation, and a tile that 30 workgroups
// of an XCD read comes from memory once.
//
// Every wait has a time budget; a workgroup that runs out of patience raises the abort flag, all others see it in their
// polls and leave, and the host falls back to the launch chain (as tri_solve_kernel does).  Factor jobs only wait for
// workgroups with a LOWER index (in-order dispatch then guarantees progress however few are resident); the inverse jobs also
// wait for later ones: FULL needs the whole grid resident (<= 136 workgroups), the mid-size form only the next column's
// first workgroup.
#pragma once
#include <type_traits>
#include "gpx_blk.hpp"
#include "gpx_cov.hpp"
#include "gpx_small.hpp"

namespace gpx {
namespace dataflow {

constexpr int ST = SMALL_TILE;  // 64
constexpr int SBLK = NB * PLD;  // one 32 x 32 block in LDS (elements), row stride PLD = 33: the layout of gpx_blk.hpp's products
// The operands of the update loop -- where the flops are -- use a second layout of the same buffers: row stride WLD = 36
// elements, so that a lane's eight consecutive k of a row are 16-byte aligned vector reads (ds_read_b128; at most 2-way bank
// conflicts).  With the 33-element rows and one element per read the LDS reads of the four waves took as long as their MFMAs.
constexpr int WLD = 36, WBLK = NB * WLD;
constexpr int DF_THREADS = 256;
constexpr int DF_LDS_PROG = 12 * WBLK + 2 * ST + 4 * ST + 3 * ST;  // step counter of the two-wave sub-block factorisation
constexpr int DF_LDS_ELEMS = DF_LDS_PROG + 2;
typedef unsigned long long u64;

#ifdef SM_TIMING
#define SM_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0 && f.dbg)                                                                \
            f.dbg[(size_t)blockIdx.x * SMALL_DBG_STAMPS + (k)] = wall_clock64();                      \
    } while (0)
#else
#define SM_STAMP(k)
#endif

__device__ __forceinline__ u64 ld_flag(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ double ld_cg(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_cg(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_cg(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// arguments of the factorisation proper, in the working type T
template <typename T>
struct FactorArgs {
    int n = 0, np = 0, nbt = 0 /* np / 64 */, nb = 0 /* tile rows that hold training points */, ntiles = 0;
    Cov<T> cov{};
    T *K = nullptr, *linv = nullptr, *d = nullptr, *dinv = nullptr;
    const T *px = nullptr, *py = nullptr, *pz = nullptr, *ps2 = nullptr;  // !FULL: centred points and sigma2 in T, np long
    u64 *flags = nullptr;  // [ntiles] factor tiles, [ntiles] inverse tiles, abort, (barrier counter), [nbt] handed-over sums
    u64 epoch = 0;
    long long wait_ticks = 2000000;  // 20 ms of the 100 MHz wall clock: after that a wait gives up (wait_budget_ticks, gpx_internal.hpp)
    int abort_idx = 0, pre_idx = 0;
    double *tmax = nullptr;
    int *tij = nullptr, *negcnt = nullptr, *badrow = nullptr;
    u64 *dbg = nullptr;
};

// thread 0 of the workgroup: wait until *f == want (false: the abort flag went up, or the time budget of the wait ran out and
// this call raised it).  The budget is CLOCK time (the constant 100 MHz counter), looked at every 32 polls: a grid that cannot
// make progress costs the caller milliseconds before the launch chain takes over (round 5 counted 2^20 polls: about a second).
__device__ inline bool poll_flag(const u64 *f, u64 want, u64 *abortf, long long budget_ticks)
{
    const u64 t0 = wall_clock64();
    for (int s = 0;; ++s) {
        if (ld_flag(f) == want)
            return true;
        if (budget_ticks <= 0)  // (tests: one look, then give up)
            break;
        if ((s & 31) == 31) {
            if (ld_flag(abortf) == want)
                return false;
            if ((long long)(wall_clock64() - t0) > budget_ticks)
                break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    st_flag(abortf, want);
    return false;
}

// all threads: wait for one or two tile flags
template <typename T>
__device__ __forceinline__ bool wait_tiles(const u64 *f1, const u64 *f2, const FactorArgs<T> &f, int *s_ok)
{
    if (threadIdx.x == 0) {
        bool ok = poll_flag(f1, f.epoch, f.flags + f.abort_idx, f.wait_ticks);
        if (ok && f2)
            ok = poll_flag(f2, f.epoch, f.flags + f.abort_idx, f.wait_ticks);
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

// 64 x 64 tile at g (leading dimension ld): element e of thread t is (row 4 e + (t >> 6), column t & 63) -- a thread's
// column is fixed, a wave reads one row of 64 contiguous elements per instruction
template <typename T>
__device__ __forceinline__ void tile_load(T (&v)[16], const T *g, long ld)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        v[e] = g[(size_t)(4 * e + (tid >> 6)) * ld + (tid & 63)];
}
// ... into four 32 x 32 LDS blocks [(r >> 5) * 2 + (c >> 5)], the thread's column scaled by s
template <typename T>
__device__ __forceinline__ void tile_to_lds(T *buf, const T (&v)[16], T s)
{
    const int tid = threadIdx.x, c = tid & 63;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = 4 * e + (tid >> 6);
        buf[((r >> 5) * 2 + (c >> 5)) * SBLK + (r & 31) * PLD + (c & 31)] = v[e] * s;
    }
}
// the same tile in the WIDE layout (update loop only): four 32 x 32 blocks of row stride WLD
template <typename T>
__device__ __forceinline__ void tile_to_lds_wide(T *buf, const T (&v)[16], T s)
{
    const int tid = threadIdx.x, c = tid & 63;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = 4 * e + (tid >> 6);
        buf[((r >> 5) * 2 + (c >> 5)) * WBLK + (r & 31) * WLD + (c & 31)] = v[e] * s;
    }
}
// C (the wave's 32 x 32 quadrant) += A B^T for one pair of 32 x 32 blocks in the wide layout (A [row][k], B [col][k]).  The
// summation index of MFMA step kk in lane group kq is k = 8 kq + kk -- any bijection serves, as long as A and B agree -- so a
// lane's eight values of a row are contiguous: two (fp32) / four (fp64) 16-byte reads per row instead of eight scalar ones.
template <typename T>
__device__ __forceinline__ void mac_nt_wide(BlkAcc<T> &c, const T *Ab, const T *Bb, int lane)
{
    typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
    constexpr int EPV = 16 / sizeof(T);      // elements per 16-byte vector
    constexpr int KH = sizeof(T) == 8 ? 2 : 8;  // k values in flight: fp64 takes the eight in four steps (register budget)
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += KH) {
        T a0[KH], a1[KH], b0[KH], b1[KH];
        auto loadk = [&](T (&dst)[KH], const T *src) {
#pragma unroll
            for (int q = 0; q < KH / EPV; ++q) {
                const vec_t v = reinterpret_cast<const vec_t *>(src + k0)[q];
#pragma unroll
                for (int w = 0; w < EPV; ++w)
                    dst[q * EPV + w] = v[w];
            }
        };
        loadk(a0, Ab + i * WLD + 8 * kq);
        loadk(a1, Ab + (16 + i) * WLD + 8 * kq);
        loadk(b0, Bb + i * WLD + 8 * kq);
        loadk(b1, Bb + (16 + i) * WLD + 8 * kq);
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            c.t[0][0] = BlkMma<T>::mma(a0[kk], b0[kk], c.t[0][0]);
            c.t[0][1] = BlkMma<T>::mma(a0[kk], b1[kk], c.t[0][1]);
            c.t[1][0] = BlkMma<T>::mma(a1[kk], b0[kk], c.t[1][0]);
            c.t[1][1] = BlkMma<T>::mma(a1[kk], b1[kk], c.t[1][1]);
        }
        if constexpr (sizeof(T) == 8)  // fp64: keep the scheduler from hoisting the next half's operands above these MFMAs (the
            __builtin_amdgcn_sched_barrier(0);  // 128 x 128 form runs two waves per SIMD on 256 registers each and would spill)
    }
}

template <typename T>
__device__ __forceinline__ void stage_tile(T *buf, const T *g, long ld)
{
    T v[16];
    tile_load(v, g, ld);
    tile_to_lds(buf, v, T(1));
}
template <typename T>
__device__ __forceinline__ T lds_tile(const T *buf, int r, int c)
{
    return buf[((r >> 5) * 2 + (c >> 5)) * SBLK + (r & 31) * PLD + (c & 31)];
}

// sign * (32 x 32 accumulator block) -> LDS block (may be null) and / or global with write-through stores (may be null)
template <typename T>
__device__ __forceinline__ void store_blk_cg(const BlkAcc<T> &b, T sign, T *lds, T *g, long ldg, int lane)
{
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i2 + BlkMma<T>::crow(lane, r), col = 16 * j2 + (lane & 15);
                const T v = sign * b.t[i2][j2][r];
                if (lds)
                    lds[row * PLD + col] = v;
                if (g)
                    st_cg(g + (size_t)row * ldg + col, v);
            }
}

// one 16 x 16 tile of a product of two 32 x 32 LDS blocks (K = 32) on one wave: rows 16 i2.. of A, columns 16 j2.. of the
// result; NT: B is [n][k] (A B^T), else [k][n].  The four waves of the workgroup share a 32 x 32 product this way.
template <typename T, bool NT, bool NEG>
__device__ __forceinline__ typename BlkMma<T>::acc_t mma16(const T *Ab, const T *Bb, int i2, int j2, int lane,
                                                            typename BlkMma<T>::acc_t t)
{
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const int k = 4 * kk + kq;
        const T av = Ab[(16 * i2 + i) * PLD + k];
        const T bv = NT ? Bb[(16 * j2 + i) * PLD + k] : Bb[k * PLD + 16 * j2 + i];
        t = BlkMma<T>::mma(NEG ? -av : av, bv, t);
    }
    return t;
}

// a tile that lies entirely in the padding: K and X are the identity there
template <typename T, bool FULL>
__device__ inline void trivial_tile(const FactorArgs<T> &f, const SmallArgs *sa, int i, int j)
{
    const int tid = threadIdx.x, np = f.np;
    const bool same128 = (i >> 1) == (j >> 1);
    for (int e = 0; e < 16; ++e) {
        const int idx = e * DF_THREADS + tid, r = idx >> 6, c = idx & 63;
        const T v = (i == j && r == c) ? T(1) : T(0);
        const size_t lo = (size_t)(ST * i + r) * np + ST * j + c, up = (size_t)(ST * j + r) * np + ST * i + c;
        f.K[lo] = v;
        if constexpr (FULL) {
            sa->X[lo] = v;
            if (i != j)
                sa->X[up] = 0.0;
            sa->XT[up] = v;  // (the transposed copy is only read on and above its diagonal)
        }
        if (same128) {
            T *lb = f.linv + (size_t)(i >> 1) * TILE * TILE;
            lb[(size_t)(ST * (i & 1) + r) * TILE + ST * (j & 1) + c] = v;
            if (i == j && !(i & 1))
                lb[(size_t)r * TILE + ST + c] = T(0);
        }
    }
    if (i == j && tid < ST) {
        f.d[ST * i + tid] = T(1);
        f.dinv[ST * i + tid] = T(1);
    }
}

// sm: DF_LDS_ELEMS elements of T.  sa: the small-model extras (FULL only).
template <typename T, int KID, bool FULL>
__device__ __forceinline__ void factor_tile(const FactorArgs<T> &f, const SmallArgs *sa, T *sm)
{
    typedef typename BlkMma<T>::acc_t acc16_t;
    T *bufA = sm, *bufB = sm + 4 * WBLK, *bufC = sm + 8 * WBLK;  // (sized for the wide layout; the other phases use 33-element rows)
    T *dvec = sm + 12 * WBLK;   // [64] D of the diagonal tile | scale vector of a staged operand
    T *dinvv = dvec + ST;       // [64] 1 / D
    T *rowp = dinvv + ST;       // [4][64] x y z s2 of the tile's rows (centred coordinates)
    T *colp = rowp + 4 * ST;    // [3][64] x y z of its columns
    __shared__ int s_ok, s_next;
    __shared__ double s_best[4];
    __shared__ int s_bi[4], s_bj[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = wave >> 1, qj = wave & 1;
    const int np = f.np, n = f.n;
    int j = 0, rem = (int)blockIdx.x;
    while (rem >= f.nbt - j) {
        rem -= f.nbt - j;
        ++j;
    }
    const int i = j + rem;
    u64 *Ff = f.flags, *Xf = f.flags + f.ntiles, *Pf = f.flags + f.pre_idx;  // factor tiles, inverse tiles, handed-over sums (per row)
    if (i >= f.nb) {
        trivial_tile<T, FULL>(f, sa, i, j);
        return;
    }
    // ---- the tile's entries of the kernel matrix, straight into the accumulator layout (gp_regressor.hpp:132-159) ----
    if (tid < ST) {
        const int r = ST * i + tid;
        const bool in = r < n;
        if constexpr (FULL) {
            rowp[tid] = in ? sa->stage[r] - sa->cen[0] : 0.0;
            rowp[ST + tid] = in ? sa->stage[np + r] - sa->cen[1] : 0.0;
            rowp[2 * ST + tid] = in ? sa->stage[2 * (size_t)np + r] - sa->cen[2] : 0.0;
            rowp[3 * ST + tid] = sa->stage[4 * (size_t)np + r];
        } else {
            rowp[tid] = f.px[r], rowp[ST + tid] = f.py[r], rowp[2 * ST + tid] = f.pz[r], rowp[3 * ST + tid] = f.ps2[r];
        }
    } else if (tid < 2 * ST) {
        const int t = tid - ST, c = ST * j + t;
        const bool in = c < n;
        if constexpr (FULL) {
            colp[t] = in ? sa->stage[c] - sa->cen[0] : 0.0;
            colp[ST + t] = in ? sa->stage[np + c] - sa->cen[1] : 0.0;
            colp[2 * ST + t] = in ? sa->stage[2 * (size_t)np + c] - sa->cen[2] : 0.0;
        } else {
            colp[t] = f.px[c], colp[ST + t] = f.py[c], colp[2 * ST + t] = f.pz[c];
        }
    }
    __syncthreads();
    BlkAcc<T> acc;
    {
        const Cov<T> cov = f.cov;
        double best = -1.0;
        int bi = 0, bj = 0;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 32 * qi + 16 * i2 + BlkMma<T>::crow(lane, r), col = 32 * qj + 16 * j2 + (lane & 15);
                    const int gi = ST * i + row, gj = ST * j + col;
                    const T dx = rowp[row] - colp[col], dy = rowp[ST + row] - colp[ST + col],
                            dz = rowp[2 * ST + row] - colp[2 * ST + col];
                    const T d2 = dx * dx + dy * dy + dz * dz;
                    T kv = cov_k<T, KID>(cov, d2);
                    if (gi == gj)
                        kv += rowp[3 * ST + row];
                    if (gi < n && gj < n) {
                        if ((double)d2 > best)
                            best = (double)d2, bi = gi, bj = gj;
                    } else {
                        kv = gi == gj ? T(1) : T(0);  // identity on the padding
                    }
                    acc.t[i2][j2][r] = kv;
                }
        // the tile's largest squared distance (Model::R = Kpp.maxCoeff(), :135); the maximum over the tiles is taken afterwards
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
            f.tmax[blockIdx.x] = best;
            f.tij[2 * blockIdx.x] = bi;
            f.tij[2 * blockIdx.x + 1] = bj;
        }
    }
    // ---- A_ij -= sum_k (L_ik D_k) L_jk^T as the tiles of the earlier columns appear ----
    const int kend = i == j ? j - 1 : j;  // (a diagonal tile's last step is the hand-over described at the top)
    SM_STAMP(0);
    if (kend > 0) {
        // fp32: the sum over k runs in CHUNKS of four tiles (256 columns, the panel width of the blocked chain) that start from
        // zero and are then added to the running value -- accumulated straight onto K_ij every one of up to 16384 fp32
        // roundings would be relative to the O(1) running value instead of to the small partial sum (measured at N = 16384:
        // variance error 1.8e-5 of max|v| against 1.2e-6 for the chain, and one more refinement step of alpha); with the chunks
        // the running value is rounded once per 256 columns, exactly as the chain's in-place updates round it
        constexpr bool CHUNKED = sizeof(T) == 4;
        T master[CHUNKED ? 16 : 1];
        if constexpr (CHUNKED) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                master[q] = acc.t[q >> 3][(q >> 2) & 1][q & 3];
            acc.zero();
        }
        T va[16], vb[16];
        T dk = T(1);
        // Flags are polled in BATCHES by wave 0 -- lane l looks at step base + l -- and one iteration before the answer is
        // needed: a tile of a late column finds a hundred earlier columns long finished, and one flag round trip to the
        // memory side (1-2 us, as long as a whole step) per step would leave the matrix cores waiting at every barrier.
        int kready = 1;  // steps [0, kready) are known to be published
        bool poll_out = false;
        int poll_base = 0;
        u64 fa = 0, fb = 0;
        auto issue = [&](int k) {
            tile_load(va, f.K + (size_t)(ST * i) * np + ST * k, np);
            tile_load(vb, f.K + (size_t)(ST * j) * np + ST * k, np);
            dk = f.d[ST * k + (tid & 63)];
        };
        auto poll_issue = [&](int base) {
            if (wave == 0) {
                const int kk = base + lane;
                fa = fb = 0;
                if (kk < kend) {
                    fa = ld_flag(Ff + tidx(i, kk));
                    fb = i != j ? ld_flag(Ff + tidx(j, kk)) : f.epoch;
                }
            }
            poll_out = true, poll_base = base;
        };
        if (!wait_tiles(Ff + tidx(i, 0), i != j ? Ff + tidx(j, 0) : nullptr, f, &s_ok))
            return;
        issue(0);
        if (kend > 1)
            poll_issue(1);
        for (int k = 0; k < kend; ++k) {
            tile_to_lds_wide(bufA, va, -dk);  // (the sign of the update goes into the operand)
            tile_to_lds_wide(bufB, vb, T(1));
            const bool consumed = poll_out;
            if (poll_out) {  // the answer of the batch issued an iteration (or more) ago
                if (wave == 0) {
                    const u64 m = __ballot(fa == f.epoch && fb == f.epoch);
                    const int cnt = ~m ? __builtin_ctzll(~m) : 64;
                    if (lane == 0)
                        s_next = poll_base + cnt;
                }
                poll_out = false;
            }
            __syncthreads();
            if (consumed)
                kready = max(kready, s_next);
            const bool ready = k + 1 < kend && k + 1 < kready;
            if (ready)
                issue(k + 1);  // in flight while the matrix cores work on step k
            if (k + 2 < kend && kready - (k + 1) < 8)
                poll_issue(max(kready, k + 1));
#pragma unroll
            for (int h = 0; h < 2; ++h)
                mac_nt_wide<T>(acc, bufA + (qi * 2 + h) * WBLK, bufB + (qj * 2 + h) * WBLK, lane);
            if constexpr (CHUNKED) {
                if ((k & 3) == 3 || k + 1 == kend) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        master[q] += acc.t[q >> 3][(q >> 2) & 1][q & 3];
                    if (k + 1 < kend)
                        acc.zero();
                    else {
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            acc.t[q >> 3][(q >> 2) & 1][q & 3] = master[q];
                    }
                }
            }
            __syncthreads();
            if (!ready && k + 1 < kend) {
                if (!wait_tiles(Ff + tidx(i, k + 1), i != j ? Ff + tidx(j, k + 1) : nullptr, f, &s_ok))
                    return;
                kready = max(kready, k + 2);
                issue(k + 1);
            }
        }
    }
    SM_STAMP(1);
    if (i == j && j >= 1) {
        const int k = j - 1;
        if (!wait_tiles(Pf + i, nullptr, f, &s_ok))
            return;
        stage_tile(bufA, f.K + (size_t)(ST * k) * np + ST * i, np);  // A_{i,k}, parked above the diagonal
        SM_STAMP(2);
        if (!wait_tiles(Ff + tidx(k, k), nullptr, f, &s_ok))
            return;
        SM_STAMP(3);
        if (tid < ST)
            dvec[tid] = f.dinv[ST * k + tid];
        stage_tile(bufB, f.linv + (size_t)(k >> 1) * TILE * TILE + (size_t)(ST * (k & 1)) * TILE + ST * (k & 1), TILE);  // Xd_k
        __syncthreads();
        SM_STAMP(4);
        BlkAcc<T> w;
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
                    const int row = 16 * i2 + BlkMma<T>::crow(lane, r), col = 16 * j2 + (lane & 15);
                    const T wv = w.t[i2][j2][r];
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
        // hand the finished sums to the diagonal tile of this row
        store_blk_cg<T>(acc, T(1), nullptr, f.K + (size_t)(ST * j + 32 * qi) * np + ST * i + 32 * qj, np, lane);
        publish_tile(Pf + i, f.epoch);
    }
    T *Ktile = f.K + (size_t)(ST * i) * np + ST * j;
    const bool same128 = (i >> 1) == (j >> 1);
    T *lb = f.linv + (size_t)(i >> 1) * TILE * TILE + (size_t)(ST * (i & 1)) * TILE + ST * (j & 1);  // the tile's quadrant of linv
    if (i == j) {
        // ---- diagonal tile: LDL^T of the 64 x 64 block and the inverse of its L (gp_regressor.hpp:161-162) ----
        acc.store(T(1), bufC + (qi * 2 + qj) * SBLK, (T *)nullptr, 0, lane);
        volatile int *prog = reinterpret_cast<volatile int *>(sm + DF_LDS_PROG);
        if (tid == 0)
            *(volatile GPX_LDS(int) *)prog = 0;  // (a ds_write: through the generic pointer it is a flat store with system scope)
        __syncthreads();
        SM_STAMP(6);
        T *Lx0 = bufA, *W21 = bufA + SBLK, *L21 = bufA + 2 * SBLK, *Lx1 = bufA + 3 * SBLK;
        T *Xd0 = bufB, *T0 = bufB + SBLK, *X10 = bufB + 2 * SBLK, *Xd1 = bufB + 3 * SBLK;
        T *A21 = bufC + 2 * SBLK, *A22 = bufC + 3 * SBLK;
        // The two 32 x 32 sub-blocks are factorised by wave 0, one rank-1 MFMA update per column, and their L inverted by wave 1
        // one step behind (gpx_blk.hpp, subblock_ldl_pair); the products between them are shared by the four waves, a
        // 16 x 16 tile each.
        const int i2 = wave >> 1, j2 = wave & 1;
        const int tcol = 16 * j2 + (lane & 15);
        T dv = T(1);
        unsigned long long mneg = 0, mbad = 0;
        if (wave <= 1)
            subblock_ldl_pair(bufC, Lx0, Xd0, lane, wave, dv, prog, 0);
        if (wave == 0) {
            if (lane < NB) {
                dvec[lane] = dv;
                dinvv[lane] = T(1) / dv;
            }
            mneg = __ballot(lane < NB && dv < T(0));
            mbad = __ballot(lane < NB && (!(fabs(dv) > T(0)) || !(fabs(dv) < pivot_huge(T(0)))));
        }
        __syncthreads();
        SM_STAMP(16);
        {   // W21 = A21 X11^T, L21 = W21 D^-1
            acc16_t t = {T(0), T(0), T(0), T(0)};
            t = mma16<T, true, false>(A21, Xd0, i2, j2, lane, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int trow = 16 * i2 + BlkMma<T>::crow(lane, r);
                W21[trow * PLD + tcol] = t[r];
                L21[trow * PLD + tcol] = t[r] * dinvv[tcol];
            }
        }
        __syncthreads();
        SM_STAMP(17);
        {   // A22 -= W21 L21^T ; T0 = L21 Xd0 (for X10, off the chain of the second sub-block)
            acc16_t c, t = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                c[r] = A22[(16 * i2 + BlkMma<T>::crow(lane, r)) * PLD + tcol];
            c = mma16<T, true, true>(W21, L21, i2, j2, lane, c);
            t = mma16<T, false, false>(L21, Xd0, i2, j2, lane, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int trow = 16 * i2 + BlkMma<T>::crow(lane, r);
                A22[trow * PLD + tcol] = c[r];
                T0[trow * PLD + tcol] = t[r];
            }
        }
        __syncthreads();
        SM_STAMP(18);
        if (wave <= 1)
            subblock_ldl_pair(A22, Lx1, Xd1, lane, wave, dv, prog, NB);
        if (wave == 0) {
            if (lane < NB) {
                dvec[NB + lane] = dv;
                dinvv[NB + lane] = T(1) / dv;
            }
            const unsigned long long mneg1 = __ballot(lane < NB && dv < T(0));
            const unsigned long long mbad1 = __ballot(lane < NB && (!(fabs(dv) > T(0)) || !(fabs(dv) < pivot_huge(T(0)))));
            if (lane == 0) {
                f.negcnt[i] = __builtin_popcountll(mneg) + __builtin_popcountll(mneg1);
                int bad = 0;
                if (mbad)
                    bad = ST * i + __builtin_ctzll(mbad) + 1;
                else if (mbad1)
                    bad = ST * i + NB + __builtin_ctzll(mbad1) + 1;
                f.badrow[i] = bad;
            }
        }
        __syncthreads();
        SM_STAMP(19);
        {   // X10 = -Xd1 (L21 Xd0)
            acc16_t t = {T(0), T(0), T(0), T(0)};
            t = mma16<T, false, true>(Xd1, T0, i2, j2, lane, t);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                X10[(16 * i2 + BlkMma<T>::crow(lane, r)) * PLD + tcol] = t[r];
        }
        __syncthreads();
        SM_STAMP(7);
        // what the tiles below wait for: L, D, 1/D and Xd (in its quadrant of the 128 x 128 inverse block); then the flag
        auto xd_at = [&](int r, int c) -> T {
            if (r < NB)
                return c < NB ? Xd0[r * PLD + c] : T(0);
            return c < NB ? X10[(r - NB) * PLD + c] : Xd1[(r - NB) * PLD + (c - NB)];
        };
        for (int e = 0; e < 16; ++e) {
            const int idx = e * DF_THREADS + tid, r = idx >> 6, c = idx & 63;
            T lv;
            if (r == c)
                lv = dvec[r];
            else if (r < c)
                lv = T(0);
            else if (r < NB)
                lv = Lx0[r * PLD + c];
            else if (c < NB)
                lv = L21[(r - NB) * PLD + c];
            else
                lv = Lx1[(r - NB) * PLD + (c - NB)];
            st_cg(Ktile + (size_t)r * np + c, lv);
            st_cg(lb + (size_t)r * TILE + c, xd_at(r, c));
        }
        if (tid < ST) {
            st_cg(f.d + ST * i + tid, dvec[tid]);
            st_cg(f.dinv + ST * i + tid, dinvv[tid]);
        }
        publish_tile(Ff + tidx(i, i), f.epoch);
        SM_STAMP(8);
        for (int e = 0; e < 16; ++e) {
            const int idx = e * DF_THREADS + tid, r = idx >> 6, c = idx & 63;
            if (!(i & 1))
                lb[(size_t)r * TILE + ST + c] = T(0);  // upper-right quadrant of the 128 x 128 inverse block
            if constexpr (FULL) {
                sa->X[(size_t)(ST * i + r) * np + ST * i + c] = xd_at(r, c);
                sa->XT[(size_t)(ST * i + r) * np + ST * i + c] = xd_at(c, r);  // transposed copy (LDS column reads: conflict-free stride)
            }
        }
        SM_STAMP(9);
        return;
    }
    // ---- tile below the diagonal: L_ij = (A_ij Xd_j^T) D_j^-1 (the panel solve as a product with the inverse block) ----
    SM_STAMP(10);
    if (!wait_tiles(Ff + tidx(j, j), nullptr, f, &s_ok))
        return;
    SM_STAMP(11);
    if (tid < ST)
        dvec[tid] = f.dinv[ST * j + tid];
    acc.store(T(1), bufA + (qi * 2 + qj) * SBLK, (T *)nullptr, 0, lane);
    stage_tile(bufB, f.linv + (size_t)(j >> 1) * TILE * TILE + (size_t)(ST * (j & 1)) * TILE + ST * (j & 1), TILE);  // Xd_j
    __syncthreads();
    {
        BlkAcc<T> w;
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
        store_blk_cg<T>(w, T(1), bufC + (qi * 2 + qj) * SBLK, Ktile + (size_t)(32 * qi) * np + 32 * qj, np, lane);
    }
    SM_STAMP(12);
    publish_tile(Ff + tidx(i, j), f.epoch);  // (its barrier also orders the LDS stores of L before the products below)
    if constexpr (!FULL) {
        // mid-size models: only the 128 x 128 inverse diagonal blocks are completed here -- the tile (2 b + 1, 2 b) inside block b:
        // X = -Xd_i (L_ij Xd_j)
        if (!(same128 && i == j + 1))
            return;
    }
    // ---- the tile of the inverse factor: X_ij = -Xd_i sum_{k = j}^{i-1} L_ik X_kj ----
    acc.zero();
#pragma unroll
    for (int h = 0; h < 2; ++h)
        acc.mac(bufC + (qi * 2 + h) * SBLK, bufB + (h * 2 + qj) * SBLK, lane);  // k = j: L_ij Xd_j
    if constexpr (FULL) {
        for (int k = j + 1; k < i; ++k) {
            if (!wait_tiles(Ff + tidx(i, k), Xf + tidx(k, j), f, &s_ok))
                return;
            stage_tile(bufA, f.K + (size_t)(ST * i) * np + ST * k, np);
            stage_tile(bufB, (const T *)sa->X + (size_t)(ST * k) * np + ST * j, np);
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h)
                acc.mac(bufA + (qi * 2 + h) * SBLK, bufB + (h * 2 + qj) * SBLK, lane);
        }
    }
    SM_STAMP(13);
    if (!wait_tiles(Ff + tidx(i, i), nullptr, f, &s_ok))
        return;
    SM_STAMP(14);
    stage_tile(bufA, f.linv + (size_t)(i >> 1) * TILE * TILE + (size_t)(ST * (i & 1)) * TILE + ST * (i & 1), TILE);  // Xd_i
    acc.store(T(1), bufC + (qi * 2 + qj) * SBLK, (T *)nullptr, 0, lane);
    __syncthreads();
    {
        BlkAcc<T> x;
        x.zero();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            x.mac(bufA + (qi * 2 + h) * SBLK, bufC + (h * 2 + qj) * SBLK, lane);
        if constexpr (FULL)
            store_blk_cg<T>(x, T(-1), bufB + (qi * 2 + qj) * SBLK, (T *)sa->X + (size_t)(ST * i + 32 * qi) * np + ST * j + 32 * qj, np, lane);
        if (same128)
            x.store(T(-1), (T *)nullptr, lb + (size_t)(32 * qi) * TILE + 32 * qj, TILE, lane);
    }
    if constexpr (!FULL) {
        // the handed-over sums sit in the upper-right quadrant of the diagonal 128 x 128 block of K: leave zeros there (the
        // diagonal tile of this row has consumed them long ago: its flag was awaited above)
        for (int e = 0; e < 16; ++e) {
            const int idx = e * DF_THREADS + tid, r = idx >> 6, c = idx & 63;
            f.K[(size_t)(ST * j + r) * np + ST * i + c] = T(0);
        }
    }
    if constexpr (FULL) {
        __syncthreads();
        for (int e = 0; e < 16; ++e) {
            const int idx = e * DF_THREADS + tid, c = idx >> 6, r = idx & 63;
            sa->XT[(size_t)(ST * j + c) * np + ST * i + r] = lds_tile(bufB, r, c);
            sa->X[(size_t)(ST * j + c) * np + ST * i + r] = 0.0;  // the tile above the diagonal: structural zeros
            f.K[(size_t)(ST * j + c) * np + ST * i + r] = T(0);
        }
        publish_tile(Xf + tidx(i, j), f.epoch);
    }
    SM_STAMP(15);
}

}  // namespace dataflow
}  // namespace gpx
