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
// Every wait has a time budget: a workgroup that runs out of patience raises the abort flag, all others see it in their
// polls and leave, and the host redoes the create with the general chain (gpx_stats.solve_fallbacks = 1) -- as
// tri_solve_kernel does.  Workgroups are enumerated column by column, so a factorisation job only ever waits for
// workgroups with a lower index; the inverse jobs also wait for later ones, which is why all (<= 136) must be resident.
#include <algorithm>
#include <cstdlib>
#include <atomic>

#include "gpx_dataflow.hpp"
#include "gpx_dataflow_wide.hpp"

namespace gpx {

using namespace dataflow;  // flags, ld_cg / st_cg, the tile workgroup (gpx_dataflow.hpp)

namespace {
constexpr int SM_THREADS = DF_THREADS;
constexpr int SM_LDS_DOUBLES = DF_LDS_ELEMS;

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

// first launch of the small-model create: the dataflow factorisation with the inverse factor (gpx_dataflow.hpp, FULL)
template <int KID>
__global__ __launch_bounds__(SM_THREADS, 1) void small_factor_kernel(const SmallArgs *__restrict__ ap)
{
    const SmallArgs &a = *ap;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < 16) {
        // state of the second launch: barrier counter, residual maxima (stream order: it starts after this grid has ended)
        if (tid == 0)
            st_flag(a.flags + a.bar_idx, 0);
        if (tid < 8)
            a.rmaxv[tid] = 0.0;
        if (tid < BLOB_META)
            a.d_meta[tid] = tid < 3 ? a.cen[tid] : (tid == 3 ? 1.0 : 0.0);
    }
    if ((int)blockIdx.x < a.nbt)  // the tiles of column 0: (i, 0), i = blockIdx.x
        scatter_rows(a, (int)blockIdx.x);
    FactorArgs<double> f;
    f.n = a.n, f.np = a.np, f.nbt = a.nbt, f.nb = a.nb, f.ntiles = a.ntiles;
    f.cov = a.cov;
    f.K = a.K, f.linv = a.linv, f.d = a.d, f.dinv = a.dinv;
    f.flags = a.flags, f.epoch = a.epoch, f.wait_ticks = a.wait_ticks, f.abort_idx = a.abort_idx, f.pre_idx = a.pre_idx;
    f.tmax = a.tmax, f.tij = a.tij, f.negcnt = a.negcnt, f.badrow = a.badrow;
    f.dbg = a.dbg;
    factor_tile<double, KID, true>(f, ap, sm);
}

// the same dataflow for mid-size models (no inverse factor, points already in the working type): replaces kbuild + the
// launch chain of the blocked LDL^T where that chain, not the flops, sets the time (profiles/r05_ldlt_sweep.txt)
template <typename T, int KID>
__global__ __launch_bounds__(DF_THREADS, 1) void mid_factor_kernel(FactorArgs<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    factor_tile<T, KID, false>(f, nullptr, reinterpret_cast<T *>(sm_raw));
}

// the 128 x 128-tile form of the same dataflow for LARGE models, where the 64 x 64 form is HBM-bound (gpx_dataflow_wide.hpp)
template <typename T, int KID>
__global__ __launch_bounds__(WIDE_THREADS, 1) void wide_factor_kernel(FactorArgs<T> f, int *info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    wide_factor_tile<T, KID>(f, info, reinterpret_cast<T *>(sm_raw));
}

// ... and what the chain reads from d_info afterwards: [0] first bad pivot, [1] negative pivots, [2..3] arg-max pair of the
// squared distance, [6] a wait gave up (the factor is void: the host redoes the create with the launch chain)
__global__ __launch_bounds__(256) void mid_finish_kernel(int nbt, int nb, int ntiles, const int *__restrict__ negcnt,
                                                         const int *__restrict__ badrow, const double *__restrict__ tmax,
                                                         const int *__restrict__ tij, const u64 *__restrict__ abortf, u64 epoch,
                                                         int *__restrict__ info)
{
    __shared__ double sb[256];
    __shared__ int st[256];
    const int tid = threadIdx.x;
    double best = -2.0;
    int bt = 0;
    // tiles column by column (index = first tile of the column + row offset); tiles in the padding (row >= nb) wrote nothing
    for (int jj = 0, base = 0; jj < nb; base += nbt - jj, ++jj)
        for (int ii = jj + tid; ii < nb; ii += 256) {
            const int t = base + (ii - jj);
            if (tmax[t] > best || (tmax[t] == best && t < bt))
                best = tmax[t], bt = t;
        }
    sb[tid] = best, st[tid] = bt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && (sb[tid + s] > sb[tid] || (sb[tid + s] == sb[tid] && st[tid + s] < st[tid])))
            sb[tid] = sb[tid + s], st[tid] = st[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        if (negcnt) {  // (the 128 x 128 form counts its pivots in info[0..1] itself, as the launch chain's diagonal blocks do)
            int bad = 0, neg = 0;
            for (int t = 0; t < nb; ++t) {
                if (!bad && badrow[t])
                    bad = badrow[t];
                neg += negcnt[t];
            }
            info[0] = bad, info[1] = neg;
        }
        info[2] = tij[2 * st[0]], info[3] = tij[2 * st[0] + 1];
        info[6] = ld_flag(abortf) == epoch ? 1 : 0;
    }
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
        const u64 t0 = wall_clock64();
        for (int s = 0;; ++s) {
            if (ld_flag(cnt) >= target) {
                ok = true;
                break;
            }
            if (a.wait_ticks <= 0 || ((s & 31) == 31 && (ld_flag(abortf) == a.epoch || (long long)(wall_clock64() - t0) > a.wait_ticks)))
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

// sum over c = c0 + lane, c0 + lane + 64, ... < c1 of row[c] * v[c]: the loads of four steps are issued before the first is used
// (one trip to memory per four steps instead of one per step: the phases of the second launch are latency-bound)
__device__ __forceinline__ double dot_strided(const double *__restrict__ row, const double *v, int c0, int c1, int lane)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = c0 + lane;
    for (; c + 192 < c1; c += 256) {
        const double a0 = row[c], a1 = row[c + 64], a2 = row[c + 128], a3 = row[c + 192];
        s0 = fma(a0, v[c], s0);
        s1 = fma(a1, v[c + 64], s1);
        s2 = fma(a2, v[c + 128], s2);
        s3 = fma(a3, v[c + 192], s3);
    }
    for (; c < c1; c += 64)
        s0 = fma(row[c], v[c], s0);
    return (s0 + s1) + (s2 + s3);
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
    __shared__ double pts[3][SMALL_CREATE_MAX_NP];  // the model's fp64 points (row corrections, residual)
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
    for (int k = tid; k < n; k += SM_THREADS)
        pts[0][k] = a.d_x[k], pts[1][k] = a.d_y[k], pts[2][k] = a.d_z[k];
    if (!aborted) {
        for (int it = 0;; ++it) {
            // ---- t = X rhs, u = D^-1 t (L y = b and the scaling of LDLT::solve, gp_regressor.hpp:163) ----
            stage_vec(it == 0 ? a.d_lab : a.d_r, np);
            for (int r = gw; r < np; r += nw) {
                double s = 0.0;
                if (r < nrows) {
                    s = wave_sum(dot_strided(a.X + (size_t)r * np, vec, 0, r + 1, lane));
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
                        double xrow[4];
                        for (int l0 = lane; l0 < lend; l0 += 256) {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                xrow[u] = l0 + 64 * u < lend ? row[l0 + 64 * u] : 0.0;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                            const int l = min(l0 + 64 * u, lend - 1);  // (past the end: xr = 0, any valid point)
                            const double xv = xrow[u];
                            double x = pts[0][l] - a.cen[0], y = pts[1][l] - a.cen[1], z = pts[2][l] - a.cen[2];
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
                    s = wave_sum(dot_strided(a.XT + (size_t)c * np, vec, c, nrows, lane));
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
                        const double px = pts[0][r], py = pts[1][r], pz = pts[2][r];
                        double s = 0.0;
                        for (int c = lane; c < n; c += 64) {
                            const double dx = px - pts[0][c], dy = py - pts[1][c], dz = pz - pts[2][c];
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
#define GPX_MID_ATTR(T, KID)                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mid_factor_kernel<T, KID>),               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DF_LDS_ELEMS * sizeof(T)))
        GPX_MID_ATTR(double, GPX_KERNEL_GAUSSIAN);
        GPX_MID_ATTR(double, GPX_KERNEL_THINPLATE);
        GPX_MID_ATTR(double, GPX_KERNEL_MATERN32);
        GPX_MID_ATTR(double, GPX_KERNEL_MATERN52);
        GPX_MID_ATTR(float, GPX_KERNEL_GAUSSIAN);
        GPX_MID_ATTR(float, GPX_KERNEL_THINPLATE);
        GPX_MID_ATTR(float, GPX_KERNEL_MATERN32);
        GPX_MID_ATTR(float, GPX_KERNEL_MATERN52);
#undef GPX_MID_ATTR
#define GPX_WIDE_ATTR(T, KID)                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_factor_kernel<T, KID>),               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes<T>())
        GPX_WIDE_ATTR(double, GPX_KERNEL_GAUSSIAN);
        GPX_WIDE_ATTR(double, GPX_KERNEL_THINPLATE);
        GPX_WIDE_ATTR(double, GPX_KERNEL_MATERN32);
        GPX_WIDE_ATTR(double, GPX_KERNEL_MATERN52);
        GPX_WIDE_ATTR(float, GPX_KERNEL_GAUSSIAN);
        GPX_WIDE_ATTR(float, GPX_KERNEL_THINPLATE);
        GPX_WIDE_ATTR(float, GPX_KERNEL_MATERN32);
        GPX_WIDE_ATTR(float, GPX_KERNEL_MATERN52);
#undef GPX_WIDE_ATTR
    });
}

template <typename T>
static void mid_factor_t(const CovHost &h, const MidFactorArgs &m, hipStream_t st)
{
    FactorArgs<T> f;
    const int tile = m.wide ? WT : ST;
    f.n = m.n, f.np = m.np, f.nbt = m.np / tile, f.nb = (m.n + tile - 1) / tile, f.ntiles = f.nbt * (f.nbt + 1) / 2;
    f.cov = lower_cov<T>(h);
    f.K = (T *)m.K, f.linv = (T *)m.linv, f.d = (T *)m.d, f.dinv = (T *)m.dinv;
    f.px = (const T *)m.px, f.py = (const T *)m.py, f.pz = (const T *)m.pz, f.ps2 = (const T *)m.ps2;
    const MidWs lay = mid_ws_layout(m.np);
    char *ws = (char *)m.ws;
    f.flags = (u64 *)(ws + lay.flags), f.epoch = m.epoch, f.wait_ticks = m.wait_ticks;
    f.abort_idx = 2 * f.ntiles, f.pre_idx = 2 * f.ntiles + 2;
    f.tmax = (double *)(ws + lay.tmax), f.tij = (int *)(ws + lay.tij);
    f.negcnt = (int *)(ws + lay.negcnt), f.badrow = (int *)(ws + lay.badrow);
    if (m.wide) {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((wide_factor_kernel<T, KID>), dim3(f.ntiles), dim3(WIDE_THREADS), wide_lds_bytes<T>(), st, f, m.info));
#ifdef WIDE_TIMING
        {  // diagnostic build: the shader-clock split of the launch, summed over its workgroups (stderr)
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> hs((size_t)f.ntiles * 8);
            (void)hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(wide_timing), hs.size() * sizeof(unsigned long long));
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int nd = 0;
            for (int t = 0, J = 0, rem = 0; t < f.ntiles && t < WIDE_TIMING_MAX_TILES; ++t) {
                for (J = 0, rem = t; rem >= f.nbt - J; rem -= f.nbt - J, ++J) {
                }
                for (int k = 0; k < 8; ++k) {
                    sum[k] += (double)hs[(size_t)t * 8 + k];
                    if (rem == 0)
                        diag[k] += (double)hs[(size_t)t * 8 + k];
                }
                nd += rem == 0;
            }
            std::fprintf(stderr, "wide_timing %s np=%d tiles=%d: workgroup cycles %.4g | flag waits %.1f %% | operands to LDS + barrier %.1f %% | MFMAs + barrier %.1f %% | diag routine %.1f %% | prologue %.1f %% | sums stored %.1f %% | epilogue + publish %.1f %% | rest %.1f %%  (diagonal tiles: %.0f cycles each, %.0f in the routine, %.0f waiting)\n",
                         sizeof(T) == 4 ? "fp32" : "fp64", f.np, f.ntiles, sum[0], 100 * sum[1] / sum[0], 100 * sum[2] / sum[0], 100 * sum[3] / sum[0],
                         100 * sum[4] / sum[0], 100 * sum[5] / sum[0], 100 * sum[6] / sum[0], 100 * sum[7] / sum[0],
                         100 * (sum[0] - sum[1] - sum[2] - sum[3] - sum[4] - sum[5] - sum[6] - sum[7]) / sum[0], diag[0] / nd, diag[4] / nd, diag[1] / nd);
        }
#endif
        hipLaunchKernelGGL(mid_finish_kernel, dim3(1), dim3(256), 0, st, f.nbt, f.nb, f.ntiles, (const int *)nullptr, (const int *)nullptr,
                           f.tmax, f.tij, f.flags + f.abort_idx, f.epoch, m.info);
        return;
    }
    const size_t lds = DF_LDS_ELEMS * sizeof(T);
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((mid_factor_kernel<T, KID>), dim3(f.ntiles), dim3(DF_THREADS), lds, st, f));
    hipLaunchKernelGGL(mid_finish_kernel, dim3(1), dim3(256), 0, st, f.nbt, f.nb, f.ntiles, f.negcnt, f.badrow, f.tmax, f.tij,
                       f.flags + f.abort_idx, f.epoch, m.info);
}

void launch_mid_factor(int prec, const CovHost &h, const MidFactorArgs &m, hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        mid_factor_t<double>(h, m, st);
    else
        mid_factor_t<float>(h, m, st);
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
    // workgroups of the second launch: its phases are latency-bound (a row or two per wave, one trip to memory each), so more
    // waves help until the barrier's cost takes over -- measured at N = 277 / 724 / 1024 (solve us): 16 workgroups 99 / 244 / 378,
    // 32: 69 / 149 / 225, 64: 58 / 108 / 157, 128: 59 / 102 / 137, 256: 66 / 113 / 152
    const int g = a.np <= 512 ? 64 : 128;
    GPX_DISPATCH_KID(kernel_id, hipLaunchKernelGGL((small_alpha_kernel<KID>), dim3(g), dim3(SM_THREADS), 0, st, d_args));
    if (ev_solve)
        (void)hipEventRecord(ev_solve, st);
    if (demote)
        hipLaunchKernelGGL(small_demote_kernel, dim3(std::max(8, a.np * a.np / (SM_THREADS * 16))), dim3(SM_THREADS), 0, st, d_args);
}

}  // namespace gpx
