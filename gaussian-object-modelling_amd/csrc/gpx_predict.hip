// gpx_predict.hip -- batched GP mean / gradient evaluation, the one-launch paths (a handful of queries; the
// AtlasBase::project loop), iso-surface compaction and small vector kernels (gfx950).
//
//   predict : f_i = sum_j k(|q_i - p_j|) alpha_j                 (reference gp_regressor.hpp:300-305, :347-353)
//             g_i = sum_j alpha_j k'(d_ij) (q_i - p_j)           (:243-249, zero-initialised)
//   One query per lane (two per thread for ILP), training points broadcast from an LDS tile of
//   256 packed {x,y,z,alpha}; no Nq x N matrix is ever materialised.  The stage is VALU-issue
//   bound (sqrt/exp), not HBM bound: 16 bytes of traffic per query against N kernel evaluations.
//   The library always runs it in fp64 (from the fp64 points and alpha, whatever the precision mode): it
//   costs 26 ms next to 2.1 s of variance at N = 16384 / 2^20 queries, and the 16384-term alternating
//   sum of a thin-plate GP is not within 1e-5 in fp32.  The kernel stays templated on the scalar type.
#include <algorithm>
#include "gpx_cov.hpp"

namespace gpx {

constexpr int PT = 256;   // training points per LDS tile
constexpr int QPB = 512;  // queries per block (2 per thread)

template <typename T>
struct alignas(16) P4 {
    T x, y, z, a;
};

static void predict_plan(long nq, int npts, int &chunks, int &chunk_len)
{
    long blocks_q = (nq + QPB - 1) / QPB;
    int ntile = npts / PT;
    long want = (1024 + blocks_q - 1) / blocks_q;
    if (want < 1)
        want = 1;
    if (want > ntile)
        want = ntile;
    int tiles_per_chunk = (int)((ntile + want - 1) / want);
    chunk_len = tiles_per_chunk * PT;
    chunks = (npts + chunk_len - 1) / chunk_len;
}

size_t predict_ws_doubles(long nq, int npts, bool grad, long plan_nq)
{
    int chunks, chunk_len;
    predict_plan(plan_nq > 0 ? plan_nq : nq, npts, chunks, chunk_len);
    return chunks > 1 ? (size_t)chunks * (size_t)nq * (grad ? 4 : 1) : 0;
}

// FASTEXP: the table-based evaluator of the exponential kernels (gpx_cov.hpp, ExpMean); the host selects it when the decay
// parameter is a positive finite number -- anything else (infinite or negative length scales: whatever the reference's
// formula gives for them) takes the general path
template <typename T, int KID, bool GRAD, bool FASTEXP>
__global__ __launch_bounds__(256) void predict_kernel(Cov<T> cov, int npts, int nlim, int chunk_len,
                                                      const T *__restrict__ px, const T *__restrict__ py,
                                                      const T *__restrict__ pz, const T *__restrict__ alpha,
                                                      long nq, const double *__restrict__ qx,
                                                      const double *__restrict__ qy, const double *__restrict__ qz,
                                                      double *__restrict__ pf, double *__restrict__ pg,
                                                      int direct)
{
    __shared__ P4<T> tile[PT];
    const int tid = threadIdx.x;
    const long qa = (long)blockIdx.x * QPB + tid, qb = qa + 256;
    const bool va = qa < nq, vb = qb < nq;
    const T ax = va ? (T)qx[qa] : T(0), ay = va ? (T)qy[qa] : T(0), az = va ? (T)qz[qa] : T(0);
    const T bx = vb ? (T)qx[qb] : T(0), by = vb ? (T)qy[qb] : T(0), bz = vb ? (T)qz[qb] : T(0);
    const int j0 = blockIdx.y * chunk_len;
    // nlim <= npts: points from nlim on are padding (alpha = 0) -- the last tile stops there, in steps of the unroll (a
    // model of 277 points is padded to 512: without the limit every query pays 512 pair evaluations for 277)
    const int j1 = min(min(npts, nlim), j0 + chunk_len);
    double fa = 0, fb = 0, gax = 0, gay = 0, gaz = 0, gbx = 0, gby = 0, gbz = 0;
    // the amplitude of the exponential kernels rides on alpha (one multiply per point and tile instead of one per pair)
    const T amp = KID == GPX_KERNEL_THINPLATE ? T(1) : cov.a;
    constexpr T TINY = sizeof(T) == 8 ? T(1e-300) : T(0);  // keeps the rsq seed of MathFast::sqrt_ finite at d = 0
    // fp64 exponential kernels: the table-based evaluator of gpx_cov.hpp
    constexpr bool TAB = FASTEXP && sizeof(T) == 8 && KID != GPX_KERNEL_THINPLATE;
    ExpMean<KID> em;
    if constexpr (TAB) {
        ExpTab::init(tid, 256);  // (the barrier at the top of the tile loop follows)
        em.prep(cov);
    }
    for (int jt = j0; jt < j1; jt += PT) {
        __syncthreads();
        tile[tid] = P4<T>{px[jt + tid], py[jt + tid], pz[jt + tid], alpha[jt + tid] * amp};
        __syncthreads();
        T sa = 0, sb = 0, tax = 0, tay = 0, taz = 0, tbx = 0, tby = 0, tbz = 0;
        const int jn = min(PT, j1 - jt);  // a multiple of 4
#pragma unroll 4
        for (int jj = 0; jj < jn; ++jj) {
            const P4<T> p = tile[jj];
            T dxa = ax - p.x, dya = ay - p.y, dza = az - p.z;
            T dxb = bx - p.x, dyb = by - p.y, dzb = bz - p.z;
            T d2a = fma(dza, dza, fma(dya, dya, fma(dxa, dxa, TINY)));
            T d2b = fma(dzb, dzb, fma(dyb, dyb, fma(dxb, dxb, TINY)));
            if constexpr (GRAD) {
                T ka, kda, kb, kdb;
                if constexpr (TAB) {
                    em.k_diff(MathFast::sqrt_(d2a), ka, kda);
                    em.k_diff(MathFast::sqrt_(d2b), kb, kdb);
                } else {
                    cov_k_diff<T, KID, MathFast, true>(cov, d2a, ka, kda);
                    cov_k_diff<T, KID, MathFast, true>(cov, d2b, kb, kdb);
                }
                sa += ka * p.a;
                sb += kb * p.a;
                T wa = kda * p.a, wb = kdb * p.a;
                tax += wa * dxa;
                tay += wa * dya;
                taz += wa * dza;
                tbx += wb * dxb;
                tby += wb * dyb;
                tbz += wb * dzb;
            } else if constexpr (TAB) {
                sa += em.k(MathFast::sqrt_(d2a)) * p.a;
                sb += em.k(MathFast::sqrt_(d2b)) * p.a;
            } else {
                sa += cov_k<T, KID, MathFast, true>(cov, d2a) * p.a;
                sb += cov_k<T, KID, MathFast, true>(cov, d2b) * p.a;
            }
        }
        fa += (double)sa;
        fb += (double)sb;
        if constexpr (GRAD) {
            gax += (double)tax;
            gay += (double)tay;
            gaz += (double)taz;
            gbx += (double)tbx;
            gby += (double)tby;
            gbz += (double)tbz;
        }
    }
    if (direct) {  // single chunk: pf = f (nq), pg = grad (nq x 3 row-major)
        if (va) {
            pf[qa] = fa;
            if constexpr (GRAD) {
                pg[3 * qa + 0] = gax;
                pg[3 * qa + 1] = gay;
                pg[3 * qa + 2] = gaz;
            }
        }
        if (vb) {
            pf[qb] = fb;
            if constexpr (GRAD) {
                pg[3 * qb + 0] = gbx;
                pg[3 * qb + 1] = gby;
                pg[3 * qb + 2] = gbz;
            }
        }
    } else {  // partials: pf[chunk][nq], pg[chunk][3][nq]
        const size_t c = blockIdx.y;
        if (va) {
            pf[c * nq + qa] = fa;
            if constexpr (GRAD) {
                pg[(c * 3 + 0) * nq + qa] = gax;
                pg[(c * 3 + 1) * nq + qa] = gay;
                pg[(c * 3 + 2) * nq + qa] = gaz;
            }
        }
        if (vb) {
            pf[c * nq + qb] = fb;
            if constexpr (GRAD) {
                pg[(c * 3 + 0) * nq + qb] = gbx;
                pg[(c * 3 + 1) * nq + qb] = gby;
                pg[(c * 3 + 2) * nq + qb] = gbz;
            }
        }
    }
}

__global__ __launch_bounds__(256) void predict_reduce_kernel(int chunks, long nq, const double *__restrict__ pf,
                                                             const double *__restrict__ pg, double *__restrict__ f,
                                                             double *__restrict__ grad)
{
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq)
        return;
    double s = 0;
    for (int c = 0; c < chunks; ++c)
        s += pf[(size_t)c * nq + q];
    f[q] = s;
    if (grad) {
        for (int d = 0; d < 3; ++d) {
            double g = 0;
            for (int c = 0; c < chunks; ++c)
                g += pg[((size_t)c * 3 + d) * nq + q];
            grad[3 * q + d] = g;
        }
    }
}

template <typename T, bool GRAD>
static void predict_t(const CovHost &h, int npts, const void *px, const void *py, const void *pz,
                      const void *alpha, long nq, const double *qx, const double *qy, const double *qz, double *f,
                      double *grad, double *ws, hipStream_t st, int nvalid, long plan_nq)
{
    const int nlim = nvalid > 0 ? std::min(npts, (nvalid + 3) / 4 * 4) : npts;
    int chunks, chunk_len;
    predict_plan(plan_nq > 0 ? plan_nq : nq, npts, chunks, chunk_len);
    Cov<T> c = lower_cov<T>(h);
    dim3 grid((unsigned)((nq + QPB - 1) / QPB), chunks);
    double *pf = chunks > 1 ? ws : f;
    double *pg = chunks > 1 ? ws + (size_t)chunks * nq : grad;
    int direct = chunks > 1 ? 0 : 1;
    if (h.s > 0 && h.s < 1e100) {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((predict_kernel<T, KID, GRAD, true>), grid, dim3(256), 0, st, c, npts,
                                                  nlim, chunk_len, (const T *)px, (const T *)py, (const T *)pz,
                                                  (const T *)alpha, nq, qx, qy, qz, pf, pg, direct));
    } else {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((predict_kernel<T, KID, GRAD, false>), grid, dim3(256), 0, st, c, npts,
                                                  nlim, chunk_len, (const T *)px, (const T *)py, (const T *)pz,
                                                  (const T *)alpha, nq, qx, qy, qz, pf, pg, direct));
    }
    if (chunks > 1)
        hipLaunchKernelGGL(predict_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, chunks, nq,
                           pf, GRAD ? pg : nullptr, f, GRAD ? grad : nullptr);
}

void launch_predict(int prec, const CovHost &cov, int npts, const void *px, const void *py, const void *pz,
                    const void *alpha, long nq, const double *qx, const double *qy, const double *qz, double *f,
                    double *grad, double *ws, hipStream_t st, int nvalid, long plan_nq)
{
    (void)prec;  // fp64 only (see the header of this file); px, py, pz, alpha are fp64 arrays
    if (grad)
        predict_t<double, true>(cov, npts, px, py, pz, alpha, nq, qx, qy, qz, f, grad, ws, st, nvalid, plan_nq);
    else
        predict_t<double, false>(cov, npts, px, py, pz, alpha, nq, qx, qy, qz, f, grad, ws, st, nvalid, plan_nq);
}

// ---- non-finite queries ----------------------------------------------------------------------------------------------------
// The reference's arithmetic answers NaN for a query with a NaN or infinite coordinate (gp_regressor.hpp:300-319: the distance is
// NaN, and so is everything computed from it).  The device kernels evaluate k with clamped fast paths (table exponential, seeded
// square root) that turn such a distance into some finite number: this pass, the last launch of every evaluate, writes NaN into
// every requested output of such a query.  It reads 24 bytes per query and writes nothing for finite ones.
__global__ __launch_bounds__(256) void poison_nonfinite_kernel(long nq, const double *__restrict__ qx, const double *__restrict__ qy,
                                                               const double *__restrict__ qz, double *f, double *v, double *grad,
                                                               double *tx, double *ty)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq)
        return;
    const double p = (qx[q] + qy[q] + qz[q]) * 0.0;
    if (p == 0.0)
        return;
    if (f)
        f[q] = p;
    if (v)
        v[q] = p;
    for (int c = 0; c < 3; ++c) {
        if (grad)
            grad[3 * q + c] = p;
        if (tx)
            tx[3 * q + c] = p;
        if (ty)
            ty[3 * q + c] = p;
    }
}

void launch_poison_nonfinite(long nq, const double *qx, const double *qy, const double *qz, double *f, double *v, double *grad,
                             double *tx, double *ty, hipStream_t st)
{
    hipLaunchKernelGGL(poison_nonfinite_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, qx, qy, qz, f, v, grad, tx,
                       ty);
}

// ---- variance epilogue: v[q] = k(0) - sum_m partial[m][q]  (gp_regressor.hpp:318-319, diagonal only)
template <typename T>
__global__ __launch_bounds__(256) void var_finish_kernel(double k0, int mtiles, long ldp,
                                                         const T *__restrict__ partial, long nq,
                                                         double *__restrict__ v)
{
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq)
        return;
    // eight loads in flight per thread (one dependent load per iteration made this 40 us for 64 x 8192 partials)
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int m = 0;
    for (; m + 8 <= mtiles; m += 8) {
        T p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            p[u] = partial[(size_t)(m + u) * ldp + q];
        s0 += (double)p[0] + (double)p[4];
        s1 += (double)p[1] + (double)p[5];
        s2 += (double)p[2] + (double)p[6];
        s3 += (double)p[3] + (double)p[7];
    }
    for (; m < mtiles; ++m)
        s0 += (double)partial[(size_t)m * ldp + q];
    v[q] = k0 - ((s0 + s1) + (s2 + s3));
}

void launch_var_finish(int prec, double k0, int mtiles, long ldp, const void *partial, long nq, double *v,
                       hipStream_t st)
{
    dim3 grid((unsigned)((nq + 255) / 256));
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(var_finish_kernel<double>, grid, dim3(256), 0, st, k0, mtiles, ldp,
                           (const double *)partial, nq, v);
    else
        hipLaunchKernelGGL(var_finish_kernel<float>, grid, dim3(256), 0, st, k0, mtiles, ldp,
                           (const float *)partial, nq, v);
}

// ---- computeTangentBasis per row (reference gp_regressor.hpp:29-44, :204-211) ---------------
__device__ __forceinline__ void tangent_basis_dev(double g0, double g1, double g2, double (&t)[3], double (&u)[3])
{
    double nrm = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    double n0 = nrm > 0 ? g0 / nrm : g0, n1 = nrm > 0 ? g1 / nrm : g1, n2 = nrm > 0 ? g2 / nrm : g2;
    // Eigen isApprox(UnitX, 1e-3): |N - e_x|^2 <= 1e-6 * min(|N|^2, 1)
    double nn = n0 * n0 + n1 * n1 + n2 * n2;
    double diff2 = (n0 - 1) * (n0 - 1) + n1 * n1 + n2 * n2;
    bool approx_x = diff2 <= 1e-6 * (nn < 1.0 ? nn : 1.0);
    double e0 = approx_x ? 0.0 : 1.0, e1 = approx_x ? 1.0 : 0.0;
    double dot = n0 * e0 + n1 * e1;
    double t0 = e0 - n0 * dot, t1 = e1 - n1 * dot, t2 = -n2 * dot;
    double tn = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    if (tn > 0) {
        t0 /= tn;
        t1 /= tn;
        t2 /= tn;
    }
    double u0 = n1 * t2 - n2 * t1, u1 = n2 * t0 - n0 * t2, u2 = n0 * t1 - n1 * t0;
    double un = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
    if (un > 0) {
        u0 /= un;
        u1 /= un;
        u2 /= un;
    }
    t[0] = t0, t[1] = t1, t[2] = t2;
    u[0] = u0, u[1] = u1, u[2] = u2;
}

__global__ __launch_bounds__(256) void tangent_basis_kernel(long nq, const double *__restrict__ grad,
                                                            double *__restrict__ tx, double *__restrict__ ty)
{
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq)
        return;
    double t[3], u[3];
    tangent_basis_dev(grad[3 * q], grad[3 * q + 1], grad[3 * q + 2], t, u);
    if (tx) {
        tx[3 * q] = t[0];
        tx[3 * q + 1] = t[1];
        tx[3 * q + 2] = t[2];
    }
    if (ty) {
        ty[3 * q] = u[0];
        ty[3 * q + 1] = u[1];
        ty[3 * q + 2] = u[2];
    }
}

// ---- a handful of queries on a small model: everything in ONE launch -------------------------------------------
// The node and the atlas call evaluate() with one query point at a time (src/gp_node.cpp:1069-1074 from hundreds of
// threads, atlas_variance.hpp:72-78, :201).  The general path needs ~7 stream operations for such a call (copy in,
// mean, K tile, GEMM, finish, copies out: 170 us); here one workgroup per query reads the point from the pinned
// staging buffer, forms k(q, .) in LDS, reduces mean and gradient, contracts k with the rows of the inverse factor
// (a wave per row, lanes along k) and writes the results back to pinned host memory.  Mean, gradient and the kernel
// values are fp64; X and 1/D are read in the model's precision.
constexpr int SE_ROWS = 64;  // rows of the inverse factor per workgroup

template <typename TX, int KID>
__global__ __launch_bounds__(256) void small_eval_kernel(Cov<double> cov, double k0, int n, int npts,
                                                         const double *__restrict__ px, const double *__restrict__ py,
                                                         const double *__restrict__ pz,
                                                         const double *__restrict__ alpha, const TX *__restrict__ X,
                                                         const TX *__restrict__ dinv, int nq,
                                                         const double *__restrict__ q, double *__restrict__ f,
                                                         double *__restrict__ v, double *__restrict__ grad,
                                                         double *__restrict__ tx, double *__restrict__ ty,
                                                         double *__restrict__ part, unsigned *__restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char se_smem[];
    double *kq = reinterpret_cast<double *>(se_smem);  // [npts]
    __shared__ double red[4][5];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int iq = blockIdx.y, rb = blockIdx.x, nrb = gridDim.x;
    const double c0 = q[iq], c1 = q[nq + iq], c2 = q[2 * nq + iq];
    // a query with a NaN or infinite coordinate answers NaN, as the reference's arithmetic does (the fast kernel evaluators
    // clamp their arguments and would answer with finite numbers): 0 for every finite query, NaN otherwise
    const double poison = (c0 + c1 + c2) * 0.0;
    double af = 0, a0 = 0, a1 = 0, a2 = 0;
    // every row block needs k(q, .) up to its last row; block 0 also reduces the mean and the gradient
    const int kmax = v ? min(npts, (rb + 1) * SE_ROWS) : 0;
    for (int j = tid; j < (rb == 0 ? npts : kmax); j += 256) {
        const double dx = c0 - px[j], dy = c1 - py[j], dz = c2 - pz[j];
        double k, kd;
        cov_k_diff<double, KID>(cov, dx * dx + dy * dy + dz * dz, k, kd);
        kq[j] = j < n ? k : 0.0;  // the padding rows of X are identity rows: they must see zeros
        if (rb == 0) {
            const double a = alpha[j];  // zero on the padding
            const double w = a * kd;
            af += a * k;
            a0 += w * dx;
            a1 += w * dy;
            a2 += w * dz;
        }
    }
    if (rb == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            af += __shfl_xor(af, off);
            a0 += __shfl_xor(a0, off);
            a1 += __shfl_xor(a1, off);
            a2 += __shfl_xor(a2, off);
        }
        if (lane == 0) {
            red[wave][0] = af;
            red[wave][1] = a0;
            red[wave][2] = a1;
            red[wave][3] = a2;
        }
    }
    __syncthreads();  // publishes kq (and red)
    if (rb == 0 && tid == 0) {
        const double fs = red[0][0] + red[1][0] + red[2][0] + red[3][0] + poison;
        const double g0 = red[0][1] + red[1][1] + red[2][1] + red[3][1] + poison;
        const double g1 = red[0][2] + red[1][2] + red[2][2] + red[3][2] + poison;
        const double g2 = red[0][3] + red[1][3] + red[2][3] + red[3][3] + poison;
        f[iq] = fs;
        if (grad) {
            grad[3 * iq] = g0;
            grad[3 * iq + 1] = g1;
            grad[3 * iq + 2] = g2;
        }
        if (tx || ty) {
            double t[3], u[3];
            tangent_basis_dev(g0, g1, g2, t, u);
            for (int c = 0; c < 3; ++c) {
                if (tx)
                    tx[3 * iq + c] = t[c];
                if (ty)
                    ty[3 * iq + c] = u[c];
            }
        }
    }
    if (!v)
        return;
    // ---- rows [rb * 64, rb * 64 + 64): a wave per row, four rows in flight per wave (the loop is latency-bound) ----
    double acc = 0.0;
    const int j0 = rb * SE_ROWS + wave * (SE_ROWS / 4);
    for (int jj = 0; jj < SE_ROWS / 4; jj += 4) {
        double w4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + jj + u;
            const TX *row = X + (size_t)j * npts;
            double wj = 0.0;
            for (int k = lane; k <= j; k += 64)
                wj += (double)row[k] * kq[k];
            w4[u] = wj;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double wj = w4[u];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                wj += __shfl_xor(wj, off);
            acc += wj * wj * (double)dinv[j0 + jj + u];
        }
    }
    if (lane == 0)
        red[wave][4] = acc;
    __syncthreads();
    if (tid == 0) {
        part[(size_t)iq * nrb + rb] = red[0][4] + red[1][4] + red[2][4] + red[3][4];
        __threadfence();
        last = atomicAdd(&done[iq], 1u) == (unsigned)(nrb - 1);
        if (last) {  // the last row block of this query adds the partial sums in a fixed order
            __threadfence();
            double s = 0.0;
            for (int r = 0; r < nrb; ++r)
                s += __hip_atomic_load(&part[(size_t)iq * nrb + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[iq] = k0 - s + poison;
            done[iq] = 0;  // ready for the next launch
        }
    }
}

size_t small_eval_scratch_bytes(int nq_max, int npts_max)
{
    return sizeof(double) * (size_t)nq_max * (npts_max / SE_ROWS) + sizeof(unsigned) * (size_t)nq_max;
}

template <typename TX>
static void small_eval_t(const CovHost &h, int n, int npts, const double *px, const double *py, const double *pz,
                         const double *alpha, const void *X, const void *dinv, int nq, int nq_max, const double *q,
                         double *f, double *v, double *grad, double *tx, double *ty, void *scratch, hipStream_t st)
{
    Cov<double> c = lower_cov<double>(h);
    const size_t shmem = sizeof(double) * (size_t)npts;
    const int nrb = v ? npts / SE_ROWS : 1;
    double *part = (double *)scratch;
    unsigned *done = (unsigned *)(part + (size_t)nq_max * (SMALL_EVAL_NP_MAX / SE_ROWS));
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((small_eval_kernel<TX, KID>), dim3((unsigned)nrb, (unsigned)nq), dim3(256),
                                              shmem, st, c, h.k0, n, npts, px, py, pz, alpha, (const TX *)X,
                                              (const TX *)dinv, nq, q, f, v, grad, tx, ty, part, done));
}

// q: qx | qy | qz (nq each); q and the outputs may be pinned host memory; v == NULL: no variance (X, dinv unused).
// scratch: small_eval_scratch_bytes(nq_max, SMALL_EVAL_NP_MAX) bytes of device memory, zeroed once.
void launch_small_eval(int prec, const CovHost &cov, int n, int npts, const double *px, const double *py,
                       const double *pz, const double *alpha, const void *X, const void *dinv, int nq, int nq_max,
                       const double *q, double *f, double *v, double *grad, double *tx, double *ty, void *scratch,
                       hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        small_eval_t<double>(cov, n, npts, px, py, pz, alpha, X, dinv, nq, nq_max, q, f, v, grad, tx, ty, scratch, st);
    else
        small_eval_t<float>(cov, n, npts, px, py, pz, alpha, X, dinv, nq, nq_max, q, f, v, grad, tx, ty, scratch, st);
}

// ---- AtlasBase::project (reference include/atlas/atlas.hpp:201-276), all start points at once ----------------
// Eigen 3.2 predicates: v.isMuchSmallerThan(1e3, 1e-1) <=> |v|^2 <= 1e4 ; v.isZero(p) <=> |v_i| <= p for all i
__device__ __forceinline__ bool atlas_vec_ok(double a, double b, double c, double zero_prec)
{
    const bool much_smaller = a * a + b * b + c * c <= 1e-1 * 1e-1 * 1e3 * 1e3;
    const bool is_zero = fabs(a) <= zero_prec && fabs(b) <= zero_prec && fabs(c) <= zero_prec;
    return much_smaller && !is_zero;
}

// :225-252 -- f at the current point is known: NaN test, |f| < f_tol, then the step along g
__global__ __launch_bounds__(256) void project_pre_kernel(long nq, double f_tol, double step_mul,
                                                          double *__restrict__ cx, double *__restrict__ cy,
                                                          double *__restrict__ cz, const double *__restrict__ g,
                                                          const double *__restrict__ f_cur, int *__restrict__ status)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq || status[i] != 0)
        return;
    const double fc = f_cur[i];
    if (isnan(fc) || isinf(fc)) {
        status[i] = -1;
        return;
    }
    if (fabs(fc) < f_tol) {
        status[i] = 1;
        return;
    }
    const double s0 = step_mul * fc * g[3 * i], s1 = step_mul * fc * g[3 * i + 1], s2 = step_mul * fc * g[3 * i + 2];
    if (atlas_vec_ok(s0, s1, s2, 1e-6)) {
        cx[i] -= s0;
        cy[i] -= s1;
        cz[i] -= s2;
    }
}

// :260-271 -- mean and gradient at the moved point are known: adopt the gradient, improvement test, count
__global__ __launch_bounds__(256) void project_post_kernel(long nq, double improve_tol, int max_iter,
                                                           const double *__restrict__ f_new,
                                                           const double *__restrict__ grad_new, double *__restrict__ g,
                                                           double *__restrict__ f_cur, int *__restrict__ iter,
                                                           int *__restrict__ status, unsigned *__restrict__ active)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq || status[i] != 0)
        return;
    const double n0 = grad_new[3 * i], n1 = grad_new[3 * i + 1], n2 = grad_new[3 * i + 2];
    if (atlas_vec_ok(n0, n1, n2, 1e-5)) {
        g[3 * i] = n0;
        g[3 * i + 1] = n1;
        g[3 * i + 2] = n2;
    }
    const double fo = f_new[i], fc = f_cur[i];
    f_cur[i] = fo;  // the mean at the (moved) current point, also what is reported with the result
    if (fabs(fo - fc) < improve_tol) {
        status[i] = 2;
        return;
    }
    const int it = iter[i] + 1;
    iter[i] = it;
    if (it >= max_iter)
        status[i] = 3;
    else
        atomicAdd(active, 1u);
}

// The whole loop of AtlasBase::project in ONE launch, for models whose points and weights fit the LDS
// (32 bytes per training point): one wave per start point, the lanes share the sum over the training points
// (butterfly reduction, so every lane holds the same f and gradient and takes the same branches), and the
// iteration runs in registers.  Replaces 4 launches per iteration of the generic path below: the reference's
// defaults (step_mul 0.001, max_iter 500) make almost every call run the full 500 iterations.
template <int KID>
__global__ __launch_bounds__(256) void project_fused_kernel(Cov<double> cov, int npts, const double *__restrict__ px,
                                                            const double *__restrict__ py,
                                                            const double *__restrict__ pz,
                                                            const double *__restrict__ alpha, long nq, double f_tol,
                                                            double improve_tol, double step_mul, int max_iter,
                                                            double *__restrict__ cx, double *__restrict__ cy,
                                                            double *__restrict__ cz, const double *__restrict__ g_in,
                                                            double *__restrict__ f_out, int *__restrict__ iter_out,
                                                            int *__restrict__ status_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char proj_smem[];
    double *sx = reinterpret_cast<double *>(proj_smem), *sy = sx + npts, *sz = sy + npts, *sa = sz + npts;
    for (int j = threadIdx.x; j < npts; j += 256) {
        sx[j] = px[j];
        sy[j] = py[j];
        sz[j] = pz[j];
        sa[j] = alpha[j];  // zero on the padding
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave_global = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long q = wave_global; q < nq; q += nwaves) {
        double c0 = cx[q], c1 = cy[q], c2 = cz[q];
        double g0 = g_in[3 * q], g1 = g_in[3 * q + 1], g2 = g_in[3 * q + 2];
        // mean and un-normalised gradient at (c0, c1, c2), identical in every lane
        auto eval = [&](double &f, double &n0, double &n1, double &n2) {
            double af = 0, a0 = 0, a1 = 0, a2 = 0;
            for (int j = lane; j < npts; j += 64) {
                const double dx = c0 - sx[j], dy = c1 - sy[j], dz = c2 - sz[j];
                double k, kd;
                cov_k_diff<double, KID>(cov, dx * dx + dy * dy + dz * dz, k, kd);
                const double w = sa[j] * kd;
                af += sa[j] * k;
                a0 += w * dx;
                a1 += w * dy;
                a2 += w * dz;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                af += __shfl_xor(af, off);
                a0 += __shfl_xor(a0, off);
                a1 += __shfl_xor(a1, off);
                a2 += __shfl_xor(a2, off);
            }
            f = af, n0 = a0, n1 = a1, n2 = a2;
        };
        double fc, t0, t1, t2;
        eval(fc, t0, t1, t2);  // :225 of the first iteration (the gradient is not used)
        int it = 0, status = 3;
        while (it < max_iter) {
            if (isnan(fc) || isinf(fc)) {
                status = -1;
                break;
            }
            if (fabs(fc) < f_tol) {
                status = 1;
                break;
            }
            const double s0 = step_mul * fc * g0, s1 = step_mul * fc * g1, s2 = step_mul * fc * g2;
            if (atlas_vec_ok(s0, s1, s2, 1e-6)) {
                c0 -= s0;
                c1 -= s1;
                c2 -= s2;
            }
            double fo, n0, n1, n2;
            eval(fo, n0, n1, n2);
            if (atlas_vec_ok(n0, n1, n2, 1e-5)) {
                g0 = n0;
                g1 = n1;
                g2 = n2;
            }
            const double df = fabs(fo - fc);
            fc = fo;
            if (df < improve_tol) {
                status = 2;
                break;
            }
            ++it;
        }
        if (lane == 0) {
            cx[q] = c0;
            cy[q] = c1;
            cz[q] = c2;
            f_out[q] = fc;
            iter_out[q] = it;
            status_out[q] = status;
        }
    }
}

size_t project_fused_lds_bytes(int npts) { return sizeof(double) * 4 * (size_t)npts; }

// false: the model does not fit the LDS, use the per-iteration path
bool launch_project_fused(const CovHost &h, int npts, const double *px, const double *py, const double *pz,
                          const double *alpha, long nq, double f_tol, double improve_tol, double step_mul,
                          int max_iter, double *cx, double *cy, double *cz, const double *g, double *f, int *iter,
                          int *status, hipStream_t st)
{
    const size_t shmem = project_fused_lds_bytes(npts);
    if (shmem > 128 * 1024)
        return false;
    Cov<double> c = lower_cov<double>(h);
    const long blocks = std::min<long>((nq + 3) / 4, 256L * 8);
    GPX_DISPATCH_KID(h.id, {
        auto kern = project_fused_kernel<KID>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), shmem, st, c, npts, px, py, pz, alpha, nq, f_tol,
                           improve_tol, step_mul, max_iter, cx, cy, cz, g, f, iter, status);
    });
    return true;
}

void launch_project_pre(long nq, double f_tol, double step_mul, double *cx, double *cy, double *cz, const double *g,
                        const double *f_cur, int *status, hipStream_t st)
{
    hipLaunchKernelGGL(project_pre_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, f_tol, step_mul,
                       cx, cy, cz, g, f_cur, status);
}

void launch_project_post(long nq, double improve_tol, int max_iter, const double *f_new, const double *grad_new,
                         double *g, double *f_cur, int *iter, int *status, unsigned *active, hipStream_t st)
{
    hipLaunchKernelGGL(project_post_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, improve_tol,
                       max_iter, f_new, grad_new, g, f_cur, iter, status, active);
}

void launch_tangent_basis(long nq, const double *grad, double *tx, double *ty, hipStream_t st)
{
    hipLaunchKernelGGL(tangent_basis_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, grad, tx,
                       ty);
}

// ---- iso-surface selection: keep the queries with |f| <= tol, in query order (src/gp_node.cpp:1075) ----
// Three small passes (count per 256-block, scan of the block counts, scatter) so the order is deterministic.
__global__ __launch_bounds__(256) void surf_count_kernel(long nq, const double *__restrict__ f, double tol,
                                                         unsigned *__restrict__ block_cnt)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const bool keep = q < nq && fabs(f[q]) <= tol;
    __shared__ unsigned wc[4];
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0)
        wc[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
        block_cnt[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ __launch_bounds__(1024) void surf_scan_kernel(long nblocks, unsigned *__restrict__ block_cnt,
                                                         unsigned long long *__restrict__ total)
{
    // exclusive scan of block_cnt in place by ONE workgroup (nblocks <= a few 10^5)
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const long per = (nblocks + 1023) / 1024;
    const long lo = t * per, hi = lo + per < nblocks ? lo + per : nblocks;
    unsigned long long s = 0;
    for (long i = lo; i < hi; ++i)
        s += block_cnt[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = t ? part[t - 1] : 0;
    for (long i = lo; i < hi; ++i) {
        const unsigned c = block_cnt[i];
        block_cnt[i] = (unsigned)run;  // offsets fit 32 bits: nq < 2^32 survivors
        run += c;
    }
    if (t == 1023)
        *total = part[1023];
}

__global__ __launch_bounds__(256) void surf_scatter_kernel(long nq, const double *__restrict__ f, double tol,
                                                           const unsigned *__restrict__ block_off, size_t capacity,
                                                           const double *__restrict__ qx,
                                                           const double *__restrict__ qy,
                                                           const double *__restrict__ qz, long long *__restrict__ idx,
                                                           double *__restrict__ fs, double *__restrict__ sx,
                                                           double *__restrict__ sy, double *__restrict__ sz,
                                                           const long long *__restrict__ idx_map)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const bool keep = q < nq && fabs(f[q]) <= tol;
    __shared__ unsigned wc[4];
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0)
        wc[w] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned base = block_off[blockIdx.x];
    for (int i = 0; i < w; ++i)
        base += wc[i];
    if (keep) {
        const size_t pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < capacity) {
            idx[pos] = idx_map ? idx_map[q] : q;  // (second stage of the screened selection: q counts candidates)
            fs[pos] = f[q];
            sx[pos] = qx[q];
            sy[pos] = qy[q];
            sz[pos] = qz[q];
        }
    }
}

void launch_surface_select(long nq, const double *f, double tol, unsigned *block_cnt, unsigned long long *total,
                           size_t capacity, const double *qx, const double *qy, const double *qz, long long *idx,
                           double *fs, double *sx, double *sy, double *sz, hipStream_t st, const long long *idx_map)
{
    const long nb = (nq + 255) / 256;
    hipLaunchKernelGGL(surf_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, nq, f, tol, block_cnt);
    hipLaunchKernelGGL(surf_scan_kernel, dim3(1), dim3(1024), 0, st, nb, block_cnt, total);
    hipLaunchKernelGGL(surf_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, st, nq, f, tol, block_cnt, capacity, qx,
                       qy, qz, idx, fs, sx, sy, sz, idx_map);
}

// ---- fp32 screen in front of the iso-surface selection (round 6; reference src/gp_node.cpp:1066-1100) -------------------
// gpx_model_sample_surface keeps the lattice points with |f| <= f_tol; deciding that for the 97 % of a grid that lie far from
// the surface does not need the 1e-10 of the fp64 mean kernel (29 fp64 instructions per pair: 30 ms of C3_surface's 133).
// Stage 1 (here): fhat = the mean in fp32 -- points relative to the cloud's centre, native v_sqrt_f32 / v_exp_f32, the
// amplitude folded into alpha, 64-term fp32 partial sums added up in fp64 -- and a PROVED bound on |fhat - f|; a query is a
// candidate when |fhat| - bound <= f_tol.  Stage 2 (caller): the fp64 mean of the candidates, then the exact test -- the
// selected set and every returned f are those of the fp64 filter, bit for bit.
// The bound, u = 2^-24, B = |q - c|_inf + max_k |p_k - c|_inf, t = s d, exponential kernels k = a e^-t P(t), P' <= P:
//   coordinates rounded to fp32 and subtracted: the difference vector is off by <= 2 u B per component, so d by <= 2 sqrt(3) u B,
//   plus 2.5 u d for the fma chain and the square root (d <= sqrt(3) B):            |dhat - d| <= 8 u B
//   t = fl(s) dhat and the argument of v_exp_f32 (t log2 e, 1 ulp instruction):      e^-t off by <= u (8 s B + 3.5) absolutely
//   the Matern factor P(t) (3 roundings, P' <= P, e^-t t^j bounded):                 |khat - k| / a <= u (16 s B + 12)
//   alpha_k a rounded once, 64-term fp32 fma chains (<= 64 u sum |terms|), fp64 sums of the chains (negligible):
//        |fhat - f| <= u (16 s B + 80) sum_k |alpha_k| a            -- the kernel applies it with a factor 2 in hand.
// The thin plate is not screened (its weights sum to 10^3 .. 10^5 k(0): the bound exceeds any f_tol).
struct alignas(16) ScreenPt {
    float x, y, z, a;
};
// one workgroup: the model's points relative to the centre and alpha a in fp32 (padding: zeros), and -- in a fixed order -- the two
// model constants of the bound: stats[0] = sum |alpha_k a|, stats[1] = max_k |p_k - c|_inf
// (coordinates are stored times sc = s log2 e, so that the kernel's distance IS the argument of v_exp_f32 and the Matern factor a
// polynomial in it: two packed instructions less per pair; a relative change of scale, the bound is unaffected)
__global__ __launch_bounds__(1024) void screen_pack_kernel(int n, int npts, double amp, double sc, const double *__restrict__ px,
                                                           const double *__restrict__ py, const double *__restrict__ pz,
                                                           const double *__restrict__ alpha, const double *__restrict__ cen,
                                                           ScreenPt *__restrict__ out, double *__restrict__ stats)
{
    __shared__ double s_sum[1024], s_max[1024];
    const int t = threadIdx.x;
    const double cx = cen[0], cy = cen[1], cz = cen[2];
    double sum = 0.0, mx = 0.0;
    for (int j = t; j < npts; j += 1024) {
        ScreenPt p{0.0f, 0.0f, 0.0f, 0.0f};
        if (j < n) {
            const double x = px[j] - cx, y = py[j] - cy, z = pz[j] - cz, a = alpha[j] * amp;
            p = ScreenPt{(float)(x * sc), (float)(y * sc), (float)(z * sc), (float)a};
            sum += fabs(a);
            mx = fmax(mx, fmax(fabs(x), fmax(fabs(y), fabs(z))));
        }
        out[j] = p;
    }
    s_sum[t] = sum, s_max[t] = mx;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (t < off)
            s_sum[t] += s_sum[t + off], s_max[t] = fmax(s_max[t], s_max[t + off]);
        __syncthreads();
    }
    if (t == 0)
        stats[0] = s_sum[0], stats[1] = s_max[0];
}

// g[q] = max(|fhat(q)| - bound(q), 0): a lower bound of |f(q)| (0 for a query whose fp32 arithmetic went non-finite: stage 2
// decides).  Two queries per thread as packed pairs; the training points broadcast from an LDS tile.
template <int KID>
__global__ __launch_bounds__(256) void screen_kernel(float s, int npts, const ScreenPt *__restrict__ pts,
                                                     const double *__restrict__ cen, const double *__restrict__ stats, long nq,
                                                     const double *__restrict__ qx, const double *__restrict__ qy,
                                                     const double *__restrict__ qz, double *__restrict__ g)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ ScreenPt tile[PT];
    const int tid = threadIdx.x;
    const long qa = (long)blockIdx.x * QPB + tid, qb = qa + 256;
    const bool va = qa < nq, vb = qb < nq;
    const double cx = cen[0], cy = cen[1], cz = cen[2];
    const double sc = (double)s * 1.44269504088896340736;  // coordinates in units of 1 / (s log2 e): d = the argument of v_exp_f32
    const double ax = va ? qx[qa] - cx : 0.0, ay = va ? qy[qa] - cy : 0.0, az = va ? qz[qa] - cz : 0.0;
    const double bx = vb ? qx[qb] - cx : 0.0, by = vb ? qy[qb] - cy : 0.0, bz = vb ? qz[qb] - cz : 0.0;
    const f2 X = {(float)(ax * sc), (float)(bx * sc)}, Y = {(float)(ay * sc), (float)(by * sc)}, Z = {(float)(az * sc), (float)(bz * sc)};
    // Matern factors in t = d ln 2: 1 + t = fma(d, ln 2, 1); 1 + t + t^2 / 3 = fma(d, fma(d, ln2^2 / 3, ln 2), 1)
    constexpr float LN2 = 0.69314718055994530942f, C52 = 0.16015100463940046f;
    double fa = 0.0, fb = 0.0;
    for (int jt = 0; jt < npts; jt += PT) {
        __syncthreads();
        tile[tid] = jt + tid < npts ? pts[jt + tid] : ScreenPt{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        const int jn = min(PT, npts - jt);  // a multiple of 4 (the caller rounds the point count up: alpha = 0 there)
        for (int j0 = 0; j0 < jn; j0 += 64) {
            f2 acc = {0.0f, 0.0f};
            const int j1 = min(64, jn - j0);
#pragma unroll 4
            for (int jj = 0; jj < j1; ++jj) {
                const ScreenPt p = tile[j0 + jj];
                const f2 dx = X - p.x, dy = Y - p.y, dz = Z - p.z;
                const f2 d2 = dz * dz + (dy * dy + dx * dx);
                const f2 d = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                const f2 e = {__builtin_amdgcn_exp2f(-d.x), __builtin_amdgcn_exp2f(-d.y)};
                f2 k;
                if constexpr (KID == GPX_KERNEL_MATERN32) {
                    k = e * (d * LN2 + 1.0f);
                } else if constexpr (KID == GPX_KERNEL_MATERN52) {
                    k = e * (d * (d * C52 + LN2) + 1.0f);
                } else {
                    k = e;
                }
                acc += k * p.a;
            }
            fa += (double)acc.x;
            fb += (double)acc.y;
        }
    }
    constexpr double U2 = 2.0 * 5.9604644775390625e-08;  // 2 u
    const double l1 = stats[0], pmax = stats[1], sd = (double)s;
    if (va) {
        const double B = fmax(fabs(ax), fmax(fabs(ay), fabs(az))) + pmax;
        g[qa] = fmax(fabs(fa) - U2 * l1 * (16.0 * sd * B + 80.0), 0.0);
    }
    if (vb) {
        const double B = fmax(fabs(bx), fmax(fabs(by), fabs(bz))) + pmax;
        g[qb] = fmax(fabs(fb) - U2 * l1 * (16.0 * sd * B + 80.0), 0.0);
    }
}

size_t surface_screen_ws_doubles(int npts) { return 2 * (size_t)npts + 2; }

// the kernels the screen exists for: a positive finite decay parameter of an exponential kernel (gpx_predict.hip, predict_t)
bool surface_screen_takes(const CovHost &h) { return h.id != GPX_KERNEL_THINPLATE && h.s > 0 && h.s < 1e30 && h.a > 0 && h.a < 1e300; }

void launch_surface_screen(const CovHost &h, int n, int npts, const double *px, const double *py, const double *pz,
                           const double *alpha, const double *cen, long nq, const double *qx, const double *qy,
                           const double *qz, double *g, double *ws, hipStream_t st)
{
    ScreenPt *pts = reinterpret_cast<ScreenPt *>(ws);
    double *stats = ws + 2 * (size_t)npts;
    const int nlim = std::min(npts, (n + 3) / 4 * 4);
    hipLaunchKernelGGL(screen_pack_kernel, dim3(1), dim3(1024), 0, st, n, npts, h.a, (double)(float)h.s * 1.44269504088896340736,
                       px, py, pz, alpha, cen, pts, stats);
    const dim3 grid((unsigned)((nq + QPB - 1) / QPB));
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((screen_kernel<KID>), grid, dim3(256), 0, st, (float)h.s, nlim, pts, cen, stats,
                                              nq, qx, qy, qz, g));
}

// ---- iterative-refinement helpers ------------------------------------------------------------
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double val)
{
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(val));
}

__global__ __launch_bounds__(256) void residual_kernel(int n, const double *__restrict__ y,
                                                       const double *__restrict__ f, const double *__restrict__ s2,
                                                       const double *__restrict__ alpha, double *__restrict__ r,
                                                       double *__restrict__ rmax)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    double a = 0;
    if (i < n) {
        double v = y[i] - f[i] - s2[i] * alpha[i];
        r[i] = v;
        a = fabs(v);
        if (!(a == a))
            a = __longlong_as_double(0x7ff0000000000000LL);  // NaN -> +inf so it is noticed
    }
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(a, off);
        a = o > a ? o : a;
    }
    if ((threadIdx.x & 63) == 0 && a > 0)
        atomic_max_nonneg(rmax, a);
}

void launch_residual(int n, const double *y, const double *f, const double *s2, const double *alpha, double *r,
                     double *rmax, hipStream_t st)
{
    hipLaunchKernelGGL(residual_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, y, f, s2, alpha, r, rmax);
}

template <typename T>
__global__ __launch_bounds__(256) void axpy_cast_kernel(int n, int npad, double *__restrict__ alpha_d,
                                                        const T *__restrict__ delta, T *__restrict__ alpha_t)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npad)
        return;
    if (i < n) {
        double a = alpha_d[i] + (double)delta[i];
        alpha_d[i] = a;
        alpha_t[i] = (T)a;
    } else {
        alpha_t[i] = T(0);
    }
}

void launch_axpy_cast(int prec, int n, int npad, double *alpha_d, const void *delta, void *alpha_t, hipStream_t st)
{
    dim3 grid((npad + 255) / 256);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(axpy_cast_kernel<double>, grid, dim3(256), 0, st, n, npad, alpha_d,
                           (const double *)delta, (double *)alpha_t);
    else
        hipLaunchKernelGGL(axpy_cast_kernel<float>, grid, dim3(256), 0, st, n, npad, alpha_d, (const float *)delta,
                           (float *)alpha_t);
}

template <typename T>
__global__ __launch_bounds__(256) void cast_vec_kernel(int n, int npad, const double *__restrict__ src,
                                                       T *__restrict__ dst, double offset)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npad)
        dst[i] = i < n ? (T)(src[i] - offset) : T(0);
}

void launch_cast_vec(int prec, int n, int npad, const double *src, void *dst, hipStream_t st, double offset)
{
    dim3 grid((npad + 255) / 256);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(cast_vec_kernel<double>, grid, dim3(256), 0, st, n, npad, src, (double *)dst, offset);
    else
        hipLaunchKernelGGL(cast_vec_kernel<float>, grid, dim3(256), 0, st, n, npad, src, (float *)dst, offset);
}

__global__ __launch_bounds__(256) void cast_d2f_kernel(size_t n, const double *__restrict__ src,
                                                       float *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = (float)src[i];
}

__global__ __launch_bounds__(256) void cast_f2d_kernel(size_t n, const float *__restrict__ src,
                                                       double *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = (double)src[i];
}

// Square np x np matrices whose content is lower block-triangular (the factor, its inverse): only the 128 x 128
// tiles on and below the block diagonal are read and converted; tiles above are skipped or (zero_upper) written as
// zeros without being read.  4 consecutive elements per lane (16-byte fp32 / 32-byte fp64 accesses).
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_lower_kernel(int np, const TS *__restrict__ src, TD *__restrict__ dst,
                                                         int zero_upper, int row_tile0)
{
    const int ti = blockIdx.y + row_tile0, tj = blockIdx.x;
    if (tj > ti && !zero_upper)
        return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 lanes x 4 columns, 8 rows per pass
#pragma unroll 4
    for (int r = ty; r < TILE; r += 8) {
        const size_t off = (size_t)(ti * TILE + r) * np + tj * TILE + tx * 4;
        TD o[4] = {TD(0), TD(0), TD(0), TD(0)};
        if (tj <= ti) {
            TS v[4];
            if constexpr (sizeof(TS) == 4) {
                const float4 t = *reinterpret_cast<const float4 *>(src + off);
                v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
            } else {
                const double2 t0 = *reinterpret_cast<const double2 *>(src + off);
                const double2 t1 = *reinterpret_cast<const double2 *>(src + off + 2);
                v[0] = t0.x, v[1] = t0.y, v[2] = t1.x, v[3] = t1.y;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                o[c] = (TD)v[c];
        }
        if constexpr (sizeof(TD) == 4) {
            *reinterpret_cast<float4 *>(dst + off) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            *reinterpret_cast<double2 *>(dst + off) = make_double2(o[0], o[1]);
            *reinterpret_cast<double2 *>(dst + off + 2) = make_double2(o[2], o[3]);
        }
    }
}

// rows [row0, row1) of the np x np matrix only (multiples of 128; row1 < 0: to the end); column tiles beyond the last
// row tile of the range are not visited
void launch_cast_lower_f2d(int np, const float *src, double *dst, bool zero_upper, hipStream_t st, int row0, int row1)
{
    const int t0 = row0 / TILE, t1 = (row1 < 0 ? np : row1) / TILE;
    if (t1 <= t0)
        return;
    hipLaunchKernelGGL((cast_lower_kernel<float, double>), dim3(zero_upper ? np / TILE : t1, t1 - t0), dim3(256), 0, st, np,
                       src, dst, zero_upper ? 1 : 0, t0);
}

void launch_cast_lower_d2f(int np, const double *src, float *dst, bool zero_upper, hipStream_t st)
{
    hipLaunchKernelGGL((cast_lower_kernel<double, float>), dim3(np / TILE, np / TILE), dim3(256), 0, st, np, src, dst,
                       zero_upper ? 1 : 0, 0);
}

void launch_cast_f2d(size_t n, const float *src, double *dst, hipStream_t st)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(cast_f2d_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, src, dst);
}

void launch_cast_d2f(size_t n, const double *src, float *dst, hipStream_t st)
{
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192)
        blocks = 8192;
    hipLaunchKernelGGL(cast_d2f_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, src, dst);
}

// Eigen row.normalize() of the training-point normals (gp_regressor.hpp:174)
__global__ __launch_bounds__(256) void normalize_rows3_kernel(long n, double *__restrict__ g)
{
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    double a = g[3 * i], b = g[3 * i + 1], c = g[3 * i + 2];
    double nrm = sqrt(a * a + b * b + c * c);
    if (nrm > 0) {
        g[3 * i] = a / nrm;
        g[3 * i + 1] = b / nrm;
        g[3 * i + 2] = c / nrm;
    }
}

void launch_normalize_rows3(long n, double *g, hipStream_t st)
{
    hipLaunchKernelGGL(normalize_rows3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, g);
}

}  // namespace gpx
