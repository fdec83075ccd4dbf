// gpx_dgp.hip -- first slice of the reference's SECOND GP library, gp::GaussianProcess (SURVEY 8f.4 / VERDICT r2 #8):
// a GP whose training data are function values AND gradients (surface normals) at the same points,
// include/gp/GaussianProcess.h of the reference: compute() :532-583 (the 4N x 4N covariance of values and
// derivatives, layout [values (N) | d/dx, d/dy, d/dz of point 0 | ... of point N-1], Cholesky), update_alpha()
// :505-528, f() :237-252 (value and gradient at a query = k_star (4 x 4N) alpha), var() :256-269,
// logLikelihood() :376-385, SampleSet (src/gp/SampleSet.cpp:23-35: targets, then the normals point by point).
//
// The reference's own code for this library does not build and has no defined results:
// CovSE::getDiff2 (include/gp/CovSE.h:84-89) lacks the delta_de / l^2 term of the second derivative and multiplies
// the noise into the kernel, ThinPlate has no getDiff2 at all, update_k_star (:459-497) indexes points up to 4N,
// var() conditions on the value block only, evaluate() (:227-235) returns zeros.  What is built here is the algorithm
// those functions are written towards, with exact derivative blocks: for a radial kernel k(r), u = x - x', r = |u|,
// g = k'(r) / r, h = g'(r) / r:
//     cov(f(x),      f(x'))       = k(r)
//     cov(d_d f(x),  f(x'))       = g u_d                    cov(f(x), d_e f(x')) = -g u_e
//     cov(d_d f(x),  d_e f(x'))   = -g delta_de - h u_d u_e
//   CovSE     (CovSE.h:70-74)     k = sf^2 exp(-r^2 / (2 l^2)),  g = -k / l^2,  h = k / l^4
//   ThinPlate (CovThinPlate.h:80-89) k = 2 r^3 - 3 R r^2 + R^3,  g = 6 r - 6 R, h = 6 / r  (h u_d u_e -> 0 at r = 0)
// noise^2 on the whole diagonal (BaseCovFunc: "dirac").  fp64 throughout.
//
// Everything dense goes through the machinery of the first library: the matrix is filled in 128 x 128 tiles of the
// lower block triangle, factorised by the blocked LDL^T on the matrix cores (factorize_matrix; for a positive
// definite matrix L D^1/2 is the reference's llt().matrixL()), solved by block substitution, and the variance is the
// same fused contraction v = k(0) - sum_j (X k_star)_j^2 / D_j with X = L^-1.
#include "gpx_model.hpp"

namespace gpx {

struct DCov {
    int id;  // GPX_KERNEL_SE or GPX_KERNEL_THINPLATE
    double sf2, inv_l2, R, R3;
};

// k, g = k'/r, h = g'/r at squared distance r2 (h_rr = h * r2 is what multiplies the unit-vector products: finite at 0)
__device__ __forceinline__ void dcov_eval(const DCov &c, double r2, double &k, double &g, double &h)
{
    if (c.id == GPX_KERNEL_THINPLATE) {
        const double r = sqrt(r2);
        k = (r - c.R) * (r - c.R) * (2.0 * r + c.R);
        g = 6.0 * (r - c.R);
        h = r > 0.0 ? 6.0 / r : 0.0;  // only ever multiplied by u_d u_e = O(r^2)
    } else {
        k = c.sf2 * exp(-0.5 * r2 * c.inv_l2);
        g = -k * c.inv_l2;
        h = k * c.inv_l2 * c.inv_l2;
    }
}

// A row of the covariance stands for one observation, coded 4 i + c: c = 0 the value at point i, c = 1..3 its derivative
// d = c - 1.  A model made by gpx_dgp_create holds them in the reference's order [values (n) | d/dx d/dy d/dz of point 0 |
// ...] (GaussianProcess.h:553-567); rows appended by gpx_dgp_add follow behind the old ones, so the order is kept as a
// table (row[a], npad entries, -1 on the padding) and not as a formula.
__device__ __forceinline__ double dgp_entry(const DCov &c, int ca, int cb, bool same_row, const double *__restrict__ x,
                                            const double *__restrict__ y, const double *__restrict__ z, double sn2)
{
    const int ia = ca >> 2, da = (ca & 3) - 1;
    const int ib = cb >> 2, db = (cb & 3) - 1;
    const double u[3] = {x[ia] - x[ib], y[ia] - y[ib], z[ia] - z[ib]};
    double k, g, h;
    dcov_eval(c, u[0] * u[0] + u[1] * u[1] + u[2] * u[2], k, g, h);
    double v;
    if (da < 0 && db < 0)
        v = k;
    else if (da >= 0 && db < 0)
        v = g * u[da];  // d/dx_{a,d} k(x_a, x_b)
    else if (da < 0)
        v = -g * u[db];  // d/dx_{b,e} k(x_a, x_b)
    else
        v = -(da == db ? g : 0.0) - h * u[da] * u[db];
    return same_row ? v + sn2 : v;
}

// tiles of the lower block triangle from tile row t_first on (0: the whole matrix; > 0: the rows of an append)
__global__ __launch_bounds__(256) void dgp_kbuild_kernel(DCov c, int t_first, int npad, const int *__restrict__ row,
                                                         const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ z, double sn2, double *__restrict__ K)
{
    int ti, tj;
    tri_decode((int)blockIdx.x + t_first * (t_first + 1) / 2, ti, tj);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll 2
    for (int r = ty; r < TILE; r += 8) {
        const int a = ti * TILE + r;
        const int ca = row[a];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int b = tj * TILE + tx * 4 + cc;
            const int cb = row[b];
            double v;
            if (ca >= 0 && cb >= 0)
                v = dgp_entry(c, ca, cb, a == b, x, y, z, sn2);
            else
                v = a == b ? 1.0 : 0.0;  // identity on the padding
            K[(size_t)a * npad + b] = v;
        }
    }
}

// alpha from the model's row order into the reference's layout [values | derivatives point by point] (what f() reads)
__global__ __launch_bounds__(256) void dgp_alpha_layout_kernel(int n, int n4, const int *__restrict__ row,
                                                               const double *__restrict__ a_rows, double *__restrict__ a_ref)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n4)
        return;
    const int code = row[a], i = code >> 2, cmp = code & 3;
    a_ref[cmp == 0 ? i : n + 3 * i + cmp - 1] = a_rows[a];
}

// value and gradient of the posterior mean at the queries: out[q][0..3] = k_star(q) (4 x 4N) alpha.
// One query per thread, the training points and their four weights through LDS.
__global__ __launch_bounds__(256) void dgp_predict_kernel(DCov c, int n, const double *__restrict__ x,
                                                          const double *__restrict__ y, const double *__restrict__ z,
                                                          const double *__restrict__ alpha, long nq,
                                                          const double *__restrict__ qx, const double *__restrict__ qy,
                                                          const double *__restrict__ qz, double *__restrict__ out)
{
    __shared__ double sx[256], sy[256], sz[256], sa[256][4];
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < nq;
    const double ax = live ? qx[q] : 0.0, ay = live ? qy[q] : 0.0, az = live ? qz[q] : 0.0;
    double f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        if (j < n) {
            sx[threadIdx.x] = x[j], sy[threadIdx.x] = y[j], sz[threadIdx.x] = z[j];
            sa[threadIdx.x][0] = alpha[j];
            sa[threadIdx.x][1] = alpha[n + 3 * j], sa[threadIdx.x][2] = alpha[n + 3 * j + 1], sa[threadIdx.x][3] = alpha[n + 3 * j + 2];
        }
        __syncthreads();
        const int cnt = min(256, n - j0);
        for (int t = 0; t < cnt; ++t) {
            const double u0 = ax - sx[t], u1 = ay - sy[t], u2 = az - sz[t];
            double k, g, h;
            dcov_eval(c, u0 * u0 + u1 * u1 + u2 * u2, k, g, h);
            const double a0 = sa[t][0], a1 = sa[t][1], a2 = sa[t][2], a3 = sa[t][3];
            const double ua = u0 * a1 + u1 * a2 + u2 * a3;  // u . (derivative weights of point t)
            f0 += k * a0 - g * ua;
            // d/dx*_d: g u_d alpha_value + sum_e (-g delta_de - h u_d u_e) alpha_e
            const double s = g * a0 - h * ua;
            f1 += s * u0 - g * a1;
            f2 += s * u1 - g * a2;
            f3 += s * u2 - g * a3;
        }
    }
    if (live) {
        out[4 * q] = f0;
        out[4 * q + 1] = f1;
        out[4 * q + 2] = f2;
        out[4 * q + 3] = f3;
    }
}

// operand of the variance contraction: row 0 of k_star for every query, Kq[q][b] = cov(f(q), observation b)
__global__ __launch_bounds__(256) void dgp_kstar_kernel(DCov c, int npad, const int *__restrict__ row,
                                                        const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ z, long nq_valid,
                                                        const double *__restrict__ qx, const double *__restrict__ qy,
                                                        const double *__restrict__ qz, double *__restrict__ Kq)
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long q0 = (long)blockIdx.y * TILE;
    const int b0 = blockIdx.x * TILE + tx * 4;
#pragma unroll 2
    for (int r = ty; r < TILE; r += 8) {
        const long q = q0 + r;
        const bool live = q < nq_valid;
        const double ax = live ? qx[q] : 0.0, ay = live ? qy[q] : 0.0, az = live ? qz[q] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int b = b0 + cc;
            const int cb = row[b];
            double v = 0.0;
            if (live && cb >= 0) {
                const int ib = cb >> 2, db = (cb & 3) - 1;
                const double u[3] = {ax - x[ib], ay - y[ib], az - z[ib]};
                double k, g, h;
                dcov_eval(c, u[0] * u[0] + u[1] * u[1] + u[2] * u[2], k, g, h);
                v = db < 0 ? k : -g * u[db];
            }
            Kq[(size_t)q * npad + b] = v;
        }
    }
}

// ---- logLikelihoodGradient (GaussianProcess.h:387-410), CovSE only (the other classes of the reference have no grad()) ----
// d/d log(l) of entry (a, b), from CovSE::grad's d k / d log(l) = k z, z = r^2 / l^2 (CovSE.h:96-101), carried through the
// derivative blocks: g = -k / l^2 -> g (z - 2), h = k / l^4 -> h (z - 4).  No noise term: sn does not depend on l.
__device__ __forceinline__ double dgp_entry_dlogl(const DCov &c, int ca, int cb, const double *__restrict__ x,
                                                  const double *__restrict__ y, const double *__restrict__ z)
{
    const int ia = ca >> 2, da = (ca & 3) - 1;
    const int ib = cb >> 2, db = (cb & 3) - 1;
    const double u[3] = {x[ia] - x[ib], y[ia] - y[ib], z[ia] - z[ib]};
    const double r2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    double k, g, h;
    dcov_eval(c, r2, k, g, h);
    const double zz = r2 * c.inv_l2;
    if (da < 0 && db < 0)
        return k * zz;
    if (da >= 0 && db < 0)
        return g * (zz - 2.0) * u[da];
    if (da < 0)
        return -g * (zz - 2.0) * u[db];
    return -(da == db ? g * (zz - 2.0) : 0.0) - h * (zz - 4.0) * u[da] * u[db];
}

// the whole matrix dK / d log(l) (both triangles: it is the A operand of a plain product), zero on the padding
__global__ __launch_bounds__(256) void dgp_dk_kernel(DCov c, int npad, const int *__restrict__ row,
                                                     const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ z, double *__restrict__ dK)
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll 2
    for (int r = ty; r < TILE; r += 8) {
        const int a = blockIdx.y * TILE + r;
        const int ca = row[a];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int b = blockIdx.x * TILE + tx * 4 + cc;
            const int cb = row[b];
            dK[(size_t)a * npad + b] = (ca >= 0 && cb >= 0) ? dgp_entry_dlogl(c, ca, cb, x, y, z) : 0.0;
        }
    }
}

// out[a] = alpha_a (dK alpha)_a : one row per workgroup (the host adds the rows up: a fixed order)
__global__ __launch_bounds__(256) void dgp_quad_rows_kernel(int n4, int npad, const double *__restrict__ dK,
                                                            const double *__restrict__ alpha, double *__restrict__ out)
{
    __shared__ double red[256];
    const int a = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < n4; b += 256)
        s += dK[(size_t)a * npad + b] * alpha[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w)
            red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[a] = alpha[a] * red[0];
}

// out[m] = (1 / D_m) sum_a X[m][a]^2 : the rows of tr(K^-1) = sum_m |X_m|^2 / D_m, X = L^-1 unit lower
__global__ __launch_bounds__(256) void dgp_trkinv_rows_kernel(int npad, const double *__restrict__ X,
                                                              const double *__restrict__ dinv, double *__restrict__ out)
{
    __shared__ double red[256];
    const int m = blockIdx.x;
    double s = 0.0;
    for (int a = threadIdx.x; a <= m; a += 256) {
        const double v = X[(size_t)m * npad + a];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w)
            red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[m] = red[0] * dinv[m];
}

// tr(K^-1 dK) = sum_m (1 / D_m) sum_a X[m][a] Z[a][m] with Z = dK X^T: one 32 x 32 tile (m, a) per workgroup, Z's tile
// through LDS (it is read across its rows); out[tile] = the tile's share (zero above the diagonal of X)
__global__ __launch_bounds__(256) void dgp_trace_tiles_kernel(int npad, const double *__restrict__ X,
                                                              const double *__restrict__ Z, const double *__restrict__ dinv,
                                                              double *__restrict__ out)
{
    __shared__ double zt[32][33];
    __shared__ double red[256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int m0 = blockIdx.y * 32, a0 = blockIdx.x * 32;
    double s = 0.0;
    if (a0 <= m0 + 31) {
        for (int rr = ty; rr < 32; rr += 8)
            zt[rr][tx] = Z[(size_t)(a0 + rr) * npad + m0 + tx];  // zt[a - a0][m - m0]
        __syncthreads();
        for (int rr = ty; rr < 32; rr += 8) {
            const int m = m0 + rr, a = a0 + tx;
            if (a <= m)
                s += X[(size_t)m * npad + a] * zt[tx][rr] * dinv[m];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w)
            red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

}  // namespace gpx

// ---- host ----------------------------------------------------------------------------------------------------------
struct gpx_dgp {
    gpx_model *m = nullptr;  // the generic engine: matrix, factor, streams, workspaces (n = 4 n_pts)
    int n_pts = 0;
    gpx::DCov cov{};
    double sn2 = 0, k0 = 0;
    double *d_x = nullptr, *d_y = nullptr, *d_z = nullptr;  // n_pts each
    double *d_alpha = nullptr;                               // 4 n_pts, the reference's layout (f() reads it)
    int *d_row = nullptr;                                    // npad: observation code of every matrix row, -1 on the padding
    std::vector<int> h_row;                                  // 4 n_pts
    std::vector<double> h_alpha, h_alpha_rows, h_y;          // reference layout; row order; targets in row order (padded)
    double loglik = 0;
    int n_negative = 0;
    int appended_from = 0;  // rows the last gpx_dgp_add carried over from the old factor (0: built from scratch)
    // what create / add were given (the optimiser refits on them)
    gpx_kernel kernel{};
    double noise = 0;
    gpx_options opt{};
    std::vector<double> in_x, in_y, in_z, in_t, in_n;  // n, n, n, n, 3 n (zeros where no normals were given)
};

extern "C" void gpx_dgp_destroy(gpx_dgp *g)
{
    if (!g)
        return;
    if (g->m) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(g->m->device);
        (void)hipStreamSynchronize(g->m->stream);
        for (void *p : {(void *)g->d_x, (void *)g->d_y, (void *)g->d_z, (void *)g->d_alpha, (void *)g->d_row})
            if (p)
                (void)hipFree(p);
        if (prev >= 0)
            (void)hipSetDevice(prev);
        gpx_model_destroy(g->m);
    }
    delete g;
}

// Everything after the arguments are checked.  old == null: compute() from scratch on the n samples.  old != null (the
// first old->n_pts samples are old's, in its row order): the leading t0 = 128 * floor(4 n_old / 128) rows and columns of old's
// factor -- L, D and the inverse diagonal blocks -- are carried over, the rows behind them are built and appended to it
// (factorize_matrix_append: the rank-n update of the first library), so the cost is that of the new rows only.
static int dgp_build(const gpx_kernel *kernel, double noise, size_t n, const double *x, const double *y, const double *z,
                     const double *target, const double *normals, const gpx_options *opt, const gpx_dgp *old, gpx_dgp **out)
{
    // the engine: an fp64 shell of order 4n on the requested device (its kernel id is irrelevant: nothing of the first
    // library's kernel-specific code runs on it)
    gpx_options o{};
    if (opt)
        o = *opt;
    else
        o.device = -1, o.ir_steps = -1;
    o.precision = GPX_PREC_F64;
    gpx_kernel gk{};
    gk.id = GPX_KERNEL_GAUSSIAN, gk.p[0] = 1.0, gk.p[1] = 1.0;
    gpx_model *m = nullptr;
    int rc = gpx_model_create_shell(&gk, 4 * n, &o, &m);
    if (rc)
        return rc;
    gpx_dgp *g = new gpx_dgp();
    g->m = m;
    g->n_pts = (int)n;
    g->cov.id = kernel->id;
    if (kernel->id == GPX_KERNEL_SE) {
        g->cov.sf2 = kernel->p[0] * kernel->p[0];
        g->cov.inv_l2 = 1.0 / (kernel->p[1] * kernel->p[1]);
        g->k0 = g->cov.sf2;
    } else {
        g->cov.R = kernel->p[0];
        g->cov.R3 = kernel->p[0] * kernel->p[0] * kernel->p[0];
        g->k0 = g->cov.R3;
    }
    g->sn2 = noise * noise;  // ThinPlate::create, CovThinPlate.h:112: sn2 = noise^2
    auto bail = [&](int code) {
        const std::string keep = g_err;
        gpx_dgp_destroy(g);
        g_err = keep;
        return code;
    };
    const int n4 = 4 * (int)n, np = m->npad;
    const int n_old = old ? old->n_pts : 0;
    const int t0 = old ? 4 * n_old / TILE * TILE : 0;
    hipStream_t s = m->stream;
#define DGP_CHK(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return bail(fail(e__ == hipErrorOutOfMemory ? GPX_E_OOM : GPX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__))); \
    } while (0)
    DGP_CHK(hipSetDevice(m->device));
    if ((rc = alloc_factor_buffers(m)))
        return bail(rc);
    DGP_CHK(hipMalloc((void **)&g->d_x, sizeof(double) * n));
    DGP_CHK(hipMalloc((void **)&g->d_y, sizeof(double) * n));
    DGP_CHK(hipMalloc((void **)&g->d_z, sizeof(double) * n));
    DGP_CHK(hipMalloc((void **)&g->d_alpha, sizeof(double) * (size_t)np));
    DGP_CHK(hipMalloc((void **)&g->d_row, sizeof(int) * (size_t)np));
    DGP_CHK(hipMemcpyAsync(g->d_x, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    DGP_CHK(hipMemcpyAsync(g->d_y, y, sizeof(double) * n, hipMemcpyHostToDevice, s));
    DGP_CHK(hipMemcpyAsync(g->d_z, z, sizeof(double) * n, hipMemcpyHostToDevice, s));
    // row order: old's rows first (a fresh model: none), then [values | derivatives point by point] of the other samples
    // -- for a fresh model the layout of compute() (:553-567) and SampleSet (src/gp/SampleSet.cpp:27-35)
    std::vector<int> hrow((size_t)np, -1);
    if (old)
        std::copy(old->h_row.begin(), old->h_row.end(), hrow.begin());
    {
        const int n_new = (int)n - n_old;
        for (int i = 0; i < n_new; ++i) {
            hrow[(size_t)4 * n_old + i] = 4 * (n_old + i);
            for (int d = 0; d < 3; ++d)
                hrow[(size_t)4 * n_old + n_new + 3 * i + d] = 4 * (n_old + i) + 1 + d;
        }
    }
    g->h_row.assign(hrow.begin(), hrow.begin() + n4);
    g->h_y.assign((size_t)np, 0.0);
    for (int a = 0; a < n4; ++a) {
        const int i = hrow[a] >> 2, c = hrow[a] & 3;
        g->h_y[a] = c == 0 ? target[i] : (normals ? normals[3 * i + c - 1] : 0.0);
    }
    DGP_CHK(hipMemcpyAsync(g->d_row, hrow.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice, s));
    DGP_CHK(hipMemcpyAsync(m->t_b, g->h_y.data(), sizeof(double) * (size_t)np, hipMemcpyHostToDevice, s));
    DGP_CHK(hipMemsetAsync(m->d_info, 0, sizeof(int) * 8, s));
    // compute(): the covariance matrix (GaussianProcess.h:545-567) and its factorisation (:578)
    const int nt = np / TILE, tf = t0 / TILE;
    (void)hipEventRecord(m->ev[EV_T0], s);
    if (t0 > 0) {
        const gpx_model *mo = old->m;
        DGP_CHK(hipStreamSynchronize(mo->stream));
        DGP_CHK(hipMemcpy2DAsync(m->Kmat, sizeof(double) * np, mo->Kmat, sizeof(double) * mo->npad, sizeof(double) * t0, t0,
                                 hipMemcpyDeviceToDevice, s));
        DGP_CHK(hipMemcpyAsync(m->linv, mo->linv, sizeof(double) * (size_t)tf * TILE * TILE, hipMemcpyDeviceToDevice, s));
        DGP_CHK(hipMemcpyAsync(m->t_d, mo->t_d, sizeof(double) * t0, hipMemcpyDeviceToDevice, s));
        DGP_CHK(hipMemcpyAsync(m->t_dinv, mo->t_dinv, sizeof(double) * t0, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(gpx::dgp_kbuild_kernel, dim3(nt * (nt + 1) / 2 - tf * (tf + 1) / 2), dim3(256), 0, s, g->cov, tf, np,
                       g->d_row, g->d_x, g->d_y, g->d_z, g->sn2, (double *)m->Kmat);
    (void)hipEventRecord(m->ev[EV_KBUILD], s);
    if (t0 > 0)
        factorize_matrix_append(m, t0);
    else
        factorize_matrix(m);
    g->appended_from = t0;
    (void)hipEventRecord(m->ev[EV_FACTOR], s);
    // update_alpha(): alpha = K^-1 y (:505-528)
    solve_factored(m, m->t_b, m->t_yv, m->t_xs);
    hipLaunchKernelGGL(gpx::dgp_alpha_layout_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, (int)n, n4, g->d_row,
                       (const double *)m->t_xs, g->d_alpha);
    (void)hipEventRecord(m->ev[EV_SOLVE], s);
    int info[8];
    std::vector<double> hd((size_t)n4);
    g->h_alpha.assign((size_t)n4, 0.0);
    g->h_alpha_rows.assign((size_t)n4, 0.0);
    DGP_CHK(hipMemcpyAsync(info, m->d_info, sizeof(info), hipMemcpyDeviceToHost, s));
    DGP_CHK(hipMemcpyAsync(hd.data(), m->t_d, sizeof(double) * (size_t)n4, hipMemcpyDeviceToHost, s));
    DGP_CHK(hipMemcpyAsync(g->h_alpha.data(), g->d_alpha, sizeof(double) * (size_t)n4, hipMemcpyDeviceToHost, s));
    DGP_CHK(hipMemcpyAsync(g->h_alpha_rows.data(), m->t_xs, sizeof(double) * (size_t)n4, hipMemcpyDeviceToHost, s));
    DGP_CHK(hipStreamSynchronize(s));
    DGP_CHK(hipGetLastError());
#undef DGP_CHK
    float ms;
    m->stats = gpx_stats{};
    if (hipEventElapsedTime(&ms, m->ev[EV_T0], m->ev[EV_KBUILD]) == hipSuccess)
        m->stats.t_kbuild_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_KBUILD], m->ev[EV_FACTOR]) == hipSuccess)
        m->stats.t_factor_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_FACTOR], m->ev[EV_SOLVE]) == hipSuccess)
        m->stats.t_solve_ms = ms;
    m->stats.n = n4, m->stats.n_padded = np, m->stats.n_negative_pivots = info[1];
    g->n_negative = info[1];
    if (info[0] != 0)
        return bail(fail(GPX_E_SINGULAR, "covariance of values and derivatives: zero or non-finite pivot at row " + std::to_string(info[0] - 1)));
    if (info[1] != 0)  // the reference takes llt() of this matrix (:578): it must be positive definite
        return bail(fail(GPX_E_SINGULAR, "covariance of values and derivatives is not positive definite (" + std::to_string(info[1]) + " negative pivots)"));
    // logLikelihood() (:376-385) with the determinant of the WHOLE matrix: -y.alpha / 2 - log det / 2 - 4n log(2 pi) / 2
    double quad = 0, logdet = 0;
    for (int i = 0; i < n4; ++i) {
        quad += g->h_y[i] * g->h_alpha_rows[i];
        logdet += std::log(hd[i]);
    }
    g->loglik = -0.5 * quad - 0.5 * logdet - 0.5 * n4 * std::log(2.0 * M_PI);
    m->ready = true;
    g->kernel = *kernel, g->noise = noise, g->opt = o;
    g->in_x.assign(x, x + n), g->in_y.assign(y, y + n), g->in_z.assign(z, z + n), g->in_t.assign(target, target + n);
    g->in_n.assign(3 * n, 0.0);
    if (normals)
        std::copy(normals, normals + 3 * n, g->in_n.begin());
    *out = g;
    return GPX_OK;
}

extern "C" int gpx_dgp_create(const gpx_kernel *kernel, double noise, size_t n, const double *x, const double *y,
                              const double *z, const double *target, const double *normals, const gpx_options *opt,
                              gpx_dgp **out)
{
    if (!out)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!kernel)
        return fail(GPX_E_NULL, "Empty kernel pointer");
    if (n == 0)
        return fail(GPX_E_EMPTY, "No training data available");  // GaussianProcess.h:239
    if (!x || !y || !z || !target)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (kernel->id != GPX_KERNEL_SE && kernel->id != GPX_KERNEL_THINPLATE)
        return fail(GPX_E_BAD_ARG, "gpx_dgp: kernel must be GPX_KERNEL_SE or GPX_KERNEL_THINPLATE");
    if (!(noise >= 0.0) || !std::isfinite(noise))
        return fail(GPX_E_BAD_ARG, "noise must be finite and non-negative");  // Desc::isValid, GaussianProcess.h:216-222
    if (n > ((size_t)1 << 18))
        return fail(GPX_E_BAD_ARG, "n too large");
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(x[i]) || !std::isfinite(y[i]) || !std::isfinite(z[i]) || !std::isfinite(target[i]) ||
            (normals && (!std::isfinite(normals[3 * i]) || !std::isfinite(normals[3 * i + 1]) || !std::isfinite(normals[3 * i + 2]))))
            return fail(GPX_E_NAN_INPUT, "non-finite value in the training data");
    return dgp_build(kernel, noise, n, x, y, z, target, normals, opt, nullptr, out);
}

// add_patterns (GaussianProcess.h:340-374): the new samples' rows are appended to the existing factor, as the reference
// appends rows to its Cholesky factor (:356-368: k = L^-1 k, L(j, :) = k^T, L(j, j) = sqrt(kappa - k.k)) -- here for ALL the
// rows a sample brings (its value and its three derivatives; the reference appends value rows only, a factor that no
// longer belongs to the matrix f() and var() assume) and block-wise on the matrix cores: the leading 128 * floor(4 n_old /
// 128) rows and columns of the old factor stay, the rest is built and eliminated against them.  The model equals
// gpx_dgp_create on the concatenated data to rounding (its rows stand in another order; alpha, f and var come back in the
// reference's layout either way).  Fewer than 128 old rows, or GPX_DGP_APPEND=0: rebuilt from scratch.  On failure the
// model is unchanged.
extern "C" int gpx_dgp_add(gpx_dgp *g, size_t n_new, const double *x, const double *y, const double *z,
                           const double *target, const double *normals)
{
    if (!g || !g->m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (n_new == 0)
        return fail(GPX_E_EMPTY, "All input data is empty!");
    if (!x || !y || !z || !target)
        return fail(GPX_E_NULL, "Empty data pointer");
    const size_t n0 = (size_t)g->n_pts, n = n0 + n_new;
    if (n > ((size_t)1 << 18))
        return fail(GPX_E_BAD_ARG, "n too large");
    for (size_t i = 0; i < n_new; ++i)
        if (!std::isfinite(x[i]) || !std::isfinite(y[i]) || !std::isfinite(z[i]) || !std::isfinite(target[i]) ||
            (normals && (!std::isfinite(normals[3 * i]) || !std::isfinite(normals[3 * i + 1]) || !std::isfinite(normals[3 * i + 2]))))
            return fail(GPX_E_NAN_INPUT, "non-finite value in the training data");
    std::vector<double> ux(g->in_x), uy(g->in_y), uz(g->in_z), ut(g->in_t), un(g->in_n);
    ux.insert(ux.end(), x, x + n_new), uy.insert(uy.end(), y, y + n_new), uz.insert(uz.end(), z, z + n_new);
    ut.insert(ut.end(), target, target + n_new);
    un.resize(3 * n, 0.0);
    if (normals)
        std::copy(normals, normals + 3 * n_new, un.begin() + 3 * n0);
    const int app_env = gpxh::switches().dgp_append;  // GPX_DGP_APPEND=0: rebuild on the union (tests compare the two)
    const bool append = app_env != 0 && 4 * n0 >= (size_t)TILE;
    gpx_dgp *fresh = nullptr;
    gpx_options o = g->opt;
    o.device = g->m->device;
    int rc;
    {
        std::lock_guard<std::mutex> lk(g->m->mtx);  // the old factor is read
        rc = dgp_build(&g->kernel, g->noise, n, ux.data(), uy.data(), uz.data(), ut.data(), un.data(), &o, append ? g : nullptr, &fresh);
    }
    if (rc)
        return rc;
    std::swap(*g, *fresh);
    gpx_dgp_destroy(fresh);  // (the old model)
    return GPX_OK;
}

// ---- logLikelihoodGradient (GaussianProcess.h:387-410) ---------------------------------------------------------------
// grad_j = 1/2 sum_ab W_ab dK_ab / d theta_j with W = alpha alpha^T - K^-1 (:398-400; the reference's loop over the lower
// triangle with the diagonal halved, :402-408, is this sum), theta = CovSE's log hyper-parameters (log l, log sf) in
// getLogHyper()'s order (CovSE.h:108-118).  Taken over the whole 4n x 4n covariance, i.e. the exact gradient of
// GPX_DGP_FIELD_LOGLIK (the reference mixes the n x n value block with the 4n-long alpha; tests check against central
// differences of the likelihood).  On the device:
//   theta = log sf : dK = 2 (K - sn^2 I)  ->  grad = alpha.(y - sn^2 alpha) - (4n - sn^2 tr K^-1),  tr K^-1 = sum_m |X_m|^2 / D_m
//   theta = log l  : dK built as a matrix (dgp_dk_kernel), Z = dK X^T on the fp64 matrix cores (X = L^-1),
//                    grad = 1/2 alpha.dK alpha - 1/2 sum_m (X_m . Z_:,m) / D_m
// Other covariances: GPX_E_BAD_ARG (BaseCovFunc::grad is empty for them, Covs.h:172).
extern "C" int gpx_dgp_loglik_gradient(const gpx_dgp *cg, double *grad2)
{
    if (!cg || !cg->m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!grad2)
        return fail(GPX_E_NULL, "Empty output pointer");
    if (cg->cov.id != GPX_KERNEL_SE)
        return fail(GPX_E_BAD_ARG, "gpx_dgp_loglik_gradient: only GPX_KERNEL_SE has hyper-parameter derivatives (CovSE::grad)");
    gpx_dgp *g = const_cast<gpx_dgp *>(cg);
    gpx_model *m = g->m;
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int n = g->n_pts, n4 = 4 * n, np = m->npad;
    const int np_rows = std::min(np, (n4 + TILE - 1) / TILE * TILE);
    int rc;
    if ((rc = build_inverse(m)))
        return rc;
    const int nt32 = np_rows / 32;
    const size_t n_part = (size_t)2 * np_rows + (size_t)nt32 * nt32;
    // workspace from the pool (an optimise call comes here up to max_iter times); alpha in row order is uploaded from the host
    // copy the likelihood terms below use -- not read from the substitution's scratch, which any later solve may overwrite
    DevGuard gdK(nullptr, true), gZ(nullptr, true), gpart(nullptr, true);
    hipError_t he;
    if ((he = big_alloc(&gdK.p, sizeof(double) * (size_t)np_rows * np)) != hipSuccess ||
        (he = big_alloc(&gZ.p, sizeof(double) * (size_t)np_rows * np)) != hipSuccess ||
        (he = big_alloc(&gpart.p, sizeof(double) * (n_part + (size_t)np_rows))) != hipSuccess) {
        (void)hipGetLastError();
        return fail(he == hipErrorOutOfMemory ? GPX_E_OOM : GPX_E_HIP, std::string("gpx_dgp_loglik_gradient: ") + hipGetErrorString(he));
    }
    double *dK = (double *)gdK.p, *Z = (double *)gZ.p, *part = (double *)gpart.p, *d_arows = part + n_part;
    auto release = [&] {
        (void)hipStreamSynchronize(s);  // nothing may still use the buffers when the guards park them
    };
    {
        std::vector<double> ar((size_t)np_rows, 0.0);
        for (int i = 0; i < n4; ++i)
            ar[i] = g->h_alpha_rows[i];
        he = hipMemcpy(d_arows, ar.data(), sizeof(double) * (size_t)np_rows, hipMemcpyHostToDevice);
        if (he != hipSuccess)
            return fail(GPX_E_HIP, std::string("gpx_dgp_loglik_gradient: ") + hipGetErrorString(he));
    }
    double *p_quad = part, *p_trk = part + np_rows, *p_tr = part + 2 * (size_t)np_rows;
    hipLaunchKernelGGL(gpx::dgp_dk_kernel, dim3(np_rows / TILE, np_rows / TILE), dim3(256), 0, s, g->cov, np, g->d_row, g->d_x,
                       g->d_y, g->d_z, dK);
    hipLaunchKernelGGL(gpx::dgp_quad_rows_kernel, dim3(n4), dim3(256), 0, s, n4, np, dK, (const double *)d_arows, p_quad);
    hipLaunchKernelGGL(gpx::dgp_trkinv_rows_kernel, dim3(n4), dim3(256), 0, s, np, (const double *)m->X,
                       (const double *)m->t_dinv, p_trk);
    GemmArgs a;  // Z[a][m] = sum_b dK[a][b] X[m][b]
    a.A = dK, a.lda = np;
    a.B = m->X, a.ldb = np;
    a.C = Z, a.ldc = np;
    a.M = np_rows, a.N = np_rows, a.K = np_rows;
    a.b_lower = 1;
    launch_gemm(GPX_PREC_F64, a, s);
    hipLaunchKernelGGL(gpx::dgp_trace_tiles_kernel, dim3(nt32, nt32), dim3(256), 0, s, np, (const double *)m->X, Z,
                       (const double *)m->t_dinv, p_tr);
    std::vector<double> h(n_part);
    he = hipMemcpyAsync(h.data(), part, sizeof(double) * n_part, hipMemcpyDeviceToHost, s);
    if (he == hipSuccess)
        he = hipStreamSynchronize(s);
    if (he == hipSuccess)
        he = hipGetLastError();
    release();
    if (he != hipSuccess)
        return fail(GPX_E_HIP, std::string("gpx_dgp_loglik_gradient: ") + hipGetErrorString(he));
    double quad_l = 0, trkinv = 0, tr_l = 0, ya = 0, aa = 0;
    for (int i = 0; i < n4; ++i) {
        quad_l += h[i];
        trkinv += h[(size_t)np_rows + i];
        ya += g->h_y[i] * g->h_alpha_rows[i];
        aa += g->h_alpha_rows[i] * g->h_alpha_rows[i];
    }
    // rows / columns of the identity padding inside the last 128-block: X = I, Z = 0 there, nothing to leave out
    for (size_t t = 0; t < (size_t)nt32 * nt32; ++t)
        tr_l += h[2 * (size_t)np_rows + t];
    grad2[0] = 0.5 * quad_l - 0.5 * tr_l;
    grad2[1] = (ya - g->sn2 * aa) - ((double)n4 - g->sn2 * trkinv);
    return GPX_OK;
}

// the model again on its own samples with another kernel (the optimiser's setLogHyper + compute()); on failure unchanged
static int dgp_refit(gpx_dgp *g, const gpx_kernel &k)
{
    gpx_dgp *fresh = nullptr;
    gpx_options o = g->opt;
    o.device = g->m->device;
    const int rc = gpx_dgp_create(&k, g->noise, (size_t)g->n_pts, g->in_x.data(), g->in_y.data(), g->in_z.data(),
                                  g->in_t.data(), g->in_n.data(), &o, &fresh);
    if (rc)
        return rc;
    std::swap(*g, *fresh);
    gpx_dgp_destroy(fresh);  // (the old model)
    return GPX_OK;
}

// ---- Optimisation (RProp), GaussianProcess.h:41-160 ---------------------------------------------------------------------
extern "C" void gpx_rprop_default(gpx_rprop *d)
{
    if (!d)
        return;
    d->delta0 = 0.1, d->delta_min = 1e-6, d->delta_max = 50, d->eta_minus = 0.5, d->eta_plus = 1.2;  // Desc::setToDefault :64-73
    d->eps_stop = 1e-4;
    d->max_iter = 100;
}

// Optimisation::find (:86-122) statement by statement, in CovSE's log hyper-parameters p = (log l, log sf): the step
// sizes adapt on the sign of gradOld .* grad, a sign change zeroes that component of the gradient, the norm test comes
// AFTER the step (so the step that meets it is never applied), the likelihood of every applied step is compared with
// the best one and the model ends on the best parameters.  A step onto parameters whose covariance does not factorise
// ends the search there (the reference's llt() would carry NaNs on); the model is left on the best parameters seen.
extern "C" int gpx_dgp_optimise(gpx_dgp *g, const gpx_rprop *desc, gpx_rprop_result *res)
{
    if (!g || !g->m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (g->cov.id != GPX_KERNEL_SE)
        return fail(GPX_E_BAD_ARG, "gpx_dgp_optimise: only GPX_KERNEL_SE has hyper-parameter derivatives (CovSE::grad)");
    gpx_rprop d;
    gpx_rprop_default(&d);
    if (desc)
        d = *desc;
    {
        auto pos = [](double v) { return std::isfinite(v) && v > 0.0; };
        if (!pos(d.delta0) || !pos(d.delta_min) || !pos(d.delta_max) || !pos(d.eta_minus) || !pos(d.eta_plus) ||
            d.eta_minus > 1.0 || d.eta_plus < 1.0 || d.delta_min > d.delta_max || !std::isfinite(d.eps_stop) || d.eps_stop < 0.0)
            return fail(GPX_E_BAD_ARG, "gpx_dgp_optimise: step sizes must be finite and positive, delta_min <= delta_max, "
                                       "eta_minus <= 1 <= eta_plus, eps_stop >= 0");
    }
    auto sign = [](double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); };
    auto kernel_of = [](const double *p) {
        gpx_kernel k{};
        k.id = GPX_KERNEL_SE;
        k.p[0] = std::exp(p[1]);  // sf  (sf2 = exp(2 p1), CovSE.h:111)
        k.p[1] = std::exp(p[0]);  // l   (ell = exp(p0),   CovSE.h:110)
        return k;
    };
    double delta[2] = {d.delta0, d.delta0}, grad_old[2] = {0, 0};
    double params[2] = {std::log(g->kernel.p[1]), std::log(g->kernel.p[0])};
    double best_params[2] = {params[0], params[1]};
    double best = -INFINITY;
    uint64_t done = 0;
    int rc = GPX_OK;
    bool at_best = true;  // the model currently holds best_params
    for (uint64_t i = 0; i < d.max_iter; ++i) {
        double grad[2];
        if ((rc = gpx_dgp_loglik_gradient(g, grad)))
            break;
        grad[0] = -grad[0], grad[1] = -grad[1];
        for (int j = 0; j < 2; ++j) {
            grad_old[j] *= grad[j];
            if (grad_old[j] > 0) {
                delta[j] = std::min(delta[j] * d.eta_plus, d.delta_max);
            } else if (grad_old[j] < 0) {
                delta[j] = std::max(delta[j] * d.eta_minus, d.delta_min);
                grad[j] = 0;
            }
            params[j] += -sign(grad[j]) * delta[j];
        }
        grad_old[0] = grad[0], grad_old[1] = grad[1];
        if (std::sqrt(grad[0] * grad[0] + grad[1] * grad[1]) < d.eps_stop)
            break;
        if ((rc = dgp_refit(g, kernel_of(params)))) {
            // parameters whose covariance does not factorise (or does not fit): the search ends HERE, as documented -- the
            // model is unchanged by the failed refit (dgp_refit), the result is the best point met, the call succeeds
            if (rc == GPX_E_SINGULAR || rc == GPX_E_OOM)
                rc = GPX_OK;
            break;
        }
        at_best = false;
        ++done;
        if (g->loglik > best) {
            best = g->loglik;
            best_params[0] = params[0], best_params[1] = params[1];
            at_best = true;
        }
    }
    const std::string keep = g_err;
    if (!at_best) {
        const int rc2 = dgp_refit(g, kernel_of(best_params));
        if (!rc)
            rc = rc2;
        else
            g_err = keep;
    }
    if (res) {
        res->loghyper[0] = best_params[0], res->loghyper[1] = best_params[1];
        res->loglik = done ? best : g->loglik;
        res->iterations = done;
    }
    return rc;
}

extern "C" int gpx_dgp_evaluate(const gpx_dgp *cg, size_t nq, const double *qx, const double *qy, const double *qz,
                                double *f4, double *var)
{
    if (!cg || !cg->m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (nq == 0)
        return fail(GPX_E_EMPTY, "All input data is empty!");
    if (!qx || !qy || !qz)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (!f4)
        return fail(GPX_E_NULL, "Empty output pointer");
    gpx_dgp *g = const_cast<gpx_dgp *>(cg);
    gpx_model *m = g->m;
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int n = g->n_pts, n4 = 4 * n, np = m->npad;
    int rc;
    // device staging: qx qy qz | out (4 nq) | var (nq)
    if ((rc = ensure(m, (void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * nq * 8)))
        return rc;
    double *d = m->ws_host_io, *dq[3] = {d, d + nq, d + 2 * nq}, *dout = d + 3 * nq, *dvar = d + 7 * nq;
    HIPCHK(hipMemcpyAsync(dq[0], qx, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dq[1], qy, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dq[2], qz, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    (void)hipEventRecord(m->ev[EV_M0], s);
    hipLaunchKernelGGL(gpx::dgp_predict_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, g->cov, n, g->d_x,
                       g->d_y, g->d_z, g->d_alpha, (long)nq, dq[0], dq[1], dq[2], dout);
    (void)hipEventRecord(m->ev[EV_M1], s);
    if (var) {
        if ((rc = build_inverse(m)))  // X = L^-1 by recursive doubling on the GEMM core, once per model
            return rc;
        const size_t qb = (size_t)std::min<size_t>((size_t)m->qbatch, ((nq + TILE - 1) / TILE) * TILE);
        const int np_rows = std::min(np, (n4 + TILE - 1) / TILE * TILE);
        if ((rc = ensure(m, &m->ws_kqp, &m->ws_kqp_bytes, sizeof(double) * qb * np)) ||
            (rc = ensure(m, &m->ws_partial, &m->ws_partial_bytes, sizeof(double) * qb * m->nblk)))
            return rc;
        for (size_t q0 = 0; q0 < nq; q0 += qb) {
            const size_t nv = std::min(qb, nq - q0), ntile = ((nv + TILE - 1) / TILE) * TILE;
            hipLaunchKernelGGL(gpx::dgp_kstar_kernel, dim3(np_rows / TILE, (unsigned)(ntile / TILE)), dim3(256), 0, s, g->cov,
                               np, g->d_row, g->d_x, g->d_y, g->d_z, (long)nv, dq[0] + q0, dq[1] + q0, dq[2] + q0,
                               (double *)m->ws_kqp);
            GemmArgs a;  // partial[mt][q] = sum_rows (X * k_star^T)^2 / D
            a.A = m->X, a.lda = np;
            a.B = m->ws_kqp, a.ldb = np;
            a.M = np_rows, a.N = (int)ntile, a.K = np;
            a.a_lower = 1;
            a.epi = EPI_COLSQ;
            a.rowweight = m->t_dinv;
            a.partial = m->ws_partial, a.ldp = (long)qb;
            a.cfg = 6;  // the one-wave fp64 tile (gpx_vargemm.hip); launch_gemm falls back to the LDS tile if it does not fit
            launch_gemm(GPX_PREC_F64, a, s);
            launch_var_finish(GPX_PREC_F64, g->k0, np_rows / gemm_rows_per_partial(GPX_PREC_F64, a), (long)qb, m->ws_partial,
                              (long)nv, dvar + q0, s);
        }
    }
    (void)hipEventRecord(m->ev[EV_V1], s);
    HIPCHK(hipMemcpyAsync(f4, dout, sizeof(double) * 4 * nq, hipMemcpyDeviceToHost, s));
    if (var)
        HIPCHK(hipMemcpyAsync(var, dvar, sizeof(double) * nq, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
    float ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_M0], m->ev[EV_M1]) == hipSuccess)
        m->stats.t_mean_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_M1], m->ev[EV_V1]) == hipSuccess)
        m->stats.t_var_ms = var ? ms : 0.0;
    return GPX_OK;
}

extern "C" int gpx_dgp_get(const gpx_dgp *g, int field, void *dst, size_t bytes)
{
    if (!g || !g->m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!dst)
        return fail(GPX_E_NULL, "Empty output pointer");
    auto need = [&](size_t b) { return bytes >= b ? GPX_OK : fail(GPX_E_SIZE_MISMATCH, "destination too small"); };
    int rc;
    switch (field) {
    case GPX_DGP_FIELD_N:
        if ((rc = need(sizeof(int64_t))))
            return rc;
        *(int64_t *)dst = g->n_pts;
        return GPX_OK;
    case GPX_DGP_FIELD_ALPHA:
        if ((rc = need(sizeof(double) * g->h_alpha.size())))
            return rc;
        std::memcpy(dst, g->h_alpha.data(), sizeof(double) * g->h_alpha.size());
        return GPX_OK;
    case GPX_DGP_FIELD_LOGLIK:
        if ((rc = need(sizeof(double))))
            return rc;
        *(double *)dst = g->loglik;
        return GPX_OK;
    case GPX_DGP_FIELD_APPENDED_FROM:
        if ((rc = need(sizeof(int64_t))))
            return rc;
        *(int64_t *)dst = g->appended_from;
        return GPX_OK;
    case GPX_DGP_FIELD_STATS:
        if ((rc = need(sizeof(gpx_stats))))
            return rc;
        std::memcpy(dst, &g->m->stats, sizeof(gpx_stats));
        return GPX_OK;
    default:
        return fail(GPX_E_BAD_ARG, "unknown field");
    }
}
