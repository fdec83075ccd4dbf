/*
 * gp_oracle.c -- CPU fp64 restatement of the reference GP-regression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product path (libgpx.so) never links, loads or calls it.
 *
 * PARITY: PINNED for the covariance functions, UNPINNED for the rest.
 *   pinned   -- orc_k / orc_kdiff / orc_kdiffdiff for Gaussian, Laplace and ThinPlate are bit-identical to the
 *               reference's own classes: kernels/{gaussian,laplace,thin_plate}.hpp are plain C++ over <cmath>
 *               and are compiled from /root/reference into oracle/_ref/libref_kernels.so (oracle/Makefile,
 *               wrapper oracle/ref_kernels_wrap.cpp); their outputs are committed as tests/golden/ref_kernels.npz
 *               (tests/golden/make_ref_kernel_golden.py) and checked by tests/test_reference_pin.py.
 *   unpinned -- everything that goes through Eigen (distance matrix, LDLT, solves, tangent basis): PARITY
 *               UNPINNED.  The reference's own tests hold no golden values and no assertions for this path
 *               (tests/test_gaussian.cpp, tests/test_gp.cpp and tests/test_eigen.cpp only print), and
 *               gp_regressor.hpp cannot be built here: its arithmetic lives in Eigen 3 (find_package(Eigen),
 *               CMakeLists.txt:25, no version pin, not vendored, not installed, no network).
 * This file restates, function by function,
 *     include/gp_regression/gp_regressor.hpp      (create / evaluate x4 / update)
 *     include/gp_regression/kernels/{gaussian,laplace,thin_plate}.hpp
 *     matlab_src/test_gp_regression_3Dsurf.m:117-123   (Matern closed forms)
 *     include/atlas/atlas.hpp:201-276               (AtlasBase::project)
 * and, for Eigen::LDLT (gp_regressor.hpp:81,:161-163), the published algorithm
 * of Eigen 3.2.x LDLT.h (ldlt_inplace<Lower>::unblocked + LDLT::solve), written
 * from its public description.  What pins the unpinned part instead: analytic known-answer
 * tests, GP identities and an independent NumPy/SciPy computation committed as
 * fixtures under tests/golden/ (tests/golden/make_golden.py).
 *
 * All arithmetic is IEEE double, as in the reference (std::vector<double>,
 * Eigen::MatrixXd).  Matrices are column-major like Eigen's default.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_GAUSSIAN 0
#define ORC_LAPLACE 1
#define ORC_THINPLATE 2
#define ORC_MATERN32 3
#define ORC_MATERN52 4

/* distance formulation */
#define ORC_DIST_DIRECT 0    /* sqrt(dx^2+dy^2+dz^2): the build's documented deviation */
#define ORC_DIST_EXPANSION 1 /* gp_regressor.hpp:548-557 literally; may yield NaN */

typedef struct {
    int id;
    double p0; /* sigma (Gaussian/Laplace/Matern)  | R (ThinPlate) */
    double p1; /* length (Gaussian/Laplace/Matern) | unused        */
} orc_kernel;

/* ---- K1..K5: scalar radial kernels on the un-squared distance -------------------------- */

/* kernels/gaussian.hpp:15-20, laplace.hpp:37-42, thin_plate.hpp:12-15,
 * matlab_src/test_gp_regression_3Dsurf.m:117-119 (Matern-3/2), :121-123 (Matern-5/2). */
double orc_k(const orc_kernel *k, double d)
{
    switch (k->id) {
    case ORC_GAUSSIAN: {
        double sigma2 = k->p0 * k->p0;
        double inv_length2 = 1.0 / (k->p1 * k->p1);
        double power = -1 * d * inv_length2; /* sic: un-squared distance */
        return sigma2 * exp(power);
    }
    case ORC_LAPLACE: {
        double inv_length = 1.0 / k->p1;
        double power = -1 * d * inv_length;
        return 2 * k->p0 * exp(power); /* sic: amplitude 2*sigma */
    }
    case ORC_THINPLATE: {
        double R = k->p0, R3 = R * R * R;
        return 2 * d * d * d - 3 * R * d * d + R3;
    }
    case ORC_MATERN32: {
        double s = sqrt(3.0) * d / k->p1;
        return k->p0 * k->p0 * (1 + s) * exp(-s);
    }
    case ORC_MATERN52: {
        double s = sqrt(5.0) * d / k->p1;
        return k->p0 * k->p0 * (1 + s + (5 * d * d) / (3 * k->p1 * k->p1)) * exp(-s);
    }
    }
    return NAN;
}

/* computediff: gaussian.hpp:22-27 (k'(d)), laplace.hpp:44-49 (k'(d)), thin_plate.hpp:17-20
 * (k'(d)/d); the new Matern kernels use k'(d)/d. */
double orc_kdiff(const orc_kernel *k, double d)
{
    switch (k->id) {
    case ORC_GAUSSIAN:
        return -1 * (1.0 / (k->p1 * k->p1)) * orc_k(k, d);
    case ORC_LAPLACE:
        return -1 * (1.0 / k->p1) * orc_k(k, d);
    case ORC_THINPLATE:
        return -6 * (k->p0 - d);
    case ORC_MATERN32: {
        double s = sqrt(3.0) * d / k->p1;
        return -3 * k->p0 * k->p0 / (k->p1 * k->p1) * exp(-s);
    }
    case ORC_MATERN52: {
        double s = sqrt(5.0) * d / k->p1;
        return -(5 * k->p0 * k->p0 / (3 * k->p1 * k->p1)) * (1 + s) * exp(-s);
    }
    }
    return NAN;
}

/* computediffdiff: 0 in every reference kernel (gaussian.hpp:29-34 etc.). */
double orc_kdiffdiff(const orc_kernel *k, double d)
{
    (void)k;
    (void)d;
    return 0.0;
}

/* ---- F1: buildEuclideanDistanceMatrix, gp_regressor.hpp:548-557 ------------------------- */
/* D (m x n, column-major) = distance between A_i and B_j. */
void orc_dist_matrix(int mode, int m, int n, const double *ax, const double *ay, const double *az,
                     const double *bx, const double *by, const double *bz, double *D)
{
    if (mode == ORC_DIST_EXPANSION) {
        /* D = -2*A*B^T; D.colwise() += rowsum(A.A); D.rowwise() += rowsum(B.B)^T; sqrt */
        for (int j = 0; j < n; ++j) {
            double bb = bx[j] * bx[j] + by[j] * by[j] + bz[j] * bz[j];
            for (int i = 0; i < m; ++i) {
                double aa = ax[i] * ax[i] + ay[i] * ay[i] + az[i] * az[i];
                double ab = ax[i] * bx[j] + ay[i] * by[j] + az[i] * bz[j];
                double v = -2 * ab;
                v += aa;
                v += bb;
                D[(size_t)j * m + i] = sqrt(v); /* NaN when v < 0, as in the reference */
            }
        }
    } else {
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i) {
                double dx = ax[i] - bx[j], dy = ay[i] - by[j], dz = az[i] - bz[j];
                D[(size_t)j * m + i] = sqrt(dx * dx + dy * dy + dz * dz);
            }
    }
}

/* ---- F4/F5: Eigen::LDLT<MatrixXd> (gp_regressor.hpp:161-163) ---------------------------- */
/* In-place, lower triangle, column-major, left-looking, "diagonal pivoting" on the stored
 * diagonal of the not-yet-processed block (Eigen 3.2.x ldlt_inplace<Lower>::unblocked).
 * On exit: strict lower = L (unit), diagonal = D, transp[k] = row swapped with k at step k.
 * Returns 0, or k+1 if the factorisation stopped early at step k (pivot below cutoff). */
int orc_ldlt(int n, double *A, int *transp)
{
    double cutoff = 0.0;
    double *temp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int ret = 0;
#define M(i, j) A[(size_t)(j) * n + (i)]
    for (int k = 0; k < n; ++k) {
        /* largest |diagonal| in the remaining block (first maximum, like maxCoeff) */
        int big = k;
        double bigv = fabs(M(k, k));
        for (int i = k + 1; i < n; ++i)
            if (fabs(M(i, i)) > bigv) {
                bigv = fabs(M(i, i));
                big = i;
            }
        if (k == 0)
            cutoff = fabs(2.220446049250313e-16 * bigv);
        if (bigv < cutoff) {
            for (int i = k; i < n; ++i)
                transp[i] = i;
            ret = k + 1;
            break;
        }
        transp[k] = big;
        if (k != big) {
            /* symmetric swap touching only the lower triangle */
            for (int j = 0; j < k; ++j) {
                double t = M(k, j);
                M(k, j) = M(big, j);
                M(big, j) = t;
            }
            for (int i = big + 1; i < n; ++i) {
                double t = M(i, k);
                M(i, k) = M(i, big);
                M(i, big) = t;
            }
            {
                double t = M(k, k);
                M(k, k) = M(big, big);
                M(big, big) = t;
            }
            for (int i = k + 1; i < big; ++i) {
                double t = M(i, k);
                M(i, k) = M(big, i);
                M(big, i) = t;
            }
        }
        int rs = n - k - 1;
        if (k > 0) {
            double acc = 0.0;
            for (int j = 0; j < k; ++j) {
                temp[j] = M(j, j) * M(k, j); /* D(0:k) .* A10^T */
                acc += M(k, j) * temp[j];
            }
            M(k, k) -= acc;
            if (rs > 0) {
                /* A21 -= A20 * temp : column-major saxpy over columns j<k */
                double *a21 = &M(k + 1, k);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (rs > 512)
#endif
                for (int ib = 0; ib < rs; ib += 256) {
                    int ie = ib + 256 < rs ? ib + 256 : rs;
                    for (int j = 0; j < k; ++j) {
                        const double *a20 = &M(k + 1, j);
                        double t = temp[j];
                        for (int i = ib; i < ie; ++i)
                            a21[i] -= a20[i] * t;
                    }
                }
            }
        }
        if (rs > 0 && fabs(M(k, k)) > cutoff) {
            double d = M(k, k);
            for (int i = k + 1; i < n; ++i)
                M(i, k) /= d;
        }
    }
#undef M
    free(temp);
    return ret;
}

/* LDLT::solve for nrhs right-hand sides (column-major n x nrhs, in place):
 * x = P^T L^-T D^+ L^-1 P b, with Eigen 3.2.x's pseudo-inverse rule for D. */
void orc_ldlt_solve(int n, const double *A, const int *transp, int nrhs, double *B)
{
#define M(i, j) A[(size_t)(j) * n + (i)]
    double dmax = 0.0;
    for (int i = 0; i < n; ++i)
        if (fabs(M(i, i)) > dmax)
            dmax = fabs(M(i, i));
    double tol = dmax * 2.220446049250313e-16;
    if (tol < 1.0 / 1.7976931348623157e308)
        tol = 1.0 / 1.7976931348623157e308;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) if (nrhs > 8)
#endif
    for (int r = 0; r < nrhs; ++r) {
        double *b = B + (size_t)r * n;
        for (int k = 0; k < n; ++k) /* b = P b */
            if (transp[k] != k) {
                double t = b[k];
                b[k] = b[transp[k]];
                b[transp[k]] = t;
            }
        for (int j = 0; j < n; ++j) { /* L y = b (unit lower), column-oriented */
            double bj = b[j];
            if (bj != 0.0)
                for (int i = j + 1; i < n; ++i)
                    b[i] -= M(i, j) * bj;
        }
        for (int i = 0; i < n; ++i) /* D^+ */
            b[i] = fabs(M(i, i)) > tol ? b[i] / M(i, i) : 0.0;
        for (int j = n - 1; j >= 0; --j) { /* L^T x = y */
            double acc = b[j];
            for (int i = j + 1; i < n; ++i)
                acc -= M(i, j) * b[i];
            b[j] = acc;
        }
        for (int k = n - 1; k >= 0; --k) /* x = P^T x */
            if (transp[k] != k) {
                double t = b[k];
                b[k] = b[transp[k]];
                b[transp[k]] = t;
            }
    }
#undef M
}

/* ---- T2: Model (gp_regressor.hpp:71-87) ------------------------------------------------- */
typedef struct {
    orc_kernel kern;
    int dist_mode;
    int n;
    int has_s2;
    double R;              /* gp->R, :135 */
    double *px, *py, *pz;  /* gp->P */
    double *Y, *S2;        /* gp->Y, gp->S2 */
    double *Kpp;           /* kernel matrix after :144-159 (column-major) */
    double *ldlt;          /* cholesker's private copy */
    int *transp;
    double *alpha;         /* :163 */
    double *normals;       /* n x 3 row-major, only with normals (:166-181) */
    int ldlt_info;
} orc_model;

static double *dup_vec(const double *v, int n)
{
    double *r = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (v && n > 0)
        memcpy(r, v, sizeof(double) * (size_t)n);
    return r;
}

void orc_free(orc_model *m)
{
    if (!m)
        return;
    free(m->px);
    free(m->py);
    free(m->pz);
    free(m->Y);
    free(m->S2);
    free(m->Kpp);
    free(m->ldlt);
    free(m->transp);
    free(m->alpha);
    free(m->normals);
    free(m);
}

static void factor_and_solve(orc_model *m)
{
    int n = m->n;
    size_t nn = (size_t)n * n;
    free(m->ldlt);
    free(m->transp);
    free(m->alpha);
    m->ldlt = (double *)malloc(sizeof(double) * nn);
    memcpy(m->ldlt, m->Kpp, sizeof(double) * nn);
    m->transp = (int *)malloc(sizeof(int) * (size_t)n);
    m->ldlt_info = orc_ldlt(n, m->ldlt, m->transp); /* :161-162 */
    m->alpha = dup_vec(m->Y, n);
    orc_ldlt_solve(n, m->ldlt, m->transp, 1, m->alpha); /* :163 */
}

/* F2/F3/F4/F5/F6: GPRegressor::create<withNormals>, gp_regressor.hpp:110-182.
 * sigma2 may be NULL (== empty vector, :154).  Normals are zero-initialised before the
 * accumulation of :172 (the reference accumulates into un-initialised storage). */
orc_model *orc_create(const orc_kernel *kern, int n, const double *x, const double *y, const double *z,
                      const double *label, const double *sigma2, int with_normals, int dist_mode)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    m->kern = *kern;
    m->dist_mode = dist_mode;
    m->n = n;
    m->has_s2 = sigma2 != NULL;
    m->px = dup_vec(x, n);
    m->py = dup_vec(y, n);
    m->pz = dup_vec(z, n);
    m->Y = dup_vec(label, n);
    m->S2 = dup_vec(sigma2, n);
    size_t nn = (size_t)n * n;
    m->Kpp = (double *)malloc(sizeof(double) * (nn ? nn : 1));
    orc_dist_matrix(dist_mode, n, n, x, y, z, x, y, z, m->Kpp); /* :132 */
    double R = -INFINITY;                                       /* :135 maxCoeff */
    for (size_t i = 0; i < nn; ++i)
        if (m->Kpp[i] > R)
            R = m->Kpp[i];
    m->R = R;
    double *Kdiff = NULL;
    if (with_normals)
        Kdiff = (double *)malloc(sizeof(double) * nn);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n > 256)
#endif
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) { /* :144-159 */
            size_t ij = (size_t)j * n + i;
            double d = m->Kpp[ij];
            if (with_normals)
                Kdiff[ij] = orc_kdiff(kern, d);
            if (sigma2 && i == j)
                m->Kpp[ij] = orc_k(kern, d) + sigma2[i];
            else
                m->Kpp[ij] = orc_k(kern, d);
        }
    factor_and_solve(m);
    if (with_normals) { /* :166-181 */
        m->normals = (double *)calloc((size_t)n * 3, sizeof(double));
        for (int i = 0; i < n; ++i) {
            double gx = 0, gy = 0, gz = 0;
            for (int j = 0; j < n; ++j) {
                double w = m->alpha[j] * Kdiff[(size_t)j * n + i];
                gx += w * (x[i] - x[j]);
                gy += w * (y[i] - y[j]);
                gz += w * (z[i] - z[j]);
            }
            double nrm = sqrt(gx * gx + gy * gy + gz * gz);
            if (nrm > 0) { /* Eigen normalize() */
                gx /= nrm;
                gy /= nrm;
                gz /= nrm;
            }
            m->normals[3 * i + 0] = gx;
            m->normals[3 * i + 1] = gy;
            m->normals[3 * i + 2] = gz;
        }
        free(Kdiff);
    }
    return m;
}

/* F11: GPRegressor::update, gp_regressor.hpp:367-479: append, rebuild Kpp blocks, refactor
 * from scratch (:457-459).  R is NOT refreshed (:454-455).  sigma2 NULL => no noise on the
 * new diagonal (the reference's block assignment would mis-size). */
int orc_update(orc_model *m, int nn_, const double *x, const double *y, const double *z, const double *label,
               const double *sigma2)
{
    int p = m->n, n = nn_, t = p + n;
    double *nx = (double *)malloc(sizeof(double) * t), *ny = (double *)malloc(sizeof(double) * t),
           *nz = (double *)malloc(sizeof(double) * t), *nY = (double *)malloc(sizeof(double) * t),
           *nS = (double *)calloc(t, sizeof(double));
    memcpy(nx, m->px, sizeof(double) * p);
    memcpy(ny, m->py, sizeof(double) * p);
    memcpy(nz, m->pz, sizeof(double) * p);
    memcpy(nY, m->Y, sizeof(double) * p);
    if (m->has_s2)
        memcpy(nS, m->S2, sizeof(double) * p);
    memcpy(nx + p, x, sizeof(double) * n);
    memcpy(ny + p, y, sizeof(double) * n);
    memcpy(nz + p, z, sizeof(double) * n);
    memcpy(nY + p, label, sizeof(double) * n);
    if (sigma2)
        memcpy(nS + p, sigma2, sizeof(double) * n);
    double *K = (double *)malloc(sizeof(double) * (size_t)t * t);
    for (int j = 0; j < p; ++j) /* conservativeResize keeps the old block (:442) */
        memcpy(K + (size_t)j * t, m->Kpp + (size_t)j * p, sizeof(double) * p);
    double *Kpn = (double *)malloc(sizeof(double) * (size_t)p * n);
    double *Knn = (double *)malloc(sizeof(double) * (size_t)n * n);
    orc_dist_matrix(m->dist_mode, n, n, x, y, z, x, y, z, Knn);             /* :397 */
    orc_dist_matrix(m->dist_mode, p, n, m->px, m->py, m->pz, x, y, z, Kpn); /* :398 */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < p; ++i) { /* :408-421, :444-445 */
            double v = orc_k(&m->kern, Kpn[(size_t)j * p + i]);
            K[(size_t)(p + j) * t + i] = v;
            K[(size_t)i * t + (p + j)] = v;
        }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) { /* :424-440, :443 */
            double v = orc_k(&m->kern, Knn[(size_t)j * n + i]);
            if (sigma2 && i == j)
                v += sigma2[i];
            K[(size_t)(p + j) * t + (p + i)] = v;
        }
    free(Kpn);
    free(Knn);
    free(m->px);
    free(m->py);
    free(m->pz);
    free(m->Y);
    free(m->S2);
    free(m->Kpp);
    m->px = nx;
    m->py = ny;
    m->pz = nz;
    m->Y = nY;
    m->S2 = nS;
    m->Kpp = K;
    m->n = t;
    factor_and_solve(m);
    return 0;
}

/* computeTangentBasis, gp_regressor.hpp:29-44.  isApprox(UnitX,1e-3):
 * ||N-UnitX||^2 <= 1e-6 * min(||N||^2, 1). */
void orc_tangent_basis(const double g[3], double N[3], double Tx[3], double Ty[3])
{
    double nrm = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    for (int c = 0; c < 3; ++c)
        N[c] = nrm > 0 ? g[c] / nrm : g[c];
    double nn = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
    double diff2 = (N[0] - 1) * (N[0] - 1) + N[1] * N[1] + N[2] * N[2];
    double mn = nn < 1.0 ? nn : 1.0;
    int approx_x = diff2 <= 1e-3 * 1e-3 * mn;
    double e[3] = {0, 0, 0};
    e[approx_x ? 1 : 0] = 1.0;
    double dot = N[0] * e[0] + N[1] * e[1] + N[2] * e[2];
    for (int c = 0; c < 3; ++c)
        Tx[c] = e[c] - N[c] * dot;
    double tn = sqrt(Tx[0] * Tx[0] + Tx[1] * Tx[1] + Tx[2] * Tx[2]);
    if (tn > 0)
        for (int c = 0; c < 3; ++c)
            Tx[c] /= tn;
    Ty[0] = N[1] * Tx[2] - N[2] * Tx[1];
    Ty[1] = N[2] * Tx[0] - N[0] * Tx[2];
    Ty[2] = N[0] * Tx[1] - N[1] * Tx[0];
    tn = sqrt(Ty[0] * Ty[0] + Ty[1] * Ty[1] + Ty[2] * Ty[2]);
    if (tn > 0)
        for (int c = 0; c < 3; ++c)
            Ty[c] /= tn;
}

/* F7..F10: GPRegressor::evaluate, gp_regressor.hpp:332-357 (f), :282-324 (f,v),
 * :222-273 (f,v,grad), :194-212 (+Tx,Ty).  Any output pointer may be NULL.
 * grad/tx/ty are nq x 3 row-major.  Variance is the diagonal of Kqq - Kqp*solve(Kpq)
 * (:316-319), computed one query column at a time so the Nq x Nq matrix never exists
 * (SURVEY D6); orc_evaluate_fullcov below is the literal form for small Nq.
 * The gradient starts from zero. */
void orc_evaluate(const orc_model *m, int nq, const double *qx, const double *qy, const double *qz, double *f,
                  double *v, double *grad, double *tx, double *ty)
{
    int n = m->n;
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        double *kq = (double *)malloc(sizeof(double) * (size_t)n);
        double *sol = (double *)malloc(sizeof(double) * (size_t)n);
        double zero = 0.0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int i = 0; i < nq; ++i) {
            double gx = 0, gy = 0, gz = 0, fi = 0;
            for (int j = 0; j < n; ++j) {
                double d;
                orc_dist_matrix(m->dist_mode, 1, 1, qx + i, qy + i, qz + i, m->px + j, m->py + j, m->pz + j, &d);
                if (grad || tx || ty) { /* :247 */
                    double w = m->alpha[j] * orc_kdiff(&m->kern, d);
                    gx += w * (qx[i] - m->px[j]);
                    gy += w * (qy[i] - m->py[j]);
                    gz += w * (qz[i] - m->pz[j]);
                }
                kq[j] = orc_k(&m->kern, d); /* :248 / :303 / :351 */
                fi += kq[j] * m->alpha[j];  /* :252 / :305 / :353 */
            }
            if (f)
                f[i] = fi;
            if (v) {
                double dqq;
                orc_dist_matrix(m->dist_mode, 1, 1, qx + i, qy + i, qz + i, qx + i, qy + i, qz + i, &dqq);
                if (m->dist_mode == ORC_DIST_DIRECT)
                    dqq = zero;
                double kqq = orc_k(&m->kern, dqq); /* :309-312 diagonal */
                memcpy(sol, kq, sizeof(double) * (size_t)n);
                orc_ldlt_solve(n, m->ldlt, m->transp, 1, sol); /* :316 */
                double acc = 0;
                for (int j = 0; j < n; ++j)
                    acc += kq[j] * sol[j]; /* :318 */
                v[i] = kqq - acc;          /* :319 */
            }
            if (grad) {
                grad[3 * i + 0] = gx;
                grad[3 * i + 1] = gy;
                grad[3 * i + 2] = gz;
            }
            if (tx || ty) { /* :204-211 */
                double g[3] = {gx, gy, gz}, N[3], Tx[3], Ty[3];
                orc_tangent_basis(g, N, Tx, Ty);
                for (int c = 0; c < 3; ++c) {
                    if (tx)
                        tx[3 * i + c] = Tx[c];
                    if (ty)
                        ty[3 * i + c] = Ty[c];
                }
            }
        }
        free(kq);
        free(sol);
    }
}

/* Literal gp_regressor.hpp:307-319: full Nq x Nq covariance, then its diagonal.
 * Only for small nq (tests); also returns the full matrix when Vfull != NULL. */
void orc_evaluate_fullcov(const orc_model *m, int nq, const double *qx, const double *qy, const double *qz,
                          double *v, double *Vfull)
{
    int n = m->n;
    double *Dqp = (double *)malloc(sizeof(double) * (size_t)nq * n);
    double *Kpq = (double *)malloc(sizeof(double) * (size_t)nq * n);
    double *Kqq = (double *)malloc(sizeof(double) * (size_t)nq * nq);
    orc_dist_matrix(m->dist_mode, nq, n, qx, qy, qz, m->px, m->py, m->pz, Dqp);
    for (size_t i = 0; i < (size_t)nq * n; ++i)
        Dqp[i] = orc_k(&m->kern, Dqp[i]); /* Kqp, nq x n col-major */
    for (int i = 0; i < nq; ++i)
        for (int j = 0; j < n; ++j)
            Kpq[(size_t)i * n + j] = Dqp[(size_t)j * nq + i]; /* :308 */
    orc_dist_matrix(m->dist_mode, nq, nq, qx, qy, qz, qx, qy, qz, Kqq);
    for (size_t i = 0; i < (size_t)nq * nq; ++i)
        Kqq[i] = orc_k(&m->kern, Kqq[i]);
    orc_ldlt_solve(n, m->ldlt, m->transp, nq, Kpq); /* :316 */
    for (int c = 0; c < nq; ++c)
        for (int r = 0; r < nq; ++r) {
            double acc = 0;
            for (int j = 0; j < n; ++j)
                acc += Dqp[(size_t)j * nq + r] * Kpq[(size_t)c * n + j];
            double val = Kqq[(size_t)c * nq + r] - acc; /* :318 */
            if (Vfull)
                Vfull[(size_t)c * nq + r] = val;
            if (r == c && v)
                v[r] = val; /* :319 */
        }
    free(Dqp);
    free(Kpq);
    free(Kqq);
}

/* ---- accessors ---------------------------------------------------------------------------- */
/* Eigen predicates used by the atlas (Eigen 3.2 DenseBase): v.isMuchSmallerThan(other, prec) for a vector
 * against a scalar is |v|^2 <= prec^2 other^2; v.isZero(prec) is |v_i| <= prec for every coefficient. */
static int vec_much_smaller(const double v[3], double other, double prec)
{
    return v[0] * v[0] + v[1] * v[1] + v[2] * v[2] <= prec * prec * other * other;
}
static int vec_is_zero(const double v[3], double prec)
{
    return fabs(v[0]) <= prec && fabs(v[1]) <= prec && fabs(v[2]) <= prec;
}

/* AtlasBase::project, reference include/atlas/atlas.hpp:201-276, one start point at a time: gradient descent
 * onto f = 0 along the (un-normalised) gradient.  Per iteration the reference evaluates the mean at the
 * current point (:225), tests |f| < f_tol (:236), steps by step_mul f g unless the step is "wrong" (:246-252),
 * evaluates mean + variance + gradient at the new point (:260; the variance only feeds a log line and is not
 * computed here), adopts the new gradient unless it is "wrong" (:261-266) and tests |f_new - f| < improve_tol
 * (:267).  status: 1 f_tol, 2 improve_tol, 3 max_iter, -1 f is NaN/Inf (the reference throws, :227-231).
 * out_f is the mean at the returned point. */
void orc_project(const orc_model *m, int n, const double *x, const double *y, const double *z, const double *normal,
                 double f_tol, double improve_tol, int max_iter, double step_mul, double *out_xyz, double *out_f,
                 int *out_iter, int *out_status)
{
    for (int i = 0; i < n; ++i) {
        double cur[3] = {x[i], y[i], z[i]};
        double g[3] = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
        int iter = 0, status = 3;
        double f_here = 0.0;
        while (iter < max_iter) {
            double fc;
            orc_evaluate(m, 1, &cur[0], &cur[1], &cur[2], &fc, NULL, NULL, NULL, NULL); /* :225 */
            f_here = fc;
            if (isnan(fc) || isinf(fc)) { /* :227 */
                status = -1;
                break;
            }
            if (fabs(fc) < f_tol) { /* :236 */
                status = 1;
                break;
            }
            double step[3] = {step_mul * fc * g[0], step_mul * fc * g[1], step_mul * fc * g[2]}; /* :245 */
            if (!(!vec_much_smaller(step, 1e3, 1e-1) || vec_is_zero(step, 1e-6))) /* :246-251 */
                for (int c = 0; c < 3; ++c)
                    cur[c] -= step[c];
            double fo, N[3];
            orc_evaluate(m, 1, &cur[0], &cur[1], &cur[2], &fo, NULL, N, NULL, NULL); /* :260 */
            f_here = fo;
            if (!(!vec_much_smaller(N, 1e3, 1e-1) || vec_is_zero(N, 1e-5))) /* :261-266 */
                for (int c = 0; c < 3; ++c)
                    g[c] = N[c];
            if (fabs(fo - fc) < improve_tol) { /* :267 */
                status = 2;
                break;
            }
            ++iter;
        }
        if (max_iter <= 0) /* loop never entered: report the mean at the start point */
            orc_evaluate(m, 1, &cur[0], &cur[1], &cur[2], &f_here, NULL, NULL, NULL, NULL);
        for (int c = 0; c < 3; ++c)
            out_xyz[3 * i + c] = cur[c];
        if (out_f)
            out_f[i] = f_here;
        if (out_iter)
            out_iter[i] = iter;
        if (out_status)
            out_status[i] = status;
    }
}

/* GaussianProcessNode::marchingSampling + marchingCubes, reference src/gp_node.cpp:1102-1291: the surface-following
 * sampler (the node's second sampling mode).  (1) Start point: the first lattice point of x, y, z in [-1.1, 1.1], step
 * 0.1 (accumulated doubles, x outermost, :1126-1152) with |f| <= f_tol, stored as float (pcl::PointXYZ).  (2) A cube of
 * side `leaf` centred there is sampled on (steps + 1)^3 points, steps = round(leaf / pass), coordinates in FLOAT
 * arithmetic as in :1210-1212 (start.x - leaf / 2 + i * pass with start.x, leaf, pass float); points with |f| <= f_tol
 * are kept with their variance (:1219-1225); a kept point on a face of the cube (i == 0, i == steps, ...) makes the
 * neighbour cube across that face a candidate (:1240-1251), centred at start -+ leaf in float (:1271-1281); every cube
 * is sampled once (:1283-1287, the octree occupancy test).
 * Differences from the reference, all about ORDER (its recursion runs one std::thread per neighbour, so its output
 * order and, through float accumulation along different paths, the last bit of a cube centre depend on thread timing):
 * cubes are processed breadth first in discovery order (faces -x +x -y +y -z +z), a cube is identified by its integer
 * offset from the start cube, and its centre is the parent's centre -+ leaf along the discovering step.  The PCL
 * VoxelGrid de-duplication that follows in the node (:1163-1168) is outside the GP path.
 * start_xyz == NULL: scan for the start point; returns the number of kept points (the first `capacity` are written),
 * -1 if no start point is found; *n_cubes = cubes sampled (stops expanding at max_cubes). */
static float marchf_add(float a, float b) { volatile float r = a + b; return r; }
static float marchf_mul(float a, float b) { volatile float r = a * b; return r; }
long orc_march_surface(const orc_model *m, const double *start_xyz, double leaf_d, double pass_d, double f_tol, long max_cubes,
                       long capacity, double *out_xyz, double *out_f, double *out_v, long *n_cubes)
{
    const float leaf = (float)leaf_d, pass = (float)pass_d;
    float sx, sy, sz;
    if (start_xyz) {
        sx = (float)start_xyz[0], sy = (float)start_xyz[1], sz = (float)start_xyz[2];
    } else {
        int found = 0;
        for (double x = -1.1; x <= 1.1 && !found; x += 0.1)
            for (double y = -1.1; y <= 1.1 && !found; y += 0.1)
                for (double z = -1.1; z <= 1.1; z += 0.1) {
                    double ff;
                    orc_evaluate(m, 1, &x, &y, &z, &ff, NULL, NULL, NULL, NULL); /* :1136 */
                    if (fabs(ff) <= f_tol) { /* :1137 */
                        sx = (float)x, sy = (float)y, sz = (float)z;
                        found = 1;
                        break;
                    }
                }
        if (!found) {
            if (n_cubes)
                *n_cubes = 0;
            return -1;
        }
    }
    const long steps = lroundf(leaf / pass); /* :1201 */
    /* breadth-first queue of cubes: integer offsets + float centres */
    long cap_q = 1024, nq = 0, head = 0;
    int *off = (int *)malloc(sizeof(int) * 3 * (size_t)cap_q);
    float *cen = (float *)malloc(sizeof(float) * 3 * (size_t)cap_q);
    /* visited set: open addressing on packed offsets */
    size_t hcap = 4096, hcnt = 0;
    long long *hkeys = (long long *)malloc(sizeof(long long) * hcap);
    for (size_t i = 0; i < hcap; ++i)
        hkeys[i] = -1;
#define MARCH_KEY(a, b, c) ((((long long)(a) + 1048576) << 42) | (((long long)(b) + 1048576) << 21) | ((long long)(c) + 1048576))
#define MARCH_HASH(k) ((size_t)(((unsigned long long)(k) * 0x9E3779B97F4A7C15ull) >> 20))
    long long k0 = MARCH_KEY(0, 0, 0);
    hkeys[MARCH_HASH(k0) % hcap] = k0;
    hcnt = 1;
    off[0] = off[1] = off[2] = 0;
    cen[0] = sx, cen[1] = sy, cen[2] = sz;
    nq = 1;
    long kept = 0;
    while (head < nq && head < max_cubes) {
        const int ox = off[3 * head], oy = off[3 * head + 1], oz = off[3 * head + 2];
        const float cx = cen[3 * head], cy = cen[3 * head + 1], cz = cen[3 * head + 2];
        ++head;
        int where[6] = {0, 0, 0, 0, 0, 0};
        const float hx = marchf_add(cx, -(leaf / 2)), hy = marchf_add(cy, -(leaf / 2)), hz = marchf_add(cz, -(leaf / 2));
        for (long i = 0; i <= steps; ++i)
            for (long j = 0; j <= steps; ++j)
                for (long k = 0; k <= steps; ++k) {
                    double x = (double)marchf_add(hx, marchf_mul((float)i, pass)); /* :1210 */
                    double y = (double)marchf_add(hy, marchf_mul((float)j, pass));
                    double z = (double)marchf_add(hz, marchf_mul((float)k, pass));
                    double ff, vv;
                    orc_evaluate(m, 1, &x, &y, &z, &ff, NULL, NULL, NULL, NULL); /* :1217 (the variance only matters for kept points) */
                    if (fabs(ff) <= f_tol) { /* :1218 */
                        orc_evaluate(m, 1, &x, &y, &z, &ff, &vv, NULL, NULL, NULL);
                        if (kept < capacity) {
                            out_xyz[3 * kept] = x, out_xyz[3 * kept + 1] = y, out_xyz[3 * kept + 2] = z;
                            out_f[kept] = ff;
                            out_v[kept] = vv;
                        }
                        ++kept;
                        if (i == 0) where[0] = 1; /* :1240-1251 */
                        if (i == steps) where[1] = 1;
                        if (j == 0) where[2] = 1;
                        if (j == steps) where[3] = 1;
                        if (k == 0) where[4] = 1;
                        if (k == steps) where[5] = 1;
                    }
                }
        for (int w = 0; w < 6; ++w) { /* :1266-1288 */
            if (!where[w])
                continue;
            int nx = ox, ny = oy, nz = oz;
            float px = cx, py = cy, pz = cz;
            if (w == 0) { nx -= 1; px = marchf_add(px, -leaf); }
            if (w == 1) { nx += 1; px = marchf_add(px, leaf); }
            if (w == 2) { ny -= 1; py = marchf_add(py, -leaf); }
            if (w == 3) { ny += 1; py = marchf_add(py, leaf); }
            if (w == 4) { nz -= 1; pz = marchf_add(pz, -leaf); }
            if (w == 5) { nz += 1; pz = marchf_add(pz, leaf); }
            long long key = MARCH_KEY(nx, ny, nz);
            size_t h = MARCH_HASH(key) % hcap;
            int seen = 0;
            while (hkeys[h] != -1) {
                if (hkeys[h] == key) {
                    seen = 1;
                    break;
                }
                h = (h + 1) % hcap;
            }
            if (seen)
                continue;
            hkeys[h] = key;
            if (++hcnt * 2 > hcap) { /* grow */
                size_t ncap = hcap * 4;
                long long *nk = (long long *)malloc(sizeof(long long) * ncap);
                for (size_t t = 0; t < ncap; ++t)
                    nk[t] = -1;
                for (size_t t = 0; t < hcap; ++t)
                    if (hkeys[t] != -1) {
                        size_t hh = MARCH_HASH(hkeys[t]) % ncap;
                        while (nk[hh] != -1)
                            hh = (hh + 1) % ncap;
                        nk[hh] = hkeys[t];
                    }
                free(hkeys);
                hkeys = nk;
                hcap = ncap;
            }
            if (nq == cap_q) {
                cap_q *= 2;
                off = (int *)realloc(off, sizeof(int) * 3 * (size_t)cap_q);
                cen = (float *)realloc(cen, sizeof(float) * 3 * (size_t)cap_q);
            }
            off[3 * nq] = nx, off[3 * nq + 1] = ny, off[3 * nq + 2] = nz;
            cen[3 * nq] = px, cen[3 * nq + 1] = py, cen[3 * nq + 2] = pz;
            ++nq;
        }
    }
#undef MARCH_KEY
#undef MARCH_HASH
    if (n_cubes)
        *n_cubes = head;
    free(off);
    free(cen);
    free(hkeys);
    return kept;
}

int orc_n(const orc_model *m) { return m->n; }
double orc_R(const orc_model *m) { return m->R; }
int orc_ldlt_info(const orc_model *m) { return m->ldlt_info; }
void orc_get_alpha(const orc_model *m, double *out) { memcpy(out, m->alpha, sizeof(double) * (size_t)m->n); }
void orc_get_Kpp(const orc_model *m, double *out)
{
    memcpy(out, m->Kpp, sizeof(double) * (size_t)m->n * m->n);
}
void orc_get_ldlt(const orc_model *m, double *out, int *transp)
{
    memcpy(out, m->ldlt, sizeof(double) * (size_t)m->n * m->n);
    memcpy(transp, m->transp, sizeof(int) * (size_t)m->n);
}
int orc_get_normals(const orc_model *m, double *out)
{
    if (!m->normals)
        return -1;
    memcpy(out, m->normals, sizeof(double) * (size_t)m->n * 3);
    return 0;
}
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int t)
{
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* =============================================================================================================
 * Second library of the reference, gp::GaussianProcess (include/gp/GaussianProcess.h): a GP trained on values AND
 * gradients.  PARITY UNPINNED, and more than that: this library of the reference does not build and its derivative
 * blocks are inconsistent (CovSE.h:84-89 lacks the delta/l^2 term of the second derivative and multiplies the noise
 * into the kernel; ThinPlate has no second derivative; update_k_star :459-497 indexes past the sample set; var()
 * :256-269 conditions on the value block only; evaluate() :227-235 returns zeros).  This is a restatement of the
 * algorithm those functions are written towards -- same matrix layout (compute() :545-567: rows [0, n) values, row
 * n + 3 i + d derivative d of point i), same target layout (src/gp/SampleSet.cpp:27-35), llt() (:578), alpha by two
 * triangular solves (:517-519), f() = k_star alpha (:246-250), var() = k(x,x) - |L^-1 k_star_0|^2 (:263-268),
 * logLikelihood (:382-384) -- with the exact derivative blocks of a radial kernel k(r), u = x - x', g = k'/r, h = g'/r:
 *   cov(d_d f(x), f(x')) = g u_d,   cov(f(x), d_e f(x')) = -g u_e,   cov(d_d f(x), d_e f(x')) = -g delta_de - h u_d u_e.
 * The blocks are checked against central differences of k in tests/test_oracle.py.
 * kid 5: CovSE (CovSE.h:70-74) k = p0^2 exp(-r^2 / (2 p1^2));  kid 2: ThinPlate (CovThinPlate.h:80-83), R = p0.
 * ============================================================================================================= */
typedef struct {
    int kid, n, info;
    double p0, p1, sn2;
    double *x, *y, *z, *L /* 4n x 4n col-major, lower = Cholesky factor */, *alpha, *yv;
    double loglik;
} orc_dgp;

static void dgp_cov(const orc_dgp *g, double r2, double *k, double *gg, double *h)
{
    if (g->kid == ORC_THINPLATE) {
        const double r = sqrt(r2), R = g->p0;
        *k = 2 * r * r * r - 3 * R * r * r + R * R * R;
        *gg = 6 * r - 6 * R;
        *h = r > 0 ? 6 / r : 0.0;
    } else {
        const double l2 = g->p1 * g->p1;
        *k = g->p0 * g->p0 * exp(-0.5 * r2 / l2);
        *gg = -*k / l2;
        *h = *k / (l2 * l2);
    }
}

/* covariance of observation a of the training set with observation b (a, b in [0, 4n)), without noise */
static double dgp_entry(const orc_dgp *g, int a, int b)
{
    const int n = g->n;
    const int ia = a < n ? a : (a - n) / 3, da = a < n ? -1 : (a - n) % 3;
    const int ib = b < n ? b : (b - n) / 3, db = b < n ? -1 : (b - n) % 3;
    const double u[3] = {g->x[ia] - g->x[ib], g->y[ia] - g->y[ib], g->z[ia] - g->z[ib]};
    double k, gg, h;
    dgp_cov(g, u[0] * u[0] + u[1] * u[1] + u[2] * u[2], &k, &gg, &h);
    if (da < 0 && db < 0)
        return k;
    if (da >= 0 && db < 0)
        return gg * u[da];
    if (da < 0)
        return -gg * u[db];
    return -(da == db ? gg : 0.0) - h * u[da] * u[db];
}

void orc_dgp_free(orc_dgp *g)
{
    if (!g)
        return;
    free(g->x), free(g->y), free(g->z), free(g->L), free(g->alpha), free(g->yv);
    free(g);
}

orc_dgp *orc_dgp_create(int kid, double p0, double p1, double noise, int n, const double *x, const double *y,
                        const double *z, const double *target, const double *normals)
{
    orc_dgp *g = (orc_dgp *)calloc(1, sizeof(orc_dgp));
    const int m = 4 * n;
    g->kid = kid, g->n = n, g->p0 = p0, g->p1 = p1, g->sn2 = noise * noise;
    g->x = (double *)malloc(sizeof(double) * n), g->y = (double *)malloc(sizeof(double) * n);
    g->z = (double *)malloc(sizeof(double) * n);
    memcpy(g->x, x, sizeof(double) * n), memcpy(g->y, y, sizeof(double) * n), memcpy(g->z, z, sizeof(double) * n);
    g->L = (double *)calloc((size_t)m * m, sizeof(double));
    g->alpha = (double *)calloc(m, sizeof(double));
    g->yv = (double *)calloc(m, sizeof(double));
    for (int i = 0; i < n; ++i) { /* SampleSet.cpp:27-35 */
        g->yv[i] = target[i];
        if (normals)
            for (int d = 0; d < 3; ++d)
                g->yv[n + 3 * i + d] = normals[3 * i + d];
    }
    /* compute() :545-567: lower triangle of the 4n x 4n matrix, noise on the diagonal */
    for (int a = 0; a < m; ++a)
        for (int b = 0; b <= a; ++b)
            g->L[(size_t)b * m + a] = dgp_entry(g, a, b) + (a == b ? g->sn2 : 0.0);
    /* :578 llt(): unblocked Cholesky, column by column */
    for (int j = 0; j < m && !g->info; ++j) {
        double d = g->L[(size_t)j * m + j];
        for (int k = 0; k < j; ++k)
            d -= g->L[(size_t)k * m + j] * g->L[(size_t)k * m + j];
        if (!(d > 0.0)) {
            g->info = j + 1;
            break;
        }
        d = sqrt(d);
        g->L[(size_t)j * m + j] = d;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int i = j + 1; i < m; ++i) {
            double s = g->L[(size_t)j * m + i];
            for (int k = 0; k < j; ++k)
                s -= g->L[(size_t)k * m + i] * g->L[(size_t)k * m + j];
            g->L[(size_t)j * m + i] = s / d;
        }
    }
    if (g->info)
        return g;
    /* update_alpha() :517-519: alpha = L^-T L^-1 y */
    memcpy(g->alpha, g->yv, sizeof(double) * m);
    for (int i = 0; i < m; ++i) {
        double s = g->alpha[i];
        for (int k = 0; k < i; ++k)
            s -= g->L[(size_t)k * m + i] * g->alpha[k];
        g->alpha[i] = s / g->L[(size_t)i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = g->alpha[i];
        for (int k = i + 1; k < m; ++k)
            s -= g->L[(size_t)i * m + k] * g->alpha[k];
        g->alpha[i] = s / g->L[(size_t)i * m + i];
    }
    /* logLikelihood() :382-384 over all 4n observations */
    double quad = 0, logdet = 0;
    for (int i = 0; i < m; ++i) {
        quad += g->yv[i] * g->alpha[i];
        logdet += 2 * log(g->L[(size_t)i * m + i]);
    }
    g->loglik = -0.5 * quad - 0.5 * logdet - 0.5 * m * log(2 * M_PI);
    return g;
}

int orc_dgp_info(const orc_dgp *g) { return g->info; }
double orc_dgp_loglik(const orc_dgp *g) { return g->loglik; }
void orc_dgp_get_alpha(const orc_dgp *g, double *out) { memcpy(out, g->alpha, sizeof(double) * 4 * (size_t)g->n); }
/* the covariance matrix itself (symmetric, row-major = col-major), for the finite-difference test of its blocks */
void orc_dgp_get_K(const orc_dgp *g, double *out)
{
    const int m = 4 * g->n;
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b)
            out[(size_t)a * m + b] = dgp_entry(g, a, b) + (a == b ? g->sn2 : 0.0);
}

/* f() :237-252 and var() :256-269 for nq queries: f4[4 q] the mean, f4[4 q + 1..3] its gradient */
void orc_dgp_evaluate(const orc_dgp *g, int nq, const double *qx, const double *qy, const double *qz, double *f4,
                      double *var)
{
    const int n = g->n, m = 4 * n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int q = 0; q < nq; ++q) {
        double *ks = (double *)malloc(sizeof(double) * 4 * (size_t)m); /* k_star, 4 x 4n (:465-489) */
        for (int j = 0; j < n; ++j) {
            const double u[3] = {qx[q] - g->x[j], qy[q] - g->y[j], qz[q] - g->z[j]};
            double k, gg, h;
            dgp_cov(g, u[0] * u[0] + u[1] * u[1] + u[2] * u[2], &k, &gg, &h);
            ks[j] = k;
            for (int e = 0; e < 3; ++e)
                ks[n + 3 * j + e] = -gg * u[e];
            for (int d = 0; d < 3; ++d) {
                ks[(size_t)(1 + d) * m + j] = gg * u[d];
                for (int e = 0; e < 3; ++e)
                    ks[(size_t)(1 + d) * m + n + 3 * j + e] = -(d == e ? gg : 0.0) - h * u[d] * u[e];
            }
        }
        for (int r = 0; r < 4; ++r) {
            double s = 0;
            for (int b = 0; b < m; ++b)
                s += ks[(size_t)r * m + b] * g->alpha[b];
            f4[4 * q + r] = s;
        }
        if (var) { /* v = L^-1 k_star_0 ; var = k(x,x) - v.v */
            double k0, gg, h, vv = 0;
            dgp_cov(g, 0.0, &k0, &gg, &h);
            for (int i = 0; i < m; ++i) {
                double s = ks[i];
                for (int k = 0; k < i; ++k)
                    s -= g->L[(size_t)k * m + i] * ks[k];
                ks[i] = s / g->L[(size_t)i * m + i];
                vv += ks[i] * ks[i];
            }
            var[q] = k0 - vv;
        }
        free(ks);
    }
}
