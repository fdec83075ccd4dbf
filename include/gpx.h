/*
 * gpx.h -- C ABI of libgpx.so: MI355X (gfx950) GP-regression hot path.
 *
 * Drop-in boundary for the reference's gp_regression::GPRegressor<Cov> (header-only C++ over
 * Eigen; /root/reference/include/gp_regression/gp_regressor.hpp).  Every entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 *   - All API scalars and arrays are IEEE double, as in the reference (std::vector<double>,
 *     Eigen::MatrixXd); gpx_options.precision selects the arithmetic used on the device.
 *   - Host entry points copy caller-owned host arrays in/out.  The *_device entry points take
 *     device pointers (HIP) and a stream and never synchronise the host.
 *   - Functions return GPX_OK (0) or a negative gpx_status; gpx_last_error() gives the
 *     thread-local message.  Nothing throws across this boundary.
 *   - There is NO CPU compute fallback: without a HIP device every compute call fails with
 *     GPX_E_HIP / GPX_E_NO_DEVICE.
 */
#ifndef GPX_H
#define GPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_VERSION_MAJOR 0
#define GPX_VERSION_MINOR 1

typedef enum {
    GPX_OK = 0,
    GPX_E_NULL = -1,          /* "Empty data pointer" / "Empty Model pointer"  gp_regressor.hpp:198,:225,:285,:335,:374,:565-566 */
    GPX_E_EMPTY = -2,         /* "All input data is empty!"                    gp_regressor.hpp:567-571 */
    GPX_E_LABELED_QUERY = -3, /* "Query is already labeled!"                   gp_regressor.hpp:230-231,:290-291,:340-341 */
    GPX_E_SIZE_MISMATCH = -4, /* new: the reference does not validate lengths */
    GPX_E_SINGULAR = -5,      /* zero / non-finite pivot in the LDL^T factorisation */
    GPX_E_NAN_INPUT = -6,
    GPX_E_HIP = -7,           /* HIP runtime error (message holds hipGetErrorString) */
    GPX_E_OOM = -8,
    GPX_E_NO_DEVICE = -9,
    GPX_E_BAD_ARG = -10,
    GPX_E_STATE = -11         /* model not ready (e.g. shell not committed) */
} gpx_status;

/* Covariance functions on the UN-squared distance d (reference semantics kept verbatim):
 *   GAUSSIAN  sigma^2 * exp(-d/l^2)              kernels/gaussian.hpp:15-27   p = {sigma, length}
 *   LAPLACE   2*sigma * exp(-d/l)                kernels/laplace.hpp:37-49    p = {sigma, length}
 *   THINPLATE 2d^3 - 3R d^2 + R^3                kernels/thin_plate.hpp:12-20 p = {R}
 *   MATERN32  sigma^2 (1+s) e^-s,  s=sqrt3 d/l   matlab_src/test_gp_regression_3Dsurf.m:117-119
 *   MATERN52  sigma^2 (1+s+s^2/3) e^-s, s=sqrt5 d/l                       ... :121-123
 * "computediff" follows each reference kernel (k'(d) for Gaussian/Laplace, k'(d)/d for
 * ThinPlate); the Matern kernels use k'(d)/d. */
typedef enum {
    GPX_KERNEL_GAUSSIAN = 0,
    GPX_KERNEL_LAPLACE = 1,
    GPX_KERNEL_THINPLATE = 2,
    GPX_KERNEL_MATERN32 = 3,
    GPX_KERNEL_MATERN52 = 4,
    GPX_KERNEL_SE = 5 /* gpx_dgp_* only: sf^2 exp(-|x - x'|^2 / (2 l^2)), p = {sf, l}   reference include/gp/CovSE.h:70-74 */
} gpx_kernel_id;

typedef struct gpx_kernel {
    int32_t id;  /* gpx_kernel_id */
    int32_t reserved;
    double p[4];
} gpx_kernel;

/* Arithmetic of the factorisation and of the variance contraction (the API is double throughout and the
 * mean / gradient are always evaluated in fp64):
 *   F32    kernel matrix, LDL^T and variance GEMM in fp32 (fp32 MFMA); alpha is refined with fp64 matrix-free
 *          residuals; the inverse factor is assembled in fp64 from the fp32 factor and rounded once; the variance
 *          GEMM contracts a centred kernel operand -- k minus a per-query parabola in the squared distance, formed in
 *          fp64 and rounded once -- and its epilogue adds the fit back, squares, weights and sums in fp64: every kernel
 *          within 1e-5 of the fp64 result at N = 16384 in max|dv| / max|v_ref| (thin plate R = 4: 4e-6, where k(0) is
 *          60 x max|v|).  Models of up to 2048 padded rows (GPX_TRAIN_F64_MAX) are trained in fp64 like MIXED: (nearly)
 *          free at that size.  THIN-PLATE models are trained in fp64 at every size whose fp64 temporaries fit the device
 *          (cond > 1e6, predictor weights of 10-100: an fp32 LDL^T shows in the variance); such models hold no factor
 *          afterwards, so gpx_model_update rebuilds them instead of appending.  A model whose kernel matrix turns out
 *          INDEFINITE (negative pivots: thin plate with R below the diameter of the cloud, as the node's R = 2) keeps
 *          its fp64 state and predicts like an F64 model, whatever precision was asked: its quadratic form has terms of
 *          both signs that cancel beyond what fp32 carries.
 *   F64    everything in fp64 (fp64 MFMA): the reference's arithmetic
 *   MIXED  train (kernel matrix, LDL^T, alpha, inverse factor) in fp64, then the inverse factor is
 *          rounded once to fp32 and the variance GEMM runs in fp32; the fp64 factor is released
 *   F32_SPLIT  as F32, but the variance contraction runs on the fp16 matrix cores with every fp32 operand
 *          carried as hi + lo fp16 halves (3 MFMA products, fp32 accumulation); the hi halves of each MFMA
 *          k-group share one quantum, which makes the matrix core's fixed-point product sums exact: measured
 *          as accurate as F32 (better for N >= 4096) at ~0.4x its time; opt-in.  Models of up to 1024 points form and split
 *          the operand inside the kernel (no operand travels through memory): 1.3-2x the F32 small-model kernel; their
 *          thin-plate form keeps the fp32 contraction */
typedef enum { GPX_PREC_F32 = 0, GPX_PREC_F64 = 1, GPX_PREC_MIXED = 2, GPX_PREC_F32_SPLIT = 3 } gpx_precision;

typedef struct gpx_options {
    int32_t precision;     /* gpx_precision */
    int32_t device;        /* HIP device ordinal; -1 = current device */
    int32_t with_normals;  /* create<true>: normals at the training points (gp_regressor.hpp:166-181) */
    int32_t ir_steps;      /* refinement steps for alpha (fp64 residuals); -1 = adaptive: at least 1, then until
                              max|y - K alpha| <= 1e-9 max|y|, at most 4 */
    int32_t prepare_variance; /* 1: build the inverse factor inside create (else lazily at first variance query) */
    int32_t query_batch;   /* queries per variance batch (multiple of 128); 0 = default */
    int32_t reserved[6];
} gpx_options;

typedef struct gpx_model gpx_model; /* opaque; replaces struct Model, gp_regressor.hpp:71-87 */

/* Per-stage device timings (HIP events) of the last create / evaluate on this model, ms.  Models of up to 1024 padded rows are
 * created by three launches (kernel matrix + LDL^T + inverse factor | alpha + refinement | rounding of the fp32 state): their
 * times are reported as t_factor_ms, t_solve_ms and t_inverse_ms, t_kbuild_ms is 0.  Larger fresh creates up to 16384 rows form the
 * kernel matrix inside the factorisation launch: t_kbuild_ms ~ 0, factor_gemm_launches = 0. */
typedef struct gpx_stats {
    double t_kbuild_ms, t_factor_ms, t_solve_ms, t_inverse_ms, t_normals_ms; /* create */
    double t_mean_ms, t_var_ms;                                              /* last evaluate */
    double t_var_gemm_ms;   /* the variance GEMM launches only (subset of t_var_ms) */
    double t_factor_gemm_ms; /* trailing-update GEMM launches only (subset of t_factor_ms) */
    int64_t n, n_padded, n_negative_pivots, ir_steps_done;
    double alpha_residual;  /* max |y - K alpha| after refinement (fp64, matrix-free) */
    int64_t var_gemm_launches, factor_gemm_launches;
    int64_t solve_fallbacks; /* > 0: a one-launch dataflow kernel (block substitution, small-model create, dataflow
                                factorisation) gave up waiting and the work was redone by its launch-per-step form (same
                                result; see gpx_factor.hip, gpx_dataflow.hpp) */
    double t_var_kqp_ms;      /* the kernel-operand (Kqp) launches of the last evaluate only (subset of t_var_ms) */
    double factor_gemm_flops; /* algorithmic flops of the event-timed trailing-update launches (lower tiles x 2 x 128^2 x K) */
    double surface_candidates; /* last gpx_model_sample_surface / _march_surface batch: queries whose mean was evaluated in fp64
                                  (= all of them unless the fp32 screen of large grids ran; gpx_predict.hip) */
} gpx_stats;

typedef enum {
    GPX_FIELD_N = 0,        /* int64  */
    GPX_FIELD_R = 1,        /* double: Model::R, max pairwise training distance (gp_regressor.hpp:135) */
    GPX_FIELD_ALPHA = 2,    /* double[n]   Model::alpha (:163), caller's point order */
    GPX_FIELD_P = 3,        /* double[3n]  Model::P, row-major n x 3 */
    GPX_FIELD_Y = 4,        /* double[n]   Model::Y */
    GPX_FIELD_S2 = 5,       /* double[n]   Model::S2 (zeros when create had no sigma2) */
    GPX_FIELD_NORMALS = 6,  /* double[3n]  Model::N, row-major; only if with_normals */
    GPX_FIELD_STATS = 7,    /* gpx_stats */
    GPX_FIELD_D = 8,        /* double[n]   diagonal D of P K P^T = L D L^T (Eigen LDLT::vectorD) */
    GPX_FIELD_PERM = 9,     /* int32[n]    internal position -> caller index */
    GPX_FIELD_KPP = 10      /* double[n*n] Model::Kpp row-major (symmetric), caller order; rebuilt on demand */
} gpx_field;

/* ---- library ---------------------------------------------------------------------------- */
const char *gpx_last_error(void); /* thread-local */
const char *gpx_version(void);
int gpx_device_count(void);       /* number of HIP devices, 0 if none / no driver */

/* ---- model: GPRegressor<Cov>::create<withNormals>(data, gp), gp_regressor.hpp:110-182 ------
 * kernel == CovType held by GPRegressor::kernel_ (:97, setCovFunction :488-491).
 * x,y,z,label: n doubles each (Data::coord_x/y/z/label, :51-54).  sigma2: n doubles or NULL
 * (== empty Data::sigma2: no diagonal noise, :154).  *out replaces any previous model, as
 * create resets gp (:116-117). */
int gpx_model_create(const gpx_kernel *kernel, size_t n, const double *x, const double *y, const double *z,
                     const double *label, const double *sigma2, const gpx_options *opt, gpx_model **out);

/* GPRegressor<Cov>::update<withNormals>(new_data, gp), gp_regressor.hpp:367-479: append n_new points.  The
 * reference refactors from scratch (:457-459); here the existing factor is extended (new kernel rows, their
 * update against the old column blocks, factorisation of the new trailing block, new solve) whenever the old
 * points stay first in Eigen's pivot order on the diagonal k(0) + sigma2 -- e.g. for one common sigma2 -- and the
 * old factor is still held in the training precision; otherwise -- and whenever the grown model still has at most 1024 padded
 * rows, where the three-launch create is faster than any append -- the model is rebuilt.  Either way the results
 * equal a fresh create on the concatenated data to rounding.  Model::R is not refreshed (:454-455). */
int gpx_model_update(gpx_model *m, size_t n_new, const double *x, const double *y, const double *z,
                     const double *label, const double *sigma2);

/* GPRegressor<Cov>::evaluate(gp, query, f [,v [,N [,Tx,Ty]]]), gp_regressor.hpp:332-357,
 * :282-324, :222-273, :194-212.  f: nq (required).  v, grad, tx, ty: NULL or nq / 3nq / 3nq / 3nq
 * doubles (row-major nq x 3).  grad is the UN-normalised gradient (:247-250), zero-initialised.
 * tx/ty follow computeTangentBasis (:29-44).  v_i = k(0) - k_i^T K^-1 k_i (diagonal of :316-319).
 * Re-entrant on a const model (src/gp_node.cpp:1027-1038 calls it from hundreds of threads). */
int gpx_model_evaluate(const gpx_model *m, size_t nq, const double *qx, const double *qy, const double *qz,
                       double *f, double *v, double *grad, double *tx, double *ty);

/* Same, with every array already resident on the model's device (double), enqueued on `stream`
 * (a hipStream_t, or NULL for the model's own stream); returns without synchronising. */
int gpx_model_evaluate_device(const gpx_model *m, size_t nq, const void *d_qx, const void *d_qy,
                              const void *d_qz, void *d_f, void *d_v, void *d_grad, void *d_tx, void *d_ty,
                              void *stream);

/* Batched form of AtlasBase::project (reference include/atlas/atlas.hpp:201-276; called per chart by
 * atlas_variance.hpp:124 and atlas_collision.hpp:55): gradient descent of nq start points onto the surface f = 0.
 * Per iteration and point, exactly as the reference: stop with status 1 if |f| < f_tol (:236); step by
 * step_mul * f * g unless the step is longer than 100 or all its components are within 1e-6 (:245-252); evaluate
 * mean and gradient at the new point (:260); adopt the new gradient unless it is longer than 100 or within 1e-5 of
 * zero (:261-266); stop with status 2 if |f_new - f| < improve_tol (:267); status 3 after max_iter iterations
 * (:272); status -1 where the reference throws "f is nan or inf" (:227-231).  The reference also computes the
 * variance in every iteration, for a log line only; it is not computed here.  The whole loop runs on the device
 * (one mean+gradient pass over the still unconverged points per iteration).
 * normal: 3*nq row-major, the un-normalised start directions (the chart gradients, atlas_variance.hpp:122).
 * out_xyz: 3*nq row-major.  out_f (mean at the returned point), out_iter, out_status: nq each, or NULL.
 * opt == NULL: the reference's defaults f_tol 1e-2, improve_tol 1e-7, max_iter 500, step_mul 0.001. */
typedef struct gpx_project_options {
    double f_tol, improve_tol, step_mul;
    int32_t max_iter;
    int32_t reserved[3];
} gpx_project_options;
int gpx_model_project(const gpx_model *m, size_t nq, const double *x, const double *y, const double *z,
                      const double *normal, const gpx_project_options *opt, double *out_xyz, double *out_f,
                      int32_t *out_iter, int32_t *out_status);

/* Iso-surface sampling: the batched form of the node's fakeDeterministicSampling / samplePoint
 * (src/gp_node.cpp:998-1100): evaluate the mean on all nq queries, keep those with |f| <= f_tol
 * (the node's 0.01, :1075) and compute the variance ONLY for the survivors -- the step right after
 * the hot path, fused so that the dominant variance cost scales with the surface, not the volume.
 * idx (query positions, ascending), f, v: host arrays of `capacity` entries; *n_out = number of
 * survivors.  If more than `capacity` survive, the first `capacity` are returned together with
 * GPX_E_SIZE_MISMATCH (and *n_out holds the full count).  v may be NULL (selection only). */
int gpx_model_sample_surface(const gpx_model *m, size_t nq, const double *qx, const double *qy, const double *qz,
                             double f_tol, size_t capacity, int64_t *idx, double *f, double *v, size_t *n_out);

/* The node's surface-following sampler, GaussianProcessNode::marchingSampling + marchingCubes
 * (src/gp_node.cpp:1102-1291), as a breadth-first frontier of device batches: starting from the cube of side `leaf`
 * around a point on the surface, every cube is sampled on (steps + 1)^3 lattice points, steps = round(leaf / pass)
 * (:1201-1212, coordinates in float as there); points with |f| <= f_tol (the node's 0.01, :1218) are kept with
 * their variance; a kept point on a face of the cube makes the neighbour across that face the next to sample
 * (:1240-1288); each cube is sampled once.  One mean batch per frontier, the variance only for the kept points.
 * start_xyz: 3 doubles, or NULL to look for the start as the node does -- the first point of the 0.1 lattice on
 * [-1.1, 1.1]^3 with |f| <= f_tol (:1126-1152; GPX_E_EMPTY "No starting point found" if there is none).
 * xyz (3 * capacity, row-major), f, v (capacity each; v may be NULL): the kept points in discovery order (cubes
 * breadth first, faces -x +x -y +y -z +z, lattice points i, j, k); *n_out = their number (GPX_E_SIZE_MISMATCH and the
 * first `capacity` points if more); *n_cubes = cubes sampled (expansion stops at max_cubes).  The voxel-grid
 * de-duplication that follows in the node (:1163-1168) is the caller's. */
int gpx_model_march_surface(const gpx_model *m, const double *start_xyz, double leaf, double pass, double f_tol,
                            size_t max_cubes, size_t capacity, double *xyz, double *f, double *v, size_t *n_out,
                            size_t *n_cubes);

/* Ensure the inverse factor needed by variance queries exists (else built at first use). */
int gpx_model_prepare_variance(gpx_model *m);

int gpx_model_get(const gpx_model *m, int field, void *dst, size_t bytes);
int gpx_model_sync(const gpx_model *m); /* hipStreamSynchronize on the model's stream */
void gpx_model_destroy(gpx_model *m);
/* Device buffers of 64 MiB and more (kernel matrix, inverse factor, assembly temporaries) are parked in a per-process
 * pool when a model releases them and handed to the next model, because allocating and freeing multi-GiB buffers costs
 * hundreds of milliseconds; GPX_POOL_MB (default 16384, 0 = off) caps the parked bytes.  gpx_trim() frees them. */
void gpx_trim(void);

/* The developer / test switches (GPX_* environment variables) are parsed ONCE per process, at the first
 * call that needs one; no call path reads the environment.  gpx_debug_reload() parses them again -- the test suites change a
 * switch between two calls of one process.  Not to be called while other threads are inside the library.  (No reference
 * counterpart: gp_regressor.hpp has no switches.) */
void gpx_debug_reload(void);

/* ---- one model, query grid sharded over ranks (one process per GPU) -----------------------
 * The rank that factorised exports its read-only state as two contiguous device blobs (part 0: points, alpha, 1/D, the
 * 14 row vectors of the variance fit, the centre of the cloud; part 1: the inverse factor); the host moves them with
 * RCCL (torch.distributed broadcast over xGMI) into the blobs of a shell created with the same kernel / n / options on
 * the other ranks, then commits it.  The byte layout depends on (n, options) only -- except that a model whose kernel
 * matrix is indefinite keeps its fp64 state (see gpx_precision): gpx_model_state_blob then fails with GPX_E_STATE and
 * source and shells must be created with GPX_PREC_F64.  The zero-communication alternative is every rank calling
 * gpx_model_create itself (bench.py --mode shard --state recompute).  No reference equivalent (single-process). */
int gpx_model_create_shell(const gpx_kernel *kernel, size_t n, const gpx_options *opt, gpx_model **out);
int gpx_model_state_blob(gpx_model *m, int part /*0: points, alpha, 1/D; 1: inverse factor*/, void **d_ptr,
                         size_t *bytes);
int gpx_model_commit(gpx_model *m, int with_variance);

/* In-process multi-device placement for a C++ caller of the reference's shape (one process, host threads sharing a
 * model: src/gp_node.cpp:1025-1038; no reference equivalent, the reference is single-device CPU code): out[i]
 * receives a read-only replica of `src` on HIP device devs[i] (state copied device to device, over xGMI where peer
 * access exists; all replicas' copies are issued together, one pair per destination on its own stream, and waited for
 * once), ready for gpx_model_evaluate* / _sample_surface / _project; evaluate on a replica returns bit-identical results
 * to `src`.  The inverse factor is built on `src` first if it was not yet.  Replicas are destroyed with
 * gpx_model_destroy; gpx_model_update on a replica rebuilds it from its host copy of the data.  On failure no replica is
 * left behind and the caller's current device is unchanged.
 * STATUS: the code specific to a SECOND device (peer enable, copies between ordinals, per-device kernel attributes) has
 * only run on hardware where the test box offers more than one GPU (tests/test_gpu_parity.py::
 * test_replicas_on_a_second_device skips otherwise); the one-GPU boxes of this project exercise devs = {0, 0} only.
 * There is no RCCL inside the library: within one process replicas travel by peer copies (this call); across
 * processes (one per GPU) the host moves the two state blobs with its own collective -- bench.py uses
 * torch.distributed broadcast (backend "nccl" = RCCL over xGMI). */
int gpx_model_replicate(const gpx_model *src, int ndev, const int *devs, gpx_model **out);

/* ONE evaluate / sampleSurface call over several replicas (the "query-grid shards" of the one-process C / C++ caller:
 * src/gp_node.cpp:1025-1038 is one process; no reference equivalent).  replicas: n_replicas DISTINCT handles that predict alike
 * -- a model and its gpx_model_replicate copies, on any devices (the same ordinal twice is allowed and is what the one-GPU test
 * boxes run).  Replica i works on contiguous slab i from its own host thread and writes its results in place into the caller's
 * arrays; arguments and statuses otherwise as gpx_model_evaluate / gpx_model_sample_surface (survivor positions refer to the
 * whole query array, ascending).  The results equal the single call's BIT FOR BIT (tests/test_gpu_sharded_call.py), which
 * decides where the cuts are: the order of the sums of an evaluate depends on the device batch a query sits in, so
 * gpx_model_evaluate_sharded distributes the slices of 2^18 queries the single call itself is pipelined in -- slab i = slices
 * gpx_slab_range(ceil(nq / 2^18), i, n_replicas), each evaluated exactly as the single call evaluates it; a call of one slice
 * stays whole on replicas[0].  gpx_model_sample_surface evaluates in an order that does not depend on the batch, so its
 * sharded form cuts the queries themselves: slab i = gpx_slab_range(nq, i, n_replicas); empty slabs are skipped.  The first
 * failing slab's status is returned, its message prefixed with the slab.  STATUS: a second ORDINAL has not run (see
 * gpx_model_replicate). */
void gpx_slab_range(size_t nq, int rank, int world, size_t *lo, size_t *hi);
int gpx_model_evaluate_sharded(const gpx_model *const *replicas, int n_replicas, size_t nq, const double *qx,
                               const double *qy, const double *qz, double *f, double *v, double *grad, double *tx, double *ty);
int gpx_model_sample_surface_sharded(const gpx_model *const *replicas, int n_replicas, size_t nq, const double *qx,
                                     const double *qy, const double *qz, double f_tol, size_t capacity, int64_t *idx, double *f,
                                     double *v, size_t *n_out);

/* ---- GP with derivative observations: first slice of the reference's second library, gp::GaussianProcess -----------
 * (include/gp/GaussianProcess.h)  Training data: a value AND a gradient (surface normal) at every point;
 * the model is the 4n x 4n covariance of values and derivatives -- compute() :532-583, layout [values (n) | d/dx d/dy
 * d/dz of point 0 | ... | of point n-1] as :553-567 and SampleSet (src/gp/SampleSet.cpp:23-35) --, its factorisation
 * (:578) and alpha = K^-1 y (update_alpha, :505-528).  kernel: GPX_KERNEL_SE {sf, l} (CovSE.h:70-74) or
 * GPX_KERNEL_THINPLATE {R} (CovThinPlate.h:80-83); noise: sn, sn^2 is added to the whole diagonal.  fp64 throughout.
 * NOT a drop-in: that library of the reference does not build and its derivative blocks are inconsistent (CovSE.h:84-89
 * lacks the delta/l^2 term, ThinPlate has no second derivative at all); this is the algorithm they are written towards,
 * with exact derivative blocks (csrc/gpx_dgp.hip), on the same device machinery as gpx_model (tiled matrix build,
 * blocked LDL^T on the matrix cores -- the matrix must be positive definite, as for the reference's llt() --, the fused
 * variance contraction).
 * normals: 3n row-major or NULL (zeros).  gpx_dgp_evaluate = f() :237-252 for nq queries: f4[4 q + 0] the mean value,
 * f4[4 q + 1..3] its gradient; var (nq, or NULL) = var() :256-269, conditioned on values AND derivatives (the reference
 * conditions on the value block only). */
typedef struct gpx_dgp gpx_dgp;
typedef enum {
    GPX_DGP_FIELD_N = 0,      /* int64: training points */
    GPX_DGP_FIELD_ALPHA = 1,  /* double[4n], layout as above */
    GPX_DGP_FIELD_LOGLIK = 2, /* double: logLikelihood() :376-385 over all 4n observations */
    GPX_DGP_FIELD_STATS = 3,  /* gpx_stats: t_kbuild_ms, t_factor_ms, t_solve_ms, t_mean_ms, t_var_ms, n = 4n, n_padded */
    GPX_DGP_FIELD_APPENDED_FROM = 4 /* int64: rows of the old factor the last gpx_dgp_add carried over (0: built from scratch) */
} gpx_dgp_field;
int gpx_dgp_create(const gpx_kernel *kernel, double noise, size_t n, const double *x, const double *y, const double *z,
                   const double *target, const double *normals, const gpx_options *opt, gpx_dgp **out);
int gpx_dgp_evaluate(const gpx_dgp *g, size_t nq, const double *qx, const double *qy, const double *qz, double *f4,
                     double *var);
/* add_patterns (GaussianProcess.h:340-374): append n_new samples (normals NULL = zeros).  Their rows -- the value and the
 * three derivatives of every new sample; the reference appends value rows only, ignoring the derivative blocks of its own
 * compute() -- are appended to the existing factor as :356-368 does row by row, here block-wise: the leading
 * 128 * floor(4 n_old / 128) rows of the old factor are kept, the rest is built and eliminated against them.  The result
 * equals gpx_dgp_create on the concatenated data to rounding (bit for bit with GPX_DGP_APPEND=0, which rebuilds; also taken
 * when the old model has fewer than 128 rows).  External exclusion against evaluate, as for gpx_model_update. */
int gpx_dgp_add(gpx_dgp *g, size_t n_new, const double *x, const double *y, const double *z, const double *target,
                const double *normals);
int gpx_dgp_get(const gpx_dgp *g, int field, void *dst, size_t bytes);
void gpx_dgp_destroy(gpx_dgp *g);
/* logLikelihoodGradient (GaussianProcess.h:387-410): grad2 = d logLikelihood / d (log l, log sf), CovSE's log
 * hyper-parameters in getLogHyper()'s order (CovSE.h:96-118), = 1/2 sum_ab (alpha alpha^T - K^-1)_ab dK_ab (:398-408) over
 * the whole 4n x 4n covariance -- the exact gradient of GPX_DGP_FIELD_LOGLIK.  GPX_KERNEL_SE models only (the other
 * covariance classes of that library have an empty grad(), Covs.h:172): GPX_E_BAD_ARG otherwise. */
int gpx_dgp_loglik_gradient(const gpx_dgp *g, double *grad2);
/* Optimisation (RProp), GaussianProcess.h:41-160.  gpx_rprop = Optimisation::Desc (:49-62), gpx_rprop_default = its
 * setToDefault (:64-73).  gpx_dgp_optimise = Optimisation::find (:86-122) on a GPX_KERNEL_SE model: at most max_iter
 * steps in (log l, log sf), each one a gradient, a step, a refit (setLogHyper + compute()) and a likelihood; the model
 * ends on the best parameters met (:120).  res (may be NULL): those parameters, their likelihood (the model's own if no
 * step was applied) and the number of steps applied.  A step onto parameters whose covariance does not factorise ends
 * the search there: the call still returns GPX_OK with the model on the best parameters met.  A descriptor with a
 * non-finite or non-positive step size, delta_min > delta_max, eta_minus > 1 or eta_plus < 1 is GPX_E_BAD_ARG.
 * External exclusion against evaluate, as for gpx_dgp_add. */
typedef struct {
    double delta0, delta_min, delta_max, eta_minus, eta_plus, eps_stop;
    uint64_t max_iter;
} gpx_rprop;
typedef struct {
    double loghyper[2]; /* log l, log sf */
    double loglik;
    uint64_t iterations;
} gpx_rprop_result;
void gpx_rprop_default(gpx_rprop *d);
int gpx_dgp_optimise(gpx_dgp *g, const gpx_rprop *desc, gpx_rprop_result *res);

/* ---- stand-alone device stages (tests, bench roofline legs) -------------------------------
 * kbuild: K[i][j] = k(|p_i-p_j|) + sigma2_i*delta_ij on the lower block-triangle of an
 * n_padded x n_padded row-major matrix of `precision` scalars (identity on the padding);
 * replaces buildEuclideanDistanceMatrix + the kernel loop, gp_regressor.hpp:132-159, :548-557.
 * d_x,d_y,d_z,d_s2: device arrays of `precision` scalars (GPX_PREC_F32 or _F64), n_padded long.
 * d_rmax: reserved, pass NULL (Model::R comes from gpx_model_get). */
int gpx_dev_kbuild(const gpx_kernel *kernel, int precision, size_t n, size_t n_padded, const void *d_x,
                   const void *d_y, const void *d_z, const void *d_s2, void *d_K, void *d_rmax, void *stream);
/* kqp: Kqp[q][j] = k(|q - p_j|) - (a_q + b_q s + c_q s^2), s = |q - p_j|^2, for nq queries (a multiple of 128) against
 * the n training points, row-major nq x n_padded `precision` scalars -- the kernel operand of one variance batch
 * (gp_regressor.hpp:300-303).  d_px,d_py,d_pz (n_padded) and d_qx,d_qy,d_qz (nq): doubles; distances, kernel and fit are
 * formed in fp64 and rounded once to `precision`.  d_fab: the per-query fit as 3 x nq doubles (rows a_q, b_q, c_q) or NULL
 * for the plain kernel values. */
int gpx_dev_kqp(const gpx_kernel *kernel, int precision, size_t n, size_t n_padded, const void *d_px, const void *d_py,
                const void *d_pz, size_t nq, const void *d_qx, const void *d_qy, const void *d_qz, const void *d_fab,
                void *d_Kqp, void *stream);
/* The same operand formed in fp32 arithmetic (what the exponential kernels use): d_px,d_py,d_pz are n_padded FLOATS, the
 * training points relative to the centre d_cen (3 doubles on the device); queries are centred before rounding; Kqp:
 * floats. */
int gpx_dev_kqp_f32(const gpx_kernel *kernel, size_t n, size_t n_padded, const void *d_px, const void *d_py,
                    const void *d_pz, const void *d_cen, size_t nq, const void *d_qx, const void *d_qy, const void *d_qz,
                    const void *d_fab, void *d_Kqp, void *stream);
size_t gpx_padded_n(size_t n); /* leading dimension / padded order used for n training points */

/* ---- PCD input + node-equivalent data preparation (host) ----------------------------------
 * gpx_pcd_read: pcl::io::loadPCDFile (src/gp_node.cpp:557) for ascii / binary /
 * binary_compressed files; xyz as float.  Returns the number of points, or a negative status.
 * If xyz == NULL only the count is returned.
 * gpx_node_training_set: deMeanAndNormalizeData + prepareExtData + prepareData + computeGP's
 * concatenation (src/gp_node.cpp:85-117, :793-914): out arrays must hold n_points+15 doubles. */
long gpx_pcd_read(const char *path, float *xyz, size_t capacity_points);
int gpx_node_training_set(const float *xyz, size_t n_points, double sigma2, double out_sphere_rad, double *x,
                          double *y, double *z, double *label, double *s2); /* returns the number of exterior
                          points appended (15) or a negative gpx_status */

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
