// gp_regression::Data / Model / GPRegressor<Cov> -- source-compatible host mirror of the reference's
// include/gp_regression/gp_regressor.hpp (:29-44, :49-87, :92-574) over the C ABI of libgpx.so.
//
// A caller such as the reference's src/gp_node.cpp (:916-922 create, :1069-1074 evaluate) or
// include/atlas/atlas_variance.hpp (:72-78) compiles against this header unchanged:
//
//     reg_ = std::make_shared<gp_regression::ThinPlateRegressor>();
//     reg_->setCovFunction(std::make_shared<gp_regression::ThinPlate>(2.0));
//     reg_->create<false>(data_gp, obj_gp);
//     reg_->evaluate(obj_gp, qq, ff, vv);
//
// What differs (documented in DESIGN.md): Model is a handle to device state with read accessors
// instead of public Eigen matrices; gradients/normals start from zero; distances are direct
// differences (no NaN); every computation runs on the GPU -- there is no host fallback.
// The Eigen-typed overloads exist only when <Eigen/Core> is available; the std::vector twins
// (row-major nq x 3) are always there.
#ifndef GPX_SHIM_GP_REGRESSOR_HPP
#define GPX_SHIM_GP_REGRESSOR_HPP

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define GPX_SHIM_HAVE_EIGEN 1
#endif
#endif

#include <gp_regression/cov_functions.h>
#include <gp_regression/gp_regression_exception.h>
#include <gpx.h>

namespace gp_regression
{

// ---- computeTangentBasis (reference :29-44), Eigen-free and inline -----------------------------
inline void computeTangentBasis(const double grad[3], double N[3], double Tx[3], double Ty[3])
{
    const double nrm = std::sqrt(grad[0] * grad[0] + grad[1] * grad[1] + grad[2] * grad[2]);
    for (int c = 0; c < 3; ++c)
        N[c] = nrm > 0 ? grad[c] / nrm : grad[c];
    const double nn = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
    const double d2 = (N[0] - 1) * (N[0] - 1) + N[1] * N[1] + N[2] * N[2];
    const bool along_x = d2 <= 1e-6 * (nn < 1.0 ? nn : 1.0);  // Eigen isApprox(UnitX, 1e-3)
    double e[3] = {along_x ? 0.0 : 1.0, along_x ? 1.0 : 0.0, 0.0};
    const double dot = N[0] * e[0] + N[1] * e[1] + N[2] * e[2];
    double tn = 0;
    for (int c = 0; c < 3; ++c) {
        Tx[c] = e[c] - N[c] * dot;
        tn += Tx[c] * Tx[c];
    }
    tn = std::sqrt(tn);
    if (tn > 0)
        for (int c = 0; c < 3; ++c)
            Tx[c] /= tn;
    Ty[0] = N[1] * Tx[2] - N[2] * Tx[1];
    Ty[1] = N[2] * Tx[0] - N[0] * Tx[2];
    Ty[2] = N[0] * Tx[1] - N[1] * Tx[0];
    tn = std::sqrt(Ty[0] * Ty[0] + Ty[1] * Ty[1] + Ty[2] * Ty[2]);
    if (tn > 0)
        for (int c = 0; c < 3; ++c)
            Ty[c] /= tn;
}
#ifdef GPX_SHIM_HAVE_EIGEN
inline void computeTangentBasis(const Eigen::Vector3d &grad, Eigen::Vector3d &N, Eigen::Vector3d &Tx,
                                Eigen::Vector3d &Ty)
{
    computeTangentBasis(grad.data(), N.data(), Tx.data(), Ty.data());
}
#endif

// ---- Data (reference :49-66) ---------------------------------------------------------------------
struct Data {
    std::vector<double> coord_x;
    std::vector<double> coord_y;
    std::vector<double> coord_z;
    std::vector<double> label;
    std::vector<double> sigma2;
    typedef std::shared_ptr<Data> Ptr;
    typedef std::shared_ptr<const Data> ConstPtr;
    void clear()
    {
        coord_x.clear();
        coord_y.clear();
        coord_z.clear();
        label.clear();
        sigma2.clear();
    }
};

// ---- Model (reference :71-87): device-resident; read accessors replace the public matrices --------
struct Model {
    double R = 0.0;  // larger pairwise distance in training (:73, :135)
    typedef std::shared_ptr<Model> Ptr;
    typedef std::shared_ptr<const Model> ConstPtr;

    Model() = default;
    Model(const Model &) = delete;
    Model &operator=(const Model &) = delete;
    ~Model()
    {
        drop_shards();
        if (handle_)
            gpx_model_destroy(handle_);
    }
    gpx_model *handle() const { return handle_; }
    // number of replicas (this model included) a large evaluate / sampleSurface call is cut over; 1 = not sharded
    size_t shards() const
    {
        std::lock_guard<std::mutex> lk(shard_mtx_);
        return shards_.empty() ? 1 : shards_.size();
    }
    size_t size() const
    {
        int64_t n = 0;
        if (handle_)
            gpx_model_get(handle_, GPX_FIELD_N, &n, sizeof(n));
        return (size_t)n;
    }
    std::vector<double> alpha() const { return vec(GPX_FIELD_ALPHA, 1); }    // Model::alpha
    std::vector<double> P() const { return vec(GPX_FIELD_P, 3); }            // Model::P, row-major n x 3
    std::vector<double> Y() const { return vec(GPX_FIELD_Y, 1); }            // Model::Y
    std::vector<double> S2() const { return vec(GPX_FIELD_S2, 1); }          // Model::S2
    std::vector<double> N() const { return vec(GPX_FIELD_NORMALS, 3); }      // Model::N (create<true>)
    std::vector<double> vectorD() const { return vec(GPX_FIELD_D, 1); }      // cholesker.vectorD()
    gpx_stats stats() const
    {
        gpx_stats s;
        std::memset(&s, 0, sizeof(s));
        if (handle_)
            gpx_model_get(handle_, GPX_FIELD_STATS, &s, sizeof(s));
        return s;
    }

private:
    template <typename>
    friend class GPRegressor;
    std::vector<double> vec(int field, size_t width) const
    {
        std::vector<double> out(size() * width);
        if (!handle_ || out.empty() || gpx_model_get(handle_, field, out.data(), out.size() * sizeof(double)) != GPX_OK)
            out.clear();
        return out;
    }
    gpx_model *handle_ = nullptr;
    // replicas of this model for GPRegressor::devices_ (made at the first large call, dropped by update); [0] == handle_
    mutable std::mutex shard_mtx_;
    mutable std::vector<gpx_model *> shards_;
    mutable bool shards_failed_ = false;
    void drop_shards() const
    {
        std::lock_guard<std::mutex> lk(shard_mtx_);
        for (size_t i = 1; i < shards_.size(); ++i)
            gpx_model_destroy(shards_[i]);
        shards_.clear();
        shards_failed_ = false;
    }
};

// ---- kernel object -> C ABI descriptor -----------------------------------------------------------
inline gpx_kernel gpx_kernel_of(const Gaussian &k) { return gpx_kernel{GPX_KERNEL_GAUSSIAN, 0, {k.sigma_, k.length_, 0, 0}}; }
inline gpx_kernel gpx_kernel_of(const Laplace &k) { return gpx_kernel{GPX_KERNEL_LAPLACE, 0, {k.sigma_, k.length_, 0, 0}}; }
inline gpx_kernel gpx_kernel_of(const ThinPlate &k) { return gpx_kernel{GPX_KERNEL_THINPLATE, 0, {k.R(), 0, 0, 0}}; }
inline gpx_kernel gpx_kernel_of(const Matern32 &k) { return gpx_kernel{GPX_KERNEL_MATERN32, 0, {k.sigma_, k.length_, 0, 0}}; }
inline gpx_kernel gpx_kernel_of(const Matern52 &k) { return gpx_kernel{GPX_KERNEL_MATERN52, 0, {k.sigma_, k.length_, 0, 0}}; }

// ---- GPRegressor (reference :92-574) ---------------------------------------------------------------
template <typename CovType>
class GPRegressor
{
public:
    std::shared_ptr<CovType> kernel_;  // :97
    // Device / precision knobs (new).  Default fp64 = the reference's arithmetic; GPX_PREC_F32 trades
    // 1e-5 norm-wise accuracy for speed.  Environment overrides: GPX_PRECISION=f32|f64, GPX_DEVICE=<n>.
    gpx_options options_;
    // Several GPUs behind the UNCHANGED caller (new; north star: query-grid shards).  devices_ = HIP ordinals; with more than one
    // entry create() places the model on devices_[0] and every evaluate / sampleSurface call of at least shard_min_nq_ queries
    // is cut into contiguous slabs over replicas on all entries (gpx_model_evaluate_sharded: whole slices of 2^18 queries,
    // so an evaluate of up to 2^18 queries stays on the model; gpx_model_sample_surface_sharded: any grid; replicas are made at
    // the first such call and after every update).  Results equal the unsharded call's bit for bit.  Environment: GPX_DEVICES=0,1,2,3 and GPX_SHARD_MIN_NQ.  The same ordinal twice is
    // allowed (two replicas on one GPU: what a one-GPU box can run).  Smaller calls -- the node's one-point evaluate from 841
    // threads -- stay on the model itself.
    std::vector<int> devices_;
    size_t shard_min_nq_ = (size_t)1 << 16;

    GPRegressor() : kernel_(std::make_shared<CovType>())  // :497-500
    {
        std::memset(&options_, 0, sizeof(options_));
        options_.precision = GPX_PREC_F64;
        options_.device = -1;
        options_.ir_steps = -1;
        if (const char *p = std::getenv("GPX_PRECISION"))
            options_.precision = (std::strcmp(p, "f32") == 0) ? GPX_PREC_F32 : GPX_PREC_F64;
        if (const char *d = std::getenv("GPX_DEVICE"))
            options_.device = std::atoi(d);
        if (const char *l = std::getenv("GPX_DEVICES")) {
            for (const char *c = l; *c;) {
                char *end = nullptr;
                const long o = std::strtol(c, &end, 10);
                if (end == c)
                    break;
                devices_.push_back((int)o);
                c = (*end == ',') ? end + 1 : end;
            }
        }
        if (const char *q = std::getenv("GPX_SHARD_MIN_NQ"))
            shard_min_nq_ = (size_t)std::strtoull(q, nullptr, 10);
    }
    virtual ~GPRegressor() {}

    void setCovFunction(const std::shared_ptr<CovType> &kernel) { kernel_ = kernel; }  // :488-491

    // create<withNormals>(data, gp), :110-182
    template <bool withNormals>
    void create(Data::ConstPtr data, Model::Ptr &gp)
    {
        assertData(data);
        gp = std::make_shared<Model>();  // "reset output", :116-117
        if (!kernel_)
            throw GPRegressionException("Empty kernel pointer");
        const size_t n = data->coord_x.size();
        if (data->coord_y.size() != n || data->coord_z.size() != n || data->label.size() != n ||
            (!data->sigma2.empty() && data->sigma2.size() != n))
            throw GPRegressionException("Input data vectors have different lengths");
        gpx_options o = options_;
        o.with_normals = withNormals ? 1 : 0;
        if (devices_.size() > 1)
            o.device = devices_[0];
        const gpx_kernel k = gpx_kernel_of(*kernel_);
        gpx_model *h = nullptr;
        const int rc = gpx_model_create(&k, n, data->coord_x.data(), data->coord_y.data(), data->coord_z.data(),
                                        data->label.data(), data->sigma2.empty() ? nullptr : data->sigma2.data(), &o,
                                        &h);
        if (rc != GPX_OK)
            throw GPRegressionException(message(rc));
        gp->handle_ = h;
        gpx_model_get(h, GPX_FIELD_R, &gp->R, sizeof(double));
    }

    // evaluate(gp, query, f), :332-357
    void evaluate(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f)
    {
        run(gp, query, f, nullptr, nullptr, nullptr, nullptr);
    }
    // evaluate(gp, query, f, v), :282-324
    void evaluate(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f, std::vector<double> &v)
    {
        run(gp, query, f, &v, nullptr, nullptr, nullptr);
    }
    // Eigen-free twins of :222-273 and :194-212: grad / tx / ty are row-major nq x 3.
    void evaluate(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f, std::vector<double> &v,
                  std::vector<double> &grad)
    {
        run(gp, query, f, &v, &grad, nullptr, nullptr);
    }
    void evaluate(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f, std::vector<double> &v,
                  std::vector<double> &grad, std::vector<double> &tx, std::vector<double> &ty)
    {
        run(gp, query, f, &v, &grad, &tx, &ty);
    }
#ifdef GPX_SHIM_HAVE_EIGEN
    // evaluate(gp, query, f, v, N), :222-273 -- N is the UN-normalised gradient
    void evaluate(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f, std::vector<double> &v,
                  Eigen::MatrixXd &N)
    {
        std::vector<double> g;
        run(gp, query, f, &v, &g, nullptr, nullptr);
        to_eigen(g, N);
    }
    // evaluate(gp, query, f, v, N, Tx, Ty), :194-212
    void evaluate(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f, std::vector<double> &v,
                  Eigen::MatrixXd &N, Eigen::MatrixXd &Tx, Eigen::MatrixXd &Ty)
    {
        std::vector<double> g, tx, ty;
        run(gp, query, f, &v, &g, &tx, &ty);
        to_eigen(g, N);
        to_eigen(tx, Tx);
        to_eigen(ty, Ty);
    }
#endif

    // Batched form of the node's fakeDeterministicSampling / samplePoint (src/gp_node.cpp:998-1100): keep the
    // queries with |f| <= f_tol (the node uses 0.01, :1075) and return their position in `query`, mean and
    // variance; the variance is computed for the survivors only.  New entry, not in the reference
    // header.
    void sampleSurface(Model::ConstPtr gp, Data::ConstPtr query, double f_tol, std::vector<size_t> &idx,
                       std::vector<double> &f, std::vector<double> &v)
    {
        if (!gp || !gp->handle_)
            throw GPRegressionException("Empty Model pointer");
        assertData(query);
        if (!query->label.empty())
            throw GPRegressionException("Query is already labeled!");
        const size_t nq = query->coord_x.size();
        if (query->coord_y.size() != nq || query->coord_z.size() != nq)
            throw GPRegressionException("Input data vectors have different lengths");
        std::vector<int64_t> pos(nq);
        f.assign(nq, 0.0);
        v.assign(nq, 0.0);
        size_t n = 0;
        std::vector<gpx_model *> sh = shards_for(*gp, nq);
        const int rc = sh.size() > 1
                           ? gpx_model_sample_surface_sharded(sh.data(), (int)sh.size(), nq, query->coord_x.data(),
                                                              query->coord_y.data(), query->coord_z.data(), f_tol, nq,
                                                              pos.data(), f.data(), v.data(), &n)
                           : gpx_model_sample_surface(gp->handle_, nq, query->coord_x.data(), query->coord_y.data(),
                                                      query->coord_z.data(), f_tol, nq, pos.data(), f.data(), v.data(), &n);
        if (rc != GPX_OK)
            throw GPRegressionException(message(rc));
        idx.assign(pos.begin(), pos.begin() + n);
        f.resize(n);
        v.resize(n);
    }

    // The node's surface-following sampler (src/gp_node.cpp:1102-1291, marchingSampling + marchingCubes) as a
    // frontier of device batches: `out` receives the kept points (coord_x/y/z), f and v their mean and variance.
    // start == nullptr: the start point is searched on the 0.1 lattice as the node does.  Returns the number of cubes
    // sampled.  New entry (gpx_model_march_surface), not in the reference header.
    size_t marchSurface(Model::ConstPtr gp, const double *start_xyz, float leaf_size, float leaf_pass, Data::Ptr out,
                        std::vector<double> &f, std::vector<double> &v, double f_tol = 0.01, size_t max_points = 1u << 18,
                        bool *truncated = nullptr)
    {
        if (!gp || !gp->handle_)
            throw GPRegressionException("Empty Model pointer");
        if (!out)
            throw GPRegressionException("Empty data pointer");
        if (truncated)
            *truncated = false;
        std::vector<double> xyz(3 * max_points);
        f.assign(max_points, 0.0);
        v.assign(max_points, 0.0);
        size_t n = 0, cubes = 0;
        const int rc = gpx_model_march_surface(gp->handle_, start_xyz, leaf_size, leaf_pass, f_tol, (size_t)1 << 24,
                                               max_points, xyz.data(), f.data(), v.data(), &n, &cubes);
        // more surface points than max_points: the first max_points (discovery order) are valid and are returned
        if (rc == GPX_E_SIZE_MISMATCH && n > max_points) {
            if (truncated)
                *truncated = true;
            n = max_points;
        } else if (rc != GPX_OK) {
            throw GPRegressionException(message(rc));
        }
        out->clear();
        out->coord_x.reserve(n), out->coord_y.reserve(n), out->coord_z.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            out->coord_x.push_back(xyz[3 * i]);
            out->coord_y.push_back(xyz[3 * i + 1]);
            out->coord_z.push_back(xyz[3 * i + 2]);
        }
        f.resize(n);
        v.resize(n);
        return cubes;
    }

    // Batched form of AtlasBase::project (include/atlas/atlas.hpp:201-276): every start point of `start` descends
    // onto f = 0 along its gradient, with the reference's tolerances, step rule and stopping criteria; `normals`
    // holds the un-normalised start directions (3 per point, row-major).  `out` receives the projected points
    // (coord_x/y/z), `status` 1 / 2 / 3 for the f_tol / improve_tol / max_iter exit.  Throws like the reference
    // ("f is nan or inf", atlas.hpp:230) if any point hits a NaN.  New entry.
    void project(Model::ConstPtr gp, Data::ConstPtr start, const std::vector<double> &normals, Data::Ptr out,
                 std::vector<int> &status, const double f_tol = 1e-2, const double improve_tol = 1e-7,
                 const unsigned int max_iter = 500, const double step_mul = 0.001)
    {
        if (!gp || !gp->handle_)
            throw GPRegressionException("Empty Model pointer");
        assertData(start);
        if (!out)
            throw GPRegressionException("Empty data pointer");
        const size_t nq = start->coord_x.size();
        if (start->coord_y.size() != nq || start->coord_z.size() != nq || normals.size() != 3 * nq)
            throw GPRegressionException("Input data vectors have different lengths");
        gpx_project_options o{};
        o.f_tol = f_tol, o.improve_tol = improve_tol, o.step_mul = step_mul, o.max_iter = (int32_t)max_iter;
        std::vector<double> xyz(3 * nq);
        std::vector<int32_t> st(nq);
        const int rc = gpx_model_project(gp->handle_, nq, start->coord_x.data(), start->coord_y.data(),
                                         start->coord_z.data(), normals.data(), &o, xyz.data(), nullptr, nullptr,
                                         st.data());
        if (rc != GPX_OK)
            throw GPRegressionException(message(rc));
        out->clear();
        status.assign(st.begin(), st.end());
        for (size_t i = 0; i < nq; ++i) {
            if (st[i] < 0)
                throw GPRegressionException("f is nan or inf");
            out->coord_x.push_back(xyz[3 * i]);
            out->coord_y.push_back(xyz[3 * i + 1]);
            out->coord_z.push_back(xyz[3 * i + 2]);
        }
    }

    // Read-only replicas of a trained model on other GPUs of the node, one per entry of `devices` (HIP ordinals):
    // the caller shards its query grid over them, e.g. one x-slab per replica from its own host threads
    // (src/gp_node.cpp:1025-1038 shares one model among threads the same way).  evaluate() on a replica returns
    // the same values as on `gp`.  New entry (gpx_model_replicate), not in the reference header.
    std::vector<Model::Ptr> replicate(Model::ConstPtr gp, const std::vector<int> &devices)
    {
        if (!gp || !gp->handle_)
            throw GPRegressionException("Empty Model pointer");
        std::vector<gpx_model *> h(devices.size(), nullptr);
        if (devices.empty())
            return {};
        const int rc = gpx_model_replicate(gp->handle_, (int)devices.size(), devices.data(), h.data());
        if (rc != GPX_OK)
            throw GPRegressionException(message(rc));
        std::vector<Model::Ptr> out;
        for (gpx_model *r : h) {
            Model::Ptr m = std::make_shared<Model>();
            m->handle_ = r;
            m->R = gp->R;
            out.push_back(m);
        }
        return out;
    }

    // update<withNormals>(new_data, gp), :367-479
    template <bool withNormals>
    void update(Data::ConstPtr new_data, Model::Ptr gp)
    {
        assertData(new_data);
        if (!gp || !gp->handle_)
            throw GPRegressionException("Empty model pointer");  // :373-374
        const size_t n = new_data->label.size();                  // :383
        if (new_data->coord_x.size() != n || new_data->coord_y.size() != n || new_data->coord_z.size() != n ||
            (!new_data->sigma2.empty() && new_data->sigma2.size() != n))
            throw GPRegressionException("Input data vectors have different lengths");
        gp->drop_shards();  // replicas of the old model: made again at the next large call
        const int rc = gpx_model_update(gp->handle_, n, new_data->coord_x.data(), new_data->coord_y.data(),
                                        new_data->coord_z.data(), new_data->label.data(),
                                        new_data->sigma2.empty() ? nullptr : new_data->sigma2.data());
        if (rc != GPX_OK)
            throw GPRegressionException(message(rc));
    }

private:
    static std::string message(int rc)
    {
        const char *m = gpx_last_error();
        return (m && *m) ? std::string(m) : ("gpx error " + std::to_string(rc));
    }
    // assertData, :563-572
    void assertData(Data::ConstPtr data) const
    {
        if (!data)
            throw GPRegressionException("Empty data pointer");
        if (data->coord_x.empty() && data->coord_y.empty() && data->coord_z.empty() && data->label.empty())
            throw GPRegressionException("All input data is empty!");
    }
    void run(Model::ConstPtr gp, Data::ConstPtr query, std::vector<double> &f, std::vector<double> *v,
             std::vector<double> *grad, std::vector<double> *tx, std::vector<double> *ty)
    {
        if (!gp || !gp->handle_)
            throw GPRegressionException("Empty Model pointer");  // :197-198, :224-225, :284-285, :334-335
        assertData(query);
        if (!query->label.empty())
            throw GPRegressionException("Query is already labeled!");  // :230-231, :290-291, :340-341
        const size_t nq = query->coord_x.size();
        if (query->coord_y.size() != nq || query->coord_z.size() != nq)
            throw GPRegressionException("Input data vectors have different lengths");
        f.assign(nq, 0.0);  // outputs are replaced, not appended (:537-540)
        if (v)
            v->assign(nq, 0.0);
        if (grad)
            grad->assign(3 * nq, 0.0);
        if (tx)
            tx->assign(3 * nq, 0.0);
        if (ty)
            ty->assign(3 * nq, 0.0);
        std::vector<gpx_model *> sh = shards_for(*gp, nq);
        const int rc = sh.size() > 1
                           ? gpx_model_evaluate_sharded(sh.data(), (int)sh.size(), nq, query->coord_x.data(),
                                                        query->coord_y.data(), query->coord_z.data(), f.data(),
                                                        v ? v->data() : nullptr, grad ? grad->data() : nullptr,
                                                        tx ? tx->data() : nullptr, ty ? ty->data() : nullptr)
                           : gpx_model_evaluate(gp->handle_, nq, query->coord_x.data(), query->coord_y.data(),
                                                query->coord_z.data(), f.data(), v ? v->data() : nullptr,
                                                grad ? grad->data() : nullptr, tx ? tx->data() : nullptr,
                                                ty ? ty->data() : nullptr);
        if (rc != GPX_OK)
            throw GPRegressionException(message(rc));
    }
    // the handles a call of nq queries is cut over: the model alone, or the model and its replicas on devices_[1..] (made
    // here, once, under the model's shard lock; a failed replication is not retried and leaves the call on the model alone)
    std::vector<gpx_model *> shards_for(const Model &gp, size_t nq) const
    {
        if (devices_.size() < 2 || nq < shard_min_nq_ || nq < devices_.size())
            return {gp.handle_};
        std::lock_guard<std::mutex> lk(gp.shard_mtx_);
        if (gp.shards_.empty() && !gp.shards_failed_) {
            std::vector<int> rest(devices_.begin() + 1, devices_.end());
            std::vector<gpx_model *> h(rest.size(), nullptr);
            if (gpx_model_replicate(gp.handle_, (int)rest.size(), rest.data(), h.data()) == GPX_OK) {
                gp.shards_.push_back(gp.handle_);
                gp.shards_.insert(gp.shards_.end(), h.begin(), h.end());
            } else {
                gp.shards_failed_ = true;
            }
        }
        return gp.shards_.empty() ? std::vector<gpx_model *>{gp.handle_} : gp.shards_;
    }
#ifdef GPX_SHIM_HAVE_EIGEN
    static void to_eigen(const std::vector<double> &rm, Eigen::MatrixXd &M)
    {
        const size_t n = rm.size() / 3;
        M.resize(n, 3);
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c)
                M(i, c) = rm[3 * i + c];
    }
#endif
};

}  // namespace gp_regression
#endif
