// gpx_eval.hip -- prediction on a ready model (gp_regressor.hpp:194-357): mean / gradient / tangent basis, the
// variance as one fused GEMM per query batch, the host-pointer paths (flat combining of small concurrent calls, the
// one-launch path for a handful of queries, the pipelined large-batch path), iso-surface sampling and the batched
// AtlasBase::project.
#include "gpx_model.hpp"
#include <thread>


namespace gpxh {

// ---- evaluate ------------------------------------------------------------------------------------
int evaluate_locked(gpx_model *m, size_t nq, const double *qx, const double *qy, const double *qz,
                           double *f, double *v, double *grad, double *tx, double *ty, hipStream_t s, bool fixed_order)
{
    const int np = m->npad;
    const long plan_nq = fixed_order ? FIXED_ORDER_PLAN_NQ : 0;
    const size_t e = m->esz;
    const bool want_basis = tx || ty;
    double *g = grad;
    if (want_basis && !g) {
        int rc = ensure(m, (void **)&m->ws_grad, &m->ws_grad_doubles, sizeof(double) * 3 * nq);
        if (rc)
            return rc;
        g = m->ws_grad;
    }
    size_t need = predict_ws_doubles((long)nq, np, g != nullptr, plan_nq) * sizeof(double);
    if (need) {
        int rc = ensure(m, (void **)&m->ws_pred, &m->ws_pred_doubles, need);
        if (rc)
            return rc;
    }
    if (v) {  // the inverse factor first: it decides the representation of X (build_inverse packs the split operands)
        int rc = build_inverse(m);
        if (rc)
            return rc;
    }
    // Small models (the reference's own sizes): every row of the product resident in one wave, the triangle of X skipped per
    // 16-row fragment, the fp64 epilogue inside the triangle, v written directly; where the operand's arithmetic is fp32
    // the wave forms it in registers and neither launch_kqp nor an operand buffer is needed (gpx_varcols.hip) -- the
    // query batch is then bounded only by the fit's coefficient array (136 bytes per query).
    const bool var_cols_on = gpxh::switches().var_cols != 0;
    const bool use_cols = v && var_cols_on && m->var_fit && !m->x_packed && m->prec != GPX_PREC_F64 &&
                          var_cols_fits(m->n, np, np, np + KQP_LDPAD);
    VarColsArgs vc;
    if (use_cols) {
        vc.X = (const float *)m->X, vc.ldx = np, vc.n = m->n, vc.np = np;
        vc.ldk = np + KQP_LDPAD;
        vc.rowcorr = m->d_corr, vc.ldrc = np;
        vc.dinv64 = m->d_dinv64, vc.k0 = m->cov.k0;
        vc.cov = m->cov, vc.op64 = m->op64;
        vc.px = (const float *)m->t_x, vc.py = (const float *)m->t_y, vc.pz = (const float *)m->t_z;
        vc.qx = qx, vc.qy = qy, vc.qz = qz;  // (set per batch below)
        vc.cen[0] = m->cen[0], vc.cen[1] = m->cen[1], vc.cen[2] = m->cen[2];
    }
    const bool cols_gen = use_cols && var_cols_gen(vc);
    // small models of the split-fp16 mode: the same structure on the fp16 matrix cores (gpx_varcols16.hip)
    const bool use_cols16 = cols_gen && m->opt.precision == GPX_PREC_F32_SPLIT && var_cols16_takes(vc);
    // small fp64 models (the header shim's default precision at the reference's own sizes): one kernel, no workspace
    // (gpx_varcols64.hip)
    const bool use_cols64 = v && m->prec == GPX_PREC_F64 && !m->var_fit && !m->x_packed && var_cols64_fits(m->n, np, np);
    const size_t nq_tiles = ((nq + TILE - 1) / TILE) * TILE;
    // (an explicitly set gpx_options.query_batch bounds the batch -- and with it the 136 bytes per query of ws_coef -- on
    // every path; the default of the small-model kernel is the whole call in slices of 2^21 queries)
    const size_t cols_cap = m->opt.query_batch > 0 ? (size_t)m->qbatch : (size_t)1 << 21;
    const size_t qbatch = cols_gen ? std::min<size_t>(nq_tiles, cols_cap) : std::min<size_t>((size_t)m->qbatch, nq_tiles);
    if (use_cols16) {
        int rc = ensure(m, &m->ws_kqp, &m->ws_kqp_bytes, var_cols16_ws_bytes(m->n));  // X split and in fragment order (per call)
        if (rc)
            return rc;
    }
    if (use_cols64) {
        int rc = ensure(m, &m->ws_kqp, &m->ws_kqp_bytes, var_cols64_ws_bytes(m->n));  // X^T of the call
        if (rc)
            return rc;
    }
    if (v && !use_cols64) {
        int rc;
        const size_t qb = qbatch;
        if (!cols_gen && (rc = ensure(m, &m->ws_kqp, &m->ws_kqp_bytes, e * qb * (np + KQP_LDPAD))))
            return rc;
        // with the fit the epilogue of the contraction runs in fp64 and writes fp64 partial sums
        if (!use_cols && (rc = ensure(m, &m->ws_partial, &m->ws_partial_bytes, (m->var_fit ? sizeof(double) : e) * qb * m->nblk)))
            return rc;
        // (the small-model kernel that forms its operand in the wave derives the 14 coefficients from a_q, b_q, c_q itself:
        // 24 instead of 136 bytes per query)
        if (m->var_fit && (rc = ensure(m, &m->ws_coef, &m->ws_coef_bytes, sizeof(double) * qb * (cols_gen ? VAR_NFIT : VAR_NCOEF))))
            return rc;
    }
    // The workspaces (prediction partials, K tile, variance partials) are shared by all evaluations of this model,
    // which may be enqueued on different streams (gpx_model_evaluate_device): order them behind the previous user.
    if (m->ws_in_flight)
        (void)hipStreamWaitEvent(s, m->ev[EV_WS], 0);
    (void)hipEventRecord(m->ev[EV_M0], s);
    // mean and gradient always in fp64 from the fp64 points and alpha (cheap next to the variance, and
    // the long alternating sum of a thin-plate GP at N = 16k is not within 1e-5 in fp32)
    // (small fp64 models without a gradient: the mean rides on the operand values of the variance kernel, gpx_varcols64.hip;
    // with a gradient request the mean kernel runs as everywhere else -- the tests compare the two)
    // (f == nullptr: variance only -- sample_surface already holds the fp64 mean of its survivors)
    const bool mean_fused = use_cols64 && !g && f;
    if (!mean_fused && f)
        launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, (long)nq, qx, qy, qz, f, g, m->ws_pred, s,
                       m->n, plan_nq);
    if (want_basis)
        launch_tangent_basis((long)nq, g, tx, ty, s);
    (void)hipEventRecord(m->ev[EV_M1], s);
    m->gemm_ev_used_var = 0;
    if (use_cols64) {
        hipEvent_t *ev = (s == m->stream) ? gemm_events(m, m->gemm_ev_used_factor) : nullptr;
        if (ev)
            (void)hipEventRecord(ev[0], s);
        launch_var_cols64(m->cov, m->n, np, (const double *)m->X, np, m->d_x, m->d_y, m->d_z, (const double *)m->t_dinv /* fp64 model: T = double */, (long)nq, qx, qy, qz, v,
                          (double *)m->ws_kqp, s, mean_fused ? m->d_alpha : nullptr, f);
        if (ev)
            (void)hipEventRecord(ev[1], s);
        m->gemm_ev_used_var = ev ? 1 : 0;
        m->kqp_ev_used = 0;
    } else if (v) {
        if (use_cols16)
            launch_var_cols16_pack(vc, m->sk, m->ws_kqp, s);
        const size_t qb = qbatch;
        const int np_rows = std::min(np, (m->n + TILE - 1) / TILE * TILE);  // 128-row blocks that hold training points
        const long ldk = (long)np + KQP_LDPAD;  // row stride of the operand buffer (not the power of two X's is)
        // One stream, one operand buffer.  (A two-deep pipeline over the query batches -- operand of batch i + 1 built on a
        // second stream beside the GEMM of batch i -- was measured in rounds 1 and 3 and removed: no gain with the LDS
        // tiles, 2421 against 1950 ms per step with the one-wave tiles, whose rounds of equal-length workgroups the operand
        // kernel's workgroups break up.)
        size_t gi = 0;
        for (size_t q0 = 0; q0 < nq; q0 += qb) {
            const size_t nv = std::min(qb, nq - q0);
            const size_t ntile = ((nv + TILE - 1) / TILE) * TILE;
            void *kqp_buf = m->ws_kqp;
            void *coef_buf = m->ws_coef;
            // The kernel operand holds k - fit with a per-query fit that is rank 14 in (q, p); the GEMM epilogue adds
            // X * fit back, in fp64, from the model's 14 row-correction vectors (gpx_internal.hpp, "low-rank fit").
            hipEvent_t *kev = nullptr;  // brackets the Kqp launch of this batch (stats; only on the model's own stream)
            if (s == m->stream) {
                while (m->kqp_ev.size() < 2 * (gi + 1)) {
                    hipEvent_t e_;
                    if (hipEventCreate(&e_) != hipSuccess)
                        break;
                    m->kqp_ev.push_back(e_);
                }
                if (m->kqp_ev.size() >= 2 * (gi + 1))
                    kev = &m->kqp_ev[2 * gi];
            }
            const double *fab = nullptr;  // rows a_q, b_q, c_q of the batch's coefficient array
            if (m->var_fit) {
                launch_var_fit(m->op64, m->cov, m->n, m->d_x, m->d_y, m->d_z, m->d_meta, (long)nv, (long)ntile, qx + q0,
                               qy + q0, qz + q0, (double *)coef_buf, (long)qb, s, cols_gen);
                fab = (const double *)coef_buf + (cols_gen ? 0 : qb * VAR_NCORR);
            }
            const int part_prec = m->var_fit ? GPX_PREC_F64 : m->prec;  // type of the partial sums
            if (kev)
                (void)hipEventRecord(kev[0], s);
            if (m->x_packed) {  // F32_SPLIT: fp16 hi/lo operands, three MFMA products per k-step
                launch_kqp_split(m->var_fit && m->op64, m->cov, m->sk, m->n, np, m->t_x, m->t_y, m->t_z, m->d_x, m->d_y,
                                 m->d_z, m->d_meta, (long)nv, (long)ntile, qx + q0, qy + q0, qz + q0, kqp_buf, s, fab,
                                 (long)qb, ldk);
                if (kev)
                    (void)hipEventRecord(kev[1], s);
                hipEvent_t *ev2 = (s == m->stream) ? gemm_events(m, m->gemm_ev_used_factor + gi) : nullptr;
                if (ev2)
                    (void)hipEventRecord(ev2[0], s);
                launch_vsplit_gemm(m->X, kqp_buf, np, (int)ntile, (const float *)m->t_dinv, m->ws_partial, (long)qb, s,
                                   np_rows, m->var_fit ? m->d_corr : nullptr, np,
                                   m->var_fit ? (const double *)coef_buf : nullptr, (long)qb, m->d_dinv64, m->d_meta + 3, ldk);
                if (ev2) {
                    (void)hipEventRecord(ev2[1], s);
                    ++gi;
                }
                launch_var_finish(part_prec, m->cov.k0, np_rows / TILE, (long)qb, m->ws_partial, (long)nv, v + q0, s);
                continue;
            }
            if (!cols_gen) {
                // fp64 models and the fp64-formed fp32 operand read the model's fp64 points (differences do not depend
                // on where the cloud sits); the fp32-formed operand reads the centred fp32 points
                const bool c64 = m->prec == GPX_PREC_F64 || (m->var_fit && m->op64);
                // (the kernel uses its n_padded argument only as the stride of the operand's rows)
                launch_kqp(c64, m->prec, m->prec == GPX_PREC_F64, m->cov, m->n, (int)ldk, c64 ? (const void *)m->d_x : m->t_x,
                           c64 ? (const void *)m->d_y : m->t_y, c64 ? (const void *)m->d_z : m->t_z, m->d_meta, (long)nv,
                           (long)ntile, qx + q0, qy + q0, qz + q0, kqp_buf, s, np_rows, fab, (long)qb);
            }
            if (kev)
                (void)hipEventRecord(kev[1], s);
            hipEvent_t *ev = (s == m->stream) ? gemm_events(m, m->gemm_ev_used_factor + gi) : nullptr;
            if (use_cols) {
                vc.Kq = (const float *)kqp_buf;
                vc.colcoef = (const double *)coef_buf, vc.ldcc = (long)qb, vc.compact_coef = cols_gen;
                vc.nq_valid = (long)nv, vc.nq_tile = (long)ntile, vc.v = v + q0;
                vc.qx = qx + q0, vc.qy = qy + q0, vc.qz = qz + q0;
                if (ev)
                    (void)hipEventRecord(ev[0], s);
                if (use_cols16)
                    launch_var_cols16(vc, m->sk, m->ws_kqp, s);
                else
                    launch_var_cols(vc, s);
                if (ev) {
                    (void)hipEventRecord(ev[1], s);
                    ++gi;
                }
                continue;
            }
            GemmArgs a;  // partial[mt][q] = sum_rows (X * Kqp^T)^2 / D
            a.A = m->X, a.lda = np;
            a.B = kqp_buf, a.ldb = ldk;
            a.M = np_rows, a.N = (int)ntile, a.K = np;  // rows of X in the identity padding see only zeros of Kqp
            a.a_lower = 1;
            a.epi = EPI_COLSQ;
            a.m_valid = m->n;
            // Tile of the fp32 product: 6 = one wave per workgroup, a 128 x 128 tile per wave, no LDS, no barrier
            // (gpx_vargemm.hip; 152 TFLOP/s at N = 16384).  GPX_VAR_TILE=3 selects the documented fallback, the LDS-staged
            // 128 x 128 tile with 64-byte k rows at three workgroups per CU (139-140 TFLOP/s) -- also what 6 falls back to
            // for a shape the one-wave kernel does not take.
            a.cfg = gpxh::switches().var_tile == 3 ? 3 : 6;
            a.no_pair = fixed_order ? 1 : 0;
            a.rowweight = m->t_dinv;
            a.partial = m->ws_partial, a.ldp = (long)qb;
            if (m->var_fit)
                a.rowcorr = m->d_corr, a.ldrc = np, a.colcoef = (const double *)coef_buf, a.ldcc = (long)qb,
                a.rowweight64 = m->d_dinv64;
            if (ev)
                (void)hipEventRecord(ev[0], s);
            launch_gemm(m->prec, a, s);
            if (ev) {
                (void)hipEventRecord(ev[1], s);
                ++gi;
            }
            const int bm = gemm_rows_per_partial(m->prec, a);
            launch_var_finish(part_prec, m->cov.k0, np_rows / bm, (long)qb, m->ws_partial, (long)nv, v + q0, s);
        }
        m->gemm_ev_used_var = gi;
        m->kqp_ev_used = (s == m->stream) ? gi : 0;
    }
    (void)hipEventRecord(m->ev[EV_V1], s);
    launch_poison_nonfinite((long)nq, qx, qy, qz, f, v, grad, tx, ty, s);  // (outside the stage events: 24 bytes per query)
    (void)hipEventRecord(m->ev[EV_WS], s);
    m->ws_in_flight = true;
    m->stats_eval_pending = true;
    m->eval_had_var = v != nullptr;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
    return GPX_OK;
}

int check_query(const gpx_model *m, size_t nq, const void *qx, const void *qy, const void *qz, const void *f)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!m->ready)
        return fail(GPX_E_STATE, "model is not ready (shell not committed or create failed)");
    if (nq == 0)
        return fail(GPX_E_EMPTY, "All input data is empty!");
    if (!qx || !qy || !qz)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (!f)
        return fail(GPX_E_NULL, "Empty output pointer");
    return GPX_OK;
}

extern "C" int gpx_model_evaluate_device(const gpx_model *cm, size_t nq, const void *d_qx, const void *d_qy,
                                         const void *d_qz, void *d_f, void *d_v, void *d_grad, void *d_tx,
                                         void *d_ty, void *stream)
{
    int rc = check_query(cm, nq, d_qx, d_qy, d_qz, d_f);
    if (rc)
        return rc;
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = stream ? (hipStream_t)stream : m->stream;
    return evaluate_locked(m, nq, (const double *)d_qx, (const double *)d_qy, (const double *)d_qz, (double *)d_f,
                           (double *)d_v, (double *)d_grad, (double *)d_tx, (double *)d_ty, s);
}

// One device batch for a list of host requests: queries are concatenated into pinned staging, evaluated
// once (the union of the requested outputs), and the results scattered back.

static int run_requests(gpx_model *m, const std::vector<gpx_pending *> &reqs)
{
    size_t total = 0;
    bool wv = false, wg = false, wtx = false, wty = false;
    for (const gpx_pending *r : reqs) {
        total += r->nq;
        wv |= r->v != nullptr;
        wg |= r->grad != nullptr;
        wtx |= r->tx != nullptr;
        wty |= r->ty != nullptr;
    }
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    // layout (host pinned and device alike): qx qy qz | f | v | grad | tx | ty
    const size_t doubles = total * (3 + 1 + 1 + 3 + 3 + 3);
    int rc;
    if ((rc = ensure(m, (void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * doubles)))
        return rc;
    if (m->pin_doubles < doubles) {
        if (m->pin)
            HIPCHK(hipHostFree(m->pin));
        m->pin = nullptr;
        m->pin_doubles = 0;
        HIPCHK(hipHostMalloc((void **)&m->pin, sizeof(double) * doubles, hipHostMallocDefault));
        m->pin_doubles = doubles;
    }
    double *h = m->pin, *d = m->ws_host_io;
    size_t off = 0;
    for (const gpx_pending *r : reqs) {
        std::memcpy(h + off, r->qx, sizeof(double) * r->nq);
        std::memcpy(h + total + off, r->qy, sizeof(double) * r->nq);
        std::memcpy(h + 2 * total + off, r->qz, sizeof(double) * r->nq);
        off += r->nq;
    }
    double *dqx = d, *dqy = d + total, *dqz = d + 2 * total, *df = d + 3 * total, *dv = d + 4 * total,
           *dg = d + 5 * total, *dtx = d + 8 * total, *dty = d + 11 * total;
    hipStream_t s = m->stream;
    // a handful of queries on a small model: one launch that reads and writes the pinned buffer directly
    const bool small_on = gpxh::switches().small_eval != 0;
    if (small_on && total <= SMALL_EVAL_MAX_NQ && m->npad <= SMALL_EVAL_NP_MAX &&
        !(wv && m->opt.precision == GPX_PREC_F32_SPLIT)) {
        if (wv && (rc = build_inverse(m)))
            return rc;
        if (!m->ws_small) {
            const size_t sb = small_eval_scratch_bytes((int)SMALL_EVAL_MAX_NQ, SMALL_EVAL_NP_MAX);
            HIPCHK(hipMalloc(&m->ws_small, sb));
            HIPCHK(hipMemsetAsync(m->ws_small, 0, sb, s));
        }
        launch_small_eval(m->prec, m->cov, m->n, m->npad, m->d_x, m->d_y, m->d_z, m->d_alpha, m->X, m->t_dinv,
                          (int)total, (int)SMALL_EVAL_MAX_NQ, h, h + 3 * total, wv ? h + 4 * total : nullptr,
                          wg ? h + 5 * total : nullptr, wtx ? h + 8 * total : nullptr,
                          wty ? h + 11 * total : nullptr, m->ws_small, s);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess)
            return fail(GPX_E_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
        HIPCHK(hipStreamSynchronize(s));
        off = 0;
        for (const gpx_pending *r : reqs) {
            std::memcpy(r->f, h + 3 * total + off, sizeof(double) * r->nq);
            if (r->v)
                std::memcpy(r->v, h + 4 * total + off, sizeof(double) * r->nq);
            if (r->grad)
                std::memcpy(r->grad, h + 5 * total + 3 * off, sizeof(double) * 3 * r->nq);
            if (r->tx)
                std::memcpy(r->tx, h + 8 * total + 3 * off, sizeof(double) * 3 * r->nq);
            if (r->ty)
                std::memcpy(r->ty, h + 11 * total + 3 * off, sizeof(double) * 3 * r->nq);
            off += r->nq;
        }
        return GPX_OK;
    }
    HIPCHK(hipMemcpyAsync(d, h, sizeof(double) * 3 * total, hipMemcpyHostToDevice, s));
    rc = evaluate_locked(m, total, dqx, dqy, dqz, df, wv ? dv : nullptr, wg ? dg : nullptr, wtx ? dtx : nullptr,
                         wty ? dty : nullptr, s);
    if (rc)
        return rc;
    HIPCHK(hipMemcpyAsync(h + 3 * total, df, sizeof(double) * total * (wv ? 2 : 1), hipMemcpyDeviceToHost, s));
    if (wg)
        HIPCHK(hipMemcpyAsync(h + 5 * total, dg, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    if (wtx)
        HIPCHK(hipMemcpyAsync(h + 8 * total, dtx, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    if (wty)
        HIPCHK(hipMemcpyAsync(h + 11 * total, dty, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    off = 0;
    for (const gpx_pending *r : reqs) {
        std::memcpy(r->f, h + 3 * total + off, sizeof(double) * r->nq);
        if (r->v)
            std::memcpy(r->v, h + 4 * total + off, sizeof(double) * r->nq);
        if (r->grad)
            std::memcpy(r->grad, h + 5 * total + 3 * off, sizeof(double) * 3 * r->nq);
        if (r->tx)
            std::memcpy(r->tx, h + 8 * total + 3 * off, sizeof(double) * 3 * r->nq);
        if (r->ty)
            std::memcpy(r->ty, h + 11 * total + 3 * off, sizeof(double) * 3 * r->nq);
        off += r->nq;
    }
    return GPX_OK;
}

// Large host batches: slices of 2^18 queries through a pinned double buffer on the model's stream.  The stream runs
// copy-in / kernels / copy-out of slice i while the host fills the other buffer with slice i+1 and, once slice i-1 has
// signalled, hands its results to the caller -- so the pageable <-> pinned copies (the larger part of the PCIe-side
// cost) hide behind the device work, and the staging stays bounded (a 256^3 grid would be 1.9 GB in one piece).
constexpr size_t EVAL_SLICE = (size_t)1 << 18;
static int run_large(gpx_model *m, const gpx_pending &r)
{
    constexpr size_t SLICE = EVAL_SLICE;
    const size_t per_q = 3 + 1 + 1 + 3 + 3 + 3;  // qx qy qz | f | v | grad | tx | ty
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    const size_t S = std::min(SLICE, r.nq);
    int rc;
    if ((rc = ensure(m, (void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * S * per_q)))
        return rc;
    if (m->pin2_doubles < S * per_q) {
        for (int b = 0; b < 2; ++b) {
            if (m->pin2[b])
                HIPCHK(hipHostFree(m->pin2[b]));
            m->pin2[b] = nullptr;
        }
        m->pin2_doubles = 0;
        for (int b = 0; b < 2; ++b)
            HIPCHK(hipHostMalloc((void **)&m->pin2[b], sizeof(double) * S * per_q, hipHostMallocDefault));
        m->pin2_doubles = S * per_q;
    }
    for (int b = 0; b < 2; ++b)
        if (!m->pin2_done[b])
            HIPCHK(hipEventCreateWithFlags(&m->pin2_done[b], hipEventDisableTiming));
    hipStream_t s = m->stream;
    double *d = m->ws_host_io;
    const size_t nslices = (r.nq + S - 1) / S;
    // results of slice i (already in pinned buffer i & 1 once its event has fired) -> the caller's arrays
    auto deliver = [&](size_t i) -> int {
        const int b = (int)(i & 1);
        const size_t q0 = i * S, nn = std::min(S, r.nq - q0);
        HIPCHK(hipEventSynchronize(m->pin2_done[b]));
        const double *h = m->pin2[b];
        std::memcpy(r.f + q0, h + 3 * S, sizeof(double) * nn);
        if (r.v)
            std::memcpy(r.v + q0, h + 4 * S, sizeof(double) * nn);
        if (r.grad)
            std::memcpy(r.grad + 3 * q0, h + 5 * S, sizeof(double) * 3 * nn);
        if (r.tx)
            std::memcpy(r.tx + 3 * q0, h + 8 * S, sizeof(double) * 3 * nn);
        if (r.ty)
            std::memcpy(r.ty + 3 * q0, h + 11 * S, sizeof(double) * 3 * nn);
        return GPX_OK;
    };
    for (size_t i = 0; i < nslices; ++i) {
        const int b = (int)(i & 1);
        const size_t q0 = i * S, nn = std::min(S, r.nq - q0);
        if (i >= 2 && (rc = deliver(i - 2)))  // frees pinned buffer b; the device is busy with slice i-1 meanwhile
            break;
        double *h = m->pin2[b];
        std::memcpy(h, r.qx + q0, sizeof(double) * nn);
        std::memcpy(h + S, r.qy + q0, sizeof(double) * nn);
        std::memcpy(h + 2 * S, r.qz + q0, sizeof(double) * nn);
        HIPCHK(hipMemcpyAsync(d, h, sizeof(double) * 3 * S, hipMemcpyHostToDevice, s));
        if ((rc = evaluate_locked(m, nn, d, d + S, d + 2 * S, d + 3 * S, r.v ? d + 4 * S : nullptr,
                                  r.grad ? d + 5 * S : nullptr, r.tx ? d + 8 * S : nullptr,
                                  r.ty ? d + 11 * S : nullptr, s)))
            break;
        HIPCHK(hipMemcpyAsync(h + 3 * S, d + 3 * S, sizeof(double) * S * (r.v ? 2 : 1), hipMemcpyDeviceToHost, s));
        if (r.grad)
            HIPCHK(hipMemcpyAsync(h + 5 * S, d + 5 * S, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, s));
        if (r.tx)
            HIPCHK(hipMemcpyAsync(h + 8 * S, d + 8 * S, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, s));
        if (r.ty)
            HIPCHK(hipMemcpyAsync(h + 11 * S, d + 11 * S, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(m->pin2_done[b], s));
    }
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    for (size_t i = nslices >= 2 ? nslices - 2 : 0; i < nslices; ++i)
        if ((rc = deliver(i)))
            return rc;
    return GPX_OK;
}


extern "C" int gpx_model_evaluate(const gpx_model *cm, size_t nq, const double *qx, const double *qy,
                                  const double *qz, double *f, double *v, double *grad, double *tx, double *ty)
{
    int rc = check_query(cm, nq, qx, qy, qz, f);
    if (rc)
        return rc;
    gpx_model *m = const_cast<gpx_model *>(cm);
    gpx_pending req{nq, qx, qy, qz, f, v, grad, tx, ty};
    if (nq > COMBINE_MAX_NQ)
        return run_large(m, req);
    // flat combining (gpx_host.hpp): the calling thread either becomes the leader of a batch or waits for one
    const int rc2 = m->combiner.submit(req, [m](std::vector<gpx_pending *> &batch, std::string &err) {
        const int brc = run_requests(m, batch);
        if (brc)
            err = g_err;  // the leader's thread-local message travels to every caller of the batch
        return brc;
    });
    if (rc2)
        g_err = req.err;
    return rc2;
}

// mean on all queries -> deterministic compaction of |f| <= f_tol -> variance of the survivors only; the caller holds
// m->mtx and has set the device.
// Round 6, large grids of an exponential kernel: the mean on ALL queries is the fp32 screen of gpx_predict.hip (a proved lower
// bound of |f|), the fp64 mean kernel then runs on its candidates only and the exact test on those -- the same selected set, the
// same f and v, bit for bit: every fp64 evaluation of this function runs in evaluate_locked's fixed_order form, in which a
// query's value does not depend on the batch it sits in (the whole grid, the candidates, the survivors, a slab of
// gpx_model_sample_surface_sharded).
constexpr size_t SURFACE_SCREEN_MIN_NQ = 32768;  // below: the fp64 mean of the whole grid costs less than the extra launches
static int sample_surface_locked(gpx_model *m, size_t nq, const double *qx, const double *qy, const double *qz,
                                 double f_tol, size_t capacity, int64_t *idx, double *f, double *v, size_t *n_out)
{
    int rc;
    hipStream_t s = m->stream;
    const size_t cap = std::min(capacity, nq);
    const size_t nb = (nq + 255) / 256;
    const bool screen = nq >= SURFACE_SCREEN_MIN_NQ && f_tol > 0.0 && surface_screen_takes(m->cov);
    // device staging: qx qy qz f_all | compacted sx sy sz fs vs (cap each) | idx (cap int64) | block counters | totals
    // screen: + candidates cx cy cz cf (nq each), cidx (nq int64) | the fp32 copy of the model
    const size_t scr_doubles = screen ? nq * 5 + surface_screen_ws_doubles(m->npad) : 0;
    const size_t doubles = nq * 4 + cap * 5 + cap + nb / 2 + 4 + scr_doubles;
    if ((rc = ensure(m, (void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * doubles)))
        return rc;
    double *d = m->ws_host_io;
    double *dqx = d, *dqy = d + nq, *dqz = d + 2 * nq, *dfa = d + 3 * nq;
    double *sx = d + 4 * nq, *sy = sx + cap, *sz = sy + cap, *fs = sz + cap, *vs = fs + cap;
    long long *didx = (long long *)(vs + cap);
    unsigned *bc = (unsigned *)(didx + cap);
    unsigned long long *dtotal = (unsigned long long *)(bc + 2 * (nb / 2 + 1));
    double *cxs = (double *)(dtotal + 2), *cys = cxs + nq, *czs = cys + nq, *cfs = czs + nq;
    long long *cidx = (long long *)(cfs + nq);
    double *scr_ws = (double *)(cidx + nq);
    HIPCHK(hipMemcpyAsync(dqx, qx, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dqy, qy, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dqz, qz, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    unsigned long long total = 0;
    if (screen) {
        launch_surface_screen(m->cov, m->n, m->npad, m->d_x, m->d_y, m->d_z, m->d_alpha, m->d_meta, (long)nq, dqx, dqy, dqz,
                              dfa, scr_ws, s);
        launch_surface_select((long)nq, dfa, f_tol, bc, dtotal, nq, dqx, dqy, dqz, cidx, cfs, cxs, cys, czs, s);
        unsigned long long ncand = 0;
        HIPCHK(hipMemcpyAsync(&ncand, dtotal, sizeof(ncand), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        m->stats.surface_candidates = (double)ncand;
        if (ncand > 0) {
            if ((rc = evaluate_locked(m, (size_t)ncand, cxs, cys, czs, cfs, nullptr, nullptr, nullptr, nullptr, s, true)))
                return rc;
            launch_surface_select((long)ncand, cfs, f_tol, bc, dtotal, cap, cxs, cys, czs, didx, fs, sx, sy, sz, s, cidx);
            HIPCHK(hipMemcpyAsync(&total, dtotal, sizeof(total), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    } else {
        if ((rc = evaluate_locked(m, nq, dqx, dqy, dqz, dfa, nullptr, nullptr, nullptr, nullptr, s, true)))
            return rc;
        launch_surface_select((long)nq, dfa, f_tol, bc, dtotal, cap, dqx, dqy, dqz, didx, fs, sx, sy, sz, s);
        HIPCHK(hipMemcpyAsync(&total, dtotal, sizeof(total), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        m->stats.surface_candidates = (double)nq;
    }
    *n_out = (size_t)total;
    const size_t ns = std::min((size_t)total, cap);
    if (ns > 0) {
        if (v) {  // variance of the survivors only
            if ((rc = evaluate_locked(m, ns, sx, sy, sz, nullptr, vs, nullptr, nullptr, nullptr, s, true)))
                return rc;
            HIPCHK(hipMemcpyAsync(v, vs, sizeof(double) * ns, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipMemcpyAsync(f, fs, sizeof(double) * ns, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(idx, didx, sizeof(int64_t) * ns, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    if (total > cap)
        return fail(GPX_E_SIZE_MISMATCH, "more surface points than capacity");
    return GPX_OK;
}

extern "C" int gpx_model_sample_surface(const gpx_model *cm, size_t nq, const double *qx, const double *qy,
                                        const double *qz, double f_tol, size_t capacity, int64_t *idx, double *f,
                                        double *v, size_t *n_out)
{
    if (!n_out || !idx)
        return fail(GPX_E_NULL, "Empty output pointer");
    *n_out = 0;
    int rc = check_query(cm, nq, qx, qy, qz, f);
    if (rc)
        return rc;
    if (!(f_tol >= 0.0))
        return fail(GPX_E_BAD_ARG, "f_tol must be non-negative");
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    return sample_surface_locked(m, nq, qx, qy, qz, f_tol, capacity, idx, f, v, n_out);
}

// ---- one call, several replicas: the query grid of ONE evaluate / sampleSurface call cut into contiguous slabs ------------------
// (north star: "query-grid shards"; the caller is the reference node's single process, src/gp_node.cpp:1025-1038.)  replicas[i]
// are models that predict alike -- a model and its gpx_model_replicate copies, on any devices -- and slab i goes to replica i from
// its own host thread, results written in place into the caller's arrays.  The cut follows gpx_slab_range (the remainder spread
// over the low ranks: sharding.slab_range of the one-process-per-GPU form), applied
//   * by evaluate to the SLICES of 2^18 queries the single call itself is pipelined in (run_large): the order of the mean's and
//     the variance's sums depends on the size of the device batch a query sits in (how the point range is split over workgroups,
//     whether the variance tiles are launched in pairs), so a slab is made of whole slices and every slice is evaluated exactly
//     as the single call evaluates it -- equal bit for bit, for every model size; a call of one slice is not cut;
//   * by sample_surface to the queries themselves: all of its fp64 arithmetic is in evaluate_locked's fixed_order form.
extern "C" void gpx_slab_range(size_t nq, int rank, int world, size_t *lo, size_t *hi)
{
    size_t a = 0, b = 0;
    if (world > 0 && rank >= 0 && rank < world) {
        const size_t base = nq / (size_t)world, rem = nq % (size_t)world, r = (size_t)rank;
        a = r * base + std::min(r, rem);
        b = a + base + (r < rem ? 1 : 0);
    }
    if (lo)
        *lo = a;
    if (hi)
        *hi = b;
}

namespace {
int check_replicas(const gpx_model *const *replicas, int n)
{
    if (!replicas)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (n <= 0)
        return fail(GPX_E_BAD_ARG, "n_replicas must be positive");
    for (int i = 0; i < n; ++i) {
        if (!replicas[i])
            return fail(GPX_E_NULL, "Empty Model pointer");
        if (!replicas[i]->ready)
            return fail(GPX_E_STATE, "model is not ready (shell not committed or create failed)");
        if (replicas[i]->n != replicas[0]->n || replicas[i]->cov.id != replicas[0]->cov.id)
            return fail(GPX_E_BAD_ARG, "replicas are not copies of one model");
        for (int j = 0; j < i; ++j)
            if (replicas[j] == replicas[i])
                return fail(GPX_E_BAD_ARG, "the same model handle twice (use gpx_model_replicate for a second copy on a device)");
    }
    return GPX_OK;
}
// run work(i) for every i in [0, n) with a non-empty slab on its own host thread (the caller's thread takes the first);
// the first failing slab's status and message are the call's
template <typename F>
int run_slabs(int n, const std::vector<size_t> &len, F work)
{
    std::vector<int> rcs((size_t)n, GPX_OK);
    std::vector<std::string> errs((size_t)n);
    std::vector<std::thread> th;
    int first = -1;
    for (int i = 0; i < n; ++i) {
        if (!len[(size_t)i])
            continue;
        if (first < 0) {
            first = i;
            continue;
        }
        auto job = [&, i]() {
            rcs[(size_t)i] = work(i);
            if (rcs[(size_t)i])
                errs[(size_t)i] = g_err;
        };
        try {
            th.emplace_back(job);
        } catch (...) {  // (no thread to be had: the slab runs on the caller's thread -- nothing may throw across the C boundary)
            job();
        }
    }
    if (first >= 0) {
        rcs[(size_t)first] = work(first);
        if (rcs[(size_t)first])
            errs[(size_t)first] = g_err;
    }
    for (std::thread &t : th)
        t.join();
    for (int i = 0; i < n; ++i)
        if (rcs[(size_t)i] && rcs[(size_t)i] != GPX_E_SIZE_MISMATCH)
            return fail(rcs[(size_t)i], "slab " + std::to_string(i) + ": " + errs[(size_t)i]);
    return GPX_OK;
}
}  // namespace

extern "C" int gpx_model_evaluate_sharded(const gpx_model *const *replicas, int n_replicas, size_t nq, const double *qx,
                                          const double *qy, const double *qz, double *f, double *v, double *grad, double *tx,
                                          double *ty)
{
    int rc = check_replicas(replicas, n_replicas);
    if (rc || (rc = check_query(replicas[0], nq, qx, qy, qz, f)))
        return rc;
    const size_t nslices = (nq + EVAL_SLICE - 1) / EVAL_SLICE;
    if (nslices < 2 || n_replicas < 2)  // one slice: the single call's own path (flat combining, the one-launch path, ...)
        return gpx_model_evaluate(replicas[0], nq, qx, qy, qz, f, v, grad, tx, ty);
    std::vector<size_t> lo((size_t)n_replicas), len((size_t)n_replicas);
    for (int i = 0; i < n_replicas; ++i) {
        size_t a, b;
        gpx_slab_range(nslices, i, n_replicas, &a, &b);
        lo[(size_t)i] = std::min(nq, a * EVAL_SLICE);
        len[(size_t)i] = std::min(nq, b * EVAL_SLICE) - lo[(size_t)i];
    }
    return run_slabs(n_replicas, len, [&](int i) {
        const size_t a = lo[(size_t)i];
        gpx_pending req{len[(size_t)i], qx + a, qy + a, qz + a, f + a, v ? v + a : nullptr, grad ? grad + 3 * a : nullptr,
                        tx ? tx + 3 * a : nullptr, ty ? ty + 3 * a : nullptr};
        return run_large(const_cast<gpx_model *>(replicas[i]), req);  // (the pipelined slices, whatever the slab's length)
    });
}

// gpx_model_sample_surface over the slabs: every replica selects and evaluates the survivors of its slab into its own
// staging, the slabs are then laid end to end (positions ascending, as the single call returns them).
extern "C" int gpx_model_sample_surface_sharded(const gpx_model *const *replicas, int n_replicas, size_t nq, const double *qx,
                                                const double *qy, const double *qz, double f_tol, size_t capacity, int64_t *idx,
                                                double *f, double *v, size_t *n_out)
{
    if (!n_out || !idx)
        return fail(GPX_E_NULL, "Empty output pointer");
    *n_out = 0;
    int rc = check_replicas(replicas, n_replicas);
    if (rc || (rc = check_query(replicas[0], nq, qx, qy, qz, f)))
        return rc;
    if (!(f_tol >= 0.0))
        return fail(GPX_E_BAD_ARG, "f_tol must be non-negative");
    const size_t nr = (size_t)n_replicas;
    std::vector<size_t> lo(nr), len(nr), got(nr, 0);
    for (size_t i = 0; i < nr; ++i) {
        size_t hi;
        gpx_slab_range(nq, (int)i, n_replicas, &lo[i], &hi);
        len[i] = hi - lo[i];
    }
    // capacity >= nq (what the header shim passes): slab i writes its survivors straight into the caller's arrays at its own offset
    // lo_i -- it has at most len_i of them -- and the slabs are closed up afterwards; a smaller capacity takes staging per slab
    const bool in_place = capacity >= nq;
    std::vector<std::vector<int64_t>> sidx(in_place ? 0 : nr);
    std::vector<std::vector<double>> sf(in_place ? 0 : nr), sv(in_place ? 0 : nr);
    rc = run_slabs(n_replicas, len, [&](int ii) {
        const size_t i = (size_t)ii, a = lo[i], cap = std::min(capacity, len[i]);
        if (in_place)
            return gpx_model_sample_surface(replicas[i], len[i], qx + a, qy + a, qz + a, f_tol, len[i], idx + a, f + a, v ? v + a : nullptr,
                                            &got[i]);
        sidx[i].resize(cap + 1);  // (+1: never a null data() for a capacity of 0)
        sf[i].resize(cap + 1);
        if (v)
            sv[i].resize(cap + 1);
        return gpx_model_sample_surface(replicas[i], len[i], qx + a, qy + a, qz + a, f_tol, cap, sidx[i].data(), sf[i].data(),
                                        v ? sv[i].data() : nullptr, &got[i]);
    });
    if (rc)
        return rc;
    size_t total = 0, w = 0;
    for (size_t i = 0; i < nr; ++i) {
        const size_t have = std::min(got[i], std::min(capacity, len[i]));
        if (in_place) {
            for (size_t k = 0; k < have; ++k)
                idx[lo[i] + k] += (int64_t)lo[i];
            if (w != lo[i] && have) {  // (w <= lo_i: the regions may overlap, the move runs upwards)
                std::memmove(idx + w, idx + lo[i], sizeof(int64_t) * have);
                std::memmove(f + w, f + lo[i], sizeof(double) * have);
                if (v)
                    std::memmove(v + w, v + lo[i], sizeof(double) * have);
            }
            w += have;
        } else {
            for (size_t k = 0; k < have && w < capacity; ++k, ++w) {
                idx[w] = sidx[i][k] + (int64_t)lo[i];
                f[w] = sf[i][k];
                if (v)
                    v[w] = sv[i][k];
            }
        }
        total += got[i];
    }
    *n_out = total;
    if (total > capacity)
        return fail(GPX_E_SIZE_MISMATCH, "more surface points than capacity");
    return GPX_OK;
}

// ---- the surface-following sampler of the node (src/gp_node.cpp:1102-1291: marchingSampling + marchingCubes) -------
// The reference walks from cube to cube with one std::thread per neighbour and one single-point evaluate(f, v) per
// lattice point.  Here the walk is a breadth-first frontier on the host -- cubes by integer offset from the start cube,
// a hash set for "already sampled" -- and every frontier is ONE device batch: the mean on all (steps + 1)^3 lattice
// points of all its cubes, compaction of |f| <= f_tol, the variance for the kept points only.  Lattice coordinates and
// cube centres are computed in float exactly as the reference does (:1210-1212, :1271-1281); order of the output:
// cubes in discovery order (faces -x +x -y +y -z +z), lattice points i, j, k -- see oracle/gp_oracle.c:orc_march_surface
// for the points where the reference's own order is left to thread timing.
namespace {
inline float f_add(float a, float b)
{
    volatile float r = a + b;  // one rounding per operation, no contraction into an fma
    return r;
}
inline float f_mul(float a, float b)
{
    volatile float r = a * b;
    return r;
}
struct MarchCube {
    int ox, oy, oz;
    float cx, cy, cz;
};
inline uint64_t march_key(int a, int b, int c)
{
    return ((uint64_t)(a + 1048576) << 42) | ((uint64_t)(b + 1048576) << 21) | (uint64_t)(c + 1048576);
}
}  // namespace

extern "C" int gpx_model_march_surface(const gpx_model *cm, const double *start_xyz, double leaf_d, double pass_d,
                                       double f_tol, size_t max_cubes, size_t capacity, double *xyz, double *f,
                                       double *v, size_t *n_out, size_t *n_cubes)
{
    if (!n_out || !xyz || !f)
        return fail(GPX_E_NULL, "Empty output pointer");
    *n_out = 0;
    if (n_cubes)
        *n_cubes = 0;
    if (!cm)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!cm->ready)
        return fail(GPX_E_STATE, "model is not ready (shell not committed or create failed)");
    const float leaf = (float)leaf_d, pass = (float)pass_d;
    if (!(leaf > 0.0f) || !(pass > 0.0f) || !(f_tol >= 0.0) || !std::isfinite(leaf) || !std::isfinite(pass))
        return fail(GPX_E_BAD_ARG, "leaf and pass must be positive, f_tol non-negative");
    const long steps = std::lround(leaf / pass);  // :1201
    if (steps < 1 || steps > 64)
        return fail(GPX_E_BAD_ARG, "round(leaf / pass) must be between 1 and 64");
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    int rc;
    float sx = 0, sy = 0, sz = 0;
    if (start_xyz) {
        sx = (float)start_xyz[0], sy = (float)start_xyz[1], sz = (float)start_xyz[2];
    } else {
        // :1126-1152: the first point of the 0.1 lattice on [-1.1, 1.1]^3 (accumulated doubles, x outermost) with
        // |f| <= f_tol, found with one batched mean instead of up to 12167 single-point calls
        std::vector<double> ax;
        for (double t = -1.1; t <= 1.1; t += 0.1)
            ax.push_back(t);
        const size_t g = ax.size(), nq = g * g * g;
        std::vector<double> qx(nq), qy(nq), qz(nq), ff(nq);
        for (size_t i = 0, q = 0; i < g; ++i)
            for (size_t j = 0; j < g; ++j)
                for (size_t k = 0; k < g; ++k, ++q)
                    qx[q] = ax[i], qy[q] = ax[j], qz[q] = ax[k];
        if ((rc = ensure(m, (void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * 4 * nq)))
            return rc;
        double *d = m->ws_host_io;
        HIPCHK(hipMemcpyAsync(d, qx.data(), sizeof(double) * nq, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(d + nq, qy.data(), sizeof(double) * nq, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(d + 2 * nq, qz.data(), sizeof(double) * nq, hipMemcpyHostToDevice, m->stream));
        if ((rc = evaluate_locked(m, nq, d, d + nq, d + 2 * nq, d + 3 * nq, nullptr, nullptr, nullptr, nullptr, m->stream)))
            return rc;
        HIPCHK(hipMemcpyAsync(ff.data(), d + 3 * nq, sizeof(double) * nq, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        size_t hit = nq;
        for (size_t q = 0; q < nq; ++q)
            if (std::fabs(ff[q]) <= f_tol) {
                hit = q;
                break;
            }
        if (hit == nq)
            return fail(GPX_E_EMPTY, "No starting point found. Relax grid pass.");  // :1155
        sx = (float)qx[hit], sy = (float)qy[hit], sz = (float)qz[hit];
    }
    const size_t per_cube = (size_t)(steps + 1) * (steps + 1) * (steps + 1);
    std::vector<MarchCube> frontier{{0, 0, 0, sx, sy, sz}}, next;
    std::vector<uint64_t> seen_keys;  // open addressing, 0 = empty (march_key is never 0)
    size_t seen_cap = 4096, seen_cnt = 0;
    seen_keys.assign(seen_cap, 0);
    auto seen_insert = [&](uint64_t key) -> bool {  // true if newly inserted
        if ((seen_cnt + 1) * 2 > seen_cap) {
            std::vector<uint64_t> old;
            old.swap(seen_keys);
            seen_cap *= 4;
            seen_keys.assign(seen_cap, 0);
            for (uint64_t k : old)
                if (k) {
                    size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) % seen_cap;
                    while (seen_keys[h])
                        h = (h + 1) % seen_cap;
                    seen_keys[h] = k;
                }
        }
        size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) % seen_cap;
        while (seen_keys[h]) {
            if (seen_keys[h] == key)
                return false;
            h = (h + 1) % seen_cap;
        }
        seen_keys[h] = key;
        ++seen_cnt;
        return true;
    };
    seen_insert(march_key(0, 0, 0));
    size_t done_cubes = 0, kept = 0;
    std::vector<double> qx, qy, qz, fs, vs;
    std::vector<int64_t> idx;
    bool truncated = false;
    while (!frontier.empty() && done_cubes < max_cubes) {
        if (frontier.size() > max_cubes - done_cubes)
            frontier.resize(max_cubes - done_cubes);
        const size_t nc = frontier.size(), nq = nc * per_cube;
        qx.resize(nq), qy.resize(nq), qz.resize(nq);
        for (size_t c = 0, q = 0; c < nc; ++c) {
            const MarchCube &cb = frontier[c];
            const float hx = f_add(cb.cx, -(leaf / 2)), hy = f_add(cb.cy, -(leaf / 2)), hz = f_add(cb.cz, -(leaf / 2));
            for (long i = 0; i <= steps; ++i)
                for (long j = 0; j <= steps; ++j)
                    for (long k = 0; k <= steps; ++k, ++q) {
                        qx[q] = (double)f_add(hx, f_mul((float)i, pass));  // :1210-1212
                        qy[q] = (double)f_add(hy, f_mul((float)j, pass));
                        qz[q] = (double)f_add(hz, f_mul((float)k, pass));
                    }
        }
        idx.resize(nq), fs.resize(nq), vs.resize(nq);
        size_t ns = 0;
        rc = sample_surface_locked(m, nq, qx.data(), qy.data(), qz.data(), f_tol, nq, idx.data(), fs.data(),
                                   v ? vs.data() : nullptr, &ns);
        if (rc)
            return rc;
        next.clear();
        std::vector<unsigned char> where(nc * 6, 0);
        for (size_t t = 0; t < ns; ++t) {
            const size_t q = (size_t)idx[t], c = q / per_cube, r = q % per_cube;
            const long i = (long)(r / ((steps + 1) * (steps + 1))), j = (long)(r / (steps + 1)) % (steps + 1),
                       k = (long)(r % (steps + 1));
            if (kept < capacity) {
                xyz[3 * kept] = qx[q], xyz[3 * kept + 1] = qy[q], xyz[3 * kept + 2] = qz[q];
                f[kept] = fs[t];
                if (v)
                    v[kept] = vs[t];
            } else {
                truncated = true;
            }
            ++kept;
            unsigned char *w = &where[c * 6];  // :1240-1251
            w[0] |= i == 0, w[1] |= i == steps, w[2] |= j == 0, w[3] |= j == steps, w[4] |= k == 0, w[5] |= k == steps;
        }
        for (size_t c = 0; c < nc; ++c)
            for (int w = 0; w < 6; ++w) {  // :1266-1288
                if (!where[c * 6 + w])
                    continue;
                MarchCube nb = frontier[c];
                switch (w) {
                case 0: nb.ox -= 1, nb.cx = f_add(nb.cx, -leaf); break;
                case 1: nb.ox += 1, nb.cx = f_add(nb.cx, leaf); break;
                case 2: nb.oy -= 1, nb.cy = f_add(nb.cy, -leaf); break;
                case 3: nb.oy += 1, nb.cy = f_add(nb.cy, leaf); break;
                case 4: nb.oz -= 1, nb.cz = f_add(nb.cz, -leaf); break;
                default: nb.oz += 1, nb.cz = f_add(nb.cz, leaf); break;
                }
                if (std::abs(nb.ox) > 1000000 || std::abs(nb.oy) > 1000000 || std::abs(nb.oz) > 1000000)
                    continue;
                if (seen_insert(march_key(nb.ox, nb.oy, nb.oz)))
                    next.push_back(nb);
            }
        done_cubes += nc;
        frontier.swap(next);
    }
    *n_out = kept;
    if (n_cubes)
        *n_cubes = done_cubes;
    if (truncated)
        return fail(GPX_E_SIZE_MISMATCH, "more surface points than capacity");
    return GPX_OK;
}

// ---- AtlasBase::project, batched and device-resident (reference include/atlas/atlas.hpp:201-276) -------------
extern "C" int gpx_model_project(const gpx_model *cm, size_t nq, const double *x, const double *y, const double *z,
                                 const double *normal, const gpx_project_options *opt, double *out_xyz, double *out_f,
                                 int32_t *out_iter, int32_t *out_status)
{
    if (!out_xyz || !normal)
        return fail(GPX_E_NULL, "Empty data pointer");
    int rc = check_query(cm, nq, x, y, z, out_xyz);
    if (rc)
        return rc;
    gpx_project_options o{1e-2, 1e-7, 0.001, 500, {0, 0, 0}};
    if (opt)
        o = *opt;
    if (!(o.f_tol >= 0.0) || !(o.improve_tol >= 0.0) || o.max_iter < 0 || !std::isfinite(o.step_mul))
        return fail(GPX_E_BAD_ARG, "project options: tolerances and max_iter must be non-negative, step_mul finite");
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // device state: cx cy cz f_cur f_new (nq each) | g grad_new (3 nq each) | iter status (nq ints each) | active
    const size_t doubles = nq * 5 + nq * 6 + nq + 2;
    if ((rc = ensure(m, (void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * doubles)))
        return rc;
    double *d = m->ws_host_io;
    double *cx = d, *cy = d + nq, *cz = d + 2 * nq, *fcur = d + 3 * nq, *fnew = d + 4 * nq;
    double *g = d + 5 * nq, *gnew = d + 8 * nq;
    int *iter = (int *)(d + 11 * nq), *status = iter + nq;
    unsigned *active = (unsigned *)(d + 12 * nq);
    HIPCHK(hipMemcpyAsync(cx, x, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(cy, y, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(cz, z, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g, normal, sizeof(double) * 3 * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(iter, 0, sizeof(int) * 2 * nq, s));
    bool fused = gpxh::switches().project_fused != 0;
    if (fused)  // the whole loop in one launch when the model fits the LDS (N <= 4096)
        fused = launch_project_fused(m->cov, m->npad, m->d_x, m->d_y, m->d_z, m->d_alpha, (long)nq, o.f_tol,
                                     o.improve_tol, o.step_mul, o.max_iter, cx, cy, cz, g, fcur, iter, status, s);
    if (!fused) {
        // the mean at the start points (:225 of the first iteration; also the answer when max_iter == 0)
        if ((rc = evaluate_locked(m, nq, cx, cy, cz, fcur, nullptr, nullptr, nullptr, nullptr, s)))
            return rc;
        for (int it = 0; it < o.max_iter; ++it) {
            HIPCHK(hipMemsetAsync(active, 0, sizeof(unsigned), s));
            launch_project_pre((long)nq, o.f_tol, o.step_mul, cx, cy, cz, g, fcur, status, s);
            if ((rc = evaluate_locked(m, nq, cx, cy, cz, fnew, nullptr, gnew, nullptr, nullptr, s)))
                return rc;
            launch_project_post((long)nq, o.improve_tol, o.max_iter, fnew, gnew, g, fcur, iter, status, active, s);
            if ((it & 7) == 7 || it + 1 == o.max_iter) {  // look at the device only every 8 iterations
                unsigned left = 0;
                HIPCHK(hipMemcpyAsync(&left, active, sizeof(left), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (left == 0)
                    break;
            }
        }
    }
    std::vector<double> hc(3 * nq);
    std::vector<int> hs(2 * nq);
    HIPCHK(hipMemcpyAsync(hc.data(), cx, sizeof(double) * 3 * nq, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hs.data(), iter, sizeof(int) * 2 * nq, hipMemcpyDeviceToHost, s));
    if (out_f)
        HIPCHK(hipMemcpyAsync(out_f, fcur, sizeof(double) * nq, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < nq; ++i) {
        out_xyz[3 * i] = hc[i];
        out_xyz[3 * i + 1] = hc[nq + i];
        out_xyz[3 * i + 2] = hc[2 * nq + i];
        int st = hs[nq + i];
        if (st == 0)
            st = 3;  // max_iter == 0: the loop of the reference is never entered
        if (out_iter)
            out_iter[i] = hs[i];
        if (out_status)
            out_status[i] = st;
    }
    return GPX_OK;
}

extern "C" int gpx_model_prepare_variance(gpx_model *m)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!m->ready)
        return fail(GPX_E_STATE, "model is not ready");
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    return build_inverse(m);
}

extern "C" int gpx_model_sync(const gpx_model *m)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GPX_OK;
}

void resolve_eval_stats(gpx_model *m)
{
    if (!m->stats_eval_pending)
        return;
    if (hipStreamSynchronize(m->stream) != hipSuccess)
        return;
    float ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_M0], m->ev[EV_M1]) == hipSuccess)
        m->stats.t_mean_ms = ms;
    m->stats.t_var_ms = 0;
    if (m->eval_had_var && hipEventElapsedTime(&ms, m->ev[EV_M1], m->ev[EV_V1]) == hipSuccess)
        m->stats.t_var_ms = ms;
    double tg = 0;
    for (size_t i = 0; i < m->gemm_ev_used_var; ++i) {
        const size_t k = m->gemm_ev_used_factor + i;
        if (hipEventElapsedTime(&ms, m->gemm_ev[2 * k], m->gemm_ev[2 * k + 1]) == hipSuccess)
            tg += ms;
    }
    m->stats.t_var_gemm_ms = tg;
    m->stats.var_gemm_launches = (int64_t)m->gemm_ev_used_var;
    double tk = 0;
    for (size_t i = 0; i < m->kqp_ev_used && 2 * i + 1 < m->kqp_ev.size(); ++i)
        if (hipEventElapsedTime(&ms, m->kqp_ev[2 * i], m->kqp_ev[2 * i + 1]) == hipSuccess)
            tk += ms;
    m->stats.t_var_kqp_ms = tk;
    m->stats_eval_pending = false;
}

}  // namespace gpxh
