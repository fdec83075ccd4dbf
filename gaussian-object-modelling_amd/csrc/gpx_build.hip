// gpx_build.hip -- everything between the host arrays and a ready model: device buffers, the kernel matrix, the
// blocked right-looking LDL^T (replaces Eigen::LDLT::compute, gp_regressor.hpp:161-162), alpha by block
// substitution with fp64 matrix-free residual refinement (:163), normals (:166-181), the inverse factor X = L^-1 by
// recursive doubling, the rank-n append of update() and the MIXED-precision demotion.
#include "gpx_model.hpp"
#include "gpx_small.hpp"

namespace gpxh {

void free_dev(gpx_model *m)
{
    quiesce(m);
    auto F = [](void *p) { big_free(p); };  // parks the buffers that came from big_alloc (>= BIG_POOL_MIN), hipFree otherwise
    F(m->dvecs);
    F(m->blob0);
    F(m->tvecs);
    F(m->Kmat);
    F(m->linv);
    F(m->Wp);
    F(m->X);
    F(m->d_info);
    F(m->d_tmax);
    F(m->d_tij);
    F(m->ws_pred);
    F(m->ws_kqp);
    F(m->ws_partial);
    F(m->ws_coef);
    F(m->ws_grad);
    F(m->ws_host_io);
    F(m->ws_small);
    m->ws_small = nullptr;
    F(m->d_normals);
    if (m->pin)
        (void)hipHostFree(m->pin);
    m->pin = nullptr;
    m->pin_doubles = 0;
    for (int b = 0; b < 2; ++b) {
        if (m->pin2[b])
            (void)hipHostFree(m->pin2[b]);
        m->pin2[b] = nullptr;
        if (m->pin2_done[b])
            (void)hipEventDestroy(m->pin2_done[b]);
        m->pin2_done[b] = nullptr;
    }
    m->pin2_doubles = 0;
    m->dvecs = nullptr;
    m->blob0 = m->tvecs = m->Kmat = m->linv = m->Wp = m->X = nullptr;
    m->d_info = nullptr;
    m->d_tmax = nullptr;
    m->d_tij = nullptr;
    m->ws_pred = nullptr;
    m->ws_kqp = m->ws_partial = m->ws_coef = nullptr;
    m->ws_coef_bytes = 0;
    m->ws_grad = nullptr;
    m->ws_host_io = nullptr;
    m->d_normals = nullptr;
    m->ws_pred_doubles = m->ws_kqp_bytes = m->ws_partial_bytes = m->ws_grad_doubles = m->ws_host_io_doubles = 0;
    for (auto &e : m->gemm_ev)
        (void)hipEventDestroy(e);
    m->gemm_ev.clear();
    for (auto &e : m->kqp_ev)
        (void)hipEventDestroy(e);
    m->kqp_ev.clear();
    m->kqp_ev_used = 0;
    // timings of an evaluate() that were not read yet refer to the events just destroyed
    m->stats_eval_pending = false;
    m->gemm_ev_used_var = m->gemm_ev_used_factor = 0;
}


// Waits for everything this model has enqueued -- its own two streams and, through the workspace event, the last
// evaluation a caller put on a stream of its own -- so that its buffers can be released or parked.  This replaces
// the device-wide synchronisation the pool did in round 2: other models and threads on the device keep running.
void quiesce(gpx_model *m)
{
    if (m->stream)
        (void)hipStreamSynchronize(m->stream);
    if (m->stream2)
        (void)hipStreamSynchronize(m->stream2);
    if (m->ws_in_flight && m->ev[EV_WS])
        (void)hipEventSynchronize(m->ev[EV_WS]);
    m->ws_in_flight = false;
}

int ensure(gpx_model *m, void **p, size_t *have, size_t need)
{
    if (*have >= need && *p)
        return GPX_OK;
    if (*p)
        quiesce(m);  // an earlier evaluation may still be using the buffer that is about to be replaced
    big_free(*p);
    *p = nullptr;
    *have = 0;
    HIPCHK(big_alloc(p, need));
    *have = need;
    return GPX_OK;
}

int alloc_blob0(gpx_model *m, size_t esz, void **blob, size_t *bytes)
{
    const size_t np = (size_t)m->npad;
    *bytes = sizeof(double) * np * (5 + VAR_NCORR) + esz * np * 4 + sizeof(double) * BLOB_META;
    HIPCHK(big_alloc(blob, *bytes));
    return GPX_OK;
}

void carve_blob0(gpx_model *m)
{
    const size_t np = (size_t)m->npad, e = m->esz;
    m->d_x = (double *)m->blob0;
    m->d_y = m->d_x + np;
    m->d_z = m->d_y + np;
    m->d_alpha = m->d_z + np;
    m->d_dinv64 = m->d_alpha + np;
    m->d_corr = m->d_dinv64 + np;
    char *b = (char *)(m->d_corr + (size_t)VAR_NCORR * np);
    m->t_x = b;
    m->t_y = b + e * np;
    m->t_z = b + 2 * e * np;
    m->t_dinv = b + 3 * e * np;
    m->d_meta = (double *)(b + 4 * e * np);
}

int alloc_model(gpx_model *m)
{
    const size_t np = (size_t)m->npad, e = m->esz;
    int rc = alloc_blob0(m, e, &m->blob0, &m->blob0_bytes);
    if (rc)
        return rc;
    carve_blob0(m);
    HIPCHK(big_alloc((void **)&m->dvecs, sizeof(double) * (np * 4 + 8)));
    m->d_lab = m->dvecs;
    m->d_s2 = m->d_lab + np;
    m->d_r = m->d_s2 + np;
    m->d_f = m->d_r + np;
    m->d_rmax = m->d_f + np;
    HIPCHK(big_alloc(&m->tvecs, e * np * 6));
    char *b = (char *)m->tvecs;
    m->t_s2 = b;
    m->t_d = b + e * np;
    m->t_b = b + 2 * e * np;
    m->t_yv = b + 3 * e * np;
    m->t_xs = b + 4 * e * np;
    m->t_alpha = b + 5 * e * np;
    HIPCHK(hipMalloc((void **)&m->d_info, sizeof(int) * 8));
    return GPX_OK;
}

hipEvent_t *gemm_events(gpx_model *m, size_t idx)
{
    while (m->gemm_ev.size() < 2 * (idx + 1)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess)
            return nullptr;
        m->gemm_ev.push_back(e);
    }
    return &m->gemm_ev[2 * idx];
}

// ---- L y = b ; y *= 1/D ; L^T x = y on T vectors (block substitution with inverse blocks) ------
// Default: one launch per direction (tri_solve_kernel).  force_steps: one launch per block step -- the path build_model
// falls back to when a workgroup of the one-launch kernel gave up waiting (info[5]; GPX_WAIT_BUDGET_US=0 forces that in the
// tests, which hold the two paths to each other).
static void solve_ldl(gpx_model *m, void *b /*consumed*/, void *ytmp, void *x, bool force_steps)
{
    if (!force_steps) {
        launch_tri_solve(m->prec, m->nblk, m->Kmat, m->npad, m->linv, m->t_dinv, b, ytmp, x, m->d_info, m->stream,
                         wait_budget_ticks(m->npad));
        return;
    }
    for (int kb = 0; kb < m->nblk; ++kb)
        launch_fwd_step(m->prec, kb, m->nblk, m->Kmat, m->npad, m->linv, b, ytmp, m->stream);
    launch_scale_vec(m->prec, m->npad, ytmp, m->t_dinv, m->stream);
    for (int kb = m->nblk - 1; kb >= 0; --kb)
        launch_bwd_step(m->prec, kb, m->Kmat, m->npad, m->linv, ytmp, x, m->stream);
}

// ---- X = L^-1 by recursive doubling: X21 = -X22 * (L21 * X11) ------------------------------------
// L, X, Tw point at the top-left corner of an np x np diagonal part of matrices with leading dimension ld.
static void trtri_levels(int prec, size_t e, char *L, char *X, char *Tw, int np, hipStream_t st, long ld = 0)
{
    if (ld == 0)
        ld = np;
    auto off = [&](size_t r, size_t c) { return (r * ld + c) * e; };
    for (long b = TILE; b < np; b *= 2) {
        // nodes p: left = [p*2b, p*2b+b), right = [p*2b+b, min(p*2b+2b, np))
        int P = 0;
        for (long base = 0; base + b < np; base += 2 * b)
            ++P;
        if (P == 0)
            break;
        const long last_base = (long)(P - 1) * 2 * b;
        const int m_last = (int)std::min<long>(b, np - (last_base + b));
        const long stride = 2 * b * ld + 2 * b;
        GemmArgs g1;  // T = L21 * X11   (B lower, [k][n])
        g1.A = L + off(b, 0), g1.lda = ld;
        g1.B = X + off(0, 0), g1.ldb = ld;
        g1.C = Tw + off(b, 0), g1.ldc = ld;
        g1.M = (int)b, g1.N = (int)b, g1.K = (int)b;
        g1.sA = g1.sB = g1.sC = stride;
        g1.batch = P, g1.M_last = m_last;
        g1.nn = 1, g1.b_lower = 1;
        launch_gemm(prec, g1, st);
        GemmArgs g2;  // X21 = -X22 * T  (A lower)
        g2.A = X + off(b, b), g2.lda = ld;
        g2.B = Tw + off(b, 0), g2.ldb = ld;
        g2.C = X + off(b, 0), g2.ldc = ld;
        g2.M = (int)b, g2.N = (int)b, g2.K = (int)b;
        g2.sA = g2.sB = g2.sC = stride;
        g2.batch = P, g2.M_last = m_last, g2.k_eq_m = 1;
        g2.nn = 1, g2.a_lower = 1;
        g2.alpha = -1.0;
        launch_gemm(prec, g2, st);
    }
}


// ---- blocked right-looking LDL^T -----------------------------------------------------------------
// Outer panels of 256 columns = 2 diagonal blocks of 128 (GPX_PANEL=512: 4 blocks): per block the diagonal LDL^T
// (+ inverse), the panel solve as a GEMM with the inverse block (W = A21 Linv^T to the workspace, L21 = W D^-1 in
// place) and the update of the remaining columns of the panel; then ONE trailing update with K = panel width.
// Measured at N = 16384 fp32: 512-wide panels give a trailing tile 16 k-tiles instead of 8 (100.6 -> 112 TFLOP/s,
// LDL^T 31.8 -> 30.8 ms), but the longer fp32 accumulations cost the ill-conditioned thin-plate system accuracy
// (alpha after two refinement steps 2.9e-5 instead of < 1e-5 of the fp64 result), so 256 stays the fp32 default;
// fp64 factorisations use 512 (GPX_PANEL=256 / 512 forces either).
// Columns before c_start (a multiple of 128) are taken as already factorised and applied (rank-n update).
static void factorize(gpx_model *m, int c_start = 0)
{
    // The kernel matrix is the identity on the padding (N is padded to a multiple of 256): 128-blocks that lie
    // entirely in it are already factorised (L = I, D = 1) and are only given their identity inverse, so the loops
    // below stop at the last block that holds a training point -- N = 277 factorises 3 diagonal blocks, not 4.
    const int np_full = m->npad;
    const int np = std::min(np_full, (m->n + TILE - 1) / TILE * TILE);
    launch_identity_blocks(m->prec, np / TILE, m->nblk, m->linv, m->t_d, m->t_dinv, m->stream);
    const int ldk = np_full;
    const size_t e = m->esz;
    char *K = (char *)m->Kmat;
    char *W = (char *)m->Wp;
    auto Kp = [&](size_t r, size_t c) { return (void *)(K + (r * ldk + c) * e); };
    auto Wpp = [&](size_t r, size_t c) { return (void *)(W + (r * WIDE_PANEL + c) * e); };
    size_t gemm_idx = 0;
    m->factor_gemm_flops = 0;
    // one 128-wide step: diagonal block, panel solve (W to column `wcol` of the workspace, L21 in place)
    // the 4-wave diagonal-block kernel is for launches that may run beside a GEMM on the other stream: by default every
    // launch off the main stream, in the round-2 schedule (narrow_explicit) whenever a rest update may be in flight
    bool narrow_diag = false, narrow_explicit = false;
    auto block_step = [&](int cc, int wcol, hipStream_t st) {
        const int r0 = cc + TILE;
        launch_diag_ldl(m->prec, Kp(cc, cc), ldk, m->linv, m->t_d, m->t_dinv, m->d_info, cc / TILE, st,
                        narrow_explicit ? narrow_diag : st != m->stream);
        if (r0 >= np)
            return;
        GemmArgs t;  // W = A21 * Linv^T ; L21 = W * D^-1 (in place)
        t.A = Kp(r0, cc), t.lda = ldk;
        t.B = (char *)m->linv + (size_t)(cc / TILE) * TILE * TILE * e, t.ldb = TILE;
        t.C = Kp(r0, cc), t.ldc = ldk;
        t.M = np - r0, t.N = TILE, t.K = TILE;
        t.b_lower = 1;
        t.epi = EPI_TRSM;
        t.W = Wpp(r0, wcol), t.ldw = WIDE_PANEL;
        t.colscale = (char *)m->t_dinv + (size_t)cc * e;
        launch_gemm(m->prec, t, st);
    };
    // the remaining columns of a panel [.., pend) (incl. the next diagonal block) -= W_h * L_h^T
    auto half_update = [&](int cc, int wcol, int pend, hipStream_t st) {
        const int r0 = cc + TILE;
        GemmArgs s;
        s.A = Wpp(r0, wcol), s.lda = WIDE_PANEL;
        s.B = Kp(r0, cc), s.ldb = ldk;
        s.C = Kp(r0, r0), s.ldc = ldk;
        s.M = np - r0, s.N = pend - r0, s.K = TILE;
        s.alpha = -1.0, s.beta = 1;
        launch_gemm(m->prec, s, st);
    };
    // trailing matrix from row / column r0 on -= W[:, wofs:wofs+kw] * L[:, c0:c0+kw]^T, lower tiles only
    auto trailing = [&](int c0, int r0, int kw, int wofs = 0, hipStream_t ts = nullptr) {
        if (!ts)
            ts = m->stream;
        GemmArgs s;
        s.A = Wpp(r0, wofs), s.lda = WIDE_PANEL;
        s.B = Kp(r0, c0), s.ldb = ldk;
        s.C = Kp(r0, r0), s.ldc = ldk;
        s.M = np - r0, s.N = np - r0, s.K = kw;
        s.alpha = -1.0, s.beta = 1;
        s.lower_only = 1;
        {
            const double mt = (double)(s.M / TILE);
            m->factor_gemm_flops += mt * (mt + 1) * 0.5 * 2.0 * TILE * TILE * kw;
        }
        hipEvent_t *ev = gemm_events(m, gemm_idx);
        if (ev)
            (void)hipEventRecord(ev[0], ts);
        launch_gemm(m->prec, s, ts);
        if (ev) {
            (void)hipEventRecord(ev[1], ts);
            ++gemm_idx;
        }
    };
    // ---- look-ahead: panel p+1 is factorised on a second stream while the trailing update of panel p runs ----
    // Per 256-panel the serial chain (2 diagonal blocks, 2 panel solves, 1 half update, the 256-column strip of the update
    // the next panel lives in: ~150 us on an idle GPU) does not shrink with the remaining rows, the update does
    // (600 us at panel 2, 125 us at 56 tile rows): the second half of the panels is chain-bound.  The update is split
    // into that strip and the rest; the two halves of the 512-wide workspace alternate; bit-identical to the plain order
    // with 256-wide panels.  What overlap can buy is bounded by one effect (scripts/diag_beside.hip):
    // a kernel that shares a CU with a workgroup of the update runs 2.3x slower (diagonal block 28 -> 60-70 us, whatever
    // its priority), and one that does not fit beside it waits for the update to drain -- hence the 4-wave variant of the
    // diagonal-block kernel for launches that run beside an update.  Setting CUs aside for the chain was tried three ways:
    // hipExtStreamCreateWithCUMask (round 1, twice: 47 ms -- kernels on masked streams pay far more than the hop),
    // stream priority / s_setprio (nothing), and a persistent update that leaves one CU per shader engine empty
    // (round 2: the chain then runs at its stand-alone speed, the update 25-30 % slower, the LDL^T the same:
    // profiles/r02_ldlt_reserved_cus.txt).
    // Default window, from measurements (LDL^T ms, look-ahead on / off): fp32, with the round-2 diagonal-block kernel and
    // schedule, N = 8192 5.35 / 5.50, 12288 10.0 / 11.3, 16384 18.3 / 20.6, 32768 111 / 119 -> from 8192 rows on (round 1:
    // 6.9 / 6.5, -, 21.3 / 22.8, 117 / 125 -> from 16384); fp64 N = 4096 4.15 / 3.95, 8192 9.7 / 10.5, 16384 36.9 / 40.0,
    // 32768 239 / 232 (there the 512-wide panels of the plain order win) -> 8192 <= rows < 32768; with the round-2 kernel
    // N = 4096 3.62 / 3.44, 6144 5.7 / 5.95, 8192 8.4 / 9.5, 16384 35.2 / 37.9 -> 6144 <= rows < 32768.
    // It works on 256-wide panels.  (The GPX_LOOKAHEAD / GPX_PANEL switches of rounds 1-5 are gone; bit-identical to the plain
    // order at the same panel width: DESIGN_LEDGER.md.)
    const bool la_window = m->prec == GPX_PREC_F64 ? (np >= 24 * PANEL && np < 128 * PANEL) : np >= 32 * PANEL;
    const bool la_env = la_window;
    if (la_env && c_start == 0 && !m->stream2 &&
        stream_acquire(m->device, &m->stream2) != hipSuccess) {
        (void)hipGetLastError();
        m->stream2 = nullptr;
    }
    if (la_env && c_start == 0 && m->stream2) {
        hipStream_t sa = m->stream, sb = m->stream2;
        constexpr int la_tail_rows = 4096;  // remaining rows from which on one trailing update on the chain stream is the faster order
        size_t ev_used = 0;
        auto next_event = [&]() -> hipEvent_t {
            if (ev_used == m->la_ev.size()) {
                hipEvent_t ev;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
                    return nullptr;
                m->la_ev.push_back(ev);
            }
            return m->la_ev[ev_used++];
        };
        auto chain = [&](int c0, int wofs, hipStream_t st) {
            const int nb = std::min(PANEL, np - c0) / TILE;
            for (int h = 0; h < nb; ++h) {
                const int cc = c0 + h * TILE;
                block_step(cc, wofs + h * TILE, st);
                if (h + 1 < nb)
                    half_update(cc, wofs + h * TILE, c0 + nb * TILE, st);
            }
        };
        {
            // The chain stream owns everything the NEXT panel waits for -- diagonal blocks, panel solves, half update AND the
            // 256-column strip of the trailing update -- so no cross-stream hop lies on the serial chain any more (the
            // round-1 schedule, removed in round 4, had strip + two hops = 44 of the ~190 us per panel there, rocprofv3 timeline in
            // profiles/r02_create_stages.txt); the main stream only runs the rest of each trailing update:
            //   chain stream:  chain_p -> [eP_p] -> wait eR_{p-1} -> strip_p -> chain_{p+1} ...
            //   main stream :  wait eP_p -> rest_p -> [eR_p]
            // rest_{p-1} must precede strip_p (same tiles: the columns of panel p+1) and chain_{p+1} (it reads the workspace
            // half chain_{p+1} writes); both hold because strip_p waits for eR_{p-1}.  rest_p and strip_p touch disjoint
            // tiles.  Same launches on the same operands as the plain order: bit-identical.
            // all events up front (2 per panel + 2), so that the loop cannot run out half way
            const size_t need_ev = 2 * (size_t)(np / PANEL) + 4;
            while (m->la_ev.size() < need_ev) {
                hipEvent_t ev;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                m->la_ev.push_back(ev);
            }
            hipEvent_t e0 = m->la_ev.size() >= need_ev ? next_event() : nullptr;
            if (e0) {
                narrow_explicit = true;
                narrow_diag = true;
                (void)hipEventRecord(e0, sa);
                (void)hipStreamWaitEvent(sb, e0, 0);
                chain(0, 0, sb);
                hipEvent_t eR_prev = nullptr;
                int p = 0;
                for (int c0 = 0; c0 + PANEL < np; c0 += PANEL, ++p) {
                    const int wofs = (p & 1) * PANEL, r0 = c0 + PANEL;
                    const int sw = std::min(PANEL, np - r0);
                    const bool tail = np - r0 <= la_tail_rows || r0 + sw >= np;
                    // the rest update is released as soon as chain_p is done -- BEFORE the chain stream waits for rest_{p-1}:
                    // in the update-bound part the chain finishes long before that, and rest_p then follows rest_{p-1} on the
                    // main stream without a hop (was: hop + start of the strip = ~37 us per panel between two rest updates)
                    hipEvent_t eP = nullptr, eR = nullptr;
                    if (!tail) {
                        eP = next_event(), eR = next_event();  // (created above)
                        (void)hipEventRecord(eP, sb);
                    }
                    if (eR_prev)
                        (void)hipStreamWaitEvent(sb, eR_prev, 0);
                    if (tail) {  // one trailing update on the chain stream
                        eR_prev = nullptr;
                        narrow_diag = false;  // nothing runs beside the chain any more: the 8-wave diagonal kernel
                        trailing(c0, r0, PANEL, wofs, sb);
                        chain(r0, wofs ^ PANEL, sb);
                        continue;
                    }
                    GemmArgs s;  // strip: C[r0:, r0:r0+sw] -= W_p L_p^T
                    s.A = Wpp(r0, wofs), s.lda = WIDE_PANEL;
                    s.B = Kp(r0, c0), s.ldb = ldk;
                    s.C = Kp(r0, r0), s.ldc = ldk;
                    s.M = np - r0, s.N = sw, s.K = PANEL;
                    s.alpha = -1.0, s.beta = 1;
                    launch_gemm(m->prec, s, sb);
                    (void)hipStreamWaitEvent(sa, eP, 0);
                    trailing(c0, r0 + sw, PANEL, wofs, sa);
                    (void)hipEventRecord(eR, sa);
                    eR_prev = eR;
                    narrow_diag = true;
                    chain(r0, wofs ^ PANEL, sb);
                }
                narrow_explicit = narrow_diag = false;
                hipEvent_t eJ = next_event();
                (void)hipEventRecord(eJ, sb);
                (void)hipStreamWaitEvent(sa, eJ, 0);
                m->gemm_ev_used_factor = gemm_idx;
                return;
            }
        }
        // (no events: the plain order below)
    }
    int c0 = c_start;
    if (c0 % PANEL) {  // start in the middle of a 256-column unit: a lone 128-wide step
        block_step(c0, 0, m->stream);
        if (c0 + TILE < np)
            trailing(c0, c0 + TILE, TILE);
        c0 += TILE;
    }
    // fp64 has no accuracy to lose to the longer accumulations: 512-wide panels there (LDL^T at N = 16384: 43.1 -> 40.0 ms)
    const int wide = m->prec == GPX_PREC_F64 ? WIDE_PANEL : PANEL;
    while (c0 < np) {
        const int pw = std::min(wide, np - c0), nb = pw / TILE;
        for (int h = 0; h < nb; ++h) {
            const int cc = c0 + h * TILE, r0 = cc + TILE;
            block_step(cc, h * TILE, m->stream);
            if (h + 1 < nb && r0 < np)
                half_update(cc, h * TILE, c0 + pw, m->stream);
        }
        if (c0 + pw < np)
            trailing(c0, c0 + pw, pw);
        c0 += pw;
    }
    m->gemm_ev_used_factor = gemm_idx;
}

// the factorisation and the substitution on a matrix somebody else has put into m->Kmat (gpx_dgp.hip: the 4N x 4N matrix of
// a GP with derivative observations); everything they touch is allocated by alloc_model + the three buffers below
int alloc_factor_buffers(gpx_model *m)
{
    const size_t np = (size_t)m->npad, e = m->esz;
    HIPCHK(big_alloc(&m->Kmat, e * np * np));
    HIPCHK(big_alloc(&m->linv, e * (size_t)m->nblk * TILE * TILE));
    HIPCHK(big_alloc(&m->Wp, e * np * WIDE_PANEL));
    return GPX_OK;
}
void factorize_matrix(gpx_model *m)
{
    factor_init(m->prec);
    factorize(m);
}
void solve_factored(gpx_model *m, void *b, void *ytmp, void *x) { solve_ldl(m, b, ytmp, x, false); }
static void factor_append_rows(gpx_model *m, int t0);
// the same on a matrix whose leading t0 rows / columns (a multiple of 128) already hold a factor -- L, D, 1/D and the
// inverse diagonal blocks -- and whose rows behind them have just been filled: the rank-n append of gpx_model_update
void factorize_matrix_append(gpx_model *m, int t0)
{
    factor_init(m->prec);
    factor_append_rows(m, t0);
    factorize(m, t0);
}

// Rank-n update, rows [t0, npad): the kernel rows have just been built; the columns [0, t0) hold the old factor.
// Column block by column block: W = A_rows,j Linv_j^T (to the workspace), L_rows,j = W D_j^-1 (in place), then
// every later column of these rows -= W L_{later rows, j}^T.  2 launches per old column block.
static void factor_append_rows(gpx_model *m, int t0)
{
    const int np = m->npad;
    const size_t e = m->esz;
    char *K = (char *)m->Kmat;
    auto Kp = [&](size_t r, size_t c) { return (void *)(K + (r * np + c) * e); };
    void *Wrows = (char *)m->Wp + ((size_t)t0 * WIDE_PANEL) * e;
    for (int cc = 0; cc < t0; cc += TILE) {
        GemmArgs t;
        t.A = Kp(t0, cc), t.lda = np;
        t.B = (char *)m->linv + (size_t)(cc / TILE) * TILE * TILE * e, t.ldb = TILE;
        t.C = Kp(t0, cc), t.ldc = np;
        t.M = np - t0, t.N = TILE, t.K = TILE;
        t.b_lower = 1;
        t.epi = EPI_TRSM;
        t.W = Wrows, t.ldw = WIDE_PANEL;
        t.colscale = (char *)m->t_dinv + (size_t)cc * e;
        launch_gemm(m->prec, t, m->stream);
        GemmArgs s;  // columns [cc + 128, np) of the new rows
        s.A = Wrows, s.lda = WIDE_PANEL;
        s.B = Kp(cc + TILE, cc), s.ldb = np;
        s.C = Kp(t0, cc + TILE), s.ldc = np;
        s.M = np - t0, s.N = np - (cc + TILE), s.K = TILE;
        s.alpha = -1.0, s.beta = 1;
        launch_gemm(m->prec, s, m->stream);
    }
}

// F32_SPLIT: the fp32 inverse factor becomes packed fp16 hi/lo halves in place, the 1/D slot the scaled weights and
// the row-correction vectors move to the accumulators' units.  No-op for the other modes, while the model still
// trains in fp64 (the packing runs on the demoted fp32 state) and when already packed.
// does an F32_SPLIT model of this size pack its inverse factor (a function of n and the options only: a shell committed from
// the state blobs of such a model must agree -- gpx_model_commit)
bool split_packs(const gpx_model *m)
{
    return !(m->n <= SPLIT_MIN_N && m->var_fit_opt && var_cols_fits(m->n, m->npad, m->npad, m->npad + KQP_LDPAD));
}

// power-of-two scale of the kernel values in the split contraction: k(0) * sk in [0.5, 1).  Every F32_SPLIT model carries it,
// packed or not: the small-model split kernel (gpx_varcols16.hip) splits (k - fit) * sk per call, and with the default of 1 an
// amplitude of 1e5 overflowed the fp16 halves and one of 1e-6 fell into their subnormals.
void set_split_scale(gpx_model *m)
{
    int e2 = 0;
    (void)std::frexp(m->cov.k0 > 0 ? m->cov.k0 : 1.0, &e2);
    m->sk = (float)std::ldexp(1.0, -e2);
}

static int pack_split(gpx_model *m)
{
    if (m->opt.precision == GPX_PREC_F32_SPLIT)
        set_split_scale(m);
    if (m->opt.precision != GPX_PREC_F32_SPLIT || m->x_packed || m->prec != GPX_PREC_F32 || !m->X)
        return GPX_OK;
    // Small models stay on the fp32 contraction of the small-model kernel (gpx_varcols_kernel.hpp), which needs no operand in
    // memory: measured on the C5 objects (scripts/c5_breakdown.py split, variance stage of 2^21 queries, packed operands vs
    // that kernel): N = 277 3.82 vs 2.24 ms, 354 3.82 vs 3.23, 447 4.95 vs 4.47, 481 4.99 vs 5.29, 724 9.18 vs 10.59 -- the
    // split contraction pays from ~470 points on.  Same accuracy class (the mode promises at least the F32 one).
    if (!split_packs(m))
        return GPX_OK;
    const int np = m->npad;
    if (!m->hD.size()) {  // keep D readable (GPX_FIELD_D) -- the 1/D slot is about to hold the weights
        std::vector<float> t((size_t)m->n);
        HIPCHK(hipMemcpy(t.data(), m->t_d, sizeof(float) * (size_t)m->n, hipMemcpyDeviceToHost));
        m->hD.assign(t.begin(), t.end());
    }
    launch_split_prepare((float *)m->X, np, (float *)m->t_dinv, m->sk, (unsigned *)(m->d_info + 4), m->stream,
                         m->d_meta + 3);
    HIPCHK(hipStreamSynchronize(m->stream));
    m->x_packed = true;
    return GPX_OK;
}

// 1/D in fp64 for the fp64 epilogue of the variance contraction (state part 0; the T copy serves the solves)
static void store_dinv64(gpx_model *m, hipStream_t s)
{
    if (m->prec == GPX_PREC_F64)
        (void)hipMemcpyAsync(m->d_dinv64, m->t_dinv, sizeof(double) * (size_t)m->npad, hipMemcpyDeviceToDevice, s);
    else
        launch_cast_f2d((size_t)m->npad, (const float *)m->t_dinv, m->d_dinv64, s);
}

int build_inverse(gpx_model *m)
{
    if (m->has_inverse)
        return GPX_OK;
    // a shell committed without its inverse factor (gpx_model_commit(m, 0)) or a replica holds no LDL^T to build one from
    if (!m->Kmat || !m->linv)
        return fail(GPX_E_STATE, "model holds no factor to build the inverse factor from (mean-only shell)");
    const int np = m->npad;
    const size_t e = m->esz;
    if (!m->X)
        HIPCHK(big_alloc(&m->X, e * (size_t)np * np));
    (void)hipEventRecord(m->ev[EV_INV0], m->stream);
    // temporaries behind guards: every early return below (HIPCHK) releases them
    DevGuard gTws(nullptr, true), gL64(nullptr, true), gX64(nullptr, true), glinv64(nullptr, true);
    bool assemble64 = m->prec == GPX_PREC_F32 && m->inv64;
    if (assemble64) {
        // three N x N fp64 temporaries: at very large N they may not fit next to K and X -- assemble in fp32 then
        const size_t nn = (size_t)np * np;
        if (big_alloc(&gL64.p, sizeof(double) * nn) != hipSuccess || big_alloc(&gX64.p, sizeof(double) * nn) != hipSuccess ||
            big_alloc(&gTws.p, sizeof(double) * nn) != hipSuccess ||
            big_alloc(&glinv64.p, sizeof(double) * (size_t)m->nblk * TILE * TILE) != hipSuccess) {
            (void)hipGetLastError();
            gL64.reset(), gX64.reset(), gTws.reset(), glinv64.reset();
            assemble64 = false;
        }
    }
    if (assemble64) {
        double *L64 = (double *)gL64.p, *X64 = (double *)gX64.p, *linv64 = (double *)glinv64.p;
        char *Tws = (char *)gTws.p;
        // The fp32 factor is kept (that is what runs on the fp32 MFMA), but its inverse is assembled in fp64 and
        // rounded once.  Measured at N = 16384 (variance error / k(0) vs the fp64 pipeline): Matern-5/2 1.0e-5 ->
        // 4.5e-6, Gaussian 1.1e-5 -> 2.3e-6, thin-plate R=4 1.05e-4 -> 2.1e-5, i.e. the level of an fp64 factor:
        // the log2(N/128) levels of products of inverses, not the LDL^T, are where fp32 loses the accuracy.
        launch_cast_lower_f2d(np, (const float *)m->Kmat, L64, false, m->stream, 0, -1);  // the upper tiles are never read
        launch_cast_f2d((size_t)m->nblk * TILE * TILE, (const float *)m->linv, linv64, m->stream);
        // (no memset of X64: the assembly reads and writes only tiles on / below the block diagonal -- place_diag
        // supplies the diagonal tiles, zeros above the diagonal inside them -- and the final cast writes the zeros of
        // the upper tiles of X without reading them)
        launch_place_diag(GPX_PREC_F64, m->nblk, linv64, X64, np, m->stream);
        trtri_levels(GPX_PREC_F64, 8, (char *)L64, (char *)X64, Tws, np, m->stream, np);
        launch_cast_lower_d2f(np, X64, (float *)m->X, true, m->stream);
        if (m->var_fit)  // from the un-rounded rows: the rounding of X then only meets the small fit residual
            launch_var_rowcorr(true, m->op64, m->n, np, X64, np, m->d_x, m->d_y, m->d_z, m->d_meta, m->d_corr, m->stream);
    } else {
        if (!gTws.p)
            HIPCHK(big_alloc(&gTws.p, e * (size_t)np * np));
        // blocks above the diagonal are structural zeros (the LDS-staged variance tiles read them)
        HIPCHK(hipMemsetAsync(m->X, 0, e * (size_t)np * np, m->stream));
        launch_place_diag(m->prec, m->nblk, m->linv, m->X, np, m->stream);
        trtri_levels(m->prec, e, (char *)m->Kmat, (char *)m->X, (char *)gTws.p, np, m->stream, np);
        if (m->var_fit)
            launch_var_rowcorr(m->prec == GPX_PREC_F64, m->op64, m->n, np, m->X, np, m->d_x, m->d_y, m->d_z, m->d_meta,
                               m->d_corr, m->stream);
    }
    store_dinv64(m, m->stream);
    {
        hipError_t le = hipGetLastError();  // launches are not checked one by one
        if (le != hipSuccess) {
            (void)hipStreamSynchronize(m->stream);  // the guards release the temporaries: nothing may still use them
            return fail(GPX_E_HIP, std::string("inverse factor: kernel launch: ") + hipGetErrorString(le));
        }
    }
    (void)hipEventRecord(m->ev[EV_INV1], m->stream);
    {
        const hipError_t se = hipStreamSynchronize(m->stream);
        if (se != hipSuccess) {
            (void)hipDeviceSynchronize();  // before the guards hand the temporaries back
            return fail(GPX_E_HIP, std::string("inverse factor: ") + hipGetErrorString(se));
        }
    }
    gTws.reset(), gL64.reset(), gX64.reset(), glinv64.reset();
    float ms = 0;
    if (hipEventElapsedTime(&ms, m->ev[EV_INV0], m->ev[EV_INV1]) == hipSuccess)
        m->stats.t_inverse_ms = ms;
    m->has_inverse = true;
    return pack_split(m);
}

// ---- rank-n update: the inverse factor grows with the factor --------------------------------------------------
// X = L^-1 of the grown factor: the leading t0 x t0 part is the old X, the new rows are
//     X22 = (L22)^-1 (recursive doubling on the small trailing block),   X21 = -X22 (L21 X11),
// two products with one thin operand (N = 16128 + 256: 2 ms) instead of the 26 ms of a fresh inverse at the next
// variance query.  Runs in the working precision (the fresh fp32 path assembles in fp64: the appended rows carry
// fp32 product rounding instead; the update tests hold the variance to the same tolerance as a rebuild).  Any
// allocation failure just leaves the inverse to be rebuilt lazily.
static void append_inverse(gpx_model *m, kept_factor *keep)
{
    const int np = m->npad, t0 = keep->t0;
    const size_t e = m->esz;
    const int np_rows = std::min(np, (m->n + TILE - 1) / TILE * TILE);
    const int m2 = np_rows - t0;
    if (m2 <= 0 || t0 <= 0 || !keep->X)
        return;
    hipStream_t s = m->stream;
    void *Tw = nullptr;
    if (big_alloc(&m->X, e * (size_t)np * np) != hipSuccess || hipMalloc(&Tw, e * (size_t)m2 * np) != hipSuccess) {
        (void)hipGetLastError();
        big_free(m->X);
        m->X = nullptr;
        return;
    }
    (void)hipEventRecord(m->ev[EV_INV0], s);
    bool ok = hipMemsetAsync(m->X, 0, e * (size_t)np * np, s) == hipSuccess;
    ok = ok && hipMemcpy2DAsync(m->X, e * np, keep->X, e * keep->np_old, e * t0, t0, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (ok) {
        launch_place_diag(m->prec, m->nblk, m->linv, m->X, np, s);
        auto at = [&](void *base, size_t r, size_t c) { return (char *)base + (r * np + c) * e; };
        trtri_levels(m->prec, e, at(m->Kmat, t0, t0), at(m->X, t0, t0), (char *)Tw, m2, s, np);
        GemmArgs g1;  // T = L21 * X11 (B lower-triangular, [k][n])
        g1.A = at(m->Kmat, t0, 0), g1.lda = np;
        g1.B = m->X, g1.ldb = np;
        g1.C = Tw, g1.ldc = np;
        g1.M = m2, g1.N = t0, g1.K = t0;
        g1.nn = 1, g1.b_lower = 1;
        launch_gemm(m->prec, g1, s);
        GemmArgs g2;  // X21 = -X22 * T (A lower-triangular)
        g2.A = at(m->X, t0, t0), g2.lda = np;
        g2.B = Tw, g2.ldb = np;
        g2.C = at(m->X, t0, 0), g2.ldc = np;
        g2.M = m2, g2.N = t0, g2.K = m2;
        g2.nn = 1, g2.a_lower = 1;
        g2.alpha = -1.0;
        launch_gemm(m->prec, g2, s);
        if (m->var_fit)  // the cloud (hence its centre) and the rows of X changed: all rows again (N^2/2 reads)
            launch_var_rowcorr(m->prec == GPX_PREC_F64, m->op64, m->n, np, m->X, np, m->d_x, m->d_y, m->d_z, m->d_meta,
                               m->d_corr, s);
        store_dinv64(m, s);
    }
    (void)hipEventRecord(m->ev[EV_INV1], s);
    ok = ok && hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
    (void)hipFree(Tw);
    if (!ok) {
        (void)hipGetLastError();
        big_free(m->X);
        m->X = nullptr;
        return;
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, m->ev[EV_INV0], m->ev[EV_INV1]) == hipSuccess)
        m->stats.t_inverse_ms = ms;
    m->has_inverse = true;
}

// ---- MIXED precision: round the fp64 state once to fp32 and release the fp64 factor -----------------
static int demote_to_f32(gpx_model *m)
{
    const size_t np = (size_t)m->npad;
    hipStream_t s = m->stream;
    m->hD.resize((size_t)m->n);
    HIPCHK(hipMemcpy(m->hD.data(), m->t_d, sizeof(double) * (size_t)m->n, hipMemcpyDeviceToHost));
    void *nb = nullptr, *nX = nullptr;
    size_t nbytes = 0;
    int rc = alloc_blob0(m, 4, &nb, &nbytes);
    if (rc)
        return rc;
    DevGuard gnb(nb, true), gnX(nullptr, true);  // released if a call below fails
    HIPCHK(big_alloc(&gnX.p, sizeof(float) * np * np));
    nX = gnX.p;
    // the fp64 part (points, alpha, 1/D, row-correction vectors) and the meta block as they are; the T part rounded
    HIPCHK(hipMemcpyAsync(nb, m->blob0, sizeof(double) * np * (5 + VAR_NCORR), hipMemcpyDeviceToDevice, s));
    float *tf = (float *)((char *)nb + sizeof(double) * np * (5 + VAR_NCORR));
    launch_cast_d2f(np * 4, (const double *)m->t_x, tf, s);  // x' y' z' 1/D are contiguous
    HIPCHK(hipMemcpyAsync(tf + 4 * np, m->d_meta, sizeof(double) * BLOB_META, hipMemcpyDeviceToDevice, s));
    launch_cast_lower_d2f((int)np, (const double *)m->X, (float *)nX, true, s);
    {
        const hipError_t se = hipStreamSynchronize(s);
        if (se != hipSuccess) {
            (void)hipDeviceSynchronize();
            return fail(GPX_E_HIP, std::string("demotion to fp32: ") + hipGetErrorString(se));
        }
    }
    (void)gnb.release(), (void)gnX.release();  // the model owns them from here on
    big_free(m->blob0);
    big_free(m->X);
    big_free(m->Kmat);
    big_free(m->linv);
    big_free(m->Wp);
    big_free(m->tvecs);
    m->Kmat = m->linv = m->Wp = m->tvecs = nullptr;
    m->t_s2 = m->t_d = m->t_b = m->t_yv = m->t_xs = m->t_alpha = nullptr;
    m->blob0 = nb;
    m->blob0_bytes = nbytes;
    m->X = nX;
    m->prec = GPX_PREC_F32;
    m->esz = 4;
    carve_blob0(m);
    return GPX_OK;
}

// ---- create of a small model in three launches (gpx_small.hip) ---------------------------------------------------
// Taken for a fresh create (no rank-n append) of a model that trains in fp64 and has at most SMALL_CREATE_MAX_NP padded
// rows -- every model of the reference's own sizes in every precision mode (F32 / F32_SPLIT models of this size train in
// fp64, set_training_precision).  GPX_DATAFLOW=0 keeps the general chain (its tested twin).
// Models above the small-model path (fresh creates in either working precision, up to 16384 padded rows): kernel matrix +
// LDL^T as one dataflow launch -- 64 x 64 tiles below 8192 rows, where the chain of diagonal tiles sets the time
// (gpx_dataflow.hpp), 128 x 128 tiles from there on, where the 64 x 64 form is HBM-bound (gpx_dataflow_wide.hpp).  It beats
// kbuild + the blocked launch chain at every size on the same box (profiles/r05_ldlt_sweep.txt: N = 16384 fp32 18.7 -> 15.6 ms,
// fp64 35.5 -> 30.4 ms; N = 4096 fp64 3.43 -> 1.75 ms).  Larger models and rank-n appends keep the chain.  GPX_DATAFLOW=0
// keeps the launch chain; GPX_DATAFLOW=64 | 128 forces that tile form at every size up to 32768 rows (tests, sweeps).
static bool mid_factor_eligible(const gpx_model *m)
{
    const int df = gpxh::switches().dataflow;  // GPX_DATAFLOW: 0 = launch chain, 64 | 128 = that tile form at EVERY size
    if (df == 0)
        return false;
    if (m->npad <= SMALL_CREATE_MAX_NP)
        return false;
    if (df == 64 || df == 128)
        return m->npad <= MID_FACTOR_FORCED_MAX_NP;  // (the flags and per-tile results scale; tested at 20480 rows)
    return m->npad <= (m->prec == GPX_PREC_F64 ? MID_FACTOR_MAX_NP_F64 : MID_FACTOR_MAX_NP_F32);
}

// The three-launch create needs its WHOLE grid resident (the inverse-factor jobs of the first launch wait for workgroups with
// a higher index, the second launch has grid barriers): ntiles <= 136 workgroups at one per CU (115 KB of LDS each) and up to
// 128 workgroups of the alpha kernel.  A device with fewer CUs (a partition, a CU mask) takes the launch chain.
static bool small_create_eligible(const gpx_model *m, const kept_factor *keep)
{
    if (keep || m->prec != GPX_PREC_F64 || m->npad > SMALL_CREATE_MAX_NP)
        return false;
    if (gpxh::switches().dataflow == 0)
        return false;
    const int nbt = m->npad / SMALL_TILE;
    return device_cu_count(m->device) >= std::max(nbt * (nbt + 1) / 2, SMALL_ALPHA_MAX_GRID / 2);  // (alpha: two workgroups fit a CU)
}

// *fell_back = true: a wait inside the launches gave up (the GPU was too busy to hold the whole grid); nothing of the
// model is valid and the caller runs the general chain.  Otherwise the model is complete -- points, factor, inverse
// factor, alpha, row corrections, and for the fp32 modes the rounded state -- exactly as build_model leaves it.
static int build_model_small(gpx_model *m, bool *fell_back)
{
    *fell_back = false;
    const int n = m->n, np = m->npad;
    small_create_init();
    std::vector<double> diag(n);
    for (int i = 0; i < n; ++i)
        diag[i] = m->cov.k0 + (m->has_s2 ? m->hs2[i] : 0.0);
    eigen_pivot_order(diag, m->perm);
    if (!m->dvecs) {
        int rc = alloc_model(m);
        if (rc)
            return rc;
        HIPCHK(big_alloc(&m->Kmat, sizeof(double) * (size_t)np * np));
        HIPCHK(big_alloc(&m->linv, sizeof(double) * (size_t)m->nblk * TILE * TILE));
        HIPCHK(big_alloc(&m->Wp, sizeof(double) * (size_t)np * WIDE_PANEL));
    }
    if (!m->X)
        HIPCHK(big_alloc(&m->X, sizeof(double) * (size_t)np * np));
    const SmallWs lay = small_ws_layout(np);
    DevGuard gws(nullptr, true), gnb(nullptr, true), gnX(nullptr, true);
    HIPCHK(big_alloc(&gws.p, lay.bytes));
    char *ws = (char *)gws.p;
    size_t nblob_bytes = 0;
    if (m->train64) {  // the fp32 state the model ends up in (demote_to_f32), filled by the third launch
        int rc = alloc_blob0(m, 4, &gnb.p, &nblob_bytes);
        if (rc)
            return rc;
        HIPCHK(big_alloc(&gnX.p, sizeof(float) * (size_t)np * np));
    }
    // host block: [5][np] staging + argument block (as in the workspace) | SmallResult | D[np]
    const size_t res_bytes = lay.res_d + sizeof(double) * (size_t)np - lay.res;
    const size_t host_bytes = lay.stage_bytes + res_bytes;
    void *hp = nullptr;
    HIPCHK(pinned_acquire(host_bytes, &hp));
    struct PinGuard {
        void *p;
        ~PinGuard() { pinned_release(p); }
    } pin_guard{hp};
    double *st = (double *)hp;
    std::memset(st, 0, sizeof(double) * 5 * (size_t)np);
    double c[3] = {0, 0, 0};
    for (int k = 0; k < n; ++k) {
        const int i = m->perm[k];
        st[k] = m->hx[i];
        st[np + k] = m->hy[i];
        st[2 * (size_t)np + k] = m->hz[i];
        st[3 * (size_t)np + k] = m->hlabel[i];
        st[4 * (size_t)np + k] = m->has_s2 ? m->hs2[i] : 0.0;
        c[0] += st[k], c[1] += st[np + k], c[2] += st[2 * (size_t)np + k];
    }
    for (int d = 0; d < 3; ++d)
        m->cen[d] = c[d] / n;
    double ymax = 0.0;
    for (int i = 0; i < n; ++i)
        ymax = std::max(ymax, std::fabs(m->hlabel[i]));
    SmallArgs a;
    a.stage = (const double *)(ws + lay.stage);
    a.n = n, a.np = np, a.nbt = np / SMALL_TILE, a.nb = (n + SMALL_TILE - 1) / SMALL_TILE;
    a.ntiles = a.nbt * (a.nbt + 1) / 2;
    a.cov = lower_cov<double>(m->cov);
    for (int d = 0; d < 3; ++d)
        a.cen[d] = m->cen[d];
    a.want_corr = m->var_fit_opt ? 1 : 0;
    a.op64 = m->op64 ? 1 : 0;
    a.ir_adaptive = m->opt.ir_steps < 0 ? 1 : 0;
    a.ir_max = a.ir_adaptive ? 4 : m->opt.ir_steps;
    a.ir_tol = 1e-9 * std::max(ymax, 1e-300);
    a.epoch = small_create_epoch();
    a.wait_ticks = wait_budget_ticks(np);
    a.abort_idx = 2 * a.ntiles, a.bar_idx = 2 * a.ntiles + 1, a.pre_idx = 2 * a.ntiles + 2;
    a.dbg = (unsigned long long *)(ws + lay.dbg);
    a.K = (double *)m->Kmat, a.X = (double *)m->X, a.linv = (double *)m->linv;
    a.d = (double *)m->t_d, a.dinv = (double *)m->t_dinv;
    a.d_x = m->d_x, a.d_y = m->d_y, a.d_z = m->d_z;
    a.t_x = (double *)m->t_x, a.t_y = (double *)m->t_y, a.t_z = (double *)m->t_z;
    a.d_lab = m->d_lab, a.d_s2 = m->d_s2, a.t_s2 = (double *)m->t_s2;
    a.d_alpha = m->d_alpha, a.t_alpha = (double *)m->t_alpha, a.d_r = m->d_r;
    a.d_corr = m->d_corr, a.d_dinv64 = m->d_dinv64, a.d_meta = m->d_meta, a.info = m->d_info;
    a.blob0 = (const double *)m->blob0, a.nblob = gnb.p, a.nX = (float *)gnX.p;
    a.XT = (double *)(ws + lay.xt);
    a.flags = (unsigned long long *)(ws + lay.flags);
    a.tmax = (double *)(ws + lay.tmax), a.rmaxv = (double *)(ws + lay.rmaxv), a.u = (double *)(ws + lay.u);
    a.tij = (int *)(ws + lay.tij), a.negcnt = (int *)(ws + lay.negcnt), a.badrow = (int *)(ws + lay.badrow);
    a.res = (SmallResult *)(ws + lay.res), a.res_d = (double *)(ws + lay.res_d);
    hipStream_t s = m->stream;
    std::memcpy((char *)hp + (lay.args - lay.stage), &a, sizeof(a));
    HIPCHK(hipMemcpyAsync(ws + lay.stage, st, lay.stage_bytes, hipMemcpyHostToDevice, s));
#ifdef SM_TIMING
    (void)hipMemsetAsync(ws + lay.dbg, 0, sizeof(unsigned long long) * (size_t)a.ntiles * SMALL_DBG_STAMPS, s);
#endif
    // Two of these grids on one device at the same time can starve each other -- each holds CUs with workgroups that wait for
    // workgroups of their own grid the other one keeps from being dispatched -- until a wait's budget is spent and both fall
    // back.  Small creates on one device therefore take turns, from the first launch to the result (0.2 - 0.5 ms each).
    static std::mutex full_mtx[MAX_DEVICES];
    std::unique_lock<std::mutex> full_lk(full_mtx[m->device >= 0 && m->device < MAX_DEVICES ? m->device : 0]);
    (void)hipEventRecord(m->ev[EV_T0], s);
    (void)hipEventRecord(m->ev[EV_KBUILD], s);
    launch_small_create(m->kern.id, a, (const SmallArgs *)(ws + lay.args), m->train64, s, m->ev[EV_FACTOR], m->ev[EV_SOLVE]);
    (void)hipEventRecord(m->ev[EV_NORMALS], s);
    char *hres = (char *)hp + lay.stage_bytes;
    HIPCHK(hipMemcpyAsync(hres, ws + lay.res, res_bytes, hipMemcpyDeviceToHost, s));
    {
        const hipError_t se = hipStreamSynchronize(s);
        full_lk.unlock();
        if (se != hipSuccess) {
            (void)hipDeviceSynchronize();
            return fail(GPX_E_HIP, std::string("small-model create: ") + hipGetErrorString(se));
        }
        HIPCHK(hipGetLastError());
    }
    const SmallResult *res = (const SmallResult *)hres;
    const double *hd = (const double *)(hres + (lay.res_d - lay.res));
#ifdef SM_TIMING
    if (const char *dump = std::getenv("GPX_SMALL_TIMING_DUMP")) {  // developer build: the stamps of every workgroup
        std::vector<unsigned long long> hs((size_t)a.ntiles * SMALL_DBG_STAMPS);
        (void)hipMemcpy(hs.data(), ws + lay.dbg, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (FILE *fp = std::fopen(dump, "w")) {
            for (int t = 0; t < a.ntiles; ++t) {
                std::fprintf(fp, "%d", t);
                for (int k = 0; k < 24; ++k)
                    std::fprintf(fp, " %llu", hs[(size_t)t * SMALL_DBG_STAMPS + k]);
                std::fprintf(fp, "\n");
            }
            std::fclose(fp);
        }
    }
#endif
    if (res->info[5] != 0) {
        *fell_back = true;
        return GPX_OK;  // (the guards hand the workspace and the unused fp32 state back)
    }
    float ms;
    m->stats = gpx_stats{};
    // kernel matrix, LDL^T and inverse factor are ONE launch here: its time is reported as the factorisation's
    if (hipEventElapsedTime(&ms, m->ev[EV_KBUILD], m->ev[EV_FACTOR]) == hipSuccess)
        m->stats.t_factor_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_FACTOR], m->ev[EV_SOLVE]) == hipSuccess)
        m->stats.t_solve_ms = ms;
    if (m->train64 && hipEventElapsedTime(&ms, m->ev[EV_SOLVE], m->ev[EV_NORMALS]) == hipSuccess)
        m->stats.t_inverse_ms = ms;  // (the rounding of the state)
    m->gemm_ev_used_factor = 0;
    m->stats.n = n;
    m->stats.n_padded = np;
    m->stats.n_negative_pivots = res->info[1];
    m->stats.ir_steps_done = res->ir_done;
    m->stats.alpha_residual = res->rmax;
    if (res->info[0] != 0)
        return fail(GPX_E_SINGULAR, "LDL^T: zero or non-finite pivot at internal row " + std::to_string(res->info[0] - 1));
    {  // Model::R (gp_regressor.hpp:135): the device found the arg-max pair, the distance is fp64 from the caller's points
        const int p = res->info[2], q = res->info[3];
        if (p >= 0 && p < n && q >= 0 && q < n) {
            const int ia = m->perm[p], ib = m->perm[q];
            const double dx = m->hx[ia] - m->hx[ib], dy = m->hy[ia] - m->hy[ib], dz = m->hz[ia] - m->hz[ib];
            m->R = std::sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    m->ready = true;
    m->has_inverse = true;
    m->var_fit = m->var_fit_opt;
    m->promoted = false;
    if (m->train64 && m->stats.n_negative_pivots > 0 && gpxh::switches().no_promote <= 0) {
        m->train64 = false;  // an indefinite kernel matrix keeps its fp64 state (see build_model)
        m->var_fit = false;
        m->promoted = true;
    } else if (m->train64 && m->stats.n_negative_pivots > 0) {
        // GPX_NO_PROMOTE (tests): the third launch skipped the rounding of an indefinite model -- do it now
        int rc = demote_to_f32(m);
        return rc ? rc : pack_split(m);
    }
    if (m->train64) {  // adopt the fp32 state the third launch has filled (what demote_to_f32 does for the chain)
        m->hD.assign(hd, hd + n);
        big_free(m->blob0);
        big_free(m->X);
        big_free(m->Kmat);
        big_free(m->linv);
        big_free(m->Wp);
        big_free(m->tvecs);
        m->Kmat = m->linv = m->Wp = m->tvecs = nullptr;
        m->t_s2 = m->t_d = m->t_b = m->t_yv = m->t_xs = m->t_alpha = nullptr;
        m->blob0 = gnb.release();
        m->blob0_bytes = nblob_bytes;
        m->X = gnX.release();
        m->prec = GPX_PREC_F32;
        m->esz = 4;
        carve_blob0(m);
    }
    if (m->opt.with_normals) {  // create<true> (gp_regressor.hpp:166-181): the gradient kernel at the training points
        size_t need = predict_ws_doubles(n, np, true) * sizeof(double);
        int rc = ensure(m, (void **)&m->ws_pred, &m->ws_pred_doubles, need);
        if (rc)
            return rc;
        if (!m->d_normals)
            HIPCHK(hipMalloc((void **)&m->d_normals, sizeof(double) * 3 * (size_t)n));
        launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, n, m->d_x, m->d_y, m->d_z, m->d_f,
                       m->d_normals, m->ws_pred, s);
        launch_normalize_rows3(n, m->d_normals, s);
        HIPCHK(hipStreamSynchronize(s));
        m->has_normals = true;
    }
    return m->prec == GPX_PREC_F32 ? pack_split(m) : GPX_OK;
}

// ---- create: everything after the host arrays are in place ---------------------------------------
int build_model(gpx_model *m, kept_factor *keep, bool no_dataflow)
{
    const int n = m->n, np = m->npad;
    const size_t e = m->esz;
    HIPCHK(hipSetDevice(m->device));
    factor_init(m->prec);
    int64_t small_fallbacks = 0;
    DevGuard mid_ws(nullptr, true);  // flags and per-tile results of the dataflow factorisation (released when this call returns)
    bool used_dataflow = false;
    // Eigen's pivot order from the original diagonal k(0) + sigma2_i
    std::vector<double> diag(n);
    for (int i = 0; i < n; ++i)
        diag[i] = m->cov.k0 + (m->has_s2 ? m->hs2[i] : 0.0);
    eigen_pivot_order(diag, m->perm);
    // host staging (internal order, zero padded)
    std::vector<double> st((size_t)np * 5, 0.0);
    for (int k = 0; k < n; ++k) {
        const int i = m->perm[k];
        st[k] = m->hx[i];
        st[np + k] = m->hy[i];
        st[2 * (size_t)np + k] = m->hz[i];
        st[3 * (size_t)np + k] = m->hlabel[i];
        st[4 * (size_t)np + k] = m->has_s2 ? m->hs2[i] : 0.0;
    }
    if (small_create_eligible(m, keep)) {
        bool fell_back = false;
        const int rc = build_model_small(m, &fell_back);
        if (!fell_back)
            return rc;
        small_fallbacks = 1;  // a wait of the dataflow launches gave up: the general chain below redoes the create
    }
    if (!m->dvecs) {
        int rc = alloc_model(m);
        if (rc)
            return rc;
        HIPCHK(big_alloc(&m->Kmat, e * (size_t)np * np));
        HIPCHK(big_alloc(&m->linv, e * (size_t)m->nblk * TILE * TILE));
        HIPCHK(big_alloc(&m->Wp, e * (size_t)np * WIDE_PANEL));
    }
    if (!m->d_tmax) {
        const int nt = np / TILE, ntiles = nt * (nt + 1) / 2;
        HIPCHK(hipMalloc((void **)&m->d_tmax, sizeof(float) * ntiles));
        HIPCHK(hipMalloc((void **)&m->d_tij, sizeof(int) * 2 * ntiles));
    }
    {  // workspace of the matrix-free residual / normals passes (n queries against npad points)
        size_t need = predict_ws_doubles(n, np, m->opt.with_normals != 0) * sizeof(double);
        if (need) {
            int rc = ensure(m, (void **)&m->ws_pred, &m->ws_pred_doubles, need);
            if (rc)
                return rc;
        }
    }
    hipStream_t s = m->stream;
    HIPCHK(hipMemcpyAsync(m->d_x, st.data(), sizeof(double) * (size_t)np * 3, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->d_lab, st.data() + 3 * (size_t)np, sizeof(double) * (size_t)np * 2,
                          hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(m->d_alpha, 0, sizeof(double) * (size_t)np, s));
    HIPCHK(hipMemsetAsync(m->d_r, 0, sizeof(double) * (size_t)np * 2 + 64, s));
    HIPCHK(hipMemsetAsync(m->d_info, 0, sizeof(int) * 8, s));
    // The working-precision copies of the points are relative to the centroid: every kernel depends on differences
    // only, so the shift is free, and fp32 arithmetic (kernel matrix, fp32 operand kernels) then rounds coordinates of
    // the size of the cloud, not of its distance from the origin (a cloud of radius 0.5 at offset 100 would lose 8
    // bits).  The fp64 points stay as given: the mean / gradient and the fp64 operand kernels use them.
    {
        double c[3] = {0, 0, 0};
        for (int k = 0; k < n; ++k)
            c[0] += st[k], c[1] += st[np + k], c[2] += st[2 * (size_t)np + k];
        for (int d = 0; d < 3; ++d)
            m->cen[d] = c[d] / n;
        double meta[BLOB_META] = {m->cen[0], m->cen[1], m->cen[2], 1.0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(m->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));  // meta lives on this stack frame
    }
    launch_cast_vec(m->prec, n, np, m->d_x, m->t_x, s, m->cen[0]);
    launch_cast_vec(m->prec, n, np, m->d_y, m->t_y, s, m->cen[1]);
    launch_cast_vec(m->prec, n, np, m->d_z, m->t_z, s, m->cen[2]);
    launch_cast_vec(m->prec, np, np, m->d_s2, m->t_s2, s);
    // ---- kernel matrix ----
    (void)hipEventRecord(m->ev[EV_T0], s);
    if (keep && keep->t0 > 0) {
        // rank-n update: the old factor goes back into the (possibly larger) matrix, only the new rows are built
        const size_t t0 = (size_t)keep->t0;
        HIPCHK(hipMemcpy2DAsync(m->Kmat, e * np, keep->K, e * keep->np_old, e * t0, t0, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->linv, keep->linv, e * (t0 / TILE) * TILE * TILE, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->t_d, keep->d, e * t0, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->t_dinv, keep->dinv, e * t0, hipMemcpyDeviceToDevice, s));
        launch_kbuild(m->prec, m->cov, n, np, m->t_x, m->t_y, m->t_z, m->t_s2, m->Kmat, m->d_tmax, m->d_tij, s,
                      keep->t0 / TILE);
        (void)hipEventRecord(m->ev[EV_KBUILD], s);
        factor_append_rows(m, keep->t0);
        factorize(m, keep->t0);
    } else if (mid_factor_eligible(m) && !no_dataflow) {
        // kernel matrix + LDL^T as one dataflow launch (gpx_dataflow.hpp): mid-size models, where the chain of the blocked
        // factorisation below -- 3-4 dependent launches per 128 columns -- and not its flops sets the time
        small_create_init();
        HIPCHK(big_alloc(&mid_ws.p, mid_ws_layout(np).bytes));
        MidFactorArgs a;
        a.n = n, a.np = np;
        a.K = m->Kmat, a.linv = m->linv, a.d = m->t_d, a.dinv = m->t_dinv;
        a.px = m->t_x, a.py = m->t_y, a.pz = m->t_z, a.ps2 = m->t_s2;
        a.ws = mid_ws.p, a.info = m->d_info, a.epoch = small_create_epoch();
        {
            const int df = gpxh::switches().dataflow;  // (64 | 128: the tests run either tile form at their own sizes)
            a.wide = df == 128 || (df != 64 && np >= WIDE_FACTOR_MIN_NP);  // 128 x 128 tiles from 8192 padded rows
        }
        a.wait_ticks = wait_budget_ticks(np);
        (void)hipEventRecord(m->ev[EV_KBUILD], s);
        launch_mid_factor(m->prec, m->cov, a, s);
        m->gemm_ev_used_factor = 0;
        m->factor_gemm_flops = 0;
        used_dataflow = true;
    } else {
        const int nmax = launch_kbuild(m->prec, m->cov, n, np, m->t_x, m->t_y, m->t_z, m->t_s2, m->Kmat, m->d_tmax, m->d_tij, s);
        launch_reduce_tilemax(nmax, m->d_tmax, m->d_tij, m->d_info + 2, s);
        (void)hipEventRecord(m->ev[EV_KBUILD], s);
        // ---- factorisation ----
        factorize(m);
    }
    (void)hipEventRecord(m->ev[EV_FACTOR], s);
    // ---- alpha = K^-1 y with fp64-residual refinement ----
    // ir_steps >= 0: exactly that many steps.  Default: adaptive -- at least one step, then until the fp64 residual
    // max|y - K alpha| is below 1e-9 max|y| (at most 4 steps).  Measured at N = 16384 with an fp32 factor, alpha
    // error vs fp64 after 1 / 2 / 3 steps: Matern-5/2 2e-9 / 7e-13 / 3e-14 (stops after 1), thin-plate R=4
    // 2e-4 / 8e-6 / 2e-7 (runs 3); each step costs one substitution pair + one matrix-free residual (2.3 ms).
    const bool ir_adaptive = m->opt.ir_steps < 0;
    const int ir_max = ir_adaptive ? 4 : m->opt.ir_steps;
    double ymax = 0.0;
    for (int i = 0; i < n; ++i)
        ymax = std::max(ymax, std::fabs(m->hlabel[i]));
    const double ir_tol = 1e-9 * std::max(ymax, 1e-300);
    int ir = 0;
    // by_steps: the launch-per-step substitution.  Returns with the stream synchronised and *gave_up telling whether
    // a workgroup of the one-launch substitution stopped waiting (its alpha is then invalid).
    auto compute_alpha = [&](bool by_steps, bool *gave_up) -> int {
        HIPCHK(hipMemsetAsync(m->d_alpha, 0, sizeof(double) * (size_t)np, s));
        for (int it = 0;; ++it) {
            // right-hand side: y (first pass) or the fp64 residual
            launch_cast_vec(m->prec, n, np, it == 0 ? m->d_lab : m->d_r, m->t_b, s);
            solve_ldl(m, m->t_b, m->t_yv, m->t_xs, by_steps);
            launch_axpy_cast(m->prec, n, np, m->d_alpha, m->t_xs, m->t_alpha, s);
            // r = y - K alpha in fp64, matrix-free from the fp64 points
            launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, n, m->d_x, m->d_y, m->d_z,
                           m->d_f, nullptr, m->ws_pred, s);
            HIPCHK(hipMemsetAsync(m->d_rmax, 0, sizeof(double), s));
            launch_residual(n, m->d_lab, m->d_f, m->d_s2, m->d_alpha, m->d_r, m->d_rmax, s);
            ir = it;
            if (it >= ir_max)
                break;
            if (ir_adaptive && it >= 1) {
                double r_now = 0.0;
                HIPCHK(hipMemcpyAsync(&r_now, m->d_rmax, sizeof(double), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (!(r_now > ir_tol))
                    break;
            }
        }
        int flag = 0;
        HIPCHK(hipMemcpyAsync(&flag, m->d_info + 5, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        *gave_up = flag != 0;
        return GPX_OK;
    };
    int64_t solve_fallbacks = 0;
    {
        bool gave_up = false;
        int rc = compute_alpha(false, &gave_up);
        if (rc)
            return rc;
        if (used_dataflow) {
            // (a dataflow factorisation that gave up leaves a void factor -- seen HERE, at the first host sync behind it,
            // instead of after a second substitution, the normals and the statistics have run on it)
            int df_gave_up = 0;
            HIPCHK(hipMemcpy(&df_gave_up, m->d_info + 6, sizeof(int), hipMemcpyDeviceToHost));
            if (df_gave_up) {
                HIPCHK(hipMemsetAsync(m->d_info, 0, sizeof(int) * 8, s));
                mid_ws.reset();
                const int rc2 = build_model(m, nullptr, true);
                m->stats.solve_fallbacks += 1;
                return rc2;
            }
        }
        if (gave_up) {
            // the one-launch substitution relies on lower block rows making progress; when a poll ran out of
            // patience its result is void: clear the flag and redo alpha with one launch per block step
            HIPCHK(hipMemsetAsync(m->d_info + 5, 0, sizeof(int), s));
            solve_fallbacks = 1;
            if ((rc = compute_alpha(true, &gave_up)))
                return rc;
        }
    }
    m->stats.ir_steps_done = ir;
    (void)hipEventRecord(m->ev[EV_SOLVE], s);
    // ---- normals at the training points (create<true>, gp_regressor.hpp:166-181) ----
    if (m->opt.with_normals) {
        if (!m->d_normals)
            HIPCHK(hipMalloc((void **)&m->d_normals, sizeof(double) * 3 * (size_t)n));
        launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, n, m->d_x, m->d_y, m->d_z, m->d_f,
                       m->d_normals, m->ws_pred, s);
        launch_normalize_rows3(n, m->d_normals, s);
        m->has_normals = true;
    }
    (void)hipEventRecord(m->ev[EV_NORMALS], s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    int info[8];
    double rmax = 0;
    HIPCHK(hipMemcpy(info, m->d_info, sizeof(info), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&rmax, m->d_rmax, sizeof(double), hipMemcpyDeviceToHost));
    float ms;
    m->stats = gpx_stats{};
    if (hipEventElapsedTime(&ms, m->ev[EV_T0], m->ev[EV_KBUILD]) == hipSuccess)
        m->stats.t_kbuild_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_KBUILD], m->ev[EV_FACTOR]) == hipSuccess)
        m->stats.t_factor_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_FACTOR], m->ev[EV_SOLVE]) == hipSuccess)
        m->stats.t_solve_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_SOLVE], m->ev[EV_NORMALS]) == hipSuccess)
        m->stats.t_normals_ms = ms;
    double tg = 0;
    for (size_t i = 0; i < m->gemm_ev_used_factor; ++i)
        if (hipEventElapsedTime(&ms, m->gemm_ev[2 * i], m->gemm_ev[2 * i + 1]) == hipSuccess)
            tg += ms;
    m->stats.t_factor_gemm_ms = tg;
    m->stats.factor_gemm_launches = (int64_t)m->gemm_ev_used_factor;
    m->stats.factor_gemm_flops = m->factor_gemm_flops;
    m->stats.n = n;
    m->stats.n_padded = np;
    m->stats.n_negative_pivots = info[1] + (keep ? keep->n_neg : 0);
    m->stats.ir_steps_done = ir;
    m->stats.alpha_residual = rmax;
    m->stats.solve_fallbacks = solve_fallbacks + small_fallbacks;
    if (used_dataflow && info[6] != 0) {
        // a wait of the dataflow factorisation gave up (the GPU could not make progress on its grid): everything computed
        // since is void -- the same create again through the launch chain
        HIPCHK(hipMemsetAsync(m->d_info, 0, sizeof(int) * 8, s));
        mid_ws.reset();
        const int rc = build_model(m, nullptr, true);
        m->stats.solve_fallbacks += 1;
        return rc;
    }
    if (info[5] != 0)  // cannot happen: the step kernels never raise it
        return fail(GPX_E_HIP, "block substitution: give-up flag set after the launch-per-step fallback");
    if (info[0] != 0)
        return fail(GPX_E_SINGULAR, "LDL^T: zero or non-finite pivot at internal row " + std::to_string(info[0] - 1));
    // Model::R (gp_regressor.hpp:135): the device found the arg-max pair, the distance is fp64
    if (!(keep && keep->t0 > 0)) {
        const int a = info[2], b = info[3];
        if (a >= 0 && a < n && b >= 0 && b < n) {
            const int ia = m->perm[a], ib = m->perm[b];
            const double dx = m->hx[ia] - m->hx[ib], dy = m->hy[ia] - m->hy[ib], dz = m->hz[ia] - m->hz[ib];
            m->R = std::sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    {  // weight offset of the variance fit (gpx_internal.hpp), from the extent of the cloud; travels in the meta block
        const double wd = m->R * m->R / VAR_FIT_WDELTA_DIV;
        HIPCHK(hipMemcpy(m->d_meta + 4, &wd, sizeof(double), hipMemcpyHostToDevice));
    }
    m->ready = true;
    // An INDEFINITE kernel matrix (negative pivots: the thin plate with R below the diameter of the cloud, as the node's
    // own R = 2, src/gp_node.cpp:919) keeps its fp64 state: v = k(0) - sum_j w_j^2 / D_j then has terms of both signs that
    // cancel by a factor fp32 cannot carry (sum |w_j^2 / D_j| = 65 k(0) on the node's clouds; the fp32 contraction of such
    // a model measured 1e-3 k(0) off on a random cloud).  The model then predicts like a GPX_PREC_F64 one, whatever
    // precision was asked for (twice the variance time of fp32, on models that are small in practice).
    m->var_fit = m->var_fit_opt;
    m->promoted = false;
    if (m->train64 && m->stats.n_negative_pivots > 0 && gpxh::switches().no_promote <= 0) {
        m->train64 = false;  // no demotion below; the fp64 factor stays (update() can append to it)
        m->var_fit = false;  // the fp64 contraction carries no fit
        m->promoted = true;
    }
    if (keep && keep->t0 > 0 && keep->X)
        append_inverse(m, keep);
    if (m->opt.prepare_variance || m->train64) {
        int rc = build_inverse(m);
        if (rc)
            return rc;
    }
    if (m->train64) {
        int rc = demote_to_f32(m);
        return rc ? rc : pack_split(m);
    }
    return GPX_OK;
}

// Arithmetic of the training stage.  MIXED trains in fp64 by definition.  F32 / F32_SPLIT models of up to
// GPX_TRAIN_F64_MAX padded rows (default 2048; 0 = never) do too: there the whole fp64 create costs under two
// milliseconds, while an fp32 LDL^T accumulates its Schur complements with errors of ~1e-5 k(0) against pivots that
// sink to the noise level sigma^2.  Larger models keep the fp32 MFMA factorisation (N = 4096: 9e-7, N = 16384: 1.1e-6
// with the centred contraction) -- except the thin plate:
// its matrices have cond > 1e6 and predictor weights |K^-1 k_q|_1 of 10-70, so that the backward error of an fp32
// LDL^T shows in the variance (4e-5 k(0) at N = 2305 on a random cloud with extrapolating queries, and k(0) / max|v| is
// ~60 at N = 16384).  Thin-plate models therefore ALWAYS train in fp64 while the fp64 temporaries fit the device
// (kernel matrix, inverse factor and the assembly workspace: 24 N^2 bytes + the fp32 state): the fp64 LDL^T costs 42.7
// instead of 18.6 ms at N = 16384 -- 1 % of a step whose inverse factor is assembled in fp64 anyway -- and the
// variance contraction stays on the fp32 matrix cores.  GPX_TRAIN_F64_MAX overrides both thresholds.
void set_training_precision(gpx_model *m)
{
    const int p = m->opt.precision;
    long thr = m->kern.id == GPX_KERNEL_THINPLATE ? (1L << 20) : 2048;
    if (gpxh::switches().train_f64_max >= 0)
        thr = gpxh::switches().train_f64_max;
    bool f64_fits = true;
    if (m->kern.id == GPX_KERNEL_THINPLATE && m->npad > 8192 && (p == GPX_PREC_F32 || p == GPX_PREC_F32_SPLIT)) {
        size_t free_b = 0, total_b = 0;
        const double need = 30.0 * (double)m->npad * (double)m->npad;  // 3 fp64 N x N buffers + the fp32 state + slack
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            f64_fits = need < 0.9 * (double)total_b;  // parked pool buffers are handed back on demand (gpx_trim in big_alloc)
        else
            (void)hipGetLastError();
    }
    m->train64 = p == GPX_PREC_MIXED || ((p == GPX_PREC_F32 || p == GPX_PREC_F32_SPLIT) && m->npad <= thr && f64_fits);
    m->prec = (p == GPX_PREC_F64 || m->train64) ? GPX_PREC_F64 : GPX_PREC_F32;
    m->esz = m->prec == GPX_PREC_F64 ? 8 : 4;
}

void set_query_batch(gpx_model *m)
{
    if (m->opt.query_batch > 0) {
        m->qbatch = m->opt.query_batch;
        return;
    }
    // ~512 MiB of Kqp per batch: 8192 queries at N = 16384, more for small models so that one variance
    // launch still fills the chip (N = 724: 131072 queries -> 6 x 1024 tiles)
    size_t qb = ((size_t)512 << 20) / ((size_t)m->npad * 4);
    qb = std::min<size_t>(std::max<size_t>(qb, 8192), 131072);
    qb = qb / 256 * 256;
    // The one-wave variance tiles run their paired launch (equal-length workgroups in step: a quarter of the L2-miss
    // traffic, gpx_vargemm.hip) when (row tiles / 2) x (query tiles) fills the SIMDs in whole rounds: round the batch down
    // to the nearest size that does, if there is one within a factor of two.
    const long mt = ((long)m->n + TILE - 1) / TILE;
    int cus = 0;
    if (mt >= 2 && mt % 2 == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device) == hipSuccess &&
        cus > 0) {
        const long slots = 4L * cus;
        long a = mt / 2, b = slots;
        while (b) {
            const long t = a % b;
            a = b, b = t;
        }
        const size_t step = (size_t)TILE * (size_t)(slots / a);  // queries per whole round of pairs
        if (step <= qb && qb / step * step >= qb / 2)
            qb = qb / step * step;
    }
    m->qbatch = (int)qb;
}

}  // namespace gpxh
