// gpx_small.hpp -- interface of the three-launch create of small models (gpx_small.hip; host side: build_model_small in
// gpx_build.hip).
#pragma once
#include "gpx_internal.hpp"

namespace gpx {

constexpr int SMALL_TILE = 64;          // tile edge of the dataflow factorisation
constexpr int SMALL_CREATE_MAX_NP = 1024;  // padded rows up to which create() takes this path (fp64 training only)
constexpr int SMALL_ALPHA_MAX_GRID = 128;  // workgroups of the second launch (64 up to 512 padded rows), all resident: it has grid barriers

// what the host reads back after the launches (one copy)
struct SmallResult {
    int info[8];   // as gpx_model::d_info: [0] first bad pivot (1-based), [1] negative pivots, [2..3] arg-max pair of the
                   // squared distance (internal indices), [5] a wait gave up (the results are void)
    double rmax;   // max |y - K alpha| of the last residual
    double d2max;  // largest squared training distance
    int ir_done;
    int pad;
};

struct SmallArgs {
    // ---- input ----
    const double *stage = nullptr;  // [5][np]: x y z label sigma2 in internal (pivot) order, zero padded
    int n = 0, np = 0, nbt = 0 /* np / 64 */, nb = 0 /* tile rows that hold training points */, ntiles = 0;
    Cov<double> cov{};
    double cen[3] = {0, 0, 0};
    int want_corr = 0, op64 = 0, ir_max = 4, ir_adaptive = 1;
    double ir_tol = 0;
    unsigned long long epoch = 0;
    long long wait_ticks = 2000000;  // time budget of every wait, 100 MHz ticks (wait_budget_ticks)
    int abort_idx = 0, bar_idx = 0, pre_idx = 0;
    // ---- model state written by the launches ----
    double *K = nullptr, *X = nullptr, *linv = nullptr, *d = nullptr, *dinv = nullptr;
    double *d_x = nullptr, *d_y = nullptr, *d_z = nullptr, *t_x = nullptr, *t_y = nullptr, *t_z = nullptr;
    double *d_lab = nullptr, *d_s2 = nullptr, *t_s2 = nullptr, *d_alpha = nullptr, *t_alpha = nullptr, *d_r = nullptr;
    double *d_corr = nullptr, *d_dinv64 = nullptr, *d_meta = nullptr;
    int *info = nullptr;
    const double *blob0 = nullptr;  // demotion: source (fp64 layout) and the fp32 state it fills
    void *nblob = nullptr;
    float *nX = nullptr;
    // ---- workspace ----
    double *XT = nullptr;              // transposed copy of X (read on and above its diagonal)
    unsigned long long *flags = nullptr;  // [ntiles] factor tiles, [ntiles] inverse tiles, abort, barrier counter, [nbt] handed-over sums
    double *tmax = nullptr, *rmaxv = nullptr, *u = nullptr, *res_d = nullptr;
    int *tij = nullptr, *negcnt = nullptr, *badrow = nullptr;
    SmallResult *res = nullptr;
    unsigned long long *dbg = nullptr;  // make EXTRA=-DSM_TIMING: [ntiles][SMALL_DBG_STAMPS] wall-clock stamps (100 MHz) per workgroup
};
constexpr int SMALL_DBG_STAMPS = 64;

// byte offsets of the workspace block for a padded order np (one big_alloc per model; layout below)
struct SmallWs {
    size_t stage, args, stage_bytes /* staging + argument block: one host-to-device copy */, xt, flags, tmax, rmaxv, u, tij, negcnt, badrow, res, res_d, dbg, bytes;
};
inline SmallWs small_ws_layout(int np)
{
    const size_t nbt = (size_t)np / SMALL_TILE, nt = nbt * (nbt + 1) / 2;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    SmallWs w{};
    size_t o = 0;
    w.stage = o, o += al(sizeof(double) * 5 * np);
    w.args = o, o += al(sizeof(SmallArgs));
    w.stage_bytes = o - w.stage;
    w.xt = o, o += al(sizeof(double) * (size_t)np * np);
    w.flags = o, o += al(sizeof(unsigned long long) * (2 * nt + 2 + nbt));
    w.tmax = o, o += al(sizeof(double) * nt);
    w.rmaxv = o, o += al(sizeof(double) * 8);
    w.u = o, o += al(sizeof(double) * np);
    w.tij = o, o += al(sizeof(int) * 2 * nt);
    w.negcnt = o, o += al(sizeof(int) * nbt);
    w.badrow = o, o += al(sizeof(int) * nbt);
    w.res = o, o += al(sizeof(SmallResult));   // res and res_d are adjacent: one device-to-host copy
    w.res_d = o, o += al(sizeof(double) * np);
    w.dbg = o, o += al(sizeof(unsigned long long) * nt * 64);
    w.bytes = o;
    return w;
}

// ---- the same dataflow factorisation for mid-size models (kernel matrix + LDL^T in one launch; the chain goes on from there)
constexpr int MID_FACTOR_MAX_NP_F32 = 16384, MID_FACTOR_MAX_NP_F64 = 16384;  // padded rows up to which a fresh create() factorises this way
constexpr int MID_FACTOR_FORCED_MAX_NP = 32768;  // ... when GPX_DATAFLOW=64|128 forces a tile form (tests run it at 20480 rows)
constexpr int WIDE_FACTOR_MIN_NP = 8192;         // padded rows from which the 128 x 128 tiles are used
struct MidWs {
    size_t flags, tmax, tij, negcnt, badrow, bytes;
};
inline MidWs mid_ws_layout(int np)
{
    const size_t nbt = (size_t)np / SMALL_TILE, nt = nbt * (nbt + 1) / 2;
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    MidWs w{};
    size_t o = 0;
    w.flags = o, o += al(sizeof(unsigned long long) * (2 * nt + 2 + nbt));
    w.tmax = o, o += al(sizeof(double) * nt);
    w.tij = o, o += al(sizeof(int) * 2 * nt);
    w.negcnt = o, o += al(sizeof(int) * nbt);
    w.badrow = o, o += al(sizeof(int) * nbt);
    w.bytes = o;
    return w;
}
struct MidFactorArgs {
    int n = 0, np = 0;
    void *K = nullptr, *linv = nullptr, *d = nullptr, *dinv = nullptr;                    // working type
    const void *px = nullptr, *py = nullptr, *pz = nullptr, *ps2 = nullptr;  // centred points, sigma2 (working type, np long)
    void *ws = nullptr;  // mid_ws_layout(np).bytes
    int *info = nullptr;  // the model's d_info: [0..3] as the chain leaves them, [6] = 1 when a wait gave up
    unsigned long long epoch = 0;
    long long wait_ticks = 2000000;  // time budget of every wait, 100 MHz ticks (wait_budget_ticks)
    bool wide = false;  // 128 x 128 tiles (gpx_dataflow_wide.hpp): large models, where the 64 x 64 form is HBM-bound
};
// kernel matrix + LDL^T (L, D in K; d, dinv; the 128 x 128 inverse diagonal blocks linv) + the info reduction; asynchronous
void launch_mid_factor(int prec, const CovHost &cov, const MidFactorArgs &m, hipStream_t st);

void small_create_init();                  // per-device kernel attributes
unsigned long long small_create_epoch();   // a value no earlier create of this process has used (flags are never cleared)
// factor (+ record ev_factor), alpha (+ record ev_solve) and, when demote, the fp32 state; asynchronous on st
// d_args: the device copy of `a` (the launches read their ~60 arguments from memory: passed by value they cost 95 spilled
// scalar registers)
void launch_small_create(int kernel_id, const SmallArgs &a, const SmallArgs *d_args, bool demote, hipStream_t st,
                         hipEvent_t ev_factor, hipEvent_t ev_solve);

}  // namespace gpx
