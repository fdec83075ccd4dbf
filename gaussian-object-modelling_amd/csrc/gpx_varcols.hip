// gpx_varcols.hip -- host side of the small-model variance contraction (kernel: gpx_varcols_kernel.hpp): eligibility,
// whether the wave forms its own operand, the launch.
#include "gpx_varcols_kernel.hpp"

namespace gpx {

bool var_cols_fits(int n, int np, long ldx, long ldk)
{
    return n > 0 && n <= VARCOLS_MAX_N && ldx % 4 == 0 && ldk % 4 == 0 && (long)np * ldx * 4 < (1L << 31) &&
           64L * ldk * 4 < (1L << 31);
}

// the operand can be formed inside the wave when its arithmetic is fp32 (every kernel but the thin plate, whose operand is
// formed in fp64 and rounded once: gpx_internal.hpp, "low-rank fit" -- that model reads the operand buffer)
bool var_cols_gen(const VarColsArgs &a)
{
    return !a.op64 && a.cov.id != GPX_KERNEL_THINPLATE && a.px && a.qx;
}

void launch_var_cols(const VarColsArgs &a, hipStream_t st)
{
    VarColsDev g;
    g.X = a.X, g.ldx = a.ldx, g.x_rows = a.np;
    g.Kq = a.Kq, g.ldk = a.ldk;
    g.rowcorr = a.rowcorr, g.ldrc = a.ldrc;
    g.colcoef = a.colcoef, g.ldcc = a.ldcc;
    g.dinv64 = a.dinv64;
    g.v = a.v, g.k0 = a.k0;
    g.nq_valid = a.nq_valid;
    g.nfrag = (a.n + 15) / 16;
    g.px = a.px, g.py = a.py, g.pz = a.pz;
    g.qx = a.qx, g.qy = a.qy, g.qz = a.qz;
    g.cen[0] = a.cen[0], g.cen[1] = a.cen[1], g.cen[2] = a.cen[2];
    g.n = a.n;
    g.cov = lower_cov<float>(a.cov);
    const bool gen = var_cols_gen(a);
    g.compact_coef = gen && a.compact_coef ? 1 : 0;
    const unsigned nwg = (unsigned)((a.nq_tile + 16 * VC_CF - 1) / (16 * VC_CF));
#ifdef VC_TIMING
    static long long *dbg = nullptr;
    if (!dbg)
        (void)hipMalloc(&dbg, (8 * 8192 + 128) * sizeof(long long));
    g.dbg = dbg;
#endif
    if (gen) {
        switch (a.cov.id) {  // (Gaussian and Laplace are the same function of (a, s): gpx_cov.hpp)
        case GPX_KERNEL_MATERN32:
            hipLaunchKernelGGL((var_cols_kernel<VC_CF, VC_FS, true, GPX_KERNEL_MATERN32>), dim3(nwg), dim3(64), 0, st, g);
            break;
        case GPX_KERNEL_MATERN52:
            hipLaunchKernelGGL((var_cols_kernel<VC_CF, VC_FS, true, GPX_KERNEL_MATERN52>), dim3(nwg), dim3(64), 0, st, g);
            break;
        default:
            hipLaunchKernelGGL((var_cols_kernel<VC_CF, VC_FS, true, GPX_KERNEL_GAUSSIAN>), dim3(nwg), dim3(64), 0, st, g);
            break;
        }
    } else {
        hipLaunchKernelGGL((var_cols_kernel<VC_CF, VC_FS, false, 0>), dim3(nwg), dim3(64), 0, st, g);
    }
#ifdef VC_TIMING
    {  // diagnostic build: per-wave phase times of the third launch per model size, per-chunk times of one wave
        static int calls = 0, last_n = 0;
        if (last_n != a.n)
            calls = 0, last_n = a.n;
        if (++calls == 3) {
            (void)hipStreamSynchronize(st);
            std::vector<long long> h(8 * 8192 + 128);
            (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            const unsigned nb = std::min(nwg, 8192u);
            double d[3] = {0, 0, 0}, cyc = 0, rt = 0;
            for (unsigned b = 0; b < nb; ++b) {
                for (int i = 0; i < 3; ++i)
                    d[i] += (double)(h[8 * b + i + 1] - h[8 * b + i]);
                cyc += (double)(h[8 * b + 3] - h[8 * b]), rt += (double)h[8 * b + 4];
            }
            fprintf(stderr, "[vc timing] n=%d F=%d waves=%u gen=%d: to the end of the last pass's prologue %.0f, its main part %.0f, tail %.0f "
                            "cycles per wave; shader clock %.3f GHz; chunks of the last pass of wave 1500:",
                    a.n, g.nfrag, nwg, (int)gen, d[0] / nb, d[1] / nb, d[2] / nb, cyc / rt / 10.0);
            for (int i = VC_FS - std::min(g.nfrag, VC_FS); i < VC_FS; ++i)
                fprintf(stderr, " %lld", h[8 * 8192 + i + 1] - h[8 * 8192 + i]);
            fprintf(stderr, "\n");
        }
    }
#endif
}

}  // namespace gpx
