// gpx_varcols16.hip -- the variance of SMALL models in the opt-in GPX_PREC_F32_SPLIT mode on the fp16 matrix cores
// (VERDICT r4 item 6): the structure of gpx_varcols64.hip (a wave owns 32 queries and up to 16 row fragments, operand formed
// in the wave, slots counted from the end of the pass, X streamed in fragment order, persistent workgroups) with the arithmetic
// of gpx_vsplit.hip (hi / lo fp16 halves on shared quanta, three v_mfma_f32_16x16x32_f16 per fragment pair into a main and a
// correction accumulator) and the fit of the fp32 modes (the operand is k - fit, X * fit is added back in fp64 from the 14
// row-correction vectors on the fp64 matrix pipe, gpx_internal.hpp "low-rank fit").
//
// Why it pays where the f32-input kernel (gpx_varcols_kernel.hpp) is stuck at 41-66 % of ITS peak: a 16 x 16 x 32 fp16 MFMA
// takes 16.7 cycles for 8x the multiply-adds of the 32-cycle f32 16 x 16 x 4, and two vector-ALU instructions issue behind
// each one for free (scripts/mfma_filler_probe2.hip, profiles/r05_mfma_filler_probe2.txt) where the f32-input form hides
// nothing: the three products of the split cost a fifth of the matrix time, and the kernel becomes a vector-ALU kernel
// (operand evaluation + split) with the matrix work largely behind it.
//
// Per call the fp32 inverse factor of the model (small F32_SPLIT models keep it unpacked, gpx_build.hip split_packs) is
// scaled, split and laid out in fragment order by two small launches (max |X|, then pack16_kernel): chunk c (32 columns) of
// row fragment f (16 rows) = 2 KB, [hi: 64 lanes x 16 B | lo: 64 lanes x 16 B], lane l = row l % 16, columns 8 (l / 16) .. + 8
// -- the 8 consecutive k of one split8 group, which is what one lane feeds the MFMA.
// Exponential kernels only (the thin plate forms its operand in fp64: those models keep the fp32 kernel).
// GPX_VAR_COLS16=0 (read per call): the fp32 small-model kernel, as before this file existed.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gpx_cov.hpp"
#include "gpx_internal.hpp"
#include "gpx_split.hpp"

namespace gpx {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
#ifndef VC16_FS
#define VC16_FS 22
#endif
#ifndef VC16_AGPR
#define VC16_AGPR 16
#endif
constexpr int FS16 = VC16_FS;  // row-fragment slots per pass
constexpr int AGPR16 = VC16_AGPR;  // slots whose accumulators (main + correction, two column fragments: 16 registers) live in AGPRs
constexpr int CF16 = 2;      // column fragments (16 queries each) per wave
constexpr int WAVES16 = 4;   // waves per workgroup, one per SIMD
constexpr int AHEAD16 = 4, RING16 = AHEAD16 + 1;
// diagnostic builds only (make EXTRA=-DVC16_DBG=n OUTDIR=../lib_dn OBJDIR=../build_dn; results wrong by construction):
// 1 operand not evaluated, 2 no MFMAs in the chunk loop, 3 no requests for X -- how the kernel's time splits
#ifndef VC16_DBG
#define VC16_DBG 0
#endif

struct VarCols16Dev {
    const half8 *Xq;  // X in fragment order (pack16_kernel)
    int fp;           // fragments per side of the packed copy
    int n, nfrag;     // F = ceil(n / 16)
    const float *px, *py, *pz;  // centred fp32 points
    const double *qx, *qy, *qz;
    double cen[3];
    const double *fab;  // [3][ldcc]: a_q, b_q, c_q of the batch (launch_var_fit, compact)
    long ldcc;
    const double *rowcorr;  // [VAR_NCORR][ldrc]
    long ldrc;
    const double *dinv64;
    const double *inv_scale;  // 1 / (sx sk), written by pack16_kernel
    float sk;
    double k0;
    double *v;
    long nq;
    Cov<float> cov;
};

// f(0), f(1), ... while the slot number is below nact: nested branches, i.e. straight-line code with one way out
template <int R, int N, class Fn>
__device__ __forceinline__ void slot_chain16(int nact, Fn &&f)
{
    if constexpr (R < N) {
        if (R < nact) {
            f(std::integral_constant<int, R>{});
            slot_chain16<R + 1, N>(nact, f);
        }
    }
}

template <int KID>
__global__ __launch_bounds__(64 * WAVES16, 1) void var_cols16_kernel(VarCols16Dev g)
{
    // LDS (dynamic, var_cols16_lds_bytes): the 14 row-correction vectors and 1 / D of the model's rows in fp64 -- the operands of
    // the add-back, which would otherwise wait for global loads once per slot -- then the centred fp32 points
    extern __shared__ double lds16[];
    const int lane = threadIdx.x & 63, r16 = lane & 15, lg = lane >> 4;
    const int F = g.nfrag, n = g.n;
    const int nr = 16 * F, npts = 16 * ((F + 1) / 2 * 2);
    double *rowc = lds16, *dinv = lds16 + VAR_NCORR * nr;
    float *lpx = reinterpret_cast<float *>(dinv + nr), *lpy = lpx + npts, *lpz = lpy + npts;
    for (int k = threadIdx.x; k < npts; k += 64 * WAVES16) {
        const bool in = k < n;
        lpx[k] = in ? g.px[k] : 0.f, lpy[k] = in ? g.py[k] : 0.f, lpz[k] = in ? g.pz[k] : 0.f;
    }
    for (int k = threadIdx.x; k < nr; k += 64 * WAVES16) {
#pragma unroll
        for (int c = 0; c < VAR_NCORR; ++c)
            rowc[c * nr + k] = g.rowcorr[(size_t)c * g.ldrc + k];
        dinv[k] = k < n ? g.dinv64[k] : 0.0;  // (rows of the padding: weight 0, so the operand needs no mask -- the points of the
                                              // padding sit at the centre and meet zero columns of X in the rows of the model)
    }
    const Cov<float> cov = g.cov;
    const double inv = *g.inv_scale;
    const float sk = g.sk;
    // v_mfma_f64_16x16x4_f64 hands rows lg, lg + 4, lg + 8, lg + 12 of a fragment to a lane where the fp32 form hands rows
    // 4 lg .. 4 lg + 3: the A operand of the add-back is fed with its rows permuted, so that result register r of a lane is
    // row 4 lg + r -- the accumulators' layout (as in gpx_varcols_kernel.hpp)
    const int prow = 4 * (r16 & 3) + (r16 >> 2);
    __syncthreads();  // (the kernel's only barrier)
    const long per_wg = 16L * CF16 * WAVES16, nblk = (g.nq + per_wg - 1) / per_wg;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long q0 = (blk * WAVES16 + (threadIdx.x >> 6)) * (16 * CF16);
        if (q0 >= g.nq)
            break;  // (a wave past the last query)
        // the lane's query of either column fragment: centred fp32 coordinates, a_q b_q c_q of the fit as the operand kernel
        // rounds them (kqp_split_kernel<float>), and its four coefficients 4 s + lg of the rank-14 add-back in fp64
        float ax[CF16], ay[CF16], az[CF16], fa[CF16], fb[CF16], fc[CF16];
        double cb[CF16][4];
#pragma unroll
        for (int j = 0; j < CF16; ++j) {  // (columns past the last query work on the last query's data and are not written)
            const long q = q0 + 16 * j + r16, qc = q < g.nq ? q : g.nq - 1;
            ax[j] = (float)(g.qx[qc] - g.cen[0]), ay[j] = (float)(g.qy[qc] - g.cen[1]), az[j] = (float)(g.qz[qc] - g.cen[2]);
            const double da = g.fab[qc], db = g.fab[g.ldcc + qc], dc = g.fab[2 * g.ldcc + qc];
            fa[j] = (float)da, fb[j] = (float)db, fc[j] = (float)dc;
            double cf[VAR_NCORR];
            var_fit_coefs(da, db, dc, (double)ax[j], (double)ay[j], (double)az[j], cf);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double val = 0.0;
#pragma unroll
                for (int c = 0; c < VAR_NCORR; ++c)
                    val = (4 * s + lg == c) ? cf[c] : val;
                cb[j][s] = val;
            }
        }
        const int npass = (F + FS16 - 1) / FS16;
        double colsum[CF16];
#pragma unroll
        for (int j = 0; j < CF16; ++j)
            colsum[j] = 0.0;
        int f_lo = 0;
        for (int p = 0; p < npass; ++p) {
            const int nfr = p == 0 ? F - FS16 * (npass - 1) : FS16, f_hi = f_lo + nfr;
            f32x4 acc[FS16][CF16], cor[FS16][CF16];
            // (AGPR accumulators zeroed by an MFMA of zeros: every definition tied to an AGPR, see gpx_varcols64.hip)
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int il = 0; il < FS16; ++il)
#pragma unroll
                for (int j = 0; j < CF16; ++j)
                    if (il < AGPR16) {
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, 0" : "=a"(acc[il][j]) : "v"(zero8));
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, 0" : "=a"(cor[il][j]) : "v"(zero8));
                    } else {
                        acc[il][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        cor[il][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            const int nchunk = (f_hi + 1) / 2;  // 32-column chunks that meet the pass's rows
#pragma nounroll
            for (int c = 0; c < nchunk; ++c) {
                // row fragments max(2 c, f_lo) .. f_hi - 1 take part (fragment f ends at column 16 f + 15 >= 32 c): the range ends
                // at the pass's last fragment, slot r = fragment f_hi - 1 - r
                const int nact = f_hi - max(2 * c, f_lo);
                const int xc = (c * g.fp + f_hi - 1) * 128;  // in 16-byte units: 2 KB per (chunk, fragment)
                half8 ahi[RING16], alo[RING16];
                auto load_a = [&](int r, half8 &hi, half8 &lo) {
                    if constexpr (VC16_DBG == 3) {
                        hi = lo = half8{1, 1, 1, 1, 1, 1, 1, 1};
                        return;
                    }
                    const half8 *src = g.Xq + (xc - 128 * min(r, nact - 1)) + lane;
                    hi = src[0], lo = src[64];
                };
#pragma unroll
                for (int u = 0; u < AHEAD16; ++u)
                    load_a(u, ahi[u], alo[u]);
                // the lane's operand values of the chunk: k(|q - p|) - fit, p = 32 c + 8 lg + e, split on one quantum per group
                half8 bh[CF16], bl[CF16];
                {
                    const int p0 = 32 * c + 8 * lg;
                    float x[8], y[8], z[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[e] = lpx[p0 + e], y[e] = lpy[p0 + e], z[e] = lpz[p0 + e];
#pragma unroll
                    for (int j = 0; j < CF16; ++j) {
                        float val[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float dx = ax[j] - x[e], dy = ay[j] - y[e], dz = az[j] - z[e];
                            const float d2 = dx * dx + dy * dy + dz * dz;
                            float kv = VC16_DBG == 1 ? d2 : cov_k<float, KID>(cov, d2);
                            if (VC16_DBG != 1)
                                kv -= fa[j] + d2 * (fb[j] + fc[j] * d2);
                            val[e] = kv;
                        }
                        if constexpr (VC16_DBG == 1) {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                bh[j][e] = bl[j][e] = (half_t)val[e];
                        } else {
                            split8(val, sk, bh[j], bl[j]);
                        }
                    }
                }
                slot_chain16<0, FS16>(nact, [&](auto slot) {
                    constexpr int il = decltype(slot)::value;
                    f32x4 &m0 = acc[il][0], &m1 = acc[il][1], &c0 = cor[il][0], &c1 = cor[il][1];
                    const half8 bh0 = bh[0], bh1 = bh[1], bl0 = bl[0], bl1 = bl[1];
                    const half8 ah = ahi[il % RING16], al = alo[il % RING16];
                    load_a(il + AHEAD16, ahi[(il + AHEAD16) % RING16], alo[(il + AHEAD16) % RING16]);
#define VC16_MFMAS(CLS_)                                                                               \
    asm volatile("s_nop 1\n"                                                                           \
                 "v_mfma_f32_16x16x32_f16 %2, %5, %6, %2\n"                                            \
                 "v_mfma_f32_16x16x32_f16 %3, %5, %7, %3\n"                                            \
                 "v_mfma_f32_16x16x32_f16 %0, %4, %6, %0\n"                                            \
                 "v_mfma_f32_16x16x32_f16 %1, %4, %7, %1\n"                                            \
                 "v_mfma_f32_16x16x32_f16 %2, %4, %8, %2\n"                                            \
                 "v_mfma_f32_16x16x32_f16 %3, %4, %9, %3"                                              \
                 : CLS_(m0), CLS_(m1), CLS_(c0), CLS_(c1)                                              \
                 : "v"(ah), "v"(al), "v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1))
                    if constexpr (VC16_DBG == 2 && il < AGPR16)
                        asm volatile("" : "+a"(m0), "+a"(m1), "+a"(c0), "+a"(c1) : "v"(ah), "v"(al), "v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
                    else if constexpr (VC16_DBG == 2)
                        asm volatile("" : "+v"(m0), "+v"(m1), "+v"(c0), "+v"(c1) : "v"(ah), "v"(al), "v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
                    else if constexpr (il < AGPR16)
                        VC16_MFMAS("+a");
                    else
                        VC16_MFMAS("+v");
#undef VC16_MFMAS
                });
            }
            // the accumulators are read by the VALU from here on: the MFMA's wait states first, every accumulator tied behind them
            asm volatile("s_nop 15\n s_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(cor[0][0]), "+a"(cor[0][1]));
#pragma unroll
            for (int il = 1; il < FS16; ++il)
                if (il < AGPR16)
                    asm volatile("" : "+a"(acc[il][0]), "+a"(acc[il][1]), "+a"(cor[il][0]), "+a"(cor[il][1]));
                else
                    asm volatile("" : "+v"(acc[il][0]), "+v"(acc[il][1]), "+v"(cor[il][0]), "+v"(cor[il][1]));
            // w = (main + 2^-11 correction) / (sx sk) + (X fit), the add-back from the row-correction vectors on the fp64 matrix
            // pipe; then w^2 / D.  Register r of lane (lg, query) in slot il is row 16 (f_hi - 1 - il) + 4 lg + r.
#pragma unroll
            for (int il = 0; il < FS16; ++il)
                if (il < nfr) {
                    const int row0 = 16 * (f_hi - 1 - il);
                    double ra[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s)  // (vectors 14, 15 do not exist: the column side is zero there)
                        ra[s] = rowc[min(4 * s + lg, VAR_NCORR - 1) * nr + row0 + prow];
                    f64x4 d[CF16];
#pragma unroll
                    for (int j = 0; j < CF16; ++j) {
                        d[j] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            d[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[s], cb[j][s], d[j], 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double di = dinv[row0 + 4 * lg + r];
#pragma unroll
                        for (int j = 0; j < CF16; ++j) {
                            const double w = fma((double)acc[il][j][r] + (double)cor[il][j][r] * (1.0 / 2048.0), inv, d[j][r]);
                            colsum[j] = fma(w * w, di, colsum[j]);
                        }
                    }
                }
            f_lo = f_hi;
        }
#pragma unroll
        for (int j = 0; j < CF16; ++j) {
            double cs = colsum[j];
            cs += __shfl_xor(cs, 16);
            cs += __shfl_xor(cs, 32);
            const long q = q0 + 16 * j + r16;
            if (lg == 0 && q < g.nq)
                g.v[q] = g.k0 - cs;
        }
    }
}

// max |X| over the leading rows x rows part (bits of a non-negative float order like the float)
__global__ __launch_bounds__(256) void absmax16_kernel(const float *X, long ldx, int rows, unsigned *out_bits)
{
    float m = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * (rows / 4); i += (long)gridDim.x * 256) {
        const long r = i / (rows / 4), c4 = i % (rows / 4);
        const float4 v = *reinterpret_cast<const float4 *>(X + r * ldx + 4 * c4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int off = 32; off > 0; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        atomicMax(out_bits, __float_as_uint(m));
}

// X (fp32, lower triangular) scaled, split and laid out in the order the kernel reads it (header); *inv_scale = 1 / (sx sk)
__global__ __launch_bounds__(64) void pack16_kernel(const float *X, long ldx, half8 *Xq, int fp, const unsigned *amax_bits,
                                                    float sk, double *inv_scale)
{
    const int f = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
    const float sx = pow2_scale_below_one(__uint_as_float(*amax_bits));
    if (f == 0 && c == 0 && lane == 0)
        *inv_scale = 1.0 / ((double)sx * (double)sk);
    if (f < 2 * c)
        return;  // (fragments that end in front of the chunk are never read)
    const float *src = X + (size_t)(16 * f + (lane & 15)) * ldx + 32 * c + 8 * (lane >> 4);
    const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    half8 hi, lo;
    split8(v, sx, hi, lo);
    half8 *dst = Xq + (size_t)(c * fp + f) * 128 + lane;
    dst[0] = hi, dst[64] = lo;
}
}  // namespace

static size_t var_cols16_lds_bytes(int n)
{
    const size_t F = ((size_t)n + 15) / 16, nr = 16 * F, npts = 16 * ((F + 1) / 2 * 2);
    return (VAR_NCORR + 1) * nr * sizeof(double) + 3 * npts * sizeof(float);
}

size_t var_cols16_ws_bytes(int n)
{
    const size_t fp = ((size_t)n + 15) / 16, nc = (fp + 1) / 2;
    return nc * fp * 2048 + 64;  // the packed copy, then max |X| (bits) and 1 / (sx sk)
}

// the kernel forms its operand in fp32 from the centred points: every kernel but the thin plate (whose operand is formed in
// fp64: those models keep the fp32 small-model kernel); GPX_VAR_COLS16=0: never
bool var_cols16_takes(const VarColsArgs &a)
{
    if (gpxh::switches().var_cols16 == 0)
        return false;
    return !a.op64 && a.cov.id != GPX_KERNEL_THINPLATE && a.px && a.qx && a.n > 0 && a.n <= VARCOLS_MAX_N &&
           a.ldx % 4 == 0 && a.np % 32 == 0;
}

void launch_var_cols16_pack(const VarColsArgs &a, float sk, void *ws, hipStream_t st)
{
    const int fp = (a.n + 15) / 16, nc = (fp + 1) / 2, rows = 16 * ((fp + 1) / 2 * 2);
    unsigned char *tail = (unsigned char *)ws + (size_t)nc * fp * 2048;
    unsigned *amax = (unsigned *)tail;
    double *inv_scale = (double *)(tail + 16);
    (void)hipMemsetAsync(amax, 0, sizeof(unsigned), st);
    hipLaunchKernelGGL(absmax16_kernel, dim3(64), dim3(256), 0, st, a.X, a.ldx, rows, amax);
    hipLaunchKernelGGL(pack16_kernel, dim3(fp, nc), dim3(64), 0, st, a.X, a.ldx, (half8 *)ws, fp, amax, sk, inv_scale);
}

void launch_var_cols16(const VarColsArgs &a, float sk, const void *ws, hipStream_t st)
{
    const int fp = (a.n + 15) / 16, nc = (fp + 1) / 2;
    const unsigned char *tail = (const unsigned char *)ws + (size_t)nc * fp * 2048;
    VarCols16Dev g;
    g.Xq = (const half8 *)ws, g.fp = fp;
    g.n = a.n, g.nfrag = fp;
    g.px = a.px, g.py = a.py, g.pz = a.pz;
    g.qx = a.qx, g.qy = a.qy, g.qz = a.qz;
    g.cen[0] = a.cen[0], g.cen[1] = a.cen[1], g.cen[2] = a.cen[2];
    g.fab = a.colcoef, g.ldcc = a.ldcc;
    g.rowcorr = a.rowcorr, g.ldrc = a.ldrc;
    g.dinv64 = a.dinv64;
    g.inv_scale = (const double *)(tail + 16);
    g.sk = sk;
    g.k0 = a.k0;
    g.v = a.v, g.nq = a.nq_valid;
    g.cov = lower_cov<float>(a.cov);
    const long per_wg = 16L * CF16 * WAVES16, nblk = (a.nq_valid + per_wg - 1) / per_wg;
    int devid = 0, ncu = 0;
    (void)hipGetDevice(&devid);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || ncu <= 0)
        ncu = 256;
    const unsigned nwg = (unsigned)std::min<long>(nblk, ncu);
    const size_t lds = var_cols16_lds_bytes(a.n);
    static PerDeviceOnce attr_once;  // (the LDS-size attribute is per device)
    attr_once.run([&] {
        const int mx = (int)var_cols16_lds_bytes(VARCOLS_MAX_N);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&var_cols16_kernel<GPX_KERNEL_MATERN32>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&var_cols16_kernel<GPX_KERNEL_MATERN52>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&var_cols16_kernel<GPX_KERNEL_GAUSSIAN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, mx);
    });
    switch (a.cov.id) {  // (Gaussian and Laplace are the same function of (a, s): gpx_cov.hpp)
    case GPX_KERNEL_MATERN32:
        hipLaunchKernelGGL((var_cols16_kernel<GPX_KERNEL_MATERN32>), dim3(nwg), dim3(64 * WAVES16), lds, st, g);
        break;
    case GPX_KERNEL_MATERN52:
        hipLaunchKernelGGL((var_cols16_kernel<GPX_KERNEL_MATERN52>), dim3(nwg), dim3(64 * WAVES16), lds, st, g);
        break;
    default:
        hipLaunchKernelGGL((var_cols16_kernel<GPX_KERNEL_GAUSSIAN>), dim3(nwg), dim3(64 * WAVES16), lds, st, g);
        break;
    }
}

}  // namespace gpx
