// gpx_vsplit.hip -- the variance contraction on the fp16 matrix cores at fp32 accuracy (GPX_PREC_F32_SPLIT):
// every fp32 operand x is carried as two halves,
//     hi = fp16(s x) on a per-group grid,   lo = fp16(2^11 (s x - hi))      (s: a power of two bringing max|x| below 1)
// and
//     x y  ~  hi_x hi_y  +  2^-11 (hi_x lo_y + lo_x hi_y)
// runs as three v_mfma_f32_16x16x32_f16 into TWO fp32 accumulator sets (main, correction) that are combined
// once in the epilogue; the dropped lo*lo term is below 2^-22 relative.  Three fp16 MFMAs move 16x more
// flops per cycle than one fp32 MFMA, so the contraction costs ~5x fewer matrix-pipe cycles; what remains
// is a staging problem (the operands are as many bytes as in fp32).
//
// Accuracy hinges on how the matrix core adds: the 8 products one lane feeds (8 consecutive k) are summed in
// fixed point, aligned to the LARGEST of them and truncated 24 bits below it (scripts/mfma_tree_probe.hip, on the
// 32x32x16 form; the 16x16x32 form shipped since round 4 gives the same errors on the headline workload), so the
// error scales with the largest product, not with the sum -- and the rows of the inverse factor times the kernel
// values cancel by 2-3 orders of magnitude (sum|x k| / |sum x k| ~ 700 at N = 1500).  With free-floating fp16 hi
// parts that cost 7x the native fp32 error (3e-5 k(0) at N = 16384).  Therefore the hi parts of each such group of
// 8 share one quantum (split8 below): all hi*hi products of a group are integer multiples of one power of two
// within 22 bits of the largest, the group sum is exact, and the result is as accurate as the fp32 MFMA path
// (measured variance error / k(0) at N = 16384: 1.9e-6 vs 4.5e-6 native for Matern-5/2, 1.8e-5 vs 2.1e-5 thin-plate).
//
// Packed layout "P16" of a [rows][K] matrix (K a multiple of 32): per row, per block of 32 k, 64 halves =
// [hi(32) | lo(32)] = 128 bytes.  A row is exactly as long as in fp32, and one k-tile of a row is one
// 128-byte line: eight lanes of one LDS-DMA instruction bring it in.
//
//   split_absmax / split_pack : X (fp32, inverse factor)  -> P16, scale chosen on the device
//   kqp_split                 : Kqp[q][j] = k(|q-p_j|)     -> P16 directly (never stored in fp32)
//   vsplit_gemm               : partial[mt][q] = sum_rows (X Kqp^T)^2 w[row],  w = 1/D / (sx sk)^2
#include "gpx_cov.hpp"
#include "gpx_split.hpp"

namespace gpx {

// ---- scale + split of the inverse factor ---------------------------------------------------------------
__global__ __launch_bounds__(256) void split_absmax_kernel(const float *__restrict__ X, size_t n,
                                                           unsigned *__restrict__ out_bits)
{
    float m = 0.0f;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(X + i);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int off = 32; off > 0; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        atomicMax(out_bits, __float_as_uint(m));  // non-negative floats order like their bits
}

// w[j] = dinv[j] / (sx sk)^2: the weights of the plain epilogue (accumulators in scaled units);
// *inv_scale = 1 / (sx sk): what the fp64 epilogue multiplies the accumulators with before the fit is added back
__global__ __launch_bounds__(256) void split_weights_kernel(int n, const float *dinv,
                                                            const unsigned *__restrict__ amax_bits, float sk, float *w,
                                                            double *inv_scale)
{
    const float sx = pow2_scale_below_one(__uint_as_float(*amax_bits));
    const float inv = 1.0f / (sx * sk);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        w[i] = dinv[i] * inv * inv;
    if (i == 0 && inv_scale)
        *inv_scale = (double)inv;
}

// In place: the 32 fp32 values of one k-block (128 bytes) become the 128 bytes [hi(32) | lo(32)] of the same
// block; one thread owns a whole block, so nothing another thread still has to read is overwritten.
__global__ __launch_bounds__(256) void split_pack_kernel(float *__restrict__ X, size_t nblocks,
                                                         const unsigned *__restrict__ amax_bits)
{
    const float s = pow2_scale_below_one(__uint_as_float(*amax_bits));
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (size_t)gridDim.x * 256) {
        float *blk = X + b * 32;
        float4 v4[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            v4[c] = *reinterpret_cast<const float4 *>(blk + 4 * c);
        half8 hi[4], lo[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v[8] = {v4[2 * c].x, v4[2 * c].y, v4[2 * c].z, v4[2 * c].w,
                                v4[2 * c + 1].x, v4[2 * c + 1].y, v4[2 * c + 1].z, v4[2 * c + 1].w};
            split8(v, s, hi[c], lo[c]);
        }
        half_t *dst = reinterpret_cast<half_t *>(blk);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<half8 *>(dst + 8 * c) = hi[c];
            *reinterpret_cast<half8 *>(dst + 32 + 8 * c) = lo[c];
        }
    }
}

// ---- Kqp tile straight into P16 --------------------------------------------------------------------------
// TC = float: plain kernel values from the centred fp32 points (queries centred by cen before rounding).
// TC = double (with a fit): k - (a_q + b_q s + c_q s^2) formed in fp64 from the fp64 points and rounded once, as in
// kqp_kernel (gpx_pairwise.hip).
template <typename TC, int KID>
__global__ __launch_bounds__(256) void kqp_split_kernel(Cov<TC> cov, float sk, int n, int npad,
                                                        const TC *__restrict__ px, const TC *__restrict__ py,
                                                        const TC *__restrict__ pz, const double *__restrict__ cen,
                                                        long nq_valid, const double *__restrict__ qx,
                                                        const double *__restrict__ qy,
                                                        const double *__restrict__ qz, half_t *__restrict__ P,
                                                        const double *__restrict__ fab, long ldcc, long ldk)
{
    __shared__ TC rx[TILE], ry[TILE], rz[TILE], rfa[TILE], rfb[TILE], rfc[TILE];
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.y * TILE;
    if (tid < TILE) {
        const long q = q0 + tid;
        const bool ok = q < nq_valid;
        const double c0 = sizeof(TC) == 4 ? cen[0] : 0.0, c1 = sizeof(TC) == 4 ? cen[1] : 0.0,
                     c2 = sizeof(TC) == 4 ? cen[2] : 0.0;
        rx[tid] = ok ? (TC)(qx[q] - c0) : TC(0);
        ry[tid] = ok ? (TC)(qy[q] - c1) : TC(0);
        rz[tid] = ok ? (TC)(qz[q] - c2) : TC(0);
        rfa[tid] = fab ? (TC)fab[q] : TC(0);  // per-query fit taken out of the kernel values (var_fit_kernel, gpx_pairwise.hip)
        rfb[tid] = fab ? (TC)fab[ldcc + q] : TC(0);
        rfc[tid] = fab ? (TC)fab[2 * ldcc + q] : TC(0);
    }
    const int tx = tid & 15, ty = tid >> 4;  // 16 lanes x 8 columns = 128 training points per row
    const int gj0 = blockIdx.x * TILE + tx * 8;
    TC cx[8], cy[8], cz[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        cx[c] = px[gj0 + c];
        cy[c] = py[gj0 + c];
        cz[c] = pz[gj0 + c];
    }
    __syncthreads();
#pragma unroll 2
    for (int r = 0; r < 8; ++r) {
        const int li = ty + 16 * r;
        const long q = q0 + li;
        const TC ax = rx[li], ay = ry[li], az = rz[li], fit_a = rfa[li], fit_b = rfb[li], fit_c = rfc[li];
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const TC dx = ax - cx[c], dy = ay - cy[c], dz = az - cz[c];
            const TC d2 = dx * dx + dy * dy + dz * dz;
            TC kv;
            if constexpr (sizeof(TC) == 8)
                kv = cov_k<TC, KID, MathFast>(cov, d2 + 1e-300);
            else
                kv = cov_k<TC, KID>(cov, d2);
            kv -= fit_a + d2 * (fit_b + fit_c * d2);
            v[c] = (q < nq_valid && gj0 + c < n) ? (float)kv : 0.0f;
        }
        half8 hi, lo;
        split8(v, sk, hi, lo);
        half_t *dst = P + (size_t)q * (2 * (size_t)ldk) + (gj0 / 32) * 64 + (gj0 % 32);  // ldk >= npad: row stride, fp32-equivalent elements
        *reinterpret_cast<half8 *>(dst) = hi;
        *reinterpret_cast<half8 *>(dst + 32) = lo;
    }
}

struct VsplitDev {
    const unsigned char *A;  // P16 X,   row stride 4 K bytes
    const unsigned char *B;  // P16 Kqp, row stride 4 K bytes
    int M, N, K;             // rows of X, queries (multiples of 128), K (multiple of 32)
    long ldkB;               // row stride of Kqp in fp32-equivalent elements (>= K; off the power of two: L2 sets)
    const float *w;          // per-row weight (1/D, scales folded in): plain epilogue
    float *partial;
    long ldp;
    // low-rank fit added back before squaring, in fp64 (colcoef != null): accumulators * inv_scale + sum_c colcoef rowcorr,
    // squared, times dinv64, summed into partial64
    const double *rowcorr, *colcoef, *dinv64, *inv_scale;
    double *partial64;
    long ldrc, ldcc;
};

// ---- the contraction -----------------------------------------------------------------------------------
// 128 x 128 tile per workgroup, 4 waves of 64 x 64 = 4 x 4 fragments of v_mfma_f32_16x16x32_f16 (one instruction per
// 32-deep k-tile and fragment pair; lane l feeds row (l & 15), k = 8 (l >> 4) .. + 8 -- the 8 consecutive k of one
// split8 group), two accumulator sets (hi*hi; hi*lo + lo*hi), two workgroups per CU.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): a k-tile (one 128-byte P16 block per row) goes from L2 straight into
// LDS, 1 KiB = 8 rows per wave-instruction, no staging registers and no ds_write_b128 -- which at 13 LDS-path cycles
// per wave-instruction (MI355X_MICROARCH.md, LDS table) cost the register-staged round-2 kernel 54 % of a CU's LDS time
// on top of the 33 % its fragment reads take (profiles/r04_pmc_vsplit.txt).  An LDS-DMA write is lane-linear, so rows
// cannot be padded against bank conflicts; instead the eight 16-byte chunks of a row are permuted: chunk c of row r sits
// at c ^ ((r >> 1) & 7), applied on the per-lane SOURCE address when writing and on the LDS address when reading.  The 16
// lanes of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper halves) then fall on
// 16 distinct 16-byte slots of the 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0).
//
// Schedule: two LDS buffers; at the top of k-tile kt the wait + barrier says that tile kt has landed and that everybody
// is done reading the other buffer; tile kt + 1 is issued into that one and has the whole compute phase (768 matrix-pipe
// cycles per wave, twice that with the CU's second workgroup) to arrive.  The kernel runs at the chip's power limit
// (1.7 GHz under it; section 4.3 of DESIGN.md): a software-pipelined form that keeps every LDS read a group ahead of its
// MFMAs raised the matrix-pipe duty from 68 to 72 % and the clock fell by the same factor.
//
// An LDS-DMA write is ordered for the other waves' ds_reads only by the issuing wave's vmcnt wait followed by a barrier;
// hipcc adds that wait to __syncthreads() for the DMAs it sees in straight-line code but not for those issued before a
// loop's back-edge, so it is spelled out.
#define VD_SYNC()                                        \
    {                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        __syncthreads();                                 \
    }
using f32x4v = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256, 2) void vsplit_gemm_kernel(VsplitDev g)
{
    constexpr int ROWB = 128;
    constexpr int TILE_B = TILE * ROWB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char vs_smem[];
    unsigned char *As = vs_smem;               // [2][TILE][ROWB]
    unsigned char *Bs = vs_smem + 2 * TILE_B;  // [2][TILE][ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nt = blockIdx.x;
    const int mt = (int)(gridDim.y - 1 - blockIdx.y);
    const int m0 = mt * TILE, n0 = nt * TILE;
    const int kt1 = min(g.K, m0 + TILE) / 32;
    const size_t ldb = (size_t)g.K * 4, ldbB = (size_t)g.ldkB * 4;

    f32x4v acc[4][4], cor[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][j][r] = 0.0f;
                cor[i][j][r] = 0.0f;
            }

    const int r_loc = lane >> 3;
    const int c_even = (lane & 7) ^ (r_loc >> 1);
    const unsigned char *a_src = g.A + (size_t)(m0 + 32 * wave + r_loc) * ldb;
    const unsigned char *b_src = g.B + (size_t)(n0 + 32 * wave + r_loc) * ldbB;
    using gptr_t = const __attribute__((address_space(1))) void *;
    using lptr_t = __attribute__((address_space(3))) void *;
    const unsigned lds_a = (unsigned)(uintptr_t)As + (unsigned)(32 * wave) * ROWB;
    const unsigned lds_b = (unsigned)(uintptr_t)Bs + (unsigned)(32 * wave) * ROWB;
#define VD_STAGE(BUF, KT)                                                                                              \
    {                                                                                                                  \
        const size_t ko_ = (size_t)(KT) * ROWB;                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
        {                                                                                                              \
            const int co_ = (c_even ^ (4 * (i & 1))) * 16;                                                             \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src + (size_t)(8 * i) * ldb + ko_ + co_),                     \
                                             (lptr_t)(uintptr_t)(lds_a + (BUF) * TILE_B + 8 * i * ROWB), 16, 0, 0);    \
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src + (size_t)(8 * i) * ldbB + ko_ + co_),                    \
                                             (lptr_t)(uintptr_t)(lds_b + (BUF) * TILE_B + 8 * i * ROWB), 16, 0, 0);    \
        }                                                                                                              \
    }

    // fragment addresses: lane l -> row (l & 15), logical chunk (l >> 4) (+ 4: the lo halves) at physical chunk
    // logical ^ ((row >> 1) & 7); fragment row offsets are multiples of 16
    const int sw = (lane >> 1) & 7;
    const int f_hi = (lane & 15) * ROWB + (((lane >> 4) ^ sw) * 16);
    const int f_lo = (lane & 15) * ROWB + ((((lane >> 4) + 4) ^ sw) * 16);
    const int a_frag = (wm * 64) * ROWB, b_frag = (wn * 64) * ROWB;

#define VD_COMPUTE(BUF)                                                                                      \
    {                                                                                                        \
        const unsigned char *as = As + (BUF) * TILE_B + a_frag;                                              \
        const unsigned char *bs = Bs + (BUF) * TILE_B + b_frag;                                              \
        half8 bh[4], bl[4];                                                                                  \
        _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                        \
        {                                                                                                    \
            bh[f] = *reinterpret_cast<const half8 *>(bs + f * 16 * ROWB + f_hi);                             \
            bl[f] = *reinterpret_cast<const half8 *>(bs + f * 16 * ROWB + f_lo);                             \
        }                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
        {                                                                                                    \
            const half8 ah = *reinterpret_cast<const half8 *>(as + i * 16 * ROWB + f_hi);                    \
            const half8 al = *reinterpret_cast<const half8 *>(as + i * 16 * ROWB + f_lo);                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
            {                                                                                                \
                cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], cor[i][j], 0, 0, 0);           \
                cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], cor[i][j], 0, 0, 0);           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[i][j], 0, 0, 0);           \
            }                                                                                                \
        }                                                                                                    \
    }

    VD_STAGE(0, 0);
    for (int kt = 0; kt < kt1; kt += 2) {  // kt1 is a multiple of 4: the buffer index stays a compile-time constant
        VD_SYNC();
        VD_STAGE(1, kt + 1);
        VD_COMPUTE(0);
        VD_SYNC();
        VD_STAGE(0, min(kt + 2, kt1 - 1));  // (past the end: the last tile once more, into the buffer nobody reads again)
        VD_COMPUTE(1);
    }
#undef VD_STAGE
#undef VD_COMPUTE
    VD_SYNC();

    // ---- epilogue; C/D layout of 16x16: col = lane & 15, row = 4 (lane >> 4) + r ----
    if (g.colcoef == nullptr) {
        float *red = reinterpret_cast<float *>(vs_smem);  // [2][TILE]
        float colsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wr = g.w[m0 + wm * 64 + i * 16 + 4 * (lane >> 4) + r];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float wv = acc[i][j][r] + cor[i][j][r] * (1.0f / 2048.0f);
                    colsum[j] += wv * wv * wr;
                }
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = colsum[j];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16)
                red[wm * TILE + wn * 64 + j * 16 + lane] = s;
        }
        VD_SYNC();
        if (tid < TILE)
            g.partial[(size_t)mt * g.ldp + n0 + tid] = red[tid] + red[TILE + tid];
        return;
    }
    double *ds = reinterpret_cast<double *>(vs_smem);
    double *rowc = ds;                       // [VAR_NCORR][TILE]
    double *colc = rowc + VAR_NCORR * TILE;  // [VAR_NCORR][TILE]
    double *roww = colc + VAR_NCORR * TILE;  // [TILE]
    double *red64 = roww + TILE;             // [2][TILE]
    const double inv = *g.inv_scale;
    for (int e = tid; e < 2 * VAR_NCORR * TILE + TILE; e += 256) {
        double v;
        if (e < VAR_NCORR * TILE) {
            v = g.rowcorr[(size_t)(e / TILE) * g.ldrc + m0 + e % TILE];
        } else if (e < 2 * VAR_NCORR * TILE) {
            const int e2 = e - VAR_NCORR * TILE;
            v = g.colcoef[(size_t)(e2 / TILE) * g.ldcc + n0 + e2 % TILE];
        } else {
            v = g.dinv64[m0 + e - 2 * VAR_NCORR * TILE];
        }
        ds[e] = v;
    }
    VD_SYNC();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lcol = wn * 64 + j * 16 + (lane & 15);
        double ca[VAR_NCORR];
#pragma unroll
        for (int c = 0; c < VAR_NCORR; ++c)
            ca[c] = colc[c * TILE + lcol];
        double sj = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = wm * 64 + i * 16 + 4 * (lane >> 4) + r;
                double w = ((double)acc[i][j][r] + (double)cor[i][j][r] * (1.0 / 2048.0)) * inv;
#pragma unroll
                for (int c = 0; c < VAR_NCORR; ++c)
                    w = fma(ca[c], rowc[c * TILE + lrow], w);
                sj = fma(w * w, roww[lrow], sj);
            }
        sj += __shfl_xor(sj, 16);
        sj += __shfl_xor(sj, 32);
        if (lane < 16)
            red64[wm * TILE + lcol] = sj;
    }
    VD_SYNC();
    if (tid < TILE)
        g.partial64[(size_t)mt * g.ldp + n0 + tid] = red64[tid] + red64[TILE + tid];
}

// ---- launchers ---------------------------------------------------------------------------------------------
void launch_split_prepare(float *X, int np, float *dinv_to_w, float sk, unsigned *amax_bits, hipStream_t st,
                          double *inv_scale)
{
    // X (fp32) -> P16 in place; dinv -> w = dinv / (sx sk)^2 in place
    (void)hipMemsetAsync(amax_bits, 0, sizeof(unsigned), st);
    const size_t n = (size_t)np * np;
    hipLaunchKernelGGL(split_absmax_kernel, dim3(2048), dim3(256), 0, st, X, n, amax_bits);
    hipLaunchKernelGGL(split_pack_kernel, dim3(4096), dim3(256), 0, st, X, n / 32, amax_bits);
    hipLaunchKernelGGL(split_weights_kernel, dim3((np + 255) / 256), dim3(256), 0, st, np, dinv_to_w, amax_bits, sk,
                       dinv_to_w, inv_scale);
}

void launch_kqp_split(bool compute64, const CovHost &h, float sk, int n, int npad, const void *px, const void *py, const void *pz,
                      const double *px64, const double *py64, const double *pz64, const double *cen, long nq_valid,
                      long nq_tile, const double *qx, const double *qy, const double *qz, void *P, hipStream_t st,
                      const double *fab, long ldcc, long ldk)
{
    if (ldk <= 0)
        ldk = npad;
    dim3 grid(npad / TILE, (unsigned)(nq_tile / TILE));
    if (compute64) {
        Cov<double> c = lower_cov<double>(h);
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((kqp_split_kernel<double, KID>), grid, dim3(256), 0, st, c, sk, n, npad,
                                                  px64, py64, pz64, cen, nq_valid, qx, qy, qz, (half_t *)P, fab, ldcc, ldk));
    } else {
        Cov<float> c = lower_cov<float>(h);
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((kqp_split_kernel<float, KID>), grid, dim3(256), 0, st, c, sk, n, npad,
                                                  (const float *)px, (const float *)py, (const float *)pz, cen, nq_valid,
                                                  qx, qy, qz, (half_t *)P, fab, ldcc, ldk));
    }
}

void launch_vsplit_gemm(const void *Xp, const void *Kp, int np, int nq_tile, const float *w, void *partial,
                        long ldp, hipStream_t st, int m_rows, const double *rowcorr, long ldrc,
                        const double *colcoef, long ldcc, const double *dinv64, const double *inv_scale, long ldk)
{
    VsplitDev g;
    const bool corr = rowcorr && colcoef && dinv64 && inv_scale;
    g.rowcorr = corr ? rowcorr : nullptr, g.colcoef = corr ? colcoef : nullptr;
    g.dinv64 = dinv64, g.inv_scale = inv_scale;
    g.ldrc = ldrc, g.ldcc = ldcc;
    g.A = (const unsigned char *)Xp;
    g.B = (const unsigned char *)Kp;
    g.M = np, g.N = nq_tile, g.K = np;
    g.ldkB = ldk > 0 ? ldk : np;
    g.w = w;
    g.partial = (float *)partial, g.partial64 = (double *)partial, g.ldp = ldp;
    constexpr size_t shmem = 4 * (size_t)TILE * 128;  // 2 buffers x 2 operands x 128 rows x 128 bytes
    static_assert(sizeof(double) * (2 * VAR_NCORR * TILE + 3 * TILE) <= shmem, "the fp64 epilogue reuses the staging buffers");
    static PerDeviceOnce attr_once;  // per device, see gpx_internal.hpp
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vsplit_gemm_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    });
    dim3 grid(nq_tile / TILE, (m_rows > 0 ? m_rows : np) / TILE);  // rows in the identity padding contribute nothing
    hipLaunchKernelGGL(vsplit_gemm_kernel, grid, dim3(256), shmem, st, g);
}

}  // namespace gpx
