// Tail of a SpectraNetBlock in ONE forward kernel (gfx950, split-bf16 arithmetic):
//     pooled = MaxPool1d(4)( Conv1d_1x1( GELU( LayerNorm(ycat) ) ) )        src/applecider/models/spectranet.py:31-40
// ycat [R, K] are the concatenated conv outputs (R = B * L rows, K = 3 * Cout channels), w [N, K] the 1x1 taps.
// Unfused, the three kernels move ycat (read) + z (written, read) + the 1x1 output (written, read) + the pooled rows:
// 6.0 GB at stage 1 of the default configuration (R = 2 097 152, K = 192, N = 64); here ycat is read from HBM once
// (statistics), again from L2 (the K loop), z is written once because the weight gradient of the 1x1 conv needs it,
// and only the pooled rows + argmax bytes leave: 3.4 GB.
//
// Structure = the 128 x 128 x 32 split-bf16 product of ac_gemm.hip (four bf16 images per stage: A_hi, A_lo, B_hi,
// B_lo; three MFMAs per fragment pair; two LDS stages, loads two K tiles ahead) with
//   * a prologue: the workgroup's 128 rows' mean / rstd (16 lanes per row, two passes over the row: the second one
//     hits L1 / L2), kept in LDS and written out for the backward pass;
//   * a transform where the A registers are split into planes: z = gelu((x - mean) * rstd * gamma + beta);
//   * an epilogue: the 128 x 128 accumulator tile (+ bias) goes through LDS, four consecutive rows are reduced to
//     their maximum (first index on ties, NaN wins: torch's rule) and only the pooled row and its argmax are stored.
#include "ac_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32;
constexpr int IMG = 128 * 32;                       // bf16 elements of one operand image
constexpr int STAGE = 4 * IMG;                      // A_hi, A_lo, B_hi, B_lo
constexpr int KMAX = 1536;                          // widest concatenated row (stage 4 of the default configuration)
constexpr int LDS_BYTES = 2 * STAGE * 2 + 2 * BM * 4 + 2 * KMAX * 4;   // two stages + mean / rstd of the 128 rows + gamma, beta

struct TailParams {
    const float *ycat;
    int64_t ld;
    const float *gamma, *beta, *w, *bias;
    float *z, *mean, *rstd, *pooled;
    uint8_t *idx;
    int R, K, N;
    float eps;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void split4(const f32x4 &v, s16x4 &hi, s16x4 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned short h = ac_f2h(v[j]);
        hi[j] = (short)h;
        lo[j] = (short)ac_f2h(v[j] - ac_h2f(h));
    }
}
// [128 rows][32 k] bf16 image: 16-byte chunks (8 k) swizzled by (row >> 2) & 3; thread chunk c (4 floats) = half a chunk
__device__ __forceinline__ int kc_off(int r, int c) { return r * 32 + (((c >> 1) ^ ((r >> 2) & 3)) << 3) + (c & 1) * 4; }
__device__ __forceinline__ bf16x8 frag_kc(const unsigned short *img, int rowbase, int s, int lane) {
    const int local = rowbase + (lane & 31), lh = lane >> 5;
    const int chunk16 = (2 * s + lh) ^ ((local >> 2) & 3);
    return *(const bf16x8 *)(img + local * 32 + chunk16 * 8);
}
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// NB = 32-column blocks per wave: 2 -> 128-column tiles, 1 -> 64-column tiles (stage 1: N = 64)
template <int NB>
__global__ __launch_bounds__(256, 2) void ln_gelu_pw_pool_fwd_kernel(TailParams p) {
    constexpr int BN = 64 * NB, NG = 2 * NB;      // tile columns, 32-row groups of w per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *sm16 = reinterpret_cast<unsigned short *>(smem);
    float *s_mean = smem + (2 * STAGE * 2) / 4, *s_rstd = s_mean + BM, *s_gamma = s_rstd + BM, *s_beta = s_gamma + KMAX;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;
    const int row0 = tm * BM, K = p.K;

    for (int k = 4 * t; k < K; k += 4 * 256) {
        *(f32x4 *)(s_gamma + k) = ac_gload<f32x4>(p.gamma + k);
        *(f32x4 *)(s_beta + k) = ac_gload<f32x4>(p.beta + k);
    }
    // ---- row statistics: 16 lanes per row, 4 rows per wave and sweep
    {
        const int l16 = lane & 15, sub = lane >> 4;
        const float invK = 1.0f / (float)K;
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int lr = wave * 32 + it * 4 + sub;
            const float *xr = p.ycat + (int64_t)(row0 + lr) * p.ld;
            float s = 0.f;
            for (int c = l16 * 4; c < K; c += 64) {
                const f32x4 v = ac_gload<f32x4>(xr + c);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
            const float mean = sum16(s) * invK;
            float qs = 0.f;
            for (int c = l16 * 4; c < K; c += 64) {
                const f32x4 v = ac_gload<f32x4>(xr + c);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                qs += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            const float rstd = rsqrtf(sum16(qs) * invK + p.eps);
            if (l16 == 0) {
                s_mean[lr] = mean;
                s_rstd[lr] = rstd;
                if (tn == 0) {
                    p.mean[row0 + lr] = mean;
                    p.rstd[row0 + lr] = rstd;
                }
            }
        }
    }
    __syncthreads();

    // ---- K loop.  This thread's rows are rbase + 32 i, its k chunk c of every tile.  N % 32 == 0: a group of 32 rows
    // of w is inside the matrix or outside it as a whole (bvalid: uniform over the workgroup).
    const int c = t & 7, rbase = t >> 3;
    const float *a0 = p.ycat + (int64_t)(row0 + rbase) * p.ld + 4 * c;
    const float *b0p = p.w + (int64_t)(tn * BN + rbase) * K + 4 * c;
    const int64_t astep = 32 * p.ld, bstep = (int64_t)32 * K;
    const int64_t zdelta = p.z ? p.z - p.ycat : 0;           // ld == K: z has the layout of ycat
    const bool write_z = p.z != nullptr && tn == 0;
    const int bgroups = (p.N - tn * BN + 31) / 32;           // groups of 32 rows of w this tile holds (1 .. NG)
    const int nkt = K / BK;

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Regs {
        f32x4 a[4], b[NG];
    };
    auto load = [&](int kt, Regs &r) {
        const int k = (kt < nkt ? kt : nkt - 1) * BK;       // clamped: a tile past the end is loaded, never stored
#pragma unroll
        for (int i = 0; i < 4; ++i) r.a[i] = ac_gload<f32x4>(a0 + i * astep + k);
#pragma unroll
        for (int i = 0; i < NG; ++i) r.b[i] = ac_gload<f32x4>(b0p + (i < bgroups ? i : 0) * bstep + k);
    };
    auto store = [&](int kt, unsigned short *stage, const Regs &r) {
        unsigned short *ah = stage, *al = ah + IMG, *bh = ah + 2 * IMG, *bl = ah + 3 * IMG;
        const f32x4 g = *(const f32x4 *)(s_gamma + kt * BK + 4 * c), bt = *(const f32x4 *)(s_beta + kt * BK + 4 * c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mu = s_mean[rbase + 32 * i], rs = s_rstd[rbase + 32 * i];
            f32x4 zv;
#pragma unroll
            for (int e = 0; e < 4; ++e) zv[e] = ac_gelu_fast((r.a[i][e] - mu) * rs * g[e] + bt[e]);
            if (write_z) *(f32x4 *)(const_cast<float *>(a0) + i * astep + kt * BK + zdelta) = zv;
            s16x4 hi, lo;
            split4(zv, hi, lo);
            const int off = kc_off(rbase + 32 * i, c);
            *(s16x4 *)(ah + off) = hi;
            *(s16x4 *)(al + off) = lo;
            if (i < NG) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                split4(i < bgroups ? r.b[i < NG ? i : 0] : zero, hi, lo);
                *(s16x4 *)(bh + off) = hi;
                *(s16x4 *)(bl + off) = lo;
            }
        }
    };
    auto compute = [&](const unsigned short *stage) {
        const unsigned short *ah = stage, *al = ah + IMG, *bh = ah + 2 * IMG, *bl = ah + 3 * IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fah[2], fal[2], fbh[NB], fbl[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fah[i] = frag_kc(ah, wm * 64 + 32 * i, s, lane);
                fal[i] = frag_kc(al, wm * 64 + 32 * i, s, lane);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                fbh[j] = frag_kc(bh, wn * 32 * NB + 32 * j, s, lane);
                fbl[j] = frag_kc(bl, wn * 32 * NB + 32 * j, s, lane);
            }
            // cross terms first, the leading term last
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = AC_MFMA16(fal[i], fbh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = AC_MFMA16(fah[i], fbl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = AC_MFMA16(fah[i], fbh[j], acc[i][j]);
        }
    };

    unsigned short *S0 = sm16, *S1 = sm16 + STAGE;
    Regs r0, r1;
    load(0, r0);
    store(0, S0, r0);
    __syncthreads();
    load(1, r0);
    for (int kt = 0; kt < nkt; kt += 2) {
        load(kt + 2, r1);
        __builtin_amdgcn_sched_barrier(0);
        compute(S0);
        if (kt + 1 < nkt) store(kt + 1, S1, r0);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        load(kt + 3, r0);
        __builtin_amdgcn_sched_barrier(0);
        compute(S1);
        if (kt + 2 < nkt) store(kt + 2, S0, r1);
        __syncthreads();
    }

    // ---- epilogue: tile (+ bias) -> LDS -> max over groups of four rows
    float *tile = smem;                                     // 128 x BN fp32 over the stages (all reads done)
    {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = wn * 32 * NB + 32 * j + li, n = tn * BN + col;
            const float bj = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    tile[(wm * 64 + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * BN + col] = acc[sa][j][e] + bj;
        }
    }
    __syncthreads();
    const int prow0 = row0 >> 2;
    for (int i = t; i < (BM / 4) * BN; i += 256) {
        const int pr = i / BN, col = i - pr * BN, n = tn * BN + col;
        if (n >= p.N) continue;
        float m = tile[(4 * pr) * BN + col];
        unsigned am = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const float v = tile[(4 * pr + j) * BN + col];
            if (v > m || (v != v && m == m)) {
                m = v;
                am = j;
            }
        }
        p.pooled[(int64_t)(prow0 + pr) * p.N + n] = m;
        p.idx[(int64_t)(prow0 + pr) * p.N + n] = (uint8_t)am;
    }
}

}  // namespace

extern "C" int ac_ln_gelu_pw_pool_fwd(const float *ycat, int64_t ld, const float *gamma, const float *beta, float eps,
                                      const float *w, const float *bias, float *z, float *mean, float *rstd,
                                      float *pooled, uint8_t *idx, int64_t rows, int32_t K, int32_t N,
                                      ac_stream_t stream) {
    if (!ycat || !gamma || !beta || !w || !mean || !rstd || !pooled || !idx || rows <= 0 || K <= 0 || N <= 0)
        return AC_EINVAL;
    if ((rows % BM) || (K % BK) || K < 2 * BK || K > KMAX || (N % 32) || ld != K ||
        rows / BM * ((N + 63) / 64) > 0x7FFFFFFF)
        return AC_EINVAL;
    if ( !ac_aligned16(ycat) || !ac_aligned16(gamma) || !ac_aligned16(beta) || !ac_aligned16(w) ||
        (z && !ac_aligned16(z)))
        return AC_EALIGN;
    TailParams p;
    p.ycat = ycat; p.ld = ld; p.gamma = gamma; p.beta = beta; p.w = w; p.bias = bias;
    p.z = z; p.mean = mean; p.rstd = rstd; p.pooled = pooled; p.idx = idx;
    p.R = (int)rows; p.K = K; p.N = N; p.eps = eps;
    const int bn = N <= 64 ? 64 : 128;
    p.tiles_m = (int)(rows / BM); p.tiles_n = (N + bn - 1) / bn;
    static const hipError_t attr1 = hipFuncSetAttribute((const void *)ln_gelu_pw_pool_fwd_kernel<1>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    static const hipError_t attr2 = hipFuncSetAttribute((const void *)ln_gelu_pw_pool_fwd_kernel<2>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr1 != hipSuccess) return -(int)attr1 - 2000;
    if (attr2 != hipSuccess) return -(int)attr2 - 2000;
    if (bn == 64)
        hipLaunchKernelGGL(ln_gelu_pw_pool_fwd_kernel<1>, dim3(p.tiles_m * p.tiles_n), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(ln_gelu_pw_pool_fwd_kernel<2>, dim3(p.tiles_m * p.tiles_n), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
