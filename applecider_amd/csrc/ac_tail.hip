// Tail of a pooled SpectraNetBlock (src/applecider/models/spectranet.py:31-40), gfx950, split-bf16 arithmetic:
//     pooled = MaxPool1d(4)( Conv1d_1x1( GELU( LayerNorm_C(ycat) ) ) )
// ycat [rows, K] are the concatenated conv outputs (rows = B * L, K = 3 * Cout channels), w [N, K] the 1x1 taps,
// N = Cout.  Unfused, the forward pass moved ycat (read) + z = gelu(LN(ycat)) (written, read) + the 1x1 output
// (written, read) and the backward pass un-pooled into a full-size tensor, read z and that tensor for the 1x1 conv's
// weight gradient, wrote and re-read d z, and re-read ycat in LayerNorm's backward: 15.8 GB per step at stage 1 of the
// default configuration (rows = 2 097 152, K = 192, N = 64).  Here NOTHING of row length K is written in the forward
// pass and nothing but the (hi, lo) operand planes of d ycat in the backward pass; z and d(1x1 output) only ever exist
// as bf16 (hi, lo) planes of 32 rows in LDS:
//   tail_fwd     ycat -> statistics -> z planes -> z . w^T (+ bias) -> max over 4 rows: pooled, argmax, mean, rstd
//   tail_bwd_dx  (ycat, mean, rstd, d pooled, argmax) -> d z = scatter(d pooled) . w -> GELU', LayerNorm backward ->
//                (hi, lo) planes of d ycat (zero-padded layout of the conv bank's gradient products), d gamma, d beta,
//                column sums of d ycat (the conv biases' gradient)
//   tail_bwd_dw  (ycat, mean, rstd, d pooled, argmax) -> z planes again -> d w += scatter(d pooled)^T . z
// 6.9 GB at stage 1.  Every product is three bf16 MFMAs on (hi, lo) halves (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32
// accumulate), as everywhere in this mode.
//
// One workgroup owns blocks of R = 32 whole rows and walks several of them (next block's rows are loaded into
// registers while this block is in the matrix cores); a row is spread over TPR adjacent lanes, so statistics and
// LayerNorm's row sums are sub-wave shuffles.  Supported shapes: (K, N) = (192, 64) and (384, 128) - stages 1 and 2
// hold 75 % of the tails' bytes; stages 3 and 4 are matrix-core-bound at these row counts and keep their products.
#include "ac_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int R = 32;                                   // rows per block = one 32-row MFMA tile

__device__ __forceinline__ void split4(const f32x4 &v, s16x4 &hi, s16x4 &lo) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, l0, h1, l1;
    ac_split_pair(v[0], v[1], h0, l0);   // 3 VALU instructions per element (ac_common.h)
    ac_split_pair(v[2], v[3], h1, l1);
    const u32x2 h = {h0, h1}, l = {l0, l1};
    hi = __builtin_bit_cast(s16x4, h);
    lo = __builtin_bit_cast(s16x4, l);
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {   // sum over G adjacent lanes (G = 8, 16, 32)
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// fragment of a [rows][k] plane, row-major ("k contiguous"): lane -> row base + (lane & 31), k = 16 s + 8 (lane >> 5) ..+7
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_kc(const unsigned short *plane, int rowbase, int s, int lane) {
    return *(const bf16x8 *)(plane + (rowbase + (lane & 31)) * PITCH + 16 * s + 8 * (lane >> 5));
}
// fragment of a [k][cols] plane (the reduction index is the ROW of the plane): hardware transpose read,
// lane -> col base + (lane & 31), k = 16 s + 8 (lane >> 5) ..+7
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_rc(const unsigned short *plane, int colbase, int s, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int k0 = 16 * s + 8 * (g >> 1);
    const unsigned short *a0 = plane + (k0 + q) * PITCH + colbase + 16 * (g & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(a0 + 4 * PITCH));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// three-MFMA product of split operands: cross terms first, the leading term last
__device__ __forceinline__ f32x16 mma3(const bf16x8 &ah, const bf16x8 &al, const bf16x8 &bh, const bf16x8 &bl, f32x16 acc) {
    acc = AC_MFMA16(al, bh, acc);
    acc = AC_MFMA16(ah, bl, acc);
    acc = AC_MFMA16(ah, bh, acc);
    return acc;
}

struct TailParams {
    const float *ycat, *gamma, *beta, *bias;
    const unsigned short *w_hi, *w_lo;      // fwd: planes of w [N][K]; bwd_dx: planes of w^T [K][N]
    float *mean, *rstd, *pooled;            // fwd outputs (bwd: mean / rstd are inputs)
    uint8_t *idx;
    const float *dpooled;
    unsigned short *dx_hi, *dx_lo;
    float *dgamma, *dbeta, *dxsum, *dw;
    int64_t rows;
    int nblocks, seg_len, seg_pitch, seg_off;
    float eps;
};

// ---- the rows of one block in registers: thread (r = t / TPR, sub = t % TPR) holds float4s at columns 4 (sub + TPR j)
template <int K, int TPR>
struct RowRegs {
    static constexpr int NV = K / (4 * TPR);
    f32x4 v[NV];
    __device__ __forceinline__ void load(const float *ycat, int64_t row, int sub) {
        const float *p = ycat + row * K + 4 * sub;
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = ac_gload<f32x4>(p + 4 * TPR * j);
    }
};

// z = gelu((x - mu) rs gamma + beta) of this thread's elements -> (hi, lo) planes [R][ZP] in LDS
template <int K, int TPR, int ZP>
__device__ __forceinline__ void write_z_planes(const RowRegs<K, TPR> &x, float mu, float rs, const float *s_gamma,
                                               const float *s_beta, unsigned short *zhi, unsigned short *zlo, int r,
                                               int sub) {
#pragma unroll
    for (int j = 0; j < RowRegs<K, TPR>::NV; ++j) {
        const int c = 4 * (sub + TPR * j);
        const f32x4 g = *(const f32x4 *)(s_gamma + c), b = *(const f32x4 *)(s_beta + c);
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = ac_gelu_fast((x.v[j][e] - mu) * rs * g[e] + b[e]);
        s16x4 hi, lo;
        split4(z, hi, lo);
        *(s16x4 *)(zhi + r * ZP + c) = hi;
        *(s16x4 *)(zlo + r * ZP + c) = lo;
    }
}

// d(1x1 output) of one block = the pooled gradient scattered to the arg-max row of each window, as (hi, lo) planes
// [R][DP].  Work item (pooled row pr < 8, four channels): at most one per thread (8 N / 4 <= NT); its global loads are
// issued one block ahead (DoutRegs::load), the LDS stores when the block's turn comes (store).
template <int N, int DP>
struct DoutRegs {
    f32x4 g;
    unsigned a;
    __device__ __forceinline__ void load(const float *dpooled, const uint8_t *idx, int64_t prow0, int t) {
        if (t < (R / 4) * (N / 4)) {
            const int pr = t / (N / 4), n = 4 * (t - pr * (N / 4));
            g = ac_gload<f32x4>(dpooled + (prow0 + pr) * N + n);
            a = ac_gload<unsigned>((const unsigned *)(idx + (prow0 + pr) * N + n));
        }
    }
    __device__ __forceinline__ void store(unsigned short *dhi, unsigned short *dlo, int t) const {
        if (t < (R / 4) * (N / 4)) {
            const int pr = t / (N / 4), n = 4 * (t - pr * (N / 4));
            s16x4 hi, lo;
            split4(g, hi, lo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool on = ((a >> (8 * e)) & 0xFFu) == (unsigned)j;
                    h[e] = on ? hi[e] : (short)0;
                    l[e] = on ? lo[e] : (short)0;
                }
                *(s16x4 *)(dhi + (4 * pr + j) * DP + n) = h;
                *(s16x4 *)(dlo + (4 * pr + j) * DP + n) = l;
            }
        }
    }
};

// =====================================================================================================================
// forward: NT = 64 * 2 * (N / 32) threads; wave w -> output tile (all 32 rows) x (32 channels nt = w % MT), half
// kh = w / MT of the reduction; its w fragments stay in registers for the whole launch
// =====================================================================================================================
template <int K, int N>
__global__ __launch_bounds__(128 * (N / 32)) void tail_fwd_kernel(TailParams p) {
    constexpr int MT = N / 32, NT = 128 * MT, TPR = NT / R, ZP = K + 8, KS = K / 32;
    static_assert(K % (4 * TPR) == 0 && (ZP * 2) % 16 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short *zhi = (unsigned short *)smem_raw, *zlo = zhi + R * ZP;
    float *red = (float *)(zlo + R * ZP);                     // [MT][16][64]
    float *s_gamma = red + MT * 16 * 64, *s_beta = s_gamma + K;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, lh = lane >> 5;
    const int r = t / TPR, sub = t % TPR, nt = w % MT, kh = w / MT;

    for (int c = 4 * t; c < K; c += 4 * NT) {
        *(f32x4 *)(s_gamma + c) = ac_gload<f32x4>(p.gamma + c);
        *(f32x4 *)(s_beta + c) = ac_gload<f32x4>(p.beta + c);
    }
    bf16x8 wh[KS], wl[KS];
    {
        const int64_t off = (int64_t)(nt * 32 + (lane & 31)) * K + 16 * (kh * KS) + 8 * lh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            wh[s] = ac_gload<bf16x8>((const short *)p.w_hi + off + 16 * s);
            wl[s] = ac_gload<bf16x8>((const short *)p.w_lo + off + 16 * s);
        }
    }
    const float bias = (kh == 0 && p.bias) ? p.bias[nt * 32 + (lane & 31)] : 0.f;
    const float invK = 1.0f / (float)K;
    RowRegs<K, TPR> x;
    int blk = blockIdx.x;
    if (blk < p.nblocks) x.load(p.ycat, (int64_t)blk * R + r, sub);
    __syncthreads();
    for (; blk < p.nblocks; blk += gridDim.x) {
        const int64_t row0 = (int64_t)blk * R;
        // ---- statistics of this thread's row (two passes over the registers), z planes
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < x.NV; ++j) s += (x.v[j][0] + x.v[j][1]) + (x.v[j][2] + x.v[j][3]);
        const float mu = group_sum<TPR>(s) * invK;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < x.NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x.v[j][e] - mu;
                q += d * d;
            }
        const float rs = rsqrtf(group_sum<TPR>(q) * invK + p.eps);
        if (sub == 0) {
            p.mean[row0 + r] = mu;
            p.rstd[row0 + r] = rs;
        }
        write_z_planes<K, TPR, ZP>(x, mu, rs, s_gamma, s_beta, zhi, zlo, r, sub);
        if (blk + (int)gridDim.x < p.nblocks) x.load(p.ycat, (int64_t)(blk + gridDim.x) * R + r, sub);
        __syncthreads();
        // ---- z . w^T over this wave's half of the channels
        f32x16 acc = zero16();
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            const bf16x8 ah = frag_kc<ZP>(zhi, 0, kh * KS + s2, lane), al = frag_kc<ZP>(zlo, 0, kh * KS + s2, lane);
            acc = mma3(ah, al, wh[s2], wl[s2], acc);
        }
        if (kh == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(nt * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (kh == 0) {
            // rows of acc[e]: 8 (e >> 2) + 4 lh + (e & 3): each group of four e is one pooling window
            const int64_t prow0 = row0 >> 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float m = acc[4 * g] + red[(nt * 16 + 4 * g) * 64 + lane] + bias;
                unsigned am = 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const float v = acc[4 * g + j] + red[(nt * 16 + 4 * g + j) * 64 + lane] + bias;
                    if (v > m || (v != v && m == m)) {      // first maximum wins, NaN propagates (torch's rule)
                        m = v;
                        am = j;
                    }
                }
                const int64_t o = (prow0 + 2 * g + lh) * N + nt * 32 + (lane & 31);
                p.pooled[o] = m;
                p.idx[o] = (uint8_t)am;
            }
        }
    }
}

// =====================================================================================================================
// backward, weight gradient of the 1x1 conv: d w [N][K] += d out^T [N][rows] . z [rows][K].  NT as in the forward
// kernel; wave w -> rows mt = w % MT of d w (32 output channels), columns half = w / MT (K / 64 tiles of 32)
// =====================================================================================================================
template <int K, int N>
__global__ __launch_bounds__(128 * (N / 32)) void tail_bwd_dw_kernel(TailParams p) {
    constexpr int MT = N / 32, NT = 128 * MT, TPR = NT / R, ZP = K + 8, DP = N + 8, CT = K / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short *zhi = (unsigned short *)smem_raw, *zlo = zhi + R * ZP, *dhi = zlo + R * ZP, *dlo = dhi + R * DP;
    float *s_gamma = (float *)(dlo + R * DP), *s_beta = s_gamma + K;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, lh = lane >> 5;
    const int r = t / TPR, sub = t % TPR, mt = w % MT, half = w / MT;

    for (int c = 4 * t; c < K; c += 4 * NT) {
        *(f32x4 *)(s_gamma + c) = ac_gload<f32x4>(p.gamma + c);
        *(f32x4 *)(s_beta + c) = ac_gload<f32x4>(p.beta + c);
    }
    f32x16 acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = zero16();
    static_assert((R / 4) * (N / 4) <= NT, "one d out work item per thread");
    RowRegs<K, TPR> x;
    DoutRegs<N, DP> dreg;
    float mu = 0.f, rs = 0.f;
    int blk = blockIdx.x;
    if (blk < p.nblocks) {
        x.load(p.ycat, (int64_t)blk * R + r, sub);
        dreg.load(p.dpooled, p.idx, ((int64_t)blk * R) >> 2, t);
        mu = p.mean[(int64_t)blk * R + r];
        rs = p.rstd[(int64_t)blk * R + r];
    }
    __syncthreads();
    for (; blk < p.nblocks; blk += gridDim.x) {
        write_z_planes<K, TPR, ZP>(x, mu, rs, s_gamma, s_beta, zhi, zlo, r, sub);
        dreg.store(dhi, dlo, t);
        if (blk + (int)gridDim.x < p.nblocks) {
            const int64_t nrow0 = (int64_t)(blk + gridDim.x) * R;
            x.load(p.ycat, nrow0 + r, sub);
            dreg.load(p.dpooled, p.idx, nrow0 >> 2, t);
            mu = p.mean[nrow0 + r];
            rs = p.rstd[nrow0 + r];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < R / 16; ++s) {
            const bf16x8 ah = frag_rc<DP>(dhi, mt * 32, s, lane), al = frag_rc<DP>(dlo, mt * 32, s, lane);
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const bf16x8 bh = frag_rc<ZP>(zhi, (half * CT + j) * 32, s, lane);
                const bf16x8 bl = frag_rc<ZP>(zlo, (half * CT + j) * 32, s, lane);
                acc[j] = mma3(ah, al, bh, bl, acc[j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = mt * 32 + 8 * (e >> 2) + 4 * lh + (e & 3), kk = (half * CT + j) * 32 + (lane & 31);
            atomicAdd(p.dw + (int64_t)n * K + kk, acc[j][e]);
        }
}

// =====================================================================================================================
// backward, gradient of the conv outputs: 512 threads (16 per row: K / 64 float4 per thread); wave w owns the
// 32-column tiles w, w + 8, ... of d z = d out [R][N] . w [N][K]; then every thread finishes LayerNorm's backward for its
// elements
// =====================================================================================================================
template <int K, int N>
__global__ __launch_bounds__(512, 2) void tail_bwd_dx_kernel(TailParams p) {
    constexpr int TPR = 16, NT = TPR * R, NW = NT / 64, DP = N + 8, ZF = K + 4, KS = N / 16, NV = K / (4 * TPR);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *dz = (float *)smem_raw;                                   // [R][ZF] fp32
    float *sacc = dz + R * ZF;                                       // [3][K]
    float *s_gamma = sacc + 3 * K, *s_beta = s_gamma + K;
    unsigned short *dhi = (unsigned short *)(s_beta + K), *dlo = dhi + R * DP;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, lh = lane >> 5;
    const int r = t / TPR, sub = t % TPR;

    for (int c = 4 * t; c < K; c += 4 * NT) {
        *(f32x4 *)(s_gamma + c) = ac_gload<f32x4>(p.gamma + c);
        *(f32x4 *)(s_beta + c) = ac_gload<f32x4>(p.beta + c);
    }
    for (int c = t; c < 3 * K; c += NT) sacc[c] = 0.f;
    f32x4 adg[NV], adb[NV], adx[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) adg[j] = adb[j] = adx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float invK = 1.0f / (float)K;
    static_assert((R / 4) * (N / 4) <= NT, "one d out work item per thread");
    constexpr bool WRES = K / 32 <= NW && KS <= 4;       // (192, 64): 32 registers of w^T per wave
    bf16x8 wrh[WRES ? KS : 1], wrl[WRES ? KS : 1];
    if constexpr (WRES) {
        const int64_t boff = (int64_t)((w < K / 32 ? w : 0) * 32 + (lane & 31)) * N + 8 * lh;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            wrh[s2] = ac_gload<bf16x8>((const short *)p.w_hi + boff + 16 * s2);
            wrl[s2] = ac_gload<bf16x8>((const short *)p.w_lo + boff + 16 * s2);
        }
    }
    RowRegs<K, TPR> x;
    DoutRegs<N, DP> dreg;
    float mu_n = 0.f, rs_n = 0.f;
    int blk = blockIdx.x;
    if (blk < p.nblocks) {
        x.load(p.ycat, (int64_t)blk * R + r, sub);
        dreg.load(p.dpooled, p.idx, ((int64_t)blk * R) >> 2, t);
        mu_n = p.mean[(int64_t)blk * R + r];
        rs_n = p.rstd[(int64_t)blk * R + r];
    }
    __syncthreads();
    for (; blk < p.nblocks; blk += gridDim.x) {
        const int64_t row0 = (int64_t)blk * R;
        dreg.store(dhi, dlo, t);
        const float mu = mu_n, rs = rs_n;
        const bool more = blk + (int)gridDim.x < p.nblocks;
        const int64_t nrow0 = (int64_t)(blk + gridDim.x) * R;
        if (more) {
            dreg.load(p.dpooled, p.idx, nrow0 >> 2, t);
            mu_n = p.mean[nrow0 + r];
            rs_n = p.rstd[nrow0 + r];
        }
        __syncthreads();
        if constexpr (WRES) {
            // one tile per wave (K / 32 <= NW): its fragments of w^T never leave the registers
            if (w < K / 32) {
                f32x16 acc = zero16();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 ah = frag_kc<DP>(dhi, 0, s, lane), al = frag_kc<DP>(dlo, 0, s, lane);
                    acc = mma3(ah, al, wrh[s], wrl[s], acc);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) dz[(8 * (e >> 2) + 4 * lh + (e & 3)) * ZF + w * 32 + (lane & 31)] = acc[e];
            }
        } else {
            for (int tile = w; tile < K / 32; tile += NW) {
                // d z tile: reduction over the N output channels; B = planes of w^T [K][N] straight from L2
                f32x16 acc = zero16();
                const int64_t boff = (int64_t)(tile * 32 + (lane & 31)) * N + 8 * lh;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 bh = ac_gload<bf16x8>((const short *)p.w_hi + boff + 16 * s);
                    const bf16x8 bl = ac_gload<bf16x8>((const short *)p.w_lo + boff + 16 * s);
                    const bf16x8 ah = frag_kc<DP>(dhi, 0, s, lane), al = frag_kc<DP>(dlo, 0, s, lane);
                    acc = mma3(ah, al, bh, bl, acc);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) dz[(8 * (e >> 2) + 4 * lh + (e & 3)) * ZF + tile * 32 + (lane & 31)] = acc[e];
            }
        }
        __syncthreads();
        // ---- LayerNorm backward of this thread's elements (ac_rows.hip layernorm_bwd_sub_kernel, with d y = d z)
        f32x4 d[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * (sub + TPR * j);
            const f32x4 g = *(const f32x4 *)(s_gamma + c), b = *(const f32x4 *)(s_beta + c);
            const f32x4 dzv = *(const f32x4 *)(dz + r * ZF + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float hh = (x.v[j][e] - mu) * rs;
                const float dd = dzv[e] * ac_gelu_grad_fast(hh * g[e] + b[e]);
                d[j][e] = dd * g[e];
                s1 += d[j][e];
                s2 += d[j][e] * hh;
                adg[j][e] += dd * hh;
                adb[j][e] += dd;
            }
        }
        const float c1 = group_sum<TPR>(s1) * invK, c2 = group_sum<TPR>(s2) * invK;
        const int64_t gr = row0 + r;
        const int64_t rr = p.seg_len ? (gr / p.seg_len) * p.seg_pitch + p.seg_off + gr % p.seg_len : gr;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * (sub + TPR * j);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = rs * (d[j][e] - c1 - (x.v[j][e] - mu) * rs * c2);
                adx[j][e] += o[e];
            }
            s16x4 hi, lo;
            split4(o, hi, lo);
            *(s16x4 *)(p.dx_hi + rr * K + c) = hi;
            *(s16x4 *)(p.dx_lo + rr * K + c) = lo;
        }
        if (more) x.load(p.ycat, nrow0 + r, sub);
    }
    // ---- column sums: the 64 / TPR rows of a wave first (shuffles), then LDS, then one atomic per column and workgroup
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = TPR; o < 64; o <<= 1) {
                adg[j][e] += __shfl_xor(adg[j][e], o, 64);
                adb[j][e] += __shfl_xor(adb[j][e], o, 64);
                adx[j][e] += __shfl_xor(adx[j][e], o, 64);
            }
    __syncthreads();
    if (lane < TPR) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * (sub + TPR * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&sacc[c + e], adg[j][e]);
                atomicAdd(&sacc[K + c + e], adb[j][e]);
                atomicAdd(&sacc[2 * K + c + e], adx[j][e]);
            }
        }
    }
    __syncthreads();
    for (int c = t; c < K; c += NT) {
        if (p.dgamma) atomicAdd(&p.dgamma[c], sacc[c]);
        if (p.dbeta) atomicAdd(&p.dbeta[c], sacc[K + c]);
        if (p.dxsum) atomicAdd(&p.dxsum[c], sacc[2 * K + c]);
    }
}

template <int K, int N>
constexpr size_t fwd_lds() { return (size_t)2 * R * (K + 8) * 2 + (size_t)(N / 32) * 16 * 64 * 4 + (size_t)2 * K * 4; }
template <int K, int N>
constexpr size_t dw_lds() { return (size_t)2 * R * (K + 8) * 2 + (size_t)2 * R * (N + 8) * 2 + (size_t)2 * K * 4; }
template <int K, int N>
constexpr size_t dx_lds() { return (size_t)R * (K + 4) * 4 + (size_t)5 * K * 4 + (size_t)2 * R * (N + 8) * 2; }

int grid_for(int nblocks, int per_cu) {
    // persistent workgroups: a multiple of the 8 XCDs, ~per_cu per CU, every workgroup walks >= 2 blocks when it can
    int g = 256 * per_cu;
    if (g > nblocks) g = nblocks;
    return g < 1 ? 1 : g;
}

template <auto KERNEL>
int launch(int grid, int threads, size_t lds, const TailParams &p, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(threads), lds, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

bool shape_ok(int64_t rows, int K, int N) {
    return rows > 0 && (rows % R) == 0 && rows / R <= 0x7FFFFFFF && ((K == 192 && N == 64) || (K == 384 && N == 128));
}

}  // namespace

extern "C" int ac_spectail_supported(int64_t rows, int32_t K, int32_t N) { return shape_ok(rows, K, N) ? 1 : 0; }

extern "C" int ac_spectail_fwd(const float *ycat, const float *gamma, const float *beta, float eps, const void *w_hi,
                               const void *w_lo, const float *bias, float *mean, float *rstd, float *pooled,
                               uint8_t *idx, int64_t rows, int32_t K, int32_t N, ac_stream_t stream) {
    if (!ycat || !gamma || !beta || !w_hi || !w_lo || !mean || !rstd || !pooled || !idx || !shape_ok(rows, K, N))
        return AC_EINVAL;
    if (!ac_aligned16(ycat) || !ac_aligned16(gamma) || !ac_aligned16(beta) || !ac_aligned16(w_hi) || !ac_aligned16(w_lo))
        return AC_EALIGN;
    TailParams p = {};
    p.ycat = ycat; p.gamma = gamma; p.beta = beta; p.bias = bias;
    p.w_hi = (const unsigned short *)w_hi; p.w_lo = (const unsigned short *)w_lo;
    p.mean = mean; p.rstd = rstd; p.pooled = pooled; p.idx = idx;
    p.rows = rows; p.nblocks = (int)(rows / R); p.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    if (K == 192) return launch<tail_fwd_kernel<192, 64>>(grid_for(p.nblocks, 4), 256, fwd_lds<192, 64>(), p, st);
    return launch<tail_fwd_kernel<384, 128>>(grid_for(p.nblocks, 1), 512, fwd_lds<384, 128>(), p, st);
}

extern "C" int ac_spectail_bwd_dx(const float *ycat, const float *mean, const float *rstd, const float *gamma,
                                  const float *beta, const float *dpooled, const uint8_t *idx, const void *wt_hi,
                                  const void *wt_lo, void *dx_hi, void *dx_lo, int32_t seg_len, int32_t seg_pitch,
                                  int32_t seg_off, float *dgamma, float *dbeta, float *dxsum, int64_t rows, int32_t K,
                                  int32_t N, ac_stream_t stream) {
    if (!ycat || !mean || !rstd || !gamma || !beta || !dpooled || !idx || !wt_hi || !wt_lo || !dx_hi || !dx_lo ||
        !shape_ok(rows, K, N))
        return AC_EINVAL;
    if (seg_len < 0 || (seg_len > 0 && ((seg_len % R) || seg_pitch < seg_len + seg_off || seg_off < 0 || (rows % seg_len))))
        return AC_EINVAL;
    if (!ac_aligned16(ycat) || !ac_aligned16(gamma) || !ac_aligned16(beta) || !ac_aligned16(wt_hi) ||
        !ac_aligned16(wt_lo) || !ac_aligned16(dpooled) || ((uintptr_t)idx & 3u) || ((uintptr_t)dx_hi & 7u) ||
        ((uintptr_t)dx_lo & 7u))
        return AC_EALIGN;
    TailParams p = {};
    p.ycat = ycat; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float *>(mean); p.rstd = const_cast<float *>(rstd);
    p.w_hi = (const unsigned short *)wt_hi; p.w_lo = (const unsigned short *)wt_lo;
    p.dpooled = dpooled; p.idx = const_cast<uint8_t *>(idx);
    p.dx_hi = (unsigned short *)dx_hi; p.dx_lo = (unsigned short *)dx_lo;
    p.dgamma = dgamma; p.dbeta = dbeta; p.dxsum = dxsum;
    p.rows = rows; p.nblocks = (int)(rows / R); p.seg_len = seg_len; p.seg_pitch = seg_pitch; p.seg_off = seg_off;
    hipStream_t st = (hipStream_t)stream;
    if (K == 192) return launch<tail_bwd_dx_kernel<192, 64>>(grid_for(p.nblocks, 2), 512, dx_lds<192, 64>(), p, st);
    return launch<tail_bwd_dx_kernel<384, 128>>(grid_for(p.nblocks, 1), 512, dx_lds<384, 128>(), p, st);
}

extern "C" int ac_spectail_bwd_dw(const float *ycat, const float *mean, const float *rstd, const float *gamma,
                                  const float *beta, const float *dpooled, const uint8_t *idx, float *dw, int64_t rows,
                                  int32_t K, int32_t N, ac_stream_t stream) {
    if (!ycat || !mean || !rstd || !gamma || !beta || !dpooled || !idx || !dw || !shape_ok(rows, K, N)) return AC_EINVAL;
    if (!ac_aligned16(ycat) || !ac_aligned16(gamma) || !ac_aligned16(beta) || !ac_aligned16(dpooled) || ((uintptr_t)idx & 3u))
        return AC_EALIGN;
    TailParams p = {};
    p.ycat = ycat; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float *>(mean); p.rstd = const_cast<float *>(rstd);
    p.dpooled = dpooled; p.idx = const_cast<uint8_t *>(idx); p.dw = dw;
    p.rows = rows; p.nblocks = (int)(rows / R);
    hipStream_t st = (hipStream_t)stream;
    if (K == 192) return launch<tail_bwd_dw_kernel<192, 64>>(grid_for(p.nblocks, 2), 256, dw_lds<192, 64>(), p, st);
    return launch<tail_bwd_dw_kernel<384, 128>>(grid_for(p.nblocks, 1), 512, dw_lds<384, 128>(), p, st);
}
