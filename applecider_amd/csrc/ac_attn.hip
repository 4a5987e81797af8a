// Key-padding-masked multi-head attention on the matrix cores (gfx950), forward and backward:
// the core of nn.TransformerEncoderLayer as HyraxBaselineCLS uses it (HyraxBaselineCLS.py:24-31,
// 73-79: d_model 128, 8 heads -> d_head 16, T = L + 1 <= 258 tokens, src_key_padding_mask, dropout on
// the attention weights in training).
//
// One 512-thread workgroup per (sample, head).  K and V (backward: also Q*scale and dO) of that head
// sit in LDS as [token][16] bf16 images (32-byte rows, the two 16-byte halves of a row swapped on
// every second group of 8 rows so that the ds_read_b128 row reads are bank-conflict free); a wave
// owns 32-token blocks.  All contractions are v_mfma_f32_32x32x16_bf16 tiles:
//   scores      S^T[key, query] = K_tile . Q^T        (contraction over d_head = 16: one MFMA)
//   forward     O[query, d]     = P . V               (P straight from the S^T accumulators: a
//               32x32 accumulator tile with its column on the lane IS the next MFMA's A operand,
//               cdna_hip_programming.md section 3; V fragments by ds_read_b64_tr_b16)
//   backward    dQ^T = K^T . dS^T ;  dV^T = dO^T . P~ ;  dK^T = (Q*scale)^T . dS
// so no T x T matrix ever leaves the registers and nothing is transposed through LDS.  The softmax
// is two passes over the key blocks (log-sum-exp first, then P = exp(S - lse)): recomputing a score
// tile costs one MFMA, which is cheaper than carrying and rescaling an output accumulator.
// Backward is two phases (queries on the lanes for dQ, keys on the lanes for dK / dV), each
// recomputing P from the saved log-sum-exp; D_i = sum_d dO_id O_id is formed while staging.
// (Measured alternative, dropped: ONE phase with the keys on the lanes, dS crossing a per-wave LDS
// transpose for dQ = dS . K and the partial dQ tiles summed with ds_add_f32 — half the exp / dropout work
// per head, but 447 us per launch against 325 us for the two-phase form at B = 512, T = 129.)
//
// SPLIT = false: bf16 operands (math mode 'bf16').  SPLIT = true: every operand is split into
// hi + lo bf16 halves and every product is 3 MFMAs (math mode 'bf16x3', ~2^-16 per product); the
// exact-fp32 mode keeps the scalar kernels of ac_seq.hip.
// Dropout uses the same counter hash and index ((b*H + h)*T + i)*T + j as those kernels.
#include "ac_common.h"
#include <hip/hip_bf16.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_TMAX = 288;   // padded tokens per head (T <= 288)
// Threads per workgroup.  Backward: 8 waves — its ~200 registers per lane allow one workgroup per CU
// either way, and T = 129 pads to 5 token blocks that 4 waves would share 2/1/1/1 (442 -> 325 us per
// launch).  Forward: 4 waves — at ~140 registers three 4-wave workgroups fit a CU, which beats one
// 8-wave workgroup (131 vs 171 us).
constexpr int ATT_NT_FWD = 256, ATT_NT_BWD = 512;

__device__ __forceinline__ unsigned short a_f2bf(float x) {
    return ac_f2h(x);
}
__device__ __forceinline__ float a_bf2f(unsigned short h) {
    return ac_h2f(h);
}

// element offset of (row, 4-element chunk c) in a [rows][16] bf16 image
__device__ __forceinline__ int img_off(int row, int c) {
    return row * 16 + 8 * ((c >> 1) ^ ((row >> 3) & 1)) + 4 * (c & 1);
}

template <bool SPLIT>
__device__ __forceinline__ void img_store4(unsigned short *hi_img, unsigned short *lo_img, int row, int c,
                                           const f32x4 &v) {
    s16x4 h, l;
    if constexpr (SPLIT) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        unsigned h0, l0, h1, l1;
        ac_split_pair(v[0], v[1], h0, l0);   // 3 VALU instructions per element (ac_common.h)
        ac_split_pair(v[2], v[3], h1, l1);
        const u32x2 hh = {h0, h1}, ll = {l0, l1};
        h = __builtin_bit_cast(s16x4, hh);
        l = __builtin_bit_cast(s16x4, ll);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (short)a_f2bf(v[j]);
    }
    const int off = img_off(row, c);
    *(s16x4 *)(hi_img + off) = h;
    if (SPLIT) *(s16x4 *)(lo_img + off) = l;
}

// MFMA operand "row of the image": 8 consecutive d (half lh) of row `row`
__device__ __forceinline__ bf16x8 row_frag(const unsigned short *img, int row, int lh) {
    return *(const bf16x8 *)(img + row * 16 + 8 * (lh ^ ((row >> 3) & 1)));
}

// MFMA operand "column of the image" for a contraction over tokens in the k order of an accumulator
// tile used as the other operand: element j of lane half h = token rbase + 8*(j>>2) + (j&3) with
// rbase = block*32 + 16*s + 4*h; the lane receives column d = lane & 15 (columns 16..31 of the MFMA
// operand are duplicates and feed output rows / columns nobody reads).  EXEC must be all ones.
__device__ __forceinline__ bf16x8 col_frag(const unsigned short *img, int rbase, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = rbase + q, r1 = r0 + 8;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_v4 *)(img + r0 * 16 + 8 * ((p >> 1) ^ ((r0 >> 3) & 1)) + 4 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_v4 *)(img + r1 * 16 + 8 * ((p >> 1) ^ ((r1 >> 3) & 1)) + 4 * (p & 1)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <bool SPLIT>
__device__ __forceinline__ void cvt8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
    if constexpr (SPLIT) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ac_split_pair(v[2 * j], v[2 * j + 1], h[j], l[j]);
        const u32x4 hh = {h[0], h[1], h[2], h[3]}, ll = {l[0], l[1], l[2], l[3]};
        hi = __builtin_bit_cast(bf16x8, hh);
        lo = __builtin_bit_cast(bf16x8, ll);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) hi[j] = (short)a_f2bf(v[j]);
    }
}

// acc += a * b with (hi, lo) halves: cross terms first, leading term last
template <bool SPLIT>
__device__ __forceinline__ f32x16 mma(const bf16x8 &ah, const bf16x8 &al, const bf16x8 &bh, const bf16x8 &bl,
                                      f32x16 acc) {
    if (SPLIT) {
        acc = AC_MFMA16(al, bh, acc);
        acc = AC_MFMA16(ah, bl, acc);
    }
    return AC_MFMA16(ah, bh, acc);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// blocks that share an XCD (bid % 8) walk consecutive (sample, head) pairs: the 8 heads of a sample
// read interleaved 64-byte pieces of the same qkv rows, so their second halves of every 128-byte line
// come from that XCD's L2 instead of the fabric
__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <bool SPLIT, bool DROP>
__global__ __launch_bounds__(ATT_NT_FWD) void mha_fwd_mfma_kernel(const float *__restrict__ qkv,
                                                           const uint8_t *__restrict__ pad,
                                                           float *__restrict__ out, float *__restrict__ lse,
                                                           int T, int H, float p_drop, uint64_t seed,
                                                           const uint64_t *stepp) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smh[];
    if (DROP) seed = ac_step_seed(seed, stepp);
    const int Tp = (T + 31) & ~31, NB = Tp >> 5;
    unsigned short *Kh = smh, *Vh = Kh + Tp * 16;
    unsigned short *Kl = Vh + Tp * 16, *Vl = Kl + (SPLIT ? Tp * 16 : 0);
    uint8_t *vm8 = (uint8_t *)(smh + (SPLIT ? 4 : 2) * Tp * 16);
    unsigned *wk_s = (unsigned *)(vm8 + Tp);                 // dropout: per-key words (ac_att_word)
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int b = wg / H, h = wg % H, D = H * 16;
    const float *base = qkv + (int64_t)b * T * 3 * D + h * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    const uint64_t tb = ((uint64_t)b * H + h) * T;
    const unsigned thr = ac_att_threshold(p_drop);

    for (int i = t; i < Tp * 4; i += ATT_NT_FWD) {
        const int tok = i >> 2, c = i & 3;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
        if (tok < T) {
            kv = ac_gload<f32x4>(base + (int64_t)tok * 3 * D + D + 4 * c);
            vv = ac_gload<f32x4>(base + (int64_t)tok * 3 * D + 2 * D + 4 * c);
        }
        img_store4<SPLIT>(Kh, Kl, tok, c, kv);
        img_store4<SPLIT>(Vh, Vl, tok, c, vv);
    }
    for (int i = t; i < Tp; i += ATT_NT_FWD) {
        vm8[i] = (i < T && !(pad && pad[(int64_t)b * T + i])) ? 1 : 0;
        if (DROP) wk_s[i] = ac_att_word(seed, tb + i, 1);
    }
    __syncthreads();

    const float inv_keep = 1.0f / (1.0f - p_drop);
    for (int qb = wave; qb < NB; qb += ATT_NT_FWD / 64) {
        const int q = qb * 32 + li;
        float qv[8];
        {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
            if (q < T) {
                a = ac_gload<f32x4>(base + (int64_t)q * 3 * D + 8 * lh);
                c = ac_gload<f32x4>(base + (int64_t)q * 3 * D + 8 * lh + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qv[j] = a[j] * (0.25f * 1.4426950408889634f);      // log2(e) / sqrt(d_head): scores in the log2 domain,
                qv[4 + j] = c[j] * (0.25f * 1.4426950408889634f);  // so that every exponential is a bare v_exp_f32
            }
        }
        bf16x8 qh, ql;
        cvt8<SPLIT>(qv, qh, ql);

        auto scores = [&](int kb) {
            const int krow = kb * 32 + li;
            const bf16x8 kh = row_frag(Kh, krow, lh);
            bf16x8 kl = kh;
            if (SPLIT) kl = row_frag(Kl, krow, lh);
            return mma<SPLIT>(kh, kl, qh, ql, zero16());   // S^T[key row, query lane]
        };

        // ---- pass 1: log-sum-exp of the query's row (this lane sees 16 of every 32 keys)
        float m = -INFINITY, l = 0.f;
        for (int kb = 0; kb < NB; ++kb) {
            const f32x16 S = scores(kb);
            float sv[16];
            float bm = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kb * 32 + 8 * g + 4 * lh;
                if (kb * 32 + 8 * g < T) {   // wave-uniform: key groups past T cost nothing
                    const unsigned w = *(const unsigned *)(vm8 + key0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sv[4 * g + e] = ((w >> (8 * e)) & 0xFFu) ? S[4 * g + e] : -INFINITY;
                        bm = fmaxf(bm, sv[4 * g + e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[4 * g + e] = -INFINITY;
                }
            }
            const float mn = fmaxf(m, bm);
            const float mref = (mn == -INFINITY) ? 0.f : mn;
            float acc = l * exp2f(m - mref);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (kb * 32 + 8 * g < T) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += exp2f(sv[4 * g + e] - mref);
                }
            l = acc;
            m = mn;
        }
        {
            const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
            const float mn = fmaxf(m, mo);
            const float mref = (mn == -INFINITY) ? 0.f : mn;
            l = l * exp2f(m - mref) + lo * exp2f(mo - mref);
            m = mref;
        }
        const float lse_q = m + __log2f(l);                                   // log2-sum-exp2 of the scaled scores
        if (lh == 0 && q < T) lse[((int64_t)b * H + h) * T + q] = lse_q * 0.6931471805599453f;   // natural units

        // ---- pass 2: O = dropout(exp(S - lse)) . V
        f32x16 O = zero16();
        const unsigned wq = DROP ? ac_att_word(seed, tb + q, 0) : 0u;
        for (int kb = 0; kb < NB; ++kb) {
            const f32x16 S = scores(kb);
            float pv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kb * 32 + 8 * g + 4 * lh;
                if (kb * 32 + 8 * g < T) {
                    const unsigned w = *(const unsigned *)(vm8 + key0);
                    uint4 wk4 = {0u, 0u, 0u, 0u};
                    if (DROP) wk4 = *(const uint4 *)(wk_s + key0);
                    const unsigned wk[4] = {wk4.x, wk4.y, wk4.z, wk4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float p = ((w >> (8 * e)) & 0xFFu) ? exp2f(S[4 * g + e] - lse_q) : 0.f;
                        if (DROP) p = ac_att_keep(wq, wk[e], thr) ? p * inv_keep : 0.f;
                        pv[4 * g + e] = p;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[4 * g + e] = 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float ps[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ps[j] = pv[8 * s + j];
                bf16x8 ph, pl;
                cvt8<SPLIT>(ps, ph, pl);
                const int rb = kb * 32 + 16 * s + 4 * lh;
                const bf16x8 vh = col_frag(Vh, rb, lane);
                bf16x8 vl = vh;
                if (SPLIT) vl = col_frag(Vl, rb, lane);
                O = mma<SPLIT>(ph, pl, vh, vl, O);   // P (A operand: X^T . B) times V: O[query row, d lane]
            }
        }
        if (li < 16) {
            float *ob = out + (int64_t)b * T * D + h * 16 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qq = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (qq < T) ob[(int64_t)qq * D] = O[e];
            }
        }
    }
}

template <bool SPLIT, bool DROP>
__global__ __launch_bounds__(ATT_NT_BWD) void mha_bwd_mfma_kernel(
    const float *__restrict__ dout, const float *__restrict__ qkv, const uint8_t *__restrict__ pad,
    const float *__restrict__ out, const float *__restrict__ lse, float *__restrict__ dqkv, int T, int H,
    float p_drop, uint64_t seed, const uint64_t *stepp) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smh[];
    if (DROP) seed = ac_step_seed(seed, stepp);
    const int Tp = (T + 31) & ~31, NB = Tp >> 5, IMG = Tp * 16;
    unsigned short *Qh = smh, *Kh = Qh + IMG, *Vh = Kh + IMG, *Gh = Vh + IMG;   // G = dO
    unsigned short *Ql = Gh + IMG, *Kl = Ql + (SPLIT ? IMG : 0), *Vl = Kl + (SPLIT ? IMG : 0),
                   *Gl = Vl + (SPLIT ? IMG : 0);
    float *lse_s = (float *)(smh + (SPLIT ? 8 : 4) * IMG);
    float *D_s = lse_s + Tp;
    uint8_t *vm8 = (uint8_t *)(D_s + Tp);
    unsigned *wq_s = (unsigned *)(vm8 + Tp), *wk_s = wq_s + Tp;   // dropout: per-query / per-key words (ac_att_word)
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int b = wg / H, h = wg % H, D = H * 16;
    const float *base = qkv + (int64_t)b * T * 3 * D + h * 16;
    const float *gbase = dout + (int64_t)b * T * D + h * 16;
    const float *obase = out + (int64_t)b * T * D + h * 16;
    float *dbase = dqkv + (int64_t)b * T * 3 * D + h * 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
    const uint64_t bh = (uint64_t)b * H + h;

    for (int i = t; i < Tp * 4; i += ATT_NT_BWD) {   // Tp*4 is a multiple of 128: whole waves stay together
        const int tok = i >> 2, c = i & 3;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f}, kv = qv, vv = qv, gv = qv, ov = qv;
        if (tok < T) {
            qv = ac_gload<f32x4>(base + (int64_t)tok * 3 * D + 4 * c) * 0.25f;
            kv = ac_gload<f32x4>(base + (int64_t)tok * 3 * D + D + 4 * c);
            vv = ac_gload<f32x4>(base + (int64_t)tok * 3 * D + 2 * D + 4 * c);
            gv = ac_gload<f32x4>(gbase + (int64_t)tok * D + 4 * c);
            ov = ac_gload<f32x4>(obase + (int64_t)tok * D + 4 * c);
        }
        img_store4<SPLIT>(Qh, Ql, tok, c, qv);
        img_store4<SPLIT>(Kh, Kl, tok, c, kv);
        img_store4<SPLIT>(Vh, Vl, tok, c, vv);
        img_store4<SPLIT>(Gh, Gl, tok, c, gv);
        float dpart = gv[0] * ov[0] + gv[1] * ov[1] + gv[2] * ov[2] + gv[3] * ov[3];
        dpart += __shfl_xor(dpart, 1, 64);
        dpart += __shfl_xor(dpart, 2, 64);
        if (c == 0) D_s[tok] = dpart;
    }
    for (int i = t; i < Tp; i += ATT_NT_BWD) {
        vm8[i] = (i < T && !(pad && pad[(int64_t)b * T + i])) ? 1 : 0;
        lse_s[i] = i < T ? lse[bh * T + i] : 1e30f;    // rows past T: exp(S - 1e30) = 0
        if (DROP) {
            wq_s[i] = ac_att_word(seed, bh * T + i, 0);
            wk_s[i] = ac_att_word(seed, bh * T + i, 1);
        }
    }
    __syncthreads();
    const float inv_keep = 1.0f / (1.0f - p_drop);
    const unsigned thr = ac_att_threshold(p_drop);

    // ---- phase A: queries on the lanes -> dQ^T[d, query] = K^T . dS^T
    for (int qb = wave; qb < NB; qb += ATT_NT_BWD / 64) {
        const int q = qb * 32 + li;
        const bf16x8 qh = row_frag(Qh, q, lh), gh = row_frag(Gh, q, lh);
        bf16x8 ql = qh, gl = gh;
        if (SPLIT) {
            ql = row_frag(Ql, q, lh);
            gl = row_frag(Gl, q, lh);
        }
        const float lse_q = lse_s[q], D_q = D_s[q];
        const unsigned wq = DROP ? wq_s[q] : 0u;
        f32x16 dQ = zero16();
        for (int kb = 0; kb < NB; ++kb) {
            const int krow = kb * 32 + li;
            const bf16x8 kh = row_frag(Kh, krow, lh), vh = row_frag(Vh, krow, lh);
            bf16x8 kl = kh, vl = vh;
            if (SPLIT) {
                kl = row_frag(Kl, krow, lh);
                vl = row_frag(Vl, krow, lh);
            }
            const f32x16 S = mma<SPLIT>(kh, kl, qh, ql, zero16());    // S^T[key, query]
            const f32x16 dP = mma<SPLIT>(vh, vl, gh, gl, zero16());   // dP~^T[key, query]
            float ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kb * 32 + 8 * g + 4 * lh;
                if (kb * 32 + 8 * g < T) {
                    const unsigned w = *(const unsigned *)(vm8 + key0);
                    uint4 wk4 = {0u, 0u, 0u, 0u};
                    if (DROP) wk4 = *(const uint4 *)(wk_s + key0);
                    const unsigned wk[4] = {wk4.x, wk4.y, wk4.z, wk4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = ((w >> (8 * e)) & 0xFFu) ? __expf(S[4 * g + e] - lse_q) : 0.f;
                        float ks = 1.f;
                        if (DROP) ks = ac_att_keep(wq, wk[e], thr) ? inv_keep : 0.f;
                        ds[4 * g + e] = p * (ks * dP[4 * g + e] - D_q);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds[4 * g + e] = 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = ds[8 * s + j];
                bf16x8 xh, xl;
                cvt8<SPLIT>(x, xh, xl);
                const int rb = kb * 32 + 16 * s + 4 * lh;
                const bf16x8 kth = col_frag(Kh, rb, lane);
                bf16x8 ktl = kth;
                if (SPLIT) ktl = col_frag(Kl, rb, lane);
                dQ = mma<SPLIT>(kth, ktl, xh, xl, dQ);   // A . X: rows d, lanes query
            }
        }
        if (q < T) {
            float *dq = dbase + (int64_t)q * 3 * D + 4 * lh;
            f32x4 a = {dQ[0], dQ[1], dQ[2], dQ[3]}, c = {dQ[4], dQ[5], dQ[6], dQ[7]};
            *(f32x4 *)dq = a * 0.25f;
            *(f32x4 *)(dq + 8) = c * 0.25f;
        }
    }

    // ---- phase B: keys on the lanes -> dV^T[d, key] = dO^T . P~ ; dK^T[d, key] = (Q scale)^T . dS
    for (int kb = wave; kb < NB; kb += ATT_NT_BWD / 64) {
        const int key = kb * 32 + li;
        const bf16x8 kh = row_frag(Kh, key, lh), vh = row_frag(Vh, key, lh);
        bf16x8 kl = kh, vl = vh;
        if (SPLIT) {
            kl = row_frag(Kl, key, lh);
            vl = row_frag(Vl, key, lh);
        }
        const bool kvalid = vm8[key] != 0;
        const unsigned wkk = DROP ? wk_s[key] : 0u;
        f32x16 dV = zero16(), dK = zero16();
        for (int qb = 0; qb < NB; ++qb) {
            const int qrow = qb * 32 + li;
            const bf16x8 qh = row_frag(Qh, qrow, lh), gh = row_frag(Gh, qrow, lh);
            bf16x8 ql = qh, gl = gh;
            if (SPLIT) {
                ql = row_frag(Ql, qrow, lh);
                gl = row_frag(Gl, qrow, lh);
            }
            const f32x16 S = mma<SPLIT>(qh, ql, kh, kl, zero16());    // S[query, key]
            const f32x16 dP = mma<SPLIT>(gh, gl, vh, vl, zero16());   // dP~[query, key]
            float pt[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qb * 32 + 8 * g + 4 * lh;
                if (qb * 32 + 8 * g < T) {
                    const f32x4 l4 = *(const f32x4 *)(lse_s + q0), d4 = *(const f32x4 *)(D_s + q0);
                    uint4 wq4 = {0u, 0u, 0u, 0u};
                    if (DROP) wq4 = *(const uint4 *)(wq_s + q0);
                    const unsigned wqv[4] = {wq4.x, wq4.y, wq4.z, wq4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = kvalid ? __expf(S[4 * g + e] - l4[e]) : 0.f;
                        float ks = 1.f;
                        if (DROP) ks = ac_att_keep(wqv[e], wkk, thr) ? inv_keep : 0.f;
                        pt[4 * g + e] = p * ks;
                        ds[4 * g + e] = p * (ks * dP[4 * g + e] - d4[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pt[4 * g + e] = ds[4 * g + e] = 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float x[8], y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[j] = pt[8 * s + j];
                    y[j] = ds[8 * s + j];
                }
                bf16x8 xh, xl, yh, yl;
                cvt8<SPLIT>(x, xh, xl);
                cvt8<SPLIT>(y, yh, yl);
                const int rb = qb * 32 + 16 * s + 4 * lh;
                const bf16x8 gth = col_frag(Gh, rb, lane), qth = col_frag(Qh, rb, lane);
                bf16x8 gtl = gth, qtl = qth;
                if (SPLIT) {
                    gtl = col_frag(Gl, rb, lane);
                    qtl = col_frag(Ql, rb, lane);
                }
                dV = mma<SPLIT>(gth, gtl, xh, xl, dV);
                dK = mma<SPLIT>(qth, qtl, yh, yl, dK);
            }
        }
        if (key < T) {
            float *dk = dbase + (int64_t)key * 3 * D + D + 4 * lh;
            float *dv = dbase + (int64_t)key * 3 * D + 2 * D + 4 * lh;
            f32x4 a = {dK[0], dK[1], dK[2], dK[3]}, c = {dK[4], dK[5], dK[6], dK[7]};
            *(f32x4 *)dk = a;
            *(f32x4 *)(dk + 8) = c;
            f32x4 e = {dV[0], dV[1], dV[2], dV[3]}, f = {dV[4], dV[5], dV[6], dV[7]};
            *(f32x4 *)dv = e;
            *(f32x4 *)(dv + 8) = f;
        }
    }
}

template <bool SPLIT, bool DROP>
int launch_fwd(const float *qkv, const uint8_t *pad, float *out, float *lse, int B, int T, int H, float p,
               uint64_t seed, const uint64_t *step, hipStream_t st) {
    const int Tp = (T + 31) & ~31;
    const size_t lds = (size_t)(SPLIT ? 4 : 2) * Tp * 32 + Tp + (size_t)Tp * sizeof(unsigned);   // + the key words
    static const hipError_t attr = hipFuncSetAttribute((const void *)mha_fwd_mfma_kernel<SPLIT, DROP>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    if (attr != hipSuccess) return -(int)attr - 2000;
    hipLaunchKernelGGL((mha_fwd_mfma_kernel<SPLIT, DROP>), dim3(B * H), dim3(ATT_NT_FWD), lds, st, qkv, pad, out, lse, T,
                       H, p, seed, step);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

template <bool SPLIT, bool DROP>
int launch_bwd(const float *dout, const float *qkv, const uint8_t *pad, const float *out, const float *lse,
               float *dqkv, int B, int T, int H, float p, uint64_t seed, const uint64_t *step,
               hipStream_t st) {
    const int Tp = (T + 31) & ~31;
    const size_t lds = (size_t)(SPLIT ? 8 : 4) * Tp * 32 + 2 * Tp * sizeof(float) + Tp + (size_t)2 * Tp * sizeof(unsigned);
    static const hipError_t attr = hipFuncSetAttribute((const void *)mha_bwd_mfma_kernel<SPLIT, DROP>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    if (attr != hipSuccess) return -(int)attr - 2000;
    hipLaunchKernelGGL((mha_bwd_mfma_kernel<SPLIT, DROP>), dim3(B * H), dim3(ATT_NT_BWD), lds, st, dout, qkv, pad, out,
                       lse, dqkv, T, H, p, seed, step);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

}  // namespace

extern "C" int ac_mha_fwd_mfma(const float *qkv, const uint8_t *pad, float *out, float *lse, int32_t B,
                               int32_t T, int32_t H, int32_t Dh, float p_drop, uint64_t seed, const uint64_t *step, int32_t split,
                               ac_stream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || T <= 0 || H <= 0) return AC_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f) return AC_EINVAL;
    if (Dh != 16 || T > ATT_TMAX || ((H * 16) % 4)) return AC_EINVAL;   // other shapes: ac_mha_fwd
    if (!ac_aligned16(qkv)) return AC_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const bool drop = p_drop > 0.f;
    if (split) return drop ? launch_fwd<true, true>(qkv, pad, out, lse, B, T, H, p_drop, seed, step, st)
                           : launch_fwd<true, false>(qkv, pad, out, lse, B, T, H, p_drop, seed, step, st);
    return drop ? launch_fwd<false, true>(qkv, pad, out, lse, B, T, H, p_drop, seed, step, st)
                : launch_fwd<false, false>(qkv, pad, out, lse, B, T, H, p_drop, seed, step, st);
}

extern "C" int ac_mha_bwd_mfma(const float *dout, const float *qkv, const uint8_t *pad, const float *out,
                               const float *lse, float *dqkv, int32_t B, int32_t T, int32_t H, int32_t Dh,
                               float p_drop, uint64_t seed, const uint64_t *step, int32_t split, ac_stream_t stream) {
    if (!dout || !qkv || !out || !lse || !dqkv || B <= 0 || T <= 0 || H <= 0) return AC_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f) return AC_EINVAL;
    if (Dh != 16 || T > ATT_TMAX) return AC_EINVAL;
    if (!ac_aligned16(qkv) || !ac_aligned16(dout) || !ac_aligned16(out) || !ac_aligned16(dqkv)) return AC_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const bool drop = p_drop > 0.f;
    if (split) return drop ? launch_bwd<true, true>(dout, qkv, pad, out, lse, dqkv, B, T, H, p_drop, seed, step, st)
                           : launch_bwd<true, false>(dout, qkv, pad, out, lse, dqkv, B, T, H, p_drop, seed, step, st);
    return drop ? launch_bwd<false, true>(dout, qkv, pad, out, lse, dqkv, B, T, H, p_drop, seed, step, st)
                : launch_bwd<false, false>(dout, qkv, pad, out, lse, dqkv, B, T, H, p_drop, seed, step, st);
}
