// Photometry-branch kernels: fused in_proj + Time2Vec + CLS embedding
// (HyraxBaselineCLS.py:58-71, Time2Vec.py:62-72) and the key-padding-masked
// multi-head attention core of nn.TransformerEncoderLayer (HyraxBaselineCLS.py:24-31,78).
//
// Attention tiles are tiny (T <= 258 tokens, d_head = 16): one workgroup per
// (sample, head) keeps K and V (or Q and dO) of that head in LDS; each thread owns
// one query (or key) row in registers and streams over the other side with
// wave-uniform LDS broadcast reads.  No T x T matrix is ever materialised; the
// backward recomputes the probabilities from the saved log-sum-exp.
#include "ac_common.h"

namespace {

#define GSTRIDE(i, n)                                                      \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); \
         i += (int64_t)gridDim.x * blockDim.x)

__global__ void embed_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                 const float *__restrict__ bias, const float *__restrict__ tw,
                                 const float *__restrict__ tb, const float *__restrict__ cls,
                                 float *__restrict__ h, int B, int L, int D) {
    const int64_t n = (int64_t)B * (L + 1) * D;
    GSTRIDE(i, n) {
        const int d = (int)(i % D);
        const int64_t tok = i / D;
        const int t = (int)(tok % (L + 1));
        const int64_t b = tok / (L + 1);
        float v;
        if (t == 0) {
            v = cls[d];
        } else {
            const float *xr = x + (b * L + (t - 1)) * 8;
            const float *wr = W + d * 8;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc = fmaf(xr[c], wr[c], acc);
            acc += bias[d];
            const float tt = xr[0];
            const float te = d == 0 ? tw[0] * tt + tb[0] : sinf(tt * tw[d] + tb[d]);
            v = acc + te;
        }
        h[i] = v;
    }
}

// thread = (channel d = tid % 128, position slot = tid / 128): the two slots walk alternate positions of the
// workgroup's slab, join through LDS, and one of them adds the slab's partial sums to the gradients
// (round 1: half the threads idle, a 64-bit division per position and 1024 workgroups x 11 atomics per
// channel on 1.4 k addresses: 0.33 ms for a 7 -> 128 projection).
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const float *__restrict__ dh, const float *__restrict__ x, const float *__restrict__ tw,
    const float *__restrict__ tb, float *__restrict__ dW, float *__restrict__ dbias,
    float *__restrict__ dtw, float *__restrict__ dtb, float *__restrict__ dcls, int B, int L,
    int D, int pos_per_block) {
    __shared__ float red[12][128];
    const int64_t npos = (int64_t)B * L;
    const int64_t p0 = (int64_t)blockIdx.x * pos_per_block;
    int64_t p1 = p0 + pos_per_block;
    if (p1 > npos) p1 = npos;
    const int lane_d = threadIdx.x & 127, slot = threadIdx.x >> 7;
    for (int d0 = 0; d0 < D; d0 += 128) {
        const int d = d0 + lane_d;
        const bool dv = d < D;
        float aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ab = 0.f, atw = 0.f, atb = 0.f, acls = 0.f;
        if (dv) {
            const float w = tw[d], bb = tb[d];
            // four positions per trip, their loads issued together (one position per trip left every load's
            // ~1 us of latency exposed: 128 dependent trips per thread)
            for (int p = (int)p0 + slot; p < (int)p1; p += 8) {
                float g[4];
                const float *xr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = p + 2 * u;
                    const bool ok = pp < (int)p1;
                    const int pc = ok ? pp : p;
                    const int bq = pc / L, lq = pc - bq * L;
                    g[u] = dh[((int64_t)bq * (L + 1) + lq + 1) * D + d];
                    xr[u] = x + (int64_t)pc * 8;
                    g[u] = ok ? g[u] : 0.f;
                    // the CLS row of a sample (token 0) is summed by whoever owns the sample's first position
                    // (one thread walking all B rows made workgroup 0 the whole kernel: 512 dependent misses)
                    if (ok && lq == 0) acls += dh[((int64_t)bq * (L + 1)) * D + d];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) aw[c] = fmaf(g[u], xr[u][c], aw[c]);
                    ab += g[u];
                    const float tt = xr[u][0];
                    if (d == 0) {
                        atw = fmaf(g[u], tt, atw);
                        atb += g[u];
                    } else {
                        const float cs = cosf(tt * w + bb);
                        atw = fmaf(g[u] * cs, tt, atw);
                        atb = fmaf(g[u], cs, atb);
                    }
                }
            }
        }
        if (slot == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) red[c][lane_d] = aw[c];
            red[8][lane_d] = ab;
            red[9][lane_d] = atw;
            red[10][lane_d] = atb;
            red[11][lane_d] = acls;
        }
        __syncthreads();
        if (slot == 0 && dv) {
#pragma unroll
            for (int c = 0; c < 8; ++c) atomicAdd(&dW[d * 8 + c], aw[c] + red[c][lane_d]);
            atomicAdd(&dbias[d], ab + red[8][lane_d]);
            atomicAdd(&dtw[d], atw + red[9][lane_d]);
            atomicAdd(&dtb[d], atb + red[10][lane_d]);
            const float cl = acls + red[11][lane_d];
            if (cl != 0.f) atomicAdd(&dcls[d], cl);
        }
        __syncthreads();
    }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float *__restrict__ qkv,
                                                      const uint8_t *__restrict__ pad,
                                                      float *__restrict__ out,
                                                      float *__restrict__ lse, int T, int H,
                                                      float p_drop, uint64_t seed,
                                                      const uint64_t *stepp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    seed = ac_step_seed(seed, stepp);
    float *Ks = sm, *Vs = sm + T * DH;
    float *msk = sm + 2 * T * DH;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * DH;
    const float *base = qkv + (int64_t)b * T * 3 * D;
    for (int i = threadIdx.x; i < T * DH; i += 256) {
        const int t = i / DH, d = i % DH;
        Ks[i] = base[(int64_t)t * 3 * D + D + h * DH + d];
        Vs[i] = base[(int64_t)t * 3 * D + 2 * D + h * DH + d];
    }
    for (int t = threadIdx.x; t < T; t += 256) msk[t] = (pad && pad[(int64_t)b * T + t]) ? 1.f : 0.f;
    __syncthreads();
    const float scale = rsqrtf((float)DH);
    const float inv_keep = 1.0f / (1.0f - p_drop);
    for (int i = threadIdx.x; i < T; i += 256) {
        float q[DH], o[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            q[d] = base[(int64_t)i * 3 * D + h * DH + d] * scale;
            o[d] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        const uint64_t tb = ((uint64_t)b * H + h) * T;
        const unsigned thr = ac_att_threshold(p_drop), wq = ac_att_word(seed, tb + i, 0);
        for (int j = 0; j < T; ++j) {
            if (msk[j] != 0.f) continue;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = fmaf(q[d], Ks[j * DH + d], s);
            const float mn = fmaxf(m, s);
            const float corr = __expf(m - mn);
            float pj = __expf(s - mn);
            l = l * corr + pj;
            if (p_drop > 0.f) pj = ac_att_keep(wq, ac_att_word(seed, tb + j, 1), thr) ? pj * inv_keep : 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = fmaf(pj, Vs[j * DH + d], o[d] * corr);
            m = mn;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < DH; ++d) out[((int64_t)b * T + i) * D + h * DH + d] = o[d] * inv;
        lse[((int64_t)b * H + h) * T + i] = m + __logf(l);
    }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_bwd_kernel(
    const float *__restrict__ dout, const float *__restrict__ qkv, const uint8_t *__restrict__ pad,
    const float *__restrict__ out, const float *__restrict__ lse, float *__restrict__ dqkv, int T,
    int H, float p_drop, uint64_t seed, const uint64_t *stepp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    seed = ac_step_seed(seed, stepp);
    float *T0 = sm, *T1 = sm + T * DH;  // phase A: K, V ; phase B: Q*scale, dO
    float *msk = sm + 2 * T * DH, *lses = msk + T, *Dv = lses + T;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * DH;
    const float *base = qkv + (int64_t)b * T * 3 * D;
    float *dbase = dqkv + (int64_t)b * T * 3 * D;
    const float scale = rsqrtf((float)DH);
    const float inv_keep = 1.0f / (1.0f - p_drop);
    const uint64_t rb = ((uint64_t)b * H + h) * T;

    for (int i = threadIdx.x; i < T * DH; i += 256) {
        const int t = i / DH, d = i % DH;
        T0[i] = base[(int64_t)t * 3 * D + D + h * DH + d];
        T1[i] = base[(int64_t)t * 3 * D + 2 * D + h * DH + d];
    }
    for (int t = threadIdx.x; t < T; t += 256) {
        msk[t] = (pad && pad[(int64_t)b * T + t]) ? 1.f : 0.f;
        lses[t] = lse[rb + t];
    }
    __syncthreads();
    // phase A: one query row per thread -> dQ
    for (int i = threadIdx.x; i < T; i += 256) {
        float q[DH], g[DH], dq[DH];
        float Di = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            q[d] = base[(int64_t)i * 3 * D + h * DH + d] * scale;
            g[d] = dout[((int64_t)b * T + i) * D + h * DH + d];
            Di = fmaf(g[d], out[((int64_t)b * T + i) * D + h * DH + d], Di);
            dq[d] = 0.f;
        }
        Dv[i] = Di;
        const float li = lses[i];
        for (int j = 0; j < T; ++j) {
            if (msk[j] != 0.f) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                s = fmaf(q[d], T0[j * DH + d], s);
                dp = fmaf(g[d], T1[j * DH + d], dp);
            }
            const float p = __expf(s - li);
            if (p_drop > 0.f)
                dp = ac_att_keep(ac_att_word(seed, rb + i, 0), ac_att_word(seed, rb + j, 1), ac_att_threshold(p_drop))
                         ? dp * inv_keep : 0.f;
            const float ds = p * (dp - Di);
#pragma unroll
            for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, T0[j * DH + d], dq[d]);
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) dbase[(int64_t)i * 3 * D + h * DH + d] = dq[d] * scale;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * DH; i += 256) {
        const int t = i / DH, d = i % DH;
        T0[i] = base[(int64_t)t * 3 * D + h * DH + d] * scale;
        T1[i] = dout[((int64_t)b * T + t) * D + h * DH + d];
    }
    __syncthreads();
    // phase B: one key row per thread -> dK, dV
    for (int j = threadIdx.x; j < T; j += 256) {
        float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            k[d] = base[(int64_t)j * 3 * D + D + h * DH + d];
            v[d] = base[(int64_t)j * 3 * D + 2 * D + h * DH + d];
            dk[d] = 0.f;
            dv[d] = 0.f;
        }
        if (msk[j] == 0.f) {
            for (int i = 0; i < T; ++i) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) {
                    s = fmaf(T0[i * DH + d], k[d], s);
                    dp = fmaf(T1[i * DH + d], v[d], dp);
                }
                const float p = __expf(s - lses[i]);
                float keep = 1.f;
                if (p_drop > 0.f)
                    keep = ac_att_keep(ac_att_word(seed, rb + i, 0), ac_att_word(seed, rb + j, 1),
                                       ac_att_threshold(p_drop)) ? inv_keep : 0.f;
                const float pk = p * keep;
                const float ds = p * (keep * dp - Dv[i]);
#pragma unroll
                for (int d = 0; d < DH; ++d) {
                    dv[d] = fmaf(pk, T1[i * DH + d], dv[d]);
                    dk[d] = fmaf(ds, T0[i * DH + d], dk[d]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            dbase[(int64_t)j * 3 * D + D + h * DH + d] = dk[d];
            dbase[(int64_t)j * 3 * D + 2 * D + h * DH + d] = dv[d];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Masked pre-training (MPTModel.train_step, HyraxBaselineCLS.py:226-319).
// mpt_mask_kernel: the selection of _mask_batch (:286-319) on the device, one workgroup per light
// curve: k = max(int(n_valid * mask_p), 3) tokens, k/3 per band (fewer when a band is short), the
// remainder from whatever valid tokens are left; channels 2..6 of the selected tokens are zeroed in
// place and the selection is returned.  Random choices are partial Fisher-Yates draws from the
// counter hash (the reference's torch.randperm stream cannot be reproduced; counts and
// eligibility are what the tests pin).
// ---------------------------------------------------------------------------------------------
constexpr int MPT_MAXL = 2048;

__global__ __launch_bounds__(64) void mpt_mask_kernel(float *__restrict__ x,
                                                      const uint8_t *__restrict__ pad,
                                                      uint8_t *__restrict__ masked, int L,
                                                      double mask_p, uint64_t seed,
                                                      const uint64_t *stepp) {
    seed = ac_step_seed(seed, stepp);
    __shared__ unsigned short lst[MPT_MAXL];
    __shared__ unsigned char sel[MPT_MAXL], band[MPT_MAXL];
    const int b = blockIdx.x;
    float *xb = x + (int64_t)b * L * 7;
    const uint8_t *pb = pad + (int64_t)b * L;
    for (int l = threadIdx.x; l < L; l += 64) {
        sel[l] = 0;
        const float x4 = xb[l * 7 + 4], x5 = xb[l * 7 + 5], x6 = xb[l * 7 + 6];
        int bd = 0;                       // argmax, first maximum wins
        if (x5 > x4) bd = 1;
        if (x6 > (x5 > x4 ? x5 : x4)) bd = 2;
        band[l] = pb[l] ? 255 : bd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nvalid = 0;
        for (int l = 0; l < L; ++l) nvalid += band[l] != 255;
        int k = (int)((double)nvalid * mask_p);
        k = k > 3 ? k : 3;
        const int num_each = k / 3, extras = k - 3 * num_each;
        uint64_t ctr = (uint64_t)b << 20;
        auto draw = [&](int n, int take) {
            take = take < n ? take : n;
            for (int i = 0; i < take; ++i) {
                const int j = i + (int)(((uint64_t)ac_hash32(seed, ctr++) * (uint64_t)(n - i)) >> 32);
                const unsigned short tmp = lst[i];
                lst[i] = lst[j];
                lst[j] = tmp;
                sel[lst[i]] = 1;
            }
        };
        for (int bd = 0; bd < 3; ++bd) {
            int n = 0;
            for (int l = 0; l < L; ++l)
                if (band[l] == bd) lst[n++] = (unsigned short)l;
            draw(n, num_each);
        }
        if (extras > 0) {
            int n = 0;
            for (int l = 0; l < L; ++l)
                if (band[l] != 255 && !sel[l]) lst[n++] = (unsigned short)l;
            draw(n, extras);
        }
    }
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += 64) {
        masked[(int64_t)b * L + l] = sel[l];
        if (sel[l]) {
#pragma unroll
            for (int c = 2; c < 7; ++c) xb[l * 7 + c] = 0.f;
        }
    }
}

// Three-term product loss over the selected tokens (:258-278): sums first, then loss and gradients.
// Head outputs carry the CLS row (token 0) of the encoder output; it gets a zero gradient.
__device__ __forceinline__ void mpt_terms(const float *f_hat, const float *b_hat, const float *dt_hat,
                                          const float *data, int b, int l, int L, float &ef,
                                          float &edt, float (&p)[3], int &tb, float &ce) {
    const int64_t tok = (int64_t)b * (L + 1) + l + 1;
    const float *d = data + ((int64_t)b * L + l) * 7;
    ef = f_hat[tok] - d[2];
    const float x4 = d[4], x5 = d[5], x6 = d[6];
    tb = 0;
    if (x5 > x4) tb = 1;
    if (x6 > (x5 > x4 ? x5 : x4)) tb = 2;
    const float gt = (l + 1 < L) ? d[7 + 1] : 0.f;   // roll(dt_prev, -1) with the last entry zeroed
    edt = dt_hat[tok] - gt;
    const float z0 = b_hat[tok * 3], z1 = b_hat[tok * 3 + 1], z2 = b_hat[tok * 3 + 2];
    const float m = fmaxf(z0, fmaxf(z1, z2));
    const float e0 = __expf(z0 - m), e1 = __expf(z1 - m), e2 = __expf(z2 - m);
    const float s = e0 + e1 + e2;
    p[0] = e0 / s; p[1] = e1 / s; p[2] = e2 / s;
    ce = -(((tb == 0 ? z0 : (tb == 1 ? z1 : z2)) - m) - __logf(s));
}

__global__ __launch_bounds__(256) void mpt_loss_sums_kernel(const float *f_hat, const float *b_hat,
                                                            const float *dt_hat, const float *data,
                                                            const uint8_t *masked, float *sums,
                                                            int B, int L) {
    __shared__ float red[4][4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int64_t n = (int64_t)B * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (!masked[i]) continue;
        float ef, edt, p[3], ce;
        int tb;
        mpt_terms(f_hat, b_hat, dt_hat, data, (int)(i / L), (int)(i % L), L, ef, edt, p, tb, ce);
        a0 += ef * ef; a1 += ce; a2 += edt * edt; a3 += 1.f;
    }
    a0 = ac_wave_sum(a0); a1 = ac_wave_sum(a1); a2 = ac_wave_sum(a2); a3 = ac_wave_sum(a3);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = a0; red[1][w] = a1; red[2][w] = a2; red[3][w] = a3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        atomicAdd(&sums[q], (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]));
    }
}

__global__ __launch_bounds__(256) void mpt_loss_grad_kernel(const float *f_hat, const float *b_hat,
                                                            const float *dt_hat, const float *data,
                                                            const uint8_t *masked, const float *sums,
                                                            float *loss, float *df, float *db,
                                                            float *ddt, int B, int L, float lf_w,
                                                            float lb_w, float ldt_w) {
    const float cnt = sums[3];
    const float lf = sums[0] / cnt, lb = sums[1] / cnt, ldt = sums[2] / cnt;   // empty selection: NaN, as torch
    const float c = lf_w * lb_w * ldt_w;
    if (blockIdx.x == 0 && threadIdx.x == 0) loss[0] = c * lf * lb * ldt;
    const int T = L + 1;
    const int64_t n = (int64_t)B * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / T), t = (int)(i % T);
        float gf = 0.f, gdt = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (t > 0 && masked[(int64_t)b * L + t - 1]) {
            float ef, edt, p[3], ce;
            int tb;
            mpt_terms(f_hat, b_hat, dt_hat, data, b, t - 1, L, ef, edt, p, tb, ce);
            gf = c * lb * ldt * 2.f * ef / cnt;
            gdt = c * lf * lb * 2.f * edt / cnt;
            const float s = c * lf * ldt / cnt;
            g0 = s * (p[0] - (tb == 0 ? 1.f : 0.f));
            g1 = s * (p[1] - (tb == 1 ? 1.f : 0.f));
            g2 = s * (p[2] - (tb == 2 ? 1.f : 0.f));
        }
        df[i] = gf;
        ddt[i] = gdt;
        db[i * 3] = g0; db[i * 3 + 1] = g1; db[i * 3 + 2] = g2;
    }
}

}  // namespace

extern "C" int ac_embed_fwd(const float *x, const float *W, const float *bias, const float *tw,
                            const float *tb, const float *cls, float *h, int32_t B, int32_t L,
                            int32_t D, ac_stream_t stream) {
    if (!x || !W || !bias || !tw || !tb || !cls || !h || B <= 0 || L <= 0 || D <= 0)
        return AC_EINVAL;
    const int64_t n = (int64_t)B * (L + 1) * D;
    int64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, W,
                       bias, tw, tb, cls, h, B, L, D);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_embed_bwd(const float *dh, const float *x, const float *tw, const float *tb,
                            float *dW, float *dbias, float *dtw, float *dtb, float *dcls,
                            int32_t B, int32_t L, int32_t D, ac_stream_t stream) {
    if (!dh || !x || !tw || !tb || !dW || !dbias || !dtw || !dtb || !dcls || B <= 0 || L <= 0 ||
        D <= 0)
        return AC_EINVAL;
    const int64_t npos = (int64_t)B * L;
    if (L < 1 || npos >= (int64_t)1 << 30) return AC_EINVAL;
    int ppb = 128;
    while ((npos + ppb - 1) / ppb > 512) ppb *= 2;   // <= 512 workgroups: two per CU, 4x fewer atomics
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((int)((npos + ppb - 1) / ppb)), dim3(256), 0,
                       (hipStream_t)stream, dh, x, tw, tb, dW, dbias, dtw, dtb, dcls, B, L, D, ppb);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_mha_fwd(const float *qkv, const uint8_t *pad, float *out, float *lse, int32_t B,
                          int32_t T, int32_t H, int32_t Dh, float p_drop, uint64_t seed,
                          const uint64_t *step, ac_stream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || T <= 0 || H <= 0) return AC_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f) return AC_EINVAL;
    const size_t lds = ((size_t)2 * T * Dh + T) * sizeof(float);
    if (lds > 65536) return AC_EINVAL;
    dim3 grid(B * H);
    if (Dh == 16)
        hipLaunchKernelGGL(mha_fwd_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, qkv, pad,
                           out, lse, T, H, p_drop, seed, step);
    else if (Dh == 32)
        hipLaunchKernelGGL(mha_fwd_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, qkv, pad,
                           out, lse, T, H, p_drop, seed, step);
    else
        return AC_EINVAL;
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_mha_bwd(const float *dout, const float *qkv, const uint8_t *pad,
                          const float *out, const float *lse, float *dqkv, int32_t B, int32_t T,
                          int32_t H, int32_t Dh, float p_drop, uint64_t seed, const uint64_t *step,
                          ac_stream_t stream) {
    if (!dout || !qkv || !out || !lse || !dqkv || B <= 0 || T <= 0 || H <= 0) return AC_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f) return AC_EINVAL;
    const size_t lds = ((size_t)2 * T * Dh + 3 * T) * sizeof(float);
    if (lds > 65536) return AC_EINVAL;
    dim3 grid(B * H);
    if (Dh == 16)
        hipLaunchKernelGGL(mha_bwd_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, dout, qkv,
                           pad, out, lse, dqkv, T, H, p_drop, seed, step);
    else if (Dh == 32)
        hipLaunchKernelGGL(mha_bwd_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, dout, qkv,
                           pad, out, lse, dqkv, T, H, p_drop, seed, step);
    else
        return AC_EINVAL;
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_mpt_mask(float *x, const uint8_t *pad, uint8_t *masked, int32_t B, int32_t L,
                           double mask_p, uint64_t seed, const uint64_t *step, ac_stream_t stream) {
    if (!x || !pad || !masked || B <= 0 || L <= 0 || L > MPT_MAXL || mask_p < 0.0 || mask_p > 1.0)
        return AC_EINVAL;
    hipLaunchKernelGGL(mpt_mask_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, pad, masked, L,
                       mask_p, seed, step);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_mpt_loss_fwd_bwd(const float *f_hat, const float *b_hat, const float *dt_hat,
                                   const float *data, const uint8_t *masked, float *sums, float *loss,
                                   float *df, float *db, float *ddt, int32_t B, int32_t L,
                                   float lambda_f, float lambda_b, float lambda_dt,
                                   ac_stream_t stream_) {
    if (!f_hat || !b_hat || !dt_hat || !data || !masked || !sums || !loss || !df || !db || !ddt ||
        B <= 0 || L <= 0)
        return AC_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(sums, 0, 4 * sizeof(float), stream);
    if (e != hipSuccess) return -(int)e - 2000;
    const int64_t n = (int64_t)B * (L + 1);
    int grid = (int)((n + 255) / 256);
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL(mpt_loss_sums_kernel, dim3(grid), dim3(256), 0, stream, f_hat, b_hat, dt_hat, data,
                       masked, sums, B, L);
    hipLaunchKernelGGL(mpt_loss_grad_kernel, dim3(grid), dim3(256), 0, stream, f_hat, b_hat, dt_hat, data,
                       masked, sums, loss, df, db, ddt, B, L, lambda_f, lambda_b, lambda_dt);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Photometry collate on the device (src/applecider/datasets/photo_dataset.py:117-152, the legacy 5-modal
// collate src/applecider/models/Time2Vec.py:18-45, HyraxBaselineCLS.to_tensor :152-166): the host ships the RAGGED light
// curves back to back (flat [sum of lengths, 7] + per-sample row offset and length) - no padded host copy, no host
// arithmetic - and one kernel pads / truncates every sample to L rows, standardises the first four channels
//     out[b, l, c] = (x - mean[c]) / (std[c] + 1e-8)        c < 4,  pad rows included (they hold zeros, as the reference's
//                                                            in-place normalisation of the padded array does)
// (IEEE division and the reference's order of operations: bit-identical to the numpy result) and writes the mask
// (1 = padding).  normalise = 0: copy / pad only (PhotoEventsDataset.collate leaves the standardisation to to_tensor).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void collate_photometry_kernel(const float *__restrict__ flat, const int64_t *__restrict__ offsets,
                                                                 const int32_t *__restrict__ lens, const float *__restrict__ mean4,
                                                                 const float *__restrict__ std4, float *__restrict__ out,
                                                                 uint8_t *__restrict__ mask, int B, int L, int normalise) {
    const int64_t n = (int64_t)B * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / L), l = (int)(i - (int64_t)b * L);
        const int len = lens[b] < L ? lens[b] : L;
        const bool valid = l < len;
        const float *src = flat + (offsets[b] + l) * 7;
        float *dst = out + i * 7;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            float v = valid ? src[c] : 0.f;
            if (normalise && c < 4) v = (v - mean4[c]) / (std4[c] + 1e-8f);
            dst[c] = v;
        }
        mask[i] = valid ? 0 : 1;
    }
}

extern "C" int ac_collate_photometry(const float *flat, const int64_t *offsets, const int32_t *lens, const float *mean4,
                                     const float *std4, float *out, uint8_t *mask, int32_t B, int32_t L,
                                     int32_t normalise, ac_stream_t stream) {
    if (!flat || !offsets || !lens || !out || !mask || B <= 0 || L <= 0 || (normalise && (!mean4 || !std4))) return AC_EINVAL;
    const int64_t n = (int64_t)B * L;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(collate_photometry_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, flat, offsets, lens, mean4,
                       std4, out, mask, B, L, normalise);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
