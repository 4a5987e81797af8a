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

// thread = (channel d, position slot); a workgroup reduces a slab of positions.
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const float *__restrict__ dh, const float *__restrict__ x, const float *__restrict__ tw,
    const float *__restrict__ tb, float *__restrict__ dW, float *__restrict__ dbias,
    float *__restrict__ dtw, float *__restrict__ dtb, float *__restrict__ dcls, int B, int L,
    int D, int pos_per_block) {
    const int64_t npos = (int64_t)B * L;
    const int64_t p0 = (int64_t)blockIdx.x * pos_per_block;
    int64_t p1 = p0 + pos_per_block;
    if (p1 > npos) p1 = npos;
    for (int d = threadIdx.x; d < D; d += 256) {
        float aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ab = 0.f, atw = 0.f, atb = 0.f;
        const float w = tw[d], bb = tb[d];
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t b = p / L;
            const int l = (int)(p - b * L);
            const float g = dh[(b * (L + 1) + l + 1) * D + d];
            const float *xr = x + p * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) aw[c] = fmaf(g, xr[c], aw[c]);
            ab += g;
            const float tt = xr[0];
            if (d == 0) {
                atw = fmaf(g, tt, atw);
                atb += g;
            } else {
                const float cs = cosf(tt * w + bb);
                atw = fmaf(g * cs, tt, atw);
                atb = fmaf(g, cs, atb);
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) atomicAdd(&dW[d * 8 + c], aw[c]);
        atomicAdd(&dbias[d], ab);
        atomicAdd(&dtw[d], atw);
        atomicAdd(&dtb[d], atb);
        if (blockIdx.x == 0) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dh[((int64_t)b * (L + 1)) * D + d];
            atomicAdd(&dcls[d], s);
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float *__restrict__ qkv,
                                                      const uint8_t *__restrict__ pad,
                                                      float *__restrict__ out,
                                                      float *__restrict__ lse, int T, int H,
                                                      float p_drop, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
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
        const uint64_t rbase = (((uint64_t)b * H + h) * T + i) * T;
        for (int j = 0; j < T; ++j) {
            if (msk[j] != 0.f) continue;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = fmaf(q[d], Ks[j * DH + d], s);
            const float mn = fmaxf(m, s);
            const float corr = __expf(m - mn);
            float pj = __expf(s - mn);
            l = l * corr + pj;
            if (p_drop > 0.f) pj = ac_rand01(seed, rbase + j) >= p_drop ? pj * inv_keep : 0.f;
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
    int H, float p_drop, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
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
            if (p_drop > 0.f) dp = ac_rand01(seed, (rb + i) * T + j) >= p_drop ? dp * inv_keep : 0.f;
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
                if (p_drop > 0.f) keep = ac_rand01(seed, (rb + i) * T + j) >= p_drop ? inv_keep : 0.f;
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
    int ppb = 64;
    while ((npos + ppb - 1) / ppb > 2048) ppb *= 2;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((int)((npos + ppb - 1) / ppb)), dim3(256), 0,
                       (hipStream_t)stream, dh, x, tw, tb, dW, dbias, dtw, dtb, dcls, B, L, D, ppb);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_mha_fwd(const float *qkv, const uint8_t *pad, float *out, float *lse, int32_t B,
                          int32_t T, int32_t H, int32_t Dh, float p_drop, uint64_t seed,
                          ac_stream_t stream) {
    if (!qkv || !out || !lse || B <= 0 || T <= 0 || H <= 0) return AC_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f) return AC_EINVAL;
    const size_t lds = ((size_t)2 * T * Dh + T) * sizeof(float);
    if (lds > 65536) return AC_EINVAL;
    dim3 grid(B * H);
    if (Dh == 16)
        hipLaunchKernelGGL(mha_fwd_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, qkv, pad,
                           out, lse, T, H, p_drop, seed);
    else if (Dh == 32)
        hipLaunchKernelGGL(mha_fwd_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, qkv, pad,
                           out, lse, T, H, p_drop, seed);
    else
        return AC_EINVAL;
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_mha_bwd(const float *dout, const float *qkv, const uint8_t *pad,
                          const float *out, const float *lse, float *dqkv, int32_t B, int32_t T,
                          int32_t H, int32_t Dh, float p_drop, uint64_t seed, ac_stream_t stream) {
    if (!dout || !qkv || !out || !lse || !dqkv || B <= 0 || T <= 0 || H <= 0) return AC_EINVAL;
    if (p_drop < 0.f || p_drop >= 1.f) return AC_EINVAL;
    const size_t lds = ((size_t)2 * T * Dh + 3 * T) * sizeof(float);
    if (lds > 65536) return AC_EINVAL;
    dim3 grid(B * H);
    if (Dh == 16)
        hipLaunchKernelGGL(mha_bwd_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, dout, qkv,
                           pad, out, lse, dqkv, T, H, p_drop, seed);
    else if (Dh == 32)
        hipLaunchKernelGGL(mha_bwd_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, dout, qkv,
                           pad, out, lse, dqkv, T, H, p_drop, seed);
    else
        return AC_EINVAL;
    AC_CHECK_LAUNCH();
    return AC_OK;
}
