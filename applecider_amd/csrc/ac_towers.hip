// Metadata towers and MoE experts of AstroMiNN as grouped, fully fused ResidualTowerBlock kernels.
//   ResidualTowerBlock  src/applecider/models/astrominn.py:44-64
//       h    = GELU(Linear(in -> hid)(x))
//       out  = Linear(hid -> out)(Drop(LN_m(h))) * sigmoid(Linear(hid -> out)(Drop(LN_g(h)))) + skip(x)
//       skip = Linear(in -> out), or the identity when in == out
//   the eight towers on column subsets of the metadata (astrominn.py:94-113, 249-261) and the four
//   fusion experts on the 288-wide concatenation (astrominn.py:129-131, 264-267, 282-295).
// The reference runs ~11 ATen ops per block and direction; round 1 ran them as ~90 launches per step.
// Here ONE launch evaluates up to 8 blocks (blockIdx.y = block, blockIdx.x = a slice of 16 samples):
// the metadata column gather, both LayerNorms (they normalise the same h, so they share mean / rstd),
// both dropouts, the gate, the skip and the write into the caller's concatenated buffer are all inside;
// the backward launch recomputes the cheap intermediates from (pre-GELU hidden, mean, rstd, main, gate)
// and adds the weight gradients of its 16 samples to the gradient buffers with fp32 atomics.
// Exact fp32 FMA arithmetic in every math mode: the top-2 routing downstream of these blocks is the
// place where rounding flips labels (astrominn.py:276), and the whole block is < 0.3 MMAC per sample.
#include "ac_common.h"

namespace {

constexpr int TW_SB = 16;          // samples per workgroup
constexpr int TW_MAX_IN = 288, TW_MAX_HID = 128, TW_MAX_OUT = 32, TW_MAX_GROUPS = 8;
constexpr int TW_NT = 256;

struct TowerParams {
    ac_tower_desc g[TW_MAX_GROUPS];
    int n, B, training;
    float p;
    uint64_t seed;
    const uint64_t *step;
};

__device__ __forceinline__ float dot_gl(const float *__restrict__ w, const float *__restrict__ v, int n) {
    float a = 0.f;
    int i = 0;
    if ((((uintptr_t)w) & 15u) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (; i + 4 <= n; i += 4) {
            const f4 wv = *(const f4 *)(w + i);
            a = fmaf(wv[0], v[i], a);
            a = fmaf(wv[1], v[i + 1], a);
            a = fmaf(wv[2], v[i + 2], a);
            a = fmaf(wv[3], v[i + 3], a);
        }
    }
    for (; i < n; ++i) a = fmaf(w[i], v[i], a);
    return a;
}

// keep / drop of element j of sample s on path `path` (0 main, 1 gate) of block `grp`
__device__ __forceinline__ float keep_scale(uint64_t seed, int grp, int path, int s, int j, float p, float inv_keep) {
    const uint64_t idx = (((uint64_t)(grp * 2 + path) << 32) | (uint32_t)s) * 128ull + (uint64_t)j;
    return ac_rand01(seed, idx) >= p ? inv_keep : 0.f;
}

struct Lds {
    float xs[TW_SB][TW_MAX_IN];
    float hs[TW_SB][TW_MAX_HID];    // fwd: GELU output, then normalised ; bwd: xhat
    float ms[TW_SB][TW_MAX_HID];    // input of the main Linear (after LN affine + dropout)
    float gs[TW_SB][TW_MAX_HID];    // input of the gate Linear
    float stat[TW_SB][2];
};

__device__ __forceinline__ void load_x(const ac_tower_desc &d, int s0, int ns, float (&xs)[TW_SB][TW_MAX_IN]) {
    for (int idx = threadIdx.x; idx < ns * d.n_in; idx += TW_NT) {
        const int s = idx / d.n_in, i = idx - s * d.n_in;
        const int c = d.gather ? d.cols[i] : i;
        xs[s][i] = d.x[(int64_t)(s0 + s) * d.ldx + c];
    }
}

// mean / rstd of hs[s][0..hid) for every sample of the slice: one wave per sample
__device__ __forceinline__ void ln_stats(int ns, int hid, float eps, const float (&hs)[TW_SB][TW_MAX_HID],
                                         float (&stat)[TW_SB][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < ns; s += TW_NT / 64) {
        float a = 0.f;
        for (int j = lane; j < hid; j += 64) a += hs[s][j];
        const float mean = ac_wave_sum(a) / (float)hid;
        float v = 0.f;
        for (int j = lane; j < hid; j += 64) {
            const float dlt = hs[s][j] - mean;
            v = fmaf(dlt, dlt, v);
        }
        const float var = ac_wave_sum(v) / (float)hid;
        if (lane == 0) {
            stat[s][0] = mean;
            stat[s][1] = rsqrtf(var + eps);
        }
    }
}

__global__ __launch_bounds__(TW_NT) void tower_blocks_fwd_kernel(TowerParams p) {
    const ac_tower_desc &d = p.g[blockIdx.y];
    const int s0 = blockIdx.x * TW_SB;
    const int ns = (p.B - s0) < TW_SB ? (p.B - s0) : TW_SB;
    if (ns <= 0) return;
    __shared__ Lds L;
    const int t = threadIdx.x;
    const int SV = d.hid + 2 * d.n_out + 2;
    const uint64_t seed = ac_step_seed(p.seed, p.step);
    const bool drop = p.training && p.p > 0.f;
    const float inv_keep = drop ? 1.0f / (1.0f - p.p) : 1.0f;

    load_x(d, s0, ns, L.xs);
    __syncthreads();
    for (int idx = t; idx < ns * d.hid; idx += TW_NT) {
        const int s = idx / d.hid, j = idx - s * d.hid;
        const float pre = d.b1[j] + dot_gl(d.w1 + (int64_t)j * d.n_in, L.xs[s], d.n_in);
        d.save[(int64_t)(s0 + s) * SV + j] = pre;
        L.hs[s][j] = ac_gelu(pre);
    }
    __syncthreads();
    ln_stats(ns, d.hid, d.eps, L.hs, L.stat);
    __syncthreads();
    for (int idx = t; idx < ns * d.hid; idx += TW_NT) {
        const int s = idx / d.hid, j = idx - s * d.hid;
        const float xh = (L.hs[s][j] - L.stat[s][0]) * L.stat[s][1];
        float m = fmaf(xh, d.lnm_g[j], d.lnm_b[j]), g = fmaf(xh, d.lng_g[j], d.lng_b[j]);
        if (drop) {
            m *= keep_scale(seed, d.group_id, 0, s0 + s, j, p.p, inv_keep);
            g *= keep_scale(seed, d.group_id, 1, s0 + s, j, p.p, inv_keep);
        }
        L.ms[s][j] = m;
        L.gs[s][j] = g;
    }
    if (t < ns) {
        d.save[(int64_t)(s0 + t) * SV + d.hid + 2 * d.n_out] = L.stat[t][0];
        d.save[(int64_t)(s0 + t) * SV + d.hid + 2 * d.n_out + 1] = L.stat[t][1];
    }
    __syncthreads();
    for (int idx = t; idx < ns * d.n_out; idx += TW_NT) {
        const int s = idx / d.n_out, o = idx - s * d.n_out;
        const float main = d.bm[o] + dot_gl(d.wm + (int64_t)o * d.hid, L.ms[s], d.hid);
        const float gate = ac_sigmoid(d.bg[o] + dot_gl(d.wg + (int64_t)o * d.hid, L.gs[s], d.hid));
        const float skip = d.ws ? d.bs[o] + dot_gl(d.ws + (int64_t)o * d.n_in, L.xs[s], d.n_in) : L.xs[s][o];
        d.y[(int64_t)(s0 + s) * d.ldy + o] = fmaf(main, gate, skip);
        float *sv = d.save + (int64_t)(s0 + s) * SV + d.hid;
        sv[o] = main;
        sv[d.n_out + o] = gate;
    }
}

struct LdsB {
    float xs[TW_SB][TW_MAX_IN];
    float xh[TW_SB][TW_MAX_HID];
    float ms[TW_SB][TW_MAX_HID];
    float gs[TW_SB][TW_MAX_HID];
    float dxh[TW_SB][TW_MAX_HID];   // then d(pre-GELU)
    float km[TW_SB][TW_MAX_HID];    // dropout scale of the main path (0 or 1/keep)
    float kg[TW_SB][TW_MAX_HID];
    float dyv[TW_SB][TW_MAX_OUT], dmain[TW_SB][TW_MAX_OUT], dgp[TW_SB][TW_MAX_OUT];
    float stat[TW_SB][2], red[TW_SB][2];
};

__global__ __launch_bounds__(TW_NT) void tower_blocks_bwd_kernel(TowerParams p) {
    const ac_tower_desc &d = p.g[blockIdx.y];
    const int s0 = blockIdx.x * TW_SB;
    const int ns = (p.B - s0) < TW_SB ? (p.B - s0) : TW_SB;
    if (ns <= 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    LdsB &L = *reinterpret_cast<LdsB *>(lds_raw);
    const int t = threadIdx.x;
    const int SV = d.hid + 2 * d.n_out + 2;
    const uint64_t seed = ac_step_seed(p.seed, p.step);
    const bool drop = p.training && p.p > 0.f;
    const float inv_keep = drop ? 1.0f / (1.0f - p.p) : 1.0f;

    // ---- recompute the forward intermediates of the slice
    load_x(d, s0, ns, L.xs);
    if (t < ns) {
        L.stat[t][0] = d.save[(int64_t)(s0 + t) * SV + d.hid + 2 * d.n_out];
        L.stat[t][1] = d.save[(int64_t)(s0 + t) * SV + d.hid + 2 * d.n_out + 1];
    }
    for (int idx = t; idx < ns * d.n_out; idx += TW_NT) {
        const int s = idx / d.n_out, o = idx - s * d.n_out;
        const float *sv = d.save + (int64_t)(s0 + s) * SV + d.hid;
        const float main = sv[o], gate = sv[d.n_out + o];
        const float dy = d.dy[(int64_t)(s0 + s) * d.lddy + o];
        L.dyv[s][o] = dy;
        L.dmain[s][o] = dy * gate;
        L.dgp[s][o] = dy * main * gate * (1.0f - gate);
    }
    __syncthreads();
    for (int idx = t; idx < ns * d.hid; idx += TW_NT) {
        const int s = idx / d.hid, j = idx - s * d.hid;
        const float pre = d.save[(int64_t)(s0 + s) * SV + j];
        const float xh = (ac_gelu(pre) - L.stat[s][0]) * L.stat[s][1];
        const float km = drop ? keep_scale(seed, d.group_id, 0, s0 + s, j, p.p, inv_keep) : 1.0f;
        const float kg = drop ? keep_scale(seed, d.group_id, 1, s0 + s, j, p.p, inv_keep) : 1.0f;
        L.xh[s][j] = xh;
        L.km[s][j] = km;
        L.kg[s][j] = kg;
        L.ms[s][j] = fmaf(xh, d.lnm_g[j], d.lnm_b[j]) * km;
        L.gs[s][j] = fmaf(xh, d.lng_g[j], d.lng_b[j]) * kg;
    }
    __syncthreads();

    // ---- output Linears: dWm, dWg (+ biases), skip Linear: dWs, dbs
    for (int idx = t; idx < d.n_out * d.hid; idx += TW_NT) {
        const int o = idx / d.hid, j = idx - o * d.hid;
        float am = 0.f, ag = 0.f;
        for (int s = 0; s < ns; ++s) {
            am = fmaf(L.dmain[s][o], L.ms[s][j], am);
            ag = fmaf(L.dgp[s][o], L.gs[s][j], ag);
        }
        atomicAdd(d.dwm + idx, am);
        atomicAdd(d.dwg + idx, ag);
    }
    if (t < d.n_out) {
        float am = 0.f, ag = 0.f, as = 0.f;
        for (int s = 0; s < ns; ++s) {
            am += L.dmain[s][t];
            ag += L.dgp[s][t];
            as += L.dyv[s][t];
        }
        atomicAdd(d.dbm + t, am);
        atomicAdd(d.dbg + t, ag);
        if (d.ws) atomicAdd(d.dbs + t, as);
    }
    if (d.ws)
        for (int idx = t; idx < d.n_out * d.n_in; idx += TW_NT) {
            const int o = idx / d.n_in, i = idx - o * d.n_in;
            float a = 0.f;
            for (int s = 0; s < ns; ++s) a = fmaf(L.dyv[s][o], L.xs[s][i], a);
            atomicAdd(d.dws + idx, a);
        }

    // ---- back through the two Linears, the dropouts and the LayerNorm affines
    for (int idx = t; idx < ns * d.hid; idx += TW_NT) {
        const int s = idx / d.hid, j = idx - s * d.hid;
        float am = 0.f, ag = 0.f;
        for (int o = 0; o < d.n_out; ++o) {
            am = fmaf(d.wm[(int64_t)o * d.hid + j], L.dmain[s][o], am);
            ag = fmaf(d.wg[(int64_t)o * d.hid + j], L.dgp[s][o], ag);
        }
        am *= L.km[s][j];
        ag *= L.kg[s][j];
        L.ms[s][j] = am;     // d(LN_m output)
        L.gs[s][j] = ag;     // d(LN_g output)
        L.dxh[s][j] = fmaf(am, d.lnm_g[j], ag * d.lng_g[j]);
    }
    __syncthreads();
    if (t < d.hid) {
        float gm = 0.f, bm = 0.f, gg = 0.f, bg = 0.f;
        for (int s = 0; s < ns; ++s) {
            gm = fmaf(L.ms[s][t], L.xh[s][t], gm);
            bm += L.ms[s][t];
            gg = fmaf(L.gs[s][t], L.xh[s][t], gg);
            bg += L.gs[s][t];
        }
        atomicAdd(d.dlnm_g + t, gm);
        atomicAdd(d.dlnm_b + t, bm);
        atomicAdd(d.dlng_g + t, gg);
        atomicAdd(d.dlng_b + t, bg);
    }
    // LayerNorm backward per sample: dh = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh))
    {
        const int lane = t & 63, wave = t >> 6;
        for (int s = wave; s < ns; s += TW_NT / 64) {
            float a = 0.f, b = 0.f;
            for (int j = lane; j < d.hid; j += 64) {
                a += L.dxh[s][j];
                b = fmaf(L.dxh[s][j], L.xh[s][j], b);
            }
            a = ac_wave_sum(a);
            b = ac_wave_sum(b);
            if (lane == 0) {
                L.red[s][0] = a / (float)d.hid;
                L.red[s][1] = b / (float)d.hid;
            }
        }
    }
    __syncthreads();
    for (int idx = t; idx < ns * d.hid; idx += TW_NT) {
        const int s = idx / d.hid, j = idx - s * d.hid;
        const float dh = L.stat[s][1] * (L.dxh[s][j] - L.red[s][0] - L.xh[s][j] * L.red[s][1]);
        L.dxh[s][j] = dh * ac_gelu_grad(d.save[(int64_t)(s0 + s) * SV + j]);   // d(pre-GELU)
    }
    __syncthreads();

    // ---- first Linear: dW1, db1 ; dx
    for (int idx = t; idx < d.hid * d.n_in; idx += TW_NT) {
        const int j = idx / d.n_in, i = idx - j * d.n_in;
        float a = 0.f;
        for (int s = 0; s < ns; ++s) a = fmaf(L.dxh[s][j], L.xs[s][i], a);
        atomicAdd(d.dw1 + idx, a);
    }
    if (t < d.hid) {
        float a = 0.f;
        for (int s = 0; s < ns; ++s) a += L.dxh[s][t];
        atomicAdd(d.db1 + t, a);
    }
    if (d.dx)
        for (int idx = t; idx < ns * d.n_in; idx += TW_NT) {
            const int s = idx / d.n_in, i = idx - s * d.n_in;
            float a = 0.f;
            for (int j = 0; j < d.hid; ++j) a = fmaf(d.w1[(int64_t)j * d.n_in + i], L.dxh[s][j], a);
            if (d.ws) {
                for (int o = 0; o < d.n_out; ++o) a = fmaf(d.ws[(int64_t)o * d.n_in + i], L.dyv[s][o], a);
            } else {
                a += L.dyv[s][i];
            }
            const int c = d.gather ? d.cols[i] : i;
            atomicAdd(d.dx + (int64_t)(s0 + s) * d.lddx + c, a);
        }
}

int check(const ac_tower_desc *g, int n, int B, float p, bool bwd) {
    if (!g || n <= 0 || n > TW_MAX_GROUPS || B <= 0 || p < 0.f || p >= 1.f) return AC_EINVAL;
    for (int i = 0; i < n; ++i) {
        const ac_tower_desc &d = g[i];
        if (d.n_in <= 0 || d.n_in > TW_MAX_IN || d.hid <= 0 || d.hid > TW_MAX_HID || d.n_out <= 0 ||
            d.n_out > TW_MAX_OUT)
            return AC_EINVAL;
        if (d.gather && d.n_in > 24) return AC_EINVAL;
        if (!d.ws && d.n_in != d.n_out) return AC_EINVAL;
        if (!d.x || !d.w1 || !d.b1 || !d.lnm_g || !d.lnm_b || !d.lng_g || !d.lng_b || !d.wm || !d.bm ||
            !d.wg || !d.bg || !d.save || (d.ws && !d.bs))
            return AC_EINVAL;
        if (!bwd && !d.y) return AC_EINVAL;
        if (bwd && (!d.dy || !d.dw1 || !d.db1 || !d.dlnm_g || !d.dlnm_b || !d.dlng_g || !d.dlng_b || !d.dwm ||
                    !d.dbm || !d.dwg || !d.dbg || (d.ws && (!d.dws || !d.dbs))))
            return AC_EINVAL;
    }
    return AC_OK;
}

}  // namespace

extern "C" int ac_tower_blocks_fwd(const ac_tower_desc *groups, int32_t n, int32_t B, float p_drop,
                                   int32_t training, uint64_t seed, const uint64_t *step, ac_stream_t stream) {
    const int rc = check(groups, n, B, p_drop, false);
    if (rc != AC_OK) return rc;
    TowerParams p;
    for (int i = 0; i < n; ++i) p.g[i] = groups[i];
    p.n = n; p.B = B; p.training = training; p.p = p_drop; p.seed = seed; p.step = step;
    hipLaunchKernelGGL(tower_blocks_fwd_kernel, dim3((B + TW_SB - 1) / TW_SB, n), dim3(TW_NT), 0,
                       (hipStream_t)stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_tower_blocks_bwd(const ac_tower_desc *groups, int32_t n, int32_t B, float p_drop,
                                   int32_t training, uint64_t seed, const uint64_t *step, ac_stream_t stream) {
    const int rc = check(groups, n, B, p_drop, true);
    if (rc != AC_OK) return rc;
    TowerParams p;
    for (int i = 0; i < n; ++i) p.g[i] = groups[i];
    p.n = n; p.B = B; p.training = training; p.p = p_drop; p.seed = seed; p.step = step;
    static const hipError_t attr = hipFuncSetAttribute((const void *)tower_blocks_bwd_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LdsB));
    if (attr != hipSuccess) return -(int)attr - 2000;
    hipLaunchKernelGGL(tower_blocks_bwd_kernel, dim3((B + TW_SB - 1) / TW_SB, n), dim3(TW_NT), sizeof(LdsB),
                       (hipStream_t)stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
