// Row-wise kernels: LayerNorm (+GELU) fwd/bwd, column sums, L2 normalise, softmax, losses.
// All are HBM-bound streaming kernels: one wave per row, lanes stride the channel
// dimension with 16-byte accesses when C % 4 == 0; re-reads of a row hit L1/L2.
#include "ac_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ROWS_BLOCK = 256;  // 4 waves per workgroup

template <bool VEC>
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_fwd_kernel(
    const float *__restrict__ x, int64_t ldx, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ y, int64_t ldy, float *__restrict__ mean,
    float *__restrict__ rstd, int64_t rows, int C, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (ROWS_BLOCK / 64);
    const float invC = 1.0f / (float)C;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float *xr = x + r * ldx;
        float *yr = y + r * ldy;
        float s = 0.f;
        if (VEC) {
            for (int c = lane * 4; c < C; c += 256) {
                f32x4 v = *(const f32x4 *)(xr + c);
                s += (v[0] + v[1]) + (v[2] + v[3]);
            }
        } else {
            for (int c = lane; c < C; c += 64) s += xr[c];
        }
        const float mu = ac_wave_sum(s) * invC;
        float q = 0.f;
        if (VEC) {
            for (int c = lane * 4; c < C; c += 256) {
                f32x4 v = *(const f32x4 *)(xr + c);
                float a = v[0] - mu, b = v[1] - mu, cc = v[2] - mu, dd = v[3] - mu;
                q += (a * a + b * b) + (cc * cc + dd * dd);
            }
        } else {
            for (int c = lane; c < C; c += 64) {
                float a = xr[c] - mu;
                q += a * a;
            }
        }
        const float var = ac_wave_sum(q) * invC;
        const float rs = rsqrtf(var + eps);
        if (lane == 0) {
            if (mean) mean[r] = mu;
            if (rstd) rstd[r] = rs;
        }
        if (VEC) {
            for (int c = lane * 4; c < C; c += 256) {
                f32x4 v = *(const f32x4 *)(xr + c);
                f32x4 g = *(const f32x4 *)(gamma + c);
                f32x4 b = *(const f32x4 *)(beta + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = (v[j] - mu) * rs * g[j] + b[j];
                    o[j] = act == AC_ACT_GELU ? ac_gelu(t) : (act == AC_ACT_GELU_FAST ? ac_gelu_fast(t) : t);
                }
                *(f32x4 *)(yr + c) = o;
            }
        } else {
            for (int c = lane; c < C; c += 64) {
                float t = (xr[c] - mu) * rs * gamma[c] + beta[c];
                yr[c] = act == AC_ACT_GELU ? ac_gelu(t) : (act == AC_ACT_GELU_FAST ? ac_gelu_fast(t) : t);
            }
        }
    }
}

// dgamma/dbeta partials are accumulated per workgroup in LDS, then one global
// atomic per (workgroup, channel).
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_bwd_kernel(
    const float *__restrict__ dy, int64_t lddy, const float *__restrict__ x, int64_t ldx,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ dx,
    int64_t lddx, float *__restrict__ dgamma, float *__restrict__ dbeta, int64_t rows, int C,
    int act) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [2*C]
    for (int c = threadIdx.x; c < 2 * C; c += ROWS_BLOCK) sacc[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (ROWS_BLOCK / 64);
    const float invC = 1.0f / (float)C;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float *xr = x + r * ldx;
        const float *dyr = dy + r * lddy;
        float *dxr = dx + r * lddx;
        const float mu = mean[r], rs = rstd[r];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float xh = (xr[c] - mu) * rs;
            float d = dyr[c];
            if (act == AC_ACT_GELU) d *= ac_gelu_grad(xh * gamma[c] + beta[c]);
                else if (act == AC_ACT_GELU_FAST) d *= ac_gelu_grad_fast(xh * gamma[c] + beta[c]);
            float g = d * gamma[c];
            s1 += g;
            s2 += g * xh;
            atomicAdd(&sacc[c], d * xh);  // ds_add_f32
            atomicAdd(&sacc[C + c], d);
        }
        const float c1 = ac_wave_sum(s1) * invC;
        const float c2 = ac_wave_sum(s2) * invC;
        for (int c = lane; c < C; c += 64) {
            float xh = (xr[c] - mu) * rs;
            float d = dyr[c];
            if (act == AC_ACT_GELU) d *= ac_gelu_grad(xh * gamma[c] + beta[c]);
                else if (act == AC_ACT_GELU_FAST) d *= ac_gelu_grad_fast(xh * gamma[c] + beta[c]);
            float g = d * gamma[c];
            dxr[c] = rs * (g - c1 - xh * c2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += ROWS_BLOCK) {
        if (dgamma) atomicAdd(&dgamma[c], sacc[c]);
        if (dbeta) atomicAdd(&dbeta[c], sacc[C + c]);
    }
}

// Register-resident LayerNorm backward for C % 4 == 0 and C <= 256*J: a wave owns a row, each lane
// keeps J float4 of x / dy in registers (one HBM read of each), accumulates its channels' dgamma /
// dbeta (and optionally the column sums of dx = the bias gradient of the producing conv/linear)
// in registers over all the wave's rows, and the workgroup flushes once through LDS.
template <int J>
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_bwd_reg_kernel(
    const float *__restrict__ dy, int64_t lddy, const float *__restrict__ x, int64_t ldx,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ dx,
    int64_t lddx, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dxsum, int64_t rows, int C, int act) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [3][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + wave;
    const int64_t nwaves = (int64_t)gridDim.x * (ROWS_BLOCK / 64);
    const float invC = 1.0f / (float)C;
    f32x4 g4[J], b4[J], adg[J], adb[J], adx[J];
    bool on[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = 4 * lane + 256 * j;
        on[j] = c < C;
        g4[j] = on[j] ? *(const f32x4 *)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b4[j] = (on[j] && beta) ? *(const f32x4 *)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        adg[j] = adb[j] = adx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = threadIdx.x; c < 3 * C; c += ROWS_BLOCK) sacc[c] = 0.f;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[J], d[J];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (on[j]) {
                const int c = 4 * lane + 256 * j;
                xh[j] = *(const f32x4 *)(x + r * ldx + c);
                d[j] = *(const f32x4 *)(dy + r * lddy + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (xh[j][e] - mu) * rs;
                    float dd = d[j][e];
                    if (act == AC_ACT_GELU) dd *= ac_gelu_grad(h * g4[j][e] + b4[j][e]);
                else if (act == AC_ACT_GELU_FAST) dd *= ac_gelu_grad_fast(h * g4[j][e] + b4[j][e]);
                    xh[j][e] = h;
                    d[j][e] = dd;
                    const float g = dd * g4[j][e];
                    s1 += g;
                    s2 += g * h;
                    adg[j][e] += dd * h;
                    adb[j][e] += dd;
                }
            }
        }
        const float c1 = ac_wave_sum(s1) * invC;
        const float c2 = ac_wave_sum(s2) * invC;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (on[j]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = rs * (d[j][e] * g4[j][e] - c1 - xh[j][e] * c2);
                    adx[j][e] += o[e];
                }
                *(f32x4 *)(dx + r * lddx + 4 * lane + 256 * j) = o;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (on[j]) {
            const int c = 4 * lane + 256 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&sacc[c + e], adg[j][e]);
                atomicAdd(&sacc[C + c + e], adb[j][e]);
                if (dxsum) atomicAdd(&sacc[2 * C + c + e], adx[j][e]);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += ROWS_BLOCK) {
        if (dgamma) atomicAdd(&dgamma[c], sacc[c]);
        if (dbeta) atomicAdd(&dbeta[c], sacc[C + c]);
        if (dxsum) atomicAdd(&dxsum[c], sacc[2 * C + c]);
    }
}

// Register-resident forward: one HBM read of the row (the 3-pass version re-read it through L1).
template <int J>
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_fwd_reg_kernel(
    const float *__restrict__ x, int64_t ldx, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ y, int64_t ldy, float *__restrict__ mean,
    float *__restrict__ rstd, int64_t rows, int C, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (ROWS_BLOCK / 64);
    const float invC = 1.0f / (float)C;
    f32x4 g4[J], b4[J];
    bool on[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = 4 * lane + 256 * j;
        on[j] = c < C;
        g4[j] = on[j] ? *(const f32x4 *)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b4[j] = on[j] ? *(const f32x4 *)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t r = wave0; r < rows; r += nwaves) {
        f32x4 v[J];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            v[j] = on[j] ? *(const f32x4 *)(x + r * ldx + 4 * lane + 256 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mu = ac_wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (on[j]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = v[j][e] - mu;
                    q += a * a;
                }
            }
        }
        const float rs = rsqrtf(ac_wave_sum(q) * invC + eps);
        if (lane == 0) {
            if (mean) mean[r] = mu;
            if (rstd) rstd[r] = rs;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (on[j]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = (v[j][e] - mu) * rs * g4[j][e] + b4[j][e];
                    o[e] = act == AC_ACT_GELU ? ac_gelu(t) : (act == AC_ACT_GELU_FAST ? ac_gelu_fast(t) : t);
                }
                *(f32x4 *)(y + r * ldy + 4 * lane + 256 * j) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sub-wave rows: a row of C = 4*G*J floats is owned by a GROUP of G lanes (G = 8..64), so a wave
// works on 64/G rows at once and every lane carries J float4 — narrow rows (C = 64: one wave per
// row used 16 of its 64 lanes and moved 256 B per load instruction) now keep all lanes and
// 1 KB per instruction in flight.  Reductions stay inside the group (xor-shuffles below G).
// The forward can also emit the bf16 copy the next matrix product reads (saves a cast pass).
// ---------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float ac_group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ unsigned short ln_bf16(float x) {
    return ac_f2h(x);
}

// 4 consecutive values of a row that is fp32 or (is16) bf16 in memory; ld in elements
__device__ __forceinline__ f32x4 ln_load4(const float *base, int64_t elem, int is16) {
    if (is16) {
        const ushort4 h = *(const ushort4 *)((const unsigned short *)base + elem);
        f32x4 v;
        v[0] = ac_h2f(h.x);
        v[1] = ac_h2f(h.y);
        v[2] = ac_h2f(h.z);
        v[3] = ac_h2f(h.w);
        return v;
    }
    return *(const f32x4 *)(base + elem);
}

template <int G, int J, bool X16>
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_fwd_sub_kernel(
    const float *__restrict__ x, int64_t ldx, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ y, int64_t ldy,
    unsigned short *__restrict__ y16, int64_t ldy16, float *__restrict__ mean,
    float *__restrict__ rstd, int64_t rows, float eps, int act) {
    constexpr int RPW = 64 / G, C = 4 * G * J;
    const int lane = threadIdx.x & 63, sub = lane % G, slot = lane / G;
    const int64_t r0 = ((int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6)) * RPW + slot;
    const int64_t rstep = (int64_t)gridDim.x * (ROWS_BLOCK / 64) * RPW;
    const float invC = 1.0f / (float)C;
    f32x4 g4[J], b4[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        g4[j] = *(const f32x4 *)(gamma + 4 * (sub + G * j));
        b4[j] = *(const f32x4 *)(beta + 4 * (sub + G * j));
    }
    for (int64_t r = r0; r < rows; r += rstep) {
        f32x4 v[J];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            v[j] = ln_load4(x, r * ldx + 4 * (sub + G * j), X16);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mu = ac_group_sum<G>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = v[j][e] - mu;
                q += a * a;
            }
        }
        const float rs = rsqrtf(ac_group_sum<G>(q) * invC + eps);
        if (sub == 0) {
            if (mean) mean[r] = mu;
            if (rstd) rstd[r] = rs;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (v[j][e] - mu) * rs * g4[j][e] + b4[j][e];
                o[e] = act == AC_ACT_GELU ? ac_gelu(t) : (act == AC_ACT_GELU_FAST ? ac_gelu_fast(t) : t);
            }
            if (y) *(f32x4 *)(y + r * ldy + 4 * (sub + G * j)) = o;
            if (y16) {
                ushort4 h;
                h.x = ln_bf16(o[0]); h.y = ln_bf16(o[1]); h.z = ln_bf16(o[2]); h.w = ln_bf16(o[3]);
                *(ushort4 *)(y16 + r * ldy16 + 4 * (sub + G * j)) = h;
            }
        }
    }
}

template <int G, int J, bool X16, bool DY16>
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_bwd_sub_kernel(
    const float *__restrict__ dy, int64_t lddy, const float *__restrict__ x, int64_t ldx,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ dx,
    int64_t lddx, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dxsum, int64_t rows, int act, unsigned short *__restrict__ dx16,
    int64_t lddx16, int seg_len, int seg_pitch, int seg_off, unsigned short *__restrict__ dxlo16) {
    constexpr int RPW = 64 / G, C = 4 * G * J;
    __shared__ __attribute__((aligned(16))) float sacc[3 * C];
    const int lane = threadIdx.x & 63, sub = lane % G, slot = lane / G;
    const int64_t r0 = ((int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6)) * RPW + slot;
    const int64_t rstep = (int64_t)gridDim.x * (ROWS_BLOCK / 64) * RPW;
    const float invC = 1.0f / (float)C;
    f32x4 g4[J], b4[J], adg[J], adb[J], adx[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        g4[j] = *(const f32x4 *)(gamma + 4 * (sub + G * j));
        b4[j] = beta ? *(const f32x4 *)(beta + 4 * (sub + G * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
        adg[j] = adb[j] = adx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = threadIdx.x; c < 3 * C; c += ROWS_BLOCK) sacc[c] = 0.f;
    for (int64_t r = r0; r < rows; r += rstep) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[J], d[J];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            xh[j] = ln_load4(x, r * ldx + 4 * (sub + G * j), X16);
            d[j] = ln_load4(dy, r * lddy + 4 * (sub + G * j), DY16);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (xh[j][e] - mu) * rs;
                float dd = d[j][e];
                if (act == AC_ACT_GELU) dd *= ac_gelu_grad(h * g4[j][e] + b4[j][e]);
                else if (act == AC_ACT_GELU_FAST) dd *= ac_gelu_grad_fast(h * g4[j][e] + b4[j][e]);
                xh[j][e] = h;
                d[j][e] = dd;
                const float g = dd * g4[j][e];
                s1 += g;
                s2 += g * h;
                adg[j][e] += dd * h;
                adb[j][e] += dd;
            }
        }
        const float c1 = ac_group_sum<G>(s1) * invC;
        const float c2 = ac_group_sum<G>(s2) * invC;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = rs * (d[j][e] * g4[j][e] - c1 - xh[j][e] * c2);
                adx[j][e] += o[e];
            }
            if (dx) *(f32x4 *)(dx + r * lddx + 4 * (sub + G * j)) = o;
            if (dx16) {
                // bf16 copy for the matrix products that consume dx; rows may land in a zero-padded
                // [batch, seg_pitch, C] buffer (row r = (b, l) -> b*seg_pitch + seg_off + l)
                const int64_t rr = seg_len ? (r / seg_len) * seg_pitch + seg_off + r % seg_len : r;
                ushort4 h;
                h.x = ln_bf16(o[0]); h.y = ln_bf16(o[1]); h.z = ln_bf16(o[2]); h.w = ln_bf16(o[3]);
                *(ushort4 *)(dx16 + rr * lddx16 + 4 * (sub + G * j)) = h;
                if (dxlo16) {
                    // split-bf16 operand planes (math mode bf16x3): lo = bf16(dx - hi), same layout as hi
                    ushort4 l;
                    l.x = ln_bf16(o[0] - ac_h2f(h.x)); l.y = ln_bf16(o[1] - ac_h2f(h.y));
                    l.z = ln_bf16(o[2] - ac_h2f(h.z)); l.w = ln_bf16(o[3] - ac_h2f(h.w));
                    *(ushort4 *)(dxlo16 + rr * lddx16 + 4 * (sub + G * j)) = l;
                }
            }
        }
    }
    // the 64/G row slots of a wave hold partial sums of the same channels: fold them with
    // shuffles first so that only one lane per channel touches LDS
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = G; o < 64; o <<= 1) {
                adg[j][e] += __shfl_xor(adg[j][e], o, 64);
                adb[j][e] += __shfl_xor(adb[j][e], o, 64);
                adx[j][e] += __shfl_xor(adx[j][e], o, 64);
            }
        }
    }
    __syncthreads();
    if (slot == 0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = 4 * (sub + G * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&sacc[c + e], adg[j][e]);
                atomicAdd(&sacc[C + c + e], adb[j][e]);
                if (dxsum) atomicAdd(&sacc[2 * C + c + e], adx[j][e]);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += ROWS_BLOCK) {
        if (dgamma) atomicAdd(&dgamma[c], sacc[c]);
        if (dbeta) atomicAdd(&dbeta[c], sacc[C + c]);
        if (dxsum) atomicAdd(&dxsum[c], sacc[2 * C + c]);
    }
}

// (G, J) with 4*G*J == C, or 0 when no exact sub-wave shape exists
inline bool ln_sub_shape(int C, int *G, int *J) {
    if (C % 4) return false;
    const int V = C / 4;
    const int gs[4] = {64, 32, 16, 8};
    for (int i = 0; i < 4; ++i) {
        if (V % gs[i] == 0) {
            const int j = V / gs[i];
            if (j == 1 || j == 2 || j == 3 || j == 6) {
                *G = gs[i];
                *J = j;
                return true;
            }
        }
    }
    return false;
}

// Wide rows (1536 < C <= 3072, e.g. the 3072-channel last SpectraNet stage): a whole workgroup per
// row, 3 float4 per thread, same register accumulation of dgamma / dbeta / column sums of dx.
__global__ __launch_bounds__(ROWS_BLOCK) void layernorm_bwd_wide_kernel(
    const float *__restrict__ dy, int64_t lddy, const float *__restrict__ x, int64_t ldx,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ dx,
    int64_t lddx, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dxsum, int64_t rows, int C, int act) {
    constexpr int J = 3;
    __shared__ float red[2][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float invC = 1.0f / (float)C;
    f32x4 g4[J], b4[J], adg[J], adb[J], adx[J];
    bool on[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = 4 * t + 1024 * j;
        on[j] = c < C;
        g4[j] = on[j] ? *(const f32x4 *)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b4[j] = (on[j] && beta) ? *(const f32x4 *)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        adg[j] = adb[j] = adx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[J], d[J];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (on[j]) {
                const int c = 4 * t + 1024 * j;
                xh[j] = *(const f32x4 *)(x + r * ldx + c);
                d[j] = *(const f32x4 *)(dy + r * lddy + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (xh[j][e] - mu) * rs;
                    float dd = d[j][e];
                    if (act == AC_ACT_GELU) dd *= ac_gelu_grad(h * g4[j][e] + b4[j][e]);
                else if (act == AC_ACT_GELU_FAST) dd *= ac_gelu_grad_fast(h * g4[j][e] + b4[j][e]);
                    xh[j][e] = h;
                    d[j][e] = dd;
                    const float g = dd * g4[j][e];
                    s1 += g;
                    s2 += g * h;
                    adg[j][e] += dd * h;
                    adb[j][e] += dd;
                }
            }
        }
        s1 = ac_wave_sum(s1);
        s2 = ac_wave_sum(s2);
        __syncthreads();  // previous row's partials consumed
        if (lane == 0) {
            red[0][wave] = s1;
            red[1][wave] = s2;
        }
        __syncthreads();
        const float c1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * invC;
        const float c2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * invC;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (on[j]) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = rs * (d[j][e] * g4[j][e] - c1 - xh[j][e] * c2);
                    adx[j][e] += o[e];
                }
                *(f32x4 *)(dx + r * lddx + 4 * t + 1024 * j) = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (on[j]) {
            const int c = 4 * t + 1024 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dgamma) atomicAdd(&dgamma[c + e], adg[j][e]);
                if (dbeta) atomicAdd(&dbeta[c + e], adb[j][e]);
                if (dxsum) atomicAdd(&dxsum[c + e], adx[j][e]);
            }
        }
    }
}

// out[n] += sum_m x[m, n]: a workgroup owns a slab of rows and a 64-column strip; thread =
// (column, one of 4 row phases) so that narrow matrices still use every lane; each wave reads
// 256 contiguous bytes per row.
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, int64_t ldx,
                                                     float *__restrict__ out, int64_t rows,
                                                     int cols, int rows_per_block) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int64_t r = r0 + ph;
        for (; r + 4 < r1; r += 8) {
            s0 += x[r * ldx + c];
            s1 += x[(r + 4) * ldx + c];
        }
        if (r < r1) s0 += x[r * ldx + c];
    }
    part[ph][cl] = s0 + s1;
    __syncthreads();
    if (ph == 0 && c < cols) atomicAdd(&out[c], (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]));
}

// y16 = bf16(x) and out[c] += sum_r x[r, c] in one pass (the output gradient of a Linear layer is
// needed as a bf16 operand AND as its bias gradient): lane = column pair, 4 row phases.
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float *__restrict__ x, int64_t ldx,
                                                          unsigned short *__restrict__ y16,
                                                          int64_t ldy, float *__restrict__ out,
                                                          int64_t rows, int cols, int rows_per_block) {
    __shared__ float part[4][128];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 128 + 2 * cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float a0 = 0.f, a1 = 0.f;
    if (c < cols) {
        for (int64_t r = r0 + ph; r < r1; r += 4) {
            const float2 v = *(const float2 *)(x + r * ldx + c);
            a0 += v.x;
            a1 += v.y;
            *(unsigned *)(y16 + r * ldy + c) = (unsigned)ac_f2h(v.x) | ((unsigned)ac_f2h(v.y) << 16);
        }
    }
    part[ph][2 * cl] = a0;
    part[ph][2 * cl + 1] = a1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = blockIdx.y * 128 + threadIdx.x;
        if (cc < cols)
            atomicAdd(&out[cc], (part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                    (part[2][threadIdx.x] + part[3][threadIdx.x]));
    }
}

// g = dy * act'(aux) and out[c] += sum_r g[r, c] in one pass: the gradient in front of an activated Linear
// layer is needed as the operand of two products AND, summed over rows, as the bias gradient (fp32 data
// flow: f32 and split-bf16 modes; the bf16 mode has cast_colsum).  Lane = column pair, 4 row phases.
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const float *__restrict__ dy,
                                                             const float *__restrict__ aux, float *__restrict__ g,
                                                             int64_t ld, float *__restrict__ out, int64_t rows,
                                                             int cols, int kind, int rows_per_block, float drop_p,
                                                             uint64_t seed, const uint64_t *stepp) {
    __shared__ float part[4][128];
    seed = ac_step_seed(seed, stepp);
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 128 + 2 * cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float a0 = 0.f, a1 = 0.f;
    if (c < cols) {
        for (int64_t r = r0 + ph; r < r1; r += 4) {
            float2 v = *(const float2 *)(dy + r * ld + c);
            if (drop_p > 0.f) {   // the forward epilogue's mask: element index r * cols + c (ld == cols there)
                const uint64_t i0 = (uint64_t)r * (uint64_t)cols + (uint64_t)c;
                v.x = ac_rand01(seed, i0) >= drop_p ? v.x * inv_keep : 0.f;
                v.y = ac_rand01(seed, i0 + 1) >= drop_p ? v.y * inv_keep : 0.f;
            }
            if (aux) {
                const float2 x = *(const float2 *)(aux + r * ld + c);
                v.x *= ac_dact(x.x, kind);
                v.y *= ac_dact(x.y, kind);
            }
            a0 += v.x;
            a1 += v.y;
            *(float2 *)(g + r * ld + c) = v;
        }
    }
    part[ph][2 * cl] = a0;
    part[ph][2 * cl + 1] = a1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = blockIdx.y * 128 + threadIdx.x;
        if (cc < cols)
            atomicAdd(&out[cc], (part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                    (part[2][threadIdx.x] + part[3][threadIdx.x]));
    }
}

// bf16 input (the bf16 hidden gradient the fused MLP backward keeps): a lane owns a column PAIR
// (one dword), a wave reads 256 contiguous bytes per row, fp32 sums.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short *__restrict__ x,
                                                          int64_t ldx, float *__restrict__ out,
                                                          int64_t rows, int cols, int rows_per_block) {
    __shared__ float part[4][128];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 128 + 2 * cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    if (c < cols) {
        int64_t r = r0 + ph;
        for (; r + 4 < r1; r += 8) {
            const unsigned u = *(const unsigned *)(x + r * ldx + c);
            const unsigned w = *(const unsigned *)(x + (r + 4) * ldx + c);
            a0 += ac_h2f((unsigned short)(u & 0xffffu));
            a1 += ac_h2f((unsigned short)(u >> 16));
            b0 += ac_h2f((unsigned short)(w & 0xffffu));
            b1 += ac_h2f((unsigned short)(w >> 16));
        }
        if (r < r1) {
            const unsigned u = *(const unsigned *)(x + r * ldx + c);
            a0 += ac_h2f((unsigned short)(u & 0xffffu));
            a1 += ac_h2f((unsigned short)(u >> 16));
        }
    }
    part[ph][2 * cl] = a0 + b0;
    part[ph][2 * cl + 1] = a1 + b1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = blockIdx.y * 128 + threadIdx.x;
        if (cc < cols)
            atomicAdd(&out[cc], (part[0][threadIdx.x] + part[1][threadIdx.x]) +
                                    (part[2][threadIdx.x] + part[3][threadIdx.x]));
    }
}

__global__ __launch_bounds__(ROWS_BLOCK) void l2norm_fwd_kernel(const float *__restrict__ x,
                                                                float *__restrict__ y,
                                                                float *__restrict__ norm,
                                                                int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float *xr = x + r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c] * xr[c];
    const float nrm = sqrtf(ac_wave_sum(s));
    if (lane == 0) norm[r] = nrm;
    const float inv = 1.0f / nrm;
    for (int c = lane; c < C; c += 64) y[r * C + c] = xr[c] * inv;
}

// y = x/n  =>  dx = (dy - y * (dy . y)) / n
__global__ __launch_bounds__(ROWS_BLOCK) void l2norm_bwd_kernel(const float *__restrict__ dy,
                                                                const float *__restrict__ y,
                                                                const float *__restrict__ norm,
                                                                float *__restrict__ dx,
                                                                int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += dy[r * C + c] * y[r * C + c];
    const float dot = ac_wave_sum(s);
    const float inv = 1.0f / norm[r];
    for (int c = lane; c < C; c += 64) dx[r * C + c] = (dy[r * C + c] - y[r * C + c] * dot) * inv;
}

__global__ __launch_bounds__(ROWS_BLOCK) void softmax_fwd_kernel(const float *__restrict__ x,
                                                                 float *__restrict__ y,
                                                                 int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (ROWS_BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[r * C + c]);
    m = ac_wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[r * C + c] - m);
    s = ac_wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < C; c += 64) y[r * C + c] = expf(x[r * C + c] - m) * inv;
}

// One thread per sample (C <= 32 classes).  loss is the batch mean.
constexpr int LOSS_MAXC = 32;
__global__ __launch_bounds__(256) void loss_kernel(const float *__restrict__ logits,
                                                   const void *__restrict__ target,
                                                   const float *__restrict__ alpha,
                                                   float *__restrict__ loss,
                                                   float *__restrict__ dlogits, int B, int C,
                                                   int kind, float gamma, float eps) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    float lb = 0.f;
    if (b < B && kind == 3) {
        // nn.MSELoss (mean over all B * C elements): SpectraNet's redshift regression (spectranet.py:178-179)
        const float *t = (const float *)target + (int64_t)b * C;
        const float inv = 1.0f / ((float)B * (float)C);
        for (int c = 0; c < C; ++c) {
            const float d = logits[(int64_t)b * C + c] - t[c];
            lb += d * d * inv;
            dlogits[(int64_t)b * C + c] = 2.f * d * inv;
        }
    } else if (b < B) {
        float z[LOSS_MAXC], y[LOSS_MAXC];
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) {
            z[c] = logits[(int64_t)b * C + c];
            m = fmaxf(m, z[c]);
        }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(z[c] - m);
        const float lse = m + logf(s);
        const float invB = 1.0f / (float)B;
        if (kind == 0) {
            const float *t = (const float *)target + (int64_t)b * C;
            float ysum = 0.f;
            for (int c = 0; c < C; ++c) {
                y[c] = t[c];
                ysum += y[c];
                lb -= y[c] * (z[c] - lse);
            }
            for (int c = 0; c < C; ++c)
                dlogits[(int64_t)b * C + c] = (expf(z[c] - lse) * ysum - y[c]) * invB;
        } else if (kind == 1) {
            const int t = (int)((const int64_t *)target)[b];
            lb = -(z[t] - lse);
            for (int c = 0; c < C; ++c)
                dlogits[(int64_t)b * C + c] = (expf(z[c] - lse) - (c == t ? 1.f : 0.f)) * invB;
        } else {
            const int t = (int)((const int64_t *)target)[b];
            float gsum = 0.f;
            float g[LOSS_MAXC];
            for (int c = 0; c < C; ++c) {
                const float logp = z[c] - lse;
                const float p = expf(logp);
                float yc = eps > 0.f ? (c == t ? 1.f - eps : eps / (float)(C - 1))
                                     : (c == t ? 1.f : 0.f);
                const float a = alpha ? alpha[c] : 1.f;
                const float om = fmaxf(1.f - p, 0.f);
                float fw, dfw;  // (1-p)^gamma and gamma*(1-p)^(gamma-1)
                if (gamma == 0.f) {
                    fw = 1.f;
                    dfw = 0.f;
                } else if (gamma == 2.f) {
                    fw = om * om;
                    dfw = 2.f * om;
                } else {
                    fw = powf(om, gamma);
                    dfw = om > 0.f ? gamma * powf(om, gamma - 1.f) : 0.f;
                }
                lb -= yc * a * fw * logp;
                g[c] = yc * a * (fw - dfw * p * logp);
                gsum += g[c];
                y[c] = p;
            }
            for (int c = 0; c < C; ++c)
                dlogits[(int64_t)b * C + c] = -(g[c] - y[c] * gsum) * invB;
        }
        lb *= invB;
    }
    lb = ac_wave_sum(lb);
    if ((threadIdx.x & 63) == 0 && lb != 0.f) atomicAdd(loss, lb);
}

inline int grid_for_rows(int64_t rows, int rows_per_block, int cap) {
    int64_t g = (rows + rows_per_block - 1) / rows_per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define LN_SUB_CASES(X) \
    X(64, 1) X(64, 2) X(64, 3) X(64, 6) X(32, 1) X(32, 3) X(16, 1) X(16, 3) X(8, 3) X(8, 1)

extern "C" int ac_layernorm_fwd(const float *x, int64_t ldx, const float *gamma,
                                const float *beta, float *y, int64_t ldy, float *mean,
                                float *rstd, int64_t rows, int32_t C, float eps, int32_t act,
                                void *y16, int64_t ldy16, int32_t x_bf16, ac_stream_t stream) {
    if (!x || !gamma || !beta || (!y && !y16) || rows < 0 || C <= 0) return AC_EINVAL;
    if (act != AC_ACT_NONE && act != AC_ACT_GELU && act != AC_ACT_GELU_FAST) return AC_EINVAL;
    if (rows == 0) return AC_OK;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) &&
                     (x_bf16 ? (((uintptr_t)x & 7u) == 0) : ac_aligned16(x)) &&
                     ac_aligned16(y) && ac_aligned16(gamma) && ac_aligned16(beta);
    int G = 0, J = 0;
    const bool sub = vec && ln_sub_shape(C, &G, &J) && (!y16 || (((uintptr_t)y16 & 7u) == 0 && ldy16 % 4 == 0));
    if (sub) {
        const int g2 = grid_for_rows(rows, 4 * (64 / G) * 2, 2048);
#define LN_FWD_SUB(GG, JJ)                                                                        \
    if (G == GG && J == JJ) {                                                                     \
        if (x_bf16)                                                                               \
            hipLaunchKernelGGL((layernorm_fwd_sub_kernel<GG, JJ, true>), dim3(g2), dim3(ROWS_BLOCK), 0, \
                               (hipStream_t)stream, x, ldx, gamma, beta, y, ldy,                  \
                               (unsigned short *)y16, ldy16, mean, rstd, rows, eps, act);         \
        else                                                                                      \
            hipLaunchKernelGGL((layernorm_fwd_sub_kernel<GG, JJ, false>), dim3(g2), dim3(ROWS_BLOCK), 0, \
                               (hipStream_t)stream, x, ldx, gamma, beta, y, ldy,                  \
                               (unsigned short *)y16, ldy16, mean, rstd, rows, eps, act);         \
    }
        LN_SUB_CASES(LN_FWD_SUB)
#undef LN_FWD_SUB
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    if (y16 || !y || x_bf16) return AC_EALIGN;  // bf16 in/out exist on the sub-wave path only
    const int grid = grid_for_rows(rows, ROWS_BLOCK / 64, 256 * 16);
    if (vec && C <= 1536) {
        const int g2 = grid_for_rows(rows, 16, 2048);
#define LN_FWD(JJ)                                                                             \
    hipLaunchKernelGGL(layernorm_fwd_reg_kernel<JJ>, dim3(g2), dim3(ROWS_BLOCK), 0,           \
                       (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C, \
                       eps, act)
        if (C <= 256) LN_FWD(1);
        else if (C <= 512) LN_FWD(2);
        else if (C <= 768) LN_FWD(3);
        else LN_FWD(6);
#undef LN_FWD
    } else if (vec)
        hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3(grid), dim3(ROWS_BLOCK), 0,
                           (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C,
                           eps, act);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3(grid), dim3(ROWS_BLOCK), 0,
                           (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C,
                           eps, act);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_layernorm_bwd(const float *dy, int64_t lddy, const float *x, int64_t ldx,
                                const float *mean, const float *rstd, const float *gamma,
                                const float *beta, float *dx, int64_t lddx, float *dgamma,
                                float *dbeta, float *dxsum, int64_t rows, int32_t C, int32_t act,
                                void *dx16, int64_t lddx16, int32_t seg_len, int32_t seg_pitch,
                                int32_t seg_off, int32_t dy_bf16, int32_t x_bf16, ac_stream_t stream_) {
    return ac_layernorm_bwd_split(dy, lddy, x, ldx, mean, rstd, gamma, beta, dx, lddx, dgamma, dbeta, dxsum, rows, C,
                                  act, dx16, nullptr, lddx16, seg_len, seg_pitch, seg_off, dy_bf16, x_bf16, stream_);
}

extern "C" int ac_layernorm_bwd_split(const float *dy, int64_t lddy, const float *x, int64_t ldx,
                                      const float *mean, const float *rstd, const float *gamma,
                                      const float *beta, float *dx, int64_t lddx, float *dgamma,
                                      float *dbeta, float *dxsum, int64_t rows, int32_t C, int32_t act,
                                      void *dx16, void *dx16_lo, int64_t lddx16, int32_t seg_len,
                                      int32_t seg_pitch, int32_t seg_off, int32_t dy_bf16, int32_t x_bf16,
                                      ac_stream_t stream_) {
    if (!dy || !x || !mean || !rstd || !gamma || (!dx && !dx16) || rows < 0 || C <= 0)
        return AC_EINVAL;
    if (dx16_lo && (!dx16 || ((uintptr_t)dx16_lo & 7u))) return AC_EINVAL;
    if (seg_len < 0 || (seg_len > 0 && (rows % seg_len || seg_pitch < seg_len + seg_off || seg_off < 0)))
        return AC_EINVAL;
    if ((act == AC_ACT_GELU || act == AC_ACT_GELU_FAST) && !beta) return AC_EINVAL;
    if (act != AC_ACT_NONE && act != AC_ACT_GELU && act != AC_ACT_GELU_FAST) return AC_EINVAL;
    if (C > 4096) return AC_EINVAL;  // 3*C floats of LDS
    if (rows == 0) return AC_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool vec4 = (C % 4 == 0) && (lddy % 4 == 0) && (ldx % 4 == 0) && (lddx % 4 == 0) &&
                      (dy_bf16 ? (((uintptr_t)dy & 7u) == 0) : ac_aligned16(dy)) &&
                      (x_bf16 ? (((uintptr_t)x & 7u) == 0) : ac_aligned16(x)) &&
                      ac_aligned16(dx) &&
                      ac_aligned16(gamma) && (!beta || ac_aligned16(beta));
    const bool vec = vec4 && C <= 1536;
    const size_t lds = 3 * (size_t)C * sizeof(float);
    int G = 0, J = 0;
    const bool dx16_ok = !dx16 || (((uintptr_t)dx16 & 7u) == 0 && lddx16 % 4 == 0);
    if (vec4 && dx16_ok && ln_sub_shape(C, &G, &J)) {
        // 512 workgroups, not 2048: every workgroup ends with 3 C atomics onto the same 3 C addresses, and at 2048 that
        // flush - not the streaming - set the time (66 048 x 128: 81 -> 62 us; 131 072 x 768: 462 -> 440 us)
        const int grid = grid_for_rows(rows, 4 * (64 / G) * 4, 512);
#define LN_BWD_ARGS                                                                              \
    dim3(grid), dim3(ROWS_BLOCK), 0, stream, dy, lddy, x, ldx, mean, rstd, gamma, beta, dx, lddx,    \
        dgamma, dbeta, dxsum, rows, act, (unsigned short *)dx16, lddx16, seg_len, seg_pitch, seg_off,     \
        (unsigned short *)dx16_lo
#define LN_BWD_SUB(GG, JJ)                                                                       \
    if (G == GG && J == JJ) {                                                                    \
        if (x_bf16 && dy_bf16)                                                                   \
            hipLaunchKernelGGL((layernorm_bwd_sub_kernel<GG, JJ, true, true>), LN_BWD_ARGS);     \
        else if (x_bf16)                                                                         \
            hipLaunchKernelGGL((layernorm_bwd_sub_kernel<GG, JJ, true, false>), LN_BWD_ARGS);    \
        else if (dy_bf16)                                                                        \
            hipLaunchKernelGGL((layernorm_bwd_sub_kernel<GG, JJ, false, true>), LN_BWD_ARGS);    \
        else                                                                                     \
            hipLaunchKernelGGL((layernorm_bwd_sub_kernel<GG, JJ, false, false>), LN_BWD_ARGS);   \
    }
        LN_SUB_CASES(LN_BWD_SUB)
#undef LN_BWD_SUB
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    if (dx16 || !dx || dy_bf16 || x_bf16) return AC_EALIGN;  // bf16 in/out exist on the sub-wave path only
    if (vec4 && !vec && C <= 3072) {
        // few workgroups: every one ends with 3*C global atomics onto the same 3*C addresses, and
        // 2048 adders per address made that flush, not the streaming, the cost (0.47 ms for 100 MB)
        int64_t g = rows < 512 ? rows : 512;
        hipLaunchKernelGGL(layernorm_bwd_wide_kernel, dim3((int)g), dim3(ROWS_BLOCK), 0, stream, dy,
                           lddy, x, ldx, mean, rstd, gamma, beta, dx, lddx, dgamma, dbeta, dxsum, rows,
                           C, act);
    } else if (vec) {
        const int grid = grid_for_rows(rows, 16, 2048);
#define LN_BWD(JJ)                                                                              \
    hipLaunchKernelGGL(layernorm_bwd_reg_kernel<JJ>, dim3(grid), dim3(ROWS_BLOCK), lds, stream, \
                       dy, lddy, x, ldx, mean, rstd, gamma, beta, dx, lddx, dgamma, dbeta,      \
                       dxsum, rows, C, act)
        if (C <= 256) LN_BWD(1);
        else if (C <= 512) LN_BWD(2);
        else if (C <= 768) LN_BWD(3);
        else LN_BWD(6);
#undef LN_BWD
    } else {
        if (dxsum) return AC_EINVAL;  // fused bias gradient only on the vector paths
        const int grid = grid_for_rows(rows, 64, 1024);
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(ROWS_BLOCK),
                           2 * (size_t)C * sizeof(float), stream, dy, lddy, x, ldx, mean, rstd,
                           gamma, beta, dx, lddx, dgamma, dbeta, rows, C, act);
    }
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_colsum(const float *x, int64_t ldx, float *out, int64_t rows, int32_t cols,
                         int32_t accumulate, ac_stream_t stream) {
    if (!x || !out || rows < 0 || cols <= 0) return AC_EINVAL;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    if (rows == 0) return AC_OK;
    int rpb = 256;
    while ((rows + rpb - 1) / rpb > 2048) rpb *= 2;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 63) / 64));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, out, rows,
                       cols, rpb);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_act_bwd_colsum(const float *dy, const float *aux, float *g, int64_t ld, float *out,
                                 int64_t rows, int32_t cols, int32_t act, int32_t accumulate, float drop_p,
                                 uint64_t drop_seed, const uint64_t *step, ac_stream_t stream) {
    if (!dy || !g || !out || rows < 0 || cols <= 0 || drop_p < 0.f || drop_p >= 1.f) return AC_EINVAL;
    if (!aux && act != AC_ACT_NONE) return AC_EINVAL;
    if (drop_p > 0.f && ld != cols) return AC_EINVAL;   // the mask index is the forward product's m * N + n
    if ((cols & 1) || (ld & 1) || ((uintptr_t)dy & 7u) || ((uintptr_t)aux & 7u) || ((uintptr_t)g & 7u))
        return AC_EALIGN;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    if (rows == 0) return AC_OK;
    int rpb = 128;
    while ((rows + rpb - 1) / rpb > 4096) rpb *= 2;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 127) / 128));
    hipLaunchKernelGGL(act_bwd_colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, aux, g, ld, out,
                       rows, cols, act, rpb, drop_p, drop_seed, step);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_cast_bf16_colsum(const float *x, int64_t ldx, void *y16, int64_t ldy, float *out,
                                   int64_t rows, int32_t cols, int32_t accumulate,
                                   ac_stream_t stream) {
    if (!x || !y16 || !out || rows < 0 || cols <= 0) return AC_EINVAL;
    if ((cols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)x & 7u) || ((uintptr_t)y16 & 3u))
        return AC_EALIGN;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    if (rows == 0) return AC_OK;
    int rpb = 128;
    while ((rows + rpb - 1) / rpb > 4096) rpb *= 2;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 127) / 128));
    hipLaunchKernelGGL(cast_colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (unsigned short *)y16, ldy, out, rows, cols, rpb);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_colsum_bf16(const void *x, int64_t ldx, float *out, int64_t rows, int32_t cols,
                              int32_t accumulate, ac_stream_t stream) {
    if (!x || !out || rows < 0 || cols <= 0) return AC_EINVAL;
    if ((cols & 1) || (ldx & 1) || ((uintptr_t)x & 3u)) return AC_EALIGN;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    if (rows == 0) return AC_OK;
    int rpb = 256;
    while ((rows + rpb - 1) / rpb > 2048) rpb *= 2;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 127) / 128));
    hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)x, ldx, out, rows, cols, rpb);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_l2norm_fwd(const float *x, float *y, float *norm, int64_t rows, int32_t C,
                             ac_stream_t stream) {
    if (!x || !y || !norm || rows <= 0 || C <= 0) return AC_EINVAL;
    const int grid = (int)((rows + 3) / 4);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid), dim3(ROWS_BLOCK), 0, (hipStream_t)stream, x,
                       y, norm, rows, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_l2norm_bwd(const float *dy, const float *y, const float *norm, float *dx,
                             int64_t rows, int32_t C, ac_stream_t stream) {
    if (!dy || !y || !norm || !dx || rows <= 0 || C <= 0) return AC_EINVAL;
    const int grid = (int)((rows + 3) / 4);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid), dim3(ROWS_BLOCK), 0, (hipStream_t)stream, dy,
                       y, norm, dx, rows, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_softmax_fwd(const float *x, float *y, int64_t rows, int32_t C,
                              ac_stream_t stream) {
    if (!x || !y || rows <= 0 || C <= 0) return AC_EINVAL;
    const int grid = (int)((rows + 3) / 4);
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(grid), dim3(ROWS_BLOCK), 0, (hipStream_t)stream, x,
                       y, rows, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_loss_fwd_bwd(const float *logits, const void *target, const float *alpha,
                               float *loss, float *dlogits, int32_t B, int32_t C, int32_t kind,
                               float gamma, float eps, ac_stream_t stream) {
    if (!logits || !target || !loss || !dlogits || B <= 0 || C <= 0 || C > LOSS_MAXC)
        return AC_EINVAL;
    if (kind < 0 || kind > 3) return AC_EINVAL;
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return -(int)e - 2000;
    hipLaunchKernelGGL(loss_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       logits, target, alpha, loss, dlogits, B, C, kind, gamma, eps);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
