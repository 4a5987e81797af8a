// Elementwise / layout / pooling kernels (all HBM-bound; grid-stride, coalesced).
#include "ac_common.h"
#include <hip/hip_bf16.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

inline int ew_grid(int64_t n, int per_block = 256) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}
#define GSTRIDE(i, n)                                                      \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); \
         i += (int64_t)gridDim.x * blockDim.x)

__global__ void act_fwd_kernel(const float *x, float *y, int64_t n, int kind) {
    GSTRIDE(i, n) y[i] = ac_act(x[i], kind);
}
__global__ void act_bwd_kernel(const float *dy, const float *aux, float *out, int64_t n, int kind) {
    GSTRIDE(i, n) out[i] = dy[i] * ac_dact(aux[i], kind);
}
__global__ void copy2d_kernel(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows,
                              int cols) {
    const int64_t n = rows * cols;
    GSTRIDE(i, n) {
        int64_t r = i / cols;
        int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}
// y[r, c] = pre[r, c] * colscale[c] + residual[r, c] (each factor optional), 4 columns per thread
__global__ void scale_add_rows_kernel(const float *__restrict__ pre, const float *__restrict__ colscale,
                                      const float *__restrict__ residual, float *__restrict__ y, int64_t n4,
                                      int cols4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    GSTRIDE(i, n4) {
        f32x4 v = ((const f32x4 *)pre)[i];
        if (colscale) v *= ((const f32x4 *)colscale)[i % cols4];
        if (residual) v += ((const f32x4 *)residual)[i];
        ((f32x4 *)y)[i] = v;
    }
}
// y = ((sum_s part[s]) + bias -> pre_out) * colscale + residual, slabs summed in index order (deterministic)
__global__ void splitk_reduce_kernel(const float *__restrict__ part, int split, const float *__restrict__ bias,
                                     float *__restrict__ pre_out, const float *__restrict__ colscale,
                                     const float *__restrict__ residual, float *__restrict__ y, int64_t n4,
                                     int cols4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    GSTRIDE(i, n4) {
        f32x4 v = ((const f32x4 *)part)[i];
        for (int s = 1; s < split; ++s) v += ((const f32x4 *)part)[(int64_t)s * n4 + i];
        const int c = (int)(i % cols4);
        if (bias) v += ((const f32x4 *)bias)[c];
        if (pre_out) ((f32x4 *)pre_out)[i] = v;
        if (colscale) v *= ((const f32x4 *)colscale)[c];
        if (residual) v += ((const f32x4 *)residual)[i];
        ((f32x4 *)y)[i] = v;
    }
}
__global__ void gather_cols_kernel(const float *src, int64_t lds, const int32_t *idx, float *dst,
                                   int64_t ldd, int64_t rows, int ncols) {
    const int64_t n = rows * ncols;
    GSTRIDE(i, n) {
        int64_t r = i / ncols;
        int c = (int)(i - r * ncols);
        dst[r * ldd + c] = src[r * lds + idx[c]];
    }
}
__global__ void gate_fwd_kernel(const float *a, const float *g, const float *s, float *out,
                                int64_t n) {
    GSTRIDE(i, n) out[i] = s ? a[i] * g[i] + s[i] : a[i] * g[i];
}
__global__ void gate_bwd_kernel(const float *dout, const float *a, const float *g, float *da,
                                float *dg, int64_t n) {
    GSTRIDE(i, n) {
        float d = dout[i];
        da[i] = d * g[i];
        dg[i] = d * a[i];
    }
}
__global__ void dropout_kernel(const float *x, float *y, int64_t n, float p, float inv_keep,
                               uint64_t seed, uint64_t offset, const uint64_t *stepp) {
    seed = ac_step_seed(seed, stepp);
    GSTRIDE(i, n) y[i] = ac_rand01(seed, offset + (uint64_t)i) >= p ? x[i] * inv_keep : 0.f;
}
// 16-byte form: same keep decision per element index as the scalar kernel
__global__ void dropout_vec_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n4, float p,
                                   float inv_keep, uint64_t seed, uint64_t offset,
                                   const uint64_t *stepp) {
    typedef float d32x4 __attribute__((ext_vector_type(4)));
    seed = ac_step_seed(seed, stepp);
    GSTRIDE(i, n4) {
        d32x4 v = *(const d32x4 *)(x + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = ac_rand01(seed, offset + (uint64_t)(4 * i + j)) >= p ? v[j] * inv_keep : 0.f;
        *(d32x4 *)(y + 4 * i) = v;
    }
}
__global__ void add_kernel(const float *a, const float *b, float *y, int64_t n, float alpha) {
    GSTRIDE(i, n) y[i] = (a[i] + b[i]) * alpha;
}
__global__ void scale_by_dev_kernel(float *x, int64_t n, const float *s) {
    const float f = s[0];
    GSTRIDE(i, n) x[i] *= f;
}

// dyl = dy*gamma; dgamma[c] += sum_rows dy*ylin.  Workgroup = slab of rows, thread = channel.
// lane = column pair (float2), 4 row phases per workgroup, one LDS fold, one atomic per column and
// workgroup (the first version walked rows serially with one thread per channel: 96 of 256 threads
// busy at C = 96 and 4-byte accesses, 2.1 TB/s)
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const float *__restrict__ dy,
                                                             const float *__restrict__ ylin,
                                                             const float *__restrict__ gamma,
                                                             float *__restrict__ dyl,
                                                             unsigned short *__restrict__ dyl16,
                                                             float *__restrict__ dgamma,
                                                             float *__restrict__ dbias, int64_t rows,
                                                             int C, int rows_per_block) {
    __shared__ float part[2][4][128];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 128 + 2 * cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f, d0 = 0.f, d1 = 0.f;
    if (c < C) {
        const float2 g = *(const float2 *)(gamma + c);
        for (int64_t r = r0 + ph; r < r1; r += 4) {
            const float2 d = *(const float2 *)(dy + r * C + c);
            const float2 yl = *(const float2 *)(ylin + r * C + c);
            s0 += d.x * yl.x; s1 += d.y * yl.y;
            d0 += d.x; d1 += d.y;
            const float o0 = d.x * g.x, o1 = d.y * g.y;
            if (dyl) *(float2 *)(dyl + r * C + c) = float2{o0, o1};
            if (dyl16) {
                const unsigned h0 = ac_f2h(o0);
                const unsigned h1 = ac_f2h(o1);
                *(unsigned *)(dyl16 + r * C + c) = h0 | (h1 << 16);
            }
        }
        d0 *= g.x; d1 *= g.y;
    }
    part[0][ph][2 * cl] = s0; part[0][ph][2 * cl + 1] = s1;
    part[1][ph][2 * cl] = d0; part[1][ph][2 * cl + 1] = d1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = blockIdx.y * 128 + threadIdx.x, k = threadIdx.x;
        if (cc < C) {
            atomicAdd(&dgamma[cc], (part[0][0][k] + part[0][1][k]) + (part[0][2][k] + part[0][3][k]));
            if (dbias) atomicAdd(&dbias[cc], (part[1][0][k] + part[1][1][k]) + (part[1][2][k] + part[1][3][k]));
        }
    }
}

__global__ __launch_bounds__(256) void layerscale_bwd_scalar_kernel(const float *dy, const float *ylin,
                                                                    const float *gamma, float *dyl,
                                                                    unsigned short *dyl16, float *dgamma,
                                                                    float *dbias, int64_t rows, int C,
                                                                    int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gamma[c];
        float s = 0.f, sd = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            const float d = dy[r * C + c];
            s += d * ylin[r * C + c];
            sd += d;
            const float o = d * g;
            if (dyl) dyl[r * C + c] = o;
            if (dyl16) dyl16[r * C + c] = ac_f2h(o);
        }
        atomicAdd(&dgamma[c], s);
        if (dbias) atomicAdd(&dbias[c], sd * g);
    }
}

// NCHW image -> [B*OH*OW, 64] patches in (ky, kx, c) order (c < Cin <= 4 ... 48 used for Cin=3)
__global__ void stem_patchify_kernel(const float *img, float *patches, int B, int Cin, int H,
                                     int W, int OH, int OW) {
    const int64_t n = (int64_t)B * OH * OW * 64;
    GSTRIDE(i, n) {
        int j = (int)(i & 63);
        int64_t p = i >> 6;
        int ox = (int)(p % OW);
        int oy = (int)((p / OW) % OH);
        int b = (int)(p / ((int64_t)OW * OH));
        float v = 0.f;
        if (j < 16 * Cin) {
            int c = j % Cin, kx = (j / Cin) & 3, ky = j / (4 * Cin);
            v = img[(((int64_t)b * Cin + c) * H + (4 * oy + ky)) * W + 4 * ox + kx];
        }
        patches[i] = v;
    }
}

__global__ void avgpool_fwd_kernel(const float *x, float *y, int B, int HW, int C) {
    const int64_t n = (int64_t)B * C;
    const float inv = 1.0f / (float)HW;
    GSTRIDE(i, n) {
        int64_t b = i / C;
        int c = (int)(i - b * C);
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += x[(b * HW + p) * C + c];
        y[i] = s * inv;
    }
}
__global__ void avgpool_bwd_kernel(const float *dy, float *dx, int B, int HW, int C) {
    const int64_t n = (int64_t)B * HW * C;
    const float inv = 1.0f / (float)HW;
    GSTRIDE(i, n) {
        int c = (int)(i % C);
        int64_t b = i / ((int64_t)HW * C);
        dx[i] = dy[b * C + c] * inv;
    }
}

__global__ void maxpool4_fwd_kernel(const float *x, float *y, int64_t y_bstride, uint8_t *idx,
                                    int B, int L, int C) {
    const int Lo = L / 4;
    const int64_t n = (int64_t)B * Lo * C;
    GSTRIDE(i, n) {
        int c = (int)(i % C);
        int64_t t = i / C;
        int lo = (int)(t % Lo);
        int64_t b = t / Lo;
        const float *xp = x + ((b * L + 4 * (int64_t)lo) * C + c);
        float m = xp[0];
        int am = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            float v = xp[(int64_t)j * C];
            if (v > m || (v != v && m == m)) {  // first max wins, NaN propagates (torch semantics)
                m = v;
                am = j;
            }
        }
        y[b * y_bstride + (int64_t)lo * C + c] = m;
        idx[i] = (uint8_t)am;
    }
}
__global__ void maxpool4_bwd_kernel(const float *dy, int64_t dy_bstride, const uint8_t *idx,
                                    float *dx, int B, int L, int C) {
    const int Lo = L / 4;
    const int64_t n = (int64_t)B * L * C;  // every dx element written once (tail rows -> 0)
    GSTRIDE(i, n) {
        int c = (int)(i % C);
        int64_t t = i / C;
        int l = (int)(t % L);
        int64_t b = t / L;
        int lo = l >> 2;
        float v = 0.f;
        if (lo < Lo) {
            int64_t o = (b * Lo + lo) * C + c;
            if (idx[o] == (l & 3)) v = dy[b * dy_bstride + (int64_t)lo * C + c];
        }
        dx[i] = v;
    }
}

// 16-byte forms (C % 4 == 0, L % 4 == 0): a thread owns 4 channels of one pooled position — one
// float4 per input row, 32-bit index arithmetic (the scalar kernels spend their time in 64-bit
// div/mod per element and 4-byte accesses: 1.7 TB/s on the 0.5 GB stage-1 tensors).
typedef float pf32x4 __attribute__((ext_vector_type(4)));
__global__ void maxpool4_fwd_vec_kernel(const float *__restrict__ x, float *__restrict__ y,
                                        int64_t y_bstride, uint8_t *__restrict__ idx, int B, int L, int C) {
    const int Lo = L / 4, C4 = C / 4;
    const unsigned n = (unsigned)B * Lo * C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned c4 = i % C4, t = i / C4;
        const unsigned lo = t % Lo, b = t / Lo;
        const float *xp = x + (((int64_t)b * L + 4 * lo) * C + 4 * c4);
        pf32x4 m = *(const pf32x4 *)xp;
        unsigned am = 0;   // one byte per channel
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const pf32x4 v = *(const pf32x4 *)(xp + (int64_t)j * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (v[e] > m[e] || (v[e] != v[e] && m[e] == m[e])) {
                    m[e] = v[e];
                    am = (am & ~(0xFFu << (8 * e))) | ((unsigned)j << (8 * e));
                }
            }
        }
        *(pf32x4 *)(y + (int64_t)b * y_bstride + (int64_t)lo * C + 4 * c4) = m;
        *(unsigned *)(idx + (int64_t)i * 4) = am;   // idx is [B, Lo, C]: element offset 4*i
    }
}
__global__ void maxpool4_bwd_vec_kernel(const float *__restrict__ dy, int64_t dy_bstride,
                                        const uint8_t *__restrict__ idx, float *__restrict__ dx, int B,
                                        int L, int C) {
    const int Lo = L / 4, C4 = C / 4;
    const unsigned n = (unsigned)B * Lo * C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned c4 = i % C4, t = i / C4;
        const unsigned lo = t % Lo, b = t / Lo;
        const pf32x4 g = *(const pf32x4 *)(dy + (int64_t)b * dy_bstride + (int64_t)lo * C + 4 * c4);
        const unsigned am = *(const unsigned *)(idx + (int64_t)i * 4);
        float *xp = dx + (((int64_t)b * L + 4 * lo) * C + 4 * c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pf32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((am >> (8 * e)) & 0xFFu) == (unsigned)j ? g[e] : 0.f;
            *(pf32x4 *)(xp + (int64_t)j * C) = o;
        }
    }
}

__global__ void globalmax_fwd_kernel(const float *x, float *y, int32_t *idx, int B, int L, int C) {
    const int64_t n = (int64_t)B * C;
    GSTRIDE(i, n) {
        int64_t b = i / C;
        int c = (int)(i - b * C);
        const float *xp = x + b * L * C + c;
        float m = xp[0];
        int am = 0;
        for (int l = 1; l < L; ++l) {
            float v = xp[(int64_t)l * C];
            if (v > m || (v != v && m == m)) {
                m = v;
                am = l;
            }
        }
        y[i] = m;
        idx[i] = am;
    }
}
__global__ void globalmax_bwd_kernel(const float *dy, const int32_t *idx, float *dx, int B, int L,
                                     int C) {
    const int64_t n = (int64_t)B * L * C;
    GSTRIDE(i, n) {
        int c = (int)(i % C);
        int64_t t = i / C;
        int l = (int)(t % L);
        int64_t b = t / L;
        dx[i] = idx[b * C + c] == l ? dy[b * C + c] : 0.f;
    }
}

__global__ void pad_rows_kernel(const float *x, float *y, int B, int L, int C, int pad_lo,
                                int Lp) {
    const int64_t n = (int64_t)B * Lp * C;
    GSTRIDE(i, n) {
        int c = (int)(i % C);
        int64_t t = i / C;
        int lp = (int)(t % Lp);
        int64_t b = t / Lp;
        int l = lp - pad_lo;
        y[i] = (l >= 0 && l < L) ? x[(b * L + l) * C + c] : 0.f;
    }
}

// same, emitting bf16 (the conv operands of the bf16 matrix-core path): 8 channels per thread
__global__ void pad_rows_bf16_kernel(const float *x, unsigned short *y, int B, int L, int C,
                                     int pad_lo, int Lp) {
    const int C8 = C >> 3;
    const unsigned n = (unsigned)B * Lp * C8;   // < 2^31 (checked by the launcher): 32-bit div/mod
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned c8 = i % C8, t = i / C8;
        const unsigned lp = t % Lp, b = t / Lp;
        const int l = (int)lp - pad_lo;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (l >= 0 && l < L) {
            const float *src = x + ((int64_t)b * L + l) * C + 8 * c8;
            const f32x4 a = *(const f32x4 *)src, bb = *(const f32x4 *)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (short)ac_f2h(a[j]);
                o[4 + j] = (short)ac_f2h(bb[j]);
            }
        }
        *(s16x8 *)(y + (int64_t)i * 8) = o;
    }
}
__global__ void pad_rows_bf16_scalar_kernel(const float *x, unsigned short *y, int B, int L, int C,
                                            int pad_lo, int Lp) {
    const int64_t n = (int64_t)B * Lp * C;
    GSTRIDE(i, n) {
        int c = (int)(i % C);
        int64_t t = i / C;
        int lp = (int)(t % Lp);
        int64_t b = t / Lp;
        int l = lp - pad_lo;
        y[i] = ac_f2h((l >= 0 && l < L) ? x[(b * L + l) * C + c] : 0.f);
    }
}

// wexp[(r*Cout + co), t'] = w[co, t' - r]  for 0 <= t'-r < k, else 0
__global__ void toeplitz_expand_kernel(const float *w, float *wexp, int Cout, int k, int Kp,
                                       int shift) {
    const int64_t n = (int64_t)8 * Cout * Kp;
    GSTRIDE(i, n) {
        int tp = (int)(i % Kp);
        int64_t rc = i / Kp;
        int co = (int)(rc % Cout);
        int r = (int)(rc / Cout);
        int t = tp - r - shift;
        wexp[i] = (t >= 0 && t < k) ? w[(int64_t)co * k + t] : 0.f;
    }
}
__global__ void toeplitz_fold_kernel(const float *dwexp, float *dw, int Cout, int k, int Kp,
                                     int shift) {
    const int64_t n = (int64_t)Cout * k;
    GSTRIDE(i, n) {
        int t = (int)(i % k);
        int co = (int)(i / k);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += dwexp[((int64_t)r * Cout + co) * Kp + t + r + shift];
        dw[i] = s;
    }
}

// top-2 of E sigmoid scores; out = w_lo*eo[e_lo] + w_hi*eo[e_hi] added in ascending expert
// order, as the per-expert loop of astrominn.py:282-295 does.  sel = (first, second) by score;
// ties pick the lower expert index.
__global__ void moe_top2_fwd_kernel(const float *scores, const float *eo, float *out, int32_t *sel,
                                    int B, int E, int C) {
    GSTRIDE(b, (int64_t)B) {
        const float *s = scores + b * E;
        int e1 = 0;
        for (int e = 1; e < E; ++e)
            if (s[e] > s[e1]) e1 = e;
        int e2 = e1 == 0 ? 1 : 0;
        for (int e = 0; e < E; ++e)
            if (e != e1 && s[e] > s[e2]) e2 = e;
        sel[2 * b] = e1;
        sel[2 * b + 1] = e2;
        const int lo = e1 < e2 ? e1 : e2, hi = e1 < e2 ? e2 : e1;
        const float wlo = s[lo], whi = s[hi];
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            v += wlo * eo[((int64_t)lo * B + b) * C + c];
            v += whi * eo[((int64_t)hi * B + b) * C + c];
            out[b * C + c] = v;
        }
    }
}
__global__ void moe_top2_bwd_kernel(const float *dout, const float *scores, const float *eo,
                                    const int32_t *sel, float *dscores, float *deo, int B, int E,
                                    int C) {
    GSTRIDE(b, (int64_t)B) {
        const int e1 = sel[2 * b], e2 = sel[2 * b + 1];
        for (int e = 0; e < E; ++e) {
            const bool on = (e == e1) || (e == e2);
            const float w = scores[b * E + e];
            float ds = 0.f;
            for (int c = 0; c < C; ++c) {
                const float d = dout[b * C + c];
                const int64_t o = ((int64_t)e * B + b) * C + c;
                deo[o] = on ? w * d : 0.f;
                if (on) ds += d * eo[o];
            }
            dscores[b * E + e] = ds;
        }
    }
}

}  // namespace

#define EW_LAUNCH(kernel, n, ...)                                                         \
    do {                                                                                  \
        if ((n) <= 0) return AC_OK;                                                       \
        hipLaunchKernelGGL(kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream,   \
                           __VA_ARGS__);                                                  \
        AC_CHECK_LAUNCH();                                                                \
        return AC_OK;                                                                     \
    } while (0)

extern "C" int ac_abi_version(void) { return AC_ABI_VERSION; }

extern "C" const char *ac_strerror(int code) {
    if (code == AC_OK) return "ok";
    if (code == AC_EINVAL) return "invalid argument or unsupported shape";
    if (code == AC_EALIGN) return "pointer or stride not 16-byte aligned";
    if (code == AC_ELAUNCH) return "kernel launch failed";
    if (code <= -2000) return hipGetErrorString((hipError_t)(-(code + 2000)));
    return "unknown error";
}

extern "C" int ac_act_fwd(const float *x, float *y, int64_t n, int32_t kind, ac_stream_t stream) {
    if (!x || !y || n < 0) return AC_EINVAL;
    EW_LAUNCH(act_fwd_kernel, n, x, y, n, kind);
}
extern "C" int ac_act_bwd(const float *dy, const float *aux, float *out, int64_t n, int32_t kind,
                          ac_stream_t stream) {
    if (!dy || !aux || !out || n < 0) return AC_EINVAL;
    EW_LAUNCH(act_bwd_kernel, n, dy, aux, out, n, kind);
}
extern "C" int ac_copy2d(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows,
                         int32_t cols, ac_stream_t stream) {
    if (!src || !dst || rows < 0 || cols < 0) return AC_EINVAL;
    EW_LAUNCH(copy2d_kernel, rows * cols, src, lds, dst, ldd, rows, cols);
}
extern "C" int ac_scale_add_rows(const float *pre, const float *colscale, const float *residual, float *y,
                                 int64_t rows, int32_t cols, ac_stream_t stream) {
    if (!pre || !y || rows < 0 || cols <= 0) return AC_EINVAL;
    if ((cols % 4) || !ac_aligned16(pre) || !ac_aligned16(y) || (colscale && !ac_aligned16(colscale)) ||
        (residual && !ac_aligned16(residual)))
        return AC_EALIGN;
    EW_LAUNCH(scale_add_rows_kernel, rows * (cols / 4), pre, colscale, residual, y, rows * (cols / 4), cols / 4);
}
extern "C" int ac_splitk_reduce(const float *part, int32_t split, const float *bias, float *pre_out,
                                const float *colscale, const float *residual, float *y, int64_t rows, int32_t cols,
                                ac_stream_t stream) {
    if (!part || !y || split < 1 || rows < 0 || cols <= 0) return AC_EINVAL;
    if ((cols % 4) || !ac_aligned16(part) || !ac_aligned16(y) || (bias && !ac_aligned16(bias)) ||
        (pre_out && !ac_aligned16(pre_out)) || (colscale && !ac_aligned16(colscale)) ||
        (residual && !ac_aligned16(residual)) || ((rows * cols) % 4))
        return AC_EALIGN;
    EW_LAUNCH(splitk_reduce_kernel, rows * (cols / 4), part, split, bias, pre_out, colscale, residual, y,
              rows * (cols / 4), cols / 4);
}
extern "C" int ac_gather_cols(const float *src, int64_t lds, const int32_t *idx, float *dst,
                              int64_t ldd, int64_t rows, int32_t ncols, ac_stream_t stream) {
    if (!src || !idx || !dst || rows < 0 || ncols < 0) return AC_EINVAL;
    EW_LAUNCH(gather_cols_kernel, rows * ncols, src, lds, idx, dst, ldd, rows, ncols);
}
extern "C" int ac_gate_fwd(const float *a, const float *g, const float *s, float *out, int64_t n,
                           ac_stream_t stream) {
    if (!a || !g || !out || n < 0) return AC_EINVAL;
    EW_LAUNCH(gate_fwd_kernel, n, a, g, s, out, n);
}
extern "C" int ac_gate_bwd(const float *dout, const float *a, const float *g, float *da,
                           float *dg, int64_t n, ac_stream_t stream) {
    if (!dout || !a || !g || !da || !dg || n < 0) return AC_EINVAL;
    EW_LAUNCH(gate_bwd_kernel, n, dout, a, g, da, dg, n);
}
extern "C" int ac_dropout(const float *x, float *y, int64_t n, float p, uint64_t seed,
                          uint64_t offset, const uint64_t *step, ac_stream_t stream) {
    if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return AC_EINVAL;
    if (n % 4 == 0 && ac_aligned16(x) && ac_aligned16(y))
        EW_LAUNCH(dropout_vec_kernel, n / 4, x, y, n / 4, p, 1.0f / (1.0f - p), seed, offset, step);
    EW_LAUNCH(dropout_kernel, n, x, y, n, p, 1.0f / (1.0f - p), seed, offset, step);
}
extern "C" int ac_add(const float *a, const float *b, float *y, int64_t n, float alpha,
                      ac_stream_t stream) {
    if (!a || !b || !y || n < 0) return AC_EINVAL;
    EW_LAUNCH(add_kernel, n, a, b, y, n, alpha);
}
// dst_j[i] += src[j * seg_len + i], j < nseg <= 4: the column sums a conv bank's backward pass forms in one buffer go
// to the (separate) bias gradients of its convolutions in one launch
struct AddSegs {
    float *dst[4];
};
__global__ void add_segments_kernel(const float *__restrict__ src, AddSegs d, int seg_len, int n) {
    GSTRIDE(i, n) {
        const int j = (int)(i / seg_len);
        d.dst[j][i - (int64_t)j * seg_len] += src[i];
    }
}
extern "C" int ac_add_segments(const float *src, float *dst0, float *dst1, float *dst2, float *dst3, int32_t seg_len,
                               int32_t nseg, ac_stream_t stream) {
    if (!src || seg_len <= 0 || nseg < 1 || nseg > 4) return AC_EINVAL;
    AddSegs d = {{dst0, dst1, dst2, dst3}};
    for (int j = 0; j < nseg; ++j)
        if (!d.dst[j]) return AC_EINVAL;
    const int64_t n = (int64_t)seg_len * nseg;
    EW_LAUNCH(add_segments_kernel, n, src, d, seg_len, (int)n);
}
extern "C" int ac_scale_by_dev(float *x, int64_t n, const float *s, ac_stream_t stream) {
    if (!x || !s || n < 0) return AC_EINVAL;
    EW_LAUNCH(scale_by_dev_kernel, n, x, n, s);
}
extern "C" int ac_layerscale_bwd(const float *dy, const float *ylin, const float *gamma,
                                 float *dyl, void *dyl16, float *dgamma, float *dbias,
                                 int64_t rows, int32_t C, ac_stream_t stream) {
    if (!dy || !ylin || !gamma || (!dyl && !dyl16) || !dgamma || rows <= 0 || C <= 0)
        return AC_EINVAL;
    const bool vec = (C % 2 == 0) && ((uintptr_t)dy % 8 == 0) && ((uintptr_t)ylin % 8 == 0) &&
                     ((uintptr_t)gamma % 8 == 0) && (!dyl || (uintptr_t)dyl % 8 == 0) &&
                     (!dyl16 || (uintptr_t)dyl16 % 4 == 0);
    if (vec) {
        int rpb = 128;
        while ((rows + rpb - 1) / rpb > 4096) rpb *= 2;
        dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((C + 127) / 128));
        hipLaunchKernelGGL(layerscale_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, ylin, gamma,
                           dyl, (unsigned short *)dyl16, dgamma, dbias, rows, C, rpb);
    } else {
        int rpb = 32;
        while ((rows + rpb - 1) / rpb > 4096) rpb *= 2;
        hipLaunchKernelGGL(layerscale_bwd_scalar_kernel, dim3((int)((rows + rpb - 1) / rpb)), dim3(256), 0,
                           (hipStream_t)stream, dy, ylin, gamma, dyl, (unsigned short *)dyl16, dgamma,
                           dbias, rows, C, rpb);
    }
    AC_CHECK_LAUNCH();
    return AC_OK;
}
extern "C" int ac_stem_patchify(const float *img, float *patches, int32_t B, int32_t Cin,
                                int32_t H, int32_t W, ac_stream_t stream) {
    if (!img || !patches || B <= 0 || Cin <= 0 || Cin > 4 || H < 4 || W < 4) return AC_EINVAL;
    const int OH = (H - 4) / 4 + 1, OW = (W - 4) / 4 + 1;
    EW_LAUNCH(stem_patchify_kernel, (int64_t)B * OH * OW * 64, img, patches, B, Cin, H, W, OH, OW);
}
extern "C" int ac_avgpool_fwd(const float *x, float *y, int32_t B, int32_t HW, int32_t C,
                              ac_stream_t stream) {
    if (!x || !y || B <= 0 || HW <= 0 || C <= 0) return AC_EINVAL;
    EW_LAUNCH(avgpool_fwd_kernel, (int64_t)B * C, x, y, B, HW, C);
}
extern "C" int ac_avgpool_bwd(const float *dy, float *dx, int32_t B, int32_t HW, int32_t C,
                              ac_stream_t stream) {
    if (!dy || !dx || B <= 0 || HW <= 0 || C <= 0) return AC_EINVAL;
    EW_LAUNCH(avgpool_bwd_kernel, (int64_t)B * HW * C, dy, dx, B, HW, C);
}
extern "C" int ac_maxpool4_fwd(const float *x, float *y, int64_t y_bstride, uint8_t *idx,
                               int32_t B, int32_t L, int32_t C, ac_stream_t stream) {
    if (!x || !y || !idx || B <= 0 || L < 4 || C <= 0) return AC_EINVAL;
    if (C % 4 == 0 && L % 4 == 0 && y_bstride % 4 == 0 && ac_aligned16(x) && ac_aligned16(y) &&
        ((uintptr_t)idx & 3u) == 0 && (int64_t)B * (L / 4) * (C / 4) < (1ll << 31)) {
        EW_LAUNCH(maxpool4_fwd_vec_kernel, (int64_t)B * (L / 4) * (C / 4), x, y, y_bstride, idx, B, L, C);
    }
    EW_LAUNCH(maxpool4_fwd_kernel, (int64_t)B * (L / 4) * C, x, y, y_bstride, idx, B, L, C);
}
extern "C" int ac_maxpool4_bwd(const float *dy, int64_t dy_bstride, const uint8_t *idx, float *dx,
                               int32_t B, int32_t L, int32_t C, ac_stream_t stream) {
    if (!dy || !idx || !dx || B <= 0 || L < 4 || C <= 0) return AC_EINVAL;
    if (C % 4 == 0 && L % 4 == 0 && dy_bstride % 4 == 0 && ac_aligned16(dy) && ac_aligned16(dx) &&
        ((uintptr_t)idx & 3u) == 0 && (int64_t)B * (L / 4) * (C / 4) < (1ll << 31)) {
        EW_LAUNCH(maxpool4_bwd_vec_kernel, (int64_t)B * (L / 4) * (C / 4), dy, dy_bstride, idx, dx, B, L, C);
    }
    EW_LAUNCH(maxpool4_bwd_kernel, (int64_t)B * L * C, dy, dy_bstride, idx, dx, B, L, C);
}
extern "C" int ac_globalmax_fwd(const float *x, float *y, int32_t *idx, int32_t B, int32_t L,
                                int32_t C, ac_stream_t stream) {
    if (!x || !y || !idx || B <= 0 || L <= 0 || C <= 0) return AC_EINVAL;
    EW_LAUNCH(globalmax_fwd_kernel, (int64_t)B * C, x, y, idx, B, L, C);
}
extern "C" int ac_globalmax_bwd(const float *dy, const int32_t *idx, float *dx, int32_t B,
                                int32_t L, int32_t C, ac_stream_t stream) {
    if (!dy || !idx || !dx || B <= 0 || L <= 0 || C <= 0) return AC_EINVAL;
    EW_LAUNCH(globalmax_bwd_kernel, (int64_t)B * L * C, dy, idx, dx, B, L, C);
}
extern "C" int ac_pad_rows(const float *x, float *y, int32_t B, int32_t L, int32_t C,
                           int32_t pad_lo, int32_t Lp, ac_stream_t stream) {
    if (!x || !y || B <= 0 || L <= 0 || C <= 0 || pad_lo < 0 || Lp < L + pad_lo) return AC_EINVAL;
    EW_LAUNCH(pad_rows_kernel, (int64_t)B * Lp * C, x, y, B, L, C, pad_lo, Lp);
}
extern "C" int ac_pad_rows_bf16(const float *x, void *y, int32_t B, int32_t L, int32_t C,
                                int32_t pad_lo, int32_t Lp, ac_stream_t stream) {
    if (!x || !y || B <= 0 || L <= 0 || C <= 0 || pad_lo < 0 || Lp < L + pad_lo) return AC_EINVAL;
    if (C % 8 == 0 && ac_aligned16(x) && ac_aligned16(y) && (int64_t)B * Lp * (C / 8) < (1ll << 31))
        EW_LAUNCH(pad_rows_bf16_kernel, (int64_t)B * Lp * (C / 8), x, (unsigned short *)y, B, L, C,
                  pad_lo, Lp);
    EW_LAUNCH(pad_rows_bf16_scalar_kernel, (int64_t)B * Lp * C, x, (unsigned short *)y, B, L, C, pad_lo,
              Lp);
}
extern "C" int ac_toeplitz_expand(const float *w, float *wexp, int32_t Cout, int32_t k,
                                  int32_t Kp, int32_t shift, ac_stream_t stream) {
    if (!w || !wexp || Cout <= 0 || k <= 0 || shift < 0 || Kp < k + 7 + shift) return AC_EINVAL;
    EW_LAUNCH(toeplitz_expand_kernel, (int64_t)8 * Cout * Kp, w, wexp, Cout, k, Kp, shift);
}
extern "C" int ac_toeplitz_fold(const float *dwexp, float *dw, int32_t Cout, int32_t k,
                                int32_t Kp, int32_t shift, ac_stream_t stream) {
    if (!dwexp || !dw || Cout <= 0 || k <= 0 || shift < 0 || Kp < k + 7 + shift) return AC_EINVAL;
    EW_LAUNCH(toeplitz_fold_kernel, (int64_t)Cout * k, dwexp, dw, Cout, k, Kp, shift);
}
extern "C" int ac_moe_top2_fwd(const float *scores, const float *expert_out, float *out,
                               int32_t *sel, int32_t B, int32_t E, int32_t C,
                               ac_stream_t stream) {
    if (!scores || !expert_out || !out || !sel || B <= 0 || E < 2 || C <= 0) return AC_EINVAL;
    EW_LAUNCH(moe_top2_fwd_kernel, (int64_t)B, scores, expert_out, out, sel, B, E, C);
}
extern "C" int ac_moe_top2_bwd(const float *dout, const float *scores, const float *expert_out,
                               const int32_t *sel, float *dscores, float *dexpert_out, int32_t B,
                               int32_t E, int32_t C, ac_stream_t stream) {
    if (!dout || !scores || !expert_out || !sel || !dscores || !dexpert_out || B <= 0 || E < 2 ||
        C <= 0)
        return AC_EINVAL;
    EW_LAUNCH(moe_top2_bwd_kernel, (int64_t)B, dout, scores, expert_out, sel, dscores, dexpert_out,
              B, E, C);
}
