// Depthwise 7x7 convolution (pad 3) on NHWC fp32 planes — the ConvNeXt block's
// spatial mixing (timm convnext_tiny, called from astrominn.py:12-17).
//
// HBM-bound (49 MAC per element).  A workgroup owns one 32-channel slice of a
// sample: the whole H x W x 32 plane sits in LDS (15x15x32 fp32 = 28.8 KB for
// stage 0), every HBM byte is read once as 128-byte lines and written once.
// Thread = (channel c = t&31, pixel slot t>>5): LDS reads are lane-contiguous
// (conflict free), the 49 taps of channel c live in registers.
#include "ac_common.h"

namespace {

constexpr int CG = 32;  // channels per workgroup
// padded plane (3-pixel zero halo) so the 49-tap loops carry no bounds checks
__host__ __device__ inline int padded_elems(int H, int W) { return (H + 6) * (W + 6) * CG; }

__device__ __forceinline__ void load_padded_plane(float *plane, const float *src, int H, int W,
                                                  int C, int cglob, bool cvalid, int c, int ps) {
    const int Wp = W + 6, np = (H + 6) * Wp;
    for (int q = ps; q < np; q += 8) {
        const int yy = q / Wp - 3, xx = q % Wp - 3;
        float v = 0.f;
        if (cvalid && yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = src[(int64_t)(yy * W + xx) * C + cglob];
        plane[q * CG + c] = v;
    }
}

__global__ __launch_bounds__(256) void dwconv7x7_fwd_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ y, int H, int W,
                                                            int C) {
    extern __shared__ __attribute__((aligned(16))) float plane[];  // [(H+6)*(W+6)][32]
    const int b = blockIdx.x, cg = blockIdx.y;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cglob = cg * CG + c;
    const bool cvalid = cglob < C;
    const int HW = H * W, Wp = W + 6;
    load_padded_plane(plane, x + (int64_t)b * HW * C, H, W, C, cglob, cvalid, c, ps);
    float wt[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) wt[k] = cvalid ? w[k * C + cglob] : 0.f;
    const float bv = (cvalid && bias) ? bias[cglob] : 0.f;
    __syncthreads();
    if (!cvalid) return;
    float *yb = y + (int64_t)b * HW * C;
    for (int p = ps; p < HW; p += 8) {
        const int py = p / W, px = p - py * W;
        const float *pp = plane + (py * Wp + px) * CG + c;  // tap (0,0) of this pixel
        float acc = bv;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) acc = fmaf(wt[ky * 7 + kx], pp[(ky * Wp + kx) * CG], acc);
        yb[(int64_t)p * C + cglob] = acc;
    }
}

// Backward.  A workgroup loops over SPB samples of one 32-channel slice.  Per sample:
//   phase 1: padded dy plane in LDS -> dx[p] = sum_k w[k] * dy[p - (k - 3)]
//   phase 2: padded x plane in LDS  -> dw[k] += dy[p] * x[p + (k - 3)]   (dy[p] re-read from L2)
// The 49 dw partial sums (+ dbias) of the thread's channel stay in registers across the
// samples, so the global atomics are 1/SPB of the per-sample count.
constexpr int SPB = 4;
__global__ __launch_bounds__(256) void dwconv7x7_bwd_kernel(const float *__restrict__ dy,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ w,
                                                            float *__restrict__ dx,
                                                            float *__restrict__ dw,
                                                            float *__restrict__ dbias, int B, int H,
                                                            int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    const int HW = H * W, Wp = W + 6;
    const int cg = blockIdx.y;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cglob = cg * CG + c;
    const bool cvalid = cglob < C;
    float wt[49], dwacc[49];
    float dbacc = 0.f;
#pragma unroll
    for (int k = 0; k < 49; ++k) {
        wt[k] = cvalid ? w[k * C + cglob] : 0.f;
        dwacc[k] = 0.f;
    }
    for (int s = 0; s < SPB; ++s) {
        const int b = blockIdx.x * SPB + s;
        if (b >= B) break;
        const float *xb = x + (int64_t)b * HW * C;
        const float *dyb = dy + (int64_t)b * HW * C;
        float *dxb = dx + (int64_t)b * HW * C;
        __syncthreads();
        load_padded_plane(plane, dyb, H, W, C, cglob, cvalid, c, ps);
        __syncthreads();
        if (cvalid) {
            for (int p = ps; p < HW; p += 8) {
                const int py = p / W, px = p - py * W;
                // dy[p - (k-3)] = padded[(py + 6 - ky), (px + 6 - kx)]
                const float *pp = plane + ((py + 6) * Wp + (px + 6)) * CG + c;
                float acc = 0.f;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx)
                        acc = fmaf(wt[ky * 7 + kx], pp[-(ky * Wp + kx) * CG], acc);
                dxb[(int64_t)p * C + cglob] = acc;
            }
        }
        __syncthreads();
        load_padded_plane(plane, xb, H, W, C, cglob, cvalid, c, ps);
        __syncthreads();
        if (cvalid) {
            for (int p = ps; p < HW; p += 8) {
                const int py = p / W, px = p - py * W;
                const float dv = dyb[(int64_t)p * C + cglob];
                const float *pp = plane + (py * Wp + px) * CG + c;
                dbacc += dv;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx)
                        dwacc[ky * 7 + kx] = fmaf(dv, pp[(ky * Wp + kx) * CG], dwacc[ky * 7 + kx]);
            }
        }
    }
    // reduce the 8 pixel slots through LDS, then one atomic per (tap, channel)
    __syncthreads();
    float *red = plane;  // [8][50][32] floats = 51.2 KB
    if (cvalid) {
#pragma unroll
        for (int k = 0; k < 49; ++k) red[(ps * 50 + k) * CG + c] = dwacc[k];
        red[(ps * 50 + 49) * CG + c] = dbacc;
    }
    __syncthreads();
    if (cvalid) {
        for (int k = ps; k < 50; k += 8) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[(q * 50 + k) * CG + c];
            if (k < 49)
                atomicAdd(&dw[k * C + cglob], s);
            else if (dbias)
                atomicAdd(&dbias[cglob], s);
        }
    }
}

}  // namespace

extern "C" int ac_dwconv7x7_fwd(const float *x, const float *w, const float *bias, float *y,
                                int32_t B, int32_t H, int32_t W, int32_t C, ac_stream_t stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return AC_EINVAL;
    size_t lds = (size_t)padded_elems(H, W) * sizeof(float);
    if (lds > 65536) return AC_EINVAL;  // up to 16x16 planes (ConvNeXt stage 0 is 15x15)
    dim3 grid(B, (C + CG - 1) / CG);
    hipLaunchKernelGGL(dwconv7x7_fwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias,
                       y, H, W, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_dwconv7x7_bwd(const float *dy, const float *x, const float *w, float *dx,
                                float *dw, float *dbias, int32_t B, int32_t H, int32_t W,
                                int32_t C, ac_stream_t stream) {
    if (!dy || !x || !w || !dx || !dw || B <= 0 || H <= 0 || W <= 0 || C <= 0) return AC_EINVAL;
    dim3 grid((B + SPB - 1) / SPB, (C + CG - 1) / CG);
    size_t planes = (size_t)padded_elems(H, W) * sizeof(float);
    size_t red = (size_t)8 * 50 * CG * sizeof(float);
    size_t lds = planes > red ? planes : red;
    if (lds > 65536) return AC_EINVAL;
    hipLaunchKernelGGL(dwconv7x7_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, dy, x, w,
                       dx, dw, dbias, B, H, W, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
