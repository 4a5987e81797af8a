// Split-bf16 operand planes (math mode 'bf16x3'): x -> hi = bf16(x), lo = bf16(x - hi), both
// round-to-nearest-even, written as two bf16 planes.  The conv products of the spectra branch read
// their operands k times (implicit GEMM) or keep them resident in LDS (window kernel): splitting once
// into planes keeps the conversion out of those inner loops (ac_gemm.hip's on-the-fly split costs
// ~16 VALU instructions per 4 elements per READ).
#include "ac_common.h"
#include <hip/hip_bf16.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short s_f2bf(float x) {
    return ac_f2h(x);
}
__device__ __forceinline__ float s_bf2f(unsigned short h) { return ac_h2f(h); }

__device__ __forceinline__ void split8(const f32x4 &a, const f32x4 &b, s16x8 &hi, s16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned short h0 = s_f2bf(a[j]), h1 = s_f2bf(b[j]);
        hi[j] = (short)h0;
        hi[4 + j] = (short)h1;
        lo[j] = (short)s_f2bf(a[j] - s_bf2f(h0));
        lo[4 + j] = (short)s_f2bf(b[j] - s_bf2f(h1));
    }
}

__global__ void split_kernel(const float *__restrict__ x, unsigned short *__restrict__ hi,
                             unsigned short *__restrict__ lo, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        s16x8 h, l;
        split8(*(const f32x4 *)(x + 8 * i), *(const f32x4 *)(x + 8 * i + 4), h, l);
        *(s16x8 *)(hi + 8 * i) = h;
        *(s16x8 *)(lo + 8 * i) = l;
    }
}
__global__ void split_tail_kernel(const float *__restrict__ x, unsigned short *__restrict__ hi,
                                  unsigned short *__restrict__ lo, int64_t begin, int64_t n) {
    const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned short h = s_f2bf(x[i]);
        hi[i] = h;
        lo[i] = s_f2bf(x[i] - s_bf2f(h));
    }
}

// y[c, r] = split(x[r, c]) through a 64x64 LDS tile
__global__ __launch_bounds__(256) void transpose_split_kernel(const float *__restrict__ x, int64_t ldx,
                                                              unsigned short *__restrict__ hi,
                                                              unsigned short *__restrict__ lo, int64_t ldy,
                                                              int64_t rows, int cols) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t rr = r0 + i;
        const int cc = c0 + tx;
        tile[i][tx] = (rr < rows && cc < cols) ? x[rr * ldx + cc] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int cc = c0 + i;
        const int64_t rr = r0 + tx;
        if (cc < cols && rr < rows) {
            const float v = tile[tx][i];
            const unsigned short h = s_f2bf(v);
            hi[(int64_t)cc * ldy + rr] = h;
            lo[(int64_t)cc * ldy + rr] = s_f2bf(v - s_bf2f(h));
        }
    }
}

// zero-padded [B, Lp, C] planes of x [B, L, C]: 8 channels per thread
__global__ void pad_rows_split_kernel(const float *__restrict__ x, unsigned short *__restrict__ hi,
                                      unsigned short *__restrict__ lo, int B, int L, int C, int pad_lo, int Lp) {
    const int C8 = C >> 3;
    const unsigned n = (unsigned)B * Lp * C8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned c8 = i % C8, t = i / C8;
        const unsigned lp = t % Lp, b = t / Lp;
        const int l = (int)lp - pad_lo;
        s16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, lw = h;
        if (l >= 0 && l < L) {
            const float *src = x + ((int64_t)b * L + l) * C + 8 * c8;
            split8(*(const f32x4 *)src, *(const f32x4 *)(src + 4), h, lw);
        }
        *(s16x8 *)(hi + (int64_t)i * 8) = h;
        *(s16x8 *)(lo + (int64_t)i * 8) = lw;
    }
}

}  // namespace

extern "C" int ac_split_bf16(const float *x, void *hi, void *lo, int64_t n, ac_stream_t stream) {
    if (!x || !hi || !lo || n < 0) return AC_EINVAL;
    if (n == 0) return AC_OK;
    const int64_t n8 = (ac_aligned16(x) && ac_aligned16(hi) && ac_aligned16(lo)) ? n / 8 : 0;
    if (n8 > 0) {
        int64_t g = (n8 + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(split_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, (unsigned short *)hi,
                           (unsigned short *)lo, n8);
    }
    if (8 * n8 < n) {
        const int64_t rem = n - 8 * n8;
        hipLaunchKernelGGL(split_tail_kernel, dim3((int)((rem + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                           (unsigned short *)hi, (unsigned short *)lo, 8 * n8, n);
    }
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_transpose_split_bf16(const float *x, int64_t ldx, void *hi, void *lo, int64_t ldy,
                                       int64_t rows, int32_t cols, ac_stream_t stream) {
    if (!x || !hi || !lo || rows <= 0 || cols <= 0) return AC_EINVAL;
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    hipLaunchKernelGGL(transpose_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (unsigned short *)hi, (unsigned short *)lo, ldy, rows, cols);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_pad_rows_split(const float *x, void *hi, void *lo, int32_t B, int32_t L, int32_t C,
                                 int32_t pad_lo, int32_t Lp, ac_stream_t stream) {
    if (!x || !hi || !lo || B <= 0 || L <= 0 || C <= 0 || pad_lo < 0 || Lp < L + pad_lo) return AC_EINVAL;
    if (C % 8 || !ac_aligned16(x) || !ac_aligned16(hi) || !ac_aligned16(lo) ||
        (int64_t)B * Lp * (C / 8) >= (1ll << 31))
        return AC_EALIGN;
    const int64_t n = (int64_t)B * Lp * (C / 8);
    int64_t g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(pad_rows_split_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short *)hi, (unsigned short *)lo, B, L, C, pad_lo, Lp);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
