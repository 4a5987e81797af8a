// On-box ceilings ("vendor-published figures AND on-box measured ceilings — a
// device-to-device copy kernel for GB/s; a register-resident MFMA loop for FLOP/s — report both").
// Measurement utilities of bench.py: they compute nothing of the model.
//   ac_ceil_copy   16-byte-per-lane streaming copy (the access shape of every HBM-bound kernel here)
//   ac_ceil_mfma   bf16 MFMA loop with every operand in registers (no LDS, no memory traffic in the
//                  loop) on the caller's random operands: what the matrix cores sustain at the clock the
//                  chip holds under that load (MI355X_MICROARCH.md, DVFS give-back).  One wave per SIMD
//                  or two (the window / weight-gradient kernels run 8-wave workgroups).
#include "ac_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// Eight 16-byte loads in flight per lane before the first store (round 3's probe kept one: 4.8-5.1 TB/s, 20 % under
// the 6.3 TB/s the MI355X guide measures for a float4 copy - latency-bound, not bandwidth-bound), streaming
// (non-temporal) accesses: the source is read once, the destination never re-read.
__global__ __launch_bounds__(256) void ceil_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst,
                                                         int64_t n16) {
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n16; i += stride) dst[i] = ac_gload<u32x4>(src + i);
}

// SHAPE 0: v_mfma_f32_16x16x32 (the form of the conv kernels), 16 accumulators of 16x16 per wave = a 64 x 64
// tile; SHAPE 1: v_mfma_f32_32x32x16, 4 accumulators of 32x32 = the same tile.
template <int SHAPE>
__global__ __launch_bounds__(512, 1) void ceil_mfma_kernel(const unsigned short *__restrict__ ops, float *__restrict__ out,
                                                           int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 8 fragments of 8 values per lane: distinct random operands per lane and wave
    const unsigned short *base = ops + ((size_t)(blockIdx.x & 63) * 8 + (wave & 7)) * 64 * 64 + lane * 64;
    bf16x8 A[4], Bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A[i] = ac_gload<bf16x8>(base + 8 * i);
        Bv[i] = ac_gload<bf16x8>(base + 32 + 8 * i);
    }
    float sum = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = AC_MFMA16S(A[i], Bv[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // two k-steps of 16 = the K of one 16x16x32 step: same FLOP per iteration as SHAPE 0
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = AC_MFMA16(A[2 * s + i], Bv[2 * s + j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;   // keeps the chain alive
}

}  // namespace

extern "C" int ac_ceil_copy(const void *src, void *dst, int64_t bytes, ac_stream_t stream) {
    if (!src || !dst || bytes <= 0 || (bytes % 16)) return AC_EINVAL;
    if (!ac_aligned16(src) || !ac_aligned16(dst)) return AC_EALIGN;
    hipLaunchKernelGGL(ceil_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src,
                       (u32x4 *)dst, bytes / 16);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// ops: 64 * 8 * 64 * 64 16-bit values (2 MiB) of the library's operand format, random, finite.
// out: workgroups * waves_per_wg * 64 floats.  FLOP of the launch = workgroups * waves_per_wg * iters * 16 *
// 2 * 16 * 16 * 32 (both shapes).
extern "C" int ac_ceil_mfma(const void *ops, float *out, int32_t shape, int32_t workgroups, int32_t waves_per_wg, int32_t iters,
                            ac_stream_t stream) {
    if (!ops || !out || workgroups <= 0 || iters <= 0 || (waves_per_wg != 4 && waves_per_wg != 8) || (shape != 0 && shape != 1))
        return AC_EINVAL;
    if (shape == 0)
        hipLaunchKernelGGL(ceil_mfma_kernel<0>, dim3(workgroups), dim3(64 * waves_per_wg), 0, (hipStream_t)stream,
                           (const unsigned short *)ops, out, iters);
    else
        hipLaunchKernelGGL(ceil_mfma_kernel<1>, dim3(workgroups), dim3(64 * waves_per_wg), 0, (hipStream_t)stream,
                           (const unsigned short *)ops, out, iters);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
