// Frequency-domain form of SpectraNet's long-tap convolutions (gfx950): the transforms.
//
// Conv1d(Cin -> Cout, k, padding k//2) on L positions (src/applecider/models/spectranet.py:18-20; stage 2 of the
// default configuration: 64 -> 128 channels, k = 251, L = 1024, default_config.toml:104-114) costs 2 B L Cout Cin k FLOP
// directly — 2.16 TFLOP per product at B = 512, three products per training step — but only
//     Y_f[b, co] = sum_ci X_f[b, ci] H_f[ci, co]        for every frequency f of a length-N transform, N >= L + k//2
// = N/2 + 1 small complex products (34 GFLOP in all) once the operands are in the frequency domain.  The complex
// products run as real batched products on the matrix cores (ac_gemm_batched; a complex number = two adjacent real
// columns); this file holds the transforms around them, all on the in-LDS core of ac_fft_core.h:
//   ac_fft_rows_fwd   channels-last real rows [B, L, C] (fp32, or a (hi, lo) bf16 plane pair) -> spectrum [F][B][2C]
//   ac_fft_rows_inv   spectrum [F][B][2C] -> channels-last real rows (cropped at an offset, + bias, or accumulated)
//   ac_fft_taps_fwd   taps w[co][t][ci] (flipped: a correlation) -> H' [F][2 Cout][2 Cin], the real block form
//                     [[Hr, -Hi], [Hi, Hr]]: the [N][K] operand of the forward product and, read as [K][N], the
//                     operand of the input-gradient product (its transpose is the block form of conj H)
//   ac_fft_taps_inv   M' [F][2 Cout][2 Cin] = G'^T X' per frequency -> conj(X_f) G_f = (M'_rr + M'_ii) + i (M'_ir - M'_ri)
//                     -> inverse transform, cropped to the k taps, flipped, added into dw[co][t][ci]
// Real sequences are transformed two at a time (z = x1 + i x2; adjacent channels are an 8-byte load) and untangled
// through Z[f] and conj Z[N - f]; a workgroup of 512 threads holds 8 complex sequences (16 channels) of N <= 2048 points
// (148 KB of LDS at N = 2048).  fp32 throughout: 2-5e-7 of the direct convolution (profiles/r03_fftconv_probe.txt).
#include "ac_common.h"
#include "ac_fft_core.h"
#include <type_traits>

namespace {

using namespace acfft;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FFT_SEQ = 8;        // complex sequences per workgroup
constexpr int FFT_THREADS = 512;

struct TwTable {
    const ac_c2 *tw;   // all levels back to back: level e (transform size N >> e) at element N - (N >> e)
    int n;
    __device__ __forceinline__ ac_c2 operator()(int e, int j) const { return ac_gload<ac_c2>(tw + (n - (n >> e)) + j); }
};

// One pass over the FFT_SEQ sequences of the workgroup.  A thread keeps PASS_U work items in flight: the LDS reads and
// twiddle loads of all of them are issued before the first butterfly (at 148 KB of LDS a CU holds ONE workgroup = two
// waves per SIMD, so nothing else hides those latencies).
// Kernels come in two register budgets: U = 4 (N >= 1024: LDS admits one or two workgroups per CU anyway) and
// U = 1 (N <= 512: ~52 VGPRs, four workgroups per CU hide the latencies between them).
template <int R, bool INVERSE, int PASS_U>
__device__ __forceinline__ void fft_pass_all(ac_c2 *buf, const TwTable &tw, int logn, int arg) {
    const int per = 1 << (logn - R), pitch = seq_pitch(logn), total = FFT_SEQ * per;
    __syncthreads();
    for (int w0 = threadIdx.x; w0 < total; w0 += FFT_THREADS * PASS_U) {
        ac_c2 v[PASS_U][1 << R], tws[PASS_U][R];
        PassItem it[PASS_U];
        ac_c2 *seq[PASS_U];
#pragma unroll
        for (int j = 0; j < PASS_U; ++j) {
            int w = w0 + j * FFT_THREADS;
            w = w < total ? w : w0;                       // (a clamped duplicate: computed, never stored)
            seq[j] = buf + (w >> (logn - R)) * pitch;
            const int u = w & (per - 1);
            it[j] = INVERSE ? dit_item<R>(arg, u) : dif_item<R>(logn, arg, u);
            if (INVERSE)
                dit_twiddles<R>(tw, logn, arg, it[j].i0, tws[j]);
            else
                dif_twiddles<R>(tw, arg, it[j].i0, tws[j]);
            pass_load<R>(seq[j], it[j], v[j]);
        }
#pragma unroll
        for (int j = 0; j < PASS_U; ++j) {
            if (INVERSE)
                dit_butterflies<R>(v[j], tws[j]);
            else
                dif_butterflies<R>(v[j], tws[j]);
        }
#pragma unroll
        for (int j = 0; j < PASS_U; ++j)
            if (w0 + j * FFT_THREADS < total) pass_store<R>(seq[j], it[j], v[j]);
    }
}

// natural order in -> bit-reversed order out
template <int U>
__device__ __forceinline__ void fft_forward(ac_c2 *buf, const TwTable &tw, int logn) {
    const int r0 = first_r(logn);
    if (r0 == 1) fft_pass_all<1, false, U>(buf, tw, logn, 0);
    if (r0 == 2) fft_pass_all<2, false, U>(buf, tw, logn, 0);
    for (int s = r0; s < logn; s += 3) fft_pass_all<3, false, U>(buf, tw, logn, s);
    __syncthreads();
}
// bit-reversed order in -> natural order out (unnormalised inverse)
template <int U>
__device__ __forceinline__ void fft_inverse(ac_c2 *buf, const TwTable &tw, int logn) {
    const int r0 = first_r(logn);
    int lh = 0;
    for (; lh + 3 <= logn - r0; lh += 3) fft_pass_all<3, true, U>(buf, tw, logn, lh);
    if (r0 == 1) fft_pass_all<1, true, U>(buf, tw, logn, lh);
    if (r0 == 2) fft_pass_all<2, true, U>(buf, tw, logn, lh);
    __syncthreads();
}

// position of the idx-th half-spectrum entry in the bit-reversed image: the even positions are the frequencies
// below N / 2, position 1 is N / 2
__device__ __forceinline__ int half_pos(int idx, int halfn) { return idx == halfn ? 1 : 2 * idx; }

// workgroups that share a batch row (its channel groups read / write the same 128-byte lines) sit on one XCD
__device__ __forceinline__ void map_block(int bid, int B, int G, int &b, int &g) {
    if ((B & 7) == 0) {
        const int xcd = bid & 7, slot = bid >> 3;
        g = slot % G;
        b = (slot / G) * 8 + xcd;
    } else {
        g = bid % G;
        b = bid / G;
    }
}

struct RowsParams {
    const void *src;       // fwd: real rows (fp32, or the hi plane); inv: spectrum
    const void *src_lo;    // fwd: lo plane (bf16) or null
    float *dst;            // fwd: spectrum; inv: real rows
    const ac_c2 *tw;
    const float *bias;     // inv only, nullable, indexed by channel
    int64_t row_stride, batch_stride;   // of the real tensor (elements)
    int col_off;           // first channel column of the real tensor
    int B, C, L;           // C channels transformed (C % 16 == 0), L valid rows
    int shift;             // real row l <-> sequence index l + shift
    int logn, accumulate;
};

template <bool PLANES, int U>
__global__ __launch_bounds__(FFT_THREADS, U == 1 ? 8 : 2) void fft_rows_fwd_kernel(RowsParams p) {
    constexpr int LB = U == 1 ? 2 : 8, SB = U == 1 ? 2 : 4;     // global loads / LDS reads in flight per thread
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const int N = 1 << p.logn, halfn = N >> 1, pitch = seq_pitch(p.logn);
    int b, g;
    map_block(blockIdx.x, p.B, p.C >> 4, b, g);
    const int q = threadIdx.x & 7, c0 = g * 16;
    const TwTable tw{p.tw, N};
    ac_c2 *seq = fbuf + q * pitch;
    const int64_t off = (int64_t)b * p.batch_stride + p.col_off + c0 + 2 * q;
    constexpr int RS = FFT_THREADS / 8;     // rows per sweep of the workgroup
    // zeros where no row lands, then the rows with eight loads in flight per thread
    for (int n = threadIdx.x >> 3; n < N; n += RS)
        if (n < p.shift || n >= p.shift + p.L) seq[phys(n)] = ac_c2{0.f, 0.f};
    for (int l0 = threadIdx.x >> 3; l0 < p.L; l0 += RS * LB) {
        ac_c2 z[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int l = l0 + u * RS;
            const int64_t a = off + (int64_t)(l < p.L ? l : l0) * p.row_stride;
            if (PLANES) {
                const unsigned h = ac_gload<unsigned>((const unsigned short *)p.src + a);
                const unsigned lo = ac_gload<unsigned>((const unsigned short *)p.src_lo + a);
                z[u] = ac_c2{ac_h2f((unsigned short)(h & 0xFFFFu)) + ac_h2f((unsigned short)(lo & 0xFFFFu)),
                             ac_h2f((unsigned short)(h >> 16)) + ac_h2f((unsigned short)(lo >> 16))};
            } else {
                z[u] = ac_gload<ac_c2>((const float *)p.src + a);
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (l0 + u * RS < p.L) seq[phys(l0 + u * RS + p.shift)] = z[u];
    }
    fft_forward<U>(fbuf, tw, p.logn);
    float *dst = p.dst + (int64_t)b * (2 * p.C) + 2 * (c0 + 2 * q);
    const int64_t fstride = (int64_t)p.B * (2 * p.C);
    for (int idx0 = threadIdx.x >> 3; idx0 <= halfn; idx0 += RS * SB) {
        ac_c2 zf[SB], zn[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = idx0 + u * RS, i = half_pos(idx <= halfn ? idx : idx0, halfn);
            zf[u] = seq[phys(i)];
            zn[u] = seq[phys(partner(i))];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = idx0 + u * RS;
            if (idx > halfn) continue;
            ac_c2 x1, x2;
            untangle(zf[u], zn[u], x1, x2);
            *(f32x4 *)(dst + (int64_t)brev(half_pos(idx, halfn), p.logn) * fstride) = f32x4{x1[0], x1[1], x2[0], x2[1]};
        }
    }
}

template <int U>
__global__ __launch_bounds__(FFT_THREADS, U == 1 ? 8 : 2) void fft_rows_inv_kernel(RowsParams p) {
    constexpr int LB = U == 1 ? 2 : 8;
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const int N = 1 << p.logn, halfn = N >> 1, pitch = seq_pitch(p.logn);
    int b, g;
    map_block(blockIdx.x, p.B, p.C >> 4, b, g);
    const int q = threadIdx.x & 7, c0 = g * 16;
    const TwTable tw{p.tw, N};
    ac_c2 *seq = fbuf + q * pitch;
    const float *src = (const float *)p.src + (int64_t)b * (2 * p.C) + 2 * (c0 + 2 * q);
    const int64_t fstride = (int64_t)p.B * (2 * p.C);
    constexpr int RS = FFT_THREADS / 8;
    for (int idx0 = threadIdx.x >> 3; idx0 <= halfn; idx0 += RS * LB) {
        f32x4 v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int idx = idx0 + u * RS;
            v[u] = ac_gload<f32x4>(src + (int64_t)brev(half_pos(idx <= halfn ? idx : idx0, halfn), p.logn) * fstride);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int idx = idx0 + u * RS;
            if (idx > halfn) continue;
            const int i = half_pos(idx, halfn);
            ac_c2 zf, zn;
            tangle(ac_c2{v[u][0], v[u][1]}, ac_c2{v[u][2], v[u][3]}, zf, zn);
            seq[phys(i)] = zf;
            if (i > 1) seq[phys(partner(i))] = zn;
        }
    }
    fft_inverse<U>(fbuf, tw, p.logn);
    const float inv = 1.0f / (float)N;
    float *dst = p.dst + (int64_t)b * p.batch_stride + p.col_off + c0 + 2 * q;
    ac_c2 bias2 = {0.f, 0.f};
    if (p.bias) bias2 = ac_gload<ac_c2>(p.bias + c0 + 2 * q);
    for (int l0 = threadIdx.x >> 3; l0 < p.L; l0 += RS * LB) {
        ac_c2 o[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int l = l0 + u * RS < p.L ? l0 + u * RS : l0;
            o[u] = seq[phys(l + p.shift)] * inv + bias2;
            if (p.accumulate) o[u] += ac_gload<ac_c2>(dst + (int64_t)l * p.row_stride);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u)
            if (l0 + u * RS < p.L) *(ac_c2 *)(dst + (int64_t)(l0 + u * RS) * p.row_stride) = o[u];
    }
}

struct TapsParams {
    const float *src;      // fwd: taps [Cout][k][Cin]; inv: M' [F][2 Cout][2 Cin]
    float *dst;            // fwd: H' [F][2 Cout][2 Cin]; inv: dw [Cout][k][Cin] (+=)
    const ac_c2 *tw;
    int Cout, Cin, k, logn;
};

// workgroup = (co, 16 input channels): h[m] = w[co][k - 1 - m][ci], pairs of ci transformed together
template <int U>
__global__ __launch_bounds__(FFT_THREADS, U == 1 ? 8 : 2) void fft_taps_fwd_kernel(TapsParams p) {
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const int N = 1 << p.logn, halfn = N >> 1, pitch = seq_pitch(p.logn);
    const int G = p.Cin >> 4, co = blockIdx.x / G, ci0 = (blockIdx.x % G) * 16;
    const int q = threadIdx.x & 7;
    const TwTable tw{p.tw, N};
    ac_c2 *seq = fbuf + q * pitch;
    const float *w = p.src + (int64_t)co * p.k * p.Cin + ci0 + 2 * q;
    for (int m = threadIdx.x >> 3; m < N; m += FFT_THREADS / 8) {
        const bool in = m < p.k;
        const ac_c2 z = ac_gload<ac_c2>(w + (int64_t)(in ? p.k - 1 - m : 0) * p.Cin);
        seq[phys(m)] = in ? z : ac_c2{0.f, 0.f};
    }
    fft_forward<U>(fbuf, tw, p.logn);
    // H'[f][(co, re)][(ci, re)] = Hr, [(co, re)][(ci, im)] = -Hi, [(co, im)][(ci, re)] = Hi, [(co, im)][(ci, im)] = Hr
    const int ld = 2 * p.Cin;
    float *o = p.dst + (int64_t)(2 * co) * ld + 2 * (ci0 + 2 * q);
    const int64_t fstride = (int64_t)(2 * p.Cout) * ld;
    for (int idx = threadIdx.x >> 3; idx <= halfn; idx += FFT_THREADS / 8) {
        const int i = half_pos(idx, halfn), f = brev(i, p.logn);
        ac_c2 h1, h2;
        untangle(seq[phys(i)], seq[phys(partner(i))], h1, h2);
        float *of = o + (int64_t)f * fstride;
        *(f32x4 *)of = f32x4{h1[0], -h1[1], h2[0], -h2[1]};          // row (co, re)
        *(f32x4 *)(of + ld) = f32x4{h1[1], h1[0], h2[1], h2[0]};     // row (co, im)
    }
}

// workgroup = (co, 16 input channels): conj(X_f) G_f for the pairs of ci from the rows (co, re), (co, im) of M'
template <int U>
__global__ __launch_bounds__(FFT_THREADS, U == 1 ? 8 : 2) void fft_taps_inv_kernel(TapsParams p) {
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const int N = 1 << p.logn, halfn = N >> 1, pitch = seq_pitch(p.logn);
    const int G = p.Cin >> 4, co = blockIdx.x / G, ci0 = (blockIdx.x % G) * 16;
    const int q = threadIdx.x & 7;
    const TwTable tw{p.tw, N};
    ac_c2 *seq = fbuf + q * pitch;
    const int ld = 2 * p.Cin;
    const float *mre = p.src + (int64_t)(2 * co) * ld + 2 * (ci0 + 2 * q);   // row (co, re); row (co, im) = + ld
    const int64_t fstride = (int64_t)(2 * p.Cout) * ld;
    for (int idx = threadIdx.x >> 3; idx <= halfn; idx += FFT_THREADS / 8) {
        const int i = half_pos(idx, halfn), f = brev(i, p.logn);
        const float *mf = mre + (int64_t)f * fstride;
        // r = [M'_rr(ci) M'_ri(ci) M'_rr(ci+1) M'_ri(ci+1)], m = [M'_ir M'_ii ...]
        const f32x4 r = ac_gload<f32x4>(mf), m = ac_gload<f32x4>(mf + ld);
        ac_c2 zf, zn;
        tangle(ac_c2{r[0] + m[1], m[0] - r[1]}, ac_c2{r[2] + m[3], m[2] - r[3]}, zf, zn);
        seq[phys(i)] = zf;
        if (i > 1) seq[phys(partner(i))] = zn;
    }
    fft_inverse<U>(fbuf, tw, p.logn);
    const float inv = 1.0f / (float)N;
    float *dw = p.dst + (int64_t)co * p.k * p.Cin + ci0 + 2 * q;
    for (int t = threadIdx.x >> 3; t < p.k; t += FFT_THREADS / 8) {
        ac_c2 *d = (ac_c2 *)(dw + (int64_t)t * p.Cin);
        *d += seq[phys(p.k - 1 - t)] * inv;
    }
}

template <typename K, typename P>
int fft_launch(K kernel, int blocks, const P &p, int logn, hipStream_t stream) {
    const size_t lds = (size_t)FFT_SEQ * seq_pitch(logn) * sizeof(ac_c2);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(FFT_THREADS), lds, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

bool logn_ok(int logn) { return logn >= 6 && logn <= 11; }

}  // namespace

extern "C" int ac_fft_rows_fwd(const void *src, const void *src_lo, int64_t batch_stride, int64_t row_stride,
                               int32_t col_off, int32_t B, int32_t L, int32_t C, int32_t shift, int32_t logn,
                               const float *tw, float *spec, ac_stream_t stream) {
    if (!src || !tw || !spec || B <= 0 || L <= 0 || C <= 0 || (C % 16) || !logn_ok(logn)) return AC_EINVAL;
    if (shift < 0 || L + shift > (1 << logn)) return AC_EINVAL;
    const uintptr_t amask = src_lo ? 3u : 7u;   // one 4-byte (two bf16) or 8-byte (two fp32) load per channel pair
    if ((row_stride % 2) || (batch_stride % 2) || (col_off % 2) || ((uintptr_t)src & amask) ||
        ((uintptr_t)src_lo & amask) || !ac_aligned16(spec) || ((uintptr_t)tw & 7u))
        return AC_EALIGN;
    RowsParams p;
    p.src = src; p.src_lo = src_lo; p.dst = spec; p.tw = (const ac_c2 *)tw; p.bias = nullptr;
    p.row_stride = row_stride; p.batch_stride = batch_stride; p.col_off = col_off;
    p.B = B; p.C = C; p.L = L; p.shift = shift; p.logn = logn; p.accumulate = 0;
    const int blocks = B * (C / 16);
    hipStream_t st = (hipStream_t)stream;
    if (src_lo) return logn >= 10 ? fft_launch(fft_rows_fwd_kernel<true, 4>, blocks, p, logn, st)
                                  : fft_launch(fft_rows_fwd_kernel<true, 1>, blocks, p, logn, st);
    return logn >= 10 ? fft_launch(fft_rows_fwd_kernel<false, 4>, blocks, p, logn, st)
                      : fft_launch(fft_rows_fwd_kernel<false, 1>, blocks, p, logn, st);
}

extern "C" int ac_fft_rows_inv(const float *spec, int32_t B, int32_t C, int32_t logn, const float *tw, float *dst,
                               int64_t batch_stride, int64_t row_stride, int32_t col_off, int32_t L, int32_t shift,
                               const float *bias, int32_t accumulate, ac_stream_t stream) {
    if (!spec || !tw || !dst || B <= 0 || L <= 0 || C <= 0 || (C % 16) || !logn_ok(logn)) return AC_EINVAL;
    if (shift < 0 || L + shift > (1 << logn)) return AC_EINVAL;
    if ((row_stride % 2) || (batch_stride % 2) || (col_off % 2) || ((uintptr_t)dst & 7u) || !ac_aligned16(spec) ||
        ((uintptr_t)tw & 7u) || (bias && ((uintptr_t)bias & 7u)))
        return AC_EALIGN;
    RowsParams p;
    p.src = spec; p.src_lo = nullptr; p.dst = dst; p.tw = (const ac_c2 *)tw; p.bias = bias;
    p.row_stride = row_stride; p.batch_stride = batch_stride; p.col_off = col_off;
    p.B = B; p.C = C; p.L = L; p.shift = shift; p.logn = logn; p.accumulate = accumulate;
    return logn >= 10 ? fft_launch(fft_rows_inv_kernel<4>, B * (C / 16), p, logn, (hipStream_t)stream)
                      : fft_launch(fft_rows_inv_kernel<1>, B * (C / 16), p, logn, (hipStream_t)stream);
}

extern "C" int ac_fft_taps_fwd(const float *w, int32_t Cout, int32_t Cin, int32_t k, int32_t logn, const float *tw,
                               float *hblock, ac_stream_t stream) {
    if (!w || !tw || !hblock || Cout <= 0 || Cin <= 0 || (Cin % 16) || k <= 0 || !logn_ok(logn) || k > (1 << logn))
        return AC_EINVAL;
    if (((uintptr_t)w & 7u) || !ac_aligned16(hblock) || ((uintptr_t)tw & 7u)) return AC_EALIGN;
    TapsParams p;
    p.src = w; p.dst = hblock; p.tw = (const ac_c2 *)tw; p.Cout = Cout; p.Cin = Cin; p.k = k; p.logn = logn;
    return logn >= 10 ? fft_launch(fft_taps_fwd_kernel<4>, Cout * (Cin / 16), p, logn, (hipStream_t)stream)
                      : fft_launch(fft_taps_fwd_kernel<1>, Cout * (Cin / 16), p, logn, (hipStream_t)stream);
}

extern "C" int ac_fft_taps_inv(const float *m, int32_t Cout, int32_t Cin, int32_t k, int32_t logn, const float *tw,
                               float *dw, ac_stream_t stream) {
    if (!m || !tw || !dw || Cout <= 0 || Cin <= 0 || (Cin % 16) || k <= 0 || !logn_ok(logn) || k > (1 << logn))
        return AC_EINVAL;
    if (!ac_aligned16(m) || ((uintptr_t)tw & 7u) || ((uintptr_t)dw & 7u)) return AC_EALIGN;
    TapsParams p;
    p.src = m; p.dst = dw; p.tw = (const ac_c2 *)tw; p.Cout = Cout; p.Cin = Cin; p.k = k; p.logn = logn;
    return logn >= 10 ? fft_launch(fft_taps_inv_kernel<4>, Cout * (Cin / 16), p, logn, (hipStream_t)stream)
                      : fft_launch(fft_taps_inv_kernel<1>, Cout * (Cin / 16), p, logn, (hipStream_t)stream);
}
