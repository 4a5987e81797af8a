// Frequency-domain form of SpectraNet's long-tap convolutions (gfx950): the transforms.
//
// Conv1d(Cin -> Cout, k, padding k//2) on L positions (src/applecider/models/spectranet.py:18-20; stage 2 of the
// default configuration: 64 -> 128 channels, k = 251, L = 1024, default_config.toml:104-114) costs 2 B L Cout Cin k FLOP
// directly — 2.16 TFLOP per product at B = 512, three products per training step — but only
//     Y_f[b, co] = sum_ci X_f[b, ci] H_f[ci, co]        for every frequency f of a length-N transform
// = N/2 + 1 small complex products (34 GFLOP in all) once the operands are in the frequency domain.  The complex
// products run as real batched products on the matrix cores (ac_gemm_batched; a complex number = two adjacent real
// columns); this file holds the transforms around them, all on the in-LDS core of ac_fft_core.h:
//   ac_fft_rows_fwd   channels-last real rows [B, L, C] (fp32, or a (hi, lo) bf16 plane pair) -> spectrum [F][B*blocks][2C]
//   ac_fft_rows_inv   spectrum -> channels-last real rows (cropped at an offset, + bias, or accumulated)
//   ac_fft_taps_fwd   taps w[co][t][ci] (flipped: a correlation) -> H' [F][2 Cout][2 Cin], the real block form
//                     [[Hr, -Hi], [Hi, Hr]]: the [N][K] operand of the forward product and, read as [K][N], the
//                     operand of the input-gradient product (its transpose is the block form of conj H)
//   ac_fft_taps_inv   M' [F][2 Cout][2 Cin] = G'^T X' per frequency -> conj(X_f) G_f = (M'_rr + M'_ii) + i (M'_ir - M'_ri)
//                     -> inverse transform, cropped to the k taps, flipped, added into dw[co][t][ci]
// A sample is either ONE sequence (N >= L + k//2: the circular wrap falls into the zero padding) or `blocks`
// overlapping windows of N points that advance by V = N - k + 1 rows (overlap-save): shorter transforms, the same
// spectrum bytes, four workgroups per CU instead of one.
// Real sequences are transformed two at a time (z = x1 + i x2; adjacent channels are an 8-byte load) and untangled
// through Z[f] and conj Z[N - f]; a workgroup of 512 threads holds 8 complex sequences (16 channels; 148 KB of LDS at
// N = 2048) or, for N <= 128, 32 of them (64 channels).  fp32 throughout: 2-5e-7 of the direct convolution.
#include "ac_common.h"
#include "ac_fft_core.h"
#include <type_traits>

// Barrier of the transform kernels (one name so that tools/fft_debug.patch, the diagnostic build of DESIGN section 7-9,
// can instrument it).
#define FFT_SYNC() __syncthreads()

namespace {

using namespace acfft;
typedef float f32x4 __attribute__((ext_vector_type(4)));


struct TwTable {
    const ac_c2 *tw;   // all levels back to back: level e (transform size N >> e) at element N - (N >> e)
    int n;
    __device__ __forceinline__ ac_c2 operator()(int e, int j) const { return ac_gload<ac_c2>(tw + (n - (n >> e)) + j); }
};

// One pass over the SEQ sequences of the workgroup.  A thread keeps PASS_U work items in flight: the LDS reads and
// twiddle loads of all of them are issued before the first butterfly (at 148 KB of LDS a CU holds ONE workgroup = two
// waves per SIMD, so nothing else hides those latencies).
// Kernels come in two register budgets: U = 4 (N >= 1024: LDS admits one or two workgroups per CU anyway) and
// U = 1 (N <= 512: ~52 VGPRs, four workgroups per CU hide the latencies between them).
template <int R, bool INVERSE, int SEQ, int PASS_U, int FFT_THREADS>
__device__ __forceinline__ void fft_pass_all(ac_c2 *buf, const TwTable &tw, int logn, int radix3, int pitch, int arg) {
    // logn = log2 of the power-of-two transform; with radix3 = a every sequence is T = 3^a of them (its pieces)
    const int T = pow3(radix3), per = 1 << (logn - R), total = T * SEQ * per, tb = third_base(1, logn);
    FFT_SYNC();
    for (int w0 = threadIdx.x; w0 < total; w0 += FFT_THREADS * PASS_U) {
        ac_c2 v[PASS_U][1 << R], tws[PASS_U][R];
        PassItem it[PASS_U];
        ac_c2 *seq[PASS_U];
#pragma unroll
        for (int j = 0; j < PASS_U; ++j) {
            int w = w0 + j * FFT_THREADS;
            w = w < total ? w : w0;                       // (a clamped duplicate: computed, never stored)
            const int vs = w >> (logn - R);               // virtual sequence: (sequence, piece)
            const int sq = vs / T;
            seq[j] = buf + sq * pitch + (vs - T * sq) * tb;
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

// the radix-3 stages of an N = 3^a * 2^logn transform over the SEQ sequences (forward: first, the whole sequence
// then its thirds; inverse: last, in the opposite order)
template <bool INVERSE, int SEQ, int FFT_THREADS>
__device__ __forceinline__ void fft_radix3_all(ac_c2 *buf, const TwTable &tw, int logn, int radix3, int pitch) {
    const int M = 1 << logn, N = pow3(radix3) * M;
    const ac_c2 *t3 = tw.tw + M;                              // exp(-2 pi i t / N), t < 2 N / 3, behind the M-point levels
    auto tw3 = [&](int t) { return ac_gload<ac_c2>(t3 + t); };
    for (int k = 0; k < radix3; ++k) {
        const int st = INVERSE ? radix3 - 1 - k : k;          // stage 0 splits the whole sequence, stage 1 its thirds
        const int s3 = st == 0 ? N / 3 : N / 9, pieces = st == 0 ? 1 : 3, tstride = st == 0 ? 1 : 3;
        const int per = pieces * s3;                          // butterflies per sequence in this stage (= N / 3)
        FFT_SYNC();
        for (int w = threadIdx.x; w < SEQ * per; w += FFT_THREADS) {
            const int sq = w / per, u = w - sq * per, pc = u / s3, j = u - pc * s3;
            ac_c2 *seq = buf + sq * pitch + phys(pc * 3 * s3);
            if (INVERSE)
                dit3_item(seq, tw3, s3, tstride, j);
            else
                dif3_item(seq, tw3, s3, tstride, j);
        }
    }
}

// natural order in -> bit-reversed order out (per third when radix3)
template <int SEQ, int U, int NT>
__device__ __forceinline__ void fft_forward(ac_c2 *buf, const TwTable &tw, int logn, int radix3, int pitch) {
    if (radix3) fft_radix3_all<false, SEQ, NT>(buf, tw, logn, radix3, pitch);
    const int r0 = first_r(logn);
    if (r0 == 1) fft_pass_all<1, false, SEQ, U, NT>(buf, tw, logn, radix3, pitch, 0);
    if (r0 == 2) fft_pass_all<2, false, SEQ, U, NT>(buf, tw, logn, radix3, pitch, 0);
    for (int s = r0; s < logn; s += 3) fft_pass_all<3, false, SEQ, U, NT>(buf, tw, logn, radix3, pitch, s);
    FFT_SYNC();
}
// bit-reversed order in -> natural order out (unnormalised inverse)
template <int SEQ, int U, int NT>
__device__ __forceinline__ void fft_inverse(ac_c2 *buf, const TwTable &tw, int logn, int radix3, int pitch) {
    const int r0 = first_r(logn);
    int lh = 0;
    for (; lh + 3 <= logn - r0; lh += 3) fft_pass_all<3, true, SEQ, U, NT>(buf, tw, logn, radix3, pitch, lh);
    if (r0 == 1) fft_pass_all<1, true, SEQ, U, NT>(buf, tw, logn, radix3, pitch, lh);
    if (r0 == 2) fft_pass_all<2, true, SEQ, U, NT>(buf, tw, logn, radix3, pitch, lh);
    if (radix3) fft_radix3_all<true, SEQ, NT>(buf, tw, logn, radix3, pitch);
    FFT_SYNC();
}

// workgroups that share a spectrum row (its channel groups read / write the same 128-byte lines) sit on one XCD
__device__ __forceinline__ void map_block(int bid, int rows, int G, int &row, int &g) {
    if ((rows & 7) == 0) {
        const int xcd = bid & 7, slot = bid >> 3;
        g = slot % G;
        row = (slot / G) * 8 + xcd;
    } else {
        g = bid % G;
        row = bid / G;
    }
}

struct RowsParams {
    ac_fft_rows_desc d;
};

// LDS image <-> spectrum [F][rows][2C]: every lane moves one (frequency, channel pair) = 16 bytes
template <int SEQ, int SB, int FFT_THREADS>
__device__ __forceinline__ void spectrum_store(const ac_c2 *seq, float *dst, int64_t fstride, int logn, int radix3) {
    const int halfn = (pow3(radix3) << logn) >> 1;
    constexpr int RS = FFT_THREADS / SEQ;
    for (int e0 = threadIdx.x / SEQ; e0 <= halfn; e0 += RS * SB) {
        ac_c2 zf[SB], zn[SB];
        int fr[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int e = e0 + u * RS;
            int pos, ppos;
            bool pair;
            half_entry(e <= halfn ? e : e0, logn, radix3, pos, ppos, fr[u], pair);
            zf[u] = seq[phys(pos)];
            zn[u] = seq[phys(ppos)];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (e0 + u * RS > halfn) continue;
            ac_c2 x1, x2;
            untangle(zf[u], zn[u], x1, x2);
            *(f32x4 *)(dst + (int64_t)fr[u] * fstride) = f32x4{x1[0], x1[1], x2[0], x2[1]};
        }
    }
}

// A rows workgroup owns its CU (fft_launch), so nothing else hides its memory latency: it walks SEVERAL tiles
// (tile = SEQ sequences of one spectrum row; tiles t, t + gridDim.x, ... — gridDim.x is a multiple of 8, so all of them
// sit on the workgroup's XCD) and the global loads of the next tile are issued, into registers, before the passes of
// the current one; the spectrum / row stores of a tile drain while the next one is transformed.
struct RawPair {
    unsigned a, b;     // two fp32 (bit patterns) or the packed (hi, hi) / (lo, lo) bf16 pairs of a channel pair
};

template <int SEQ, int U, bool PLANES, int FFT_THREADS>
__global__ __launch_bounds__(FFT_THREADS, U == 1 && FFT_THREADS >= 1024 ? 4 : 2) void fft_rows_fwd_kernel(RowsParams p) {
    constexpr int SB = U == 1 ? 2 : 4;                           // LDS reads in flight per thread in the spectrum store
    constexpr int RS = FFT_THREADS / SEQ;                        // rows per sweep of the workgroup
    constexpr int EPT = U == 1 ? 16 : 32;                        // rows of one sequence per thread: N / RS at most
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const ac_fft_rows_desc &d = p.d;
    const int N = pow3(d.radix3) << d.logn, pitch = seq_pitch_n(N, SEQ);
    const int rows = d.B * d.blocks, G = d.C / (2 * SEQ), total = rows * G, W = gridDim.x;
    const int q = threadIdx.x & (SEQ - 1), r0 = threadIdx.x / SEQ;
    const TwTable tw{(const ac_c2 *)d.tw, 1 << d.logn};
    ac_c2 *seq = fbuf + q * pitch;
    const int64_t fstride = (int64_t)rows * (2 * d.C);
    int tile = blockIdx.x;
    if (tile >= total) return;

    // sequence indices [n0, n1) of tile `t` take the rows l = rv + n - shift; everything else is zero
    int row, c0, n0, n1;
    int64_t off;
    auto locate = [&](int t) __attribute__((always_inline)) {
        int g;
        map_block(t, rows, G, row, g);
        const int b = row / d.blocks, rv = (row - b * d.blocks) * d.block_step;
        c0 = g * 2 * SEQ;
        n0 = d.shift - rv;
        n1 = d.L + d.shift - rv;
        n0 = n0 > d.n_lo ? n0 : d.n_lo;
        n1 = n1 < d.n_hi ? n1 : d.n_hi;
        n1 = n1 < N ? n1 : N;
        n1 = n1 > n0 ? n1 : n0;
        off = (int64_t)b * d.batch_stride + (int64_t)(rv + n0 - d.shift) * d.row_stride + d.col_off + c0 + 2 * q;
    };
    RawPair z[EPT];
    auto issue = [&]() __attribute__((always_inline)) {
        const int cnt = n1 - n0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (u * RS >= cnt) break;
            const int j = r0 + u * RS;
            const int64_t a = off + (int64_t)(j < cnt ? j : 0) * d.row_stride;     // clamped; masked when written to LDS
            if (PLANES) {
                z[u].a = ac_gload<unsigned>((const unsigned short *)d.rows + a);
                z[u].b = ac_gload<unsigned>((const unsigned short *)d.rows_lo + a);
            } else {
                const ac_c2 v = ac_gload<ac_c2>((const float *)d.rows + a);
                z[u].a = __float_as_uint(v[0]);
                z[u].b = __float_as_uint(v[1]);
            }
        }
    };
    locate(tile);
    issue();
    for (;;) {
        const int cnt = n1 - n0;
        for (int n = r0; n < N; n += RS)
            if (n < n0 || n >= n1) seq[phys(n)] = ac_c2{0.f, 0.f};
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (u * RS >= cnt) break;
            const int j = r0 + u * RS;
            if (j >= cnt) continue;
            ac_c2 v;
            if (PLANES)
                v = ac_c2{ac_h2f((unsigned short)(z[u].a & 0xFFFFu)) + ac_h2f((unsigned short)(z[u].b & 0xFFFFu)),
                          ac_h2f((unsigned short)(z[u].a >> 16)) + ac_h2f((unsigned short)(z[u].b >> 16))};
            else
                v = ac_c2{__uint_as_float(z[u].a), __uint_as_float(z[u].b)};
            seq[phys(n0 + j)] = v;
        }
        float *dst = d.spec + (int64_t)row * (2 * d.C) + 2 * (c0 + 2 * q);
        tile += W;
        const bool more = tile < total;
        if (more) {
            locate(tile);
            issue();
        }
        fft_forward<SEQ, U, FFT_THREADS>(fbuf, tw, d.logn, d.radix3, pitch);
        spectrum_store<SEQ, SB, FFT_THREADS>(seq, dst, fstride, d.logn, d.radix3);
        if (!more) break;
        FFT_SYNC();                                         // the image is rewritten
    }
}

// half spectra of the pairs (y1, y2) -> Z in the bit-reversed image; `load(f)` returns [y1.re y1.im y2.re y2.im]
template <int SEQ, int LB, int FFT_THREADS, typename LOAD>
__device__ __forceinline__ void spectrum_load(ac_c2 *seq, int logn, int radix3, LOAD load) {
    const int halfn = (pow3(radix3) << logn) >> 1;
    constexpr int RS = FFT_THREADS / SEQ;
    for (int e0 = threadIdx.x / SEQ; e0 <= halfn; e0 += RS * LB) {
        f32x4 v[LB];
        int pos[LB], ppos[LB];
        bool pair[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int e = e0 + u * RS;
            int f;
            half_entry(e <= halfn ? e : e0, logn, radix3, pos[u], ppos[u], f, pair[u]);
            v[u] = load(f);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            if (e0 + u * RS > halfn) continue;
            ac_c2 zf, zn;
            tangle(ac_c2{v[u][0], v[u][1]}, ac_c2{v[u][2], v[u][3]}, zf, zn);
            seq[phys(pos[u])] = zf;
            if (pair[u]) seq[phys(ppos[u])] = zn;
        }
    }
}

template <int SEQ, int U, int FFT_THREADS>
__global__ __launch_bounds__(FFT_THREADS, U == 1 && FFT_THREADS >= 1024 ? 4 : 2) void fft_rows_inv_kernel(RowsParams p) {
    constexpr int LB = U == 1 ? 2 : 8;
    constexpr int RS = FFT_THREADS / SEQ;
    constexpr int EPS = U == 1 ? 9 : 17;                          // half-spectrum entries per thread: (N / 2 + 1) / RS at most
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const ac_fft_rows_desc &d = p.d;
    const int N = pow3(d.radix3) << d.logn, halfn = N >> 1, pitch = seq_pitch_n(N, SEQ);
    const int rows = d.B * d.blocks, G = d.C / (2 * SEQ), total = rows * G, W = gridDim.x;
    const int q = threadIdx.x & (SEQ - 1), r0 = threadIdx.x / SEQ;
    const TwTable tw{(const ac_c2 *)d.tw, 1 << d.logn};
    ac_c2 *seq = fbuf + q * pitch;
    const int64_t fstride = (int64_t)rows * (2 * d.C);
    const float inv = 1.0f / (float)N;
    int tile = blockIdx.x;
    if (tile >= total) return;

    int row, c0;
    auto locate = [&](int t) __attribute__((always_inline)) {
        int g;
        map_block(t, rows, G, row, g);
        c0 = g * 2 * SEQ;
    };
    f32x4 v[EPS];
    // `rr` = r0 behind an optimisation barrier, renewed per tile: the entry -> (position, partner, frequency) maps
    // depend on the thread only, and hoisted out of the tile loop they would sit in ~70 registers through every pass
    int rr = r0;
    auto issue = [&]() __attribute__((always_inline)) {
        const float *src = d.spec + (int64_t)row * (2 * d.C) + 2 * (c0 + 2 * q);
#pragma unroll
        for (int u = 0; u < EPS; ++u) {
            if (u * RS > halfn) break;
            const int e = rr + u * RS;
            int pos, ppos, f;
            bool pair;
            half_entry(e <= halfn ? e : 0, d.logn, d.radix3, pos, ppos, f, pair);
            v[u] = ac_gload<f32x4>(src + (int64_t)f * fstride);
        }
    };
    locate(tile);
    issue();
    for (;;) {
        asm volatile("" : "+v"(rr));
#pragma unroll
        for (int u = 0; u < EPS; ++u) {
            if (u * RS > halfn) break;
            const int e = rr + u * RS;
            if (e > halfn) continue;
            int pos, ppos, f;
            bool pair;
            half_entry(e, d.logn, d.radix3, pos, ppos, f, pair);
            ac_c2 zf, zn;
            tangle(ac_c2{v[u][0], v[u][1]}, ac_c2{v[u][2], v[u][3]}, zf, zn);
            seq[phys(pos)] = zf;
            if (pair) seq[phys(ppos)] = zn;
        }
        const int b = row / d.blocks, rv = (row - b * d.blocks) * d.block_step, cc0 = c0;
        tile += W;
        const bool more = tile < total;
        if (more) {
            locate(tile);
            issue();
        }
        fft_inverse<SEQ, U, FFT_THREADS>(fbuf, tw, d.logn, d.radix3, pitch);
        // this block's output rows rv + j, j < cnt, = sequence index j + shift
        int cnt = d.L - rv;
        if (d.blocks > 1 && cnt > d.block_step) cnt = d.block_step;
        float *dst = (float *)d.rows + (int64_t)b * d.batch_stride + (int64_t)rv * d.row_stride + d.col_off + cc0 + 2 * q;
        ac_c2 bias2 = {0.f, 0.f};
        if (d.bias) bias2 = ac_gload<ac_c2>(d.bias + cc0 + 2 * q);
        for (int j0 = r0; j0 < cnt; j0 += RS * LB) {
            ac_c2 o[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int j = j0 + u * RS < cnt ? j0 + u * RS : j0;
                o[u] = seq[phys(j + d.shift)] * inv + bias2;
                if (d.accumulate) o[u] += ac_gload<ac_c2>(dst + (int64_t)j * d.row_stride);
            }
#pragma unroll
            for (int u = 0; u < LB; ++u)
                if (j0 + u * RS < cnt) *(ac_c2 *)(dst + (int64_t)(j0 + u * RS) * d.row_stride) = o[u];
        }
        if (!more) break;
        FFT_SYNC();                                         // the image is rewritten
    }
}

struct TapsParams {
    const float *src;      // fwd: taps [Cout][k][Cin]; inv: M' [F][2 Cout][2 Cin]
    float *dst;            // fwd: H' [F][2 Cout][2 Cin]; inv: dw [Cout][k][Cin] (+=)
    const ac_c2 *tw;
    int Cout, Cin, k, logn, radix3;
};

// workgroup = (co, 2 SEQ input channels): h[m] = w[co][k - 1 - m][ci], pairs of ci transformed together
template <int SEQ, int U, int FFT_THREADS>
__global__ __launch_bounds__(FFT_THREADS, U == 1 ? 8 : 2) void fft_taps_fwd_kernel(TapsParams p) {
    constexpr int RS = FFT_THREADS / SEQ;
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const int N = pow3(p.radix3) << p.logn, halfn = N >> 1, pitch = seq_pitch_n(N, SEQ);
    const int G = p.Cin / (2 * SEQ), co = blockIdx.x / G, ci0 = (blockIdx.x % G) * 2 * SEQ;
    const int q = threadIdx.x & (SEQ - 1);
    const TwTable tw{p.tw, 1 << p.logn};
    ac_c2 *seq = fbuf + q * pitch;
    const float *w = p.src + (int64_t)co * p.k * p.Cin + ci0 + 2 * q;
    for (int m = threadIdx.x / SEQ; m < N; m += RS) {
        const bool in = m < p.k;
        const ac_c2 z = ac_gload<ac_c2>(w + (int64_t)(in ? p.k - 1 - m : 0) * p.Cin);
        seq[phys(m)] = in ? z : ac_c2{0.f, 0.f};
    }
    fft_forward<SEQ, U, FFT_THREADS>(fbuf, tw, p.logn, p.radix3, pitch);
    // H'[f][(co, re)][(ci, re)] = Hr, [(co, re)][(ci, im)] = -Hi, [(co, im)][(ci, re)] = Hi, [(co, im)][(ci, im)] = Hr
    const int ld = 2 * p.Cin;
    float *o = p.dst + (int64_t)(2 * co) * ld + 2 * (ci0 + 2 * q);
    const int64_t fstride = (int64_t)(2 * p.Cout) * ld;
    for (int e = threadIdx.x / SEQ; e <= halfn; e += RS) {
        int pos, ppos, f;
        bool pair;
        half_entry(e, p.logn, p.radix3, pos, ppos, f, pair);
        ac_c2 h1, h2;
        untangle(seq[phys(pos)], seq[phys(ppos)], h1, h2);
        float *of = o + (int64_t)f * fstride;
        *(f32x4 *)of = f32x4{h1[0], -h1[1], h2[0], -h2[1]};          // row (co, re)
        *(f32x4 *)(of + ld) = f32x4{h1[1], h1[0], h2[1], h2[0]};     // row (co, im)
    }
}

// workgroup = (co, 2 SEQ input channels): conj(X_f) G_f for the pairs of ci from the rows (co, re), (co, im) of M'
template <int SEQ, int U, int FFT_THREADS>
__global__ __launch_bounds__(FFT_THREADS, U == 1 ? 8 : 2) void fft_taps_inv_kernel(TapsParams p) {
    constexpr int LB = U == 1 ? 1 : 4;
    constexpr int RS = FFT_THREADS / SEQ;
    extern __shared__ __attribute__((aligned(16))) ac_c2 fbuf[];
    const int N = pow3(p.radix3) << p.logn, pitch = seq_pitch_n(N, SEQ);
    const int G = p.Cin / (2 * SEQ), co = blockIdx.x / G, ci0 = (blockIdx.x % G) * 2 * SEQ;
    const int q = threadIdx.x & (SEQ - 1);
    const TwTable tw{p.tw, 1 << p.logn};
    ac_c2 *seq = fbuf + q * pitch;
    const int ld = 2 * p.Cin;
    const float *mre = p.src + (int64_t)(2 * co) * ld + 2 * (ci0 + 2 * q);   // row (co, re); row (co, im) = + ld
    const int64_t fstride = (int64_t)(2 * p.Cout) * ld;
    spectrum_load<SEQ, LB, FFT_THREADS>(seq, p.logn, p.radix3, [&](int f) {
        // r = [M'_rr(ci) M'_ri(ci) M'_rr(ci+1) M'_ri(ci+1)], m = [M'_ir M'_ii ...]
        const float *mf = mre + (int64_t)f * fstride;
        const f32x4 r = ac_gload<f32x4>(mf), m = ac_gload<f32x4>(mf + ld);
        return f32x4{r[0] + m[1], m[0] - r[1], r[2] + m[3], m[2] - r[3]};
    });
    fft_inverse<SEQ, U, FFT_THREADS>(fbuf, tw, p.logn, p.radix3, pitch);
    const float inv = 1.0f / (float)N;
    float *dw = p.dst + (int64_t)co * p.k * p.Cin + ci0 + 2 * q;
    for (int t = threadIdx.x / SEQ; t < p.k; t += RS) {
        ac_c2 *d = (ac_c2 *)(dw + (int64_t)t * p.Cin);
        *d += seq[phys(p.k - 1 - t)] * inv;
    }
}

template <auto KERNEL, typename P>
int fft_launch(int blocks, const P &p, int n, int nseq, hipStream_t stream, int FFT_THREADS = 512, bool lds_exact = false) {
    // EVERY transform workgroup asks for the whole LDS of a CU (160 KB), so that it shares its CU with nothing.
    // History (profiles/r03_fft_coresidency_root_cause.txt): beside workgroups of certain OTHER kernels
    // (attention, the photometry forward, the image backward) a transform workgroup computed wrong imaginary parts in up to
    // 100 % of the launches.  In-kernel self-checks (tools/fft_debug.patch) found inputs, twiddles, LDS traffic and barriers
    // right and the arithmetic wrong: the HIGH halves of packed-fp32 instructions (v_pk_add/mul/fma_f32), which only
    // this file's float2 math produced.  The library is built without them (Makefile NOPK; tests/test_no_packed_fp32.py
    // disassembles the code objects) and is exact with shared CUs too (ac_fft_rows_desc.lds_exact = 1: the exact
    // request, diagnostic); the whole-CU request stays as a second line of defence, at no measurable cost.
    const size_t lds = lds_exact ? (size_t)nseq * seq_pitch_n(n, nseq) * sizeof(ac_c2) : (size_t)160 * 1024;
    static bool configured = false;   // once per kernel instance (benign race: same value from any thread)
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(FFT_THREADS), lds, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// transform sizes: N = 2^logn (5 <= logn <= 11), N = 3 * 2^logn (radix3 = 1, 3 <= logn <= 9) or N = 9 * 2^logn
// (radix3 = 2, 3 <= logn <= 7): 24 ... 2048 points
bool size_ok(int logn, int radix3) {
    if (radix3 == 0) return logn >= 5 && logn <= 11;
    return (radix3 == 1 || radix3 == 2) && logn >= 3 && logn <= (radix3 == 1 ? 9 : 7);
}
int size_n(int logn, int radix3) { return pow3(radix3) << logn; }
// sequences per workgroup: 32 (64 channels) for the short transforms, 8 (16 channels) otherwise.  Measured and
// dropped for N = 2048: 4 sequences per 256-thread workgroup (74 KB of LDS, two workgroups per CU so that one loads /
// stores while the other transforms) — the 32-byte row segments cost more than the overlap returns: stage 2's k = 251
// convolution 2.27 ms against 1.96 ms (tools/bench_fftconv.py).
// With the CU to itself a workgroup should be as large as it can: 32 sequences x 1024 threads for N <= 512 (148 KB of
// images at N = 512) when the channel count allows, else 8 sequences x 512 threads.
// The shortest transforms (N <= 96: stages 4 and 5) take 128 sequences = 256 channels per workgroup.
int nseq_for(int n, int channels) {
    if (n <= 96 && channels % 256 == 0) return 128;
    return (n <= 512 && channels % 64 == 0) ? 32 : 8;
}

// The long transforms (N >= 1024, 8 sequences per workgroup) run on 16 waves, one work item per thread and pass: with
// the CU to itself a workgroup hides the LDS latency of a pass with waves, not with items in flight per thread
// (stage 2's k = 251 convolution 1.81 -> 1.64 ms, whole step -0.25 ms against the 8-wave form with four items in
// flight per thread, which the taps transforms keep).

// workgroups of a rows launch: every one takes ~4 tiles (latency of the next tile's loads hidden behind the passes of
// the current one), never fewer workgroups than CUs, a multiple of 8 so that a workgroup's tiles stay on its XCD
int rows_grid(int total) {
    constexpr int tpw = 4;
    if (total <= 256) return total;
    int w = (total + tpw - 1) / tpw;
    w = w < 256 ? 256 : w;
    w = (w + 7) & ~7;
    return w < total ? w : total;
}

int rows_check(const ac_fft_rows_desc &d, bool inverse) {
    if (!d.rows || !d.tw || !d.spec || d.B <= 0 || d.L <= 0 || d.C <= 0 || (d.C % 16) || !size_ok(d.logn, d.radix3))
        return AC_EINVAL;
    const int N = size_n(d.logn, d.radix3);
    if (d.blocks < 1 || d.shift < 0 || d.shift >= N) return AC_EINVAL;
    if (d.blocks > 1 && (d.block_step < 1 || d.block_step > N || (int64_t)d.blocks * d.block_step < d.L)) return AC_EINVAL;
    if (inverse) {
        const int cnt = d.blocks > 1 ? d.block_step : d.L;
        if (cnt + d.shift > N || d.rows_lo) return AC_EINVAL;
    } else {
        if (d.blocks == 1 && d.L + d.shift > N) return AC_EINVAL;       // one sequence per sample must hold every row
        if (d.n_lo < 0 || d.n_hi > N || d.n_lo >= d.n_hi) return AC_EINVAL;
    }
    const uintptr_t amask = d.rows_lo ? 3u : 7u;   // one 4-byte (two bf16) or 8-byte (two fp32) access per channel pair
    if ((d.row_stride % 2) || (d.batch_stride % 2) || (d.col_off % 2) || ((uintptr_t)d.rows & amask) ||
        ((uintptr_t)d.rows_lo & amask) || !ac_aligned16(d.spec) || ((uintptr_t)d.tw & 7u) || ((uintptr_t)d.bias & 7u))
        return AC_EALIGN;
    return AC_OK;
}

}  // namespace

extern "C" int ac_fft_rows_fwd(const ac_fft_rows_desc *dp, ac_stream_t stream) {
    if (!dp) return AC_EINVAL;
    RowsParams p;
    p.d = *dp;
    ac_fft_rows_desc &d = p.d;
    if (d.n_hi == 0 && d.n_lo == 0 && size_ok(d.logn, d.radix3)) d.n_hi = size_n(d.logn, d.radix3);
    const int rc = rows_check(d, false);
    if (rc != AC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool ex = d.lds_exact != 0;
    const int n = size_n(d.logn, d.radix3), ns = nseq_for(n, d.C), blocks = rows_grid(d.B * d.blocks * (d.C / (2 * ns)));
    if (ns == 128)
        return d.rows_lo ? fft_launch<fft_rows_fwd_kernel<128, 1, true, 1024>>(blocks, p, n, 128, st, 1024, ex)
                         : fft_launch<fft_rows_fwd_kernel<128, 1, false, 1024>>(blocks, p, n, 128, st, 1024, ex);
    if (ns == 32)
        return d.rows_lo ? fft_launch<fft_rows_fwd_kernel<32, 1, true, 1024>>(blocks, p, n, 32, st, 1024, ex)
                         : fft_launch<fft_rows_fwd_kernel<32, 1, false, 1024>>(blocks, p, n, 32, st, 1024, ex);
    if (n >= 1024)
        return d.rows_lo ? fft_launch<fft_rows_fwd_kernel<8, 1, true, 1024>>(blocks, p, n, 8, st, 1024, ex)
                         : fft_launch<fft_rows_fwd_kernel<8, 1, false, 1024>>(blocks, p, n, 8, st, 1024, ex);
    return d.rows_lo ? fft_launch<fft_rows_fwd_kernel<8, 1, true, 512>>(blocks, p, n, 8, st, 512, ex)
                     : fft_launch<fft_rows_fwd_kernel<8, 1, false, 512>>(blocks, p, n, 8, st, 512, ex);
}

extern "C" int ac_fft_rows_inv(const ac_fft_rows_desc *dp, ac_stream_t stream) {
    if (!dp) return AC_EINVAL;
    RowsParams p;
    p.d = *dp;
    const ac_fft_rows_desc &d = p.d;
    const int rc = rows_check(d, true);
    if (rc != AC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool ex = d.lds_exact != 0;
    const int n = size_n(d.logn, d.radix3), ns = nseq_for(n, d.C), blocks = rows_grid(d.B * d.blocks * (d.C / (2 * ns)));
    if (ns == 128) return fft_launch<fft_rows_inv_kernel<128, 1, 1024>>(blocks, p, n, 128, st, 1024, ex);
    if (ns == 32) return fft_launch<fft_rows_inv_kernel<32, 1, 1024>>(blocks, p, n, 32, st, 1024, ex);
    if (n >= 1024) return fft_launch<fft_rows_inv_kernel<8, 1, 1024>>(blocks, p, n, 8, st, 1024, ex);
    return fft_launch<fft_rows_inv_kernel<8, 1, 512>>(blocks, p, n, 8, st, 512, ex);
}

extern "C" int ac_fft_taps_fwd(const float *w, int32_t Cout, int32_t Cin, int32_t k, int32_t logn, int32_t radix3,
                               const float *tw, float *hblock, ac_stream_t stream) {
    if (!w || !tw || !hblock || Cout <= 0 || Cin <= 0 || (Cin % 16) || k <= 0 || !size_ok(logn, radix3) ||
        k > size_n(logn, radix3))
        return AC_EINVAL;
    if (((uintptr_t)w & 7u) || !ac_aligned16(hblock) || ((uintptr_t)tw & 7u)) return AC_EALIGN;
    TapsParams p;
    p.src = w; p.dst = hblock; p.tw = (const ac_c2 *)tw; p.Cout = Cout; p.Cin = Cin; p.k = k; p.logn = logn; p.radix3 = radix3;
    hipStream_t st = (hipStream_t)stream;
    const int n = size_n(logn, radix3), ns = nseq_for(n, Cin), blocks = Cout * (Cin / (2 * ns));
    if (ns == 128) return fft_launch<fft_taps_fwd_kernel<128, 1, 1024>>(blocks, p, n, 128, st, 1024);
    if (ns == 32) return fft_launch<fft_taps_fwd_kernel<32, 1, 1024>>(blocks, p, n, 32, st, 1024);
    if (n >= 1024) return fft_launch<fft_taps_fwd_kernel<8, 4, 512>>(blocks, p, n, 8, st);
    return fft_launch<fft_taps_fwd_kernel<8, 1, 512>>(blocks, p, n, 8, st);
}

extern "C" int ac_fft_taps_inv(const float *m, int32_t Cout, int32_t Cin, int32_t k, int32_t logn, int32_t radix3,
                               const float *tw, float *dw, ac_stream_t stream) {
    if (!m || !tw || !dw || Cout <= 0 || Cin <= 0 || (Cin % 16) || k <= 0 || !size_ok(logn, radix3) ||
        k > size_n(logn, radix3))
        return AC_EINVAL;
    if (!ac_aligned16(m) || ((uintptr_t)tw & 7u) || ((uintptr_t)dw & 7u)) return AC_EALIGN;
    TapsParams p;
    p.src = m; p.dst = dw; p.tw = (const ac_c2 *)tw; p.Cout = Cout; p.Cin = Cin; p.k = k; p.logn = logn; p.radix3 = radix3;
    hipStream_t st = (hipStream_t)stream;
    const int n = size_n(logn, radix3), ns = nseq_for(n, Cin), blocks = Cout * (Cin / (2 * ns));
    if (ns == 128) return fft_launch<fft_taps_inv_kernel<128, 1, 1024>>(blocks, p, n, 128, st, 1024);
    if (ns == 32) return fft_launch<fft_taps_inv_kernel<32, 1, 1024>>(blocks, p, n, 32, st, 1024);
    if (n >= 1024) return fft_launch<fft_taps_inv_kernel<8, 4, 512>>(blocks, p, n, 8, st);
    return fft_launch<fft_taps_inv_kernel<8, 1, 512>>(blocks, p, n, 8, st);
}
