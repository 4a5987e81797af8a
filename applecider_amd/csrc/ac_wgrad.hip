// Weight gradient of a 'same' Conv1d on the bf16 matrix cores with the input window resident in LDS
// (gfx950).  dW[co, t, ci] = sum_{b, l} dy[b, l, co] * xpad[b, l + t + off, ci]
// (SpectraNetBlock conv bank, src/applecider/models/spectranet.py:18-20,25; torch's conv1d backward).
//
// As a plain TN product (M = Cout, N = k*Cin, K = B*L) the B operand of tap t is the padded input
// shifted by t rows, so a 128 x 256 output tile (4 taps x 64 channels) streams four shifted copies of
// the same rows: 3.8x more fabric traffic than the data holds (profiles/r01_pmc_hbm_traffic.json).
// Here a 512-thread workgroup owns 128 output channels x 8 TAPS x 64 input channels and, per K step
// of 64 positions, loads ONE dy tile [64 x 128] and ONE input window [(64 + 7) x 64]; the B fragments
// of tap t are transposed reads (ds_read_b64_tr_b16) of the window at a row offset of t.  Operand bytes
// per FLOP drop 4x against the 128 x 256 tile, which moves the product from the L2-feed bound
// (~700 TF) towards the MFMA bound.
//   waves   8 = 2 (64 output channels each) x 4 (2 taps each); 8 accumulator tiles of 32x32 per wave
//   LDS     dy image [64][128 + 32], window image [72][64 + 32] bf16, two stages
//   K split over workgroups (blockIdx.y), fp32 atomics into dW (two 128-byte segments per instruction)
// SPLIT: operands are (hi, lo) bf16 planes (math mode bf16x3): 3 MFMAs per product, images doubled.
#include "ac_common.h"
#include <hip/hip_bf16.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_TAPS = 8, WG_KP = 64, WG_CO = 128, WG_CI = 64;
// window rows: 64 positions + 8 taps of one sample; short sequences (L = 16 / 32: SpectraNet stages 4-5 have
// L = 64 / 16) put 64 / L whole samples in a K step, each with its own L + 8 rows -> at most 4 x 24 = 96
constexpr int B_ROWS = 96;
// Toeplitz form (ac_wgrad_desc.tap_row_step = 8): tap t reads window rows (position + 8 t): 64 + 8 * 8 rows
constexpr int B_ROWS_TOEP = 128, TOEP_STEP = 8;
// Row pitches (elements) by MFMA shape.  32x32x16 form: a 32-lane half reads 4 rows x 64 bytes, pitch = 16
// banks (mod 64).  16x16x32 form: a half reads 8 consecutive rows x 32 bytes, pitch = an odd multiple of
// 8 banks.  Both conflict-free for ds_read_b64_tr_b16.
template <bool S16, bool TOEP = false> struct Img {
    static constexpr int A_PITCH = S16 ? WG_CO + 16 : WG_CO + 32;
    static constexpr int B_PITCH = S16 ? WG_CI + 16 : WG_CI + 32;
    static constexpr int A_IMG = WG_KP * A_PITCH, B_IMG = (TOEP ? B_ROWS_TOEP : B_ROWS) * B_PITCH;
};

struct WgradParams {
    ac_wgrad_desc d;
    int co_tiles, ci_tiles, tap_chunks, steps_total, steps_per_split;
};

// 8 consecutive k (positions) of column `col` of an [k][cols] image: two transposed 4x16 block reads
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_t(const unsigned short *img, int colbase, int s, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int k0 = 16 * s + 8 * (g >> 1);
    const unsigned short *a0 = img + (k0 + q) * PITCH + colbase + 16 * (g & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(a0 + 4 * PITCH));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// 16x16x32 form: lane l holds column (l & 15) of a 16-column block and 8 of the step's 32 k (positions);
// the k of lane group g = l >> 4 are rows 16(g>>1) + 4(g&1) + {0..3, 8..11} of the step — any assignment
// works as long as both operands use the same one, and this one lets a 32-lane half read 8 consecutive
// rows (conflict-free at the pitches above).
// seg_rows > 0: the image holds 64 / Ls segments of seg_rows rows (one per sample of a short-sequence
// step); the lane's 8 positions lie inside one aligned 16-block, hence inside one segment.
// krow >= 0: the image row of the lane group's first position, precomputed by the caller (short sequences).
template <int PITCH, bool SEG = false>
__device__ __forceinline__ bf16x8 frag16_t(const unsigned short *img, int colbase, int s2, int lane, int krow = 0) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    int k0 = 32 * s2 + 16 * (g >> 1) + 4 * (g & 1);
    if constexpr (SEG) k0 = krow;
    const unsigned short *a0 = img + (k0 + q) * PITCH + colbase + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(a0 + 8 * PITCH));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// SHORT: L = 16 / 32 (64 / L whole samples per K step); a separate instantiation so that the long-sequence
// kernel keeps its compile-time addressing (as one runtime-switched kernel it lost 18 %).
// TOEP: the Toeplitz form (header: ac_wgrad_desc.tap_row_step) — window rows 8 per tap, blocked dy columns
template <bool SPLIT, bool S16, bool SHORT, bool TOEP = false>
__global__ __launch_bounds__(512, 1) void conv1d_wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smw[];
    const ac_wgrad_desc &d = p.d;
    constexpr int A_PITCH = Img<S16, TOEP>::A_PITCH, B_PITCH = Img<S16, TOEP>::B_PITCH;
    constexpr int A_IMG = Img<S16, TOEP>::A_IMG, B_IMG = Img<S16, TOEP>::B_IMG;
    constexpr int TSTEP = TOEP ? TOEP_STEP : 1;
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int STAGE = NPL * (A_IMG + B_IMG);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // One flat grid, each XCD takes a contiguous run of (K split, output tile) pairs with the tap chunk
    // fastest: the workgroups of one K split (they stream the same dy tiles and overlapping input
    // windows) sit behind ONE L2.  As a 2-D grid (tiles x splits) the hardware dealt every split's
    // tiles round-robin over the 8 XCDs and each L2 fetched the whole of dy: 3.9x the algorithmic bytes
    // on the fabric counters.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgl = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tiles = p.tap_chunks * p.ci_tiles * p.co_tiles;
    const int ksplit = wgl / tiles, wg = wgl - ksplit * tiles;
    const int tc = wg % p.tap_chunks;
    const int cit = (wg / p.tap_chunks) % p.ci_tiles;
    const int cot = wg / (p.tap_chunks * p.ci_tiles);
    const int t0 = tc * WG_TAPS;

    const int s_begin = ksplit * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split;
    if (s_end > p.steps_total) s_end = p.steps_total;
    if (s_begin >= s_end) return;

    const unsigned short *dy = (const unsigned short *)d.dy, *x = (const unsigned short *)d.x;
    // loader roles: dy tile 64 rows x 16 chunks = 1024 chunks (2 per thread); window 72 rows x 8 chunks
    const int ar = t >> 4, ac = t & 15;            // + 32 rows for the second chunk
    const int br = t >> 3, bc = t & 7;             // rows 0..63 ; second chunk: rows 64..71 (t < 64)
    constexpr bool shortL = SHORT;                       // 64 / L whole samples per K step (S16 form only)
    const int steps_per_seq = shortL ? 1 : d.L / WG_KP;
    const int lsh = shortL ? 31 - __builtin_clz(d.L) : 6;  // log2 of the positions per sample in a step
    const int seg = shortL ? d.L + WG_TAPS : 0;           // window rows per sample (short sequences)
    const int nbrow = shortL ? (WG_KP >> lsh) * seg : WG_KP + TSTEP * WG_TAPS;

    auto gload_step = [&](int step, u32x4 (&ra)[2 * NPL], u32x4 (&rb)[2 * NPL]) {
        int b, l0;
        int64_t ao0, ao1;
        if constexpr (shortL) {
            b = step << (6 - lsh);
            l0 = 0;
            // dy tile row r (position r of the step) -> (sample b + (r >> lsh), position r & (L - 1))
            const int r0 = ar, r1 = ar + 32;
            ao0 = (int64_t)(r0 >> lsh) * d.dy_batch_stride + (int64_t)(d.dy_row_base + (r0 & ((1 << lsh) - 1))) * d.dy_row_stride;
            ao1 = (int64_t)(r1 >> lsh) * d.dy_batch_stride + (int64_t)(d.dy_row_base + (r1 & ((1 << lsh) - 1))) * d.dy_row_stride;
        } else {
            b = step / steps_per_seq;
            l0 = (step - b * steps_per_seq) * WG_KP;
            ao0 = (int64_t)(d.dy_row_base + l0 + ar) * d.dy_row_stride;
            ao1 = ao0 + 32 * d.dy_row_stride;
        }
        int acol = cot * WG_CO + ac * 8;   // column of the dy row; blocked layout: (c / block) * stride + c % block
        if constexpr (TOEP) acol = (acol / d.dy_block) * (int)d.dy_block_stride + acol % d.dy_block;
        const unsigned short *ap = dy + (int64_t)b * d.dy_batch_stride + d.dy_col_off + acol;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            ra[2 * pl] = ac_gload<u32x4>(ap + pl * d.dy_lo_off + ao0);
            ra[2 * pl + 1] = ac_gload<u32x4>(ap + pl * d.dy_lo_off + ao1);
        }
        // window rows: image row w -> (sample w / seg, input row w % seg) for short sequences
        int w0 = br, w1 = 64 + (t >> 3);          // second pass: rows 64.. (up to 95: all 256 threads)
        w1 = w1 < nbrow ? w1 : nbrow - 1;
        int s0 = 0, s1 = 0;
        if constexpr (shortL) {
            s0 = w0 / seg; w0 -= s0 * seg;
            s1 = w1 / seg; w1 -= s1 * seg;
        }
        int row0 = d.x_row_base + l0 + TSTEP * t0 + w0;
        int row1 = d.x_row_base + l0 + TSTEP * t0 + w1;
        row0 = row0 < d.x_rows ? row0 : d.x_rows - 1;
        row1 = row1 < d.x_rows ? row1 : d.x_rows - 1;
        const unsigned short *bp = x + (int64_t)b * d.x_batch_stride + cit * WG_CI + bc * 8;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            rb[2 * pl] = ac_gload<u32x4>(bp + pl * d.x_lo_off + (int64_t)s0 * d.x_batch_stride + (int64_t)row0 * d.x_row_stride);
            rb[2 * pl + 1] = ac_gload<u32x4>(bp + pl * d.x_lo_off + (int64_t)s1 * d.x_batch_stride + (int64_t)row1 * d.x_row_stride);
        }
    };
    auto lds_store = [&](unsigned short *stage, const u32x4 (&ra)[2 * NPL], const u32x4 (&rb)[2 * NPL]) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            unsigned short *ai = stage + pl * A_IMG, *bi = stage + NPL * A_IMG + pl * B_IMG;
            *(u32x4 *)(ai + ar * A_PITCH + ac * 8) = ra[2 * pl];
            *(u32x4 *)(ai + (ar + 32) * A_PITCH + ac * 8) = ra[2 * pl + 1];
            *(u32x4 *)(bi + br * B_PITCH + bc * 8) = rb[2 * pl];
            if (64 + (t >> 3) < nbrow) *(u32x4 *)(bi + (64 + (t >> 3)) * B_PITCH + bc * 8) = rb[2 * pl + 1];
        }
    };

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 acc[2][4];     // 32x32x16 form: [32 output channels][tap pair x 32 input channels]
    f32x4 acs[4][8];      // 16x16x32 form: [16 output channels][tap (2) x 16 input channels (4)]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acs[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // short sequences: window row of this lane group's first position in step half s2 (the group's 8
    // positions lie inside one aligned 16-block, i.e. inside one sample's segment of the window image)
    int krow[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int k0 = 32 * s2 + 16 * ((lane >> 4) >> 1) + 4 * ((lane >> 4) & 1);
        krow[s2] = (k0 >> lsh) * seg + (k0 & ((1 << lsh) - 1));
    }
    auto compute16 = [&](const unsigned short *stage) {
        const unsigned short *ah = stage, *al = stage + A_IMG;
        const unsigned short *bh = stage + NPL * A_IMG, *bl = bh + B_IMG;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 a_h[4], a_l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a_h[i] = frag16_t<A_PITCH>(ah, wm * 64 + 16 * i, s2, lane);
                if (SPLIT) a_l[i] = frag16_t<A_PITCH>(al, wm * 64 + 16 * i, s2, lane);
            }
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int tap = 2 * wn + tp;
                // Toeplitz form: K is padded to 64-element taps and the last chunk of 8 is mostly past the end
                // (k = 1021: 17 taps) — a wave skips the taps that do not exist (wave-uniform)
                if (TOEP && t0 + tap >= d.k) continue;
                bf16x8 b_h[4], b_l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b_h[j] = frag16_t<B_PITCH, SHORT>(bh + TSTEP * tap * B_PITCH, 16 * j, s2, lane, krow[s2]);
                    if (SPLIT) b_l[j] = frag16_t<B_PITCH, SHORT>(bl + TSTEP * tap * B_PITCH, 16 * j, s2, lane, krow[s2]);
                }
                if (SPLIT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acs[i][4 * tp + j] = AC_MFMA16S(a_l[i], b_h[j], acs[i][4 * tp + j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acs[i][4 * tp + j] = AC_MFMA16S(a_h[i], b_l[j], acs[i][4 * tp + j]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acs[i][4 * tp + j] = AC_MFMA16S(a_h[i], b_h[j], acs[i][4 * tp + j]);
            }
        }
    };
    auto compute32 = [&](const unsigned short *stage) {
        const unsigned short *ah = stage, *al = stage + A_IMG;
        const unsigned short *bh = stage + NPL * A_IMG, *bl = bh + B_IMG;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a_h[2], a_l[2], b_h[4], b_l[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a_h[i] = frag_t<A_PITCH>(ah, wm * 64 + 32 * i, s, lane);
                if (SPLIT) a_l[i] = frag_t<A_PITCH>(al, wm * 64 + 32 * i, s, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tap = 2 * wn + (j >> 1), ch = 32 * (j & 1);
                b_h[j] = frag_t<B_PITCH>(bh + tap * B_PITCH, ch, s, lane);
                if (SPLIT) b_l[j] = frag_t<B_PITCH>(bl + tap * B_PITCH, ch, s, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (SPLIT) {
                        acc[i][j] = AC_MFMA16(a_l[i], b_h[j], acc[i][j]);
                        acc[i][j] = AC_MFMA16(a_h[i], b_l[j], acc[i][j]);
                    }
                    acc[i][j] = AC_MFMA16(a_h[i], b_h[j], acc[i][j]);
                }
        }
    };

    auto compute = [&](const unsigned short *stage) {
        if constexpr (S16) compute16(stage);
        else compute32(stage);
    };
    unsigned short *S0 = smw, *S1 = smw + STAGE;
    u32x4 ra[2 * NPL], rb[2 * NPL];
    gload_step(s_begin, ra, rb);
    lds_store(S0, ra, rb);
    __syncthreads();
    int cur = 0;
    for (int step = s_begin; step < s_end; ++step) {
        const bool more = step + 1 < s_end;
        if (more) gload_step(step + 1, ra, rb);
        compute(cur ? S1 : S0);
        if (more) lds_store(cur ? S0 : S1, ra, rb);
        __syncthreads();
        cur ^= 1;
    }

    // dW[co, t, ci] += acc (fp32 atomics; lanes 0..31 of a register cover 32 consecutive ci)
    float *dw = d.dw;
    if constexpr (S16) {
        // 16x16x32 accumulators: register e of lane l = (output channel 4(l>>4) + e, input channel l & 15)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = t0 + 2 * wn + (j >> 2);
            if (tap >= d.k) continue;
            const int ci = cit * WG_CI + 16 * (j & 3) + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = cot * WG_CO + wm * 64 + 16 * i + 4 * (lane >> 4) + e;
                    atomicAdd(dw + (int64_t)co * d.ldw + (int64_t)tap * d.Cin + ci, acs[i][j][e]);
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tap = t0 + 2 * wn + (j >> 1);
        if (tap >= d.k) continue;
        const int ci = cit * WG_CI + 32 * (j & 1) + li;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = cot * WG_CO + wm * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                atomicAdd(dw + (int64_t)co * d.ldw + (int64_t)tap * d.Cin + ci, acc[i][j][e]);
            }
    }
}

// Round 3, measured and NOT kept: the same product on a ring of four LDS-DMA half-steps with the fragments of the
// next half read under the MFMAs of this one (the recipe that gave conv1d_window_x3r_kernel +7 %): 8.18 ms against
// 7.54 ms for this kernel over the nine stage-2..4 products (profiles/r03_wgrad_ring_ab.txt).  Both operands are
// re-streamed every step here (26 DMA instructions per half against 16 in the window kernel, each ~60-100 issue
// cycles of a wave that then feeds the matrix pipe nothing), the barrier interval halves, and the input rows' halo
// is fetched twice; this kernel's 192 MFMAs per wave and barrier already hide most of what the ring removes.
template <bool SPLIT, bool S16, bool SHORT = false, bool TOEP = false>
int launch_wgrad(WgradParams &p, hipStream_t stream) {
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr size_t LDS = (size_t)2 * NPL * (Img<S16, TOEP>::A_IMG + Img<S16, TOEP>::B_IMG) * sizeof(short);
    static_assert(LDS <= 160 * 1024, "two stages in the LDS");
    static const hipError_t attr = hipFuncSetAttribute((const void *)conv1d_wgrad_kernel<SPLIT, S16, SHORT, TOEP>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (attr != hipSuccess) return -(int)attr - 2000;
    const ac_wgrad_desc &d = p.d;
    dim3 grid(p.co_tiles * p.ci_tiles * p.tap_chunks * ((p.steps_total + p.steps_per_split - 1) / p.steps_per_split));
    hipLaunchKernelGGL((conv1d_wgrad_kernel<SPLIT, S16, SHORT, TOEP>), grid, dim3(512), LDS, stream, p);
    AC_CHECK_LAUNCH();
    (void)d;
    return AC_OK;
}

}  // namespace

extern "C" int ac_conv1d_wgrad_bf16(const ac_wgrad_desc *dp, ac_stream_t stream) {
    if (!dp) return AC_EINVAL;
    WgradParams p;
    p.d = *dp;
    const ac_wgrad_desc &d = p.d;
    if (!d.dy || !d.x || !d.dw || d.B <= 0 || d.L <= 0 || d.k <= 0 || d.Cout <= 0 || d.Cin <= 0) return AC_EINVAL;
    // shapes outside the tile grid go back to the caller's generic TN product
    const bool short_seq = d.L < WG_KP && (d.L == 16 || d.L == 32) && ((int64_t)d.B * d.L) % WG_KP == 0 && d.variant != 2;
    if (((d.L % WG_KP) && !short_seq) || (d.Cout % WG_CO) || (d.Cin % WG_CI)) return AC_EINVAL;
    if (!ac_aligned16(d.dy) || !ac_aligned16(d.x) || (d.dy_row_stride % 8) || (d.dy_batch_stride % 8) ||
        (d.dy_col_off % 8) || (d.x_row_stride % 8) || (d.x_batch_stride % 8) || (d.dy_lo_off % 8) ||
        (d.x_lo_off % 8))
        return AC_EALIGN;
    if (d.x_rows <= 0 || d.ldw < (int64_t)d.k * d.Cin) return AC_EINVAL;
    p.co_tiles = d.Cout / WG_CO;
    p.ci_tiles = d.Cin / WG_CI;
    p.tap_chunks = (d.k + WG_TAPS - 1) / WG_TAPS;
    p.steps_total = (int)(((int64_t)d.B * d.L) / WG_KP);
    int split = d.split_k > 0 ? d.split_k : 1;
    if (split > p.steps_total) split = p.steps_total;
    p.steps_per_split = (p.steps_total + split - 1) / split;
    const bool sp = d.dy_lo_off != 0 || d.x_lo_off != 0;
    if (sp && (d.dy_lo_off == 0 || d.x_lo_off == 0)) return AC_EINVAL;
    if (d.tap_row_step > 1 || d.dy_block > 0) {   // Toeplitz form: split-bf16, 16x16x32 tiles, long sequences
        if (d.tap_row_step != TOEP_STEP || d.dy_block <= 0 || (d.dy_block % 8) || (d.dy_block_stride % 8) || !sp || short_seq ||
            d.Cin != WG_CI || d.variant == 2)
            return AC_EINVAL;
        return launch_wgrad<true, true, false, true>(p, (hipStream_t)stream);
    }
    // variant 2: the 32x32x16 form (A/B measurements); default: v_mfma_f32_16x16x32 (higher sustained clock)
    if (d.variant == 2)
        return sp ? launch_wgrad<true, false>(p, (hipStream_t)stream) : launch_wgrad<false, false>(p, (hipStream_t)stream);
    if (short_seq)
        return sp ? launch_wgrad<true, true, true>(p, (hipStream_t)stream)
                  : launch_wgrad<false, true, true>(p, (hipStream_t)stream);
    return sp ? launch_wgrad<true, true>(p, (hipStream_t)stream) : launch_wgrad<false, true>(p, (hipStream_t)stream);
}
