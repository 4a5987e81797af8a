// LDS-resident-window Conv1d on the bf16 matrix cores (gfx950).
//
// The generic gather-GEMM treats a 'same' Conv1d as an implicit GEMM whose A rows are overlapping
// windows of the padded sequence, so each input element is re-read k times through L1/L2 (and from
// MALL/HBM once the per-workgroup window outgrows L1: the stage-2 input-gradient product streamed
// 33 GB that way).  Here a workgroup loads its whole input window
//        rows [l0, l0 + BM + k - 1)  x  C channels   (bf16, up to 130 KB of the 160 KB LDS)
// ONCE, and the K loop (taps x channel tiles) reads the A fragments straight from that window at a
// row offset of one row per tap; only the weights stream (global -> registers -> double-buffered
// LDS, one barrier per 64-deep K tile).  Used for the forward product (A = padded input) and the
// input-gradient product (A = padded output gradient, taps flipped) of SpectraNetBlock
// (src/applecider/models/spectranet.py:18-20,25) when L is a multiple of the row tile.
#include "ac_common.h"
#include <hip/hip_bf16.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned short win_bf16(float x) {
    return ac_f2h(x);
}

struct ConvWinParams {
    ac_convwin_desc d;
    int tiles_l, tiles_n, cchunks;  // cchunks = C / 8 (16-byte chunks per window row)
    int vec_epi;
};

// window image: row r, 16-byte chunk cc -> element offset.  128-byte rows (C = 64) alternate the
// two halves of the 256-byte bank row, wider rows start on a bank-row boundary.
template <int C>
__device__ __forceinline__ int win_off(int r, int cc) {
    if (C == 64) return r * 64 + ((cc ^ ((r >> 1) & 7)) << 3);
    return r * C + (((cc & ~15) | ((cc & 15) ^ (r & 15))) << 3);
}

template <int WM, int WN, int C>
__global__ __launch_bounds__(WM *WN * 64, 1) void conv1d_window_kernel(ConvWinParams p) {
    constexpr int NT = WM * WN * 64, BM = WM * 64, BN = WN * 64;
    constexpr int BCH = BN * 8 / NT;  // weight chunks per thread per K tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *win = reinterpret_cast<unsigned short *>(smem);
    const ac_convwin_desc &d = p.d;
    const int W = BM + d.k - 1;
    unsigned short *bst = win + W * C;  // two weight stages of BN*64 elements

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // tile order: n fastest, then l tiles, then batch (blocks of one XCD share a weight panel)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n;
    const int tl = (wg / p.tiles_n) % p.tiles_l;
    const int b = wg / (p.tiles_n * p.tiles_l);
    const int l0 = tl * BM;

    // ---- window: global -> LDS, once
    {
        const unsigned short *a = (const unsigned short *)d.a + (int64_t)b * d.a_batch_stride +
                                  (int64_t)(d.row_base + l0) * d.a_row_stride + d.a_col_off;
        const int total = W * (C / 8);
        for (int idx = t; idx < total; idx += NT) {
            const int r = idx / (C / 8), cc = idx % (C / 8);
            const u32x4 v = ac_gload<u32x4>(a + (int64_t)r * d.a_row_stride + cc * 8);
            *(u32x4 *)(win + win_off<C>(r, cc)) = v;
        }
    }

    // ---- weight loader: BN rows x 64 k per K tile, thread -> rows (t>>3) + (NT/8)*i, chunk t&7
    const unsigned short *wptr = (const unsigned short *)d.w;
    int64_t wbase[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        int n = tn * BN + (t >> 3) + (NT / 8) * i;
        n = n < d.N ? n : d.N - 1;
        wbase[i] = (int64_t)n * d.w_row_stride + 8 * (t & 7);
    }
    const int ctiles = C / 64;
    const int nkt = d.k * ctiles;
    auto koff = [&](int kt) -> int64_t {
        const int tap = kt / ctiles, c0 = (kt % ctiles) * 64;
        return (int64_t)(d.flip ? d.k - 1 - tap : tap) * d.w_tap_stride + c0;
    };
    auto wload = [&](int kt, u32x4 (&v)[BCH]) {
        const int64_t ko = koff(kt);
#pragma unroll
        for (int i = 0; i < BCH; ++i) v[i] = ac_gload<u32x4>(wptr + wbase[i] + ko);
    };
    auto wstore = [&](unsigned short *tile, const u32x4 (&v)[BCH]) {
        const int c = t & 7;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = (t >> 3) + (NT / 8) * i;
            *(u32x4 *)(tile + r * 64 + ((c ^ ((r >> 1) & 7)) << 3)) = v[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Weight stream: prefetch distance 2 through two register sets, branch-free steady state (tail
    // iterations re-load the last tile into an unused stage), so hipcc keeps counted vmcnt waits.
    auto compute = [&](int kt, const unsigned short *bt) {
        const int tap = kt / ctiles, cc0 = (kt % ctiles) * 8;
        const int r0 = wm * 64 + li + tap, r1 = r0 + 32;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cc = cc0 + 2 * s + lh;
            const bf16x8 a0 = *(const bf16x8 *)(win + win_off<C>(r0, cc));
            const bf16x8 a1 = *(const bf16x8 *)(win + win_off<C>(r1, cc));
            const int n0 = wn * 64 + li, n1 = n0 + 32;
            const int chunk = 2 * s + lh;
            const bf16x8 b0 = *(const bf16x8 *)(bt + n0 * 64 + ((chunk ^ ((n0 >> 1) & 7)) << 3));
            const bf16x8 b1 = *(const bf16x8 *)(bt + n1 * 64 + ((chunk ^ ((n1 >> 1) & 7)) << 3));
            acc[0][0] = AC_MFMA16(a0, b0, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0, b1, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1, b0, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1, b1, acc[1][1]);
        }
    };
    unsigned short *S0 = bst, *S1 = bst + BN * 64;
    const int last = nkt - 1;
    u32x4 rb0[BCH], rb1[BCH];
    wload(0, rb0);
    wstore(S0, rb0);
    __syncthreads();  // window + first weight tile visible
    wload(1 < last ? 1 : last, rb0);
    // two K tiles per trip, no exit between them (a `break` after the first half gives the loop header a back edge on
    // which the first half's loads are in flight, and hipcc then waits for every load at the top of each trip)
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        wload(kt + 2 < last ? kt + 2 : last, rb1);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt, S0);
        wstore(S1, rb0);
        __syncthreads();
        wload(kt + 3 < last ? kt + 3 : last, rb0);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt + 1, S1);
        wstore(S0, rb1);
        __syncthreads();
    }
    if (kt < nkt) {
        compute(kt, S0);
        __syncthreads();
    }

    // ---- epilogue: out[b, l0 + m, n] (+)= acc (+ bias).  The wave parks each 32x64 half of its
    // tile in 8 KB of the (now idle) window and re-reads it row-major: float4 stores.
    float *cb = d.c + ((int64_t)b * d.L + l0) * d.ldc;
    if (p.vec_epi) {
        float *wbuf = smem + wave * 2048;
        const int rsub = lane >> 4, c4 = 4 * (lane & 15);
        const int n = tn * BN + wn * 64 + c4;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (d.bias && n < d.N) bias4 = *(const f32x4 *)(d.bias + n);
#pragma unroll
        for (int sa = 0; sa < 2; ++sa) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                wbuf[r * 64 + li] = acc[sa][0][e];
                wbuf[r * 64 + 32 + li] = acc[sa][1][e];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + rsub;
                f32x4 v = *(const f32x4 *)(wbuf + r * 64 + c4) + bias4;
                if (n < d.N) {
                    const int64_t row = wm * 64 + sa * 32 + r;
                    if (d.c16) {
                        ushort4 h;
                        h.x = win_bf16(v[0]); h.y = win_bf16(v[1]); h.z = win_bf16(v[2]); h.w = win_bf16(v[3]);
                        *(ushort4 *)((unsigned short *)d.c16 + ((int64_t)b * d.L + l0 + row) * d.ldc16 + n) = h;
                    }
                    if (d.c) {
                        f32x4 *dst = (f32x4 *)(cb + row * d.ldc + n);
                        if (d.accumulate) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
        return;
    }
    const int nn0 = tn * BN + wn * 64 + li, nn1 = nn0 + 32;
    const float bias0 = (d.bias && nn0 < d.N) ? d.bias[nn0] : 0.f;
    const float bias1 = (d.bias && nn1 < d.N) ? d.bias[nn1] : 0.f;
#pragma unroll
    for (int sa = 0; sa < 2; ++sa) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = wm * 64 + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v0 = acc[sa][0][e] + bias0, v1 = acc[sa][1][e] + bias1;
            if (d.c16) {
                unsigned short *r16 = (unsigned short *)d.c16 + ((int64_t)b * d.L + l0 + m) * d.ldc16;
                if (nn0 < d.N) r16[nn0] = win_bf16(v0);
                if (nn1 < d.N) r16[nn1] = win_bf16(v1);
            }
            if (d.c) {
                float *row = cb + (int64_t)m * d.ldc;
                if (nn0 < d.N) row[nn0] = d.accumulate ? row[nn0] + v0 : v0;
                if (nn1 < d.N) row[nn1] = d.accumulate ? row[nn1] + v1 : v1;
            }
        }
    }
}

template <int WM, int WN, int C>
int launch(ConvWinParams &p, hipStream_t stream) {
    constexpr int BM = WM * 64, BN = WN * 64, NT = WM * WN * 64;
    const ac_convwin_desc &d = p.d;
    const size_t lds = ((size_t)(BM + d.k - 1) * C + 2 * BN * 64) * sizeof(short);
    if (lds > 160 * 1024) return AC_EINVAL;
    p.tiles_l = d.L / BM;
    p.tiles_n = (d.N + BN - 1) / BN;
    p.cchunks = C / 8;
    p.vec_epi = (d.N % 4 == 0) && (!d.c || ((d.ldc % 4 == 0) && ac_aligned16(d.c))) &&
                (!d.bias || ac_aligned16(d.bias)) &&
                (!d.c16 || (d.ldc16 % 4 == 0 && ((uintptr_t)d.c16 & 7u) == 0));
    static size_t configured = 0;  // grow-only attribute (benign race: same value from any thread)
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void *)conv1d_window_kernel<WM, WN, C>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = 160 * 1024;
    }
    dim3 grid(d.B * p.tiles_l * p.tiles_n);
    hipLaunchKernelGGL((conv1d_window_kernel<WM, WN, C>), grid, dim3(NT), lds, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// N <= 64 (the input-gradient product of a stage whose input has 64 channels): a 256 x 64 output tile
// gives 4 waves only — one per SIMD, nothing to overlap a barrier or an LDS round trip with (878 TF).
// This variant runs 8 waves as two groups of 4 that share the window and take ALTERNATE K tiles
// (group g multiplies tiles 2p + g), each with its own accumulators; the groups are summed through
// LDS at the end.  Weight stages hold a pair of tiles (2 x 8 KB), two stages: 32 KB beside the window.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int C>
__global__ __launch_bounds__(2 * WM * WN * 64, 1) void conv1d_window_ks2_kernel(ConvWinParams p) {
    constexpr int GW = WM * WN, NT = 2 * GW * 64, BM = WM * 64, BN = WN * 64;
    constexpr int TILE = BN * 64;          // one 64-deep weight tile (elements)
    constexpr int BCH = 8 / WM;            // weight chunks per thread per tile pair
    constexpr int HALF = NT / 2, RPP = HALF / 8;   // threads per tile, rows per pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *win = reinterpret_cast<unsigned short *>(smem);
    const ac_convwin_desc &d = p.d;
    const int W = BM + d.k - 1;
    unsigned short *bst = win + W * C;  // 2 stages x 2 tiles x (BN x 64)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int grp = wave / GW, wv = wave % GW;
    const int wm = wv / WN, wn = wv % WN;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n;
    const int tl = (wg / p.tiles_n) % p.tiles_l;
    const int b = wg / (p.tiles_n * p.tiles_l);
    const int l0 = tl * BM;
    {
        const unsigned short *a = (const unsigned short *)d.a + (int64_t)b * d.a_batch_stride +
                                  (int64_t)(d.row_base + l0) * d.a_row_stride + d.a_col_off;
        const int total = W * (C / 8);
        for (int idx = t; idx < total; idx += NT) {
            const int r = idx / (C / 8), cc = idx % (C / 8);
            const u32x4 v = ac_gload<u32x4>(a + (int64_t)r * d.a_row_stride + cc * 8);
            *(u32x4 *)(win + win_off<C>(r, cc)) = v;
        }
    }
    // weight pair loader: thread -> tile (t / HALF), rows ((t % HALF) >> 3) + RPP i, chunk t & 7
    const unsigned short *wptr = (const unsigned short *)d.w;
    const int ltile = t / HALF, tt = t % HALF;
    int64_t wbase[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        int n = tn * BN + (tt >> 3) + RPP * i;
        n = n < d.N ? n : d.N - 1;
        wbase[i] = (int64_t)n * d.w_row_stride + 8 * (tt & 7);
    }
    const int ctiles = C / 64;
    const int nkt = d.k * ctiles;
    const int npairs = (nkt + 1) / 2;
    auto wload = [&](int pair, u32x4 (&v)[BCH], unsigned &mask) {
        int kt = 2 * pair + ltile;
        mask = (pair < npairs && kt < nkt) ? 0xFFFFFFFFu : 0u;
        kt = kt < nkt ? kt : nkt - 1;
        const int tap = kt / ctiles, c0 = (kt % ctiles) * 64;
        const int64_t ko = (int64_t)(d.flip ? d.k - 1 - tap : tap) * d.w_tap_stride + c0;
#pragma unroll
        for (int i = 0; i < BCH; ++i) v[i] = ac_gload<u32x4>(wptr + wbase[i] + ko);
    };
    auto wstore = [&](unsigned short *stage, const u32x4 (&v)[BCH], unsigned mask) {
        const int c = tt & 7;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = (tt >> 3) + RPP * i;
            u32x4 w = v[i];
            w[0] &= mask; w[1] &= mask; w[2] &= mask; w[3] &= mask;
            *(u32x4 *)(stage + ltile * TILE + r * 64 + ((c ^ ((r >> 1) & 7)) << 3)) = w;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto compute = [&](int pair, const unsigned short *stage) {
        int kt = 2 * pair + grp;
        kt = kt < nkt ? kt : nkt - 1;   // past-the-end tile: finite A rows x zero weights
        const unsigned short *bt = stage + grp * TILE;
        const int tap = kt / ctiles, cc0 = (kt % ctiles) * 8;
        const int r0 = wm * 64 + li + tap, r1 = r0 + 32;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cc = cc0 + 2 * s + lh;
            const bf16x8 a0 = *(const bf16x8 *)(win + win_off<C>(r0, cc));
            const bf16x8 a1 = *(const bf16x8 *)(win + win_off<C>(r1, cc));
            const int n0 = wn * 64 + li, n1 = n0 + 32;
            const int chunk = 2 * s + lh;
            const bf16x8 b0 = *(const bf16x8 *)(bt + n0 * 64 + ((chunk ^ ((n0 >> 1) & 7)) << 3));
            const bf16x8 b1 = *(const bf16x8 *)(bt + n1 * 64 + ((chunk ^ ((n1 >> 1) & 7)) << 3));
            acc[0][0] = AC_MFMA16(a0, b0, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0, b1, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1, b0, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1, b1, acc[1][1]);
        }
    };
    unsigned short *S0 = bst, *S1 = bst + 2 * TILE;
    u32x4 rb0[BCH], rb1[BCH];
    unsigned m0, m1;
    wload(0, rb0, m0);
    wstore(S0, rb0, m0);
    __syncthreads();
    wload(1, rb0, m0);
    for (int pr = 0; pr < npairs; pr += 2) {
        wload(pr + 2, rb1, m1);
        __builtin_amdgcn_sched_barrier(0);
        compute(pr, S0);
        wstore(S1, rb0, m0);
        __syncthreads();
        wload(pr + 3, rb0, m0);
        __builtin_amdgcn_sched_barrier(0);
        compute(pr + 1, S1);   // a pair past the end holds zero weights
        wstore(S0, rb1, m1);
        __syncthreads();
    }
    // ---- sum the two groups through the (idle) window, then group 0 writes the tile
    float *xch = smem + wv * 4096;
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) xch[((i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += xch[((i * 2 + j) * 16 + e) * 64 + lane];
    }
    __syncthreads();
    if (grp == 1) return;
    float *cb = d.c + ((int64_t)b * d.L + l0) * d.ldc;
    if (p.vec_epi) {
        float *wbuf = smem + wv * 2048;
        const int rsub = lane >> 4, c4 = 4 * (lane & 15);
        const int n = tn * BN + wn * 64 + c4;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (d.bias && n < d.N) bias4 = *(const f32x4 *)(d.bias + n);
#pragma unroll
        for (int sa = 0; sa < 2; ++sa) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                wbuf[r * 64 + li] = acc[sa][0][e];
                wbuf[r * 64 + 32 + li] = acc[sa][1][e];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + rsub;
                f32x4 v = *(const f32x4 *)(wbuf + r * 64 + c4) + bias4;
                if (n < d.N) {
                    const int64_t row = wm * 64 + sa * 32 + r;
                    if (d.c16) {
                        ushort4 h;
                        h.x = win_bf16(v[0]); h.y = win_bf16(v[1]); h.z = win_bf16(v[2]); h.w = win_bf16(v[3]);
                        *(ushort4 *)((unsigned short *)d.c16 + ((int64_t)b * d.L + l0 + row) * d.ldc16 + n) = h;
                    }
                    if (d.c) {
                        f32x4 *dst = (f32x4 *)(cb + row * d.ldc + n);
                        if (d.accumulate) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
        return;
    }
    const int nn0 = tn * BN + wn * 64 + li, nn1 = nn0 + 32;
    const float bias0 = (d.bias && nn0 < d.N) ? d.bias[nn0] : 0.f;
    const float bias1 = (d.bias && nn1 < d.N) ? d.bias[nn1] : 0.f;
#pragma unroll
    for (int sa = 0; sa < 2; ++sa) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = wm * 64 + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v0 = acc[sa][0][e] + bias0, v1 = acc[sa][1][e] + bias1;
            if (d.c16) {
                unsigned short *r16 = (unsigned short *)d.c16 + ((int64_t)b * d.L + l0 + m) * d.ldc16;
                if (nn0 < d.N) r16[nn0] = win_bf16(v0);
                if (nn1 < d.N) r16[nn1] = win_bf16(v1);
            }
            if (d.c) {
                float *row = cb + (int64_t)m * d.ldc;
                if (nn0 < d.N) row[nn0] = d.accumulate ? row[nn0] + v0 : v0;
                if (nn1 < d.N) row[nn1] = d.accumulate ? row[nn1] + v1 : v1;
            }
        }
    }
}

template <int WM, int WN, int C>
size_t ks2_lds_bytes(int k) {
    return ((size_t)(WM * 64 + k - 1) * C + 4 * WN * 64 * 64) * sizeof(short);
}

template <int WM, int WN, int C>
int launch_ks2(ConvWinParams &p, hipStream_t stream) {
    const ac_convwin_desc &d = p.d;
    const size_t lds = ks2_lds_bytes<WM, WN, C>(d.k);
    // (the group exchange parks WM*WN accumulator tiles of 16 KB in the window)
    if (lds > 160 * 1024 || (size_t)(WM * 64 + d.k - 1) * C * 2 < (size_t)WM * WN * 16384) return AC_EINVAL;
    p.tiles_l = d.L / (WM * 64);
    p.tiles_n = (d.N + WN * 64 - 1) / (WN * 64);
    p.cchunks = C / 8;
    p.vec_epi = (d.N % 4 == 0) && (!d.c || ((d.ldc % 4 == 0) && ac_aligned16(d.c))) &&
                (!d.bias || ac_aligned16(d.bias)) &&
                (!d.c16 || (d.ldc16 % 4 == 0 && ((uintptr_t)d.c16 & 7u) == 0));
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)conv1d_window_ks2_kernel<WM, WN, C>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    hipLaunchKernelGGL((conv1d_window_ks2_kernel<WM, WN, C>), dim3(d.B * p.tiles_l * p.tiles_n),
                       dim3(2 * WM * WN * 64), lds, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 (math mode bf16x3) window conv in ONE launch: A and the taps come as (hi, lo) bf16 planes
// and every fragment pair is three MFMAs (lo*hi + hi*lo + hi*hi, fp32 accumulate) — against three
// passes of the kernel above this reads each operand plane once, writes the output once and carries
// 1.5x more MFMA work per LDS byte.  Two window planes do not fit beside the weight stages for the long
// taps, so the K loop is cut into CHUNKS: a chunk = 64 input channels x up to TC taps, whose window
// [(BM + TC - 1) rows x 64 channels x 2 planes] is (re)loaded between chunks while the accumulators
// stay in registers.  TC is chosen by the host so that the chunk fits the 160 KB LDS (k = 251 at
// 64 channels: two chunks of 129 + 122 taps).
// ---------------------------------------------------------------------------------------------
// GR = 2 (N <= 64: the tile has only WM x 1 waves): a second group of waves works on the same output tile
// and the same window but on the ODD taps of the chunk (group 0: even taps), each group with its own
// weight stages; the groups add their accumulators through LDS at the end.  Two waves per SIMD instead
// of one, which the 4-wave tile lacked (one wave cannot cover its own LDS latency).
template <int WM, int WN, bool S16, int GR>
__global__ __launch_bounds__(GR *WM *WN * 64, 1) void conv1d_window_x3_kernel(ConvWinParams p, int TC) {
    constexpr int GT = WM * WN * 64, NT = GR * GT, BM = WM * 64, BN = WN * 64, CC = 64;
    constexpr int BCH = BN * 8 / GT;          // weight chunks per thread per plane per K tile (per group)
    constexpr int WT = BN * 64;               // one weight plane tile (elements)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *win_h = reinterpret_cast<unsigned short *>(smem);
    const ac_convwin_desc &d = p.d;
    const int Wrows = (d.L < BM ? (BM / d.L) * (d.L + TC - 1) : BM + TC - 1);
    unsigned short *win_l = win_h + Wrows * CC;
    const int t = threadIdx.x, lane = t & 63;
    const int grp = (t >> 6) / (WM * WN), wave = (t >> 6) % (WM * WN), tg = t % GT;
    unsigned short *bst = win_l + Wrows * CC + grp * 4 * WT;  // per group: 2 stages x (hi tile | lo tile)
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n;
    // A tile is BM consecutive rows of the [B*L] output.  L >= BM: a slice of one sample.  L < BM
    // (SpectraNet stages 4-5, L = 64 / 16): BM / L whole samples, each with its own zero-padded window.
    const int R0 = (wg / p.tiles_n) * BM;
    const int b = R0 / d.L, l0 = R0 - b * d.L;
    const int Ls = d.L < BM ? d.L : BM, spt = BM / Ls;   // rows per sample in the tile, samples per tile
    const int lsh = 31 - __builtin_clz(Ls);               // Ls is a power of two when spt > 1

    const unsigned short *aptr = (const unsigned short *)d.a + (int64_t)b * d.a_batch_stride +
                                 (int64_t)(d.row_base + l0) * d.a_row_stride + d.a_col_off;
    const unsigned short *wptr = (const unsigned short *)d.w;
    int64_t wbase[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        int n = tn * BN + (tg >> 3) + (GT / 8) * i;
        n = n < d.N ? n : d.N - 1;
        wbase[i] = (int64_t)n * d.w_row_stride + 8 * (tg & 7);
    }

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 acc[2][2];      // 32x32x16 form: [row half][column half] of the wave's 64 x 64 tile
    f32x4 acs[4][4];       // 16x16x32 form (S16): [16-row block][16-column block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acs[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // gridDim.y > 1: the channel chunks are split over workgroups (small grids: SpectraNet stage 5 has 32 row
    // tiles), partial tiles are added to the output with atomics (the host zeroed it unless accumulate)
    const int cchunks = d.C / CC, csplit = gridDim.y;
    const int cper = (cchunks + csplit - 1) / csplit;
    const int c_begin = blockIdx.y * cper, c_end = (c_begin + cper) < cchunks ? (c_begin + cper) : cchunks;
    for (int cch = c_begin; cch < c_end; ++cch) {
        for (int t0 = 0; t0 < d.k; t0 += TC) {
            const int tc = (d.k - t0) < TC ? (d.k - t0) : TC;   // taps in this chunk (K tiles: one per tap)
            __syncthreads();   // every wave is done with the previous chunk's window and weight stages
            const int wr = Ls + tc - 1;   // window rows of one sample
            {
                const unsigned short *a = aptr + (int64_t)t0 * d.a_row_stride + cch * CC;
                // four index groups per trip, all eight loads issued before the first store: one group per trip was a
                // load round trip per trip (ld ld / s_waitcnt vmcnt(0) in the generated code), and this workgroup is
                // alone on its CU - nothing else covered it
                const int nidx = spt * wr * 8;
                for (int idx = t; idx < nidx; idx += 4 * NT) {
                    u32x4 vh[4], vl[4];
                    int off[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int id = idx + u * NT;
                        const int idc = id < nidx ? id : idx;      // clamped: a valid address, the value is not stored
                        const int rq = idc >> 3, cc = idc & 7;
                        const int sidx = spt > 1 ? rq / wr : 0, r = rq - sidx * wr;
                        const unsigned short *src = a + (int64_t)sidx * d.a_batch_stride + (int64_t)r * d.a_row_stride + cc * 8;
                        vh[u] = ac_gload<u32x4>(src);
                        vl[u] = ac_gload<u32x4>(src + d.a_lo_off);
                        off[u] = id < nidx ? win_off<64>(rq, cc) : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (off[u] >= 0) {
                            *(u32x4 *)(win_h + off[u]) = vh[u];
                            *(u32x4 *)(win_l + off[u]) = vl[u];
                        }
                }
            }
            // K tile kt of this group = tap GR * kt + grp of the chunk; a tile past the chunk's last tap loads
            // the last tap (in bounds) and is zeroed when stored
            const int ntile = (tc + GR - 1) / GR;
            auto wload = [&](int kt, u32x4 (&v)[2 * BCH]) {
                int tl = GR * kt + grp;
                tl = tl < tc ? tl : tc - 1;
                const int tap = t0 + tl;
                const int64_t ko = (int64_t)(d.flip ? d.k - 1 - tap : tap) * d.w_tap_stride + cch * CC;
#pragma unroll
                for (int i = 0; i < BCH; ++i) {
                    v[i] = ac_gload<u32x4>(wptr + wbase[i] + ko);
                    v[BCH + i] = ac_gload<u32x4>(wptr + wbase[i] + ko + d.w_lo_off);
                }
            };
            auto wstore = [&](unsigned short *stage, int kt, u32x4 (&v)[2 * BCH]) {
                const int c = tg & 7;
                if (GR > 1 && GR * kt + grp >= tc) {
#pragma unroll
                    for (int i = 0; i < 2 * BCH; ++i) v[i] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int i = 0; i < BCH; ++i) {
                    const int r = (tg >> 3) + (GT / 8) * i;
                    const int off = r * 64 + ((c ^ ((r >> 1) & 7)) << 3);
                    *(u32x4 *)(stage + off) = v[i];
                    *(u32x4 *)(stage + WT + off) = v[BCH + i];
                }
            };
            // One K tile (one tap x 64 channels) = 4 steps of 16 channels; the operand reads of step s + 1
            // are issued ahead of the 12 MFMAs of step s (register double buffer, pinned by sched_barriers:
            // left alone the scheduler sinks every read to just before its first use), so only step 0's
            // reads are exposed.
            // 16x16x32 form of the same tile: a K tile = 2 steps of 32 channels; per step the wave reads its
            // 8 B fragments (4 column blocks x (hi, lo)) once and walks the 4 row blocks in two halves, the
            // next half's A fragments / next step's B fragments in flight under the 24 MFMAs of this half.
            auto compute16 = [&](int kt, const unsigned short *bt) {
                const int lr = lane & 15, g = lane >> 4;
                int ar[4], bo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = wm * 64 + 16 * i + lr;
                    ar[i] = (m >> lsh) * wr + (m & (Ls - 1)) + kt;
                    bo[i] = wn * 64 + 16 * i + lr;
                }
                bf16x8 Bf[2][8], Af[2][4];   // Bf: [hi j0..3 | lo j0..3] ; Af: [hi i0 i1 | lo i0 i1] of one half
                auto ldB = [&](int s2, bf16x8 (&Bv)[8]) {
                    const int cc = 4 * s2 + g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o = bo[j] * 64 + ((cc ^ ((bo[j] >> 1) & 7)) << 3);
                        Bv[j] = *(const bf16x8 *)(bt + o);
                        Bv[4 + j] = *(const bf16x8 *)(bt + WT + o);
                    }
                };
                auto ldA = [&](int s2, int h, bf16x8 (&A)[4]) {
                    const int cc = 4 * s2 + g;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int o = win_off<64>(ar[2 * h + i], cc);
                        A[i] = *(const bf16x8 *)(win_h + o);
                        A[2 + i] = *(const bf16x8 *)(win_l + o);
                    }
                };
                auto mm = [&](int h, const bf16x8 (&A)[4], const bf16x8 (&Bv)[8]) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acs[2 * h + i][j] = AC_MFMA16S(A[2 + i], Bv[j], acs[2 * h + i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acs[2 * h + i][j] = AC_MFMA16S(A[i], Bv[4 + j], acs[2 * h + i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acs[2 * h + i][j] = AC_MFMA16S(A[i], Bv[j], acs[2 * h + i][j]);
                };
                ldB(0, Bf[0]);
                ldA(0, 0, Af[0]);
                ldA(0, 1, Af[1]);
                __builtin_amdgcn_sched_barrier(0);
                mm(0, Af[0], Bf[0]);
                __builtin_amdgcn_sched_barrier(0);
                ldB(1, Bf[1]);
                ldA(1, 0, Af[0]);
                __builtin_amdgcn_sched_barrier(0);
                mm(1, Af[1], Bf[0]);
                __builtin_amdgcn_sched_barrier(0);
                ldA(1, 1, Af[1]);
                __builtin_amdgcn_sched_barrier(0);
                mm(0, Af[0], Bf[1]);
                mm(1, Af[1], Bf[1]);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto compute32 = [&](int kt, const unsigned short *bt) {
                const int m0 = wm * 64 + li, m1 = m0 + 32;   // tile rows -> (sample, position) -> window rows
                const int r0 = (m0 >> lsh) * wr + (m0 & (Ls - 1)) + kt, r1 = (m1 >> lsh) * wr + (m1 & (Ls - 1)) + kt;
                const int n0 = wn * 64 + li, n1 = n0 + 32;
                bf16x8 fa[2][4], fb[2][4];   // [buffer][a0h a1h a0l a1l] / [b0h b1h b0l b1l]
                auto ld = [&](int s_, bf16x8 (&A)[4], bf16x8 (&Bv)[4]) {
                    const int cc = 2 * s_ + lh;
                    const int ao0 = win_off<64>(r0, cc), ao1 = win_off<64>(r1, cc);
                    const int bo0 = n0 * 64 + ((cc ^ ((n0 >> 1) & 7)) << 3), bo1 = n1 * 64 + ((cc ^ ((n1 >> 1) & 7)) << 3);
                    A[2] = *(const bf16x8 *)(win_l + ao0);
                    A[3] = *(const bf16x8 *)(win_l + ao1);
                    Bv[0] = *(const bf16x8 *)(bt + bo0);
                    Bv[1] = *(const bf16x8 *)(bt + bo1);
                    A[0] = *(const bf16x8 *)(win_h + ao0);
                    A[1] = *(const bf16x8 *)(win_h + ao1);
                    Bv[2] = *(const bf16x8 *)(bt + WT + bo0);
                    Bv[3] = *(const bf16x8 *)(bt + WT + bo1);
                };
                ld(0, fa[0], fb[0]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8(&A)[4] = fa[s & 1];
                    const bf16x8(&Bv)[4] = fb[s & 1];
                    if (s < 3) ld(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);   // reads of step s + 1 are in flight before step s's MFMAs
                    acc[0][0] = AC_MFMA16(A[2], Bv[0], acc[0][0]);
                    acc[0][1] = AC_MFMA16(A[2], Bv[1], acc[0][1]);
                    acc[1][0] = AC_MFMA16(A[3], Bv[0], acc[1][0]);
                    acc[1][1] = AC_MFMA16(A[3], Bv[1], acc[1][1]);
                    acc[0][0] = AC_MFMA16(A[0], Bv[2], acc[0][0]);
                    acc[0][1] = AC_MFMA16(A[0], Bv[3], acc[0][1]);
                    acc[1][0] = AC_MFMA16(A[1], Bv[2], acc[1][0]);
                    acc[1][1] = AC_MFMA16(A[1], Bv[3], acc[1][1]);
                    acc[0][0] = AC_MFMA16(A[0], Bv[0], acc[0][0]);
                    acc[0][1] = AC_MFMA16(A[0], Bv[1], acc[0][1]);
                    acc[1][0] = AC_MFMA16(A[1], Bv[0], acc[1][0]);
                    acc[1][1] = AC_MFMA16(A[1], Bv[1], acc[1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto compute = [&](int kt, const unsigned short *bt) {
                if constexpr (S16) compute16(kt, bt);
                else compute32(kt, bt);
            };
            unsigned short *S0 = bst, *S1 = bst + 2 * WT;
            const int last = ntile - 1;
            // window row offset of this group's K tile kt (clamped: a padded tile multiplies zero weights
            // with rows that must still be finite window data)
            auto tapoff = [&](int kt) { const int o = GR * kt + grp; return o < tc ? o : tc - 1; };
            u32x4 rb0[2 * BCH], rb1[2 * BCH];
            wload(0, rb0);
            wstore(S0, 0, rb0);
            __syncthreads();  // window + first weight tile visible
            wload(1 < last ? 1 : last, rb0);
            int kt = 0;   // two K tiles per trip, no exit between them (see conv1d_window_kernel)
            for (; kt + 1 < ntile; kt += 2) {
                wload(kt + 2 < last ? kt + 2 : last, rb1);
                __builtin_amdgcn_sched_barrier(0);
                compute(tapoff(kt), S0);
                wstore(S1, kt + 1, rb0);
                __syncthreads();
                wload(kt + 3 < last ? kt + 3 : last, rb0);
                __builtin_amdgcn_sched_barrier(0);
                compute(tapoff(kt + 1), S1);
                wstore(S0, kt + 2, rb1);
                __syncthreads();
            }
            if (kt < ntile) {
                compute(tapoff(kt), S0);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if constexpr (GR == 2) {   // group 1 hands its accumulators to group 0 through the (idle) window
        float *xch = smem + wave * 4096;
        if (grp == 1) {
            if constexpr (S16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xch[((i * 4 + j) * 4 + e) * 64 + lane] = acs[i][j][e];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) xch[((i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
            }
        }
        __syncthreads();
        if (grp == 1) return;
        if constexpr (S16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acs[i][j][e] += xch[((i * 4 + j) * 4 + e) * 64 + lane];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += xch[((i * 2 + j) * 16 + e) * 64 + lane];
        }
        // (a wave parks only its own 8 KB below and reads its own 16 KB above: no barrier needed)
    }

    // ---- epilogue (fp32 output only): out[R0 + m, n] (+)= acc (+ bias), 16-byte stores through LDS
    float *cb = d.c + (int64_t)R0 * d.ldc;
    float *wbuf = smem + wave * (GR == 2 ? 4096 : 2048);   // GR = 2: inside the wave's own exchange region
    const int rsub = lane >> 4, c4 = 4 * (lane & 15);
    const int n = tn * BN + wn * 64 + c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (d.bias && n < d.N && blockIdx.y == 0) bias4 = *(const f32x4 *)(d.bias + n);
#pragma unroll
    for (int sa = 0; sa < 2; ++sa) {
        if constexpr (S16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        wbuf[(16 * i + 4 * (lane >> 4) + e) * 64 + 16 * j + (lane & 15)] = acs[2 * sa + i][j][e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                wbuf[r * 64 + li] = acc[sa][0][e];
                wbuf[r * 64 + 32 + li] = acc[sa][1][e];
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + rsub;
            f32x4 v = *(const f32x4 *)(wbuf + r * 64 + c4) + bias4;
            if (n < d.N) {
                const int64_t row = wm * 64 + sa * 32 + r;
                f32x4 *dst = (f32x4 *)(cb + row * d.ldc + n);
                if (csplit > 1) {
                    float *df = (float *)dst;
                    atomicAdd(df, v[0]);
                    atomicAdd(df + 1, v[1]);
                    atomicAdd(df + 2, v[2]);
                    atomicAdd(df + 3, v[3]);
                } else {
                    if (d.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 3: the same product with the weight stream on a RING of four half-stages filled by LDS-DMA.
// Counters on the kernel above (profiles/r03_pmc_conv_counters.json): matrix pipe busy 0.59-0.64 of the cycles,
// every wave parked 0.28 of its time at s_waitcnt / s_barrier, 23 % of the LDS cycles bank conflicts.  Its K tile
// (one tap x 64 channels, 96 MFMAs per wave) ends in a barrier behind which all eight waves start the next tile by
// reading 16 fragments each before their first MFMA — the two waves of a SIMD in lock-step, so nothing covers it
// (~600 of ~3 100 cycles per tile) — and the weights travel global -> VGPR -> ds_write_b128.
// Here:
//   * a stage holds HALF a K tile (32 channels x BN rows x (hi, lo) = 16 KB at BN = 128) and the ring has four:
//     half j + 3 is fetched by LDS-DMA (two wave-instructions per wave, no VGPRs, no ds_write) while half j is
//     multiplied; `s_waitcnt vmcnt(2)` + a raw s_barrier per half publish half j + 2;
//   * so the fragments of half j + 1 are visible one barrier EARLY and are read under the MFMAs of half j, across
//     the barrier (A fragments come from the resident window and never depended on it): a wave leaves every barrier
//     with 24 MFMAs of operands in registers — no read burst in front of the matrix pipe;
//   * LDS images that are conflict-free for the 16x16x32 fragment shape: window rows (128 B) rotate their 16-byte
//     chunks by 2 * (row >> 1) (any tap offset), weight rows (64 B) by 2 * (row >> 2); the DMA writes lane-linear,
//     so the weight rotation is applied to each lane's SOURCE address (cdna_hip_programming.md rule 21).
// Same tiles, chunking, two-group (N <= 64) form and epilogue as the kernel above; WM = 4 only.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int win_off16(int r, int cc) { return r * 64 + (((cc + (r & ~1)) & 7) << 3); }
typedef __attribute__((address_space(3))) void cw_lds_void;
typedef __attribute__((address_space(1))) const void cw_gbl_cvoid;

// SHORT: L < BM (whole samples per tile): window rows per row block by the general (sample, position) map; else
// the wave's four row blocks are 16 rows apart and share one address (immediate offsets), ~6 VALU per half for A
template <int WM, int WN, int GR, bool SHORT, int NS>
__global__ __launch_bounds__(GR *WM *WN * 64, 1) void conv1d_window_x3r_kernel(ConvWinParams p, int TC) {
    constexpr int GT = WM * WN * 64, NT = GR * GT, BM = WM * 64, BN = WN * 64, CC = 64;
    constexpr int NWG = WM * WN;              // waves per group
    constexpr int HP = BN * 32, HS = 2 * HP;  // half-stage plane / half-stage (hi | lo), elements
    static_assert(WM == 4 && BN == 16 * NWG, "one 16-row DMA block per wave and plane");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *win_h = reinterpret_cast<unsigned short *>(smem);
    const ac_convwin_desc &d = p.d;
    const int tstep = d.tap_row_step > 1 ? d.tap_row_step : 1;   // window rows per tap (8: Toeplitz form, see the header)
    const int Wrows = (d.L < BM ? (BM / d.L) * (d.L + tstep * (TC - 1)) : BM + tstep * (TC - 1));
    unsigned short *win_l = win_h + Wrows * CC;
    const int t = threadIdx.x, lane = t & 63;
    const int grp = (t >> 6) / NWG, wave = (t >> 6) % NWG;
    unsigned short *ring = win_l + Wrows * CC + grp * NS * HS;   // NS half-stages per group
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, g = lane >> 4;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n;
    const int R0 = (wg / p.tiles_n) * BM;
    const int b = R0 / d.L, l0 = R0 - b * d.L;
    const int Ls = d.L < BM ? d.L : BM, spt = BM / Ls;
    const int lsh = 31 - __builtin_clz(Ls);

    const unsigned short *aptr = (const unsigned short *)d.a + (int64_t)b * d.a_batch_stride +
                                 (int64_t)(d.row_base + l0) * d.a_row_stride + d.a_col_off;
    // DMA source of this lane: weight row 16 * wave + (lane >> 2), the chunk that belongs at position lane & 3
    const unsigned short *wsrc;
    {
        int n = tn * BN + 16 * wave + (lane >> 2);
        n = n < d.N ? n : d.N - 1;
        const int chunk = ((lane & 3) - 2 * (lane >> 4)) & 3;
        wsrc = (const unsigned short *)d.w + (int64_t)n * d.w_row_stride + 8 * chunk;
    }
    // fragment read offsets (elements): weight rows 16 j + lr of the wave's 64 columns, chunk g rotated
    const int boff = (wn * 64 + lr) * 32 + (((g + 2 * (lr >> 2)) & 3) << 3);

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acs[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acs[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cchunks = d.C / CC, csplit = gridDim.y;
    const int cper = (cchunks + csplit - 1) / csplit;
    const int c_begin = blockIdx.y * cper, c_end = (c_begin + cper) < cchunks ? (c_begin + cper) : cchunks;
    for (int cch = c_begin; cch < c_end; ++cch) {
        for (int t0 = 0; t0 < d.k; t0 += TC) {
            const int tc = (d.k - t0) < TC ? (d.k - t0) : TC;
            __syncthreads();   // previous chunk: window and ring idle, every DMA drained
            const int wr = Ls + tstep * (tc - 1);
            {
                const unsigned short *a = aptr + (int64_t)t0 * tstep * d.a_row_stride + cch * CC;
                // four index groups per trip, all eight loads issued before the first store: one group per trip was a
                // load round trip per trip (ld ld / s_waitcnt vmcnt(0) in the generated code), and this workgroup is
                // alone on its CU - nothing else covered it
                const int nidx = spt * wr * 8;
                for (int idx = t; idx < nidx; idx += 4 * NT) {
                    u32x4 vh[4], vl[4];
                    int off[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int id = idx + u * NT;
                        const int idc = id < nidx ? id : idx;      // clamped: a valid address, the value is not stored
                        const int rq = idc >> 3, cc = idc & 7;
                        const int sidx = spt > 1 ? rq / wr : 0, r = rq - sidx * wr;
                        const unsigned short *src = a + (int64_t)sidx * d.a_batch_stride + (int64_t)r * d.a_row_stride + cc * 8;
                        vh[u] = ac_gload<u32x4>(src);
                        vl[u] = ac_gload<u32x4>(src + d.a_lo_off);
                        off[u] = id < nidx ? win_off16(rq, cc) : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (off[u] >= 0) {
                            *(u32x4 *)(win_h + off[u]) = vh[u];
                            *(u32x4 *)(win_l + off[u]) = vl[u];
                        }
                }
            }
            const int ntile = (tc + GR - 1) / GR, nhalf = 2 * ntile;
            // half jj of this group's sequence: tile jj >> 1 (tap GR * tile + grp of the chunk, clamped: a tile past
            // the chunk's last tap is fetched from the last tap and not multiplied), channels 32 * (jj & 1) ..
            auto tapl = [&](int jj) { const int o = GR * (jj >> 1) + grp; return o < tc ? o : tc - 1; };
            auto issue_dma = [&](int jslot, int jj) {
                const int tap = t0 + tapl(jj);
                const int64_t ko = (int64_t)(d.flip ? d.k - 1 - tap : tap) * d.w_tap_stride + cch * CC + 32 * (jj & 1);
                unsigned short *dst = ring + (jslot % NS) * HS + wave * 16 * 32;
                __builtin_amdgcn_global_load_lds((cw_gbl_cvoid *)(wsrc + ko), (cw_lds_void *)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((cw_gbl_cvoid *)(wsrc + ko + d.w_lo_off), (cw_lds_void *)(dst + HP), 16, 0, 0);
            };
            int ar[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = wm * 64 + 16 * i + lr;
                ar[i] = (m >> lsh) * wr + (m & (Ls - 1));
            }
            bf16x8 Bf[2][8], Af[2][2][4];   // [parity of the half][row half h][hi i0 i1 | lo i0 i1]
            auto ldB = [&](int jj, bf16x8 (&Bv)[8]) {
                const unsigned short *bt = ring + (jj % NS) * HS + boff;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Bv[j] = *(const bf16x8 *)(bt + j * 16 * 32);
                    Bv[4 + j] = *(const bf16x8 *)(bt + HP + j * 16 * 32);
                }
            };
            auto ldA = [&](int jj, int h, bf16x8 (&A)[4]) {
                const int to = tstep * tapl(jj), cc = 4 * (jj & 1) + g;
                if constexpr (SHORT) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int o = win_off16(ar[2 * h + i] + to, cc);
                        A[i] = *(const bf16x8 *)(win_h + o);
                        A[2 + i] = *(const bf16x8 *)(win_l + o);
                    }
                } else {
                    // rows r + 16 i: the rotation only sees (r & 6), so one address serves the four row blocks
                    const int r = ar[0] + to;
                    const int o = r * 64 + (((cc + (r & 6)) & 7) << 3);
                    const unsigned short *ph = win_h + o, *pl = win_l + o;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        A[i] = *(const bf16x8 *)(ph + (2 * h + i) * 1024);
                        A[2 + i] = *(const bf16x8 *)(pl + (2 * h + i) * 1024);
                    }
                }
            };
            auto mm = [&](int h, const bf16x8 (&A)[4], const bf16x8 (&Bv)[8]) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acs[2 * h + i][j] = AC_MFMA16S(A[2 + i], Bv[j], acs[2 * h + i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acs[2 * h + i][j] = AC_MFMA16S(A[i], Bv[4 + j], acs[2 * h + i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acs[2 * h + i][j] = AC_MFMA16S(A[i], Bv[j], acs[2 * h + i][j]);
            };
            const int lasth = nhalf - 1;
#pragma unroll
            for (int q_ = 0; q_ < NS - 1; ++q_) issue_dma(q_, q_ < lasth ? q_ : lasth);
            __syncthreads();   // window visible, halves 0 .. NS-2 landed (a __syncthreads waits for the DMA queue as well)
            ldB(0, Bf[0]);
            ldA(0, 0, Af[0][0]);
            ldA(0, 1, Af[0][1]);
            // one half: MFMAs of half j on its own register set; all 16 fragments of half j + 1 are read into the
            // other set under them (two full A / B sets: 96 operand registers), so the wait in front of the next
            // half's first MFMA is for reads that are 24 MFMAs old
            auto half = [&](int j, const bf16x8 (&Bcur)[8], bf16x8 (&Bnext)[8], const bf16x8 (&Acur)[2][4],
                            bf16x8 (&Anext)[2][4]) {
                const int jn = j + 1 < lasth ? j + 1 : lasth;
                const bool valid = GR == 1 || GR * (j >> 1) + grp < tc;
                issue_dma(j + NS - 1, j + NS - 1 < lasth ? j + NS - 1 : lasth);
                __builtin_amdgcn_sched_barrier(0);
                if (valid) mm(0, Acur[0], Bcur);
                ldB(jn, Bnext);
                ldA(jn, 0, Anext[0]);
                ldA(jn, 1, Anext[1]);
                if (valid) mm(1, Acur[1], Bcur);
                // one fragment read per three MFMAs: issued back to back the 16 reads hold this wave's issue slot for
                // ~100 cycles in which it feeds the matrix pipe nothing (and its SIMD partner is at the same point)
                // (two MFMAs per read: the last read is issued 16 MFMAs before the half ends, so the wait in front of the
                //  next half's first MFMA finds it landed)
#pragma unroll
                for (int q_ = 0; q_ < 16; ++q_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_barrier(0);
                // all but the DMA instructions of the last NS - 3 halves are done: half j + 2 has landed in every
                // wave's share of the ring
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 3)) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int j = 0; j < nhalf; j += 2) {
                half(j, Bf[0], Bf[1], Af[0], Af[1]);
                half(j + 1, Bf[1], Bf[0], Af[1], Af[0]);
            }
        }
    }
    __syncthreads();
    if constexpr (GR == 2) {   // group 1 hands its accumulators to group 0 through the (idle) window
        float *xch = smem + wave * 4096;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xch[((i * 4 + j) * 4 + e) * 64 + lane] = acs[i][j][e];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acs[i][j][e] += xch[((i * 4 + j) * 4 + e) * 64 + lane];
    }
    float *cb = d.c + (int64_t)R0 * d.ldc;
    float *wbuf = smem + wave * (GR == 2 ? 4096 : 2048);
    const int rsub = lane >> 4, c4 = 4 * (lane & 15);
    const int n = tn * BN + wn * 64 + c4;
    // column n -> offset inside an output row (plain, or blocks of c_block columns c_block_stride apart)
    const int64_t ncol = d.c_block > 0 ? (int64_t)(n / d.c_block) * d.c_block_stride + n % d.c_block : n;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (d.bias && n < d.N && blockIdx.y == 0) bias4 = *(const f32x4 *)(d.bias + n);
#pragma unroll
    for (int sa = 0; sa < 2; ++sa) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    wbuf[(16 * i + 4 * (lane >> 4) + e) * 64 + 16 * j + (lane & 15)] = acs[2 * sa + i][j][e];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + rsub;
            f32x4 v = *(const f32x4 *)(wbuf + r * 64 + c4) + bias4;
            if (n < d.N) {
                const int64_t row = wm * 64 + sa * 32 + r;
                f32x4 *dst = (f32x4 *)(cb + row * d.ldc + ncol);
                if (csplit > 1) {
                    float *df = (float *)dst;
                    atomicAdd(df, v[0]);
                    atomicAdd(df + 1, v[1]);
                    atomicAdd(df + 2, v[2]);
                    atomicAdd(df + 3, v[3]);
                } else {
                    if (d.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <int WM, int WN, bool S16 = true, int GR = 1>
int launch_x3(ConvWinParams &p, hipStream_t stream) {
    if (S16 && p.d.variant == 2) return launch_x3<WM, WN, false, 1>(p, stream);   // variant 2: the 32x32x16 form
    if (GR == 2 && p.d.variant == 3) return launch_x3<WM, WN, S16, 1>(p, stream);    // variant 3: one wave group
    constexpr int BM = WM * 64, BN = WN * 64, NT = GR * WM * WN * 64;
    const ac_convwin_desc &d = p.d;
    const size_t stages = (size_t)GR * 2 * 2 * BN * 64 * sizeof(short);
    const int rows_budget = (int)((160 * 1024 - stages) / (2 * 64 * sizeof(short)));
    const int spt = d.L < BM ? BM / d.L : 1, Ls = d.L < BM ? d.L : BM;
    int TC = rows_budget / spt - Ls + 1;
    if (TC < 1) return AC_EINVAL;
    if (TC > d.k) TC = d.k;
    const size_t lds = (size_t)spt * (Ls + TC - 1) * 64 * 2 * sizeof(short) + stages;
    if (lds < (size_t)WM * WN * (GR == 2 ? 16384 : 8192)) return AC_EINVAL;   // epilogue: 8 KB per wave (group exchange: 16)
    p.tiles_l = d.L / BM;
    p.tiles_n = (d.N + BN - 1) / BN;
    p.cchunks = 8;
    p.vec_epi = 1;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)conv1d_window_x3_kernel<WM, WN, S16, GR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    const int row_tiles = (int)(((int64_t)d.B * d.L) / BM);
    // small grids (<= one workgroup per two CUs) with several channel chunks: split the chunks over
    // gridDim.y and add the partial tiles with atomics
    int csplit = 1;
    const int wgs = row_tiles * p.tiles_n, cchunks = d.C / 64;
    if (wgs <= 128 && cchunks >= 4 && d.ldc == d.N) {
        csplit = 256 / wgs;
        if (csplit > cchunks / 2) csplit = cchunks / 2;
        if (csplit > 4) csplit = 4;
        if (csplit < 1) csplit = 1;
    }
    if (csplit > 1 && !d.accumulate) {
        hipError_t e = hipMemsetAsync(d.c, 0, (size_t)d.B * d.L * d.ldc * sizeof(float), stream);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    hipLaunchKernelGGL((conv1d_window_x3_kernel<WM, WN, S16, GR>), dim3(wgs, csplit), dim3(NT), lds, stream, p, TC);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

template <int WM, int WN, int GR, bool SHORT = false, int NS = 4>
int launch_x3r(ConvWinParams &p, hipStream_t stream) {
    constexpr int BM = WM * 64, BN = WN * 64, NT = GR * WM * WN * 64;
    const ac_convwin_desc &d = p.d;
    if (!SHORT && d.L < BM) return launch_x3r<WM, WN, GR, true, NS>(p, stream);
    if (NS == 4 && d.variant == 6) return launch_x3r<WM, WN, GR, SHORT, 5>(p, stream);   // A/B: a fifth half-stage
    const size_t stages = (size_t)GR * NS * 2 * BN * 32 * sizeof(short);   // ring of NS half-stages per group
    const int rows_budget = (int)((160 * 1024 - stages) / (2 * 64 * sizeof(short)));
    const int spt = d.L < BM ? BM / d.L : 1, Ls = d.L < BM ? d.L : BM;
    const int tstep = d.tap_row_step > 1 ? d.tap_row_step : 1;
    int TC = (rows_budget / spt - Ls) / tstep + 1;
    if (TC < 1) return AC_EINVAL;
    if (TC > d.k) TC = d.k;
    const size_t lds = (size_t)spt * (Ls + tstep * (TC - 1)) * 64 * 2 * sizeof(short) + stages;
    if (lds < (size_t)WM * WN * (GR == 2 ? 16384 : 8192)) return AC_EINVAL;
    p.tiles_l = d.L / BM;
    p.tiles_n = (d.N + BN - 1) / BN;
    p.cchunks = 8;
    p.vec_epi = 1;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void *)conv1d_window_x3r_kernel<WM, WN, GR, SHORT, NS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    const int row_tiles = (int)(((int64_t)d.B * d.L) / BM);
    int csplit = 1;
    const int wgs = row_tiles * p.tiles_n, cchunks = d.C / 64;
    if (wgs <= 128 && cchunks >= 4 && d.ldc == d.N && d.c_block == 0) {
        csplit = 256 / wgs;
        if (csplit > cchunks / 2) csplit = cchunks / 2;
        if (csplit > 4) csplit = 4;
        if (csplit < 1) csplit = 1;
    }
    if (csplit > 1 && !d.accumulate) {
        hipError_t e = hipMemsetAsync(d.c, 0, (size_t)d.B * d.L * d.ldc * sizeof(float), stream);
        if (e != hipSuccess) return -(int)e - 2000;
    }
    hipLaunchKernelGGL((conv1d_window_x3r_kernel<WM, WN, GR, SHORT, NS>), dim3(wgs, csplit), dim3(NT), lds, stream, p, TC);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

template <int C>
int dispatch(ConvWinParams &p, hipStream_t stream) {
    const ac_convwin_desc &d = p.d;
    const size_t win256 = (size_t)(256 + d.k - 1) * C * 2;
    const bool wide = d.N > 64;
    // N <= 64 with a handful of taps: the window load dominates and the generic gather-GEMM is faster
    // (0.080 vs 0.105 ms at k = 3); AC_EINVAL sends the caller there
    if (!wide && d.k < 8) return AC_EINVAL;
    // N <= 64, long tap loops: two K-parity groups of 4 waves (variant = 1 forces the 4-wave kernel)
    if (!wide && d.L % 256 == 0 && d.k >= 8 && p.d.variant != 1 &&
        launch_ks2<4, 1, C>(p, stream) == AC_OK)
        return AC_OK;
    if (d.L % 256 == 0 && win256 + 2 * (wide ? 128 : 64) * 64 * 2 <= 160 * 1024)
        return wide ? launch<4, 2, C>(p, stream) : launch<4, 1, C>(p, stream);
    // 128-row tiles (the 256-row window does not fit): 4 waves alone on the CU -> two K-parity groups
    if (wide && d.L % 128 == 0 && d.k >= 8 && p.d.variant != 1 && launch_ks2<2, 2, C>(p, stream) == AC_OK)
        return AC_OK;
    if (d.L % 128 == 0) return wide ? launch<2, 2, C>(p, stream) : launch<2, 1, C>(p, stream);
    return AC_EINVAL;
}

}  // namespace

extern "C" int ac_conv1d_window_x3(const ac_convwin_desc *dp, ac_stream_t stream_) {
    if (!dp) return AC_EINVAL;
    ConvWinParams p;
    p.d = *dp;
    const ac_convwin_desc &d = p.d;
    if (!d.a || !d.w || !d.c || d.c16 || d.B <= 0 || d.L <= 0 || d.k <= 0 || d.N <= 0) return AC_EINVAL;
    if (d.a_lo_off == 0 || d.w_lo_off == 0) return AC_EINVAL;
    // L a multiple of the 128 / 256-row tile, or a power of two below it (whole samples per 256-row tile)
    const bool short_seq = d.L < 128 && d.L >= 8 && (d.L & (d.L - 1)) == 0 && (((int64_t)d.B * d.L) % 256) == 0;
    if (d.C <= 0 || (d.C % 64) || ((d.L % 128) && !short_seq)) return AC_EINVAL;
    if (!ac_aligned16(d.a) || !ac_aligned16(d.w) || !ac_aligned16(d.c)) return AC_EALIGN;
    if ((d.a_row_stride % 8) || (d.a_batch_stride % 8) || (d.a_col_off % 8) || (d.w_row_stride % 8) ||
        (d.w_tap_stride % 8) || (d.a_lo_off % 8) || (d.w_lo_off % 8) || (d.N % 4) || (d.ldc % 4) ||
        (d.bias && !ac_aligned16(d.bias)))
        return AC_EALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    const bool wide = d.N > 64;
    // N <= 64: one column of waves only -> a second wave group on the odd taps (needs a few taps to share)
    const bool two = !wide && d.k >= 4;
    const bool extended = d.tap_row_step > 1 || d.c_block > 0;   // Toeplitz form / blocked output columns: ring kernel only
    if (d.tap_row_step < 0 || d.c_block < 0 || (d.c_block > 0 && ((d.c_block % 4) || (d.c_block_stride % 4))))
        return AC_EINVAL;
    if (extended && !((d.variant == 0 || d.variant == 5) && (d.L % 256 == 0 || short_seq) && (wide || two)))
        return AC_EINVAL;
    // default (variant 0 / 5): the ring kernel (weights by LDS-DMA into four half-stages, fragments prefetched across
    // the barrier); 6 = the same with a fifth half-stage; 4 = the round-2 kernel (two register-staged stages),
    // 2 / 3 = its 32x32x16 / one-group forms (A/B measurements, tests)
    if ((d.variant == 0 || d.variant == 5 || d.variant == 6) && (d.L % 256 == 0 || short_seq) && (wide || two))
        return wide ? launch_x3r<4, 2, 1>(p, stream) : launch_x3r<4, 1, 2>(p, stream);
    if (d.L % 256 == 0 || short_seq)
        return wide ? launch_x3<4, 2>(p, stream) : (two ? launch_x3<4, 1, true, 2>(p, stream) : launch_x3<4, 1>(p, stream));
    return wide ? launch_x3<2, 2>(p, stream) : (two ? launch_x3<2, 1, true, 2>(p, stream) : launch_x3<2, 1>(p, stream));
}

extern "C" int ac_conv1d_window_bf16(const ac_convwin_desc *dp, ac_stream_t stream_) {
    if (!dp) return AC_EINVAL;
    ConvWinParams p;
    p.d = *dp;
    const ac_convwin_desc &d = p.d;
    if (!d.a || !d.w || (!d.c && !d.c16) || d.B <= 0 || d.L <= 0 || d.k <= 0 || d.N <= 0) return AC_EINVAL;
    if (!d.c && d.accumulate) return AC_EINVAL;
    if (!ac_aligned16(d.a) || !ac_aligned16(d.w)) return AC_EALIGN;
    if ((d.a_row_stride % 8) || (d.a_batch_stride % 8) || (d.a_col_off % 8) || (d.w_row_stride % 8) ||
        (d.w_tap_stride % 8))
        return AC_EALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    switch (d.C) {
        case 64: return dispatch<64>(p, stream);
        case 128: return dispatch<128>(p, stream);
        case 256: return dispatch<256>(p, stream);
        default: return AC_EINVAL;
    }
}
