// Depthwise 7x7 convolution (pad 3) on NHWC fp32 planes — the ConvNeXt block's
// spatial mixing (timm convnext_tiny, called from astrominn.py:12-17).
//
// HBM-bound by arithmetic (49 MAC per element, 24.5 FLOP/B at bf16 — SURVEY §8d).  A workgroup owns
// one 32-channel slice of a sample: the zero-haloed H x W x 32 plane sits in LDS, every HBM byte is
// read once as 128-byte lines and written once.  Thread = (channel c = t&31, work item): an item is
// XS adjacent output pixels of one row, so a row of XS+6 LDS reads feeds 7*XS FMAs (register
// tiling: ~15 LDS reads per output instead of 49); the 49 taps of channel c live in registers.
#include "ac_common.h"

namespace {

constexpr int CG = 32;  // channels per workgroup

struct Geo {
    int H, W, C, nseg, Wp, np;  // nseg segments per row, padded pitch (pixels), padded plane pixels
};
// WC > 0: the plane is WC x WC at compile time (15 x 15 and 7 x 7, the ConvNeXt stages at 63 x 63
// input), so every index division below is by a constant.  With run-time extents each thread spent
// ~5000 instructions per sample on `q / Wp`, `q % Wp`, `p / W`, `it / nseg` (integer division is a
// ~35-instruction sequence on gfx950): more than the 1400 FMAs of its share of the convolution.
template <int XS, int WC = 0>
__host__ __device__ inline Geo make_geo(int H, int W, int C) {
    Geo g;
    if (WC > 0) H = W = WC;
    g.H = H;
    g.W = W;
    g.C = C;
    g.nseg = (W + XS - 1) / XS;
    g.Wp = g.nseg * XS + 6;
    g.np = (H + 6) * g.Wp;
    return g;
}

// Fill the zero-haloed plane: halo cells are LDS stores only; the interior is fetched eight pixels
// at a time per thread so that eight independent 128-byte-line loads are in flight per wave (the
// naive one-load-one-store loop was latency bound at 2 workgroups per CU).
__device__ __forceinline__ void load_padded_plane(float *plane, const float *src, const Geo &g,
                                                  int cglob, bool cvalid, int c, int ps) {
    for (int q = ps; q < g.np; q += 8) {
        const int yy = q / g.Wp - 3, xx = q % g.Wp - 3;
        if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) plane[q * CG + c] = 0.f;
    }
    const int HW = g.H * g.W;
    for (int p0 = ps; p0 < HW; p0 += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 8 * u;
            v[u] = src[(int64_t)(p < HW ? p : HW - 1) * g.C + (cvalid ? cglob : 0)];   // clamped, masked below
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 8 * u;
            if (p < HW) {
                const int py = p / g.W, px = p - py * g.W;
                plane[((py + 3) * g.Wp + px + 3) * CG + c] = cvalid ? v[u] : 0.f;
            }
        }
    }
}

// The same with 16-byte global loads: 8 threads fetch the 32 channels of a pixel as one 128-byte
// line, a wave covers 8 pixels per instruction (4x fewer load and LDS-store instructions), eight
// independent lines in flight per thread.  Needs C % 4 == 0 and a 16-byte aligned plane.
__device__ __forceinline__ void load_padded_plane_v4(float *plane, const float *src, const Geo &g, int cg0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int cq = threadIdx.x & 7, slot = threadIdx.x >> 3;   // 32 pixel slots
    const bool cv = cg0 + 4 * cq < g.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int q = slot; q < g.np; q += 32) {
        const int yy = q / g.Wp - 3, xx = q % g.Wp - 3;
        if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) *(f32x4 *)(plane + q * CG + 4 * cq) = zero;
    }
    const int HW = g.H * g.W;
    const float *sp = src + (cv ? cg0 + 4 * cq : 0);
    for (int p0 = slot; p0 < HW; p0 += 256) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 32 * u;
            v[u] = *(const f32x4 *)(sp + (int64_t)(p < HW ? p : HW - 1) * g.C);   // clamped, masked below
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 32 * u;
            if (p < HW) {
                const int py = p / g.W, px = p - py * g.W;
                *(f32x4 *)(plane + ((py + 3) * g.Wp + px + 3) * CG + 4 * cq) = cv ? v[u] : zero;
            }
        }
    }
}

// out[o] += sum_{ky,kx} W(ky,kx) * plane[(y+ky), (x0+o+kx)] with W = taps (forward) or the taps
// flipped on both axes (input gradient).  `pp` points at padded pixel (y, x0).
template <int XS, bool FLIP>
__device__ __forceinline__ void conv_rows(const float *pp, int Wp, const float (&wt)[49],
                                          float (&out)[XS]) {
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        float in[XS + 6];
#pragma unroll
        for (int j = 0; j < XS + 6; ++j) in[j] = pp[(ky * Wp + j) * CG];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float w = FLIP ? wt[(6 - ky) * 7 + (6 - kx)] : wt[ky * 7 + kx];
#pragma unroll
            for (int o = 0; o < XS; ++o) out[o] = fmaf(w, in[o + kx], out[o]);
        }
    }
}

template <int XS, int WC, bool V4>
__global__ __launch_bounds__(256) void dwconv7x7_fwd_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ y, int H, int W,
                                                            int C) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    const Geo g = make_geo<XS, WC>(H, W, C);
    const int b = blockIdx.x, cg = blockIdx.y;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cglob = cg * CG + c;
    const bool cvalid = cglob < C;
    const int HW = g.H * g.W;
    if (V4)
        load_padded_plane_v4(plane, x + (int64_t)b * HW * C, g, cg * CG);
    else
        load_padded_plane(plane, x + (int64_t)b * HW * C, g, cglob, cvalid, c, ps);
    float wt[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) wt[k] = w[k * C + (cvalid ? cglob : 0)];
    const float bv = (cvalid && bias) ? bias[cglob] : 0.f;
    __syncthreads();
    if (!cvalid) return;
    float *yb = y + (int64_t)b * HW * C;
    const int items = g.H * g.nseg;
    for (int it = ps; it < items; it += 8) {
        const int py = it / g.nseg, x0 = (it % g.nseg) * XS;
        float out[XS];
#pragma unroll
        for (int o = 0; o < XS; ++o) out[o] = bv;
        conv_rows<XS, false>(plane + (py * g.Wp + x0) * CG + c, g.Wp, wt, out);
#pragma unroll
        for (int o = 0; o < XS; ++o)
            if (x0 + o < g.W) yb[(int64_t)(py * g.W + x0 + o) * C + cglob] = out[o];
    }
}

// Backward.  A workgroup loops over SPB samples of one 32-channel slice.  Per sample:
//   phase 1: padded dy plane in LDS -> dx = correlation of dy with the flipped taps
//   phase 2: padded x plane in LDS  -> dw[ky,kx] += sum_o dy[o] * x[o + (ky-3, kx-3)]  (dy from L2)
// The 49 dw partial sums (+ dbias) of the thread's channel stay in registers across the samples,
// so the global atomics are 1/SPB of the per-sample count.
constexpr int SPB = 4;
template <int XS, int WC, bool V4>
__global__ __launch_bounds__(256) void dwconv7x7_bwd_kernel(const float *__restrict__ dy,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ w,
                                                            float *__restrict__ dx,
                                                            float *__restrict__ dw,
                                                            float *__restrict__ dbias, int B, int H,
                                                            int W, int C, const float *__restrict__ dres) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    const Geo g = make_geo<XS, WC>(H, W, C);
    const int HW = g.H * g.W;
    W = g.W;
    const int cg = blockIdx.y;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cglob = cg * CG + c;
    const bool cvalid = cglob < C;
    const int items = g.H * g.nseg;
    float wt[49], dwacc[49];
    float dbacc = 0.f;
#pragma unroll
    for (int k = 0; k < 49; ++k) {
        wt[k] = w[k * C + (cvalid ? cglob : 0)];
        dwacc[k] = 0.f;
    }
    for (int s = 0; s < SPB; ++s) {
        const int b = blockIdx.x * SPB + s;
        if (b >= B) break;
        const float *xb = x + (int64_t)b * HW * C;
        const float *dyb = dy + (int64_t)b * HW * C;
        float *dxb = dx + (int64_t)b * HW * C;
        __syncthreads();
        if (V4)
            load_padded_plane_v4(plane, dyb, g, cg * CG);
        else
            load_padded_plane(plane, dyb, g, cglob, cvalid, c, ps);
        __syncthreads();
        if (cvalid) {
            for (int it = ps; it < items; it += 8) {
                const int py = it / g.nseg, x0 = (it % g.nseg) * XS;
                float out[XS];
#pragma unroll
                for (int o = 0; o < XS; ++o) out[o] = 0.f;
                conv_rows<XS, true>(plane + (py * g.Wp + x0) * CG + c, g.Wp, wt, out);
#pragma unroll
                for (int o = 0; o < XS; ++o)
                    if (x0 + o < W) {
                        const int64_t off = (int64_t)(py * W + x0 + o) * C + cglob;
                        dxb[off] = dres ? out[o] + dres[(int64_t)b * HW * C + off] : out[o];
                    }
            }
        }
        __syncthreads();
        if (V4)
            load_padded_plane_v4(plane, xb, g, cg * CG);
        else
            load_padded_plane(plane, xb, g, cglob, cvalid, c, ps);
        __syncthreads();
        if (cvalid) {
            for (int it = ps; it < items; it += 8) {
                const int py = it / g.nseg, x0 = (it % g.nseg) * XS;
                float d[XS];
#pragma unroll
                for (int o = 0; o < XS; ++o) {
                    d[o] = (x0 + o < W) ? dyb[(int64_t)(py * W + x0 + o) * C + cglob] : 0.f;
                    dbacc += d[o];
                }
                const float *pp = plane + (py * g.Wp + x0) * CG + c;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    float in[XS + 6];
#pragma unroll
                    for (int j = 0; j < XS + 6; ++j) in[j] = pp[(ky * g.Wp + j) * CG];
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        float a = dwacc[ky * 7 + kx];
#pragma unroll
                        for (int o = 0; o < XS; ++o) a = fmaf(d[o], in[o + kx], a);
                        dwacc[ky * 7 + kx] = a;
                    }
                }
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

// ---------------------------------------------------------------------------------------------
// Whole-row kernels for the square W x W planes of the benchmark geometry (W = 15, 7).  A thread owns
// one channel and whole output ROWS, so the horizontal halo is known at compile time (taps that fall
// outside a row are simply not multiplied) and the vertical halo is a per-row predicate: the LDS
// plane is the bare [W][W][32 channels] image — 28.8 KB at 15 x 15 instead of 56 KB with halos, i.e.
// five workgroups per CU instead of two, which is what this latency-bound kernel (HBM load ->
// barrier -> LDS reads -> FMAs) lacked.  Per output row: 15 LDS reads feed up to 105 FMAs per tap row.
// Planes are fetched with 16-byte loads (8 threads = the 32 channels of a pixel = one 128-byte line).
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void load_plane_rows(float *plane, const float *src, int C, int cg0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int HW = W * W, NIT = (HW + 31) / 32;
    const int cq = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const bool cv = cg0 + 4 * cq < C;
    // unconditional loads from clamped addresses, masked when stored: a "load or zero" select makes
    // hipcc branch around every load and wait for each in turn (cdna_hip_programming.md s.5 trap (c))
    const float *sp = src + (cv ? cg0 + 4 * cq : 0);
    f32x4 v[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int p = slot + 32 * u;
        v[u] = *(const f32x4 *)(sp + (int64_t)(p < HW ? p : HW - 1) * C);
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int p = slot + 32 * u;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (p < HW) *(f32x4 *)(plane + p * CG + 4 * cq) = cv ? v[u] : zero;
    }
}

// out[o] += sum_kx wrow[kx] * in[o + kx - 3] over the taps that stay inside the row (compile time)
template <int W, bool FLIP>
__device__ __forceinline__ void row_taps(const float (&in)[W], const float (&wt)[49], int ky, float (&out)[W]) {
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
        const float w = FLIP ? wt[(6 - ky) * 7 + (6 - kx)] : wt[ky * 7 + kx];
#pragma unroll
        for (int o = 0; o < W; ++o)
            if (o + kx - 3 >= 0 && o + kx - 3 < W) out[o] = fmaf(w, in[o + kx - 3], out[o]);
    }
}

template <int W>
__global__ __launch_bounds__(256) void dwconv_rows_fwd_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ y, int C) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    constexpr int HW = W * W;
    const int b = blockIdx.x, cg = blockIdx.y;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cglob = cg * CG + c;
    const bool cvalid = cglob < C;
    load_plane_rows<W>(plane, x + (int64_t)b * HW * C, C, cg * CG);
    float wt[49];
    {
        const float *wp = w + (cvalid ? cglob : 0);   // running address (49 separate ones cost 98 registers)
#pragma unroll
        for (int k = 0; k < 49; ++k) {
            wt[k] = *wp;
            wp += C;
        }
    }
    const float bv = (cvalid && bias) ? bias[cglob] : 0.f;
    __syncthreads();
    if (!cvalid) return;
    float *yb = y + (int64_t)b * HW * C + cglob;
#pragma unroll 1
    for (int py = ps; py < W; py += 8) {
        float out[W];
#pragma unroll
        for (int o = 0; o < W; ++o) out[o] = bv;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int yy = py + ky - 3;
            if (yy >= 0 && yy < W) {
                float in[W];
#pragma unroll
                for (int j = 0; j < W; ++j) in[j] = plane[(yy * W + j) * CG + c];
                row_taps<W, false>(in, wt, ky, out);
            }
            // keep the LDS reads of the next tap row behind this row's FMAs: hoisted together, the seven
            // rows need 105 registers of inputs and the kernel drops to two waves per SIMD
            __builtin_amdgcn_sched_barrier(0);
        }
        float *yp = yb + (int64_t)py * W * C;   // one running address: W separate 64-bit addresses cost 2W registers
#pragma unroll
        for (int o = 0; o < W; ++o) {
            *yp = out[o];
            yp += C;
        }
    }
}

// Backward: both planes (dy, x) of a sample in LDS; per output row the thread forms dx (correlation of
// dy with the flipped taps) and adds its share of dw / dbias; the 49 + 1 partial sums stay in
// registers over the SPB samples of the workgroup and are reduced through LDS at the end.
template <int W>
__global__ __launch_bounds__(256) void dwconv_rows_bwd_kernel(const float *__restrict__ dy,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ w,
                                                              float *__restrict__ dx, float *__restrict__ dw,
                                                              float *__restrict__ dbias, int B, int C,
                                                              const float *__restrict__ dres) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    constexpr int HW = W * W;
    float *pdy = plane, *px = plane + HW * CG;
    const int cg = blockIdx.y;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cglob = cg * CG + c;
    const bool cvalid = cglob < C;
    float wt[49], dwacc[49];
    float dbacc = 0.f;
#pragma unroll
    for (int k = 0; k < 49; ++k) {
        wt[k] = w[k * C + (cvalid ? cglob : 0)];
        dwacc[k] = 0.f;
    }
    for (int s = 0; s < SPB; ++s) {
        const int b = blockIdx.x * SPB + s;
        if (b >= B) break;
        __syncthreads();
        load_plane_rows<W>(pdy, dy + (int64_t)b * HW * C, C, cg * CG);
        load_plane_rows<W>(px, x + (int64_t)b * HW * C, C, cg * CG);
        __syncthreads();
        if (cvalid) {
            float *dxb = dx + (int64_t)b * HW * C + cglob;
            for (int py = ps; py < W; py += 8) {
                float out[W], d[W];
#pragma unroll
                for (int o = 0; o < W; ++o) {
                    out[o] = 0.f;
                    d[o] = pdy[(py * W + o) * CG + c];
                    dbacc += d[o];
                }
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const int yy = py + ky - 3;
                    if (yy >= 0 && yy < W) {
                        float in[W];
#pragma unroll
                        for (int j = 0; j < W; ++j) in[j] = pdy[(yy * W + j) * CG + c];
                        row_taps<W, true>(in, wt, ky, out);
#pragma unroll
                        for (int j = 0; j < W; ++j) in[j] = px[(yy * W + j) * CG + c];
#pragma unroll
                        for (int kx = 0; kx < 7; ++kx) {
                            float a = dwacc[ky * 7 + kx];
#pragma unroll
                            for (int o = 0; o < W; ++o)
                                if (o + kx - 3 >= 0 && o + kx - 3 < W) a = fmaf(d[o], in[o + kx - 3], a);
                            dwacc[ky * 7 + kx] = a;
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < W; ++o) {
                    const int64_t off = (int64_t)(py * W + o) * C;
                    dxb[off] = dres ? out[o] + dres[(int64_t)b * HW * C + cglob + off] : out[o];
                }
            }
        }
    }
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
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += red[(q * 50 + k) * CG + c];
            if (k < 49)
                atomicAdd(&dw[k * C + cglob], sum);
            else if (dbias)
                atomicAdd(&dbias[cglob], sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Pipelined whole-row kernels (round 3).  The kernels above are load -> barrier -> compute -> store per
// workgroup with ~1.2 waves of workgroups on the chip: every CU's loads, FMAs and stores happen one after
// the other, so the launch runs at load + compute + store time (0.40 of HBM forward, 0.18 backward) although
// each phase alone is faster than the HBM stream.  Here a PERSISTENT workgroup walks several (sample,
// 32-channel slice) items with TWO plane buffers in LDS: the next item's plane arrives by LDS-DMA
// (global_load_lds_dwordx4: 8 lanes = the 128-byte line of one pixel's 32 channels, one wave-instruction =
// 8 pixels = 1 KiB, no VGPR round trip) while this item's rows are convolved and stored.  The DMA is waited
// for with a COUNTED vmcnt (this item's output stores, issued later, stay in flight) and a raw s_barrier —
// __syncthreads() would drain everything (cdna_hip_programming.md, "Pipelining across barriers").
// The channel slice of a workgroup is fixed, so its 49 taps are loaded once.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

template <int W>
struct PipeGeo {
    static constexpr int HW = W * W;
    static constexpr int NI = (HW + 7) / 8;          // DMA wave-instructions per plane (8 pixels each)
    static constexpr int PLANE = NI * 8 * CG;        // floats per plane buffer (pixels padded to 8)
};

// wave `wv` of `nw` issues instructions wv, wv + nw, ... of one plane.  A RUNTIME loop with running addresses on
// purpose: unrolled, hipcc hoists the 29 lane-constant 64-bit pixel offsets out of the item loop, spills them, and
// every reload's vmcnt(0) drains the queue in front of each DMA (cdna_hip_programming.md, Appendix B pitfalls).
template <int W>
__device__ __forceinline__ void dma_plane(float *buf, const float *src, int C, int wv, int nw) {
    constexpr int HW = W * W, NI = PipeGeo<W>::NI;
    const int lane = threadIdx.x & 63;
    const int pl = lane >> 3, ch = (lane & 7) * 4;
    const float *sp = src + ch;
#pragma unroll 1
    for (int j = wv; j < NI; j += nw) {
        int p = 8 * j + pl;
        p = p < HW ? p : HW - 1;     // the pad pixels of the last instruction re-read the last pixel (never used)
        __builtin_amdgcn_global_load_lds((gbl_cvoid *)(sp + (unsigned)(p * C)), (lds_void *)(buf + 8 * j * CG), 16, 0, 0);
    }
}

// SLOTS row slots per workgroup (threads = 32 channels x SLOTS), ROUNDS = rows per slot
// PROBE (diagnostic builds of the same kernel, variant 8 / 9 of ac_dwconv7x7_fwd_v; results are NOT a convolution):
// 1 = the memory side alone (each output row = the centre input row: DMA, LDS reads of one row, stores), 2 = no
// stores (taps only) — where the 27 us of the real kernel go (profiles/r03_dwconv_probe.txt).
template <int W, int SLOTS, int PROBE = 0>
__global__ __launch_bounds__(32 * SLOTS, (SLOTS * 64) / 256) void dwconv_pipe_fwd_kernel(const float *__restrict__ x,
                                                                      const float *__restrict__ w,
                                                                      const float *__restrict__ bias,
                                                                      float *__restrict__ y, int B, int C, int ncg) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    constexpr int HW = W * W, PL = PipeGeo<W>::PLANE, NWV = SLOTS / 2, ROUNDS = (W + SLOTS - 1) / SLOTS;
    constexpr int NI = PipeGeo<W>::NI;
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // item = (sample, slice): the slice of a workgroup is fixed (gridDim.x is a multiple of ncg)
    const int cg = blockIdx.x % ncg, b0 = blockIdx.x / ncg, bstep = gridDim.x / ncg;
    const int cglob = cg * CG + c;
    float wt[49];
    {
        const float *wp = w + cglob;
#pragma unroll
        for (int k = 0; k < 49; ++k) {
            wt[k] = *wp;
            wp += C;
        }
    }
    const float bv = bias ? bias[cglob] : 0.f;
    if (b0 >= B) return;
    dma_plane<W>(plane, x + (int64_t)b0 * HW * C + cg * CG, C, wv, NWV);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int b = b0; b < B; b += bstep) {
        const bool more = b + bstep < B;
        if (more) dma_plane<W>(plane + (cur ^ 1) * PL, x + (int64_t)(b + bstep) * HW * C + cg * CG, C, wv, NWV);
        const float *pb = plane + cur * PL + c;
        float *yb = y + (int64_t)b * HW * C + cglob;
#pragma unroll 1
        for (int r = 0; r < ROUNDS; ++r) {
            const int py_ = ps + SLOTS * r;
            const bool rv = py_ < W;
            const int py = rv ? py_ : W - 1;
            float out[W];
#pragma unroll
            for (int o = 0; o < W; ++o) out[o] = bv;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int yy = py + ky - 3;
                if (PROBE == 1 && ky != 3) continue;
                if (yy >= 0 && yy < W) {
                    float in[W];
#pragma unroll
                    for (int j = 0; j < W; ++j) in[j] = pb[(yy * W + j) * CG];
                    if (PROBE == 1) {
#pragma unroll
                        for (int j = 0; j < W; ++j) out[j] += in[j] * wt[24];
                    } else {
                        row_taps<W, false>(in, wt, ky, out);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PROBE == 2) {   // keep the sums alive, store one value per row
                float acc_ = 0.f;
#pragma unroll
                for (int o = 0; o < W; ++o) acc_ += out[o];
                if (acc_ == 123.456f) yb[(int64_t)py * W * C] = acc_;
                continue;
            }
            float *yp = yb + (int64_t)py * W * C;
            // W store instructions per round in EVERY wave — the count the vmcnt below relies on.  A slot without a
            // row of its own (W % SLOTS != 0) recomputes the last row and stores the same values to the same
            // addresses as that row's owner: no predicate, no branch around a store.
            (void)rv;
#pragma unroll
            for (int o = 0; o < W; ++o) {
                *yp = out[o];
                yp += C;
            }
        }
        if (more) {
            // all but the youngest ROUNDS * W vector-memory operations (this item's stores) are done: the DMA of
            // the next plane, issued before them, has landed
            static_assert(ROUNDS * W <= 63, "vmcnt field");
            if (PROBE == 2)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROUNDS * W) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        cur ^= 1;
    }
}

template <int W, int SLOTS, int PROBE = 0>
int launch_pipe_fwd(const float *x, const float *w, const float *bias, float *y, int B, int C, hipStream_t stream) {
    const int ncg = C / CG;
    const size_t lds = (size_t)2 * PipeGeo<W>::PLANE * sizeof(float);
    const int per_cu = (int)((160 * 1024) / lds) < (2048 / (32 * SLOTS)) ? (int)((160 * 1024) / lds) : (2048 / (32 * SLOTS));   // LDS- or thread-limited
    int wgs = 256 * (per_cu < 1 ? 1 : per_cu);
    wgs -= wgs % ncg;
    if (wgs > B * ncg) wgs = B * ncg;
    static bool configured = false;
    if (!configured && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)dwconv_pipe_fwd_kernel<W, SLOTS, PROBE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    hipLaunchKernelGGL((dwconv_pipe_fwd_kernel<W, SLOTS, PROBE>), dim3(wgs), dim3(32 * SLOTS), lds, stream, x, w, bias, y, B, C, ncg);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// Backward, pipelined the same way: (dy, x) plane pairs double-buffered (4 buffers = 119 KB at 15 x 15: one
// workgroup of 512 threads per CU).  Two thread maps per item, so that neither needs many registers:
//   phase A  thread = (channel, output row): dx row = correlation of dy with the flipped taps (49 taps in registers);
//            the centre row it reads is dy[row] itself -> its share of dbias
//   phase B  thread = (channel, tap row ky, half of the rows): dw[ky][0..6] += sum over its rows of dy[row] x[row+ky-3]
//            — SEVEN running sums per thread instead of the 49 a (channel, row) thread would carry over all items
//            (that form needed 256 registers and spilled)
// The sums stay in registers over all items of the workgroup (fixed channel slice) and meet in LDS once at the end.
template <int W, int SLOTS, bool RES>
__global__ __launch_bounds__(32 * SLOTS, (32 * SLOTS) / 256) void dwconv_pipe_bwd_kernel(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ dx,
    float *__restrict__ dw, float *__restrict__ dbias, int B, int C, int ncg, const float *__restrict__ dres) {
    extern __shared__ __attribute__((aligned(16))) float plane[];
    static_assert(SLOTS >= W && (SLOTS == 8 || SLOTS == 16), "one output row per thread; 8 tap-row slots per half");
    constexpr int HW = W * W, PL = PipeGeo<W>::PLANE, NWV = SLOTS / 2;
    constexpr int HALVES = SLOTS / 8, RH = (W + HALVES - 1) / HALVES;   // rows per half in phase B
    const int c = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cg = blockIdx.x % ncg, b0 = blockIdx.x / ncg, bstep = gridDim.x / ncg;
    const int cglob = cg * CG + c;
    const bool rv = ps < W;                  // phase A: the slot owns a row (else it repeats the last row's dx)
    const int py = rv ? ps : W - 1;
    const int kyb = ps & 7, hb = ps >> 3;    // phase B: tap row, half
    const bool bv = kyb < 7;
    const int r_begin = hb * RH, r_end = (r_begin + RH) < W ? (r_begin + RH) : W;
    float wt[49], dwacc[7];
    float dbacc = 0.f;
    {
        const float *wp = w + cglob;
#pragma unroll
        for (int k = 0; k < 49; ++k) {
            wt[k] = *wp;
            wp += C;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) dwacc[k] = 0.f;
    }
    if (b0 < B) {
        dma_plane<W>(plane, dy + (int64_t)b0 * HW * C + cg * CG, C, wv, NWV);
        dma_plane<W>(plane + PL, x + (int64_t)b0 * HW * C + cg * CG, C, wv, NWV);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int b = b0; b < B; b += bstep) {
        const bool more = b + bstep < B;
        // the shortcut's gradient of this row (ConvNeXt block: d x = depthwise backward + d shortcut): loaded now,
        // behind the next planes' DMA and ahead of the row's arithmetic, added when the row is stored.  The counted
        // wait below still leaves exactly this item's W stores in flight.
        // RES is a template parameter on purpose: with a run-time `dres ? load : 0` the two arms met in register copies
        // right here, the copies waited for these loads - and for everything older on the in-order counter, i.e. the
        // next planes' DMA issued just above: the double buffering did not overlap anything (60 vs 45 us per launch).
        // The loads are issued from inline assembly ON PURPOSE as well: with compiler-visible loads in flight hipcc puts
        // `s_waitcnt vmcnt(0)` in front of the first use of every LDS read of phase A (an LDS read may alias a pending
        // LDS-DMA, and with ordinary loads pending too it no longer counts) - the same stall.  Nothing touches rres
        // between here and the explicit wait in front of the stores.
        float rres[W];
        if constexpr (RES) {
            const float *rp = dres + (int64_t)b * HW * C + cglob + (int64_t)py * W * C;
#pragma unroll
            for (int o = 0; o < W; ++o) {
                const float *q = rp + (int64_t)o * C;
                asm volatile("global_load_dword %0, %1, off" : "=&v"(rres[o]) : "v"(q) : "memory");
            }
        }
        if (more) {
            float *nb = plane + (cur ^ 1) * 2 * PL;
            dma_plane<W>(nb, dy + (int64_t)(b + bstep) * HW * C + cg * CG, C, wv, NWV);
            dma_plane<W>(nb + PL, x + (int64_t)(b + bstep) * HW * C + cg * CG, C, wv, NWV);
        }
        const float *pdy = plane + cur * 2 * PL + c, *px = pdy + PL;
        // ---- phase A: dx row
        {
            float out[W];
#pragma unroll
            for (int o = 0; o < W; ++o) out[o] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int yy = py + ky - 3;
                if (yy >= 0 && yy < W) {
                    float in[W];
#pragma unroll
                    for (int j = 0; j < W; ++j) in[j] = pdy[(yy * W + j) * CG];
                    if (ky == 3 && rv) {
#pragma unroll
                        for (int j = 0; j < W; ++j) dbacc += in[j];
                    }
                    row_taps<W, true>(in, wt, ky, out);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float *dxp = dx + (int64_t)b * HW * C + cglob + (int64_t)py * W * C;
            if constexpr (RES) {
                // rres (and, older on the in-order counter, nothing else of this item; younger: the next planes' DMA,
                // which has had all of phase A to land)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(rres[0]), "+v"(rres[W - 1])::"memory");
#pragma unroll
                for (int o = 1; o < W - 1; ++o) asm volatile("" : "+v"(rres[o]));
            }
#pragma unroll
            for (int o = 0; o < W; ++o) {   // W stores in every wave (a slot without a row repeats the last row)
                if constexpr (RES)
                    *dxp = out[o] + rres[o];
                else
                    *dxp = out[o];
                dxp += C;
            }
        }
        // ---- phase B: dw[ky][:] over this half's rows
        if (bv) {
#pragma unroll 1
            for (int r = r_begin; r < r_end; ++r) {
                const int yy = r + kyb - 3;
                if (yy < 0 || yy >= W) continue;
                float d[W], in[W];
#pragma unroll
                for (int j = 0; j < W; ++j) d[j] = pdy[(r * W + j) * CG];
#pragma unroll
                for (int j = 0; j < W; ++j) in[j] = px[(yy * W + j) * CG];
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    float a = dwacc[kx];
#pragma unroll
                    for (int o = 0; o < W; ++o)
                        if (o + kx - 3 >= 0 && o + kx - 3 < W) a = fmaf(d[o], in[o + kx - 3], a);
                    dwacc[kx] = a;
                }
            }
        }
        if (more) {
            static_assert(W <= 63, "vmcnt field");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        cur ^= 1;
    }
    // the halves' tap-row sums and the row slots' dbias shares meet in LDS (the plane buffers are idle now)
    __syncthreads();
    float *red = plane;   // [SLOTS][8][32]: 7 tap sums (phase B) + the slot's dbias share (phase A)
#pragma unroll
    for (int k = 0; k < 7; ++k) red[(ps * 8 + k) * CG + c] = bv ? dwacc[k] : 0.f;
    red[(ps * 8 + 7) * CG + c] = dbacc;
    __syncthreads();
    if (ps < 7) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            float sum = 0.f;
#pragma unroll
            for (int h = 0; h < HALVES; ++h) sum += red[((h * 8 + ps) * 8 + kx) * CG + c];
            atomicAdd(&dw[(ps * 7 + kx) * C + cglob], sum);
        }
    } else if (ps == 7 && dbias) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) sum += red[(q * 8 + 7) * CG + c];
        atomicAdd(&dbias[cglob], sum);
    }
}

template <int W, int SLOTS>
int launch_pipe_bwd(const float *dy, const float *x, const float *w, float *dx, float *dw, float *dbias, int B, int C,
                    hipStream_t stream, const float *dres) {
    const int ncg = C / CG;
    const size_t planes = (size_t)4 * PipeGeo<W>::PLANE * sizeof(float), red = (size_t)SLOTS * 8 * CG * sizeof(float);
    const size_t lds = planes > red ? planes : red;
    if (lds > 160 * 1024) return AC_EINVAL;
    int per_cu = (int)((160 * 1024) / lds);
    const int by_threads = 2048 / (32 * SLOTS), by_regs = 512 / (32 * SLOTS);    // 2 waves per SIMD (the kernels take ~200-256 registers)
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu > by_regs) per_cu = by_regs;
    if (per_cu < 1) per_cu = 1;
    int wgs = 256 * per_cu;
    wgs -= wgs % ncg;
    if (wgs > B * ncg) wgs = B * ncg;
    static bool configured = false;
    if (!configured && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)dwconv_pipe_bwd_kernel<W, SLOTS, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)dwconv_pipe_bwd_kernel<W, SLOTS, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(int)e - 2000;
        configured = true;
    }
    if (dres)
        hipLaunchKernelGGL((dwconv_pipe_bwd_kernel<W, SLOTS, true>), dim3(wgs), dim3(32 * SLOTS), lds, stream, dy, x, w, dx,
                           dw, dbias, B, C, ncg, dres);
    else
        hipLaunchKernelGGL((dwconv_pipe_bwd_kernel<W, SLOTS, false>), dim3(wgs), dim3(32 * SLOTS), lds, stream, dy, x, w, dx,
                           dw, dbias, B, C, ncg, dres);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

template <int W>
int launch_rows_fwd(const float *x, const float *w, const float *bias, float *y, int B, int C, hipStream_t stream) {
    const size_t lds = (size_t)W * W * CG * sizeof(float);
    hipLaunchKernelGGL(dwconv_rows_fwd_kernel<W>, dim3(B, (C + CG - 1) / CG), dim3(256), lds, stream, x, w, bias, y, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
template <int W>
int launch_rows_bwd(const float *dy, const float *x, const float *w, float *dx, float *dw, float *dbias, int B,
                    int C, hipStream_t stream, const float *dres) {
    const size_t planes = (size_t)2 * W * W * CG * sizeof(float), red = (size_t)8 * 50 * CG * sizeof(float);
    const size_t lds = planes > red ? planes : red;
    hipLaunchKernelGGL(dwconv_rows_bwd_kernel<W>, dim3((B + SPB - 1) / SPB, (C + CG - 1) / CG), dim3(256), lds,
                       stream, dy, x, w, dx, dw, dbias, B, C, dres);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// Small planes (S x S, S = 1 or 3: the last two ConvNeXt stages at 63x63 input).  Every output
// pixel sees every input pixel (|offset| <= 2 < 4), so the convolution is a dense S^2 x S^2 product
// per channel over the (2S-1)^2 central taps: thread = channel, all pixels of a sample in
// registers, samples looped per thread, no LDS plane and no barriers in the loop.  The general
// kernel spent ~100 us per launch here on its per-sample load -> barrier -> compute chain.
// ---------------------------------------------------------------------------------------------
constexpr int SM_SLOTS = 4;  // sample slots per workgroup (256 threads = 64 channels x 4 slots)

template <int S>
__global__ __launch_bounds__(256) void dwconv_small_fwd_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ bias,
                                                               float *__restrict__ y, int B, int C,
                                                               int spb) {
    constexpr int P = S * S, T = 2 * S - 1;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slot = threadIdx.x >> 6;
    if (c >= C) return;
    float wt[T * T];
#pragma unroll
    for (int ty = 0; ty < T; ++ty)
#pragma unroll
        for (int tx = 0; tx < T; ++tx) wt[ty * T + tx] = w[((ty + 4 - S) * 7 + (tx + 4 - S)) * C + c];
    const float bv = bias ? bias[c] : 0.f;
    const int b1 = min(B, (int)(blockIdx.y + 1) * spb);
    for (int b = blockIdx.y * spb + slot; b < b1; b += SM_SLOTS) {
        float xi[P];
#pragma unroll
        for (int i = 0; i < P; ++i) xi[i] = x[((int64_t)b * P + i) * C + c];
#pragma unroll
        for (int o = 0; o < P; ++o) {
            float a = bv;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int ty = i / S - o / S + S - 1, tx = i % S - o % S + S - 1;
                a = fmaf(xi[i], wt[ty * T + tx], a);
            }
            y[((int64_t)b * P + o) * C + c] = a;
        }
    }
}

template <int S>
__global__ __launch_bounds__(256) void dwconv_small_bwd_kernel(const float *__restrict__ dy,
                                                               const float *__restrict__ x,
                                                               const float *__restrict__ w,
                                                               float *__restrict__ dx,
                                                               float *__restrict__ dw,
                                                               float *__restrict__ dbias, int B,
                                                               int C, int spb, const float *__restrict__ dres) {
    constexpr int P = S * S, T = 2 * S - 1;
    __shared__ float red[SM_SLOTS][T * T + 1][64];
    const int cl = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cv = c < C;
    float wt[T * T], dwa[T * T];
    float dba = 0.f;
#pragma unroll
    for (int ty = 0; ty < T; ++ty)
#pragma unroll
        for (int tx = 0; tx < T; ++tx) {
            wt[ty * T + tx] = w[((ty + 4 - S) * 7 + (tx + 4 - S)) * C + (cv ? c : 0)];   // unconditional (clamped) load
            dwa[ty * T + tx] = 0.f;
        }
    const int b1 = min(B, (int)(blockIdx.y + 1) * spb);
    if (cv) {
        for (int b = blockIdx.y * spb + slot; b < b1; b += SM_SLOTS) {
            // the shortcut's gradient is loaded with the planes (at the store, each of the P stores waited for its own
            // load: a load round trip per output pixel)
            float xi[P], d[P], rr[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                xi[i] = x[((int64_t)b * P + i) * C + c];
                d[i] = dy[((int64_t)b * P + i) * C + c];
                rr[i] = dres ? dres[((int64_t)b * P + i) * C + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < P; ++i) dba += d[i];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float a = 0.f;
#pragma unroll
                for (int o = 0; o < P; ++o) {
                    const int ty = i / S - o / S + S - 1, tx = i % S - o % S + S - 1;
                    a = fmaf(d[o], wt[ty * T + tx], a);
                    dwa[ty * T + tx] = fmaf(d[o], xi[i], dwa[ty * T + tx]);
                }
                dx[((int64_t)b * P + i) * C + c] = a + rr[i];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < T * T; ++k) red[slot][k][cl] = dwa[k];
    red[slot][T * T][cl] = dba;
    __syncthreads();
    if (cv) {
        for (int k = slot; k <= T * T; k += SM_SLOTS) {
            const float sum = (red[0][k][cl] + red[1][k][cl]) + (red[2][k][cl] + red[3][k][cl]);
            if (k < T * T) {
                const int ty = k / T, tx = k % T;
                atomicAdd(&dw[((ty + 4 - S) * 7 + (tx + 4 - S)) * C + c], sum);
            } else if (dbias) {
                atomicAdd(&dbias[c], sum);
            }
        }
    }
}

inline int small_spb(int B) {  // samples per workgroup: ~32 workgroups along the batch
    int spb = (B + 31) / 32;
    return spb < SM_SLOTS ? SM_SLOTS : spb;
}

template <int XS, int WC = 0>
int launch_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W,
               int C, hipStream_t stream) {
    const Geo g = make_geo<XS, WC>(H, W, C);
    const size_t lds = (size_t)g.np * CG * sizeof(float);
    if (lds > 65536) return AC_EINVAL;  // up to ~16x16 planes (ConvNeXt stage 0 is 15x15)
    const dim3 grid(B, (C + CG - 1) / CG);
    if (WC > 0 && C % 4 == 0 && ac_aligned16(x))
        hipLaunchKernelGGL((dwconv7x7_fwd_kernel<XS, WC, true>), grid, dim3(256), lds, stream, x, w, bias, y, H, W, C);
    else
        hipLaunchKernelGGL((dwconv7x7_fwd_kernel<XS, WC, false>), grid, dim3(256), lds, stream, x, w, bias, y, H, W, C);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
template <int XS, int WC = 0>
int launch_bwd(const float *dy, const float *x, const float *w, float *dx, float *dw, float *dbias,
               int B, int H, int W, int C, hipStream_t stream, const float *dres) {
    const Geo g = make_geo<XS, WC>(H, W, C);
    const size_t planes = (size_t)g.np * CG * sizeof(float);
    const size_t red = (size_t)8 * 50 * CG * sizeof(float);
    const size_t lds = planes > red ? planes : red;
    if (lds > 65536) return AC_EINVAL;
    const dim3 grid((B + SPB - 1) / SPB, (C + CG - 1) / CG);
    if (WC > 0 && C % 4 == 0 && ac_aligned16(x) && ac_aligned16(dy))
        hipLaunchKernelGGL((dwconv7x7_bwd_kernel<XS, WC, true>), grid, dim3(256), lds, stream, dy, x, w, dx, dw,
                           dbias, B, H, W, C, dres);
    else
        hipLaunchKernelGGL((dwconv7x7_bwd_kernel<XS, WC, false>), grid, dim3(256), lds, stream, dy, x, w, dx, dw,
                           dbias, B, H, W, C, dres);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

}  // namespace

extern "C" int ac_dwconv7x7_fwd_v(const float *x, const float *w, const float *bias, float *y,
                                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t variant, ac_stream_t stream_) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return AC_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    // pipelined persistent kernels (variant 0 = automatic, 1 = the one-item-per-workgroup kernels below)
    if ((variant == 8 || variant == 9) && H == 15 && W == 15 && C % CG == 0 && ac_aligned16(x) && B * (C / CG) >= 256)
        return variant == 8 ? launch_pipe_fwd<15, 8, 1>(x, w, bias, y, B, C, stream)
                            : launch_pipe_fwd<15, 8, 2>(x, w, bias, y, B, C, stream);
    if (variant != 1 && H == W && (W == 15 || W == 7) && C % CG == 0 && ac_aligned16(x) && B * (C / CG) >= 256)
        return W == 15 ? (variant == 7 ? launch_pipe_fwd<15, 8>(x, w, bias, y, B, C, stream)
                                       : launch_pipe_fwd<15, 16>(x, w, bias, y, B, C, stream))
                       : launch_pipe_fwd<7, 8>(x, w, bias, y, B, C, stream);
    if (H == W && (W == 1 || W == 3)) {
        const int spb = small_spb(B);
        dim3 grid((C + 63) / 64, (B + spb - 1) / spb);
        if (W == 1)
            hipLaunchKernelGGL(dwconv_small_fwd_kernel<1>, grid, dim3(256), 0, stream, x, w, bias, y, B, C, spb);
        else
            hipLaunchKernelGGL(dwconv_small_fwd_kernel<3>, grid, dim3(256), 0, stream, x, w, bias, y, B, C, spb);
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    if (W == 1) return launch_fwd<1>(x, w, bias, y, B, H, W, C, stream);
    if (W <= 3) return launch_fwd<3>(x, w, bias, y, B, H, W, C, stream);
    if (H == W && (W == 15 || W == 7) && C % 4 == 0 && ac_aligned16(x))
        return W == 15 ? launch_rows_fwd<15>(x, w, bias, y, B, C, stream) : launch_rows_fwd<7>(x, w, bias, y, B, C, stream);
    if (H == 15 && W == 15) return launch_fwd<5, 15>(x, w, bias, y, B, H, W, C, stream);
    if (H == 7 && W == 7) return launch_fwd<7, 7>(x, w, bias, y, B, H, W, C, stream);
    if (W == 7) return launch_fwd<7>(x, w, bias, y, B, H, W, C, stream);
    return launch_fwd<5>(x, w, bias, y, B, H, W, C, stream);
}

extern "C" int ac_dwconv7x7_fwd(const float *x, const float *w, const float *bias, float *y,
                                int32_t B, int32_t H, int32_t W, int32_t C, ac_stream_t stream_) {
    return ac_dwconv7x7_fwd_v(x, w, bias, y, B, H, W, C, 0, stream_);
}

extern "C" int ac_dwconv7x7_bwd(const float *dy, const float *x, const float *w, float *dx,
                                float *dw, float *dbias, int32_t B, int32_t H, int32_t W,
                                int32_t C, ac_stream_t stream_) {
    return ac_dwconv7x7_bwd_res(dy, x, w, nullptr, dx, dw, dbias, B, H, W, C, 0, stream_);
}

extern "C" int ac_dwconv7x7_bwd_v(const float *dy, const float *x, const float *w, float *dx,
                                  float *dw, float *dbias, int32_t B, int32_t H, int32_t W,
                                  int32_t C, int32_t variant, ac_stream_t stream_) {
    return ac_dwconv7x7_bwd_res(dy, x, w, nullptr, dx, dw, dbias, B, H, W, C, variant, stream_);
}

extern "C" int ac_dwconv7x7_bwd_res(const float *dy, const float *x, const float *w, const float *dres, float *dx,
                                    float *dw, float *dbias, int32_t B, int32_t H, int32_t W,
                                    int32_t C, int32_t variant, ac_stream_t stream_) {
    if (!dy || !x || !w || !dx || !dw || B <= 0 || H <= 0 || W <= 0 || C <= 0) return AC_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    if (variant != 1 && H == W && (W == 15 || W == 7) && C % CG == 0 && ac_aligned16(x) && ac_aligned16(dy) &&
        B * (C / CG) >= 256)
        return W == 15 ? launch_pipe_bwd<15, 16>(dy, x, w, dx, dw, dbias, B, C, stream, dres)
                       : launch_pipe_bwd<7, 8>(dy, x, w, dx, dw, dbias, B, C, stream, dres);
    if (H == W && (W == 1 || W == 3)) {
        const int spb = small_spb(B);
        dim3 grid((C + 63) / 64, (B + spb - 1) / spb);
        if (W == 1)
            hipLaunchKernelGGL(dwconv_small_bwd_kernel<1>, grid, dim3(256), 0, stream, dy, x, w, dx, dw, dbias, B, C, spb, dres);
        else
            hipLaunchKernelGGL(dwconv_small_bwd_kernel<3>, grid, dim3(256), 0, stream, dy, x, w, dx, dw, dbias, B, C, spb, dres);
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    if (W == 1) return launch_bwd<1>(dy, x, w, dx, dw, dbias, B, H, W, C, stream, dres);
    if (W <= 3) return launch_bwd<3>(dy, x, w, dx, dw, dbias, B, H, W, C, stream, dres);
    if (H == W && (W == 15 || W == 7) && C % 4 == 0 && ac_aligned16(x) && ac_aligned16(dy))
        return W == 15 ? launch_rows_bwd<15>(dy, x, w, dx, dw, dbias, B, C, stream, dres)
                       : launch_rows_bwd<7>(dy, x, w, dx, dw, dbias, B, C, stream, dres);
    if (H == 15 && W == 15) return launch_bwd<5, 15>(dy, x, w, dx, dw, dbias, B, H, W, C, stream, dres);
    if (H == 7 && W == 7) return launch_bwd<7, 7>(dy, x, w, dx, dw, dbias, B, H, W, C, stream, dres);
    if (W == 7) return launch_bwd<7>(dy, x, w, dx, dw, dbias, B, H, W, C, stream, dres);
    return launch_bwd<5>(dy, x, w, dx, dw, dbias, B, H, W, C, stream, dres);
}
