// Gather-GEMM for gfx950 (see include/applecider_hip.h, "Gather-GEMM").
//
// Tile: 128x128x32 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave as
// 2x2 MFMA 32x32 tiles -> 64 accumulator registers).  Operands are staged
// global -> registers -> LDS (double buffered, one barrier per K-tile); the
// global loads of tile k+1 are in flight while tile k feeds the matrix cores.
//
// Operand images in LDS:
//   "KC" (inner index = reduction k, e.g. activations [M,K], weights [N,K]):
//       [128 rows][32 k] fp32, 16-byte chunks XOR-swizzled by ((row>>1)&7) so the
//       ds_read_b128 fragment reads are bank-conflict free (MI355X_MICROARCH §LDS).
//       One b128 read gives 4 k values; they feed 4 consecutive MFMAs, i.e. the
//       k order inside a K-tile is permuted identically for A and B.
//   "RC" (outer index = reduction k, inner = m or n; used by NN's B, TN's A and B):
//       [32 k][128 cols] fp32, read with conflict-free ds_read_b32.
//
// math = AC_MATH_F32 : v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
// math = AC_MATH_BF16: operands rounded to bf16 when written to LDS,
//                      v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
#include "ac_common.h"
#include <hip/hip_bf16.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_FLOATS = 128 * 32;  // both operand images are 16 KB

struct GemmParams {
    ac_gemm_desc d;
    int tiles_m, tiles_n, nkt, kt_per_split;
    int vec_epi;  // 1: 16-byte epilogue (all C-side pointers/strides 16-byte aligned, N % 4 == 0)
    int epi_var;  // compile-time epilogue variant (AC_EPI_VARIANTS index) or EPI_GENERIC
    // ac_gemm_batched: gridDim.x = batch * tiles; product z reads / writes its operands bs_* elements further on
    int64_t bs_a, bs_b, bs_c;
    // ac_gemm_grouped: product z takes its three base pointers from here instead (ngrp > 0): independent products of one
    // shape whose operands are separate allocations (the weight gradients of a ConvNeXt stage's blocks, of the encoder
    // layers).  By value in the kernel arguments: nothing to upload, and a captured hipGraph replays them.
    static constexpr int MAX_GROUPS = 16;
    const void *grp[3 * MAX_GROUPS];
    int ngrp;
};

// Workgroup -> (output tile, K piece).  One K piece: XCD-aware tile order (workgroups that share an XCD, bid % 8, walk
// contiguous tiles).  Split-K with a multiple of 8 pieces: the PIECES are dealt to the XCDs and an XCD walks the tiles
// of one piece after another — all tiles of a piece read the same K rows of A and B, so each XCD's L2 fetches them
// once (the 2-D grid spread the tiles of a piece over all eight L2s: the weight-gradient family moved 2.06x its
// algorithmic bytes over the fabric, profiles/r03_pmc_hbm_traffic_bf16x3.json).
__device__ __forceinline__ void map_workgroup(int &wg, int &piece) {
    const int nwg = gridDim.x, pieces = gridDim.y;
    if (pieces > 1 && (pieces & 7) == 0) {
        const int lin = blockIdx.x + nwg * blockIdx.y;
        const int slot = lin >> 3, round = slot / nwg;
        piece = (lin & 7) + 8 * round;
        wg = slot - round * nwg;
        return;
    }
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    piece = blockIdx.y;
}

// Batched instantiations: workgroup id -> (product z, tile), operand pointers of product z.  The XCD-aware order
// is taken over the flattened (z, tile) range, so the tiles of one product (which share its A and B) run on one XCD.
__device__ __forceinline__ void batch_select(const GemmParams &p, int &wg, ac_gemm_desc &d) {
    const int per = p.tiles_m * p.tiles_n, z = wg / per;
    wg -= z * per;
    d = p.d;
    if (p.ngrp > 0) {
        d.a.ptr = p.grp[3 * z];
        d.b.ptr = p.grp[3 * z + 1];
        d.c.ptr = const_cast<void *>(p.grp[3 * z + 2]);
        return;
    }
    d.a.ptr = (const float *)d.a.ptr + (int64_t)z * p.bs_a;
    d.b.ptr = (const float *)d.b.ptr + (int64_t)z * p.bs_b;
    d.c.ptr = (float *)d.c.ptr + (int64_t)z * p.bs_c;
}

// round-to-nearest-even fp32 -> bf16 (the rounding ac_cast_bf16 applies)
__device__ __forceinline__ unsigned short epi_bf16(float x) {
    return ac_f2h(x);
}
__device__ __forceinline__ float epi_bf16_to_f32(unsigned short h) {
    return ac_h2f(h);
}

__device__ __forceinline__ int64_t inner_off(const int32_t *goff, int i) {
    return goff ? (int64_t)goff[i >> 5] + (i & 31) : (int64_t)i;
}

// GELU means the rational-erf form in every math mode but the exact-fp32 one (the compile-time variants: E_FAST)
__device__ __forceinline__ int eff_act(const ac_gemm_desc &d, int code) {
    return (code == AC_ACT_GELU && d.math != AC_MATH_F32) ? AC_ACT_GELU_FAST : code;
}

__device__ __forceinline__ void epilogue_store(const ac_gemm_desc &d, uint64_t dseed, int m, int n, float acc,
                                               int64_t caddr) {
    float v = acc * d.alpha;
    if (d.bias) v += d.bias[n];
    if (d.pre_out) d.pre_out[(int64_t)m * d.ld_pre + n] = v;
    v = ac_act(v, eff_act(d, d.act));
    if (d.dact) v *= ac_dact(d.aux[(int64_t)m * d.ld_aux + n], eff_act(d, d.dact));
    if (d.mask16) v = epi_bf16_to_f32(((const unsigned short *)d.mask16)[(int64_t)m * d.ld_mask16 + n]) > 0.f ? v : 0.f;
    if (d.colscale) v *= d.colscale[n];
    if (d.drop_p > 0.f)
        v = ac_rand01(dseed, (uint64_t)m * (uint64_t)d.N + (uint64_t)n) >= d.drop_p
                ? v * (1.0f / (1.0f - d.drop_p)) : 0.f;
    if (d.residual) v += d.residual[(int64_t)m * d.ld_res + n];
    if (d.c16) ((unsigned short *)d.c16)[(int64_t)m * d.ld_c16 + inner_off(d.c.goff, n)] = epi_bf16(v);
    if (!d.c.ptr) return;
    float *c = (float *)d.c.ptr + caddr;
    if (d.accumulate == 2)
        atomicAdd(c, v);
    else if (d.accumulate == 1)
        *c += v;
    else
        *c = v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// epilogue: reg e of a 32x32 tile holds row (e&3)+8*(e>>2)+4*lh, col li.
// Compile-time register indices keep the accumulators out of scratch.
__device__ __forceinline__ void store_tile(const ac_gemm_desc &d, uint64_t dseed, const f32x16 (&acc)[2][2],
                                           int row_base, int col_base, int li, int lh) {
    const int n0 = col_base + li, n1 = n0 + 32;
    const int64_t c0 = inner_off(d.c.goff, n0 < d.N ? n0 : 0);
    const int64_t c1 = inner_off(d.c.goff, n1 < d.N ? n1 : 0);
    static_for<0, 32>([&](auto idx) {
        constexpr int sa = decltype(idx)::value / 16, e = decltype(idx)::value % 16;
        const int m = row_base + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < d.M) {
            const int64_t crow = ac_rowaddr(d.c.rows, m);
            if (n0 < d.N) epilogue_store(d, dseed, m, n0, acc[sa][0][e], crow + c0);
            if (n1 < d.N) epilogue_store(d, dseed, m, n1, acc[sa][1][e], crow + c1);
        }
    });
}

// Split-K epilogue: straight-line no-return float atomics (plain row-major C, alpha only).  The
// general per-element epilogue costs ~25 us per 128x128 tile (branches around every element), more
// than the K loop of a skinny weight-gradient product; this form leaves its 64 atomics per lane in
// flight.  Each wave-instruction adds two 128-byte row segments (the full-rate shape).
__device__ __forceinline__ void store_tile_atomic(const ac_gemm_desc &d, const f32x16 (&acc)[2][2],
                                                  int row_base, int col_base, int li, int lh) {
    const int n0 = col_base + li, n1 = n0 + 32;
    const bool v0 = n0 < d.N, v1 = n1 < d.N;
    float *c = (float *)d.c.ptr + n0;
    const int64_t ldc = d.c.rows.s3;
    const float alpha = d.alpha;
    static_for<0, 32>([&](auto idx) {
        constexpr int sa = decltype(idx)::value / 16, e = decltype(idx)::value % 16;
        const int m = row_base + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < d.M) {
            float *cr = c + (int64_t)m * ldc;
            if (v0) atomicAdd(cr, acc[sa][0][e] * alpha);
            if (v1) atomicAdd(cr + 32, acc[sa][1][e] * alpha);
        }
    });
}

// Deterministic split-K (accumulate = 3): K piece s stores its partial tile plainly into slab s of a [split_k, M, N]
// buffer; a reduce pass (ac_splitk_reduce) sums the slabs in a fixed order and applies the epilogue.  Forward
// products use this instead of atomics: two runs of the same forward stay bit-identical.
__device__ __forceinline__ void store_tile_slab(const ac_gemm_desc &d, const f32x16 (&acc)[2][2], int split,
                                                int row_base, int col_base, int li, int lh) {
    const int n0 = col_base + li, n1 = n0 + 32;
    const bool v0 = n0 < d.N, v1 = n1 < d.N;
    float *c = (float *)d.c.ptr + (int64_t)split * d.M * d.N + n0;
    const float alpha = d.alpha;
    static_for<0, 32>([&](auto idx) {
        constexpr int sa = decltype(idx)::value / 16, e = decltype(idx)::value % 16;
        const int m = row_base + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < d.M) {
            float *cr = c + (int64_t)m * d.N;
            if (v0) cr[0] = acc[sa][0][e] * alpha;
            if (v1) cr[32] = acc[sa][1][e] * alpha;
        }
    });
}

// 16-byte epilogue: the wave parks each 32x64 half of its accumulator tile in 8 KB of (now idle)
// LDS and re-reads it row-major, so every lane handles 4 consecutive columns of one row: bias /
// aux / residual come in as float4 and C goes out as float4 — 4x fewer memory instructions than the
// one-float-per-lane accumulator layout (the memory-bound small-K products were store-issue bound).
__device__ __forceinline__ f32x4 epilogue_vec(const ac_gemm_desc &d, uint64_t dseed, int m, int n, f32x4 v,
                                              int64_t caddr) {
    v *= d.alpha;
    if (d.bias) v += *(const f32x4 *)(d.bias + n);
    if (d.pre_out) *(f32x4 *)(d.pre_out + (int64_t)m * d.ld_pre + n) = v;
    if (d.act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ac_act(v[j], eff_act(d, d.act));
    }
    if (d.dact) {
        const f32x4 a = *(const f32x4 *)(d.aux + (int64_t)m * d.ld_aux + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= ac_dact(a[j], eff_act(d, d.dact));
    }
    if (d.mask16) {
        const ushort4 k = *(const ushort4 *)((const unsigned short *)d.mask16 + (int64_t)m * d.ld_mask16 + n);
        v[0] = epi_bf16_to_f32(k.x) > 0.f ? v[0] : 0.f;
        v[1] = epi_bf16_to_f32(k.y) > 0.f ? v[1] : 0.f;
        v[2] = epi_bf16_to_f32(k.z) > 0.f ? v[2] : 0.f;
        v[3] = epi_bf16_to_f32(k.w) > 0.f ? v[3] : 0.f;
    }
    if (d.colscale) v *= *(const f32x4 *)(d.colscale + n);
    if (d.drop_p > 0.f) {
        const float inv_keep = 1.0f / (1.0f - d.drop_p);
        const uint64_t i0 = (uint64_t)m * (uint64_t)d.N + (uint64_t)n;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ac_rand01(dseed, i0 + j) >= d.drop_p ? v[j] * inv_keep : 0.f;
    }
    if (d.residual) v += *(const f32x4 *)(d.residual + (int64_t)m * d.ld_res + n);
    if (d.c16) {
        ushort4 h;
        h.x = epi_bf16(v[0]); h.y = epi_bf16(v[1]); h.z = epi_bf16(v[2]); h.w = epi_bf16(v[3]);
        *(ushort4 *)((unsigned short *)d.c16 + (int64_t)m * d.ld_c16 + inner_off(d.c.goff, n)) = h;
    }
    if (!d.c.ptr) return v;
    f32x4 *c = (f32x4 *)((float *)d.c.ptr + caddr);
    if (d.accumulate == 1)
        *c += v;
    else
        *c = v;
    return v;
}

// Compile-time epilogue variants.  The generic epilogue_vec tests a dozen descriptor fields per
// float4 and (worse) evaluates `ac_act(v, kind)` with a run-time kind inside unrolled loops, which
// hipcc turns into "compute GELU, sigmoid and tanh, then select": bias + ReLU cost +50 % on a
// 66048 x 512 x 128 product.  The host picks the variant whose feature mask equals the
// descriptor's; anything else (and gathered / scattered C) takes the generic path.
enum : unsigned {
    E_BIAS = 1u, E_PRE = 2u, E_GELU = 4u, E_RELU = 8u, E_DGELU = 16u, E_MASK16 = 32u, E_CSCALE = 64u,
    E_DROP = 128u, E_RES = 256u, E_C16 = 512u, E_C32 = 1024u, E_ACC = 2048u,
    E_FAST = 4096u,  // bf16 / split-bf16 math modes: rational erf inside GELU / GELU'
    E_GOFF = 8192u,  // C columns through the offset table (conv outputs scattered into the cat buffer)
    E_DRELU = 16384u // v *= (aux > 0): ReLU' of the layer whose hidden gradient this product forms
};
#define AC_EPI_VARIANTS(X)                                                                      \
    X(0, E_C32) X(1, E_C32 | E_BIAS) X(2, E_C16 | E_BIAS | E_GELU | E_PRE)                       \
    X(3, E_C16 | E_BIAS | E_RELU | E_DROP) X(4, E_C16 | E_DGELU) X(5, E_C16 | E_MASK16)          \
    X(6, E_C32 | E_BIAS | E_PRE | E_CSCALE | E_RES) X(7, E_C32 | E_BIAS | E_DROP | E_RES)        \
    X(8, E_C32 | E_ACC) X(9, E_C32 | E_BIAS | E_RES) X(10, E_C16 | E_BIAS | E_RELU)              \
    X(11, E_C16 | E_BIAS | E_GELU) X(12, E_C32 | E_BIAS | E_CSCALE | E_RES) X(13, E_C32 | E_BIAS | E_GELU) \
    X(14, E_C16 | E_BIAS | E_GELU | E_PRE | E_FAST) X(15, E_C16 | E_DGELU | E_FAST)               \
    X(16, E_C16 | E_BIAS | E_GELU | E_FAST) X(17, E_C32 | E_BIAS | E_GELU | E_FAST) X(18, E_C16)     \
    X(19, E_C32 | E_BIAS | E_GOFF) X(20, E_C32 | E_GOFF) X(21, E_C16 | E_BIAS | E_GOFF)                    \
    X(22, E_C32 | E_BIAS | E_GELU | E_PRE) X(23, E_C32 | E_BIAS | E_RELU)                                 \
    X(24, E_C32 | E_BIAS | E_RELU | E_PRE | E_RES) X(25, E_C32 | E_BIAS | E_GELU | E_PRE | E_RES)       \
    X(26, E_C32 | E_BIAS | E_GELU | E_PRE | E_FAST) X(27, E_C32 | E_BIAS | E_GELU | E_PRE | E_RES | E_FAST)   \
    X(28, E_C32 | E_DGELU | E_FAST) X(29, E_C32 | E_DGELU) X(30, E_C32 | E_DRELU) X(31, E_C32 | E_DRELU | E_DROP)
constexpr int EPI_GENERIC = 255;

template <unsigned F>
__device__ __forceinline__ f32x4 epilogue_vec_t(const ac_gemm_desc &d, uint64_t dseed, int m, int n, f32x4 v,
                                                const f32x4 &bias4, const f32x4 &cs4, int64_t ccol) {
    v *= d.alpha;
    if constexpr (F & E_BIAS) v += bias4;  // per-column vectors are loaded once per tile
    if constexpr (F & E_PRE) *(f32x4 *)(d.pre_out + (int64_t)m * d.ld_pre + n) = v;
    if constexpr (F & E_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (F & E_FAST) ? ac_gelu_fast(v[j]) : ac_gelu(v[j]);
    }
    if constexpr (F & E_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    if constexpr (F & E_DGELU) {
        const f32x4 a = *(const f32x4 *)(d.aux + (int64_t)m * d.ld_aux + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (F & E_FAST) ? ac_gelu_grad_fast(a[j]) : ac_gelu_grad(a[j]);
    }
    if constexpr (F & E_DRELU) {
        const f32x4 a = *(const f32x4 *)(d.aux + (int64_t)m * d.ld_aux + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a[j] > 0.f ? v[j] : 0.f;
    }
    if constexpr (F & E_MASK16) {
        const ushort4 k = *(const ushort4 *)((const unsigned short *)d.mask16 + (int64_t)m * d.ld_mask16 + n);
        v[0] = epi_bf16_to_f32(k.x) > 0.f ? v[0] : 0.f;
        v[1] = epi_bf16_to_f32(k.y) > 0.f ? v[1] : 0.f;
        v[2] = epi_bf16_to_f32(k.z) > 0.f ? v[2] : 0.f;
        v[3] = epi_bf16_to_f32(k.w) > 0.f ? v[3] : 0.f;
    }
    if constexpr (F & E_CSCALE) v *= cs4;
    if constexpr (F & E_DROP) {
        const float inv_keep = 1.0f / (1.0f - d.drop_p);
        const uint64_t i0 = (uint64_t)m * (uint64_t)d.N + (uint64_t)n;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ac_rand01(dseed, i0 + j) >= d.drop_p ? v[j] * inv_keep : 0.f;
    }
    if constexpr (F & E_RES) v += *(const f32x4 *)(d.residual + (int64_t)m * d.ld_res + n);
    if constexpr (F & E_C16) {
        ushort4 h;
        h.x = epi_bf16(v[0]); h.y = epi_bf16(v[1]); h.z = epi_bf16(v[2]); h.w = epi_bf16(v[3]);
        *(ushort4 *)((unsigned short *)d.c16 + (int64_t)m * d.ld_c16 + ((F & E_GOFF) ? ccol : (int64_t)n)) = h;
    }
    if constexpr (F & E_C32) {
        // plain row-major C; with E_GOFF the column offset comes from the table (looked up once per tile)
        f32x4 *c = (f32x4 *)((float *)d.c.ptr + (int64_t)m * d.c.rows.s3 + ((F & E_GOFF) ? ccol : (int64_t)n));
        if constexpr (F & E_ACC) *c += v; else *c = v;
    }
    return v;
}

template <int VAR>
__device__ __forceinline__ void store_tile_vec_t(const ac_gemm_desc &d, uint64_t dseed, const f32x16 (&acc)[2][2],
                                                 float *wbuf, int row_base, int col_base, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int rsub = lane >> 4, c4 = 4 * (lane & 15);
    const int n = col_base + c4;
    const int64_t coff = (d.c.goff != nullptr && n < d.N) ? inner_off(d.c.goff, n) : (int64_t)n;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, cs4 = {1.f, 1.f, 1.f, 1.f}, csum = {0.f, 0.f, 0.f, 0.f};
    if (VAR != EPI_GENERIC && n < d.N) {
        if (d.bias) bias4 = *(const f32x4 *)(d.bias + n);
        if (d.colscale) cs4 = *(const f32x4 *)(d.colscale + n);
    }
    // compile-time indices into acc throughout: a loop the compiler declines to unroll would index
    // the accumulators dynamically and push all 64 of them to scratch
    static_for<0, 2>([&](auto sidx) {
        constexpr int sa = decltype(sidx)::value;
        static_for<0, 16>([&](auto idx) {
            constexpr int e = decltype(idx)::value;
            const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
            wbuf[r * 64 + li] = acc[sa][0][e];
            wbuf[r * 64 + 32 + li] = acc[sa][1][e];
        });
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + rsub;
            const f32x4 v = *(const f32x4 *)(wbuf + r * 64 + c4);
            const int m = row_base + sa * 32 + r;
            if (m < d.M && n < d.N) {
                f32x4 o;
                if constexpr (VAR == EPI_GENERIC) {
                    o = epilogue_vec(d, dseed, m, n, v, ac_rowaddr(d.c.rows, m) + coff);
                } else {
#define AC_EPI_CALL(I, F) if constexpr (VAR == I) o = epilogue_vec_t<(F)>(d, dseed, m, n, v, bias4, cs4, coff);
                    AC_EPI_VARIANTS(AC_EPI_CALL)
#undef AC_EPI_CALL
                }
                csum += o;
            }
        }
    });
    // ac_gemm_desc.colsum: column sums of what this wave stored (its 64 rows) -> one atomic per column.  The bias
    // gradient of the layer whose activation backward the epilogue applied (dact): no separate pass over the tensor.
    if (d.colsum) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            csum[j] += __shfl_xor(csum[j], 16, 64);
            csum[j] += __shfl_xor(csum[j], 32, 64);
        }
        if (rsub == 0 && n < d.N) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(d.colsum + n + j, csum[j]);
        }
    }
}

__device__ __forceinline__ void store_tile_vec(const ac_gemm_desc &d, uint64_t dseed, int variant,
                                               const f32x16 (&acc)[2][2], float *wbuf, int row_base,
                                               int col_base, int lane) {
    switch (variant) {
#define AC_EPI_CASE(I, F) case I: store_tile_vec_t<I>(d, dseed, acc, wbuf, row_base, col_base, lane); break;
        AC_EPI_VARIANTS(AC_EPI_CASE)
#undef AC_EPI_CASE
        default: store_tile_vec_t<EPI_GENERIC>(d, dseed, acc, wbuf, row_base, col_base, lane); break;
    }
}

// ---------------------------------------------------------------------------
// Operand loaders.  KC: 128 rows x 8 chunks; thread t owns rows (t>>3)+32*i, chunk t&7.
// RC: 32 k-rows x 32 chunks; thread t owns k (t>>5)+8*i, chunk t&31.
// ---------------------------------------------------------------------------
template <bool KC>
struct Loader {
    const float *ptr;
    const int32_t *goff;
    ac_rowmap rows;
    int64_t base[4];  // KC: row address + 4*c ; RC: inner offset (same for all i)
    int64_t fbase[4]; // load_fast: everything of the address that does not depend on the K tile
    bool ok[4];
    int outer_n, inner_n;  // extents of the outer / inner index
    int t;

    __device__ __forceinline__ void init(const ac_mat &m, int outer_extent, int inner_extent,
                                         int tile_origin, int tid) {
        ptr = (const float *)m.ptr;
        goff = m.goff;
        rows = m.rows;
        outer_n = outer_extent;
        inner_n = inner_extent;
        t = tid;
        if (KC) {
            int c = t & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = tile_origin + (t >> 3) + 32 * i;
                ok[i] = r < outer_n;
                r = r < outer_n ? r : outer_n - 1;
                base[i] = ac_rowaddr(rows, r) + 4 * c;
                fbase[i] = base[i];
            }
        } else {
            int col = tile_origin + 4 * (t & 31);
            bool cv = col < inner_n;
            int64_t io = cv ? inner_off(goff, col) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                base[i] = io;
                ok[i] = cv;
                fbase[i] = io + (int64_t)((t >> 5) + 8 * i) * rows.s3;
            }
        }
    }

    // Loads are UNCONDITIONAL (addresses clamped into the operand) and masked afterwards: a
    // "load or zero" select makes hipcc branch around every load and wait for each one in turn
    // (cdna_hip_programming.md section 5, trap (c)) - a K tile then costs 4-8 dependent round trips.
    __device__ __forceinline__ void load(int kt, f32x4 (&v)[4]) const {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int k = kt * BK + 4 * (t & 7);
            const bool kv = k < inner_n;
            const int ktc = kv ? kt : 0;
            const int64_t ko = kv ? (goff ? (int64_t)goff[ktc] : (int64_t)ktc * BK) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 val = ac_gload<f32x4>(ptr + base[i] + (kv ? ko : -4 * (t & 7)));
                v[i] = kv ? val : zero;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kg = kt * BK + (t >> 5) + 8 * i;
                const bool okk = ok[i] && kg < outer_n;
                const f32x4 val = ac_gload<f32x4>(ptr + ac_rowaddr(rows, kg < outer_n ? kg : outer_n - 1) + base[i]);
                v[i] = okk ? val : zero;
            }
        }
    }

    // Raw form for software pipelines deeper than one tile: nothing consumes the loaded registers in
    // the iteration that issued them (mask bit i = element group i is inside the operand; applied by
    // whoever writes the registers to LDS).
    __device__ __forceinline__ void load_raw(int kt, f32x4 (&v)[4], unsigned &mask) const {
        if (KC) {
            const int k = kt * BK + 4 * (t & 7);
            const bool kv = k < inner_n;
            // The table entry is read through a UNIFORM index and an always-valid pointer (scalar load, no branch): a
            // conditional load put control flow into the pipelined loop, and hipcc then waits for EVERY load in flight
            // at the loop header (s_waitcnt vmcnt(0)) - the prefetch distance of 2 was 1 in effect.
            const int ktu = kt * BK < inner_n ? kt : 0;
            const int32_t gv = ac_gload<int32_t>(goff ? goff + ktu : (const int32_t *)ptr);
            const int64_t ku = goff ? (int64_t)gv : (int64_t)ktu * BK;
            mask = kv ? 0xFu : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ac_gload<f32x4>(ptr + base[i] + (kv ? ku : -4 * (t & 7)));
        } else {
            mask = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kg = kt * BK + (t >> 5) + 8 * i;
                const bool in = kg < outer_n;
                v[i] = ac_gload<f32x4>(ptr + ac_rowaddr(rows, in ? kg : outer_n - 1) + base[i]);
                mask |= (ok[i] && in) ? (1u << i) : 0u;
            }
        }
    }

    // Fast form, chosen per workgroup (uniform): the 128 rows / columns of this operand's tile are all inside, K is a
    // multiple of 32 and (RC) the row map is the plain one.  No masks, no selects, no branches; a prefetch past the K range
    // re-reads the last tile (its registers are never consumed).
    // (KC: no K-offset table either - its scalar load and the wait for it sat in front of every tile's loads.  RC: the
    // per-thread part of the row address is folded into fbase at init, the K tile adds one uniform offset.)
    __device__ __forceinline__ void load_fast(int kt, int nkt, f32x4 (&v)[4]) const {
        const int ktc = kt < nkt ? kt : nkt - 1;
        const float *tp = ptr + (KC ? (int64_t)ktc * BK : (int64_t)ktc * BK * rows.s3);   // uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ac_gload<f32x4>(tp + fbase[i]);
    }

    __device__ __forceinline__ void store(float *tile, const f32x4 (&v)[4]) const {
        if (KC) {
            int c = t & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = (t >> 3) + 32 * i;
                *(f32x4 *)(tile + r * 32 + ((c ^ ((r >> 1) & 7)) << 2)) = v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int k = (t >> 5) + 8 * i;
                *(f32x4 *)(tile + k * 128 + 4 * (t & 31)) = v[i];
            }
        }
    }
};

// fragment for k-step s (8 k values; this lane's half takes 4 of them)
template <bool KC>
__device__ __forceinline__ f32x4 read_frag(const float *tile, int local /* row or col 0..127 */,
                                           int s, int lh) {
    if (KC) {
        int chunk = (2 * s + lh) ^ ((local >> 1) & 7);
        return *(const f32x4 *)(tile + local * 32 + (chunk << 2));
    } else {
        int k = 8 * s + 4 * lh;
        f32x4 r;
        r[0] = tile[(k + 0) * 128 + local];
        r[1] = tile[(k + 1) * 128 + local];
        r[2] = tile[(k + 2) * 128 + local];
        r[3] = tile[(k + 3) * 128 + local];
        return r;
    }
}

template <bool A_KC, bool B_KC, bool BATCH = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint64_t dseed = ac_step_seed(p.d.drop_seed, p.d.drop_step);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: workgroups that share an XCD (bid % 8) walk contiguous tiles
    int wg, piece;
    map_workgroup(wg, piece);
    ac_gemm_desc dz;
    if constexpr (BATCH) batch_select(p, wg, dz);
    const ac_gemm_desc &d = BATCH ? dz : p.d;
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;

    const int kt_begin = piece * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > p.nkt) kt_end = p.nkt;
    if (kt_begin >= kt_end) return;

    Loader<A_KC> la;
    Loader<B_KC> lb;
    // KC: outer = row (M or N), inner = K.  RC: outer = K, inner = row.
    if (A_KC)
        la.init(d.a, d.M, d.K, tm * BM, t);
    else
        la.init(d.a, d.K, d.M, tm * BM, t);
    if (B_KC)
        lb.init(d.b, d.N, d.K, tn * BN, t);
    else
        lb.init(d.b, d.K, d.N, tn * BN, t);


    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[4], rb[4];
    la.load(kt_begin, ra);
    lb.load(kt_begin, rb);
    la.store(smem, ra);
    lb.store(smem + TILE_FLOATS, rb);
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) {
            la.load(kt + 1, ra);
            lb.load(kt + 1, rb);
        }
        const float *at = smem + cur * 2 * TILE_FLOATS, *bt = at + TILE_FLOATS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 af[2], bf[2];
            af[0] = read_frag<A_KC>(at, wm * 64 + li, s, lh);
            af[1] = read_frag<A_KC>(at, wm * 64 + 32 + li, s, lh);
            bf[0] = read_frag<B_KC>(bt, wn * 64 + li, s, lh);
            bf[1] = read_frag<B_KC>(bt, wn * 64 + 32 + li, s, lh);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][j], bf[0][j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][j], bf[1][j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][j], bf[0][j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][j], bf[1][j], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
            la.store(smem + (cur ^ 1) * 2 * TILE_FLOATS, ra);
            lb.store(smem + (cur ^ 1) * 2 * TILE_FLOATS + TILE_FLOATS, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (p.vec_epi == 3)
        store_tile_slab(d, acc, piece, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
    else if (p.vec_epi == 2)
        store_tile_atomic(d, acc, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
    else if (p.vec_epi)
        store_tile_vec(d, dseed, p.epi_var, acc, smem + wave * 2048, tm * BM + wm * 64, tn * BN + wn * 64, lane);
    else
        store_tile(d, dseed, acc, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
}

// ---------------------------------------------------------------------------
// bf16 matrix-core variant: same tiling and staging, operands rounded to bf16
// (round-to-nearest-even via the hardware convert) when they are written to LDS.
// LDS images hold bf16: KC image [128 rows][32 k] = 64-byte rows, 16-byte chunks
// (8 k) swizzled by ((row>>2)&3); RC image [32 k][128 cols] bf16.
// v_mfma_f32_32x32x16_bf16: lane (i=l&31, h=l>>5) supplies A[i][8h..8h+7], B[8h..8h+7][i].
// ---------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float x) {
    return ac_f2h(x);
}

template <bool KC>
__device__ __forceinline__ void store_bf16(const Loader<KC> &L, unsigned short *tile,
                                           const f32x4 (&v)[4]) {
    const int t = L.t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x4 h;
        h[0] = (short)f2bf(v[i][0]);
        h[1] = (short)f2bf(v[i][1]);
        h[2] = (short)f2bf(v[i][2]);
        h[3] = (short)f2bf(v[i][3]);
        if (KC) {
            int c = t & 7;  // 4-float chunk index along k: k = 4c..4c+3
            int r = (t >> 3) + 32 * i;
            int chunk16 = (c >> 1) ^ ((r >> 2) & 3);  // 16-byte chunk (8 bf16) index, swizzled
            *(bf16x4 *)(tile + r * 32 + chunk16 * 8 + (c & 1) * 4) = h;
        } else {
            int k = (t >> 5) + 8 * i;
            *(bf16x4 *)(tile + k * 128 + 4 * (t & 31)) = h;
        }
    }
}

// fragment for 16-deep k-step s (s = 0,1): lane half lh takes k = 16s + 8lh .. +7
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag_bf16(const unsigned short *tile, int local, int s,
                                                 int lh) {
    if (KC) {
        int chunk16 = (2 * s + lh) ^ ((local >> 2) & 3);
        return *(const bf16x8 *)(tile + local * 32 + chunk16 * 8);
    } else {
        int k = 16 * s + 8 * lh;
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (short)tile[(k + j) * 128 + local];
        return r;
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *sm16 = reinterpret_cast<unsigned short *>(smem);
    const ac_gemm_desc &d = p.d;
    const uint64_t dseed = ac_step_seed(p.d.drop_seed, p.d.drop_step);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    int wg, piece;
    map_workgroup(wg, piece);
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;

    const int kt_begin = piece * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > p.nkt) kt_end = p.nkt;
    if (kt_begin >= kt_end) return;

    Loader<A_KC> la;
    Loader<B_KC> lb;
    if (A_KC)
        la.init(d.a, d.M, d.K, tm * BM, t);
    else
        la.init(d.a, d.K, d.M, tm * BM, t);
    if (B_KC)
        lb.init(d.b, d.N, d.K, tn * BN, t);
    else
        lb.init(d.b, d.K, d.N, tn * BN, t);


    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[4], rb[4];
    la.load(kt_begin, ra);
    lb.load(kt_begin, rb);
    store_bf16<A_KC>(la, sm16, ra);
    store_bf16<B_KC>(lb, sm16 + TILE_FLOATS, rb);
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) {
            la.load(kt + 1, ra);
            lb.load(kt + 1, rb);
        }
        const unsigned short *at = sm16 + cur * 2 * TILE_FLOATS, *bt = at + TILE_FLOATS;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a0 = read_frag_bf16<A_KC>(at, wm * 64 + li, s, lh);
            bf16x8 a1 = read_frag_bf16<A_KC>(at, wm * 64 + 32 + li, s, lh);
            bf16x8 b0 = read_frag_bf16<B_KC>(bt, wn * 64 + li, s, lh);
            bf16x8 b1 = read_frag_bf16<B_KC>(bt, wn * 64 + 32 + li, s, lh);
            acc[0][0] = AC_MFMA16(a0, b0, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0, b1, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1, b0, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1, b1, acc[1][1]);
        }
        if (more) {
            store_bf16<A_KC>(la, sm16 + (cur ^ 1) * 2 * TILE_FLOATS, ra);
            store_bf16<B_KC>(lb, sm16 + (cur ^ 1) * 2 * TILE_FLOATS + TILE_FLOATS, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    if (p.vec_epi == 3)
        store_tile_slab(d, acc, piece, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
    else if (p.vec_epi == 2)
        store_tile_atomic(d, acc, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
    else if (p.vec_epi)
        store_tile_vec(d, dseed, p.epi_var, acc, smem + wave * 2048, tm * BM + wm * 64, tn * BN + wn * 64, lane);
    else
        store_tile(d, dseed, acc, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
}

// ---------------------------------------------------------------------------
// bf16-OPERAND kernels (math = AC_MATH_BF16_IN): A and B are already bf16 in HBM (cast once per
// call by ac_cast_bf16 / ac_transpose_cast_bf16, so the K-fold re-reads of the implicit-GEMM
// operands move half the bytes and the inner loop carries no conversion).  Tile 128x128x64,
// 4 waves, 16 MFMA 32x32x16 per wave per K-tile, double-buffered LDS, register staging.
//   NT: both operands "KC" ([128 rows][64 k] bf16 = 128-byte rows, the same chunk swizzle as the
//       fp32 image; one ds_read_b128 = one MFMA operand).
//   TN: both operands "RC" ([64 k][128 cols] bf16, row pitch 320 B); MFMA operands are gathered
//       with the hardware transpose read ds_read_b64_tr_b16 (two per operand), bank-conflict free
//       at that pitch.
// ---------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int BK16 = 64;

// ROWS x 64 bf16 image, NT threads: thread t owns rows (t>>3) + (NT/8)*i, 16-byte chunk t&7
template <int ROWS, int NT>
struct LoaderKC16 {
    static constexpr int NCH = ROWS * 8 / NT;
    const unsigned short *ptr;
    const int32_t *goff;
    int64_t base[NCH];
    int inner_n, t;
    __device__ __forceinline__ void init(const ac_mat &m, int outer_n, int inner_extent, int origin,
                                         int tid) {
        ptr = (const unsigned short *)m.ptr;
        goff = m.goff;
        inner_n = inner_extent;
        t = tid;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int r = origin + (t >> 3) + (NT / 8) * i;
            r = r < outer_n ? r : outer_n - 1;
            base[i] = ac_rowaddr(m.rows, r);
        }
    }
    // Raw (unconditional, clamped) loads; the K-tail mask is applied when the registers are
    // written to LDS one iteration later, so nothing consumes the loaded values in the iteration
    // that issued them (a select right after the load made hipcc wait for its own prefetch).
    template <bool G, bool FAST = true>
    __device__ __forceinline__ void load(int kt, u32x4 (&v)[NCH], unsigned &mask) const {
        const int e = kt * BK16 + 8 * (t & 7);
        const bool kv = e < inner_n;
        const int ec = kv ? e : 0;
        int64_t ko = ec;
        if constexpr (G) ko = (int64_t)ac_gload<int32_t>(goff + (ec >> 5)) + (ec & 31);
        mask = kv ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[i] = ac_gload<u32x4>(ptr + base[i] + ko);
    }
    __device__ __forceinline__ void store(unsigned short *tile, const u32x4 (&v)[NCH],
                                          unsigned mask) const {
        const int c = t & 7;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = (t >> 3) + (NT / 8) * i;
            u32x4 w = v[i];
            w[0] &= mask; w[1] &= mask; w[2] &= mask; w[3] &= mask;
            *(u32x4 *)(tile + r * 64 + ((c ^ ((r >> 1) & 7)) << 3)) = w;
        }
    }
};

__device__ __forceinline__ bf16x8 frag_kc16(const unsigned short *tile, int local, int s, int lh) {
    const int chunk = (2 * s + lh) ^ ((local >> 1) & 7);
    return *(const bf16x8 *)(tile + local * 64 + (chunk << 3));
}

// 64 x COLS bf16 image (row pitch COLS + 32): thread t owns k rows t/(COLS/8) + (NT*8/COLS)*i,
// 16-byte chunk t % (COLS/8)
template <int COLS, int NT>
struct LoaderRC16 {
    static constexpr int CPR = COLS / 8;          // chunks per k-row
    static constexpr int KSTEP = NT / CPR;        // k rows covered per pass
    static constexpr int NCH = 64 / KSTEP;
    static constexpr int PITCH = COLS + 32;
    const unsigned short *ptr;
    ac_rowmap rows;
    int64_t io;
    bool cv, fast;
    int outer_n, t;
    // 1- and 2-level row maps keep a running element address per chunk: one 64-bit add (+ a wrap
    // correction when the row index crosses a batch boundary) per K tile instead of re-deriving
    // (batch, position) with divisions and 64-bit multiplies for every chunk of every K tile
    int64_t addr[NCH], step, wrap_fix;
    int rem[NCH];
    __device__ __forceinline__ void init(const ac_mat &m, int outer_extent, int inner_extent,
                                         int origin, int tid, int kt0) {
        ptr = (const unsigned short *)m.ptr;
        rows = m.rows;
        outer_n = outer_extent;
        t = tid;
        const int col = origin + 8 * (t % CPR);
        cv = col < inner_extent;
        io = cv ? (m.goff ? (int64_t)m.goff[col >> 5] + (col & 31) : (int64_t)col) : 0;
        fast = rows.r1 == 0 || (rows.r2 == rows.r1 && rows.r1 >= BK16);
        step = (int64_t)BK16 * rows.s3;
        wrap_fix = rows.s1 - (int64_t)rows.r1 * rows.s3;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kg = kt0 * BK16 + t / CPR + KSTEP * i;
            if (rows.r1 == 0) {
                rem[i] = 0;
                addr[i] = (int64_t)kg * rows.s3 + io;
            } else {
                const int q1 = kg / rows.r1;
                rem[i] = kg - q1 * rows.r1;
                addr[i] = (int64_t)q1 * rows.s1 + (int64_t)rem[i] * rows.s3 + io;
            }
        }
    }
    // loads K tile kt (raw, clamped); must be called with consecutive kt starting at kt0.
    // mask bit i = chunk i is inside the operand (applied by store()).
    template <bool G, bool FAST = true>
    __device__ __forceinline__ void load(int kt, u32x4 (&v)[NCH], unsigned &mask) {
        mask = 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kg = kt * BK16 + t / CPR + KSTEP * i;
            const bool in = kg < outer_n;
            int64_t ra;
            if constexpr (FAST) {
                ra = addr[i];
                addr[i] += step;
                rem[i] += BK16;
                const bool wrap = rows.r1 != 0 && rem[i] >= rows.r1;  // r1 >= 64 on the fast path
                rem[i] -= wrap ? rows.r1 : 0;
                addr[i] += wrap ? wrap_fix : 0;
            } else {
                ra = ac_rowaddr(rows, in ? kg : 0) + io;
            }
            ra = in ? ra : io;  // keep the load inside the operand
            v[i] = ac_gload<u32x4>(ptr + ra);
            mask |= (cv && in) ? (1u << i) : 0u;
        }
    }
    __device__ __forceinline__ void store(unsigned short *tile, const u32x4 (&v)[NCH],
                                          unsigned mask) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int k = t / CPR + KSTEP * i;
            const unsigned m = (mask >> i) & 1u ? 0xFFFFFFFFu : 0u;
            u32x4 w = v[i];
            w[0] &= m; w[1] &= m; w[2] &= m; w[3] &= m;
            *(u32x4 *)(tile + k * PITCH + 8 * (t % CPR)) = w;
        }
    }
};

// MFMA operand (8 consecutive k of column `local`) from an RC image via two transpose reads.
// Lane l: 16-lane group g = l>>4 covers columns 16*(g&1).. and k half (g>>1) = l>>5; lane 4q+p of
// the group addresses row q, columns 4p..4p+3 of the 4x16 block and receives column (l&15).
// Row pitches of COLS+32 elements step 16 banks per k row: conflict free.
template <int PITCH>
__device__ __forceinline__ bf16x8 frag_rc16(const unsigned short *tile, int colbase, int s, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int k0 = 16 * s + 8 * (g >> 1);
    const unsigned short *a0 = tile + (k0 + q) * PITCH + colbase + 16 * (g & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(a0 + 4 * PITCH));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// ---------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant, math = AC_MATH_BF16X3: fp32 operands in HBM (the fp32 data flow of
// AC_MATH_F32 is unchanged), every element x is split when its tile is written to LDS into
//     hi = bf16(x)   and   lo = bf16(x - hi)          (both round-to-nearest-even),
// and every product of fragments is three matrix-core instructions, fp32 accumulate:
//     a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (the dropped lo*lo term is <= 2^-16 relative).
// hi + lo carries 16 mantissa bits, so one product is good to ~2^-16 instead of bf16's 2^-8: the mode
// meets the 1e-3 logit parity bar of the path at 3 of the 16 x cheaper bf16 MFMAs per product instead
// of the fp32 MFMA.  Same 128x128x32 tile / staging as the kernels above; LDS holds four bf16 images
// per stage (A_hi, A_lo, B_hi, B_lo).  RC images ([k][cols], row pitch 160 elements) are read with
// the hardware transpose read like the bf16-operand TN kernel.
// ---------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));
constexpr int X3_RC_PITCH = 128 + 32;                 // bf16 elements per k row of an RC image
constexpr int X3_KC_ELEMS = 128 * 32;
constexpr int X3_RC_ELEMS = 32 * X3_RC_PITCH;

// x -> (hi, lo) for 4 floats; packed 4 x bf16 each
__device__ __forceinline__ void split4(const f32x4 &v, s16x4_t &hi, s16x4_t &lo) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, l0, h1, l1;
    ac_split_pair(v[0], v[1], h0, l0);
    ac_split_pair(v[2], v[3], h1, l1);
    const u32x2 h = {h0, h1}, l = {l0, l1};
    hi = __builtin_bit_cast(s16x4_t, h);
    lo = __builtin_bit_cast(s16x4_t, l);
}

template <bool KC, bool FAST = false>
__device__ __forceinline__ void store_x3(const Loader<KC> &L, unsigned short *img_hi, unsigned short *img_lo,
                                         const f32x4 (&v)[4], unsigned mask) {
    const int t = L.t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s16x4_t hi, lo;
        const bool in = FAST || ((mask >> i) & 1u);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        split4(in ? v[i] : zero, hi, lo);
        int off;
        if (KC) {
            const int c = t & 7, r = (t >> 3) + 32 * i;
            off = r * 32 + (((c >> 1) ^ ((r >> 2) & 3)) << 3) + (c & 1) * 4;
        } else {
            const int k = (t >> 5) + 8 * i;
            off = k * X3_RC_PITCH + 4 * (t & 31);
        }
        *(s16x4_t *)(img_hi + off) = hi;
        *(s16x4_t *)(img_lo + off) = lo;
    }
}

template <bool KC>
__device__ __forceinline__ bf16x8 frag_x3(const unsigned short *img, int colbase, int s, int lane) {
    if (KC) {
        const int local = colbase + (lane & 31), lh = lane >> 5;
        const int chunk16 = (2 * s + lh) ^ ((local >> 2) & 3);
        return *(const bf16x8 *)(img + local * 32 + chunk16 * 8);
    } else {
        return frag_rc16<X3_RC_PITCH>(img, colbase, s, lane);
    }
}

// B operand from CACHED (hi, lo) planes (ac_gemm_desc.b_hi / b_lo: a weight matrix split once per optimizer step by
// ac_split_bf16): a K tile is two 16-byte loads per plane and thread, written to the LDS images as they are - no split
// arithmetic in the loop and half the operand bytes.  Every workgroup used to split the same weight tile again
// (~3 VALU instructions per element per read; the kernel's small-K launches are bound by VALU issue and latency).
template <bool KC>
struct PlaneLoader {
    const unsigned short *hi, *lo;
    int64_t ld;
    int outer_n, inner_n, origin, t;   // KC: outer = row (N), inner = K.  RC: outer = K, inner = column (N).

    __device__ __forceinline__ void init(const ac_gemm_desc &d, int outer_extent, int inner_extent, int tile_origin, int tid) {
        hi = (const unsigned short *)d.b_hi;
        lo = (const unsigned short *)d.b_lo;
        ld = d.ld_bpl;
        outer_n = outer_extent;
        inner_n = inner_extent;
        origin = tile_origin;
        t = tid;
    }
    __device__ __forceinline__ void load_raw(int kt, bf16x8 (&vh)[2], bf16x8 (&vl)[2], unsigned &mask) const {
        mask = 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t off;
            bool in;
            if (KC) {
                const int r = origin + (t >> 2) + 64 * i, k = kt * BK + 8 * (t & 3);
                in = r < outer_n && k < inner_n;
                off = in ? (int64_t)r * ld + k : 0;
            } else {
                const int kr = kt * BK + (t >> 4) + 16 * i, c = origin + 8 * (t & 15);
                in = kr < outer_n && c < inner_n;
                off = in ? (int64_t)kr * ld + c : 0;
            }
            vh[i] = ac_gload<bf16x8>((const short *)hi + off);
            vl[i] = ac_gload<bf16x8>((const short *)lo + off);
            mask |= in ? (1u << i) : 0u;
        }
    }
    // fast form (see Loader::load_fast): the tile's 128 N entries inside, K a multiple of 32
    __device__ __forceinline__ void load_fast(int kt, int nkt, bf16x8 (&vh)[2], bf16x8 (&vl)[2]) const {
        const int ktc = kt < nkt ? kt : nkt - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t off = KC ? (int64_t)(origin + (t >> 2) + 64 * i) * ld + ktc * BK + 8 * (t & 3)
                                   : (int64_t)(ktc * BK + (t >> 4) + 16 * i) * ld + origin + 8 * (t & 15);
            vh[i] = ac_gload<bf16x8>((const short *)hi + off);
            vl[i] = ac_gload<bf16x8>((const short *)lo + off);
        }
    }
    template <bool FAST = false>
    __device__ __forceinline__ void store(unsigned short *img_hi, unsigned short *img_lo, const bf16x8 (&vh)[2],
                                          const bf16x8 (&vl)[2], unsigned mask) const {
        const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int off;
            if (KC) {
                const int r = (t >> 2) + 64 * i, c16 = t & 3;
                off = r * 32 + ((c16 ^ ((r >> 2) & 3)) << 3);
            } else {
                off = ((t >> 4) + 16 * i) * X3_RC_PITCH + 8 * (t & 15);
            }
            const bool in = FAST || ((mask >> i) & 1u);
            *(bf16x8 *)(img_hi + off) = in ? vh[i] : zero;
            *(bf16x8 *)(img_lo + off) = in ? vl[i] : zero;
        }
    }
};

template <bool A_KC, bool B_KC, bool BATCH = false, bool B_PL = false>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *sm16 = reinterpret_cast<unsigned short *>(smem);
    constexpr int A_IMG = A_KC ? X3_KC_ELEMS : X3_RC_ELEMS;
    constexpr int B_IMG = B_KC ? X3_KC_ELEMS : X3_RC_ELEMS;
    constexpr int STAGE = 2 * A_IMG + 2 * B_IMG;
    const uint64_t dseed = ac_step_seed(p.d.drop_seed, p.d.drop_step);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    int wg, piece;
    map_workgroup(wg, piece);
    ac_gemm_desc dz;
    if constexpr (BATCH) batch_select(p, wg, dz);
    const ac_gemm_desc &d = BATCH ? dz : p.d;
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;

    const int kt_begin = piece * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > p.nkt) kt_end = p.nkt;
    if (kt_begin >= kt_end) return;

    Loader<A_KC> la;
    Loader<B_KC> lb;
    PlaneLoader<B_KC> lp;
    if (A_KC)
        la.init(d.a, d.M, d.K, tm * BM, t);
    else
        la.init(d.a, d.K, d.M, tm * BM, t);
    if constexpr (B_PL) {
        if (B_KC)
            lp.init(d, d.N, d.K, tn * BN, t);
        else
            lp.init(d, d.K, d.N, tn * BN, t);
    } else {
        if (B_KC)
            lb.init(d.b, d.N, d.K, tn * BN, t);
        else
            lb.init(d.b, d.K, d.N, tn * BN, t);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Software pipeline, prefetch distance 2 (two register sets, two LDS stages, one barrier per K
    // tile): while tile kt feeds the matrix cores, tile kt+1 waits in registers and the loads of tile
    // kt+2 are in flight.  The small-K products of the image / photometry branches run one workgroup
    // per CU or less, where a distance-1 pipeline exposed the whole L2 latency in every K tile
    // (1.9 us per 32-deep tile against 0.3 us of MFMA work).  Loads past the K range are clamped and
    // zero-masked by the loaders, so the steady state needs no data-dependent branch.
    auto compute = [&](const unsigned short *stage) {
        const unsigned short *ah = stage, *al = ah + A_IMG;
        const unsigned short *bh = ah + 2 * A_IMG, *bl = bh + B_IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 a0h = frag_x3<A_KC>(ah, wm * 64, s, lane), a1h = frag_x3<A_KC>(ah, wm * 64 + 32, s, lane);
            const bf16x8 b0h = frag_x3<B_KC>(bh, wn * 64, s, lane), b1h = frag_x3<B_KC>(bh, wn * 64 + 32, s, lane);
            const bf16x8 a0l = frag_x3<A_KC>(al, wm * 64, s, lane), a1l = frag_x3<A_KC>(al, wm * 64 + 32, s, lane);
            const bf16x8 b0l = frag_x3<B_KC>(bl, wn * 64, s, lane), b1l = frag_x3<B_KC>(bl, wn * 64 + 32, s, lane);
            // cross terms first, the leading term last
            acc[0][0] = AC_MFMA16(a0l, b0h, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0l, b1h, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1l, b0h, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1l, b1h, acc[1][1]);
            acc[0][0] = AC_MFMA16(a0h, b0l, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0h, b1l, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1h, b0l, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1h, b1l, acc[1][1]);
            acc[0][0] = AC_MFMA16(a0h, b0h, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0h, b1h, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1h, b0h, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1h, b1h, acc[1][1]);
        }
    };
    // B registers of one K tile: four fp32 float4s (split when stored) or, from cached planes, two 16-byte pieces of
    // each plane (stored as they are)
    struct BRegs {
        f32x4 f[B_PL ? 1 : 4];
        bf16x8 h[B_PL ? 2 : 1], l[B_PL ? 2 : 1];
    };
    // The loop exists in four copies, chosen per workgroup and operand (uniform): FA / FB = the operand's tile takes the
    // fast loader (all 128 rows / columns inside, K a multiple of 32, no K-offset table for a KC operand, plain row map
    // for an RC operand): no masks, no selects behind the split, no scalar loads or control flow between the loads -
    // interior tiles, i.e. nearly all of them.  Edge
    // tiles and gathered K rows keep the general loaders.
    const int nkt = p.nkt;
    auto mainloop = [&](auto fa_c, auto fb_c) {
        constexpr bool FA = decltype(fa_c)::value, FB = decltype(fb_c)::value;
        auto load_a = [&](int kt, f32x4 (&r)[4], unsigned &m) {
            if constexpr (FA) {
                la.load_fast(kt, nkt, r);
                m = 0xFu;
            } else {
                la.load_raw(kt, r, m);
            }
        };
        auto load_b = [&](int kt, BRegs &r, unsigned &m) {
            if constexpr (B_PL) {
                if constexpr (FB) {
                    lp.load_fast(kt, nkt, r.h, r.l);
                    m = 0x3u;
                } else {
                    lp.load_raw(kt, r.h, r.l, m);
                }
            } else if constexpr (FB) {
                lb.load_fast(kt, nkt, r.f);
                m = 0xFu;
            } else {
                lb.load_raw(kt, r.f, m);
            }
        };
        auto stage_store = [&](unsigned short *stage, const f32x4 (&ra)[4], const BRegs &rb, unsigned ma, unsigned mb) {
            store_x3<A_KC, FA>(la, stage, stage + A_IMG, ra, ma);
            if constexpr (B_PL)
                lp.template store<FB>(stage + 2 * A_IMG, stage + 2 * A_IMG + B_IMG, rb.h, rb.l, mb);
            else
                store_x3<B_KC, FB>(lb, stage + 2 * A_IMG, stage + 2 * A_IMG + B_IMG, rb.f, mb);
        };
        unsigned short *S0 = sm16, *S1 = sm16 + STAGE;
        f32x4 ra0[4], ra1[4];
        BRegs rb0, rb1;
        unsigned ma0, mb0, ma1, mb1;
        // Prologue: the first TWO tiles' loads go out back to back (most launches of the step are short-K - 3 to 16
        // tiles - and a workgroup that asked for tile 1 only after tile 0 had landed and been stored paid two load
        // latencies before its second tile); tile 0 is staged in S1 and the first step below is the second half of a
        // trip, peeled: tile 0 computed from S1, tile 1 stored to S0, tile 2 requested.
        load_a(kt_begin, ra0, ma0);
        load_b(kt_begin, rb0, mb0);
        load_a(kt_begin + 1, ra1, ma1);
        load_b(kt_begin + 1, rb1, mb1);
        __builtin_amdgcn_sched_barrier(0);
        stage_store(S1, ra0, rb0, ma0, mb0);
        __syncthreads();
        load_a(kt_begin + 2, ra0, ma0);
        load_b(kt_begin + 2, rb0, mb0);
        __builtin_amdgcn_sched_barrier(0);
        compute(S1);
        stage_store(S0, ra1, rb1, ma1, mb1);
        __syncthreads();
        // Two K tiles per trip and NO exit between them: a `break` after the first half gave the loop header a second
        // back edge on which the first half's loads are still in flight, and hipcc then waits for every load
        // (s_waitcnt vmcnt(0)) at the top of each trip - the prefetch distance of 2 was 1 in effect.  The last trip of
        // an even count stores one tile nobody reads; an odd count ends with the tile left in S0.
        int kt = kt_begin + 1;
        for (; kt + 1 < kt_end; kt += 2) {
            load_a(kt + 2, ra1, ma1);
            load_b(kt + 2, rb1, mb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(S0);
            stage_store(S1, ra0, rb0, ma0, mb0);
            __syncthreads();
            load_a(kt + 3, ra0, ma0);
            load_b(kt + 3, rb0, mb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(S1);
            stage_store(S0, ra1, rb1, ma1, mb1);
            __syncthreads();
        }
        if (kt < kt_end) {
            compute(S0);
            __syncthreads();   // the epilogues park their tiles in this buffer
        }
    };
    {
        const bool kfull = (d.K % BK) == 0;
        const bool fa = kfull && tm * BM + BM <= d.M && (A_KC ? d.a.goff == nullptr : d.a.rows.r1 == 0);
        const bool fb = kfull && tn * BN + BN <= d.N && (B_PL || (B_KC ? d.b.goff == nullptr : d.b.rows.r1 == 0));
        if (fa && fb)
            mainloop(std::true_type{}, std::true_type{});
        else if (fa)
            mainloop(std::true_type{}, std::false_type{});
        else if (fb)
            mainloop(std::false_type{}, std::true_type{});
        else
            mainloop(std::false_type{}, std::false_type{});
    }

    if (p.vec_epi == 3)
        store_tile_slab(d, acc, piece, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
    else if (p.vec_epi == 2)
        store_tile_atomic(d, acc, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
    else if (p.vec_epi)
        store_tile_vec(d, dseed, p.epi_var, acc, smem + wave * 2048, tm * BM + wm * 64, tn * BN + wn * 64, lane);
    else
        store_tile(d, dseed, acc, tm * BM + wm * 64, tn * BN + wn * 64, li, lh);
}

template <bool A_KC, bool B_KC, bool BATCH = false, bool B_PL = false>
int launch_x3(const GemmParams &p, dim3 grid, hipStream_t stream) {
    constexpr int A_IMG = A_KC ? X3_KC_ELEMS : X3_RC_ELEMS;
    constexpr int B_IMG = B_KC ? X3_KC_ELEMS : X3_RC_ELEMS;
    constexpr int LDS = 2 * (2 * A_IMG + 2 * B_IMG) * 2;   // two stages, bytes
    static_assert(LDS >= 4 * 2048 * 4, "the 16-byte epilogue parks 8 KB per wave in this buffer");
    static const hipError_t attr = hipFuncSetAttribute((const void *)gemm_x3_kernel<A_KC, B_KC, BATCH, B_PL>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr != hipSuccess) return -(int)attr - 2000;
    hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, BATCH, B_PL>), grid, dim3(256), LDS, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// WM x WN waves, each owning a 64x64 output block: tile (64*WM) x (64*WN) x 64.
template <bool TN, int WM, int WN>
struct Bf16Cfg {
    static constexpr int NT = WM * WN * 64;
    static constexpr int TM = WM * 64, TNn = WN * 64;
    static constexpr int A_TILE = TN ? 64 * (TM + 32) : TM * 64;   // ushort elements
    static constexpr int B_TILE = TN ? 64 * (TNn + 32) : TNn * 64;
    static constexpr int STAGE = A_TILE + B_TILE;
    static constexpr int LDS_BYTES = 2 * STAGE * 2;
};

template <bool TN, int WM, int WN>
__global__ __launch_bounds__(WM *WN * 64, (WM * WN >= 8) ? 1 : 2) void gemm_bf16in_kernel(GemmParams p) {
    using Cfg = Bf16Cfg<TN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short *sm = reinterpret_cast<unsigned short *>(smem);
    const ac_gemm_desc &d = p.d;
    const uint64_t dseed = ac_step_seed(p.d.drop_seed, p.d.drop_step);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    int wg, piece;
    map_workgroup(wg, piece);
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;

    const int kt_begin = piece * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split;
    if (kt_end > p.nkt) kt_end = p.nkt;
    if (kt_begin >= kt_end) return;

    using LA = typename std::conditional<TN, LoaderRC16<Cfg::TM, Cfg::NT>, LoaderKC16<Cfg::TM, Cfg::NT>>::type;
    using LB = typename std::conditional<TN, LoaderRC16<Cfg::TNn, Cfg::NT>, LoaderKC16<Cfg::TNn, Cfg::NT>>::type;
    LA la;
    LB lb;
    if constexpr (TN) {
        la.init(d.a, d.K, d.M, tm * Cfg::TM, t, kt_begin);
        lb.init(d.b, d.K, d.N, tn * Cfg::TNn, t, kt_begin);
    } else {
        la.init(d.a, d.M, d.K, tm * Cfg::TM, t);
        lb.init(d.b, d.N, d.K, tn * Cfg::TNn, t);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Software pipeline, prefetch distance 2: while tile kt feeds the matrix cores, tile kt+1 is
    // landing in one register set and the loads of tile kt+2 are issued into the other (a K-tile
    // iteration was latency bound at ~1.3 us with distance 1: the global loads of the next tile
    // only had the 0.2 us of MFMA work to hide under).  Two LDS stages, one barrier per tile.
    auto compute = [&](const unsigned short *at, const unsigned short *bt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a0, a1, b0, b1;
            if (TN) {
                a0 = frag_rc16<Cfg::TM + 32>(at, wm * 64, s, lane);
                a1 = frag_rc16<Cfg::TM + 32>(at, wm * 64 + 32, s, lane);
                b0 = frag_rc16<Cfg::TNn + 32>(bt, wn * 64, s, lane);
                b1 = frag_rc16<Cfg::TNn + 32>(bt, wn * 64 + 32, s, lane);
            } else {
                a0 = frag_kc16(at, wm * 64 + li, s, lh);
                a1 = frag_kc16(at, wm * 64 + 32 + li, s, lh);
                b0 = frag_kc16(bt, wn * 64 + li, s, lh);
                b1 = frag_kc16(bt, wn * 64 + 32 + li, s, lh);
            }
            acc[0][0] = AC_MFMA16(a0, b0, acc[0][0]);
            acc[0][1] = AC_MFMA16(a0, b1, acc[0][1]);
            acc[1][0] = AC_MFMA16(a1, b0, acc[1][0]);
            acc[1][1] = AC_MFMA16(a1, b1, acc[1][1]);
        }
    };
    unsigned short *S0 = sm, *S1 = sm + Cfg::STAGE;
    // Steady-state loop without data-dependent branches: the offset-table lookups are compiled in or
    // out per operand (AG / BG), and tail iterations re-load the last tile into an unused register
    // set / LDS stage instead of skipping, so that each half of the loop is ONE basic block and
    // hipcc uses counted vmcnt waits (a load inside an `if` made it wait vmcnt(0) in front of the LDS
    // store, i.e. for the prefetch it had just issued).
    auto run = [&](auto AGt, auto BGt, auto FSt) {
        constexpr bool AG = decltype(AGt)::value, BG = decltype(BGt)::value, FS = decltype(FSt)::value;
        u32x4 ra0[LA::NCH], rb0[LB::NCH], ra1[LA::NCH], rb1[LB::NCH];
        unsigned ma0, mb0, ma1, mb1;
        la.template load<AG, FS>(kt_begin, ra0, ma0);
        lb.template load<BG, FS>(kt_begin, rb0, mb0);
        la.store(S0, ra0, ma0);
        lb.store(S0 + Cfg::A_TILE, rb0, mb0);
        __syncthreads();
        // Tiles past the end of this block's K range are still loaded (valid memory: rows / k
        // beyond the operand are redirected to offset 0 and masked) and land in a stage that is
        // never multiplied; the TN loader's incremental row state needs strictly consecutive calls.
        la.template load<AG, FS>(kt_begin + 1, ra0, ma0);
        lb.template load<BG, FS>(kt_begin + 1, rb0, mb0);
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            la.template load<AG, FS>(kt + 2, ra1, ma1);
            lb.template load<BG, FS>(kt + 2, rb1, mb1);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch issue ahead of the MFMA phase
            compute(S0, S0 + Cfg::A_TILE);
            // NT: the mask + LDS store of the prefetched registers (and the vmcnt waits in front of
            // them) must not be hoisted between the MFMAs — hipcc otherwise gates MFMA #1 on a load
            // issued only half an iteration earlier (+5 % on 8192x8192x4096; the TN loop, whose
            // fragment reads are twice as many, measured 4 % slower with the pin and keeps hipcc's order)
            if constexpr (!TN) __builtin_amdgcn_sched_barrier(0);
            la.store(S1, ra0, ma0);
            lb.store(S1 + Cfg::A_TILE, rb0, mb0);
            __syncthreads();
            if (kt + 1 >= kt_end) break;
            la.template load<AG, FS>(kt + 3, ra0, ma0);
            lb.template load<BG, FS>(kt + 3, rb0, mb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(S1, S1 + Cfg::A_TILE);
            if constexpr (!TN) __builtin_amdgcn_sched_barrier(0);
            la.store(S0, ra1, ma1);
            lb.store(S0 + Cfg::A_TILE, rb1, mb1);
            __syncthreads();
        }
    };
    if constexpr (TN) {
        // the division-based 3-level row map (2x2 patch gathers) is a separate loop instance so the
        // common running-address loader carries none of its code
        if (la.fast && lb.fast)
            run(std::false_type{}, std::false_type{}, std::true_type{});
        else
            run(std::false_type{}, std::false_type{}, std::false_type{});
    } else {
        const bool ag = d.a.goff != nullptr, bg = d.b.goff != nullptr;
        if (ag && bg)
            run(std::true_type{}, std::true_type{}, std::true_type{});
        else if (ag)
            run(std::true_type{}, std::false_type{}, std::true_type{});
        else if (bg)
            run(std::false_type{}, std::true_type{}, std::true_type{});
        else
            run(std::false_type{}, std::false_type{}, std::true_type{});
    }
    if (p.vec_epi == 2)
        store_tile_atomic(d, acc, tm * Cfg::TM + wm * 64, tn * Cfg::TNn + wn * 64, li, lh);
    else if (p.vec_epi)
        store_tile_vec(d, dseed, p.epi_var, acc, smem + wave * 2048, tm * Cfg::TM + wm * 64, tn * Cfg::TNn + wn * 64, lane);
    else
        store_tile(d, dseed, acc, tm * Cfg::TM + wm * 64, tn * Cfg::TNn + wn * 64, li, lh);
}

template <bool TN, int WM, int WN>
int launch_bf16in(GemmParams &p, hipStream_t stream) {
    using Cfg = Bf16Cfg<TN, WM, WN>;
    const ac_gemm_desc &d = p.d;
    p.tiles_m = (d.M + Cfg::TM - 1) / Cfg::TM;
    p.tiles_n = (d.N + Cfg::TNn - 1) / Cfg::TNn;
    p.nkt = (d.K + BK16 - 1) / BK16;
    p.kt_per_split = (p.nkt + d.split_k - 1) / d.split_k;
    static const hipError_t attr = hipFuncSetAttribute(
        (const void *)gemm_bf16in_kernel<TN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize,
        Cfg::LDS_BYTES);
    if (attr != hipSuccess) return -(int)attr - 2000;
    dim3 grid(p.tiles_m * p.tiles_n, d.split_k);
    hipLaunchKernelGGL((gemm_bf16in_kernel<TN, WM, WN>), grid, dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// fp32 -> bf16 (round to nearest even), 8 elements per thread
__global__ void cast_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ y,
                                 int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4 *)(x + 8 * i), b = *(const f32x4 *)(x + 8 * i + 4);
        bf16x8 o;
        o[0] = (short)f2bf(a[0]); o[1] = (short)f2bf(a[1]); o[2] = (short)f2bf(a[2]); o[3] = (short)f2bf(a[3]);
        o[4] = (short)f2bf(b[0]); o[5] = (short)f2bf(b[1]); o[6] = (short)f2bf(b[2]); o[7] = (short)f2bf(b[3]);
        *(bf16x8 *)(y + 8 * i) = o;
    }
}
__global__ void cast_bf16_tail_kernel(const float *__restrict__ x, unsigned short *__restrict__ y,
                                      int64_t begin, int64_t n) {
    int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f2bf(x[i]);
}

// y[c, r] = bf16(x[r, c]) through a 64x64 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float *__restrict__ x,
                                                                  int64_t ldx,
                                                                  unsigned short *__restrict__ y,
                                                                  int64_t ldy, int64_t rows,
                                                                  int cols) {
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
        if (cc < cols && rr < rows) y[(int64_t)cc * ldy + rr] = f2bf(tile[tx][i]);
    }
}

// All 2-D weights of a flat parameter buffer in ONE launch: segment s = {offset, rows, cols,
// first tile}; dst holds, at the same offset, the [cols, rows] bf16 transpose.  (One launch per
// weight per optimizer step cost ~100 launches of ~10 us.)
__global__ __launch_bounds__(256) void transpose_cast_segments_kernel(
    const float *__restrict__ src, unsigned short *__restrict__ dst, const int64_t *__restrict__ segs,
    int nseg) {
    __shared__ float tile[64][65];
    __shared__ int sseg;
    const int tileid = blockIdx.x;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nseg - 1;  // last segment whose first tile <= tileid
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[4 * mid + 3] <= tileid) lo = mid; else hi = mid - 1;
        }
        sseg = lo;
    }
    __syncthreads();
    const int64_t off = segs[4 * sseg], rows = segs[4 * sseg + 1], cols = segs[4 * sseg + 2];
    const int local = tileid - (int)segs[4 * sseg + 3];
    const int tcols = (int)((cols + 63) / 64);
    const int64_t r0 = (int64_t)(local / tcols) * 64;
    const int c0 = (local % tcols) * 64;
    const float *x = src + off;
    unsigned short *y = dst + off;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t rr = r0 + i;
        const int cc = c0 + tx;
        tile[i][tx] = (rr < rows && cc < cols) ? x[rr * cols + cc] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int cc = c0 + i;
        const int64_t rr = r0 + tx;
        if (cc < cols && rr < rows) y[(int64_t)cc * rows + rr] = f2bf(tile[tx][i]);
    }
}

// ---------------------------------------------------------------------------
// Scalar kernel for tiny or unaligned products (metadata towers, router, class
// heads: K, N of 2..48).  One thread per output element.
// ---------------------------------------------------------------------------
__global__ void gemm_simple_kernel(GemmParams p) {
    const ac_gemm_desc &d = p.d;
    const uint64_t dseed = ac_step_seed(p.d.drop_seed, p.d.drop_step);
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)d.M * d.N) return;
    int m = (int)(idx / d.N), n = (int)(idx % d.N);
    const float *A = (const float *)d.a.ptr, *B = (const float *)d.b.ptr;
    float acc = 0.f;
    if (d.mode == AC_GEMM_NT) {
        int64_t ar = ac_rowaddr(d.a.rows, m), br = ac_rowaddr(d.b.rows, n);
        for (int k = 0; k < d.K; ++k)
            acc = fmaf(A[ar + inner_off(d.a.goff, k)], B[br + inner_off(d.b.goff, k)], acc);
    } else if (d.mode == AC_GEMM_NN) {
        int64_t ar = ac_rowaddr(d.a.rows, m), bi = inner_off(d.b.goff, n);
        for (int k = 0; k < d.K; ++k)
            acc = fmaf(A[ar + inner_off(d.a.goff, k)], B[ac_rowaddr(d.b.rows, k) + bi], acc);
    } else {
        int64_t ai = inner_off(d.a.goff, m), bi = inner_off(d.b.goff, n);
        for (int k = 0; k < d.K; ++k)
            acc = fmaf(A[ac_rowaddr(d.a.rows, k) + ai], B[ac_rowaddr(d.b.rows, k) + bi], acc);
    }
    epilogue_store(d, dseed, m, n, acc, ac_rowaddr(d.c.rows, m) + inner_off(d.c.goff, n));
}

// TN with tiny M x N and a long reduction (tower / router weight gradients: K = batch rows):
// one wave per output element, lanes stride the reduction, shuffle-reduce.
__global__ __launch_bounds__(256) void gemm_simple_tn_wave_kernel(GemmParams p) {
    const ac_gemm_desc &d = p.d;
    const uint64_t dseed = ac_step_seed(p.d.drop_seed, p.d.drop_step);
    const int64_t widx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (widx >= (int64_t)d.M * d.N) return;
    const int lane = threadIdx.x & 63;
    const int m = (int)(widx / d.N), n = (int)(widx % d.N);
    const float *A = (const float *)d.a.ptr, *B = (const float *)d.b.ptr;
    const int64_t ai = inner_off(d.a.goff, m), bi = inner_off(d.b.goff, n);
    float acc = 0.f;
    for (int k = lane; k < d.K; k += 64)
        acc = fmaf(A[ac_rowaddr(d.a.rows, k) + ai], B[ac_rowaddr(d.b.rows, k) + bi], acc);
    acc = ac_wave_sum(acc);
    if (lane == 0) epilogue_store(d, dseed, m, n, acc, ac_rowaddr(d.c.rows, m) + inner_off(d.c.goff, n));
}

bool rowmap_aligned(const ac_rowmap &r) {
    return (r.s1 % 4 == 0) && (r.s2 % 4 == 0) && (r.s3 % 4 == 0);
}

int epilogue_variant(const ac_gemm_desc &d, int accumulate) {
    if (d.c.rows.r1 != 0 || accumulate == 2) return EPI_GENERIC;
    unsigned f = d.c.goff ? E_GOFF : 0;
    if (d.bias) f |= E_BIAS;
    if (d.pre_out) f |= E_PRE;
    if (d.act == AC_ACT_GELU || d.act == AC_ACT_GELU_FAST) f |= E_GELU;
    else if (d.act == AC_ACT_RELU) f |= E_RELU;
    else if (d.act != AC_ACT_NONE) return EPI_GENERIC;
    if (d.dact == AC_ACT_GELU || d.dact == AC_ACT_GELU_FAST) f |= E_DGELU;
    else if (d.dact == AC_ACT_RELU) f |= E_DRELU;
    else if (d.dact != AC_ACT_NONE) return EPI_GENERIC;
    if (d.mask16) f |= E_MASK16;
    if (d.colscale) f |= E_CSCALE;
    if (d.drop_p > 0.f) f |= E_DROP;
    if (d.residual) f |= E_RES;
    if (d.c16) f |= E_C16;
    if (d.c.ptr) f |= E_C32;
    if (accumulate == 1) f |= E_ACC;
    // every mode but the exact-fp32 one takes the rational erf (the backward kernels of a split-bf16 step get
    // AC_ACT_GELU_FAST from the host, so forward and backward differentiate the same function)
    if ((d.math != AC_MATH_F32 || d.act == AC_ACT_GELU_FAST || d.dact == AC_ACT_GELU_FAST) && (f & (E_GELU | E_DGELU)))
        f |= E_FAST;
#define AC_EPI_FIND(I, F) if (f == (F)) return I;
    AC_EPI_VARIANTS(AC_EPI_FIND)
#undef AC_EPI_FIND
    return EPI_GENERIC;
}

int vec_epilogue_ok(const ac_gemm_desc &d, int accumulate) {
    if (accumulate == 2) {  // atomics keep the 128-byte-per-row lane layout
        const bool lean = !d.bias && !d.pre_out && !d.act && !d.dact && !d.colscale && !d.residual &&
                          !d.mask16 && !d.c16 && d.drop_p <= 0.f && d.c.rows.r1 == 0 && !d.c.goff;
        return lean ? 2 : 0;
    }
    if (d.N % 4) return 0;
    if (d.c.ptr && (!ac_aligned16(d.c.ptr) || !rowmap_aligned(d.c.rows))) return 0;
    if (d.bias && !ac_aligned16(d.bias)) return 0;
    if (d.colscale && !ac_aligned16(d.colscale)) return 0;
    if (d.pre_out && (!ac_aligned16(d.pre_out) || (d.ld_pre % 4))) return 0;
    if (d.aux && (!ac_aligned16(d.aux) || (d.ld_aux % 4))) return 0;
    if (d.residual && (!ac_aligned16(d.residual) || (d.ld_res % 4))) return 0;
    if (d.c16 && (((uintptr_t)d.c16 & 7u) || (d.ld_c16 % 4))) return 0;
    if (d.mask16 && (((uintptr_t)d.mask16 & 7u) || (d.ld_mask16 % 4))) return 0;
    return 1;  // c.goff entries are multiples of 4 by contract
}

}  // namespace

static int gemm_run(const ac_gemm_desc *dp, ac_stream_t stream_, int batch, int64_t bs_a, int64_t bs_b, int64_t bs_c,
                    const void *const *grp = nullptr) {
    if (!dp) return AC_EINVAL;
    ac_gemm_desc d = *dp;
    if (grp) {   // grouped: the descriptor's pointers are those of product 0
        d.a.ptr = grp[0];
        d.b.ptr = grp[1];
        d.c.ptr = const_cast<void *>(grp[2]);
    }
    hipStream_t stream = (hipStream_t)stream_;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return AC_EINVAL;
    if (d.mode < 0 || d.mode > 2) return AC_EINVAL;
    if (!d.a.ptr || !d.b.ptr || (!d.c.ptr && !d.c16)) return AC_EINVAL;
    if (!d.c.ptr && (d.accumulate || d.split_k > 1)) return AC_EINVAL;  // bf16-only output: plain store
    if (d.c16 && d.split_k > 1) return AC_EINVAL;
    if (d.drop_p < 0.f || d.drop_p >= 1.f) return AC_EINVAL;
    if ((d.drop_p > 0.f || d.mask16) && d.split_k > 1) return AC_EINVAL;
    if (d.dact && !d.aux) return AC_EINVAL;
    if (d.a.rows.r1 != 0 && d.a.rows.r2 == 0) return AC_EINVAL;
    if (d.b.rows.r1 != 0 && d.b.rows.r2 == 0) return AC_EINVAL;
    if (d.c.rows.r1 != 0 && d.c.rows.r2 == 0) return AC_EINVAL;
    if (d.split_k < 1) d.split_k = 1;

    if (d.math == AC_MATH_BF16_IN) {
        // operands are bf16 in memory: strides / offsets in bf16 elements, multiples of 8 (16 B)
        if (batch > 1) return AC_EINVAL;
        if (d.mode == AC_GEMM_NN) return AC_EINVAL;  // transpose B at cast time and use NT
        auto al8 = [](const ac_rowmap &r) { return r.s1 % 8 == 0 && r.s2 % 8 == 0 && r.s3 % 8 == 0; };
        const int ai = d.mode == AC_GEMM_TN ? d.M : d.K, bi = d.mode == AC_GEMM_TN ? d.N : d.K;
        if (!ac_aligned16(d.a.ptr) || !ac_aligned16(d.b.ptr) || !al8(d.a.rows) || !al8(d.b.rows) ||
            (ai % 8) || (bi % 8) || d.force_simple == 1)
            return AC_EALIGN;
        GemmParams p;
        p.d = d;
        if (d.split_k > 1) p.d.accumulate = 2;
        if (d.split_k == 1 && d.accumulate == 2) p.d.accumulate = 1;  // one workgroup per tile: plain +=
        if (d.split_k > 1 && (d.bias || d.pre_out || d.act || d.dact || d.residual)) return AC_EINVAL;
        p.vec_epi = vec_epilogue_ok(p.d, p.d.accumulate);
        p.epi_var = epilogue_variant(p.d, p.d.accumulate);
        // tile shape: force_tile (tests/tuning) or by output shape
        const int tile = d.tile;  // 0 auto, 1 = 128x128, 2 = 256x64, 3 = 256x128 (8 waves)
        if (d.mode == AC_GEMM_TN) {
            if (tile == 3) return launch_bf16in<true, 4, 2>(p, stream);
            if (tile == 4) return launch_bf16in<true, 2, 4>(p, stream);   // 128 x 256, 8 waves
            return launch_bf16in<true, 2, 2>(p, stream);
        }
        if (tile == 2 || (tile == 0 && d.N <= 64 && d.M >= 256)) return launch_bf16in<false, 4, 1>(p, stream);
        // long reductions with enough 256 x 128 tiles to give every CU its own 8-wave workgroup: the
        // larger tile pulls 25 % fewer operand bytes per FLOP through L2 (630 -> 830 TF at 8192^2 x 4096,
        // +5..10 % on the generic conv products); short-K and narrow products keep 2 x (128 x 128) per CU
        const int64_t big_tiles = (int64_t)((d.M + 255) / 256) * ((d.N + 127) / 128);
        if (tile == 3 || (tile == 0 && d.split_k == 1 && d.K >= 1024 && big_tiles >= 256))
            return launch_bf16in<false, 4, 2>(p, stream);
        return launch_bf16in<false, 2, 2>(p, stream);
    }
    const bool a_kc = d.mode != AC_GEMM_TN;
    const bool b_kc = d.mode == AC_GEMM_NT;
    // inner extents must be multiples of 4 so that 16-byte chunks never straddle the edge
    const int a_inner = a_kc ? d.K : d.M;
    const int b_inner = b_kc ? d.K : d.N;
    bool aligned = ac_aligned16(d.a.ptr) && ac_aligned16(d.b.ptr) && rowmap_aligned(d.a.rows) &&
                   rowmap_aligned(d.b.rows) && (a_inner % 4 == 0) && (b_inner % 4 == 0);
    const double macs = (double)d.M * d.N * d.K;
    bool use_mfma = aligned && !d.force_simple && (macs >= 262144.0 || batch > 1);

    GemmParams p;
    p.d = d;
    p.bs_a = bs_a; p.bs_b = bs_b; p.bs_c = bs_c;
    p.ngrp = 0;
    if (grp) {
        if (batch < 1 || batch > GemmParams::MAX_GROUPS || !use_mfma) return AC_EINVAL;
        p.ngrp = batch;
        for (int i = 0; i < 3 * batch; ++i) {
            if (!grp[i] || !ac_aligned16(grp[i])) return AC_EALIGN;
            p.grp[i] = grp[i];
        }
    }
    if (!use_mfma) {
        if (d.accumulate == 3 || batch > 1) return AC_EINVAL;   // split-K slabs / batches exist on the matrix-core kernels only
        // split_k is only a scheduling hint: the scalar kernel computes whole dot products
        // and honours the caller's accumulate mode.
        p.tiles_m = p.tiles_n = p.nkt = p.kt_per_split = p.vec_epi = 0;
        p.epi_var = EPI_GENERIC;
        int64_t total = (int64_t)d.M * d.N;
        if (d.mode == AC_GEMM_TN && d.K >= 128 && total <= 65536) {
            hipLaunchKernelGGL(gemm_simple_tn_wave_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0,
                               stream, p);
            AC_CHECK_LAUNCH();
            return AC_OK;
        }
        int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL(gemm_simple_kernel, dim3(blocks), dim3(256), 0, stream, p);
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    const bool slabs = d.accumulate == 3;   // deterministic split-K: c is [split_k, M, N], reduced by ac_splitk_reduce
    if (slabs && (d.math == AC_MATH_BF16 || d.bias || d.pre_out || d.act || d.dact || d.residual || d.colscale ||
                  d.mask16 || d.c16 || d.drop_p > 0.f || d.c.rows.r1 != 0 || d.c.goff || d.c.rows.s3 != d.N))
        return AC_EINVAL;
    if (d.split_k > 1 && !slabs) p.d.accumulate = 2;
    if (d.split_k == 1 && d.accumulate == 2) p.d.accumulate = 1;  // one workgroup per tile: plain +=
    if (d.split_k > 1 && (d.bias || d.pre_out || d.act || d.dact || d.residual)) return AC_EINVAL;
    p.vec_epi = slabs ? 3 : vec_epilogue_ok(p.d, p.d.accumulate);
    p.epi_var = slabs ? EPI_GENERIC : epilogue_variant(p.d, p.d.accumulate);
    if (d.colsum && (p.vec_epi != 1 || d.split_k != 1 || !d.c.ptr || batch > 1 || grp)) return AC_EINVAL;
    p.tiles_m = (d.M + BM - 1) / BM;
    p.tiles_n = (d.N + BN - 1) / BN;
    p.nkt = (d.K + BK - 1) / BK;
    p.kt_per_split = (p.nkt + d.split_k - 1) / d.split_k;
    // a K piece without tiles returns before it stores: in slab form its slab would stay unwritten and
    // ac_splitk_reduce would sum uninitialised memory - the caller must pass a split_k whose last piece is not empty
    // (split_k = ceil(nkt / ceil(nkt / wanted)); hipops._exact_split)
    if (slabs && (int64_t)(d.split_k - 1) * p.kt_per_split >= p.nkt) return AC_EINVAL;
    dim3 grid(p.tiles_m * p.tiles_n, d.split_k);
    const size_t lds_f32 = 4 * TILE_FLOATS * sizeof(float);  // 64 KB
    const size_t lds_bf16 = 4 * TILE_FLOATS * sizeof(short); // 32 KB
    if (batch > 1 || p.ngrp > 0) {
        // the per-frequency products of the frequency-domain convolutions (ac_fft.hip): plain fp32 matrices, no
        // epilogue beyond store / +=, exact fp32 or split-bf16 arithmetic.  Grouped products (ac_gemm_grouped) may also
        // be cut over K: their pieces meet in C with atomics (weight gradients into the gradient sinks).
        if ((d.split_k != 1 && !(p.ngrp > 0 && p.d.accumulate == 2)) || d.math == AC_MATH_BF16 ||
            (int64_t)batch * grid.x > 0x7FFFFFFF)
            return AC_EINVAL;
        if (d.bias || d.pre_out || d.act || d.dact || d.residual || d.colscale || d.mask16 || d.c16 || d.drop_p > 0.f)
            return AC_EINVAL;
        if ((bs_a % 4) || (bs_b % 4) || (bs_c % 4)) return AC_EALIGN;
        grid.x *= batch;
        if (d.math == AC_MATH_BF16X3) {
            if (d.mode == AC_GEMM_NT) return launch_x3<true, true, true>(p, grid, stream);
            if (d.mode == AC_GEMM_NN) return launch_x3<true, false, true>(p, grid, stream);
            return launch_x3<false, false, true>(p, grid, stream);
        }
        if (d.mode == AC_GEMM_NT)
            hipLaunchKernelGGL((gemm_f32_kernel<true, true, true>), grid, dim3(256), lds_f32, stream, p);
        else if (d.mode == AC_GEMM_NN)
            hipLaunchKernelGGL((gemm_f32_kernel<true, false, true>), grid, dim3(256), lds_f32, stream, p);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<false, false, true>), grid, dim3(256), lds_f32, stream, p);
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    if (d.math == AC_MATH_BF16X3) {
        if (d.b_hi || d.b_lo) {
            // B from cached planes: a plain row-major matrix, 16-byte aligned rows of whole 8-element pieces
            const int inner = d.mode == AC_GEMM_NT ? d.K : d.N;
            if (!d.b_hi || !d.b_lo || d.mode == AC_GEMM_TN || d.b.rows.r1 != 0 || d.b.goff || (d.ld_bpl % 8) ||
                (inner % 8) || d.ld_bpl < inner)
                return AC_EINVAL;
            if (!ac_aligned16(d.b_hi) || !ac_aligned16(d.b_lo)) return AC_EALIGN;
            if (d.mode == AC_GEMM_NT) return launch_x3<true, true, false, true>(p, grid, stream);
            return launch_x3<true, false, false, true>(p, grid, stream);
        }
        if (d.mode == AC_GEMM_NT) return launch_x3<true, true>(p, grid, stream);
        if (d.mode == AC_GEMM_NN) return launch_x3<true, false>(p, grid, stream);
        return launch_x3<false, false>(p, grid, stream);
    }
    if (d.math == AC_MATH_BF16) {
        if (d.mode == AC_GEMM_NT)
            hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(256), lds_bf16, stream, p);
        else if (d.mode == AC_GEMM_NN)
            hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, dim3(256), lds_bf16, stream, p);
        else
            hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(256), lds_bf16, stream, p);
    } else {
        if (d.mode == AC_GEMM_NT)
            hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), lds_f32, stream, p);
        else if (d.mode == AC_GEMM_NN)
            hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), lds_f32, stream, p);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), lds_f32, stream, p);
    }
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_gemm(const ac_gemm_desc *dp, ac_stream_t stream) { return gemm_run(dp, stream, 1, 0, 0, 0); }

extern "C" int ac_gemm_batched(const ac_gemm_desc *dp, int32_t batch, int64_t bs_a, int64_t bs_b, int64_t bs_c,
                               ac_stream_t stream) {
    if (batch < 1) return AC_EINVAL;
    return gemm_run(dp, stream, batch, bs_a, bs_b, bs_c);
}

extern "C" int ac_gemm_grouped(const ac_gemm_desc *dp, int32_t groups, const void *const *ptrs, ac_stream_t stream) {
    if (groups < 1 || groups > GemmParams::MAX_GROUPS || !ptrs) return AC_EINVAL;
    return gemm_run(dp, stream, groups, 0, 0, 0, ptrs);
}

extern "C" int ac_cast_bf16(const float *x, void *y, int64_t n, ac_stream_t stream) {
    if (!x || !y || n < 0) return AC_EINVAL;
    if (n == 0) return AC_OK;
    const int64_t n8 = (ac_aligned16(x) && ac_aligned16(y)) ? n / 8 : 0;
    if (n8 > 0) {
        int64_t g = (n8 + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x,
                           (unsigned short *)y, n8);
    }
    if (8 * n8 < n) {
        const int64_t rem = n - 8 * n8;
        hipLaunchKernelGGL(cast_bf16_tail_kernel, dim3((int)((rem + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x, (unsigned short *)y, 8 * n8, n);
    }
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_transpose_cast_bf16(const float *x, int64_t ldx, void *y, int64_t ldy,
                                      int64_t rows, int32_t cols, ac_stream_t stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return AC_EINVAL;
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    hipLaunchKernelGGL(transpose_cast_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (unsigned short *)y, ldy, rows, cols);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_transpose_cast_segments(const float *src, void *dst, const int64_t *segs,
                                          int32_t nseg, int32_t total_tiles, ac_stream_t stream) {
    if (!src || !dst || !segs || nseg <= 0 || total_tiles <= 0) return AC_EINVAL;
    hipLaunchKernelGGL(transpose_cast_segments_kernel, dim3(total_tiles), dim3(256), 0,
                       (hipStream_t)stream, src, (unsigned short *)dst, segs, nseg);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
