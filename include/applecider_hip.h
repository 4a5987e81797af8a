/*
 * applecider_hip.h — C ABI of libapplecider_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary of the AppleCiDEr forward/backward hot path
 * (SURVEY.md §8b).  The reference (skyportal/applecider) is pure Python and has
 * no FFI of its own: every entry point below replaces the stock ATen op(s) that
 * the cited reference lines dispatch to.  The Python host modules under
 * applecider_amd/models call these through ctypes from
 * torch.autograd.Function.forward/backward.
 *
 * Conventions (all entry points):
 *   - extern "C"; plain pointers and sizes; no torch types.
 *   - every pointer is a caller-owned DEVICE pointer (tensor.data_ptr()) that
 *     must stay alive until the stream work has finished; fp32 unless stated.
 *   - last argument is the hipStream_t to launch on (passed as void*).
 *   - returns 0 on success, a negative AC_E* code on bad arguments, or the
 *     negated hipError_t of a failed launch.  Never allocates, never
 *     synchronises, keeps no global mutable state (re-entrant).
 *   - "ld*" arguments are leading dimensions in ELEMENTS.
 */
#ifndef APPLECIDER_HIP_H
#define APPLECIDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_ABI_VERSION 5

#define AC_OK 0
#define AC_EINVAL (-22)     /* bad argument / unsupported shape            */
#define AC_EALIGN (-1001)   /* pointer / stride alignment requirement unmet */
#define AC_ELAUNCH (-1002)  /* generic launch failure                       */

typedef void *ac_stream_t; /* hipStream_t */

int ac_abi_version(void);
const char *ac_strerror(int code);

/* ------------------------------------------------------------------------
 * Gather-GEMM on the matrix cores.
 *
 * One kernel family carries every dense contraction of the path:
 *   nn.Linear            astrominn.py:23-35,46-58,133-139; HyraxBaselineCLS.py:19,38;
 *                        spectranet.py:138-155
 *   nn.Conv1d (k up to 1021, implicit GEMM over a zero-padded [B,Lp,C] buffer)
 *                        spectranet.py:18-20,25 (C1)
 *   ConvNeXt 1x1 / 2x2-s2 / 4x4-s4 convolutions (timm convnext_tiny, A2)
 *   transformer QKV / out / FFN projections (nn.TransformerEncoderLayer, B2)
 * and their dX / dW backward products.
 *
 * A matrix operand is described by an "outer" index (a row of memory) and an
 * "inner" index (the contiguous direction):
 *     elem(outer o, inner i) = ptr + rowaddr(o) + (goff ? goff[i/32] + i%32 : i)
 *     rowaddr(o) = r1 == 0 ? o*s3
 *                          : (o/r1)*s1 + ((o%r1)/r2)*s2 + (o%r2)*s3
 * which expresses plain strided matrices, overlapping-row (Toeplitz) views of
 * a padded sequence (conv1d), and 2x2 patch gathers (ConvNeXt downsample).
 * All strides and goff entries must be multiples of 4 elements (16 B).
 *
 * mode AC_GEMM_NT : C[m,n] = sum_k A[m,k] * B[n,k]   A outer=m inner=k, B outer=n inner=k
 * mode AC_GEMM_NN : C[m,n] = sum_k A[m,k] * B[k,n]   A outer=m inner=k, B outer=k inner=n
 * mode AC_GEMM_TN : C[m,n] = sum_k A[k,m] * B[k,n]   A outer=k inner=m, B outer=k inner=n
 * C: outer=m inner=n (same addressing scheme, so results can be scattered).
 *
 * Epilogue, per element, in this order:
 *   v = alpha*acc (+ bias[n]); if pre_out: pre_out[m,n] = v; v = act(v);
 *   if dact: v *= act'(aux[m,n]); if mask16: v = mask16[m,n] > 0 ? v : 0;
 *   if colscale: v *= colscale[n];
 *   if drop_p > 0: v = keep(drop_seed, m*N + n) ? v/(1-drop_p) : 0   (ac_dropout's generator);
 *   if residual: v += residual[m,n];
 *   if c16: c16[m,n] = bf16(v)  (row stride ld_c16, columns through c.goff when that is set;
 *           c.ptr may then be NULL: bf16-only output);
 *   accumulate: 0 store, 1 C += v, 2 atomicAdd(C, v)  (2 is forced by split_k > 1), 3 = split-K slabs: C is
 *   [split_k, M, N] and K piece s stores alpha*acc into slab s (no epilogue; see ac_splitk_reduce).
 * mask16 is a bf16 matrix: with act = RELU (and dropout) in the forward product, the forward's bf16
 * output is its own backward mask — alpha = 1/(1-p) then rebuilds dropout's scale
 * (Time2Vec.py:96-101 feed-forward: linear2(dropout(relu(linear1(x))))).
 * ---------------------------------------------------------------------- */
enum { AC_GEMM_NT = 0, AC_GEMM_NN = 1, AC_GEMM_TN = 2 };
enum { AC_ACT_NONE = 0, AC_ACT_GELU = 1, AC_ACT_RELU = 2, AC_ACT_SIGMOID = 3, AC_ACT_TANH = 4,
       /* erf-form GELU through a rational erf (max abs error 4.5e-7) instead of the library erff: what the split-bf16
          and bf16 math modes use (their products carry 2^-16 / 2^-8 of rounding; the library routine is 2-3x the VALU
          instructions, and the row / epilogue kernels around a GELU are bound by VALU issue).  Accepted wherever
          AC_ACT_GELU is; GEMM epilogues choose it by ac_gemm_desc.math. */
       AC_ACT_GELU_FAST = 5 };
/* dact: derivative taken from aux.  GELU/RELU expect the PRE-activation in aux,
 * SIGMOID/TANH expect the activation OUTPUT in aux. */
/* MFMA input type.  F32 / BF16: operands are fp32 in memory (BF16 rounds them while staging).
 * BF16_IN: A and B are already bf16 in memory (ac_cast_bf16 / ac_transpose_cast_bf16); strides,
 * goff entries and inner extents are in bf16 elements and multiples of 8; modes NT and TN only.
 * BF16X3: fp32 operands in memory like F32; each element is split while staging into hi = bf16(x)
 * and lo = bf16(x - hi), and each product is three bf16 matrix-core instructions (hi*hi + hi*lo +
 * lo*hi, fp32 accumulate): ~2^-16 relative per product — the mode that meets the path's 1e-3 logit
 * parity bar at bf16 matrix-core speed.
 * C, bias, aux, residual stay fp32 in every mode. */
enum { AC_MATH_F32 = 0, AC_MATH_BF16 = 1, AC_MATH_BF16_IN = 2, AC_MATH_BF16X3 = 3 };

typedef struct ac_rowmap {
    int32_t r1, r2;
    int64_t s1, s2, s3;
} ac_rowmap;

typedef struct ac_mat {
    const void *ptr;
    ac_rowmap rows;
    const int32_t *goff; /* nullable; one entry per 32 inner elements */
} ac_mat;

typedef struct ac_gemm_desc {
    int32_t mode;
    int32_t math;
    int32_t M, N, K;
    int32_t act;
    int32_t dact;
    int32_t accumulate;
    int32_t split_k;      /* >= 1 */
    int32_t force_simple; /* 1: use the scalar reference kernel (tests) */
    float alpha;
    int32_t tile;         /* BF16_IN only: 0 auto, 1 = 128x128, 2 = 256x64, 3 = 256x128 tile */
    ac_mat a, b, c;
    const float *bias;
    float *pre_out;
    int64_t ld_pre;
    const float *aux;
    int64_t ld_aux;
    const float *colscale;
    const float *residual;
    int64_t ld_res;
    void *c16;            /* nullable: bf16 copy of the output, row-major */
    int64_t ld_c16;
    const void *mask16;   /* nullable: bf16 matrix, v is zeroed where mask16[m,n] <= 0 */
    int64_t ld_mask16;
    float drop_p;         /* 0: no dropout */
    uint64_t drop_seed;
    const uint64_t *drop_step; /* nullable DEVICE counter mixed into drop_seed (see ac_step_advance) */
    /* AC_MATH_BF16X3, NT / NN, one product (not batched / grouped), nullable: the B operand as CACHED (hi, lo) bf16
       planes (ac_split_bf16 of the same matrix, e.g. a weight split once per optimizer step) - plain row-major with
       row pitch ld_bpl elements (% 8 == 0, rows 16-byte aligned, inner extent % 8 == 0).  The kernel then reads B
       from the planes (b.ptr is ignored): no split arithmetic per read, half the operand bytes. */
    const void *b_hi, *b_lo;
    int64_t ld_bpl;
    /* nullable: the column sums of the fp32 values this product stores (after its whole epilogue) are ADDED to
       colsum[0 .. N) with atomics.  With dact (+ drop_p: the forward epilogue's mask, same seed and index) the product
       that forms a hidden layer's gradient applies that layer's activation backward itself and leaves the layer's bias
       gradient here - no separate pass over the hidden tensor.  Needs the 16-byte epilogue (N % 4 == 0, aligned
       pointers), one K piece, one product. */
    float *colsum;
} ac_gemm_desc;

int ac_gemm(const ac_gemm_desc *d, ac_stream_t stream);
/* y = bf16(x) elementwise (round to nearest even); y is a uint16 buffer. */
int ac_cast_bf16(const float *x, void *y, int64_t n, ac_stream_t stream);
/* y[c, r] = bf16(x[r, c]) : the k-contiguous bf16 copy of an operand stored row-contiguous. */
/* Split-bf16 planes for math mode bf16x3: hi = bf16(x), lo = bf16(x - hi) (round to nearest even).
 * ac_split_bf16: elementwise; ac_transpose_split_bf16: [rows, cols] -> two [cols, rows] planes;
 * ac_pad_rows_split: the zero-padded [B, Lp, C] planes of x [B, L, C] (C % 8 == 0). */
int ac_split_bf16(const float *x, void *hi, void *lo, int64_t n, ac_stream_t stream);
int ac_transpose_split_bf16(const float *x, int64_t ldx, void *hi, void *lo, int64_t ldy, int64_t rows,
                            int32_t cols, ac_stream_t stream);
int ac_pad_rows_split(const float *x, void *hi, void *lo, int32_t B, int32_t L, int32_t C, int32_t pad_lo,
                      int32_t Lp, ac_stream_t stream);
int ac_transpose_cast_bf16(const float *x, int64_t ldx, void *y, int64_t ldy, int64_t rows,
                           int32_t cols, ac_stream_t stream);
/* Batched form over a flat parameter buffer: segs (device, 4 x int64 per segment) =
 * {element offset, rows, cols, index of the segment's first 64x64 tile}; for every segment
 * dst[offset + c*rows + r] = bf16(src[offset + r*cols + c]).  One launch per optimizer step. */
int ac_transpose_cast_segments(const float *src, void *dst, const int64_t *segs, int32_t nseg,
                               int32_t total_tiles, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Conv1d with an LDS-resident input window (bf16 operands, fp32 output) — the fast path of the
 * SpectraNetBlock conv bank (spectranet.py:18-20,25) for both the forward product and the
 * input-gradient product:
 *   out[b, l, n] (+)= bias[n] + sum_{t<k} sum_{c<C} A[b, row_base + l + t, a_col_off + c]
 *                                                  * W[n, tap(t), c],   tap(t) = flip ? k-1-t : t
 * A: bf16, element (b, row, col) at a + b*a_batch_stride + row*a_row_stride + col (zero-padded rows).
 * W: bf16, element (n, tap, c) at w + n*w_row_stride + tap*w_tap_stride + c.
 * Requirements: C in {64,128,256}; L % 128 == 0; strides/offsets multiples of 8 elements;
 * (tile_rows + k - 1)*C*2 B + weight stages <= 160 KB (else AC_EINVAL: use ac_gemm).
 * ---------------------------------------------------------------------- */
typedef struct ac_convwin_desc {
    const void *a;
    int64_t a_batch_stride, a_row_stride;
    int32_t a_col_off, row_base;
    int32_t B, L, C, k;
    const void *w;
    int64_t w_row_stride, w_tap_stride;
    int32_t flip, N;
    float *c;
    int64_t ldc;
    const float *bias;
    int32_t accumulate; /* 0 store, 1 out += */
    int32_t variant;    /* 0 auto.  ac_conv1d_window_bf16: 1 = never use the 8-wave two-group kernel for N <= 64.
                           ac_conv1d_window_x3: 4 = the round-2 kernel (register-staged weight stages) instead of
                           the LDS-DMA ring kernel, 2 / 3 = its 32x32x16 / one-group forms, 6 = ring of five
                           (A/B measurements, tests) */
    void *c16;          /* nullable: bf16 output (element (b, l, n) at c16 + (b*L + l)*ldc16 + n);
                           c may then be NULL (bf16-only output, accumulate must be 0) */
    int64_t ldc16;
    int64_t a_lo_off, w_lo_off; /* ac_conv1d_window_x3 only: element offsets from the hi plane of A / W
                                   to its lo plane (split-bf16 operands, math mode bf16x3) */
    /* ac_conv1d_window_x3, ring kernel only (0 = plain conv).  tap_row_step > 1: tap t reads window rows
     * (l + tap_row_step * t); with a_row_stride = 8, C = 64 and tap_row_step = 8 the "window rows" are overlapping
     * 64-element views of ONE sequence and the product is the Toeplitz form of a Cin = 1 convolution (SpectraNet
     * stage 1, spectranet.py:18-20 with in_channels = 1): row = 8 consecutive output positions, column n =
     * (position p, channel co), K = the k + 7 window in 64-element "taps".  c_block > 0: output column n lands at
     * c[row * ldc + (n / c_block) * c_block_stride + n % c_block] — the 8 positions of a Toeplitz row are 8
     * rows of the channels-last output (c_block = Cout, c_block_stride = its row pitch, ldc = 8 * that). */
    int32_t tap_row_step, c_block;
    int64_t c_block_stride;
} ac_convwin_desc;
int ac_conv1d_window_bf16(const ac_convwin_desc *d, ac_stream_t stream);
/* The same product on split-bf16 operands in one launch: 3 MFMAs per fragment pair, the K loop cut
 * into chunks of 64 channels x TC taps whose two window planes fit the LDS (any C % 64 == 0, L % 128
 * == 0, fp32 output only: c16 must be NULL).  AC_EINVAL: shape not covered. */
int ac_conv1d_window_x3(const ac_convwin_desc *d, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Row-wise LayerNorm over the last dimension (nn.LayerNorm: astrominn.py:25,34,47,52;
 * spectranet.py:21,31; HyraxBaselineCLS.py:34 and the two norms inside each
 * nn.TransformerEncoderLayer; timm ConvNeXt block norm / LayerNorm2d in NHWC).
 * act = AC_ACT_GELU fuses the GELU of spectranet.py:35 (y = gelu(ln(x))).
 * bwd accumulates dgamma/dbeta with atomics: zero them first.  dxsum (nullable, C % 4 == 0 and
 * C <= 3072 only) additionally accumulates the column sums of dx, i.e. the bias gradient of the
 * conv/linear that produced x, so that no separate pass over dx is needed.
 * ---------------------------------------------------------------------- */
int ac_layernorm_fwd(const float *x, int64_t ldx, const float *gamma, const float *beta,
                     float *y, int64_t ldy, float *mean, float *rstd, int64_t rows, int32_t C,
                     float eps, int32_t act, void *y16, int64_t ldy16, int32_t x_bf16,
                     ac_stream_t stream);
/* y16 (nullable): bf16 copy of the output for the matrix product that follows (y may then be NULL);
 * available when C = 4*G*J with G in {8,16,32,64}, J in {1,2,3,6} (AC_EALIGN otherwise).
 * x_bf16 != 0 (both directions, same C restriction): x points at a bf16 matrix (ldx in elements) —
 * the conv-bank output kept in bf16 between the Conv1d products and LayerNorm. */
int ac_layernorm_bwd(const float *dy, int64_t lddy, const float *x, int64_t ldx,
                     const float *mean, const float *rstd, const float *gamma, const float *beta,
                     float *dx, int64_t lddx, float *dgamma, float *dbeta, float *dxsum,
                     int64_t rows, int32_t C, int32_t act, void *dx16, int64_t lddx16,
                     int32_t seg_len, int32_t seg_pitch, int32_t seg_off, int32_t dy_bf16,
                     int32_t x_bf16, ac_stream_t stream);
/* The same with a second 16-bit output: dx16_lo (nullable) receives bf16(dx - dx16), the lo plane of the split-bf16
 * operand pair (math mode bf16x3) in the layout of dx16 — LayerNorm's backward then writes the (hi, lo) planes
 * of the Conv1d gradient products itself and no fp32 dx (dx = NULL) nor a separate split pass is needed. */
int ac_layernorm_bwd_split(const float *dy, int64_t lddy, const float *x, int64_t ldx,
                           const float *mean, const float *rstd, const float *gamma, const float *beta,
                           float *dx, int64_t lddx, float *dgamma, float *dbeta, float *dxsum,
                           int64_t rows, int32_t C, int32_t act, void *dx16, void *dx16_lo, int64_t lddx16,
                           int32_t seg_len, int32_t seg_pitch, int32_t seg_off, int32_t dy_bf16,
                           int32_t x_bf16, ac_stream_t stream);
/* dx16 (nullable, same C restriction as y16; dx may then be NULL): bf16 copy of dx.  With
 * seg_len > 0 row r = (b, l), l < seg_len, is written to row b*seg_pitch + seg_off + l of dx16 —
 * the zero-padded [B, Lp, C] operand of the Conv1d gradient products (pads are the caller's).
 * dy_bf16 != 0: dy points at a bf16 matrix (lddy in elements) — the input gradient of a 1x1 conv
 * written by ac_gemm's c16 output; same C restriction. */

/* ------------------------------------------------------------------------
 * Elementwise / reduction helpers.
 * ---------------------------------------------------------------------- */
/* out[n] (+)= sum_m x[m,n]  — bias gradients. */
int ac_colsum(const float *x, int64_t ldx, float *out, int64_t rows, int32_t cols,
              int32_t accumulate, ac_stream_t stream);
/* g = dropmask * dy * act'(aux) elementwise AND out[c] (+)= sum_r g[r, c] in one pass over [rows, cols] (row
 * stride ld for all three): the backward entry of an activated Linear (astrominn.py:19-28 heads, timm ConvNeXt
 * fc1 + GELU, nn.TransformerEncoderLayer linear1 + ReLU + dropout, linear2 + dropout) needs both.  aux as in
 * ac_act_bwd (pre-activation for GELU / ReLU, output for sigmoid / tanh; NULL with act none).  drop_p > 0:
 * the mask ac_gemm's epilogue drew in the forward product (same seed / step counter, element index
 * m * cols + n, ld == cols), scaled by 1/(1-p).  cols and ld even, pointers 8-byte aligned. */
int ac_act_bwd_colsum(const float *dy, const float *aux, float *g, int64_t ld, float *out, int64_t rows,
                      int32_t cols, int32_t act, int32_t accumulate, float drop_p, uint64_t drop_seed,
                      const uint64_t *step, ac_stream_t stream);
/* y16 = bf16(x) and out[n] (+)= sum_m x[m,n] in one pass (cols, ldx, ldy even). */
int ac_cast_bf16_colsum(const float *x, int64_t ldx, void *y16, int64_t ldy, float *out,
                        int64_t rows, int32_t cols, int32_t accumulate, ac_stream_t stream);
/* The same for a bf16 matrix (cols and ldx even): bias gradient of a bf16 hidden gradient. */
int ac_colsum_bf16(const void *x, int64_t ldx, float *out, int64_t rows, int32_t cols,
                   int32_t accumulate, ac_stream_t stream);
/* out = dy * act'(aux)  (kinds as in ac_gemm dact); in-place allowed. */
int ac_act_bwd(const float *dy, const float *aux, float *out, int64_t n, int32_t kind,
               ac_stream_t stream);
/* y = act(x) elementwise. */
int ac_act_fwd(const float *x, float *y, int64_t n, int32_t kind, ac_stream_t stream);
/* 2-D strided copy: dst[r, c] = src[r, c]. */
int ac_copy2d(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows,
              int32_t cols, ac_stream_t stream);
/* y[r, c] = pre[r, c] * colscale[c] + residual[r, c] on contiguous [rows, cols] matrices (colscale / residual
 * nullable; cols % 4 == 0): the layer-scale + skip epilogue of a ConvNeXt block's fc2 (timm convnext_tiny,
 * astrominn.py:12-17) when that product ran split over K with atomics and could not carry it itself. */
int ac_scale_add_rows(const float *pre, const float *colscale, const float *residual, float *y, int64_t rows,
                      int32_t cols, ac_stream_t stream);
/* Deterministic split-K: ac_gemm with accumulate = 3 and split_k = S (fp32 / split-bf16 math, plain [M, N] output,
 * no epilogue) stores the partial product of K piece s into slab s of c = [S, M, N]; this pass sums the slabs in
 * index order and applies the epilogue the product could not:
 *     v = sum_s part[s] (+ bias[c]);  pre_out = v (nullable);  y = v (* colscale[c]) (+ residual).
 * Used for forward products whose output grid leaves the chip idle (cols % 4 == 0). */
int ac_splitk_reduce(const float *part, int32_t split, const float *bias, float *pre_out, const float *colscale,
                     const float *residual, float *y, int64_t rows, int32_t cols, ac_stream_t stream);
/* dst[r, j] = src[r, idx[j]]  — metadata column gathers, astrominn.py:249-261. */
int ac_gather_cols(const float *src, int64_t lds, const int32_t *idx, float *dst, int64_t ldd,
                   int64_t rows, int32_t ncols, ac_stream_t stream);
/* out = a*g + s (s nullable)  — ResidualTowerBlock combine astrominn.py:62, head product :41. */
int ac_gate_fwd(const float *a, const float *g, const float *s, float *out, int64_t n,
                ac_stream_t stream);
/* da = dout*g, dg = dout*a. */
int ac_gate_bwd(const float *dout, const float *a, const float *g, float *da, float *dg,
                int64_t n, ac_stream_t stream);
/* y = x * keep(seed', offset + i) / (1-p); the same call with dy gives dx (nn.Dropout).
 * seed' = seed when step is NULL, else seed + step[0] * 0x9E3779B97F4A7C15 (ac_step_advance). */
int ac_dropout(const float *x, float *y, int64_t n, float p, uint64_t seed, uint64_t offset,
               const uint64_t *step, ac_stream_t stream);
/* ConvNeXt layer-scale backward: dyl = dy*gamma[n] (fp32 and/or bf16 copy, either nullable);
 * dgamma[n] += sum_m dy*ylin; dbias[n] (nullable) += sum_m dyl = the bias gradient of the linear
 * layer under the scale (atomics: zero dgamma / dbias first). */
int ac_layerscale_bwd(const float *dy, const float *ylin, const float *gamma, float *dyl,
                      void *dyl16, float *dgamma, float *dbias, int64_t rows, int32_t C,
                      ac_stream_t stream);
/* y = (a + b) * alpha. */
int ac_add(const float *a, const float *b, float *y, int64_t n, float alpha, ac_stream_t stream);
/* x *= s[0] * scale  (device scalar; gradient clipping). */
int ac_scale_by_dev(float *x, int64_t n, const float *s, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Image branch (timm convnext_tiny called at astrominn.py:12-17).
 * Activations are NHWC fp32: [B, H, W, C].
 * ---------------------------------------------------------------------- */
/* NCHW image -> 4x4 stride-4 patches [B*OH*OW, 64] ((ky,kx,c) order, 48 used, zero padded);
 * OH = (H-4)/4+1.  The stem conv then is a plain GEMM. */
int ac_stem_patchify(const float *img, float *patches, int32_t B, int32_t Cin, int32_t H,
                     int32_t W, ac_stream_t stream);
/* depthwise 7x7, pad 3.  w is [49, C] (tap-major), bias [C]. */
int ac_dwconv7x7_fwd(const float *x, const float *w, const float *bias, float *y, int32_t B,
                     int32_t H, int32_t W, int32_t C, ac_stream_t stream);
/* dx = corr(dy, w); dw[49,C] += ..., dbias[C] += ... (atomics; zero first). */
int ac_dwconv7x7_bwd(const float *dy, const float *x, const float *w, float *dx, float *dw,
                     float *dbias, int32_t B, int32_t H, int32_t W, int32_t C,
                     ac_stream_t stream);
/* The same two entry points with the kernel family selectable (A/B measurements, tests): variant 0 = automatic
 * (what the plain entry points do), 1 = the one-item-per-workgroup kernels of round 2 (no LDS-DMA pipeline). */
int ac_dwconv7x7_fwd_v(const float *x, const float *w, const float *bias, float *y, int32_t B, int32_t H,
                       int32_t W, int32_t C, int32_t variant, ac_stream_t stream);
int ac_dwconv7x7_bwd_v(const float *dy, const float *x, const float *w, float *dx, float *dw, float *dbias,
                       int32_t B, int32_t H, int32_t W, int32_t C, int32_t variant, ac_stream_t stream);
/* ... and with the ConvNeXt block's shortcut folded in (timm block at astrominn.py:12-17: out = x + gamma * mlp(..dw(x))):
 * dx = depthwise backward(dy) + dres, dres [B, H, W, C] nullable - the block input's two gradient contributions meet
 * where dx is stored instead of in a separate elementwise pass. */
int ac_dwconv7x7_bwd_res(const float *dy, const float *x, const float *w, const float *dres, float *dx, float *dw,
                         float *dbias, int32_t B, int32_t H, int32_t W, int32_t C, int32_t variant,
                         ac_stream_t stream);
/* mean over the HW positions: [B, HW, C] -> [B, C]; bwd broadcasts dy/HW. */
int ac_avgpool_fwd(const float *x, float *y, int32_t B, int32_t HW, int32_t C, ac_stream_t stream);
int ac_avgpool_bwd(const float *dy, float *dx, int32_t B, int32_t HW, int32_t C,
                   ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Spectra branch (spectranet.py:7-41, 163-170).  Sequences are [B, L, C].
 * ---------------------------------------------------------------------- */
/* MaxPool1d(4) along L; writes y[b, l/4, c] at y + b*y_bstride + (l/4)*C (so it can
 * land inside the next stage's zero-padded buffer); idx records the argmax (0..3). */
int ac_maxpool4_fwd(const float *x, float *y, int64_t y_bstride, uint8_t *idx, int32_t B,
                    int32_t L, int32_t C, ac_stream_t stream);
int ac_maxpool4_bwd(const float *dy, int64_t dy_bstride, const uint8_t *idx, float *dx,
                    int32_t B, int32_t L, int32_t C, ac_stream_t stream);
/* adaptive_max_pool1d(x, 1): [B, L, C] -> [B, C] with argmax. */
int ac_globalmax_fwd(const float *x, float *y, int32_t *idx, int32_t B, int32_t L, int32_t C,
                     ac_stream_t stream);
int ac_globalmax_bwd(const float *dy, const int32_t *idx, float *dx, int32_t B, int32_t L,
                     int32_t C, ac_stream_t stream);
/* flux [B, L] -> zero-padded [B, Lp] at offset pad (stage-1 Toeplitz operand). */
int ac_pad_rows(const float *x, float *y, int32_t B, int32_t L, int32_t C, int32_t pad_lo,
                int32_t Lp, ac_stream_t stream);
/* same with a bf16 destination (operand of the bf16 matrix-core kernels). */
int ac_pad_rows_bf16(const float *x, void *y, int32_t B, int32_t L, int32_t C, int32_t pad_lo,
                     int32_t Lp, ac_stream_t stream);
/* stage-1 (Cin = 1) weight expansion for the 8-phase Toeplitz GEMM:
 *   w[Cout, k] -> wexp[8*Cout, Kp],  wexp[(r,co), t'] = w[co, t' - r - shift]  (0 elsewhere)
 * and the matching gradient fold dw[co,t] = sum_r dwexp[(r,co), t + r + shift].
 * Needs Kp >= k + 7 + shift. */
int ac_toeplitz_expand(const float *w, float *wexp, int32_t Cout, int32_t k, int32_t Kp,
                       int32_t shift, ac_stream_t stream);
int ac_toeplitz_fold(const float *dwexp, float *dw, int32_t Cout, int32_t k, int32_t Kp,
                     int32_t shift, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Photometry branch (HyraxBaselineCLS.py:49-86, Time2Vec.py:48-72).
 * ---------------------------------------------------------------------- */
/* h[b,0,:] = cls; h[b,1+l,:] = W x[b,l,:] + bias + time2vec(x[b,l,0]).
 * x is [B, L, 8] (7 channels + one zero pad), W is [D, 8], tw/tb are [D]
 * (tw[0]=w0, tb[0]=b0, tw[1:]=w, tb[1:]=b). */
int ac_embed_fwd(const float *x, const float *W, const float *bias, const float *tw,
                 const float *tb, const float *cls, float *h, int32_t B, int32_t L, int32_t D,
                 ac_stream_t stream);
/* dW[D,8], dbias, dtw, dtb, dcls accumulate with atomics (zero first). */
int ac_embed_bwd(const float *dh, const float *x, const float *tw, const float *tb, float *dW,
                 float *dbias, float *dtw, float *dtb, float *dcls, int32_t B, int32_t L,
                 int32_t D, ac_stream_t stream);
/* Multi-head self-attention core with key-padding mask (True/1 = ignore key).
 * qkv is [B, T, 3*D] (q | k | v, heads contiguous inside each D), out is [B, T, D].
 * lse [B, H, T] is saved for backward.  p_drop > 0 applies dropout to the
 * attention probabilities with the counter RNG. */
int ac_mha_fwd(const float *qkv, const uint8_t *pad, float *out, float *lse, int32_t B,
               int32_t T, int32_t H, int32_t Dh, float p_drop, uint64_t seed,
               const uint64_t *step, ac_stream_t stream);
int ac_mha_bwd(const float *dout, const float *qkv, const uint8_t *pad, const float *out,
               const float *lse, float *dqkv, int32_t B, int32_t T, int32_t H, int32_t Dh,
               float p_drop, uint64_t seed, const uint64_t *step, ac_stream_t stream);

/* BatchNorm1d over the columns of a [rows, cols] channels-last tensor (rows = B*L positions) fused with
 * an activation: SpectraNetBlock with use_ln = False (spectranet.py:21,33-37).  training != 0: batch
 * statistics (biased variance), running statistics updated with `momentum` (unbiased variance) when the
 * pointers are given; training == 0: running statistics.  stats = 4*cols floats {mean, rstd, scale,
 * shift} written by the forward and read by the backward; sums (2*cols, training only) and the first
 * 2*cols floats of work (5*cols) must be zero on entry.  dgamma / dbeta are accumulated into. */
int ac_batchnorm_fwd(const float *x, int64_t ld, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, float *y, int64_t ldy, float *stats, float *sums, int64_t rows,
                     int32_t cols, float eps, float momentum, int32_t training, int32_t act, ac_stream_t stream);
int ac_batchnorm_bwd(const float *dy, int64_t lddy, const float *x, int64_t ld, const float *gamma,
                     const float *stats, float *dx, int64_t lddx, float *dgamma, float *dbeta, float *work,
                     int64_t rows, int32_t cols, int32_t training, int32_t act, ac_stream_t stream);

/* Weight gradient of a 'same' Conv1d with the input window resident in LDS (ac_wgrad.hip):
 *   dw[co, t*Cin + ci] += sum_{b, l} dy[b, dy_row_base + l, dy_col_off + co] * x[b, x_row_base + l + t, ci]
 * dy and x are bf16 ([B, rows, row_stride] views, strides / offsets in elements, multiples of 8);
 * L % 64 == 0, Cout % 128 == 0, Cin % 64 == 0 (AC_EINVAL otherwise: use the generic TN ac_gemm).
 * dy_lo_off / x_lo_off != 0: element offsets from the hi plane to the lo plane of split-bf16 operands
 * (math mode bf16x3).  dw is fp32 [Cout, ldw], accumulated with atomics (zero or pre-filled by caller).
 * Replaces the weight-gradient half of torch's conv1d backward for spectranet.py:18-20. */
typedef struct ac_wgrad_desc {
    const void *dy;
    int64_t dy_batch_stride, dy_row_stride;
    int32_t dy_row_base, dy_col_off;
    const void *x;
    int64_t x_batch_stride, x_row_stride;
    int32_t x_row_base, x_rows;   /* x_rows: rows per batch of the padded input (loads are clamped) */
    int32_t B, L, Cout, Cin, k;
    int32_t split_k;              /* workgroups along the B*L reduction */
    float *dw;
    int64_t ldw;
    int64_t dy_lo_off, x_lo_off;
    int32_t variant;              /* 0: v_mfma_f32_16x16x32 tiles (default); 2: the 32x32x16 form (A/B tests) */
    /* Toeplitz form (0 = plain): the weight gradient of a Cin = 1 convolution (SpectraNet stage 1) as
     * dWexp[(p, co), kk] = sum_m dy[m, (p, co)] * x[8 m + kk], m = groups of 8 output positions.  tap_row_step = 8
     * with x_row_stride = 8 and Cin = 64: "input row" r is the 64-element view x[8 r ..], tap t reads rows
     * (m + 8 t), i.e. kk = 64 t + ci; dw = dWexp with ldw = its row length (k * 64).  dy_block > 0: column c of a
     * dy row lives at (c / dy_block) * dy_block_stride + c % dy_block (the 8 positions of a group are 8 rows of
     * the channels-last gradient; dy_block = Cout of the conv, dy_block % 8 == 0).  Split-bf16 operands only. */
    int32_t tap_row_step, dy_block;
    int64_t dy_block_stride;
} ac_wgrad_desc;
int ac_conv1d_wgrad_bf16(const ac_wgrad_desc *d, ac_stream_t stream);

/* The same attention on the matrix cores (d_head = 16, T <= 288; v_mfma_f32_32x32x16_bf16 tiles for
 * Q K^T, P V and the three backward contractions; same lse / dropout-index contract as above, so the
 * two forward / backward pairs are interchangeable).  split = 0: bf16 operands (math mode bf16);
 * split = 1: (hi, lo) bf16 halves, 3 MFMAs per product (math mode bf16x3).  The exact-fp32 mode keeps
 * ac_mha_fwd / ac_mha_bwd.  Replaces nn.MultiheadAttention inside nn.TransformerEncoderLayer
 * (HyraxBaselineCLS.py:24-31,73-79). */
int ac_mha_fwd_mfma(const float *qkv, const uint8_t *pad, float *out, float *lse, int32_t B, int32_t T,
                    int32_t H, int32_t Dh, float p_drop, uint64_t seed, const uint64_t *step, int32_t split,
                    ac_stream_t stream);
int ac_mha_bwd_mfma(const float *dout, const float *qkv, const uint8_t *pad, const float *out,
                    const float *lse, float *dqkv, int32_t B, int32_t T, int32_t H, int32_t Dh,
                    float p_drop, uint64_t seed, const uint64_t *step, int32_t split, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Towers / MoE / fusion head (astrominn.py:264-295; _archive core/model.py:40-67).
 * ---------------------------------------------------------------------- */
/* scores [B,E] (sigmoid outputs), expert_out [E,B,C] -> out [B,C]; sel [B,2] int32. */
int ac_moe_top2_fwd(const float *scores, const float *expert_out, float *out, int32_t *sel,
                    int32_t B, int32_t E, int32_t C, ac_stream_t stream);
int ac_moe_top2_bwd(const float *dout, const float *scores, const float *expert_out,
                    const int32_t *sel, float *dscores, float *dexpert_out, int32_t B,
                    int32_t E, int32_t C, ac_stream_t stream);
/* Grouped, fully fused ResidualTowerBlock (astrominn.py:44-64): up to 8 blocks per launch — the eight
 * metadata towers (astrominn.py:94-113 on the column subsets of :249-261) or the four fusion experts
 * (:129-131) — forward and backward.  Per block and sample:
 *   h    = GELU(W1 x + b1)                    x = metadata[:, cols] when gather != 0, else x[:, 0..n_in)
 *   xhat = (h - mean(h)) * rsqrt(var(h) + eps)            (both LayerNorms normalise the same h)
 *   y    = (Wm drop(xhat*lnm_g + lnm_b) + bm) * sigmoid(Wg drop(xhat*lng_g + lng_b) + bg) + skip(x)
 *   skip = Ws x + bs, or x when ws == NULL (n_in == n_out)
 * y is written at y + sample*ldy (the caller points it into the concatenated feature buffer / the
 * stacked expert outputs).  save [B, hid + 2*n_out + 2] keeps (pre-GELU hidden | main | gate | mean, rstd)
 * for the backward call, which recomputes the rest, adds every parameter gradient to d* with fp32
 * atomics (buffers zeroed or pre-filled by the caller) and, when dx != NULL, adds dL/dx into
 * dx[sample*lddx + column] (atomics too: the four experts share one input).
 * Dropout (p_drop, training != 0): keep decisions from (seed', group_id, path, sample, hidden index) with
 * seed' as in ac_dropout; forward and backward must be given the same seed / step counter value.
 * Limits: n_in <= 288 (<= 24 with gather), hid <= 128, n_out <= 32, n <= 8.  Exact fp32 FMA arithmetic. */
typedef struct ac_tower_desc {
    const float *x;
    const float *w1, *b1;                                /* [hid, n_in], [hid] */
    const float *lnm_g, *lnm_b, *lng_g, *lng_b;          /* LayerNorm affine of the main / gate path, [hid] */
    const float *wm, *bm, *wg, *bg;                      /* [n_out, hid], [n_out] */
    const float *ws, *bs;                                /* skip Linear [n_out, n_in], [n_out]; NULL = identity */
    float *y;
    float *save;
    const float *dy;                                     /* backward only from here */
    float *dx;
    float *dw1, *db1, *dlnm_g, *dlnm_b, *dlng_g, *dlng_b, *dwm, *dbm, *dwg, *dbg, *dws, *dbs;
    int64_t ldx, ldy, lddy, lddx;
    int32_t n_in, hid, n_out;
    int32_t gather;                                      /* 1: x columns are cols[0..n_in) */
    int32_t group_id;                                    /* distinguishes the blocks' dropout streams */
    float eps;
    uint8_t cols[24];
} ac_tower_desc;
int ac_tower_blocks_fwd(const ac_tower_desc *groups_host, int32_t n, int32_t B, float p_drop, int32_t training,
                        uint64_t seed, const uint64_t *step, ac_stream_t stream);
int ac_tower_blocks_bwd(const ac_tower_desc *groups_host, int32_t n, int32_t B, float p_drop, int32_t training,
                        uint64_t seed, const uint64_t *step, ac_stream_t stream);
/* y = x / ||x||_2 per row. */
int ac_l2norm_fwd(const float *x, float *y, float *norm, int64_t rows, int32_t C,
                  ac_stream_t stream);
int ac_l2norm_bwd(const float *dy, const float *y, const float *norm, float *dx, int64_t rows,
                  int32_t C, ac_stream_t stream);
/* softmax over the last dim (use_probabilities, astrominn.py:297). */
int ac_softmax_fwd(const float *x, float *y, int64_t rows, int32_t C, ac_stream_t stream);
/* dst_j[i] += src[j * seg_len + i] for j < nseg <= 4 (unused dst pointers may be NULL): one launch hands the
 * concatenated column sums of a conv bank's backward pass to the bias gradients of its convolutions. */
int ac_add_segments(const float *src, float *dst0, float *dst1, float *dst2, float *dst3, int32_t seg_len,
                    int32_t nseg, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Losses: one kernel computes the mean loss AND dlogits (= dloss/dlogits for mean).
 *   kind 0: CrossEntropyLoss with class-probability targets  (astrominn.py:147,315)
 *   kind 1: CrossEntropyLoss with int64 class indices         (brew_cider.py:1229)
 *   kind 2: FocalLoss(gamma, alpha, eps)                      (HyraxBaselineCLS.py:169-191)
 *   kind 3: MSELoss, float targets [B, C], mean over B * C    (SpectraNet redshift regression, spectranet.py:178-179;
 *           "logits" are the predictions)
 * loss is a single device float, zeroed by the call.
 * ---------------------------------------------------------------------- */
int ac_loss_fwd_bwd(const float *logits, const void *target, const float *alpha, float *loss,
                    float *dlogits, int32_t B, int32_t C, int32_t kind, float gamma, float eps,
                    ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Photometry collate on the device (photo_dataset.py:117-152, Time2Vec.py:18-45, HyraxBaselineCLS.py:152-166): the
 * ragged light curves arrive back to back, flat [sum of lengths, 7], sample b = rows offsets[b] .. offsets[b] + lens[b];
 * every sample is padded with zeros / truncated to L rows, out [B, L, 7]; mask [B, L] = 1 on padding.  normalise != 0:
 * channels 0..3 become (x - mean4[c]) / (std4[c] + 1e-8) on EVERY row, padding included - the reference standardises
 * the padded array in place - with IEEE division: bit-identical to numpy's float32 result.
 * ---------------------------------------------------------------------- */
int ac_collate_photometry(const float *flat, const int64_t *offsets, const int32_t *lens, const float *mean4,
                          const float *std4, float *out, uint8_t *mask, int32_t B, int32_t L, int32_t normalise,
                          ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Masked pre-training step of MPTModel (HyraxBaselineCLS.py:226-319).
 * ac_mpt_mask: _mask_batch (:286-319) on the device.  x [B, L, 7] (in place: channels 2..6 of the
 * selected tokens become 0), pad [B, L] (1 = padding), masked [B, L] out.  Per light curve
 * k = max(int(n_valid * mask_p), 3) tokens: k/3 from each band (argmax of channels 4..6), the
 * remainder from the valid tokens still unselected; draws come from the counter hash of `seed`.
 * ac_mpt_loss_fwd_bwd: loss = lambda_f*MSE(f_hat, x[...,2]) * lambda_b*CE(b_hat, argmax x[...,4:7])
 * * lambda_dt*MSE(dt_hat, roll(x[...,1], -1) with a zero last entry), each a mean over the selected
 * tokens, targets read from the MASKED x exactly as the reference does (:262-271).  f_hat / dt_hat
 * [B, L+1], b_hat [B, L+1, 3] are the heads applied to the whole encoder output (token 0 = CLS, zero
 * gradient); outputs: loss (1 float), df, db, ddt in the same shapes.  sums: 4 floats of scratch.
 * ---------------------------------------------------------------------- */
int ac_mpt_mask(float *x, const uint8_t *pad, uint8_t *masked, int32_t B, int32_t L, double mask_p,
                uint64_t seed, const uint64_t *step, ac_stream_t stream);
int ac_mpt_loss_fwd_bwd(const float *f_hat, const float *b_hat, const float *dt_hat,
                        const float *data, const uint8_t *masked, float *sums, float *loss, float *df,
                        float *db, float *ddt, int32_t B, int32_t L, float lambda_f, float lambda_b,
                        float lambda_dt, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimizers over flat fp32 buffers (torch.optim.AdamW astrominn.py:151-218,
 * Adam HyraxBaselineCLS.py:41, SGD injected by Hyrax for SpectraNet).
 * seg_* describe param groups as [begin,end) element ranges with their own
 * hyper-parameters; step is the 1-based step count.
 * ---------------------------------------------------------------------- */
typedef struct ac_adam_seg {
    int64_t begin, end;
    float lr, beta1, beta2, eps, weight_decay;
    int32_t decoupled; /* 1 = AdamW, 0 = Adam (L2 into grad) */
} ac_adam_seg;
/* segs is a HOST array; grad_scale_dev (nullable) is a DEVICE scalar multiplied into the
 * gradient (the clip coefficient of ac_clip_coef), so clipping needs no host sync. */
int ac_adam_flat(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                 const ac_adam_seg *segs_host, int32_t nseg, int32_t step,
                 const float *grad_scale_dev, ac_stream_t stream);
/* Same step with the 1-based step count read from DEVICE memory (bias corrections computed in the
 * kernel): nothing about the step number is baked into the launch, so a captured hipGraph of a
 * whole training step (zero_grad .. optimizer) replays correctly.  torch keeps `state["step"]` as
 * a tensor for the same reason (capturable=True). */
int ac_adam_flat_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                     const ac_adam_seg *segs_host, int32_t nseg, const int64_t *step_dev,
                     const float *grad_scale_dev, ac_stream_t stream);
int ac_sgd_flat(float *param, const float *grad, float *momentum_buf, int64_t n, float lr,
                float momentum, float weight_decay, int32_t first_step, ac_stream_t stream);
/* Device-resident step counter for graph capture.
 *   ac_step_advance(counter, stream): counter[0] += 1 on the stream (one tiny kernel).
 *   Every entry point that draws random numbers (ac_dropout, ac_gemm's epilogue dropout through
 *   ac_gemm_desc.drop_step, ac_mha_* attention dropout, ac_mpt_mask) takes a nullable `step` pointer to
 *   such a counter and uses seed' = seed + step[0] * 0x9E3779B97F4A7C15 instead of the seed its launch
 *   was given (a counter at 0 and a NULL pointer are the same thing).  Forward and backward of one
 *   training step read the same counter value, so their masks agree; a replayed hipGraph (same host
 *   seeds baked in) draws new masks once the counter advanced.  The library keeps no state: the
 *   counter is the caller's.  This is what torch's philox (seed, offset) pair kept on the device does
 *   for CUDA graphs (nn.Dropout / MultiheadAttention dropout: photo_events.py:55-64,
 *   HyraxBaselineCLS.py:31). */
int ac_step_advance(uint64_t *counter_dev, ac_stream_t stream);
/* out[0] = sum x^2 (zeroed by the call); clip coefficient computed on device:
 * coef[0] = min(1, max_norm / (sqrt(sumsq) + 1e-6)). */
int ac_sumsq(const float *x, int64_t n, float *out, ac_stream_t stream);
int ac_clip_coef(const float *sumsq, float max_norm, float *coef, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Measured ceilings of the box (bench.py `ceilings`; SURVEY.md section 8d asks for the on-box copy rate
 * and the register-resident MFMA rate beside the vendor peaks).  They replace nothing of the reference:
 * measurement utilities on the same C ABI so that the bench needs no second library.
 *   ac_ceil_copy: dst[0..bytes) = src[0..bytes), 16 bytes per lane, grid-stride (bytes % 16 == 0).
 *   ac_ceil_mfma: `workgroups` x `waves_per_wg` (4 = one wave per SIMD, 8 = two) waves each run `iters`
 *       rounds of 16 x v_mfma_f32_16x16x32 (shape 0) or 8 x v_mfma_f32_32x32x16 (shape 1) on 16-bit
 *       operands taken once from `ops` (64*8*64*64 values of the library's operand format) into
 *       registers; FLOP = workgroups * waves_per_wg * iters * 16 * 2*16*16*32.  `out` receives one
 *       float per thread (the accumulator sums: keeps the work alive).
 */
int ac_ceil_copy(const void *src, void *dst, int64_t bytes, ac_stream_t stream);
int ac_ceil_mfma(const void *ops, float *out, int32_t shape, int32_t workgroups, int32_t waves_per_wg,
                 int32_t iters, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Frequency-domain form of the long-tap Conv1d products of a SpectraNetBlock (spectranet.py:18-20,25 and torch's
 * conv1d backward; stage 2 of default_config.toml:104-114 is Conv1d(64 -> 128, k = 251) on 1024 positions).  With
 * N = 2^logn >= L + k/2 the three products of one convolution are
 *     y  = irfft( rfft(x) H )          dx = irfft( rfft(dy) conj(H)^T )          dw = irfft( conj(rfft(x))^T rfft(dy) )
 * per frequency f <= N/2 — about k / log2(N) times fewer FLOP than the direct form, in fp32 throughout.
 * Spectra are [N/2 + 1][B][2C] fp32 (a complex number = two adjacent columns: re, im).
 *
 * ac_gemm_batched: `batch` independent products of one shape in one launch; product z takes its operands
 *   bs_a / bs_b / bs_c ELEMENTS after those of product z - 1 (multiples of 4).  math = AC_MATH_F32 or
 *   AC_MATH_BF16X3, split_k = 1, 16-byte aligned operands (the matrix-core kernels; AC_EINVAL otherwise).
 * Transform sizes: N = 2^logn, N = 3 * 2^logn (radix3 = 1) or N = 9 * 2^logn (radix3 = 2) — a 'same' convolution needs
 *   L + k/2 points, rarely a power of two (stage 2's k = 251: 1149 -> 1152 instead of 2048).  Write M = 2^logn.
 * Twiddle table `tw`, provided by the caller (the library never allocates): M complex (2 floats each) entries —
 *   level e (0 <= e < logn) holds exp(-2 pi i j / (M >> e)), j < M >> (e + 1), at element offset M - (M >> e), one pad —
 *   followed, for radix3 > 0, by exp(-2 pi i t / N), t < 2 N / 3.
 * ac_fft_rows_fwd: every sample becomes `blocks` zero-filled length-N sequences; in block r the row
 *   rows[b, l, col_off + c] (element at b * batch_stride + l * row_stride + col_off + c; fp32, or — rows_lo non-null — a
 *   (hi, lo) bf16 plane pair of the same strides whose sum is the value), l < L, c < C, sits at sequence index
 *   n = l - r * block_step + shift when 0 <= n < N and n_lo <= n < n_hi (n_lo = n_hi = 0: no mask)
 *   -> spec [N/2 + 1][B * blocks][2C], spectrum row b * blocks + r.  C % 16 == 0, 5 <= logn <= 11.
 *   blocks = 1: one sequence per sample (L + shift <= N).  blocks > 1: overlap-save windows advancing by block_step.
 * ac_fft_rows_inv: the inverse, scaled by 1/N: rows[b, r * block_step + j, col_off + c] (+)= seq_r[j + shift] + bias[c]
 *   for j < block_step (blocks > 1; j < L for blocks = 1) and r * block_step + j < L; bias nullable; fp32 rows only.
 * ac_fft_taps_fwd: taps w[co][t][ci] -> hblock [N/2 + 1][2 Cout][2 Cin], the real block form [[Hr, -Hi], [Hi, Hr]]
 *   of the spectrum of h[m] = w[co][k - 1 - m][ci].  One sequence per sample: x at shift 0, y = x * h read at shift
 *   k - 1 - k/2 is the 'same' correlation of nn.Conv1d (N >= L + k/2).  Overlap-save (block_step = N - k + 1): x windows
 *   at shift k/2, y read at shift k - 1.  Forward product: NT with A = spectrum of x, B = hblock; input gradient: NN
 *   with A = spectrum of dy (same placement as y), B = hblock, dx read at shift 0.  Cin % 16 == 0.
 * ac_fft_taps_inv: m [N/2 + 1][2 Cout][2 Cin] = (spectrum of dy)^T (spectrum of x) per frequency (TN product; overlap-
 *   save: x masked to the block's own block_step rows at shift 0, n_hi = block_step)
 *   -> dw[co][t][ci] += the weight gradient (dw is accumulated into: zero it or pass the gradient sink).
 * ---------------------------------------------------------------------- */
typedef struct ac_fft_rows_desc {
    const void *rows;    /* the real tensor (forward: source; inverse: destination, written through this pointer) */
    const void *rows_lo; /* forward only, nullable: lo plane (rows = hi plane), both bf16 */
    float *spec;         /* spectrum [N/2 + 1][B * blocks][2C] */
    const float *tw;
    const float *bias;   /* inverse only, nullable */
    int64_t batch_stride, row_stride;
    int32_t col_off, B, L, C, logn;
    int32_t radix3;      /* radix-3 stages: 0: N = 2^logn (5 <= logn <= 11); 1: N = 3 * 2^logn (3 <= logn <= 9);
                            2: N = 9 * 2^logn (3 <= logn <= 7) */
    int32_t blocks, block_step, shift;
    int32_t n_lo, n_hi;  /* forward only */
    int32_t accumulate;  /* inverse only: rows += */
    int32_t lds_exact;   /* 0 (default): every transform workgroup requests a whole CU's LDS and shares its CU with
                            nothing; 1 (diagnostic): the exact request, workgroups of other kernels may share the CU */
} ac_fft_rows_desc;
int ac_gemm_batched(const ac_gemm_desc *d, int32_t batch, int64_t bs_a, int64_t bs_b, int64_t bs_c, ac_stream_t stream);
/* `groups` (<= 16) independent products of ONE shape / mode / math in one launch: product g uses the descriptor with
 * a.ptr, b.ptr, c.ptr replaced by ptrs[3 g], ptrs[3 g + 1], ptrs[3 g + 2] (HOST array of device pointers, 16-byte
 * aligned; read during the call).  No epilogue beyond store / += ; split_k > 1 with accumulate = 2 (atomics) is allowed:
 * the weight gradients of a stage's ConvNeXt blocks / of the encoder layers in one launch that fills the chip. */
int ac_gemm_grouped(const ac_gemm_desc *d, int32_t groups, const void *const *ptrs, ac_stream_t stream);
int ac_fft_rows_fwd(const ac_fft_rows_desc *d, ac_stream_t stream);
int ac_fft_rows_inv(const ac_fft_rows_desc *d, ac_stream_t stream);
int ac_fft_taps_fwd(const float *w, int32_t Cout, int32_t Cin, int32_t k, int32_t logn, int32_t radix3, const float *tw,
                    float *hblock, ac_stream_t stream);
int ac_fft_taps_inv(const float *m, int32_t Cout, int32_t Cin, int32_t k, int32_t logn, int32_t radix3, const float *tw,
                    float *dw, ac_stream_t stream);

/* ------------------------------------------------------------------------
 * Tail of a pooled SpectraNetBlock (spectranet.py:31-40), split-bf16 arithmetic, fused so that nothing of row length
 * K is written in the forward pass and only the operand planes of d ycat in the backward pass:
 *     pooled[r / 4, n] = max_{j < 4} ( bias[n] + sum_k gelu(LN(ycat[4 (r/4) + j, :]))[k] * w[n, k] ),  idx = argmax j
 * ycat [rows, K] contiguous fp32 (rows = B * L, pooling windows = rows 4i .. 4i + 3), gamma / beta [K], w [N, K].
 * Shapes: ac_spectail_supported(rows, K, N) != 0  <=>  rows % 32 == 0 and (K, N) in {(192, 64), (384, 128)}.
 * GELU is the erf form through a rational erf (max abs error 4.5e-7).
 *   ac_spectail_fwd     w_hi / w_lo: (hi, lo) bf16 planes of w [N][K] (ac_split_bf16).  Outputs mean / rstd [rows]
 *                       (biased variance, eps inside the root), pooled [rows / 4, N], idx uint8 [rows / 4, N] (first
 *                       maximum wins, NaN propagates: torch.nn.MaxPool1d).
 *   ac_spectail_bwd_dx  d pooled [rows / 4, N] -> (hi, lo) planes of d ycat [.., K] (row r = (b, l) lands at
 *                       b * seg_pitch + seg_off + l when seg_len = L > 0, at r otherwise: the zero-padded layout the
 *                       conv bank's gradient products read; the caller zeroes the pad rows), and ACCUMULATES
 *                       dgamma / dbeta / dxsum [K] (dxsum = column sums of d ycat = the conv biases' gradient; each
 *                       nullable).  wt_hi / wt_lo: planes of the TRANSPOSE w^T [K][N] (ac_transpose_split_bf16).
 *   ac_spectail_bwd_dw  dw [N][K] += scatter(d pooled)^T . gelu(LN(ycat))   (fp32 atomics).
 * The 1x1 conv's bias gradient is the column sum of d pooled (ac_colsum).
 * ---------------------------------------------------------------------- */
int ac_spectail_supported(int64_t rows, int32_t K, int32_t N);
int ac_spectail_fwd(const float *ycat, const float *gamma, const float *beta, float eps, const void *w_hi,
                    const void *w_lo, const float *bias, float *mean, float *rstd, float *pooled, uint8_t *idx,
                    int64_t rows, int32_t K, int32_t N, ac_stream_t stream);
int ac_spectail_bwd_dx(const float *ycat, const float *mean, const float *rstd, const float *gamma, const float *beta,
                       const float *dpooled, const uint8_t *idx, const void *wt_hi, const void *wt_lo, void *dx_hi,
                       void *dx_lo, int32_t seg_len, int32_t seg_pitch, int32_t seg_off, float *dgamma, float *dbeta,
                       float *dxsum, int64_t rows, int32_t K, int32_t N, ac_stream_t stream);
int ac_spectail_bwd_dw(const float *ycat, const float *mean, const float *rstd, const float *gamma, const float *beta,
                       const float *dpooled, const uint8_t *idx, float *dw, int64_t rows, int32_t K, int32_t N,
                       ac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* APPLECIDER_HIP_H */
