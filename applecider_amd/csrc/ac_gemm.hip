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
};

__device__ __forceinline__ void epilogue_store(const ac_gemm_desc &d, int m, int n, float acc,
                                               int64_t caddr) {
    float v = acc * d.alpha;
    if (d.bias) v += d.bias[n];
    if (d.pre_out) d.pre_out[(int64_t)m * d.ld_pre + n] = v;
    v = ac_act(v, d.act);
    if (d.dact) v *= ac_dact(d.aux[(int64_t)m * d.ld_aux + n], d.dact);
    if (d.colscale) v *= d.colscale[n];
    if (d.residual) v += d.residual[(int64_t)m * d.ld_res + n];
    float *c = (float *)d.c.ptr + caddr;
    if (d.accumulate == 2)
        atomicAdd(c, v);
    else if (d.accumulate == 1)
        *c += v;
    else
        *c = v;
}

__device__ __forceinline__ int64_t inner_off(const int32_t *goff, int i) {
    return goff ? (int64_t)goff[i >> 5] + (i & 31) : (int64_t)i;
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
__device__ __forceinline__ void store_tile(const ac_gemm_desc &d, const f32x16 (&acc)[2][2], int tm,
                                           int tn, int wm, int wn, int li, int lh) {
    const int n0 = tn * BN + wn * 64 + li, n1 = n0 + 32;
    const int64_t c0 = inner_off(d.c.goff, n0 < d.N ? n0 : 0);
    const int64_t c1 = inner_off(d.c.goff, n1 < d.N ? n1 : 0);
    static_for<0, 32>([&](auto idx) {
        constexpr int sa = decltype(idx)::value / 16, e = decltype(idx)::value % 16;
        const int m = tm * BM + wm * 64 + sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < d.M) {
            const int64_t crow = ac_rowaddr(d.c.rows, m);
            if (n0 < d.N) epilogue_store(d, m, n0, acc[sa][0][e], crow + c0);
            if (n1 < d.N) epilogue_store(d, m, n1, acc[sa][1][e], crow + c1);
        }
    });
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
            }
        } else {
            int col = tile_origin + 4 * (t & 31);
            bool cv = col < inner_n;
            int64_t io = cv ? inner_off(goff, col) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                base[i] = io;
                ok[i] = cv;
            }
        }
    }

    __device__ __forceinline__ void load(int kt, f32x4 (&v)[4]) const {
        if (KC) {
            int k = kt * BK + 4 * (t & 7);
            bool kv = k < inner_n;
            int64_t ko = goff ? (int64_t)goff[kt] : (int64_t)kt * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kv)
                    v[i] = *(const f32x4 *)(ptr + base[i] + ko);
                else
                    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int kg = kt * BK + (t >> 5) + 8 * i;
                if (ok[i] && kg < outer_n)
                    v[i] = *(const f32x4 *)(ptr + ac_rowaddr(rows, kg) + base[i]);
                else
                    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
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

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ac_gemm_desc &d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: workgroups that share an XCD (bid % 8) walk contiguous tiles
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;

    const int kt_begin = blockIdx.y * p.kt_per_split;
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

    store_tile(d, acc, tm, tn, wm, wn, li, lh);
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
    __hip_bfloat16 b = __float2bfloat16(x);
    return *reinterpret_cast<unsigned short *>(&b);
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
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = wg % p.tiles_n, tm = wg / p.tiles_n;

    const int kt_begin = blockIdx.y * p.kt_per_split;
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
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            store_bf16<A_KC>(la, sm16 + (cur ^ 1) * 2 * TILE_FLOATS, ra);
            store_bf16<B_KC>(lb, sm16 + (cur ^ 1) * 2 * TILE_FLOATS + TILE_FLOATS, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    store_tile(d, acc, tm, tn, wm, wn, li, lh);
}

// ---------------------------------------------------------------------------
// Scalar kernel for tiny or unaligned products (metadata towers, router, class
// heads: K, N of 2..48).  One thread per output element.
// ---------------------------------------------------------------------------
__global__ void gemm_simple_kernel(GemmParams p) {
    const ac_gemm_desc &d = p.d;
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
    epilogue_store(d, m, n, acc, ac_rowaddr(d.c.rows, m) + inner_off(d.c.goff, n));
}

bool rowmap_aligned(const ac_rowmap &r) {
    return (r.s1 % 4 == 0) && (r.s2 % 4 == 0) && (r.s3 % 4 == 0);
}

}  // namespace

extern "C" int ac_gemm(const ac_gemm_desc *dp, ac_stream_t stream_) {
    if (!dp) return AC_EINVAL;
    ac_gemm_desc d = *dp;
    hipStream_t stream = (hipStream_t)stream_;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return AC_EINVAL;
    if (d.mode < 0 || d.mode > 2) return AC_EINVAL;
    if (!d.a.ptr || !d.b.ptr || !d.c.ptr) return AC_EINVAL;
    if (d.dact && !d.aux) return AC_EINVAL;
    if (d.a.rows.r1 != 0 && d.a.rows.r2 == 0) return AC_EINVAL;
    if (d.b.rows.r1 != 0 && d.b.rows.r2 == 0) return AC_EINVAL;
    if (d.c.rows.r1 != 0 && d.c.rows.r2 == 0) return AC_EINVAL;
    if (d.split_k < 1) d.split_k = 1;

    const bool a_kc = d.mode != AC_GEMM_TN;
    const bool b_kc = d.mode == AC_GEMM_NT;
    // inner extents must be multiples of 4 so that 16-byte chunks never straddle the edge
    const int a_inner = a_kc ? d.K : d.M;
    const int b_inner = b_kc ? d.K : d.N;
    bool aligned = ac_aligned16(d.a.ptr) && ac_aligned16(d.b.ptr) && rowmap_aligned(d.a.rows) &&
                   rowmap_aligned(d.b.rows) && (a_inner % 4 == 0) && (b_inner % 4 == 0);
    const double macs = (double)d.M * d.N * d.K;
    bool use_mfma = aligned && !d.force_simple && macs >= 262144.0;

    GemmParams p;
    p.d = d;
    if (!use_mfma) {
        // split_k is only a scheduling hint: the scalar kernel computes whole dot products
        // and honours the caller's accumulate mode.
        p.tiles_m = p.tiles_n = p.nkt = p.kt_per_split = 0;
        int64_t total = (int64_t)d.M * d.N;
        int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL(gemm_simple_kernel, dim3(blocks), dim3(256), 0, stream, p);
        AC_CHECK_LAUNCH();
        return AC_OK;
    }
    if (d.split_k > 1) p.d.accumulate = 2;
    if (d.split_k > 1 && (d.bias || d.pre_out || d.act || d.dact || d.residual)) return AC_EINVAL;
    p.tiles_m = (d.M + BM - 1) / BM;
    p.tiles_n = (d.N + BN - 1) / BN;
    p.nkt = (d.K + BK - 1) / BK;
    p.kt_per_split = (p.nkt + d.split_k - 1) / d.split_k;
    dim3 grid(p.tiles_m * p.tiles_n, d.split_k);
    const size_t lds_f32 = 4 * TILE_FLOATS * sizeof(float);  // 64 KB
    const size_t lds_bf16 = 4 * TILE_FLOATS * sizeof(short); // 32 KB
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
