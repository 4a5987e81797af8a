// BatchNorm1d over the channels of a channels-last sequence tensor + activation: the `use_ln=False`
// form of SpectraNetBlock (src/applecider/models/spectranet.py:21-37: nn.BatchNorm1d(norm_channels)
// on [B, C, L], then GELU).  Rows = B*L positions, columns = channels: statistics are COLUMN moments.
// Training: batch mean / biased variance normalise, running statistics move with `momentum` (unbiased
// variance), exactly nn.BatchNorm1d.  All kernels are HBM streaming passes:
//   forward   moments (1 read)            -> finalize (C elements) -> apply (1 read, 1 write)
//   backward  reduce dz, dz*x (2 reads)   -> finalize              -> apply (2 reads, 1 write)
// with dz = dy * act'(z), z = x*scale + shift recomputed from x (no pre-activation is stored).
#include "ac_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// column sums of (u, v) over a slab of rows: lane = 4 consecutive columns, 4 row phases per workgroup
template <typename F>
__device__ __forceinline__ void col_reduce2(F &&rowfn, float *out_u, float *out_v, int64_t rows, int cols,
                                            int rows_per_block) {
    __shared__ float part[4][2][256];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + 4 * cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    f32x4 su = {0.f, 0.f, 0.f, 0.f}, sv = su;
    if (c < cols)
        for (int64_t r = r0 + ph; r < r1; r += 4) rowfn(r, c, su, sv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[ph][0][4 * cl + j] = su[j];
        part[ph][1][4 * cl + j] = sv[j];
    }
    __syncthreads();
    const int cc = blockIdx.y * 256 + threadIdx.x;
    if (cc < cols) {
        const int i = threadIdx.x;
        atomicAdd(out_u + cc, (part[0][0][i] + part[1][0][i]) + (part[2][0][i] + part[3][0][i]));
        atomicAdd(out_v + cc, (part[0][1][i] + part[1][1][i]) + (part[2][1][i] + part[3][1][i]));
    }
}

__global__ __launch_bounds__(256) void bn_moments_kernel(const float *__restrict__ x, int64_t ld,
                                                         float *__restrict__ sum, float *__restrict__ sumsq,
                                                         int64_t rows, int cols, int rpb) {
    // SHIFTED moments: sums of (x - K) and (x - K)^2 with K = the column's first row.  E[x^2] - mean^2 in one
    // fp32 pass cancels catastrophically when |mean| >> std (2M rows at the benchmark shape); about a value of
    // the column itself the two terms are of the size of the variance, as in a two-pass scheme, at one read.
    const int cb = blockIdx.y * 256 + 4 * (threadIdx.x & 63);
    f32x4 K = {0.f, 0.f, 0.f, 0.f};
    if (cb < cols) K = *(const f32x4 *)(x + cb);
    col_reduce2([&](int64_t r, int c, f32x4 &su, f32x4 &sv) {
        const f32x4 v = *(const f32x4 *)(x + r * ld + c) - K;
        su += v;
        sv += v * v;
    }, sum, sumsq, rows, cols, rpb);
}

// per channel: mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; running statistics
__global__ void bn_finalize_fwd_kernel(const float *x_row0, const float *sum, const float *sumsq, const float *gamma,
                                       const float *beta, float *running_mean, float *running_var,
                                       float *mean, float *rstd, float *scale, float *shift, int cols,
                                       float n, float eps, float momentum, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float m, var;
    if (training) {
        const float d = sum[c] / n;                 // mean of (x - K), K = x[0][c] (bn_moments_kernel)
        m = x_row0[c] + d;
        var = fmaxf(sumsq[c] / n - d * d, 0.f);     // biased, as the normalisation uses it
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
        }
    } else {
        m = running_mean[c];
        var = running_var[c];
    }
    const float rs = rsqrtf(var + eps);
    mean[c] = m;
    rstd[c] = rs;
    scale[c] = gamma[c] * rs;
    shift[c] = beta[c] - m * gamma[c] * rs;
}

__global__ void bn_apply_fwd_kernel(const float *__restrict__ x, int64_t ld, const float *__restrict__ scale,
                                    const float *__restrict__ shift, float *__restrict__ y, int64_t ldy,
                                    int64_t rows, int cols, int act) {
    const int c4 = cols >> 2;
    const int64_t n = rows * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        f32x4 v = *(const f32x4 *)(x + r * ld + c) * *(const f32x4 *)(scale + c) + *(const f32x4 *)(shift + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ac_act(v[j], act);
        *(f32x4 *)(y + r * ldy + c) = v;
    }
}

__device__ __forceinline__ float bn_dz(float dy, float x, float sc, float sh, int act) {
    return act ? dy * ac_dact(act == AC_ACT_GELU || act == AC_ACT_RELU ? x * sc + sh : ac_act(x * sc + sh, act), act) : dy;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dy, int64_t lddy,
                                                            const float *__restrict__ x, int64_t ld,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            const float *__restrict__ mean,
                                                            float *__restrict__ sum_dz, float *__restrict__ sum_dzx,
                                                            int64_t rows, int cols, int act, int rpb) {
    const int cb = blockIdx.y * 256 + 4 * (threadIdx.x & 63);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, mu = sc;
    if (cb < cols) {
        sc = *(const f32x4 *)(scale + cb);
        sh = *(const f32x4 *)(shift + cb);
        mu = *(const f32x4 *)(mean + cb);
    }
    col_reduce2([&](int64_t r, int c, f32x4 &su, f32x4 &sv) {
        const f32x4 g = *(const f32x4 *)(dy + r * lddy + c), v = *(const f32x4 *)(x + r * ld + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dz = bn_dz(g[j], v[j], sc[j], sh[j], act);
            su[j] += dz;
            sv[j] += dz * (v[j] - mu[j]);   // centred: sum dz*x - mean*sum dz cancels when |mean| >> std
        }
    }, sum_dz, sum_dzx, rows, cols, rpb);
}

// dx = dz*A + x*Bc + C0 ;  dgamma = sum dz*xhat ; dbeta = sum dz
__global__ void bn_finalize_bwd_kernel(const float *sum_dz, const float *sum_dzx, const float *gamma,
                                       const float *mean, const float *rstd, float *A, float *Bc, float *C0,
                                       float *dgamma, float *dbeta, int cols, float n, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const float m = mean[c], rs = rstd[c], a = gamma[c] * rs;
    const float s1 = sum_dz[c], s2 = rs * sum_dzx[c];   // sum dz, sum dz*xhat (the reduction is centred)
    dgamma[c] += s2;
    dbeta[c] += s1;
    A[c] = a;
    if (training) {
        const float m1 = s1 / n, m2 = s2 / n;
        Bc[c] = -a * rs * m2;
        C0[c] = -a * m1 + a * rs * m * m2;
    } else {   // running statistics are constants
        Bc[c] = 0.f;
        C0[c] = 0.f;
    }
}

__global__ void bn_bwd_apply_kernel(const float *__restrict__ dy, int64_t lddy, const float *__restrict__ x,
                                    int64_t ld, const float *__restrict__ scale, const float *__restrict__ shift,
                                    const float *__restrict__ A, const float *__restrict__ Bc,
                                    const float *__restrict__ C0, float *__restrict__ dx, int64_t lddx,
                                    int64_t rows, int cols, int act) {
    const int c4 = cols >> 2;
    const int64_t n = rows * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const f32x4 g = *(const f32x4 *)(dy + r * lddy + c), v = *(const f32x4 *)(x + r * ld + c);
        const f32x4 sc = *(const f32x4 *)(scale + c), sh = *(const f32x4 *)(shift + c);
        const f32x4 a = *(const f32x4 *)(A + c), b = *(const f32x4 *)(Bc + c), c0 = *(const f32x4 *)(C0 + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = bn_dz(g[j], v[j], sc[j], sh[j], act) * a[j] + v[j] * b[j] + c0[j];
        *(f32x4 *)(dx + r * lddx + c) = o;
    }
}

inline int rows_per_block(int64_t rows) {
    int rpb = 64;
    while ((rows + rpb - 1) / rpb > 1024) rpb *= 2;
    return rpb;
}

}  // namespace

// stats: 4*cols floats of scratch owned by the caller = {mean, rstd, scale, shift}; sums: 2*cols zeroed floats
extern "C" int ac_batchnorm_fwd(const float *x, int64_t ld, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, float *y, int64_t ldy, float *stats,
                                float *sums, int64_t rows, int32_t cols, float eps, float momentum,
                                int32_t training, int32_t act, ac_stream_t stream_) {
    if (!x || !gamma || !beta || !y || !stats || rows <= 0 || cols <= 0) return AC_EINVAL;
    if (!training && (!running_mean || !running_var)) return AC_EINVAL;
    if (training && !sums) return AC_EINVAL;
    if ((cols % 4) || (ld % 4) || (ldy % 4) || !ac_aligned16(x) || !ac_aligned16(y) || !ac_aligned16(stats))
        return AC_EALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    float *mean = stats, *rstd = stats + cols, *scale = stats + 2 * cols, *shift = stats + 3 * cols;
    if (training) {
        const int rpb = rows_per_block(rows);
        dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 255) / 256));
        hipLaunchKernelGGL(bn_moments_kernel, grid, dim3(256), 0, stream, x, ld, sums, sums + cols, rows, cols, rpb);
    }
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, x, sums,
                       sums ? sums + cols : nullptr, gamma, beta, running_mean, running_var, mean, rstd, scale, shift,
                       cols, (float)rows, eps, momentum, training);
    int64_t g = (rows * (cols / 4) + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, ld, scale, shift, y, ldy, rows,
                       cols, act);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

// work: 5*cols floats of scratch = {sum_dz, sum_dzx (both zeroed by the caller), A, Bc, C0};
// dgamma / dbeta are accumulated into (+=)
extern "C" int ac_batchnorm_bwd(const float *dy, int64_t lddy, const float *x, int64_t ld, const float *gamma,
                                const float *stats, float *dx, int64_t lddx, float *dgamma, float *dbeta,
                                float *work, int64_t rows, int32_t cols, int32_t training, int32_t act,
                                ac_stream_t stream_) {
    if (!dy || !x || !gamma || !stats || !dx || !dgamma || !dbeta || !work || rows <= 0 || cols <= 0) return AC_EINVAL;
    if ((cols % 4) || (ld % 4) || (lddy % 4) || (lddx % 4) || !ac_aligned16(x) || !ac_aligned16(dy) ||
        !ac_aligned16(dx) || !ac_aligned16(stats) || !ac_aligned16(work))
        return AC_EALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    const float *mean = stats, *rstd = stats + cols, *scale = stats + 2 * cols, *shift = stats + 3 * cols;
    float *s1 = work, *s2 = work + cols, *A = work + 2 * cols, *Bc = work + 3 * cols, *C0 = work + 4 * cols;
    const int rpb = rows_per_block(rows);
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 255) / 256));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, stream, dy, lddy, x, ld, scale, shift, mean, s1, s2, rows,
                       cols, act, rpb);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, s1, s2, gamma, mean,
                       rstd, A, Bc, C0, dgamma, dbeta, cols, (float)rows, training);
    int64_t g = (rows * (cols / 4) + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)g), dim3(256), 0, stream, dy, lddy, x, ld, scale, shift, A,
                       Bc, C0, dx, lddx, rows, cols, act);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
