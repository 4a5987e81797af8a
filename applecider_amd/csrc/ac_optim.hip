// Optimizer steps over flat fp32 parameter / gradient buffers (pure HBM streaming:
// AdamW reads p,g,m,v and writes p,m,v = 28 B per parameter).
//   torch.optim.AdamW  astrominn.py:151-218 (11 groups -> 11 segments)
//   torch.optim.Adam   HyraxBaselineCLS.py:41 ; brew_cider.py:1211
//   torch.optim.SGD    injected by Hyrax for SpectraNet (spectranet.py:172-184)
//   clip_grad_norm_    HyraxBaselineCLS.py:112
#include "ac_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p,
                                                   const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   int64_t begin, int64_t end, float lr,
                                                   float beta1, float beta2, float eps, float wd,
                                                   int decoupled, float bc1, float bc2_sqrt,
                                                   const float *__restrict__ gscale,
                                                   const int64_t *__restrict__ step_dev) {
    const float gs = gscale ? gscale[0] : 1.0f;
    if (step_dev) {  // step count lives in HBM (captured hipGraph): bias corrections computed here
        const float st = (float)step_dev[0];
        bc1 = 1.0f - powf(beta1, st);
        bc2_sqrt = sqrtf(1.0f - powf(beta2, st));
    }
    const float step_size = lr / bc1;
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
         i += (int64_t)gridDim.x * blockDim.x) {
        float pv = p[i];
        float gv = g[i] * gs;
        if (decoupled)
            pv *= (1.0f - lr * wd);
        else
            gv = fmaf(wd, pv, gv);
        // torch: exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
        float mv = m[i] + (gv - m[i]) * (1.0f - beta1);
        float vv = v[i] * beta2 + (1.0f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pv - step_size * (mv / denom);
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ p,
                                                  const float *__restrict__ g,
                                                  float *__restrict__ buf, int64_t n, float lr,
                                                  float momentum, float wd, int first_step) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float gv = fmaf(wd, p[i], g[i]);
        if (momentum != 0.f) {
            float bv = first_step ? gv : momentum * buf[i] + gv;
            buf[i] = bv;
            gv = bv;
        }
        p[i] -= lr * gv;
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ x, int64_t n,
                                                    float *__restrict__ out) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        s = fmaf(x[i], x[i], s);
    s = ac_wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ void step_advance_kernel(uint64_t *c) { c[0] += 1; }

__global__ void clip_coef_kernel(const float *sumsq, float max_norm, float *coef) {
    const float total = sqrtf(sumsq[0]);
    const float c = max_norm / (total + 1e-6f);
    coef[0] = c < 1.0f ? c : 1.0f;
}

inline int stream_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

int adam_launch(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                const ac_adam_seg *segs, int32_t nseg, int32_t step, const int64_t *step_dev,
                const float *grad_scale_dev, ac_stream_t stream);

}  // namespace

extern "C" int ac_step_advance(uint64_t *counter_dev, ac_stream_t stream) {
    if (!counter_dev) return AC_EINVAL;
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_adam_flat(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                            const ac_adam_seg *segs, int32_t nseg, int32_t step,
                            const float *grad_scale_dev, ac_stream_t stream) {
    if (step < 1) return AC_EINVAL;
    return adam_launch(param, grad, exp_avg, exp_avg_sq, segs, nseg, step, nullptr, grad_scale_dev,
                       stream);
}

extern "C" int ac_adam_flat_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                const ac_adam_seg *segs, int32_t nseg, const int64_t *step_dev,
                                const float *grad_scale_dev, ac_stream_t stream) {
    if (!step_dev) return AC_EINVAL;
    return adam_launch(param, grad, exp_avg, exp_avg_sq, segs, nseg, 1, step_dev, grad_scale_dev,
                       stream);
}

namespace {
int adam_launch(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                const ac_adam_seg *segs, int32_t nseg, int32_t step, const int64_t *step_dev,
                const float *grad_scale_dev, ac_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !segs || nseg <= 0)
        return AC_EINVAL;
    for (int s = 0; s < nseg; ++s) {
        const ac_adam_seg &g = segs[s];
        if (g.end < g.begin) return AC_EINVAL;
        if (g.end == g.begin) continue;
        const float bc1 = 1.0f - powf(g.beta1, (float)step);
        const float bc2 = 1.0f - powf(g.beta2, (float)step);
        hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(g.end - g.begin)), dim3(256), 0,
                           (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, g.begin, g.end,
                           g.lr, g.beta1, g.beta2, g.eps, g.weight_decay, g.decoupled, bc1,
                           sqrtf(bc2), grad_scale_dev, step_dev);
        AC_CHECK_LAUNCH();
    }
    return AC_OK;
}
}  // namespace

extern "C" int ac_sgd_flat(float *param, const float *grad, float *momentum_buf, int64_t n,
                           float lr, float momentum, float weight_decay, int32_t first_step,
                           ac_stream_t stream) {
    if (!param || !grad || n < 0) return AC_EINVAL;
    if (momentum != 0.f && !momentum_buf) return AC_EINVAL;
    if (n == 0) return AC_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, param,
                       grad, momentum_buf, n, lr, momentum, weight_decay, first_step);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_sumsq(const float *x, int64_t n, float *out, ac_stream_t stream) {
    if (!x || !out || n < 0) return AC_EINVAL;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return -(int)e - 2000;
    if (n == 0) return AC_OK;
    hipLaunchKernelGGL(sumsq_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n,
                       out);
    AC_CHECK_LAUNCH();
    return AC_OK;
}

extern "C" int ac_clip_coef(const float *sumsq, float max_norm, float *coef, ac_stream_t stream) {
    if (!sumsq || !coef) return AC_EINVAL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm,
                       coef);
    AC_CHECK_LAUNCH();
    return AC_OK;
}
