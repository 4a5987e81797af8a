// Diagnostic (not part of the product): do co-resident workgroups of two kernels with different dynamic LDS sizes
// overlap in LDS?  Every workgroup fills its whole dynamic allocation with a pattern derived from (kernel tag, block),
// spins, and verifies.  Also: the hardware's LDS allocation register of a workgroup (eager vs graph launches).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lds_probe_kernel(uint32_t *out) {
    extern __shared__ float buf[];
    if (threadIdx.x == 0) {
        out[blockIdx.x] = __builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11));   // HW_REG_LDS_ALLOC
        buf[0] = 1.f;
    }
}

__global__ void lds_pattern_kernel(uint32_t *errors, uint32_t tag, int words, int spin) {
    extern __shared__ uint32_t w[];
    const uint32_t key = tag * 0x9E3779B1u + blockIdx.x * 0x85EBCA77u;
    for (int i = threadIdx.x; i < words; i += blockDim.x) w[i] = key ^ (uint32_t)i;
    __syncthreads();
    uint32_t bad = 0;
    for (int s = 0; s < spin; ++s) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) bad += (w[i] != (key ^ (uint32_t)i));
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += blockDim.x) w[i] = key ^ (uint32_t)i;   // rewrite
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}

extern "C" int lds_probe(uint32_t *out, int blocks, int threads, size_t lds, void *stream, int via_pointer) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)lds_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(lds_probe_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}

extern "C" int lds_pattern(uint32_t *errors, uint32_t tag, int blocks, int threads, size_t lds, int spin, void *stream) {
    (void)hipFuncSetAttribute((const void *)lds_pattern_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(lds_pattern_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, errors, tag, (int)(lds / 4), spin);
    return (int)hipGetLastError();
}
