// Diagnostic (not part of the product): what dynamic LDS does a workgroup really get?  The kernel stores the hardware's
// LDS allocation register (HW_REG_LDS_ALLOC: base and size granules of this workgroup) — eager launch vs the same launch
// captured into a hipGraph.  Built and used by tools/dbg_lds_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lds_probe_kernel(uint32_t *out) {
    extern __shared__ float buf[];
    if (threadIdx.x == 0) {
        // hwreg id 6 = HW_REG_LDS_ALLOC, offset 0, size 32
        out[blockIdx.x] = __builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11));
        buf[0] = 1.f;
    }
}

extern "C" int lds_probe(uint32_t *out, int blocks, int threads, size_t lds, void *stream, int via_pointer) {
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void *)lds_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (via_pointer) {
        auto k = lds_probe_kernel;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, out);
    } else {
        hipLaunchKernelGGL(lds_probe_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, out);
    }
    return (int)hipGetLastError();
}
