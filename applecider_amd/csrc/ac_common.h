// Internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/applecider_hip.h"

#define AC_WAVE 64

#define AC_CHECK_LAUNCH()                              \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return -(int)e__ - 2000; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// 16-bit operand format of THIS BUILD of the library.  The training library
// (libapplecider_hip.so) rounds matrix-core operands and 16-bit hand-overs to bfloat16: 8 exponent
// bits, so activations and gradients never leave the range.  `make` also builds the same sources with
// -DAC_HALF_F16 into libapplecider_hip_f16.so, the INFERENCE library of BASELINE configs[4]
// ("inference-only fused forward, fp16"): IEEE fp16 operands on v_mfma_f32_32x32x16_f16 — the same
// matrix-core rate, 3 more mantissa bits, forward only (fp16 gradients would underflow).
// Every conversion in the kernels goes through ac_f2h / ac_h2f / AC_MFMA16.
// ---------------------------------------------------------------------------------------------
#include <hip/hip_bf16.h>
typedef short ac_s16x8 __attribute__((ext_vector_type(8)));
#ifdef AC_HALF_F16
typedef _Float16 ac_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short ac_f2h(float x) {
    return __builtin_bit_cast(unsigned short, (_Float16)x);       // round to nearest even
}
__device__ __forceinline__ float ac_h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
#define AC_MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ac_h8, a), __builtin_bit_cast(ac_h8, b), c, 0, 0, 0)
#define AC_MFMA16S(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ac_h8, a), __builtin_bit_cast(ac_h8, b), c, 0, 0, 0)
#define AC_HALF_NAME "f16"
#else
__device__ __forceinline__ unsigned short ac_f2h(float x) {
    return __builtin_bit_cast(unsigned short, __float2bfloat16(x));   // v_cvt_pk_bf16_f32, RNE
}
__device__ __forceinline__ float ac_h2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
#define AC_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// 16x16x32 shape (16 cycles): lane l holds A[row l&15][k = 8(l>>4)..+7], B[k = 8(l>>4)..+7][col l&15],
// D[row 4(l>>4)+e][col l&15].  Same FLOP per cycle as 32x32x16; the chip holds a higher clock on it.
#define AC_MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define AC_HALF_NAME "bf16"
#endif

// x -> (hi, lo) for a PAIR of floats, each packed as 2 x 16 bit (first element in the low half): hi = round16(x),
// lo = round16(x - hi), both round-to-nearest-even - the split of the "bf16x3" arithmetic.  bf16 build: 3 VALU
// instructions per element (one packed convert yields both hi halves, a shift and a mask turn them back into floats, one
// packed convert for lo); the element-wise form (convert, shift, subtract, convert, and a v_perm to pack) took 4.5, and
// the split sits in loops that are bound by VALU issue (wave64 without packed fp32: 4 cycles per instruction).
// Bit-identical to the element-wise form: the same hardware convert.
__device__ __forceinline__ void ac_split_pair(float a, float b, unsigned &hi, unsigned &lo) {
#ifdef AC_HALF_F16
    const unsigned short ha = ac_f2h(a), hb = ac_f2h(b);
    hi = (unsigned)ha | ((unsigned)hb << 16);
    lo = (unsigned)ac_f2h(a - ac_h2f(ha)) | ((unsigned)ac_f2h(b - ac_h2f(hb)) << 16);
#else
    typedef float ac_f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 ac_b2 __attribute__((ext_vector_type(2)));
    const ac_f2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, ac_b2));
    const ac_f2 l = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, ac_b2));
#endif
}

static inline bool ac_aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

__device__ __forceinline__ float ac_gelu(float x) {
    // exact erf GELU (torch default approximate='none')
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float ac_gelu_grad(float x) {
    const float kInvSqrt2Pi = 0.39894228040143267794f;
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}
// Rational erf (odd polynomial / even polynomial on |x| <= 4, the form Eigen and XLA use for
// float): max abs error 4.5e-7 against the exact function, ~15 FMAs and one reciprocal instead of
// the branchy library routine.  Used by the bf16-mode GEMM epilogues, where GELU over a 4C-wide hidden layer cost as
// much as the product itself, and by the fused SpectraNet tail (split-bf16 mode); the exact-fp32 mode keeps erff.
__device__ __forceinline__ float ac_erf_fast(float x) {
    x = fminf(fmaxf(x, -4.f), 4.f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    // v_rcp_f32 (1 ulp), not the correctly rounded quotient: that is a ten-instruction sequence per element, and the
    // fused SpectraNet tail kernels are bound by VALU issue (non-packed fp32: 4 cycles per wave instruction)
    return x * p * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float ac_gelu_fast(float x) {
    return 0.5f * x * (1.0f + ac_erf_fast(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float ac_gelu_grad_fast(float x) {
    const float kInvSqrt2Pi = 0.39894228040143267794f;
    const float cdf = 0.5f * (1.0f + ac_erf_fast(x * 0.70710678118654752440f));
    return cdf + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float ac_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float ac_act(float v, int kind) {
    switch (kind) {
        case AC_ACT_GELU: return ac_gelu(v);
        case AC_ACT_GELU_FAST: return ac_gelu_fast(v);
        case AC_ACT_RELU: return v > 0.f ? v : 0.f;
        case AC_ACT_SIGMOID: return ac_sigmoid(v);
        case AC_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative from aux: GELU/RELU take the pre-activation, SIGMOID/TANH the output
__device__ __forceinline__ float ac_dact(float aux, int kind) {
    switch (kind) {
        case AC_ACT_GELU: return ac_gelu_grad(aux);
        case AC_ACT_GELU_FAST: return ac_gelu_grad_fast(aux);
        case AC_ACT_RELU: return aux > 0.f ? 1.f : 0.f;
        case AC_ACT_SIGMOID: return aux * (1.f - aux);
        case AC_ACT_TANH: return 1.f - aux * aux;
        default: return 1.f;
    }
}

__device__ __forceinline__ float ac_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ac_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based keep/drop decision shared by forward and backward (dropout): a 32-bit avalanche
// hash (two multiply-xorshift rounds, the murmur3 finaliser) of (seed, index) -> uniform in [0,1).
// 32-bit arithmetic on purpose: inside a GEMM epilogue the generator runs at 2 waves per SIMD, and
// a 64-bit splitmix (three 64x64 multiplies = ~30 VALU ops per element) cost more than the product.
__device__ __forceinline__ unsigned ac_hash32(uint64_t seed, uint64_t idx) {
    unsigned h = (unsigned)idx * 0x9E3779B1u + (unsigned)(idx >> 32) * 0x85EBCA77u;
    h ^= (unsigned)seed;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h += (unsigned)(seed >> 32);
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float ac_rand01(uint64_t seed, uint64_t idx) {
    return (float)(ac_hash32(seed, idx) >> 8) * (1.0f / 16777216.0f);
}

// Dropout on ATTENTION WEIGHTS (nn.MultiheadAttention's dropout, HyraxBaselineCLS.py:24-31): the keep / drop decision
// of (query row, key) is one mixing round over the XOR of two per-token words,
//     keep  <=>  mix(word_q ^ word_k) >= p * 2^32,
// word_q = ac_att_word(seed, (b H + h) T + q, 0), word_k = ac_att_word(seed, (b H + h) T + k, 1): a kernel computes
// the words once per token (T per workgroup, kept on the lane or in a small LDS table) and pays ONE 32-bit multiply
// per score element instead of the three of ac_hash32 plus its 64-bit index arithmetic - the attention kernels are
// bound by VALU issue and drew 2-3 masks per element and step (forward, two backward phases).  Bernoulli(1 - p) to
// 2^-32; the scalar kernels of ac_seq.hip and the matrix-core kernels of ac_attn.hip draw identical masks.
__device__ __forceinline__ unsigned ac_att_word(uint64_t seed, uint64_t token, unsigned is_key) {
    return ac_hash32(seed + (is_key ? 0x9E3779B97F4A7C15ull : 0ull), token);
}
__device__ __forceinline__ unsigned ac_att_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}
__device__ __forceinline__ bool ac_att_keep(unsigned word_q, unsigned word_k, unsigned thr) {
    unsigned h = word_q ^ word_k;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    return h >= thr;
}

// Device-resident step counter: every entry point that draws random numbers takes an optional
// `step` pointer (uint64 in HBM) and mixes step[0] into the seed its launch was given, so a captured
// hipGraph of a whole training step draws a new mask at every replay (the host-side seed baked into
// the graph stays the same; ac_step_advance bumps the counter inside the graph).  Null: the seed is
// used as given; a counter at 0 likewise.
__device__ __forceinline__ uint64_t ac_step_seed(uint64_t seed, const uint64_t *stepp) {
    return stepp ? seed + stepp[0] * 0x9E3779B97F4A7C15ull : seed;
}

__device__ __forceinline__ int64_t ac_rowaddr(const ac_rowmap &m, int r) {
    if (m.r1 == 0) return (int64_t)r * m.s3;
    int q1 = r / m.r1;
    int rem = r - q1 * m.r1;
    int q2 = rem / m.r2;
    int q3 = rem - q2 * m.r2;
    return (int64_t)q1 * m.s1 + (int64_t)q2 * m.s2 + (int64_t)q3 * m.s3;
}

// Explicit global-address-space loads.  Pointers that travel through by-value kernel-argument
// structs are not always inferred as global by hipcc; a FLAT load counts on lgkmcnt as well as
// vmcnt, so the `s_waitcnt lgkmcnt(0)` in front of the MFMAs (meant for the LDS fragment reads)
// also waits for every prefetch in flight and serialises the whole software pipeline.
template <typename V, typename T>
__device__ __forceinline__ V ac_gload(const T *p) {
    typedef __attribute__((address_space(1))) const V gV;
    return *(gV *)(p);
}
