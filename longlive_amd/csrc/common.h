// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA bf16, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/longlive_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LL_WAVE 64

// bf16 <-> f32.  (bf16)f lowers to v_cvt_pk_bf16_f32 on gfx950: round-to-nearest-even, NaN stays NaN.
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float f) { return (bf16)f; }
// round an fp32 value to bf16 precision and come back (the reference's intermediate rounding points)
__device__ __forceinline__ float rbf(float f) { return (float)((bf16)f); }

// pairs of elements: a 32-bit word of a bf16 row is two values; fp32 steps on pairs compile to v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32,
// a bf16 rounding point is ONE v_cvt_pk_bf16_f32 per pair (+ shift / mask when the value is used again)
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ f32x2 unpack2(unsigned u) {
  f32x2 r;
  r.x = __builtin_bit_cast(float, u << 16);
  r.y = __builtin_bit_cast(float, u & 0xffff0000u);
  return r;
}
__device__ __forceinline__ unsigned pack2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 rbf2(f32x2 v) { return unpack2(pack2(v)); }
__device__ __forceinline__ f32x2 splat2(float v) {
  f32x2 r = {v, v};
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over a 256-thread workgroup in a FIXED order (per wave by wave_sum, then ((w0 + w1) + w2) + w3); sh = 4 floats of LDS
__device__ __forceinline__ float t5_block_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// gelu(tanh approx) as x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3); identical to 0.5x(1+tanh u).
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  // exp(-2u) via exp2; saturates cleanly: u -> +inf gives x, u -> -inf gives 0 * x -> -0/0 guarded by rcp(inf)=0
  float e = __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ float silu(float x) {
  float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// host-side error plumbing (api.hip)
void ll_set_error(const char* fmt, ...);
#define LL_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ll_set_error(__VA_ARGS__);         \
      return LL_ERR_INVALID_ARG;         \
    }                                    \
  } while (0)

int ll_check_launch(const char* what);
int ll_lds_attr(const void* fn, int bytes);      // dynamic-LDS limit of a kernel, set once per (kernel, device); LL_OK or an error code
void ll_plan_append_knobs(char* out, int cap);   // " [generator knobs: ...]" when the library was built with non-default ASM_* knobs
