// Hand-scheduled bf16 GEMMs for the block linears on gfx950: 4 waves per 256 x WN output tile, one wave per SIMD, accumulators in
// AGPRs, generated body (gen/gemm_asm_gen.py -> build/gemm_asm_<WN>_<EPI>.inc; structure, pipeline and CPU checks are described
// there).  This file computes each workgroup's tile and scalar arguments, pins them to the registers the text expects and launches.
// Replaces gemm_kernel_v5 / v2 for the shapes ll_gemm_bf16 routes here (tuning key gemm_asm); rounding points as gemm_common.h.
#include "gemm_common.h"

#define GA_NAME gemm_asm_224_gelu
#define GA_WN 224
#define GA_INC "build/gemm_asm_224_1.inc"
#include "gemm_asm_kernel.inl"
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asm_192_bias
#define GA_WN 192
#define GA_INC "build/gemm_asm_192_0.inc"
#include "gemm_asm_kernel.inl"
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asm_128_bias
#define GA_WN 128
#define GA_INC "build/gemm_asm_128_0.inc"
#include "gemm_asm_kernel.inl"
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asm_128_gate_res
#define GA_WN 128
#define GA_INC "build/gemm_asm_128_2.inc"
#include "gemm_asm_kernel.inl"
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asm_128_res
#define GA_WN 128
#define GA_INC "build/gemm_asm_128_3.inc"
#include "gemm_asm_kernel.inl"
#undef GA_NAME
#undef GA_WN
#undef GA_INC

// tile width of the generated kernel that covers this call, 0 = none (the caller takes the HIP kernels).
// plain = no int8 scales, no per-batch modulation vector; v_ok = no V-cache output, or one the 192-wide kernel can redirect per
// tile (one batch element, the V third starting on a tile boundary)
int gemm_asm_width(int M, int N, int K, int ldx, int epilogue, bool plain, bool has_v, bool v_ok, int frame_len) {
  if (!plain || M <= 0 || K % 64 != 0 || K < 256 || (ldx % 8) != 0) return 0;
  if (epilogue == LL_EPI_BIAS_GATE_RES && frame_len <= 0) return 0;
  if ((long long)256 * ldx * 2 >= 0x7fffffffLL || (long long)256 * K * 2 >= 0x7fffffffLL) return 0;
  if (has_v) return (v_ok && epilogue == LL_EPI_BIAS && N % 192 == 0) ? 192 : 0;
  if (epilogue == LL_EPI_BIAS_GELU) return N % 224 == 0 ? 224 : 0;
  if (epilogue == LL_EPI_BIAS && N > 2048 && N % 192 == 0) return 192;
  if (N % 128 == 0 && N <= 2048 && (epilogue == LL_EPI_BIAS || epilogue == LL_EPI_BIAS_GATE_RES || epilogue == LL_EPI_BIAS_RES)) return 128;
  return 0;
}

// 1 = launched; 0 = shape / epilogue not covered here (the caller takes the HIP kernels)
int gemm_asm_launch(const bf16* x, const bf16* w, bf16* out, int M, int N, int K, int ldx, int ldo, int epilogue, const EpiArgs& ea,
                    int gm, hipStream_t s) {
  const bool has_v = ea.v_out != nullptr;
  const bool v_ok = has_v && ea.v_L == M && ea.v_col0 % 192 == 0 && ea.v_C > 0;
  const int wn = gemm_asm_width(M, N, K, ldx, epilogue, ea.sx == nullptr && ea.mod == nullptr, has_v, v_ok, ea.frame_len);
  if (!wn) return 0;
  const void* fn = wn == 224 ? (const void*)gemm_asm_224_gelu
                   : wn == 192 ? (const void*)gemm_asm_192_bias
                   : epilogue == LL_EPI_BIAS ? (const void*)gemm_asm_128_bias
                   : epilogue == LL_EPI_BIAS_GATE_RES ? (const void*)gemm_asm_128_gate_res : (const void*)gemm_asm_128_res;
  static bool attr[5] = {false, false, false, false, false};
  const int slot = wn == 224 ? 0 : wn == 192 ? 4 : fn == (const void*)gemm_asm_128_bias ? 1 : fn == (const void*)gemm_asm_128_gate_res ? 2 : 3;
  const int lds = 3 * wn * 128 + 4 * 2 * 8192;      // gen/gemm_asm_gen.py Cfg.lds_bytes: 3 W slots of WN rows x 128 B + 2 X units of 8 KiB per wave
  if (!attr[slot]) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr[slot] = true; }
  const int ntm = (M + 255) / 256, ntn = N / wn;
  const bf16* gate = epilogue == LL_EPI_BIAS_GATE_RES ? ea.e + (size_t)ea.gate_idx * N : nullptr;
  const int gstride = ea.nmod * N * 2;
  bf16* v_out = ea.v_out;
  int v_col0 = ea.v_col0, v_C = ea.v_C, v_shift = ea.v_write_start - ea.v_roped_offset, v_lo = ea.v_roped_offset,
      v_hi = ea.v_roped_offset + ea.v_write_len;
  void* args[] = {(void*)&x, (void*)&w, (void*)&ea.bias, (void*)&out, (void*)&ea.res, (void*)&gate, (void*)&M, (void*)&N, (void*)&K,
                  (void*)&ldx, (void*)&ldo, (void*)&ea.frame_len, (void*)&gstride, (void*)&ntm, (void*)&ntn, (void*)&gm,
                  (void*)&v_out, (void*)&v_col0, (void*)&v_C, (void*)&v_shift, (void*)&v_lo, (void*)&v_hi};
  (void)hipLaunchKernel(fn, dim3(ntm * ntn), dim3(256), args, (size_t)lds, s);
  return 1;
}

const char* gemm_asm_plan(int M, int N, int wn, int epilogue, char* out, int cap) {
  const char* tail = wn == 224 ? "gelu" : epilogue == LL_EPI_BIAS ? "bias" : epilogue == LL_EPI_BIAS_GATE_RES ? "gate_res" : "res";      // wn == 192: bias
  snprintf(out, (size_t)cap, "gemm_asm_%d_%s<bf16> tile 256x%d (4 waves x 64 rows, one wave per SIMD, generated schedule), %d workgroups", wn,
           tail, wn, ((M + 255) / 256) * (N / wn));
  return out;
}
