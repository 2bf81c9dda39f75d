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
#define GA_NAME gemm_asm_256_bias
#define GA_WN 256
#define GA_INC "build/gemm_asm_256_0.inc"
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

#define GA_NAME gemm_asmp_224_gelu
#define GA_INC "build/gemm_asmp_224_1.inc"
#include "gemm_asm_kernel_p.inl"
#undef GA_NAME
#undef GA_INC
#define GA_NAME gemm_asmp_192_bias
#define GA_INC "build/gemm_asmp_192_0.inc"
#include "gemm_asm_kernel_p.inl"
#undef GA_NAME
#undef GA_INC
#define GA_NAME gemm_asmp_128_bias
#define GA_INC "build/gemm_asmp_128_0.inc"
#include "gemm_asm_kernel_p.inl"
#undef GA_NAME
#undef GA_INC
#define GA_NAME gemm_asmp_128_gate_res
#define GA_INC "build/gemm_asmp_128_2.inc"
#include "gemm_asm_kernel_p.inl"
#undef GA_NAME
#undef GA_INC
#define GA_NAME gemm_asmp_128_res
#define GA_INC "build/gemm_asmp_128_3.inc"
#include "gemm_asm_kernel_p.inl"
#undef GA_NAME
#undef GA_INC

#define GA_NAME gemm_asm_128_bias_ssq
#define GA_WN 128
#define GA_INC "build/gemm_asm_128_5.inc"
#define GA_SSQ 1
#include "gemm_asm_kernel.inl"
#undef GA_SSQ
#undef GA_NAME
#undef GA_WN
#undef GA_INC

#define GA_NAME gemm_asm_128_partial
#define GA_WN 128
#define GA_INC "build/gemm_asm_128_4.inc"
#define GA_PARTIAL 1
#include "gemm_asm_kernel.inl"
#undef GA_PARTIAL
#undef GA_NAME
#undef GA_WN
#undef GA_INC

#define GA_NAME gemm_asmq_224_gelu
#define GA_WN 224
#define GA_INC "build/gemm_asmq_224_1.inc"
#define GA_I8 1
#include "gemm_asm_kernel.inl"
#undef GA_I8
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asmq_192_bias
#define GA_WN 192
#define GA_INC "build/gemm_asmq_192_0.inc"
#define GA_I8 1
#include "gemm_asm_kernel.inl"
#undef GA_I8
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asmq_128_bias
#define GA_WN 128
#define GA_INC "build/gemm_asmq_128_0.inc"
#define GA_I8 1
#include "gemm_asm_kernel.inl"
#undef GA_I8
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asmq_128_gate_res
#define GA_WN 128
#define GA_INC "build/gemm_asmq_128_2.inc"
#define GA_I8 1
#include "gemm_asm_kernel.inl"
#undef GA_I8
#undef GA_NAME
#undef GA_WN
#undef GA_INC
#define GA_NAME gemm_asmq_128_res
#define GA_WN 128
#define GA_INC "build/gemm_asmq_128_3.inc"
#define GA_I8 1
#include "gemm_asm_kernel.inl"
#undef GA_I8
#undef GA_NAME
#undef GA_WN
#undef GA_INC

int g_gemm_asm_persistent = 1;      // tuning key gemm_asm bit 5 (set from gemm.hip's ll_set_tuning)
static int gemm_asm_cus() {
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus[dev] = n > 0 ? n : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}

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
  if (epilogue == LL_EPI_BIAS && M <= 1024 && N >= 16384 && N % 256 == 0) return 256;      // umT5's gated FFN (512 x 20480 x 4096): 160 tiles of 256 x 256 in ONE round,
                                                                                             // half the L2 bytes per FLOP of the 128-wide kernel (which is L2-bound at ~29 B/clk/CU)
  if (N % 128 == 0 && (N <= 2048 || M <= 1024) &&      // wide outputs of few rows (umT5's gated FFN, 512 x 20480): the HIP choice there is 256 x 128 as well
      (epilogue == LL_EPI_BIAS || epilogue == LL_EPI_BIAS_GATE_RES || epilogue == LL_EPI_BIAS_RES)) return 128;
  return 0;
}

// 1 = launched; 0 = shape / epilogue not covered here (the caller takes the HIP kernels); < 0 = an LL_ERR_* code (attribute / launch failed)
int gemm_asm_launch(const bf16* x, const bf16* w, bf16* out, int M, int N, int K, int ldx, int ldo, int epilogue, const EpiArgs& ea,
                    int gm, hipStream_t s) {
  const bool has_v = ea.v_out != nullptr;
  const bool v_ok = has_v && ea.v_L == M && ea.v_col0 % 192 == 0 && ea.v_C > 0;
  const int wn = gemm_asm_width(M, N, K, ldx, epilogue, ea.sx == nullptr && ea.mod == nullptr, has_v, v_ok, ea.frame_len);
  if (!wn) return 0;
  const void* fn = wn == 224 ? (const void*)gemm_asm_224_gelu
                   : wn == 256 ? (const void*)gemm_asm_256_bias
                   : wn == 192 ? (const void*)gemm_asm_192_bias
                   : epilogue == LL_EPI_BIAS ? (const void*)gemm_asm_128_bias
                   : epilogue == LL_EPI_BIAS_GATE_RES ? (const void*)gemm_asm_128_gate_res : (const void*)gemm_asm_128_res;
  const int lds = 3 * wn * 128 + 4 * 2 * 8192;      // gen/gemm_asm_gen.py Cfg.lds_bytes: 3 W slots of WN rows x 128 B + 2 X units of 8 KiB per wave
  const int ntm = (M + 255) / 256, ntn = N / wn;
  // persistent form (tuning key gemm_asm bit 5): a launch with more tiles than CUs runs ONE workgroup per CU that walks its tiles and
  // stages the next tile's first pieces under the current epilogue (FFN1: 760 tiles, QKV: 456, the recache forward's N = 1536
  // linears: 888) -- one pipeline fill per launch instead of one per round
  const int cus = gemm_asm_cus() & ~7;
  const void* pfn = wn == 224 ? (const void*)gemm_asmp_224_gelu
                    : wn == 192 ? (const void*)gemm_asmp_192_bias
                    : wn != 128 ? nullptr
                    : epilogue == LL_EPI_BIAS ? (const void*)gemm_asmp_128_bias
                    : epilogue == LL_EPI_BIAS_GATE_RES ? (const void*)gemm_asmp_128_gate_res : (const void*)gemm_asmp_128_res;
  const bool persistent = g_gemm_asm_persistent && pfn != nullptr && cus >= 8 && ntm * ntn > cus;
  if (persistent) fn = pfn;
  if (int rc = ll_lds_attr(fn, lds)) return rc;
  const bf16* gate = epilogue == LL_EPI_BIAS_GATE_RES ? ea.e + (size_t)ea.gate_idx * N : nullptr;
  const int gstride = ea.nmod * N * 2;
  bf16* v_out = ea.v_out;
  int v_col0 = ea.v_col0, v_C = ea.v_C, v_shift = ea.v_write_start - ea.v_roped_offset, v_lo = ea.v_roped_offset,
      v_hi = ea.v_roped_offset + ea.v_write_len;
  void* args[] = {(void*)&x, (void*)&w, (void*)&ea.bias, (void*)&out, (void*)&ea.res, (void*)&gate, (void*)&M, (void*)&N, (void*)&K,
                  (void*)&ldx, (void*)&ldo, (void*)&ea.frame_len, (void*)&gstride, (void*)&ntm, (void*)&ntn, (void*)&gm,
                  (void*)&v_out, (void*)&v_col0, (void*)&v_C, (void*)&v_shift, (void*)&v_lo, (void*)&v_hi};
  if (hipLaunchKernel(fn, dim3(persistent ? cus : ntm * ntn), dim3(256), args, (size_t)lds, s) != hipSuccess) return ll_check_launch("gemm_asm");
  return 1;
}

// The bias kernel that also leaves per-row sums of squares of its outputs: ssq[N / 128][M] fp32 (plane = n-tile).  1 = launched,
// 0 = not covered, < 0 = error.
int gemm_asm_ssq_launch(const bf16* x, const bf16* w, const bf16* bias, bf16* out, float* ssq, int M, int N, int K, int ldx, int ldo, int gm,
                        hipStream_t s) {
  if (gemm_asm_width(M, N, K, ldx, LL_EPI_BIAS, true, false, false, 0) != 128) return 0;
  const void* fn = (const void*)gemm_asm_128_bias_ssq;
  const int lds = 3 * 128 * 128 + 4 * 2 * 8192;
  if (int rc = ll_lds_attr(fn, lds)) return rc;
  const int ntm = (M + 255) / 256, ntn = N / 128;
  const bf16* nullb = nullptr;
  bf16* nov = nullptr;
  int zero = 0;
  void* args[] = {(void*)&x, (void*)&w, (void*)&bias, (void*)&out, (void*)&nullb, (void*)&nullb, (void*)&M, (void*)&N, (void*)&K,
                  (void*)&ldx, (void*)&ldo, (void*)&zero, (void*)&zero, (void*)&ntm, (void*)&ntn, (void*)&gm,
                  (void*)&nov, (void*)&zero, (void*)&zero, (void*)&zero, (void*)&zero, (void*)&zero, (void*)&ssq};
  if (hipLaunchKernel(fn, dim3(ntm * ntn), dim3(256), args, (size_t)lds, s) != hipSuccess) return ll_check_launch("gemm_asm_128_bias_ssq");
  return 1;
}

const char* gemm_asm_plan(int M, int N, int wn, int epilogue, char* out, int cap, bool i8) {
  const char* tail = wn == 224 ? "gelu" : wn == 256 ? "bias" : epilogue == LL_EPI_BIAS ? "bias" : epilogue == LL_EPI_BIAS_GATE_RES ? "gate_res" : "res";      // wn == 192: bias
  const int tiles = ((M + 255) / 256) * (N / wn), cus = gemm_asm_cus() & ~7;
  if (!i8 && g_gemm_asm_persistent && wn != 256 && cus >= 8 && tiles > cus)
    snprintf(out, (size_t)cap, "gemm_asmp_%d_%s<bf16> tile 256x%d (4 waves x 64 rows, one wave per SIMD, generated schedule), %d persistent workgroups "
             "walk %d tiles, next tile staged under the epilogue", wn, tail, wn, cus, tiles);
  else
    snprintf(out, (size_t)cap, "gemm_asm%s_%d_%s<%s> tile 256x%d (4 waves x 64 rows, one wave per SIMD, generated schedule), %d workgroups", i8 ? "q" : "", wn,
             tail, i8 ? "i8" : "bf16", wn, tiles);
  return out;
}

// tile width gemm_asm_launch_i8 takes for a W8A8 call (0 = not covered): the same rule as in that launcher
int gemm_asm_width_i8(int M, int N, int K, int epilogue, bool plain, bool has_v, bool v_ok, int frame_len) {
  if (!plain || M <= 0 || K % 128 != 0 || K < 512 || (long long)256 * K >= 0x7fffffffLL) return 0;
  if (epilogue == LL_EPI_BIAS_GATE_RES && frame_len <= 0) return 0;
  if (has_v) return (v_ok && epilogue == LL_EPI_BIAS && N % 192 == 0) ? 192 : 0;
  if (epilogue == LL_EPI_BIAS_GELU) return N % 224 == 0 ? 224 : 0;
  if (epilogue == LL_EPI_BIAS && N > 2048 && N % 192 == 0) return 192;
  if (N % 128 == 0 && N <= 2048) return 128;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Small-M split-K (umT5's linears at 512 tokens, wan/modules/t5.py:65-117; the text K/V projections): a grid of
// ceil(M / 256) x (N / 128) tiles fills a quarter of the device, so K is cut into `splits` ranges -- each workgroup of
// gemm_asm_128_partial writes its tile's fp32 accumulators to workspace[split][M][N], and one elementwise pass sums the ranges in
// a fixed order and applies the epilogue (bias, or bias + residual) with gemm_common.h's rounding points.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_ksplit_reduce_kernel(const float* __restrict__ part, int splits, int M, int N,
                                                                 const bf16* __restrict__ bias, const bf16* __restrict__ res,
                                                                 bf16* __restrict__ out, int ldo) {
  const int n8 = N / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)M * n8) return;
  const int m = (int)(idx / n8), n = (int)(idx - (long long)m * n8) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int s = 0; s < splits; ++s) {                  // fixed order: bit-identical run to run
    const float4* p = reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * N + n);
    float4 a = p[0], b = p[1];
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
  }
  bf16x8 bv = *reinterpret_cast<const bf16x8*>(bias + n), o;
  bf16x8 rv;
  if (EPI == LL_EPI_BIAS_RES) rv = *reinterpret_cast<const bf16x8*>(res + (size_t)m * ldo + n);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bf16 v = (bf16)(acc[j] + (float)bv[j]);
    o[j] = EPI == LL_EPI_BIAS_RES ? (bf16)((float)rv[j] + (float)v) : v;
  }
  *reinterpret_cast<bf16x8*>(out + (size_t)m * ldo + n) = o;
}

// The same pass for the residual stream of umT5 (t5.py:119-160: x = x + linear(...); h = T5LayerNorm(x)): one workgroup per row sums
// the K-ranges, adds bias and residual (x_new, written to `out`) and applies the T5 RMSNorm to that row at once (h_out) -- the
// arithmetic, its order and its rounding points are gemm_ksplit_reduce_kernel's followed by t5_rmsnorm_kernel's (t5.hip: thread t
// owns columns 8 t + 2048 i, t5_block_sum), so the pair of outputs is bit-identical to the two launches it replaces.
// N <= 4096 (two column groups per thread).
__global__ __launch_bounds__(256) void gemm_ksplit_reduce_norm_kernel(const float* __restrict__ part, int splits, int M, int N,
                                                                      const bf16* __restrict__ bias, const bf16* __restrict__ res,
                                                                      bf16* __restrict__ out, int ldo, const bf16* __restrict__ nw,
                                                                      float eps, bf16* __restrict__ h_out) {
  __shared__ float sh[4];
  const int m = blockIdx.x;
  bf16x8 xn[2];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = threadIdx.x * 8 + 2048 * i;
    if (n < N) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int s = 0; s < splits; ++s) {
        const float4* p = reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * N + n);
        float4 a = p[0], b = p[1];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      }
      bf16x8 bv = *reinterpret_cast<const bf16x8*>(bias + n), rv = *reinterpret_cast<const bf16x8*>(res + (size_t)m * ldo + n), o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bf16 v = (bf16)(acc[j] + (float)bv[j]);
        o[j] = (bf16)((float)rv[j] + (float)v);
        ss += (float)o[j] * (float)o[j];
      }
      xn[i] = o;
      *reinterpret_cast<bf16x8*>(out + (size_t)m * ldo + n) = o;
    }
  }
  ss = t5_block_sum(ss, sh);
  const float r = rsqrtf(ss / (float)N + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = threadIdx.x * 8 + 2048 * i;
    if (n < N) {
      bf16x8 g = *reinterpret_cast<const bf16x8*>(nw + n), o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)g[j] * rbf((float)xn[i][j] * r));
      *reinterpret_cast<bf16x8*>(h_out + (size_t)m * N + n) = o;
    }
  }
}

// K-ranges the small-M path cuts this call into on a device of `cus` compute units; 0 = not taken.  The NUMBER of ranges depends
// on N and K only (the order of the fp32 sum must not change with the batch: M = 512 and M = 1024 give the same bits per row);
// M only decides whether the path is taken at all (the plain kernels would fill less than half of the device).
int gemm_ksplit_splits(int M, int N, int K, int cus) {
  if (M <= 0 || N <= 0 || N % 128 != 0 || K % 64 != 0 || K < 1024 || cus <= 0) return 0;
  const long long tiles = (long long)((M + 255) / 256) * (N / 128);
  if (tiles * 2 > cus) return 0;                        // the plain kernels already fill half the device
  const int nk = K / 64;
  int S = (int)((long long)cus * 64 / N);               // fills the device at 512 rows (tiles = N / 64 there)
  if (S > 4) S = 4;
  if (S > nk / 8) S = nk / 8;                           // at least 8 K-steps per range: the pipeline fill is ~3
  while (S >= 2 && (S - 1) * ((nk + S - 1) / S) >= nk) --S;      // every range non-empty
  if (S < 2 || tiles * S > 2 * cus) return 0;
  return S;
}

// 1 = launched (two launches), 0 = not covered, < 0 = an LL_ERR_* code
int gemm_asm_ksplit_launch(const bf16* x, const bf16* w, const bf16* bias, bf16* out, int M, int N, int K, int ldx, int ldo,
                           int epilogue, const bf16* res, float* workspace, int splits, int gm, hipStream_t s, const bf16* norm_w,
                           float eps, bf16* h_out) {
  if (splits < 2 || (epilogue != LL_EPI_BIAS && epilogue != LL_EPI_BIAS_RES) || (ldx % 8) != 0 || (ldo % 8) != 0) return 0;
  if ((long long)256 * ldx * 2 >= 0x7fffffffLL || (long long)256 * K * 2 >= 0x7fffffffLL) return 0;
  const int lds = 3 * 128 * 128 + 4 * 2 * 8192;
  if (int rc = ll_lds_attr((const void*)gemm_asm_128_partial, lds)) return rc;
  const int ntm = (M + 255) / 256, ntn = N / 128, per = (K / 64 + splits - 1) / splits;
  const bf16* nullb = nullptr;
  bf16* ws = (bf16*)workspace;
  bf16* nov = nullptr;
  int zero = 0, ldw = N, flen = 0;
  void* args[] = {(void*)&x, (void*)&w, (void*)&nullb, (void*)&ws, (void*)&nullb, (void*)&nullb, (void*)&M, (void*)&N, (void*)&K,
                  (void*)&ldx, (void*)&ldw, (void*)&flen, (void*)&zero, (void*)&ntm, (void*)&ntn, (void*)&gm,
                  (void*)&nov, (void*)&per, (void*)&zero, (void*)&zero, (void*)&zero, (void*)&zero};
  if (hipLaunchKernel((const void*)gemm_asm_128_partial, dim3(ntm * ntn * splits), dim3(256), args, (size_t)lds, s) != hipSuccess)
    return ll_check_launch("gemm_asm_128_partial");
  const long long threads = (long long)M * (N / 8);
  dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (norm_w != nullptr)       // (the caller checked: bias + residual, N <= 4096)
    hipLaunchKernelGGL(gemm_ksplit_reduce_norm_kernel, dim3(M), block, 0, s, (const float*)workspace, splits, M, N, bias, res,
                       out, ldo, norm_w, eps, h_out);
  else if (epilogue == LL_EPI_BIAS_RES)
    hipLaunchKernelGGL((gemm_ksplit_reduce_kernel<LL_EPI_BIAS_RES>), grid, block, 0, s, (const float*)workspace, splits, M, N, bias, res, out, ldo);
  else
    hipLaunchKernelGGL((gemm_ksplit_reduce_kernel<LL_EPI_BIAS>), grid, block, 0, s, (const float*)workspace, splits, M, N, bias, res, out, ldo);
  return 1;
}

// W8A8 form of gemm_asm_launch (ll_gemm_w8a8 / ll_gemm_w8a8_qkv): int8 operands with row strides K, fp32 scales sx [M] / sw [N];
// the same tile widths, epilogues and V-cache redirect.  Integer sums are exact and the epilogue applies gemm_common.h's operations
// in its order, so the results are bit-identical to the HIP W8A8 kernels.  1 = launched, 0 = not covered.
int gemm_asm_launch_i8(const int8_t* x, const int8_t* w, bf16* out, int M, int N, int K, int ldo, int epilogue, const EpiArgs& ea,
                       int gm, hipStream_t s) {
  if (ea.sx == nullptr || ea.sw == nullptr) return 0;
  const bool has_v = ea.v_out != nullptr;
  const int wn = gemm_asm_width_i8(M, N, K, epilogue, ea.mod == nullptr, has_v, has_v && ea.v_L == M && ea.v_col0 % 192 == 0 && ea.v_C > 0,
                                   ea.frame_len);
  if (!wn) return 0;
  const void* fn = wn == 224 ? (const void*)gemm_asmq_224_gelu
                   : wn == 192 ? (const void*)gemm_asmq_192_bias
                   : epilogue == LL_EPI_BIAS ? (const void*)gemm_asmq_128_bias
                   : epilogue == LL_EPI_BIAS_GATE_RES ? (const void*)gemm_asmq_128_gate_res : (const void*)gemm_asmq_128_res;
  const int lds = 3 * wn * 128 + 4 * 2 * 8192;
  if (int rc = ll_lds_attr(fn, lds)) return rc;
  const int ntm = (M + 255) / 256, ntn = N / wn;
  const bf16* gate = epilogue == LL_EPI_BIAS_GATE_RES ? ea.e + (size_t)ea.gate_idx * N : nullptr;
  const int gstride = ea.nmod * N * 2;
  bf16* v_out = ea.v_out;
  int v_col0 = ea.v_col0, v_C = ea.v_C, v_shift = ea.v_write_start - ea.v_roped_offset, v_lo = ea.v_roped_offset,
      v_hi = ea.v_roped_offset + ea.v_write_len, ldx = K;
  void* args[] = {(void*)&x, (void*)&w, (void*)&ea.bias, (void*)&out, (void*)&ea.res, (void*)&gate, (void*)&M, (void*)&N, (void*)&K,
                  (void*)&ldx, (void*)&ldo, (void*)&ea.frame_len, (void*)&gstride, (void*)&ntm, (void*)&ntn, (void*)&gm,
                  (void*)&v_out, (void*)&v_col0, (void*)&v_C, (void*)&v_shift, (void*)&v_lo, (void*)&v_hi, (void*)&ea.sx, (void*)&ea.sw};
  if (hipLaunchKernel(fn, dim3(ntm * ntn), dim3(256), args, (size_t)lds, s) != hipSuccess) return ll_check_launch("gemm_asm");
  return 1;
}
