// Hand-scheduled self-attention for long contiguous key ranges on gfx950: 4 waves x 64 query rows, one wave per SIMD, the whole
// 512-register file owned by the kernel text.  The body is GENERATED (gen/attn_asm_gen.py -> build/attn_asm_body.inc: structure,
// register map, pipeline and the checks it passes on the CPU are described there); this file only computes the workgroup's scalar
// arguments, pins them to the registers the text expects and launches.  Replaces attention() + the sink/window gather
// (wan/modules/attention.py:43-197, causal_model.py:331-360) for the launches ll_flash_attn routes here (tuning key attn_asm).
#include "common.h"

#define ASM_KT 64

#define LL_ASM_NAME flash_attn_asm_kernel
#define LL_ASM_INC "build/attn_asm_body.inc"
#include "attention_asm_kernel.inl"
#undef LL_ASM_NAME
#undef LL_ASM_INC
#define LL_ASM_NAME flash_attn_asm_qn_kernel
#define LL_ASM_INC "build/attn_asm_body_qn.inc"
#define LL_ASM_QNORM 1
#include "attention_asm_kernel.inl"
#undef LL_ASM_QNORM
#undef LL_ASM_NAME
#undef LL_ASM_INC
// QNORM form: q = the raw q projection [B, Lq, H * 128] with per-(plane, row) sums of squares ssq[nplanes][B * Lq]; the kernel
// applies WanRMSNorm (weight nw, eps) to its 256 rows x one head in the prologue (wan/modules/model.py:78-86,172)
int flash_attn_asm_qn_launch(const bf16* q, const float* ssq, int nplanes, const bf16* nw, float eps, const bf16* k, const bf16* v,
                             bf16* out, int B, int Lq, int H, int ldq, int ldo, int ldk, long long k_batch_stride, int kstart, int nkeys,
                             float c, int xcd, hipStream_t stream) {
  if (int rc = ll_lds_attr((const void*)flash_attn_asm_qn_kernel, 128 * 1024)) return rc;
  const int nqt = (Lq + 255) / 256;
  hipLaunchKernelGGL(flash_attn_asm_qn_kernel, dim3(nqt * H, 1, B), dim3(256), 128 * 1024, stream, q, k, v, out, Lq, ldq, ldo, ldk,
                     k_batch_stride, kstart, nkeys, c, nqt, xcd, ssq, nplanes, (long long)B * Lq, nw, 1.0f / (float)(H * 128), eps);
  return ll_check_launch("ll_flash_attn_qnorm(asm)");
}

int flash_attn_asm_launch(const bf16* q, const bf16* k, const bf16* v, bf16* out, int B, int Lq, int H, int ldq, int ldo, int ldk,
                          long long k_batch_stride, int kstart, int nkeys, float c, int xcd, int form, hipStream_t stream) {
  if (int rc = ll_lds_attr((const void*)flash_attn_asm_kernel, 128 * 1024)) return rc;
  const int nqt = (Lq + 255) / 256;
  (void)form;
  hipLaunchKernelGGL(flash_attn_asm_kernel, dim3(nqt * H, 1, B), dim3(256), 128 * 1024, stream, q, k, v, out, Lq, ldq, ldo, ldk,
                       k_batch_stride, kstart, nkeys, c, nqt, xcd);
  return ll_check_launch("ll_flash_attn(asm)");
}
