// included by attention_asm.hip once per staging form with LL_ASM_NAME / LL_ASM_INC defined
__global__ __launch_bounds__(256, 1) void LL_ASM_NAME(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                const bf16* __restrict__ Vc, bf16* __restrict__ O, int Lq,
                                                                int ldq, int ldo, int ldk, long long k_batch_stride, int kstart,
                                                                int nkeys, float c, int nqt, int xcd_placement
#ifdef LL_ASM_DIAG
                                                                , unsigned long long* dbg
#endif
#ifdef LL_ASM_QNORM
                                                                // Q is the raw projection output; ssq[plane][B * Lq] (fp32) = its per-row sums of squares per
                                                                // 128-column plane, nw[H * 128] the RMSNorm weight (gen/attn_asm_gen.py: QNORM form)
                                                                , const float* __restrict__ ssq, int nplanes, long long plane_stride, const bf16* __restrict__ nw,
                                                                float inv_c, float eps
#endif
) {
  // XCD-aware placement as flash_attn_pipe_kernel: workgroup ids that share an XCD (id % 8) take a contiguous head-major range
  const int b = blockIdx.z;
  int nwg_ = gridDim.x, bid_ = blockIdx.x;
  int qq_ = nwg_ >> 3, rr_ = nwg_ & 7, xcd_ = bid_ & 7;
  int lid_ = bid_;
  if (xcd_placement) {
    int start_ = 0, nbig_ = 0, nsmall_ = 0, mine_ = 0;
#pragma unroll
    for (int k_ = 0; k_ < 8; ++k_) {
      bool big_ = ((k_ + 1) * rr_) / 8 > (k_ * rr_) / 8;
      int id_ = big_ ? nbig_++ : rr_ + nsmall_++;
      mine_ = id_ == xcd_ ? start_ : mine_;
      start_ += big_ ? qq_ + 1 : qq_;
    }
    lid_ = mine_ + (bid_ >> 3);
  }
  const int head = lid_ / nqt, qtile = lid_ % nqt;
  const int q0 = qtile * 256;
  const int nt = (nkeys + ASM_KT - 1) / ASM_KT;
  unsigned long long qb = (unsigned long long)(Q + ((size_t)b * Lq + q0) * ldq + head * 128);
  unsigned long long ob = (unsigned long long)(O + ((size_t)b * Lq + q0) * ldo + head * 128);
  unsigned long long kb = (unsigned long long)(Kc + (size_t)b * k_batch_stride + (size_t)kstart * ldk + head * 128);
  unsigned long long vb = (unsigned long long)(Vc + (size_t)b * k_batch_stride + (size_t)kstart * ldk + head * 128);
  unsigned ldq_b = (unsigned)ldq * 2u, ldo_b = (unsigned)ldo * 2u, ldk_b = (unsigned)ldk * 2u;
  unsigned rows = (unsigned)(Lq - q0 < 256 ? Lq - q0 : 256);
  unsigned unt = (unsigned)nt, lastv = (unsigned)(nkeys - (nt - 1) * ASM_KT);
  unsigned nrec = (unsigned)(nkeys - 1) * ldk_b + 256u;          // bytes from the head's first key to the end of its last key
  unsigned tid = threadIdx.x;
#ifdef LL_ASM_QNORM
  unsigned long long sqb = (unsigned long long)(ssq + (size_t)b * Lq + q0), nwb = (unsigned long long)(nw + head * 128);
  unsigned sq_stride = (unsigned)(plane_stride * 4), unp = (unsigned)nplanes;
#endif
  asm volatile(
#include LL_ASM_INC
      :
      : "{s[8:9]}"(qb), "{s[10:11]}"(ob), "{s[12:13]}"(kb), "{s[14:15]}"(vb), "{s16}"(ldq_b), "{s17}"(ldo_b), "{s18}"(ldk_b),
        "{s19}"(rows), "{s20}"(unt), "{s21}"(lastv), "{s22}"(c), "{s23}"(nrec), "{v0}"(tid)
#ifdef LL_ASM_DIAG
        , "{s[56:57]}"(dbg), "{s58}"(lid_ + gridDim.x * b)
#endif
#ifdef LL_ASM_QNORM
        , "{s[80:81]}"(sqb), "{s82}"(sq_stride), "{s83}"(unp), "{s[84:85]}"(nwb), "{s86}"(inv_c), "{s87}"(eps)
#endif
      : "memory", "v255", "a255", "s79", "s95", "vcc");
  __builtin_unreachable();
}

