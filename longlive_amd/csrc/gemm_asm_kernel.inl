// included by gemm_asm.hip once per (tile width, epilogue) with GA_NAME / GA_WN / GA_INC defined; GA_PARTIAL: the split-K form that
// stores fp32 accumulators; GA_I8: W8A8 operands (X, W int8 with row strides in BYTES = elements, K-steps of 128, sx / sw scales)
__global__ __launch_bounds__(256, 1) void GA_NAME(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                  const bf16* __restrict__ bias, bf16* __restrict__ Y,
                                                  const bf16* __restrict__ res, const bf16* __restrict__ gate, int M, int N,
                                                  int K, int ldx, int ldo, int frame_len, int gate_stride, int ntm, int ntn,
                                                  int gm, bf16* __restrict__ v_out, int v_col0, int v_C, int v_shift, int v_lo,
                                                  int v_hi
#ifdef GA_I8
                                                  , const float* __restrict__ sx, const float* __restrict__ sw
#endif
#ifdef GA_SSQ
                                                  , float* __restrict__ ssq      // [ntn][M]: row sums of squares of this n-tile's outputs
#endif
                                                  ) {
  int mt, nt;
#ifdef GA_PARTIAL
  // one K-range of a split-K call: logical ids [split][tile], consecutive ids (one XCD) share the K-range and neighbouring panels;
  // v_col0 carries the number of K-steps per split, Y is the fp32 workspace [splits][M][ldo], ldo in floats
  const int ntiles = ntm * ntn, lid = xcd_remap(blockIdx.x, gridDim.x), split = lid / ntiles;
  tile_of(lid - split * ntiles, ntm, ntn, gm, mt, nt);
#else
  tile_of(xcd_remap(blockIdx.x, gridDim.x), ntm, ntn, gm, mt, nt);
#endif
  const int m0 = mt * 256, n0 = nt * (GA_WN);
#ifdef GA_I8
  unsigned long long xb = (unsigned long long)((const char*)X + (size_t)m0 * ldx), wb = (unsigned long long)((const char*)W + (size_t)n0 * K);
  unsigned long long sxb = (unsigned long long)(sx + m0), swb = (unsigned long long)(sw + n0);
#else
  unsigned long long xb = (unsigned long long)(X + (size_t)m0 * ldx), wb = (unsigned long long)(W + (size_t)n0 * K);
#endif
  unsigned long long yb = (unsigned long long)(Y + (size_t)m0 * ldo + n0), bb = (unsigned long long)(bias + n0);
  unsigned long long rb = (unsigned long long)(res ? res + (size_t)m0 * ldo + n0 : Y), gb = (unsigned long long)(gate ? gate + n0 : bias);
#ifdef GA_I8
  unsigned ldx_b = (unsigned)ldx, ldw_b = (unsigned)K, ldo_b = (unsigned)ldo * 2u;
  unsigned rows = (unsigned)(M - m0), cols = (unsigned)(N - n0), nk = (unsigned)(K / 128);
#else
  unsigned ldx_b = (unsigned)ldx * 2u, ldw_b = (unsigned)K * 2u, ldo_b = (unsigned)ldo * 2u;
  unsigned rows = (unsigned)(M - m0), cols = (unsigned)(N - n0), nk = (unsigned)(K / 64);
#endif
  unsigned flen = (unsigned)(frame_len > 0 ? frame_len : 1), gstride = (unsigned)gate_stride, um0 = (unsigned)m0;
  unsigned tid = threadIdx.x, row_lo = 0;
#ifdef GA_PARTIAL
  {
    const int per = v_col0, k0 = split * per, left = K / 64 - k0;
    nk = (unsigned)(left < per ? left : per);
    xb += (unsigned long long)k0 * 128ull; wb += (unsigned long long)k0 * 128ull;
    yb = (unsigned long long)((float*)Y + ((size_t)split * M + m0) * ldo + n0);
    ldo_b = (unsigned)ldo * 4u;
  }
#else
  if (v_out != nullptr && n0 >= v_col0) {
    // a V tile of the fused QKV projection (one batch element): token t -> cache row t + v_shift for v_lo <= t < v_hi (epi_dest)
    const int hi = (M < v_hi ? M : v_hi) - m0, lo = v_lo > m0 ? v_lo - m0 : 0;
    if (hi <= lo) return;                                     // nothing of this tile is stored
    rows = (unsigned)hi; row_lo = (unsigned)lo;
    yb = (unsigned long long)(v_out + ((long long)m0 + v_shift) * (long long)v_C + (n0 - v_col0));
    ldo_b = (unsigned)v_C * 2u;
  }
#endif
#ifdef GA_SSQ
  unsigned long long ssqb = (unsigned long long)(ssq + (size_t)nt * M + m0);
#endif
  asm volatile(
#include GA_INC
      :
      : "{s[8:9]}"(xb), "{s[10:11]}"(wb), "{s[12:13]}"(yb), "{s[14:15]}"(bb), "{s[16:17]}"(rb), "{s[18:19]}"(gb),
        "{s20}"(ldx_b), "{s21}"(ldw_b), "{s22}"(ldo_b), "{s23}"(rows), "{s24}"(cols), "{s25}"(nk), "{s26}"(flen),
        "{s27}"(gstride), "{s28}"(um0), "{s29}"(row_lo), "{v0}"(tid)
#ifdef GA_I8
        , "{s[64:65]}"(sxb), "{s[66:67]}"(swb)
      : "memory", "v255", "a255", "s79", "vcc");
#elif defined(GA_SSQ)
        , "{s[64:65]}"(ssqb)
      : "memory", "v255", "a255", "s79", "vcc");
#else
      : "memory", "v255", "a255", "s63", "vcc");
#endif
  __builtin_unreachable();
}
