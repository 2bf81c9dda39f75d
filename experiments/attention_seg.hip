// NOT COMPILED INTO THE LIBRARY -- kept for the record (round 2).  Drop-in for csrc/attention.hip (it uses that file's helpers:
// store_o_rows, xhalf_*, pack_bf16x2, TILE_B, KT, DIAG_ON) next to flash_attn_pipe_kernel; it was launched from
// flash_attn_pipe_launch under attn_variant = 3 with (SEG_KST + SEG_VST) * TILE_B bytes of dynamic LDS.
// Correct (tests/test_ops_gpu.py attention cases under LL_TUNING_TEST=attn_variant=3), but slower than the ping-pong kernel:
// 4011 cycles per 64-key tile against 3277 (tools/attn_diag.hip; compute segments alone 4 x 810, the LDS-DMA pieces +630).
// See experiments/README.md, round 2, "attention tile-cycle diagnostics".

// =================================================================================================================
// "Segment" schedule (attn_variant 3): compute segments with register-resident operands against load segments.
//
// tools/attn_diag.hip on the ping-pong kernel above: 3276 cycles per 64-key tile where the softmax phase alone takes 955
// (+212 for its four LDS-DMA pieces) and the matrix phase alone 1183 -- the two phases of a SIMD's wave pair overlap badly
// (the softmax wave gets what is left of the vector issue port between the partner's MFMAs, the matrix wave waits on its
// just-in-time LDS fragment reads).  Here the work is cut so that the two waves of a SIMD never want the same unit:
//   compute segment C(h), h = half tile of 32 keys:  16 MFMAs whose A operands are ALREADY in registers --
//         S(h+1) = K(h+1) Q^T (8) and O^T += V(h-1)^T P(h-1)^T (8) -- with the softmax of S(h) (VALU, independent of both)
//         interleaved into the MFMA gaps of the same wave (in-order issue = exact placement); no LDS, no memory
//   load segment L(h):  K(h+2) and V(h) fragments LDS -> registers (8 ds_read_b128 + 16 ds_read_b64_tr_b16), every second one
//         also the wave's four LDS-DMA pieces of tile t+3; no VALU to speak of, no MFMA
// separated by workgroup barriers, waves 4..7 one segment behind waves 0..3: on every SIMD one wave computes while the other
// loads.  Online softmax per half tile.  Rings: K 3 x 16 KiB, V 4 x 16 KiB; tile t+3 is issued seven segments before its first
// read and waited for with a counted vmcnt (its successor's pieces stay in flight).
#define SEG_KST 3
#define SEG_VST 4
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void flash_attn_seg_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                    const bf16* __restrict__ Vc, bf16* __restrict__ O,
                                                                    int Lq, int ldq, int ldo, int ldk,
                                                                    long long k_batch_stride, int kstart, int nkeys,
                                                                    float c, int nqt, int xcd_placement) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(NW == 8, "two wave groups of four, two K and two V pieces per wave and tile");
  char* const ksm = smem;
  char* const vsm = smem + SEG_KST * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  int nwg_ = gridDim.x, bid_ = blockIdx.x;      // XCD-aware (head, q-tile) placement: see flash_attn_pipe_kernel
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
  const int q0 = qtile * (NW * 32) + wave * 32;
  const int nt = (nkeys + KT - 1) / KT;
  const int last_valid = nkeys - (nt - 1) * KT;
  const bf16* kh = Kc + (size_t)b * k_batch_stride + (size_t)kstart * ldk + head * 128;
  const bf16* vh = Vc + (size_t)b * k_batch_stride + (size_t)kstart * ldk + head * 128;

  bf16x8 qf[8];
  {
    int qr = q0 + r;
    qr = qr < Lq ? qr : Lq - 1;
    const bf16* qp = Q + ((size_t)b * Lq + qr) * ldq + head * 128 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }

  constexpr int NDMA = 16 / NW;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  int dma_key[NDMA], dma_kch[NDMA], dma_vch[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    int key = 4 * (wave * NDMA + i) + (lane >> 4), pos = lane & 15;
    dma_key[i] = key;
    dma_kch[i] = (pos ^ (key & 15)) * 16;
    dma_vch[i] = (pos ^ ((key & 3) << 2)) * 16;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define SEG_DMA(T, KS, VS)                                                                       \
  {                                                                                              \
    int t_ = (T) < nt ? (T) : nt - 1;                                                            \
    int valid_ = (t_ == nt - 1) ? last_valid : KT;                                               \
    const char* kt_ = reinterpret_cast<const char*>(kh) + (size_t)t_ * KT * ldk * 2;            \
    const char* vt_ = reinterpret_cast<const char*>(vh) + (size_t)t_ * KT * ldk * 2;            \
    _Pragma("unroll") for (int i_ = 0; i_ < NDMA; ++i_) {                                        \
      int key_ = dma_key[i_] < valid_ ? dma_key[i_] : valid_ - 1;                                \
      unsigned row_ = (unsigned)key_ * (unsigned)ldk * 2u;                                       \
      int j_ = wave_u * NDMA + i_;                                                               \
      __builtin_amdgcn_global_load_lds((gptr_t)(kt_ + row_ + dma_kch[i_]), (lptr_t)(ksm + (KS) * TILE_B + j_ * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gptr_t)(vt_ + row_ + dma_vch[i_]), (lptr_t)(vsm + (VS) * TILE_B + j_ * 1024), 16, 0, 0); \
    }                                                                                            \
  }

  int k_off[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = r * 256 + (((2 * ks + h) ^ (r & 15)) << 4);   // + 8192 for kb = 1
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  int v_off[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    int dbyte = (32 * db + 16 * tg1 + 4 * tp) * 2;
    v_off[db] = (4 * h + tq) * 256 + (((dbyte >> 4) ^ (tq << 2)) << 4) + (dbyte & 15);   // + (32kb+16s2[+8])*256
  }
  typedef __attribute__((ext_vector_type(8))) short sg_s16x8;

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  bf16x8 kbuf[8];            // A fragments of one 32-key half of K
  s16x4 vlo[8], vhi[8];      // transposed A fragments of one 32-key half of V: [s2][db]
  f32x16 s_buf[2];           // S(h) (being softmaxed) and S(h+1) (being accumulated)
  uint4 pw[2][2];            // P(h-1), P(h): [half parity][s2]
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    vlo[i] = s16x4{0, 0, 0, 0};
    vhi[i] = s16x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) pw[a][s2] = make_uint4(0, 0, 0, 0);

#define SEG_LOAD_K(SLOT, KB)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                               \
    kbuf[ks] = *reinterpret_cast<const bf16x8*>(ksm + (SLOT) * TILE_B + k_off[ks] + (KB) * 8192);
#define SEG_LOAD_V(SLOT, KB)                                                                     \
  _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                               \
  _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                             \
    const char* a0 = vsm + (SLOT) * TILE_B + v_off[db] + (32 * (KB) + 16 * s2) * 256;            \
    vlo[4 * s2 + db] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));             \
    vhi[4 * s2 + db] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 8 * 256));   \
  }
  // S(next) = K(kbuf) Q^T: one accumulation chain starting from the inline constant 0
#define SEG_S_MFMA(DST)                                                                          \
  {                                                                                              \
    f32x16 acc_;                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc_[i] = 0.f;                                \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                             \
      acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kbuf[ks], qf[ks], acc_, 0, 0, 0);           \
    s_buf[DST] = acc_;                                                                           \
  }
#define SEG_PV_MFMA(PSRC)                                                                        \
  _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                             \
    bf16x8 pfrag = __builtin_bit_cast(bf16x8, pw[PSRC][s2]);                                     \
    _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                           \
      sg_s16x8 vv = __builtin_shufflevector(vlo[4 * s2 + db], vhi[4 * s2 + db], 0, 1, 2, 3, 4, 5, 6, 7); \
      o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pfrag, o[db], 0, 0, 0); \
    }                                                                                            \
  }
#ifndef LL_SEG_VALU
#define LL_SEG_VALU 4
#endif
#ifndef LL_SEG_HEAD
#define LL_SEG_HEAD 4
#endif
  // compute segment of half tile (T, KB): SCUR = KB, the other score buffer receives S(h+1); P(h) -> pw[KB], P(h-1) = pw[KB ^ 1]
#define SEG_COMPUTE(T, KB)                                                                       \
  {                                                                                              \
    if ((T) == nt - 1 && last_valid < KT) { /* ragged last tile (uniform branch) */              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                           \
        int key = 32 * (KB) + (i & 3) + 8 * (i >> 2) + 4 * h;                                    \
        if (key >= last_valid) s_buf[KB][i] = -INFINITY;                                         \
      }                                                                                          \
    }                                                                                            \
    SEG_S_MFMA((KB) ^ 1)                                                                         \
    SEG_PV_MFMA((KB) ^ 1)                                                                        \
    float mx = s_buf[KB][0];                                                                     \
    _Pragma("unroll") for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s_buf[KB][i]);                 \
    mx = xhalf_max(mx);                                                                          \
    float m_new = fmaxf(m_run, mx);                                                              \
    float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);                                   \
    float mc = m_new * c;                                                                        \
    float rs = 0.f;                                                                              \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                           \
      float p[8];                                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
        p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_buf[KB][8 * s2 + j], c, -mc));            \
        rs += p[j];                                                                              \
      }                                                                                          \
      pw[KB][s2] = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), \
                              pack_bf16x2(p[6], p[7]));                                          \
    }                                                                                            \
    rs = xhalf_sum(rs);                                                                          \
    l_run = l_run * alpha + rs;                                                                  \
    /* in-order issue: an MFMA that meets a busy matrix pipe blocks the VALU work behind it, so every MFMA is followed by   \
       ~24 issue cycles of softmax work (plain VALU 4, transcendental 8): the row maximum (no exponentials yet) under the     \
       first LL_SEG_HEAD MFMAs, then three plain + one or two transcendental per gap */           \
    _Pragma("unroll") for (int g_ = 0; g_ < LL_SEG_HEAD; ++g_) {                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x002, LL_SEG_VALU, 0);                               \
    }                                                                                            \
    _Pragma("unroll") for (int g_ = LL_SEG_HEAD; g_ < 16; ++g_) {                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);                                         \
      if (g_ & 1) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);                             \
      else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
    }                                                                                            \
    if (__any(m_new != m_run)) { /* O holds P(<= h-1) V: rescale before P(h) V(h) is added in the next compute segment */ \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) o[d][i] *= alpha;                           \
    }                                                                                            \
    m_run = m_new;                                                                               \
  }

  // ---- prologue: tiles 0..2 staged; S(0); K half 1 in registers ---------------------------------------------------------
  SEG_DMA(0, 0, 0);
  SEG_DMA(1, 1, 1);
  SEG_DMA(2, 2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  SEG_LOAD_K(0, 0);
  SEG_S_MFMA(0)
  SEG_LOAD_K(0, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const int late = __builtin_amdgcn_readfirstlane(wave >= NW / 2);
  // opaque-to-the-compiler "true" (each segment its own basic block = its own scheduling region), false for waves whose 32
  // query rows are all padding: they only stage and synchronise
  const bool own_block = xcd_placement >= 0 && qtile * (NW * 32) + wave_u * 32 < Lq;
#ifdef LL_ATTN_DIAG
  unsigned long long dg0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (late) __syncthreads();
  int kn = 1 % SEG_KST, vc = 0, kd = 0, vd = 3 % SEG_VST;      // slots: K(t+1), V(t), DMA targets K(t+3) = slot of K(t), V(t+3)
  for (int t = 0; t < nt; ++t) {
    if (own_block && !DIAG_ON(4)) SEG_COMPUTE(t, 0);         // C(2t):   S(2t+1), P(2t-1).V(2t-1), softmax S(2t)
    __syncthreads();
    if (!DIAG_ON(8)) {
      SEG_LOAD_K(kn, 0);                                     // L(2t):   K half 2t+2, V half 2t, stage tile t+3
      SEG_LOAD_V(vc, 0);
    }
    if (!DIAG_ON(2)) SEG_DMA(t + 3, kd, vd);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (own_block && !DIAG_ON(4)) SEG_COMPUTE(t, 1);         // C(2t+1): S(2t+2), P(2t).V(2t), softmax S(2t+1)
    __syncthreads();
    if (!DIAG_ON(8)) {
      SEG_LOAD_K(kn, 1);                                     // L(2t+1): K half 2t+3, V half 2t+1; tile t+2 has landed
      SEG_LOAD_V(vc, 1);
    }
    if (DIAG_ON(2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    kd = kd == SEG_KST - 1 ? 0 : kd + 1;
    kn = kn == SEG_KST - 1 ? 0 : kn + 1;
    vc = vc == SEG_VST - 1 ? 0 : vc + 1;
    vd = vd == SEG_VST - 1 ? 0 : vd + 1;
  }
#ifdef LL_ATTN_DIAG
  if (tid == 0 && blockIdx.x < 2048) {
    g_attn_diag[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - dg0;
    g_attn_diag[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - dr0;
    g_attn_diag[4 * blockIdx.x + 2] = nt;
  }
#endif
  if (own_block) SEG_PV_MFMA(1)                               // P(2nt-1).V(2nt-1)
  if (!late) __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the clamped look-ahead pieces still target this workgroup's LDS
#undef SEG_DMA
#undef SEG_LOAD_K
#undef SEG_LOAD_V
#undef SEG_S_MFMA
#undef SEG_PV_MFMA
#undef SEG_COMPUTE
  {
    const int qr = q0 + r;
    const int qc = qr < Lq ? qr : Lq - 1;
    store_o_rows(o, 1.0f / l_run, O + ((size_t)b * Lq + qc) * ldo + head * 128, h, qr < Lq);
  }
}

