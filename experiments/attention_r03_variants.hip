// NOT COMPILED -- kept for the record (round 4 moved them out of longlive_amd/csrc/attention.hip).
// Two variants of flash_attn_pipe_kernel<8, 1> that lost their A/B and are no fallback for any shipped shape:
//   flash_attn_pipe16_kernel   the ping-pong loop on v_mfma_f32_16x16x32_bf16: 2.0 % slower per launch, 0.9 % end to end
//                              (profiles/r03_ab_attn_mfma16.md)
//   flash_attn_sk_kernel + flash_attn_sk_combine_kernel   stream-K over all CUs + log-sum-exp merge: -5 % end to end in round 2
//                              (profiles/r02_ab_streamk_groupm.txt); round 4 re-measured the balanced grid on the generated kernel with a
//                              timing-only launch (profiles/r04_attn_balance.md): the clock gives back three quarters of the gain.
//                              Its W > tile-units case also read uninitialised workspace slots (found in round 4's GPU suite).
// They compile against attention.hip's helpers at commit 4b0bd8e (git show 4b0bd8e:longlive_amd/csrc/attention.hip).

// =================================================================================================================
// flash_attn_pipe_kernel<8, 1>'s ping-pong loop on v_mfma_f32_16x16x32_bf16 (tuning key attn_mfma16 = 1).  Same workgroup
// (8 waves x 32 query rows), same per-wave output tile, same LDS rings, same SM / MM cut, same barrier-shifted wave groups; only
// the MFMA shape and what follows from it differ.  Why it exists: where the chip holds its clock down under load the clock it
// holds depends on the MFMA shape (MI355X_MICROARCH.md DVFS give-back item 7: 16x16x32 loops delivered 1.12-1.15x the FLOP/s of
// 32x32x16 loops at equal cycles per FLOP), so the shape is decided by wall time of the real loop on random data.
//   lane = 16 g + i.   S^T = K Q^T per (16-key block kb, 16-query block qb): lane holds query 16 qb + i, keys 16 kb + 4 g + reg.
//   A query's 64 scores are spread over the 4 lanes i, i+16, i+32, i+48: row max = in-lane max of 16 + v_permlane16_swap +
//   v_permlane32_swap; the row SUM stays lane-partial through the loop (all four lanes share the running max, so the rescale
//   factor is common) and is reduced once in the epilogue.
//   P^T as the B operand of O^T += V^T P^T (k = 32): element j of lane group g <-> key 32 ks + 16 (j >> 2) + 4 g + (j & 3) =
//   accumulator registers of key blocks 2 ks and 2 ks + 1, no lane movement.  V^T fragments by two ds_read_b64_tr_b16 per
//   (16-d block, k-step) in that same key order.  A 32-lane half then reads key rows 4 apart in the same columns, which the
//   32x32 kernel's swizzle would serve 2-way: V chunk c of key r sits at c ^ (((r & 3) << 2) | (((r >> 2) & 1) << 1)).
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void flash_attn_pipe16_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                       const bf16* __restrict__ Vc, bf16* __restrict__ O,
                                                                       int Lq, int ldq, int ldo, int ldk,
                                                                       long long k_batch_stride, int kstart, int nkeys,
                                                                       float c, int nqt, int xcd_placement) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3 K stages][4 V stages] x 16 KiB
  constexpr int KSTAGES = PIPE_KSTAGES + 1, VSTAGES = PIPE_VSTAGES + 1;
  char* const ksm = smem;
  char* const vsm = smem + KSTAGES * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  int nwg_ = gridDim.x, bid_ = blockIdx.x;
  int qq_ = nwg_ >> 3, rr_ = nwg_ & 7, xcd_ = bid_ & 7;
  int lid_ = bid_;
  if (xcd_placement) {                                   // as flash_attn_pipe_kernel: head-major ranges per XCD
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

  // Q^T fragments (B operand): lane (i, g) holds Q[q0 + 16 qb + i][32 ks + 8 g .. +7]
  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qr = q0 + 16 * qb + li;
    qr = qr < Lq ? qr : Lq - 1;
    const bf16* qp = Q + ((size_t)b * Lq + qr) * ldq + head * 128 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
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
    dma_vch[i] = (pos ^ (((key & 3) << 2) | (((key >> 2) & 1) << 1))) * 16;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define P16_DMA(T, KS, VS)                                                                       \
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

  // K A-fragment: key = 16 kb + i, 16-byte chunk 4 ks + g at position chunk ^ (key & 15)            (+ 4096 kb)
  int k_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_off[ks] = li * 256 + (((4 * ks + g) ^ li) << 4);
  // V tr-read: lane supplies row (key) K0 + 4 g + q, columns 16 db + 4 p .. +3  (q = i >> 2, p = i & 3)      (+ 256 K0)
  const int tq = li >> 2, tp = li & 3;
  int v_off[8];
#pragma unroll
  for (int db = 0; db < 8; ++db) {
    int chunk = 2 * db + (tp >> 1);
    v_off[db] = (4 * g + tq) * 256 + ((chunk ^ ((tq << 2) | ((g & 1) << 1))) << 4) + (tp & 1) * 8;
  }

  f32x4 o[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 8; ++db) o[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  P16_DMA(0, 0, 0);
  P16_DMA(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 s_cur[2][4];
#define P16_QK(KN)                                                                               \
  {                                                                                              \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                             \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) s_cur[qb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};  \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                             \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                           \
      bf16x8 kf = *reinterpret_cast<const bf16x8*>((KN) + k_off[ks] + kb * 4096);                \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                           \
        s_cur[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], s_cur[qb][kb], 0, 0, 0); \
    }                                                                                            \
  }
  P16_QK(ksm);
  __syncthreads();   // K stage 0 is overwritten by iteration 0's staging

  uint4 pw[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) pw[qb][k2] = make_uint4(0, 0, 0, 0);
  typedef __attribute__((ext_vector_type(8))) short p16_s16x8;
  // SM(t): softmax of S(t) -> P(t); VALU + transcendental only
#define P16_SM(T)                                                                                \
  {                                                                                              \
    if ((T) == nt - 1 && last_valid < KT) {                                                      \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                           \
      _Pragma("unroll") for (int kb = 0; kb < 4; ++kb)                                           \
      _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                           \
        if (16 * kb + 4 * g + r_ >= last_valid) s_cur[qb][kb][r_] = -INFINITY;                   \
    }                                                                                            \
    float alpha[2];                                                                              \
    bool moved = false;                                                                          \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                           \
      float mx = s_cur[qb][0][0];                                                                \
      _Pragma("unroll") for (int kb = 0; kb < 4; ++kb)                                           \
      _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) mx = fmaxf(mx, s_cur[qb][kb][r_]);        \
      mx = xrow_max4(mx);                                                                        \
      float m_new = fmaxf(m_run[qb], mx);                                                        \
      alpha[qb] = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);                               \
      moved = moved || (m_new != m_run[qb]);                                                     \
      m_run[qb] = m_new;                                                                         \
      float mc = m_new * c;                                                                      \
      float rs = 0.f;                                                                            \
      _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                                         \
        float p_[8];                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                          \
          p_[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[qb][2 * k2 + (j >> 2)][j & 3], c, -mc)); \
          rs += p_[j];                                                                           \
        }                                                                                        \
        pw[qb][k2] = make_uint4(pack_bf16x2(p_[0], p_[1]), pack_bf16x2(p_[2], p_[3]), pack_bf16x2(p_[4], p_[5]), \
                                pack_bf16x2(p_[6], p_[7]));                                      \
      }                                                                                          \
      l_run[qb] = l_run[qb] * alpha[qb] + rs;      /* lane-partial: the 4 lanes of a query share m, hence alpha */ \
    }                                                                                            \
    if (__any(moved)) {                                                                          \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                           \
      _Pragma("unroll") for (int db = 0; db < 8; ++db)                                           \
      _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) o[qb][db][r_] *= alpha[qb];               \
    }                                                                                            \
  }
  // MM: O^T += V(VS)^T P^T (32 MFMAs, 32 transposed reads), then S = K(KS) Q^T (32 MFMAs, 16 reads)
#define P16_MM(VS, KS)                                                                           \
  {                                                                                              \
    const char* vc = vsm + (VS) * TILE_B;                                                        \
    _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                                             \
    _Pragma("unroll") for (int db = 0; db < 8; ++db) {                                           \
      const char* a0 = vc + v_off[db] + (32 * k2) * 256;                                         \
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));                   \
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 16 * 256));        \
      p16_s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                    \
      bf16x8 vf = __builtin_bit_cast(bf16x8, vv);                                                \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                           \
        o[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pw[qb][k2]), o[qb][db], 0, 0, 0); \
    }                                                                                            \
    P16_QK(ksm + (KS) * TILE_B);                                                                 \
    /* LDS reads ~8 MFMAs ahead: 8 transposed reads up front, then 2 per 2 MFMAs; K reads 1 per 2 MFMAs */ \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                           \
    _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                         \
    }                                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); \
  }
  const int late = __builtin_amdgcn_readfirstlane(wave >= NW / 2);
  const bool own_block = xcd_placement >= 0 && qtile * (NW * 32) + wave_u * 32 < Lq;   // opaque always-true for waves with rows (see pipe kernel)
  if (late) __syncthreads();
  int kq = 1, kd = 2, vq = 0, vd = 2;      // slots: K(t+1), K(t+2), V(t), V(t+2)
  for (int t = 0; t < nt; ++t) {
    P16_DMA(t + 2, kd, vd);
    if (own_block) P16_SM(t);
    __syncthreads();
    __builtin_amdgcn_s_setprio(1);
    if (own_block) P16_MM(vq, kq);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    kq = kd;
    kd = kd == KSTAGES - 1 ? 0 : kd + 1;
    vq = vq == VSTAGES - 1 ? 0 : vq + 1;
    vd = vd == VSTAGES - 1 ? 0 : vd + 1;
  }
  if (!late) __syncthreads();
#undef P16_DMA
#undef P16_QK
#undef P16_SM
#undef P16_MM

  // epilogue: lane holds O^T[d = 16 db + 4 g + (0..3)][query 16 qb + i].  v_permlane16_swap of the packed words of db = k (vdst)
  // and k + 1 (src) gives every lane 16 contiguous bytes: d = 16 (k + (g & 1)) + 8 (g >> 1) .. +7.
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = xrow_sum4(l_run[qb]);
    float inv = 1.0f / l;
    const int qr = q0 + 16 * qb + li;
    const int qc = qr < Lq ? qr : Lq - 1;
    bf16* orow = O + ((size_t)b * Lq + qc) * ldo + head * 128;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      unsigned ax = pack_bf16x2(o[qb][k][0] * inv, o[qb][k][1] * inv), ay = pack_bf16x2(o[qb][k][2] * inv, o[qb][k][3] * inv);
      unsigned bx = pack_bf16x2(o[qb][k + 1][0] * inv, o[qb][k + 1][1] * inv), by = pack_bf16x2(o[qb][k + 1][2] * inv, o[qb][k + 1][3] * inv);
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
      if (qr < Lq) *reinterpret_cast<uint4*>(orow + 16 * (k + (g & 1)) + 8 * (g >> 1)) = make_uint4(ax, ay, bx, by);
    }
  }
}

// =================================================================================================================
// Stream-K form of the ping-pong kernel: the (batch, head, 256-row q-tile) x key-tile work of a launch is cut into W equal
// CONTIGUOUS ranges of 64-key tile units, one per workgroup, W = number of CUs.
//
// Why: a launch is ONE round of workgroups (one per CU, 112 KiB of LDS each), so its duration is the tile count of the
// busiest CU.  The steady-state self-attention has 12 heads x 19 q-tiles = 228 (head, q-tile) pairs of 293 key tiles: 228
// CUs walk 293 tiles, 28 CUs idle.  Cut as 256 x 261 tiles every CU works and the walk is 11 % shorter (the recache launch,
// 888 pairs: 4 rounds of 293 -> 1017).  A workgroup's range covers the tail of one pair, possibly whole pairs, and the head
// of another; a part that is not a whole pair leaves (unnormalised O^T in fp32, running max, running sum) in a workspace
// slot and flash_attn_sk_combine_kernel merges the 2-3 parts of every split pair (log-sum-exp merge) into the bf16 output.
//
// Order inside a workgroup: LAST pair first.  All workgroups then start at key tile 0 together and, after the switch to
// their earlier pair, sit at one common offset again: the workgroups of a head form two fronts that stream the same K/V
// tiles at the same time, so the per-XCD L2 still serves most of them (walking the range in ascending order instead puts
// every workgroup at a different key offset: ~2 GB of L2 misses per launch).
//
// Everything inside a part -- LDS-DMA staging, rings, SM / MM phases, the barrier-shifted wave groups, register layout -- is
// flash_attn_pipe_kernel<NW, 1>'s, tile by tile; outputs of unsplit pairs are bit-identical to it.
#define SK_KSTAGES 3
#define SK_VSTAGES 4
#define SK_SLOT_FLOATS(NW) ((NW) * 32 * 128 + 2 * (NW) * 32)      // O^T image + m + l of one part

// first tile unit of workgroup w; the launcher guarantees U * W < 2^31 (32-bit divisions: the 64-bit ones are software loops)
__device__ __forceinline__ int sk_start(int w, int U, int W) { return (int)(((unsigned)w * (unsigned)U) / (unsigned)W); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void flash_attn_sk_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                    const bf16* __restrict__ Vc, bf16* __restrict__ O,
                                                                    float* __restrict__ ws, int Lq, int ldq, int ldo, int ldk,
                                                                    long long k_batch_stride, int kstart, int nkeys, float c,
                                                                    int nqt, int H, int npairs, int xcd_placement) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3 K stages][4 V stages] x 16 KiB
  char* const ksm = smem;
  char* const vsm = smem + SK_KSTAGES * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int W = gridDim.x;
  const int bid = blockIdx.x;
  // ids that share an XCD (id % 8) take consecutive ranges: one head's K/V stream meets ~2 of the 8 L2s
  const int lid = (xcd_placement > 0 && (W & 7) == 0) ? (bid & 7) * (W >> 3) + (bid >> 3) : bid;
  const int nt = (nkeys + KT - 1) / KT;
  const int last_valid = nkeys - (nt - 1) * KT;
  const int U = npairs * nt;
  const int u0 = sk_start(lid, U, W), u1 = sk_start(lid + 1, U, W);
  if (u0 >= u1) return;                                   // more workgroups than tile units (uniform: no barrier is pending)
  const int p_first = u0 / nt, p_last = (u1 - 1) / nt;

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
  const int late = __builtin_amdgcn_readfirstlane(wave >= NW / 2);
  int k_off[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_off[ks] = r * 256 + (((2 * ks + h) ^ (r & 15)) << 4);
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  int v_off[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    int dbyte = (32 * db + 16 * tg1 + 4 * tp) * 2;
    v_off[db] = (4 * h + tq) * 256 + (((dbyte >> 4) ^ (tq << 2)) << 4) + (dbyte & 15);
  }
  typedef __attribute__((ext_vector_type(8))) short sk_s16x8;

  for (int p = p_last; p >= p_first; --p) {
    const int pb = p * nt;
    const int tb = (u0 > pb ? u0 : pb) - pb;                              // this part = key tiles [tb, te) of pair p
    const int te = (u1 < pb + nt ? u1 : pb + nt) - pb;
    const int qtile = p % nqt, head = (p / nqt) % H, b = p / (nqt * H);
    const int q0 = qtile * (NW * 32) + wave * 32;
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
#define SK_DMA(T, KS, VS)                                                                        \
  {                                                                                              \
    int t_ = (T) < te ? (T) : te - 1;                                                            \
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
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- prologue: tiles tb, tb + 1 into LDS, S(tb) ----------------------------------------------------------------------
    SK_DMA(tb, 0, 0);
    SK_DMA(tb + 1, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s_cur[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s_cur[kb][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(ksm + k_off[ks] + kb * 8192);
        s_cur[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_cur[kb], 0, 0, 0);
      }
    __syncthreads();   // K stage 0 is overwritten by the first iteration's staging

    uint4 pw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) pw[kb][s2] = make_uint4(0, 0, 0, 0);
#define SK_SM(T)                                                                                 \
  {                                                                                              \
    if ((T) == nt - 1 && last_valid < KT) {                                                      \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                           \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                           \
        int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;                                      \
        if (key >= last_valid) s_cur[kb][i] = -INFINITY;                                         \
      }                                                                                          \
    }                                                                                            \
    float mx = s_cur[0][0];                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s_cur[kb][i]);                 \
    mx = xhalf_max(mx);                                                                          \
    float m_new = fmaxf(m_run, mx);                                                              \
    float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);                                   \
    float mc = m_new * c;                                                                        \
    float rs = 0.f;                                                                              \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                           \
      float p_[8];                                                                               \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
        p_[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kb][8 * s2 + j], c, -mc));           \
        rs += p_[j];                                                                             \
      }                                                                                          \
      pw[kb][s2] = make_uint4(pack_bf16x2(p_[0], p_[1]), pack_bf16x2(p_[2], p_[3]), pack_bf16x2(p_[4], p_[5]), \
                              pack_bf16x2(p_[6], p_[7]));                                        \
    }                                                                                            \
    rs = xhalf_sum(rs);                                                                          \
    l_run = l_run * alpha + rs;                                                                  \
    if (__any(m_new != m_run)) {                                                                 \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) o[d][i] *= alpha;                           \
    }                                                                                            \
    m_run = m_new;                                                                               \
  }
#define SK_MM(VS, KS)                                                                            \
  {                                                                                              \
    const char* vc = vsm + (VS) * TILE_B;                                                        \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                           \
      bf16x8 pfrag = __builtin_bit_cast(bf16x8, pw[kb][s2]);                                     \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                         \
        const char* a0 = vc + v_off[db] + (32 * kb + 16 * s2) * 256;                             \
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));                 \
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 8 * 256));       \
        sk_s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                   \
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pfrag, o[db], 0, 0, 0); \
      }                                                                                          \
    }                                                                                            \
    const char* kn = ksm + (KS) * TILE_B;                                                        \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s_cur[kb][i] = 0.f;                           \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                             \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                           \
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(kn + k_off[ks] + kb * 8192);                  \
      s_cur[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_cur[kb], 0, 0, 0);       \
    }                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);                                          \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                         \
    }                                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
  }
    // always true for waves that own query rows, but opaque to the compiler: each phase becomes its own scheduling region
    const bool own_block = xcd_placement >= 0 && qtile * (NW * 32) + wave_u * 32 < Lq;
    if (late) __syncthreads();
    int kq = 1, kd = 2, vq = 0, vd = 2;      // slots: K(t+1), K(t+2), V(t), V(t+2)
    for (int t = tb; t < te; ++t) {
      SK_DMA(t + 2, kd, vd);
      if (own_block) SK_SM(t);
      __syncthreads();
      __builtin_amdgcn_s_setprio(1);
      if (own_block) SK_MM(vq, kq);
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      kq = kd;
      kd = kd == SK_KSTAGES - 1 ? 0 : kd + 1;
      vq = vq == SK_VSTAGES - 1 ? 0 : vq + 1;
      vd = vd == SK_VSTAGES - 1 ? 0 : vd + 1;
    }
    if (!late) __syncthreads();
#undef SK_DMA
#undef SK_SM
#undef SK_MM

    const int qr = q0 + r;
    if (tb == 0 && te == nt) {                           // a whole pair: final output, as the unsplit kernel writes it
      const int qc = qr < Lq ? qr : Lq - 1;
      store_o_rows(o, 1.0f / l_run, O + ((size_t)b * Lq + qc) * ldo + head * 128, h, qr < Lq);
    } else {                                             // a part: the lane image of O^T (fully coalesced 1 KiB stores), m, l
      const int slot = 2 * lid + (tb > 0 ? 0 : 1);
      float* sp = ws + (size_t)slot * SK_SLOT_FLOATS(NW);
      float4* op = reinterpret_cast<float4*>(sp) + (size_t)wave * 16 * 64 + lane;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          op[(db * 4 + g4) * 64] = make_float4(o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]);
      if (h == 0) {
        sp[NW * 32 * 128 + wave * 32 + r] = m_run;
        sp[NW * 32 * 128 + NW * 32 + wave * 32 + r] = l_run;
      }
    }
  }
}

// Merges the parts of every split (batch, head, q-tile) pair: O = sum_i O_i 2^(c (m_i - M)) / sum_i l_i 2^(c (m_i - M)),
// M = max_i m_i.  One workgroup per pair, the attention kernel's thread geometry (each lane re-reads exactly the float4s a
// lane of that geometry wrote).  The enumeration of a pair's parts mirrors the kernel's range arithmetic (sk_start).
template <int NW>
__global__ __launch_bounds__(NW * 64) void flash_attn_sk_combine_kernel(const float* __restrict__ ws, bf16* __restrict__ O,
                                                                        int Lq, int ldo, int nkeys, float c, int nqt, int H,
                                                                        int npairs, int W) {
  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nt = (nkeys + KT - 1) / KT;
  const int U = npairs * nt;
  const int pb = p * nt, pe = pb + nt;
  int w = (int)(((unsigned)pb * (unsigned)W) / (unsigned)U);   // owner of unit pb: largest w with sk_start(w) <= pb
  while (w + 1 < W && sk_start(w + 1, U, W) <= pb) ++w;
  while (w > 0 && sk_start(w, U, W) > pb) --w;
  if (sk_start(w, U, W) <= pb && sk_start(w + 1, U, W) >= pe) return;      // unsplit pair: the kernel wrote the output
  const int qtile = p % nqt, head = (p / nqt) % H, b = p / (nqt * H);
  const int qr = qtile * (NW * 32) + wave * 32 + r;
  if (qtile * (NW * 32) + wave * 32 >= Lq) return;       // padding wave of the last q-tile (nothing was computed for it)
  const int w0 = w;
  float M = -INFINITY;
  for (w = w0; w < W && sk_start(w, U, W) < pe; ++w) {
    int slot = 2 * w + (sk_start(w, U, W) > pb ? 0 : 1);
    M = fmaxf(M, ws[(size_t)slot * SK_SLOT_FLOATS(NW) + NW * 32 * 128 + wave * 32 + r]);
  }
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.f;
  float L = 0.f;
  for (w = w0; w < W && sk_start(w, U, W) < pe; ++w) {
    int slot = 2 * w + (sk_start(w, U, W) > pb ? 0 : 1);
    const float* sp = ws + (size_t)slot * SK_SLOT_FLOATS(NW);
    float f = __builtin_amdgcn_exp2f((sp[NW * 32 * 128 + wave * 32 + r] - M) * c);
    L += sp[NW * 32 * 128 + NW * 32 + wave * 32 + r] * f;
    const float4* op = reinterpret_cast<const float4*>(sp) + (size_t)wave * 16 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float4 v = op[i * 64];
      acc[4 * i] += v.x * f; acc[4 * i + 1] += v.y * f; acc[4 * i + 2] += v.z * f; acc[4 * i + 3] += v.w * f;
    }
  }
  if (qr < Lq) {
    float inv = 1.0f / L;
    bf16* op = O + ((size_t)b * Lq + qr) * ldo + head * 128 + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        int i = (db * 4 + g4) * 4;
        uint2 wv;
        wv.x = pack_bf16x2(acc[i] * inv, acc[i + 1] * inv);
        wv.y = pack_bf16x2(acc[i + 2] * inv, acc[i + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * db + 8 * g4) = wv;
      }
  }
}

