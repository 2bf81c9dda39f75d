// Flash attention forward for head_dim 128 on gfx950: frame-sink + sliding-window self-attention straight out of the
// KV cache (no gather/cat), and cross-attention over the 512 text tokens.  Non-causal, no mask: every query of the
// block sees [seg0 | seg1] (wan/modules/causal_model.py:331-360).
//
// Structure (per workgroup: NW waves x 32 query rows of one head; per wave the whole 32 x 128 Q slab in registers):
//   S^T = K . Q^T      "swapped" product (A := K tile from LDS, B := Q^T in registers, v_mfma_f32_32x32x16_bf16): a lane
//                       then holds 32 scores of ONE query (its MFMA column), so the softmax row reductions are 31 in-lane
//                       max/adds + one lane^32 wave shuffle; no LDS round trip, no P buffer
//   P  = exp2(c.S^T - c.m) in registers -> bf16; the 32x32 f32 accumulator tile IS the B operand of the next MFMA
//                       (k = key index permuted as 16s + 8(j>>2) + 4h + (j&3))
//   O^T += V^T . P^T    A := V^T fetched from the row-major V tile in LDS with ds_read_b64_tr_b16 (hardware transpose) in
//                       that same permuted key order; O^T accumulators (128 d x 32 q = 64 f32/lane) hold one query per lane
//                       so the online-softmax rescale is a per-lane scalar
//   K/V tiles (64 keys x 128 d, 16 KiB each) : coalesced 16-B global loads -> registers (issued before the tile's MFMAs so
//                       HBM latency hides under compute) -> XOR-swizzled LDS (double buffered, one barrier per tile).
//   LDS swizzles: K row = 256 B; 16-B chunk c of key r at position c ^ (r & 15)  -> ds_read_b128 A-fragments conflict-free
//                 V row = 256 B; 16-B chunk c of key r at position c ^ ((r & 3) << 2) -> tr_b16 reads conflict-free
#include "common.h"

#define KT 64                       // keys per tile
#define TILE_B (KT * 128 * 2)       // 16 KiB

struct Segs {
  int s0, n0, s1, n1, nt0, nt;      // two key ranges and their tile counts
};

__device__ __forceinline__ void tile_range(const Segs& sg, int t, int& base, int& valid) {
  if (t < sg.nt0) {
    base = sg.s0 + t * KT;
    valid = sg.n0 - t * KT;
  } else {
    int u = t - sg.nt0;
    base = sg.s1 + u * KT;
    valid = sg.n1 - u * KT;
  }
  valid = valid < KT ? valid : KT;
}

// Staging registers are plain unrolled arrays indexed by compile-time constants, and the prefetch is UNCONDITIONAL
// (the tile index is clamped on the last iteration): a struct passed by reference across an `if (more)` branch was
// placed in scratch by hipcc, which put an s_waitcnt + scratch_store behind every global load (3x slower kernel).
#define ATTN_LOAD_TILE(BASE, VALID)                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < NCHUNK; ++i_) {                                        \
    int cid_ = tid + i_ * NW * 64;                                                               \
    int key_ = cid_ >> 4, ch_ = cid_ & 15;                                                       \
    int row_ = (BASE) + (key_ < (VALID) ? key_ : (VALID) - 1); /* masked keys re-read a valid row */ \
    size_t off_ = (size_t)row_ * ldk + ch_ * 8;                                                  \
    kr[i_] = *reinterpret_cast<const uint4*>(kh + off_);                                         \
    vr[i_] = *reinterpret_cast<const uint4*>(vh + off_);                                         \
  }
#define ATTN_STORE_TILE(KL, VL)                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < NCHUNK; ++i_) {                                        \
    int cid_ = tid + i_ * NW * 64;                                                               \
    int key_ = cid_ >> 4, ch_ = cid_ & 15;                                                       \
    *reinterpret_cast<uint4*>((KL) + key_ * 256 + ((ch_ ^ (key_ & 15)) << 4)) = kr[i_];          \
    *reinterpret_cast<uint4*>((VL) + key_ * 256 + ((ch_ ^ ((key_ & 3) << 2)) << 4)) = vr[i_];    \
  }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                const bf16* __restrict__ Vc, bf16* __restrict__ O,
                                                                int Lq, int ldq, int ldo, int ldk,
                                                                long long k_batch_stride, Segs sg, float c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K tile | V tile] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (NW * 32) + wave * 32;

  const bf16* kh = Kc + (size_t)b * k_batch_stride + head * 128;
  const bf16* vh = Vc + (size_t)b * k_batch_stride + head * 128;

  // Q^T fragments: lane (r, h) holds Q[q0 + r][16 ks + 8 h .. +7] for ks = 0..7  (B operand of S^T = K Q^T)
  bf16x8 qf[8];
  {
    int qr = q0 + r;
    qr = qr < Lq ? qr : Lq - 1;
    const bf16* qp = Q + ((size_t)b * Lq + qr) * ldq + head * 128 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  constexpr int NCHUNK = 1024 / (NW * 64);
  uint4 kr[NCHUNK], vr[NCHUNK];
  int base, valid;
  tile_range(sg, 0, base, valid);
  ATTN_LOAD_TILE(base, valid);
  ATTN_STORE_TILE(smem, smem + TILE_B);
  __syncthreads();

  // per-lane LDS offsets
  // K A-fragment: key = 32 kb + r, chunk = 2 ks + h  ->  off = key*256 + ((chunk ^ (key & 15)) << 4)
  // V tr-read   : lane = 16 g + 4 q + p; key = K0 + q (K0 = 32 kb + 16 s2 + 4 h [+8]); d0 = 32 db + 16 (g & 1) + 4 p
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;

  for (int t = 0; t < sg.nt; ++t) {
    const char* kl = smem + (t & 1) * 2 * TILE_B;
    const char* vl = kl + TILE_B;
    int cur_valid = valid;
    {
      int tn = t + 1 < sg.nt ? t + 1 : t;               // clamped: the last iteration re-fetches its own tile
      tile_range(sg, tn, base, valid);
      ATTN_LOAD_TILE(base, valid);                      // global -> regs, consumed after this tile's MFMAs
    }

    // ---- S^T = K Q^T : 2 key blocks x 8 k-steps ------------------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        int key = 32 * kb + r;
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(kl + key * 256 + (((2 * ks + h) ^ (key & 15)) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // lane now holds, for query r: s[kb][i] = score of key 32 kb + (i & 3) + 8 (i >> 2) + 4 h
    if (cur_valid < KT) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= cur_valid) s[kb][i] = -INFINITY;
        }
    }

    // ---- online softmax (base-2, scale folded: p = exp2(c s - c m)) ------------------------------------------
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float m_new = fmaxf(m_run, mx);
    float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    float mc = m_new * c;
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], c, -mc));
        s[kb][i] = p;
        rs += p;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    if (__any(m_new != m_run)) {   // wave-uniform; exact: alpha == 1 for every lane otherwise
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
    m_run = m_new;

    // P^T fragments (B operand): k-step (kb, s2) uses accumulator registers 8 s2 .. 8 s2 + 7 of s[kb]
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kb][s2][j] = (bf16)s[kb][8 * s2 + j];

    // ---- O^T += V^T P^T : 4 d-blocks x 4 k-steps -----------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        int key0 = 32 * kb + 16 * s2 + 4 * h + tq;   // row of the first 4-key block this lane addresses
        int key1 = key0 + 8;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          int dbyte = (32 * db + 16 * tg1 + 4 * tp) * 2;     // byte offset of d0 inside the 256-B row
          int ch = dbyte >> 4, sub = dbyte & 15;
          int a0 = key0 * 256 + ((ch ^ ((key0 & 3) << 2)) << 4) + sub;
          int a1 = key1 * 256 + ((ch ^ ((key1 & 3) << 2)) << 4) + sub;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vl + a0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vl + a1));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s2], o[db], 0, 0, 0);
        }
      }
    }

    {
      char* nk = smem + ((t + 1) & 1) * 2 * TILE_B;     // other stage: last read one barrier ago
      ATTN_STORE_TILE(nk, nk + TILE_B);
    }
    __syncthreads();
  }

  // ---- epilogue: O^T[d][q] / l  ->  out[q][head*128 + d]; lane holds d = 32 db + 8 g4 + 4 h + (0..3) --------------
  int qr = q0 + r;
  if (qr < Lq) {
    float inv = 1.0f / l_run;
    bf16* op = O + ((size_t)b * Lq + qr) * ldo + head * 128 + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (bf16)(o[db][4 * g4 + j] * inv);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * g4) = w;
      }
  }
}

// =================================================================================================================
// Software-pipelined variant (single contiguous key range -- the LongLive steady state and cross-attention).
//
// Inside ONE wave the matrix pipe and the VALU run concurrently (an MFMA occupies the issue port for 8 of its 32
// cycles), but a tile's own chain QK^T -> softmax -> PV is serial.  So iteration t overlaps two different tiles:
//     phase A:  S(t+1) = K(t+1) Q^T        [16 MFMA + 16 ds_read_b128]   ||   softmax of S(t) -> P(t)      [VALU]
//     phase B:  O^T   += V(t)^T P(t)^T     [16 MFMA + 32 ds_read_b64_tr] ||   stage tile t+2 into LDS      [ds_write]
// one barrier per tile.  K(t+1) is read in iteration t and V(t) in iteration t, so K lives in a 2-stage ring and V in a
// 3-stage ring (tile t+2 may be written while V(t) is still being read by slower waves).  Row reductions use
// v_permlane32_swap (VALU) instead of ds_bpermute so the softmax never waits on LDS.
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);   // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, r);
}

// Cross-half (lane ^ 32) reductions on the VALU: v_permlane32_swap exchanges lanes 32-63 of its first operand with
// lanes 0-31 of the second, so with both operands = x the results are [x_lo | x_lo] and [x_hi | x_hi].  The hazard
// "VALU write -> v_permlane read" needs 2 wait states; the builtin form read stale registers here (wrong row sums), so
// the instruction is issued from inline asm with the s_nop inside the statement.
__device__ __forceinline__ void xhalf_swap(float x, float& lo, float& hi) {
  lo = x;
  hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ float xhalf_max(float x) {
  float a, b;
  xhalf_swap(x, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xhalf_sum(float x) {
  float a, b;
  xhalf_swap(x, a, b);
  return a + b;
}

#define PIPE_KSTAGES 2
#define PIPE_VSTAGES 3

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void flash_attn_pipe_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                     const bf16* __restrict__ Vc, bf16* __restrict__ O,
                                                                     int Lq, int ldq, int ldo, int ldk,
                                                                     long long k_batch_stride, int kstart, int nkeys,
                                                                     float c, int nqt, int xcd_placement) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K stage 0,1][V stage 0,1,2] x 16 KiB
  char* const ksm = smem;
  char* const vsm = smem + PIPE_KSTAGES * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  // XCD-aware placement: workgroup ids that share an XCD (id % 8) get a contiguous, head-major range of (head, q-tile)
  // pairs, so one head's K/V stream is pulled into ~2 of the 8 L2s instead of all 8 (PMC: 8 x 57.5 MB per launch before).
  int nwg_ = gridDim.x, bid_ = blockIdx.x;
  int qq_ = nwg_ >> 3, rr_ = nwg_ & 7, xcd_ = bid_ & 7;
  int lid_ = xcd_placement ? (xcd_ < rr_ ? xcd_ * (qq_ + 1) : rr_ * (qq_ + 1) + (xcd_ - rr_) * qq_) + (bid_ >> 3) : bid_;
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

  constexpr int NCHUNK = 1024 / (NW * 64);
  uint4 kr[NCHUNK], vr[NCHUNK];
  // per-thread staging coordinates (loop invariant)
  int st_goff[NCHUNK], st_koff[NCHUNK], st_voff[NCHUNK], st_key[NCHUNK];
#pragma unroll
  for (int i = 0; i < NCHUNK; ++i) {
    int cid = tid + i * NW * 64;
    int key = cid >> 4, ch = cid & 15;
    st_key[i] = key;
    st_goff[i] = ch * 8;
    st_koff[i] = key * 256 + ((ch ^ (key & 15)) << 4);
    st_voff[i] = key * 256 + ((ch ^ ((key & 3) << 2)) << 4);
  }
#define PIPE_LOAD(T)                                                                             \
  {                                                                                              \
    int t_ = (T) < nt ? (T) : nt - 1;                                                            \
    int valid_ = (t_ == nt - 1) ? last_valid : KT;                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < NCHUNK; ++i_) {                                      \
      int key_ = st_key[i_] < valid_ ? st_key[i_] : valid_ - 1;                                  \
      size_t off_ = (size_t)(t_ * KT + key_) * ldk + st_goff[i_];                                \
      kr[i_] = *reinterpret_cast<const uint4*>(kh + off_);                                       \
      vr[i_] = *reinterpret_cast<const uint4*>(vh + off_);                                       \
    }                                                                                            \
  }
#define PIPE_STORE(KS, VS)                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < NCHUNK; ++i_) {                                        \
    *reinterpret_cast<uint4*>(ksm + (KS) * TILE_B + st_koff[i_]) = kr[i_];                       \
    *reinterpret_cast<uint4*>(vsm + (VS) * TILE_B + st_voff[i_]) = vr[i_];                       \
  }

  // per-lane LDS read offsets (loop invariant; the stage base is added per iteration)
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

  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // ---- prologue: tiles 0 and 1 into LDS, S(0) ------------------------------------------------------------------
  PIPE_LOAD(0);
  PIPE_STORE(0, 0);
  PIPE_LOAD(1);
  PIPE_STORE(1, 1);
  __syncthreads();
  f32x16 s_cur[2], s_nxt[2];
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
  __syncthreads();   // K stage 0 is overwritten by iteration 0's staging: every wave must be done reading it

  int vstage = 0;
  for (int t = 0; t < nt; ++t) {
    const char* kn = ksm + ((t + 1) & 1) * TILE_B;       // K(t+1)
    const char* vc = vsm + vstage * TILE_B;              // V(t)
    PIPE_LOAD(t + 2);                                    // global -> regs; written to LDS in phase B

    if (t == nt - 1 && last_valid < KT) {                // ragged last tile: mask (uniform branch, own region)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= last_valid) s_cur[kb][i] = -INFINITY;
        }
    }

    // ---- phase A: S(t+1) on the matrix pipe  ||  softmax(S(t)) on the VALU --------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s_nxt[kb][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(kn + k_off[ks] + kb * 8192);
        s_nxt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_nxt[kb], 0, 0, 0);
      }
    float mx = s_cur[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s_cur[kb][i]);
    mx = xhalf_max(mx);
    float m_new = fmaxf(m_run, mx);
    float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    float mc = m_new * c;
    float rs = 0.f;
    uint4 pw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kb][8 * s2 + j], c, -mc));
          rs += p[j];
        }
        pw[kb][s2] = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]),
                                pack_bf16x2(p[6], p[7]));
      }
    rs = xhalf_sum(rs);
    l_run = l_run * alpha + rs;
#ifdef LL_ATTN_SCHED
    // pin the phase-A interleave: per MFMA one K-fragment read and a slice of the softmax VALU/transcendental work
    // (LLVM SchedGroupMask: VALU 0x2, MFMA 0x8, DS_READ 0x100, TRANS 0x400)
#pragma unroll
    for (int g_ = 0; g_ < 16; ++g_) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, LL_ATTN_SCHED, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
    }
#endif

    if (__any(m_new != m_run)) {   // wave-uniform, exact: alpha == 1 in every lane otherwise
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
    m_run = m_new;

    // ---- phase B: O^T += V(t)^T P(t)^T  ||  stage tile t+2 -------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pfrag = __builtin_bit_cast(bf16x8, pw[kb][s2]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const char* a0 = vc + v_off[db] + (32 * kb + 16 * s2) * 256;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 8 * 256));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pfrag, o[db], 0, 0, 0);
        }
      }
    {
      int vs2 = vstage + 2;
      vs2 = vs2 >= PIPE_VSTAGES ? vs2 - PIPE_VSTAGES : vs2;
      PIPE_STORE(t & 1, vs2);      // K(t+2) over K(t) (read an iteration ago), V(t+2) over V(t-1)
    }
#ifdef LL_ATTN_SCHED_B
    // phase-B interleave: per MFMA the two transposed V reads of the NEXT MFMA; staging writes spread behind them
#pragma unroll
    for (int g_ = 0; g_ < 16; ++g_) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
      __builtin_amdgcn_sched_group_barrier(0x002, LL_ATTN_SCHED_B, 1);
    }
#endif
    __syncthreads();
    s_cur[0] = s_nxt[0];
    s_cur[1] = s_nxt[1];
    vstage = vstage == PIPE_VSTAGES - 1 ? 0 : vstage + 1;
  }

  int qr = q0 + r;
  if (qr < Lq) {
    float inv = 1.0f / l_run;
    bf16* op = O + ((size_t)b * Lq + qr) * ldo + head * 128 + 4 * h;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 w;
        w.x = pack_bf16x2(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv);
        w.y = pack_bf16x2(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * db + 8 * g4) = w;
      }
  }
}

static int g_attn_variant = 1;
static int g_attn_xcd = 1;
void ll_set_attn_xcd_internal(int v) { g_attn_xcd = v; }   // 0: simple kernel, 1: software-pipelined kernel (single key range)
void ll_set_attn_variant_internal(int v) { g_attn_variant = v; }

static int flash_attn_pipe_launch(const ll_bf16* q, const ll_bf16* k, const ll_bf16* v, ll_bf16* out, int B,
                                         int Lq, int H, int ldq, int ldo, int ldk, long long k_batch_stride, int kstart,
                                         int nkeys, float c, ll_stream stream) {
  constexpr int NW = 8;
  size_t lds = (PIPE_KSTAGES + PIPE_VSTAGES) * TILE_B;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)flash_attn_pipe_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  int nqt = (Lq + NW * 32 - 1) / (NW * 32);
  dim3 grid(nqt * H, 1, B), block(NW * 64);
  hipLaunchKernelGGL(flash_attn_pipe_kernel<NW>, grid, block, lds, (hipStream_t)stream, (const bf16*)q, (const bf16*)k,
                     (const bf16*)v, (bf16*)out, Lq, ldq, ldo, ldk, k_batch_stride, kstart, nkeys, c, nqt, g_attn_xcd);
  return ll_check_launch("ll_flash_attn(pipe)");
}

extern "C" int ll_flash_attn(const ll_bf16* q, const ll_bf16* k, const ll_bf16* v, ll_bf16* out, int B, int Lq, int H,
                             int ldq, int ldo, int ldk, long long k_batch_stride, int seg0_start, int seg0_len,
                             int seg1_start, int seg1_len, float scale, ll_stream stream) {
  LL_REQUIRE(ldq % 8 == 0 && ldo % 4 == 0 && ldk % 8 == 0, "ll_flash_attn: row strides must be multiples of 8 elements");
  LL_REQUIRE(ldq >= H * 128 && ldo >= H * 128 && ldk >= H * 128, "ll_flash_attn: row stride smaller than H*128");
  LL_REQUIRE(seg0_len > 0 && seg1_len >= 0 && seg0_start >= 0 && seg1_start >= 0, "ll_flash_attn: needs a non-empty first key range");
  if (B == 0 || Lq == 0 || H == 0) return LL_OK;
  Segs sg;
  sg.s0 = seg0_start; sg.n0 = seg0_len; sg.s1 = seg1_start; sg.n1 = seg1_len;
  if (sg.n1 > 0 && sg.s1 == sg.s0 + sg.n0) { sg.n0 += sg.n1; sg.n1 = 0; }   // contiguous: one range
  sg.nt0 = (sg.n0 + KT - 1) / KT;
  sg.nt = sg.nt0 + (sg.n1 + KT - 1) / KT;
  float c = scale * 1.4426950408889634f;
  if (g_attn_variant >= 1 && sg.n1 == 0)
    return flash_attn_pipe_launch(q, k, v, out, B, Lq, H, ldq, ldo, ldk, k_batch_stride, sg.s0, sg.n0, c, stream);
  constexpr int NW = 4;
  dim3 grid((Lq + NW * 32 - 1) / (NW * 32), H, B), block(NW * 64);
  hipLaunchKernelGGL(flash_attn_kernel<NW>, grid, block, 4 * TILE_B, (hipStream_t)stream, (const bf16*)q, (const bf16*)k,
                     (const bf16*)v, (bf16*)out, Lq, ldq, ldo, ldk, k_batch_stride, sg, c);
  return ll_check_launch("ll_flash_attn");
}

