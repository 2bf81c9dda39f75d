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
#include <stdio.h>

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

// Diagnostic build only (-DLL_ATTN_DIAG=bits, tools/attn_diag.hip; never defined for the shipped library): in-kernel clock
// stamps around the ping-pong main loop (s_memtime = shader cycles, s_memrealtime = 100 MHz) and switches that drop one part of
// the loop at a time (results are wrong by construction): 2 = no K/V staging in the loop, 4 = no softmax phase, 8 = no matrix
// phase, 16 = no wait for the staged tile.
#ifdef LL_ATTN_DIAG
__device__ unsigned long long g_attn_diag[4 * 2048];
extern "C" int ll_attn_diag_read(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_diag), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define DIAG_ON(bit) ((LL_ATTN_DIAG) & (bit))
#else
#define DIAG_ON(bit) 0
#endif

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

// Final O^T / l -> bf16 rows.  A lane (r, h) holds, for query r, d = 32 db + 8 g4 + 4 h + (0..3): 16 x 8-byte stores per lane.
// v_permlane32_swap of the packed words of g4 = k (vdst) and k + 1 (src) hands lanes 0-31 the upper half's group k and lanes
// 32-63 the lower half's group k + 1: every lane then owns 16 contiguous bytes (d = 32 db + 8 (k + h) .. +7) -> 8 x 16-byte
// stores, same bytes, same addresses, half the store instructions (the tail of this kernel is store-issue bound).  The swap
// crosses lanes: it runs with EXEC all ones, only the store is predicated.
__device__ __forceinline__ void store_o_rows(const f32x16 (&o)[4], float inv, bf16* __restrict__ orow, int h, bool valid) {
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int kp = 0; kp < 4; kp += 2) {
      unsigned ax = pack_bf16x2(o[db][4 * kp] * inv, o[db][4 * kp + 1] * inv);
      unsigned ay = pack_bf16x2(o[db][4 * kp + 2] * inv, o[db][4 * kp + 3] * inv);
      unsigned bx = pack_bf16x2(o[db][4 * kp + 4] * inv, o[db][4 * kp + 5] * inv);
      unsigned by = pack_bf16x2(o[db][4 * kp + 6] * inv, o[db][4 * kp + 7] * inv);
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
      if (valid) *reinterpret_cast<uint4*>(orow + 32 * db + 8 * (kp + h)) = make_uint4(ax, ay, bx, by);
    }
}

#define PIPE_KSTAGES 2
#define PIPE_VSTAGES 3

// PP = 1: "ping-pong" schedule.  Waves w and w + 4 share a SIMD.  In the PP = 0 kernel every wave runs phase A (softmax of
// tile t on the VALU with the S(t+1) MFMAs interleaved) and then phase B (P.V on the matrix pipe) in lock step between the
// per-tile barriers, so the two waves of a SIMD contend for the VALU in A and for the matrix pipe in B: measured ~4150
// cycles per tile against ~1050 VALU + ~1024 MFMA cycles of work per wave.  With PP the work of a tile is cut the other
// way -- SM(t) = softmax(S(t)) (VALU only) and MM(t) = P(t).V(t) + S(t+1) = K(t+1).Q (32 MFMAs, no VALU) -- and waves 4..7
// run half a tile behind: [SM(t), MM(t)] on waves 0..3 against [MM(t-1), SM(t)] on waves 4..7, so on every SIMD one wave
// is on the VALU while the other is on the matrix pipe.  S(t+1) overwrites S(t) (no second score tile: 32 registers fewer).
// Costs one more stage of K and of V (the late group still reads K(t), V(t-1) while tile t+2 is staged): 7 x 16 KiB of LDS,
// one workgroup per CU (the grid has fewer workgroups than CUs anyway at Lq = 4680).
template <int NW, int PP>
__global__ __launch_bounds__(NW * 64, PP ? 1 : 2) void flash_attn_pipe_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kc,
                                                                     const bf16* __restrict__ Vc, bf16* __restrict__ O,
                                                                     int Lq, int ldq, int ldo, int ldk,
                                                                     long long k_batch_stride, int kstart, int nkeys,
                                                                     float c, int nqt, int xcd_placement) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K stages][V stages] x 16 KiB
  constexpr int KSTAGES = PIPE_KSTAGES + PP, VSTAGES = PIPE_VSTAGES + PP;
  char* const ksm = smem;
  char* const vsm = smem + KSTAGES * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  // XCD-aware placement: workgroup ids that share an XCD (id % 8) get a contiguous, head-major range of (head, q-tile)
  // pairs, so one head's K/V stream is pulled into ~2 of the 8 L2s instead of all 8 (PMC: 8 x 57.5 MB per launch before).
  int nwg_ = gridDim.x, bid_ = blockIdx.x;
  int qq_ = nwg_ >> 3, rr_ = nwg_ & 7, xcd_ = bid_ & 7;
  // The rr_ XCDs that hold one workgroup more than the others are spread evenly over the id range (ranges of 29, 28, 29, 28 ...
  // rather than 29 x 4, 28 x 4): with 12 heads x 19 q-tiles the range boundaries then fall on multiples of 9.5 q-tiles, every
  // head meets at most two XCDs without one-tile slivers, 16 (head, XCD) pairs instead of 19.  Still a bijection.
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

  // K/V staging by LDS-DMA (global_load_lds, 16 B per lane, no staging registers): instruction j of a tile fills LDS bytes
  // [1024 j, 1024 (j+1)) linearly = keys 4j..4j+3; the lane at (key, pos) fetches the 16-byte chunk that the XOR swizzle
  // puts there (K: pos ^ (key & 15); V: pos ^ ((key & 3) << 2)) -- the swizzle lives on the source side.  Each wave issues
  // 16 / NW instructions for K and as many for V per tile.
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
#ifndef LL_ATTN_PRIO
#define LL_ATTN_PRIO 0           // experiment: 0 = matrix phase at priority 1 (shipped), 1 = no priorities, 2 = softmax phase at priority 1
#endif
#define PIPE_DMA_SETUP(T)                                                                        \
    int t_ = (T) < nt ? (T) : nt - 1;                                                            \
    int valid_ = (t_ == nt - 1) ? last_valid : KT;                                               \
    const char* kt_ = reinterpret_cast<const char*>(kh) + (size_t)t_ * KT * ldk * 2;            \
    const char* vt_ = reinterpret_cast<const char*>(vh) + (size_t)t_ * KT * ldk * 2;
#define PIPE_DMA(T, KS, VS)                                                                      \
  {                                                                                              \
    PIPE_DMA_SETUP(T)                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < NDMA; ++i_) {                                        \
      int key_ = dma_key[i_] < valid_ ? dma_key[i_] : valid_ - 1;                                \
      unsigned row_ = (unsigned)key_ * (unsigned)ldk * 2u;                                       \
      int j_ = wave_u * NDMA + i_;                                                               \
      __builtin_amdgcn_global_load_lds((gptr_t)(kt_ + row_ + dma_kch[i_]), (lptr_t)(ksm + (KS) * TILE_B + j_ * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gptr_t)(vt_ + row_ + dma_vch[i_]), (lptr_t)(vsm + (VS) * TILE_B + j_ * 1024), 16, 0, 0); \
    }                                                                                            \
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
  PIPE_DMA(0, 0, 0);
  PIPE_DMA(1, 1, 1);
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
  __syncthreads();   // K stage 0 is overwritten by iteration 0's staging: every wave must be done reading it

  // ---- phase A(t): S(t+1) on the matrix pipe  ||  softmax(S(t)) on the VALU; ends with s_cur = S(t+1) ------------------
#ifdef LL_ATTN_SCHED
  // pin the phase-A interleave: per MFMA one K-fragment read and a slice of the softmax VALU/transcendental work
  // (LLVM SchedGroupMask: VALU 0x2, MFMA 0x8, DS_READ 0x100, TRANS 0x400)
#define PHASE_A_SCHED                                                                            \
  _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                           \
    __builtin_amdgcn_sched_group_barrier(0x002, LL_ATTN_SCHED, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                                           \
  }
#else
#define PHASE_A_SCHED
#endif
#define PHASE_A(T)                                                                               \
  {                                                                                              \
    const char* kn = ksm + (((T) + 1) & 1) * TILE_B; /* K(t+1) */                                \
    if ((T) == nt - 1 && last_valid < KT) { /* ragged last tile: mask (uniform branch) */        \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                           \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                           \
        int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;                                      \
        if (key >= last_valid) s_cur[kb][i] = -INFINITY;                                         \
      }                                                                                          \
    }                                                                                            \
    f32x16 s_nxt[2];                                                                             \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s_nxt[kb][i] = 0.f;                           \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                             \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                           \
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(kn + k_off[ks] + kb * 8192);                  \
      s_nxt[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_nxt[kb], 0, 0, 0);       \
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
      float p[8];                                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
        p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[kb][8 * s2 + j], c, -mc));            \
        rs += p[j];                                                                              \
      }                                                                                          \
      pw[kb][s2] = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), \
                              pack_bf16x2(p[6], p[7]));                                          \
    }                                                                                            \
    rs = xhalf_sum(rs);                                                                          \
    l_run = l_run * alpha + rs;                                                                  \
    PHASE_A_SCHED                                                                                \
    if (__any(m_new != m_run)) { /* wave-uniform, exact: alpha == 1 in every lane otherwise */   \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) o[d][i] *= alpha;                           \
    }                                                                                            \
    m_run = m_new;                                                                               \
    s_cur[0] = s_nxt[0];                                                                         \
    s_cur[1] = s_nxt[1];                                                                         \
  }
  // ---- phase B: O^T += V^T P^T with the P of the last phase A and the V stage VS ---------------------------------------
#define PHASE_B(VS)                                                                              \
  {                                                                                              \
    const char* vc = vsm + (VS) * TILE_B;                                                        \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                           \
      bf16x8 pfrag = __builtin_bit_cast(bf16x8, pw[kb][s2]);                                     \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                         \
        const char* a0 = vc + v_off[db] + (32 * kb + 16 * s2) * 256;                             \
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));                 \
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 8 * 256));       \
        pp_s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                   \
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pfrag, o[db], 0, 0, 0); \
      }                                                                                          \
    }                                                                                            \
  }
  typedef __attribute__((ext_vector_type(8))) short pp_s16x8;
  uint4 pw[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) pw[kb][s2] = make_uint4(0, 0, 0, 0);
  if constexpr (PP == 0) {
    int vstage = 0;
    for (int t = 0; t < nt; ++t) {
      {                                                  // tile t+2: K over K(t) (read an interval ago), V over V(t-1)
        int vs2 = vstage + 2;
        vs2 = vs2 >= VSTAGES ? vs2 - VSTAGES : vs2;
        PIPE_DMA(t + 2, t & 1, vs2);
      }
      PHASE_A(t);
      PHASE_B(vstage);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile t+2 has landed in LDS
      __syncthreads();
      vstage = vstage == VSTAGES - 1 ? 0 : vstage + 1;
    }
  } else {
    // SM(t): softmax of S(t) in s_cur -> P(t) in pw, running max / sum, O rescale.  VALU + transcendental only.
#define LL_EXP2(x) (DIAG_ON(32) ? (x) : __builtin_amdgcn_exp2f(x))
#define PP_SM(T)                                                                                 \
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
      float p[8];                                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
        p[j] = LL_EXP2(__builtin_fmaf(s_cur[kb][8 * s2 + j], c, -mc));                           \
        rs += p[j];                                                                              \
      }                                                                                          \
      pw[kb][s2] = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), \
                              pack_bf16x2(p[6], p[7]));                                          \
    }                                                                                            \
    rs = xhalf_sum(rs);                                                                          \
    l_run = l_run * alpha + rs;                                                                  \
    if (__any(m_new != m_run)) {                                                                 \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                              \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) o[d][i] *= alpha;                           \
    }                                                                                            \
    m_run = m_new;                                                                               \
  }
    // MM: O^T += V(VS)^T P^T, then s_cur = K(KS) Q^T (the next score tile over the consumed one).  Matrix pipe + LDS reads.
#define PP_MM(VS, KS)                                                                            \
  {                                                                                              \
    PHASE_B(VS);                                                                                 \
    const char* kn = ksm + (KS) * TILE_B;                                                        \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) s_cur[kb][i] = 0.f;                           \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                             \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                           \
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(kn + k_off[ks] + kb * 8192);                  \
      s_cur[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_cur[kb], 0, 0, 0);       \
    }                                                                                            \
    /* LDS reads eight MFMAs ahead (16 transposed V reads / 8 K reads in flight), not all up front (registers) */ \
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
    const int late = __builtin_amdgcn_readfirstlane(wave >= NW / 2);
    // always true, but opaque to the compiler: each phase becomes its own basic block = its own scheduling region.  Without
    // it LLVM mixes the two phases' instructions across the barrier and the kernel is 15 % slower (594 vs 517 us in situ).
    const bool own_block = xcd_placement >= 0 && qtile * (NW * 32) + wave_u * 32 < Lq;   // ... and false for waves whose 32 query rows are all padding
                                                            // (last q-tile): they only stage and sync -- the chip runs at
                                                            // its power cap, an idle wave is clock for the others
#ifdef LL_ATTN_DIAG
    unsigned long long dg0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (late) __syncthreads();
    int kq = 1, kd = 2, vq = 0, vd = 2;      // slots: K(t+1), K(t+2), V(t), V(t+2)
    for (int t = 0; t < nt; ++t) {
      if (!DIAG_ON(2)) PIPE_DMA(t + 2, kd, vd);
      if (own_block && !DIAG_ON(4)) PP_SM(t);
      __syncthreads();
      if (LL_ATTN_PRIO == 0) __builtin_amdgcn_s_setprio(1);      // MM is the longer phase: its MFMA / LDS issue wins over the partner's softmax VALU
      if (LL_ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      if (own_block && !DIAG_ON(8)) PP_MM(vq, kq);
      if (LL_ATTN_PRIO == 0) __builtin_amdgcn_s_setprio(0);
      if (LL_ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(1);
      if (!DIAG_ON(16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      kq = kd;
      kd = kd == KSTAGES - 1 ? 0 : kd + 1;
      vq = vq == VSTAGES - 1 ? 0 : vq + 1;
      vd = vd == VSTAGES - 1 ? 0 : vd + 1;
    }
    if (!late) __syncthreads();
#ifdef LL_ATTN_DIAG
    if (tid == 0 && blockIdx.x < 2048) {
      g_attn_diag[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - dg0;
      g_attn_diag[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - dr0;
      g_attn_diag[4 * blockIdx.x + 2] = nt;
    }
#endif
#undef PP_SM
#undef PP_MM
  }
#undef PHASE_A
#undef PHASE_B
#undef PHASE_A_SCHED

  {
    const int qr = q0 + r;
    const int qc = qr < Lq ? qr : Lq - 1;
    store_o_rows(o, 1.0f / l_run, O + ((size_t)b * Lq + qc) * ldo + head * 128, h, qr < Lq);
  }
}



// Reductions over the four 16-lane rows of a wave (lanes i, i+16, i+32, i+48) on the VALU: v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of the second, v_permlane32_swap the upper half of the first with the lower half
// of the second; with both operands = x each leaves the two values to combine in the same lane.  Hazard nops inside the
// statements (VALU write -> v_permlane read: 2 wait states).
__device__ __forceinline__ float xrow_max4(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  float y = fmaxf(a, b);
  float c2 = y, d = y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c2), "+v"(d));
  return fmaxf(c2, d);
}
__device__ __forceinline__ float xrow_sum4(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  float y = a + b;
  float c2 = y, d = y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c2), "+v"(d));
  return c2 + d;
}

#ifndef LL_ATTN_VARIANT_DEFAULT
#define LL_ATTN_VARIANT_DEFAULT 2
#endif
static int g_attn_variant = LL_ATTN_VARIANT_DEFAULT;   // 0: simple kernel, 1: software-pipelined, 2: + ping-pong wave groups for long key ranges
static int g_attn_pp_min_keys = 16 * KT;   // key ranges at least this long run the ping-pong loop (cross-attention's 512 keys: one-barrier loop)
void ll_set_attn_pp_min_internal(int v) { g_attn_pp_min_keys = v; }
static int g_attn_xcd = 1;
static int g_attn_asm = 1;        // tuning key attn_asm (DEFAULT 1; 0 = flash_attn_pipe_kernel<8, 1>): long contiguous key ranges run flash_attn_asm_kernel
                                  // (attention_asm.hip: 4 waves x 64 rows, one wave per SIMD, generated hand-scheduled body)
void ll_set_attn_asm_internal(int v) { g_attn_asm = v; }
int flash_attn_asm_launch(const bf16* q, const bf16* k, const bf16* v, bf16* out, int B, int Lq, int H, int ldq, int ldo, int ldk,
                          long long k_batch_stride, int kstart, int nkeys, float c, int xcd, int form, hipStream_t stream);
static int g_attn_asm_min_keys = 8 * KT;    // the generated kernel from 512 keys on (cross-attention: 22.7 vs 26.8 us, profiles/r03_cross_attn_asm.txt)
void ll_set_attn_asm_min_internal(int v) { g_attn_asm_min_keys = v; }
static bool attn_asm_eligible(int nkeys, int ldk) {
  // (the generated kernel needs at least two key tiles: its first tile is a special case and so is its last; whatever the tuning
  //  key says, shorter ranges stay on the HIP kernels -- tests/test_attn_asm_emu.py::test_two_tiles_is_the_shortest_range)
  const int min_keys = g_attn_asm_min_keys > 2 * KT ? g_attn_asm_min_keys : 2 * KT;
  return g_attn_asm && g_attn_variant >= 2 && nkeys >= min_keys && (long long)nkeys * ldk * 2 < 0x7fffffffLL;
}
void ll_set_attn_xcd_internal(int v) { g_attn_xcd = v; }   // 0: simple kernel, 1: software-pipelined kernel (single key range)
void ll_set_attn_variant_internal(int v) { g_attn_variant = v; }

static int flash_attn_pipe_launch(const ll_bf16* q, const ll_bf16* k, const ll_bf16* v, ll_bf16* out, int B,
                                         int Lq, int H, int ldq, int ldo, int ldk, long long k_batch_stride, int kstart,
                                         int nkeys, float c, ll_stream stream) {
  constexpr int NW = 8;
  {
    (void)ll_lds_attr((const void*)flash_attn_pipe_kernel<NW, 0>, (int)((PIPE_KSTAGES + PIPE_VSTAGES) * TILE_B));
    (void)ll_lds_attr((const void*)flash_attn_pipe_kernel<NW, 1>, (int)((PIPE_KSTAGES + PIPE_VSTAGES + 2) * TILE_B));
  }
  int nqt = (Lq + NW * 32 - 1) / (NW * 32);
  dim3 grid(nqt * H, 1, B), block(NW * 64);
  if (attn_asm_eligible(nkeys, ldk))
    return flash_attn_asm_launch((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, B, Lq, H, ldq, ldo, ldk, k_batch_stride,
                                 kstart, nkeys, c, g_attn_xcd, g_attn_asm, (hipStream_t)stream);
  if (g_attn_variant >= 2 && nkeys >= g_attn_pp_min_keys)   // short ranges (cross-attention, 512 keys): the one-barrier loop is faster
    hipLaunchKernelGGL((flash_attn_pipe_kernel<NW, 1>), grid, block, (PIPE_KSTAGES + PIPE_VSTAGES + 2) * TILE_B,
                       (hipStream_t)stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, Lq, ldq, ldo, ldk,
                       k_batch_stride, kstart, nkeys, c, nqt, g_attn_xcd);
  else
    hipLaunchKernelGGL((flash_attn_pipe_kernel<NW, 0>), grid, block, (PIPE_KSTAGES + PIPE_VSTAGES) * TILE_B,
                       (hipStream_t)stream, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, Lq, ldq, ldo, ldk,
                       k_batch_stride, kstart, nkeys, c, nqt, g_attn_xcd);
  return ll_check_launch("ll_flash_attn(pipe)");
}

// Which kernel instance ll_flash_attn launches for these key ranges under the current tuning (host only).
extern "C" int ll_flash_attn_plan(int Lq, int H, int B, int seg0_len, int seg1_len, int seg_adjacent, char* out, int cap) {
  LL_REQUIRE(out != nullptr && cap > 0, "ll_flash_attn_plan: needs an output buffer");
  int n0 = seg0_len, n1 = seg1_len;
  if (n1 > 0 && seg_adjacent) { n0 += n1; n1 = 0; }
  if (g_attn_variant >= 1 && n1 == 0) {
    int nqt = (Lq + 255) / 256;
    bool pp = g_attn_variant >= 2 && n0 >= g_attn_pp_min_keys;
    if (attn_asm_eligible(n0, H * 128))
      snprintf(out, (size_t)cap, "%s (4 waves x 64 rows, one wave per SIMD, generated schedule), %d workgroups of 256 query rows%s",
               "flash_attn_asm_kernel", nqt * H * B, g_attn_xcd ? ", XCD-aware placement" : ""), ll_plan_append_knobs(out, cap);
    else
      snprintf(out, (size_t)cap, "flash_attn_pipe_kernel<8, %d> (%s), %d workgroups of 256 query rows%s", pp ? 1 : 0,
               pp ? "ping-pong wave groups" : "one-barrier loop", nqt * H * B, g_attn_xcd ? ", XCD-aware placement" : "");
  } else {
    snprintf(out, (size_t)cap, "flash_attn_kernel<4>, %d workgroups of 128 query rows", ((Lq + 127) / 128) * H * B);
  }
  return LL_OK;
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

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention with the q RMSNorm folded into the attention kernel (wan/modules/model.py:172,189: q = norm_q(self.q(x)), then
// attention over the text keys): q is the RAW q projection, ssq[nplanes][B * Lq] its per-row sums of squares per 128-column plane
// (ll_gemm_bf16_ssq).  Only the generated kernel has this prologue: ll_flash_attn_qnorm_ok says whether a launch is covered.
int flash_attn_asm_qn_launch(const bf16* q, const float* ssq, int nplanes, const bf16* nw, float eps, const bf16* k, const bf16* v,
                             bf16* out, int B, int Lq, int H, int ldq, int ldo, int ldk, long long k_batch_stride, int kstart, int nkeys,
                             float c, int xcd, hipStream_t stream);
extern "C" int ll_flash_attn_qnorm_ok(int H, int nkeys) {
  return (attn_asm_eligible(nkeys, H * 128) && H >= 1 && H <= 16) ? 1 : 0;      // one plane per 128 channels, at most 16 planes (NPART_MAX)
}
extern "C" int ll_flash_attn_qnorm(const ll_bf16* q, const float* ssq, const ll_bf16* norm_w, float eps, const ll_bf16* k,
                                   const ll_bf16* v, ll_bf16* out, int B, int Lq, int H, int ldq, int ldo, int ldk,
                                   long long k_batch_stride, int key_start, int nkeys, float scale, ll_stream stream) {
  LL_REQUIRE(ldq == H * 128, "ll_flash_attn_qnorm: q rows must be the whole projection output (ldq = H * 128 = %d), got %d", H * 128, ldq);
  LL_REQUIRE(ldo % 4 == 0 && ldk % 8 == 0 && ldo >= H * 128 && ldk >= H * 128, "ll_flash_attn_qnorm: bad row strides");
  LL_REQUIRE(ssq != nullptr && norm_w != nullptr, "ll_flash_attn_qnorm: ssq and norm_w are required");
  LL_REQUIRE(key_start >= 0 && nkeys > 0, "ll_flash_attn_qnorm: needs a non-empty key range");
  LL_REQUIRE(ll_flash_attn_qnorm_ok(H, nkeys), "ll_flash_attn_qnorm: %d keys x %d heads is not covered by the generated kernel under the "
             "current tuning (ask ll_flash_attn_qnorm_ok first and run ll_rmsnorm + ll_flash_attn instead)", nkeys, H);
  if (B == 0 || Lq == 0) return LL_OK;
  return flash_attn_asm_qn_launch((const bf16*)q, ssq, H, (const bf16*)norm_w, eps, (const bf16*)k, (const bf16*)v, (bf16*)out, B, Lq, H,
                                  ldq, ldo, ldk, k_batch_stride, key_start, nkeys, scale * 1.4426950408889634f, g_attn_xcd,
                                  (hipStream_t)stream);
}
