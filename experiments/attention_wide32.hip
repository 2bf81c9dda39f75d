// Flash attention, "wide32" variant for long key ranges: ONE wave per SIMD, 64 query rows per wave, 32-KEY tiles.
// (32-key tiles keep both S sets in architectural VGPRs; with 64-key tiles hipcc parked S in AGPRs: experiments/.)
//
//   workgroup = 4 waves = 256 query rows of one head; each wave owns 64 rows (two 32-row q-blocks) so every K / V^T
//   fragment read from LDS feeds TWO MFMAs -> half the LDS bytes per FLOP of the 32-row kernels (which are limited by
//   LDS-read latency/bandwidth at 2 waves/SIMD and ~250 registers), and 64 MFMAs per 64-key tile give the in-order
//   wave enough matrix work to hide its own VALU and LDS latency.
//
//   per tile t (one iteration, one barrier):
//     region 1:  S(t+1) = K(t+1) Q^T  [32 MFMA]   ||  P(t) = exp2(c S(t) - c m), row sums, bf16 pack        [VALU]
//     region 2:  O^T += V(t)^T P(t)^T [32 MFMA]   ||  row max of S(t+1), alpha, stage tile t+2 into LDS     [VALU, DS]
//     then (rare, wave-uniform branch) O *= alpha when some lane's running max moved
//   i.e. the softmax is itself split in two pipeline stages so that both regions carry ~1000 cycles of MFMA and
//   ~600-1100 cycles of VALU.  K ring: 2 stages, V ring: 3 stages (80 KiB LDS).
//
//   workgroup id -> (head, q-tile) is XCD-aware: ids that share an XCD (id % 8) get a contiguous range of
//   (head-major) tiles, so one head's K/V stream is pulled into ~2 of the 8 L2s instead of all 8.
#include "common.h"

#define KT 32
#define TILE_B (KT * 128 * 2)   // 8 KiB
#define NKB (KT / 32)
#define W_KSTAGES 2
#define W_VSTAGES 3
#define W_NW 4
#define W_ROWS (W_NW * 64)

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ unsigned w_pack2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
// PV accumulation with the O^T tile pinned in the accumulator (AGPR) half of the register file: left to itself hipcc
// allocates O in architectural VGPRs and shuttles the S tiles through AGPRs with hundreds of v_accvgpr moves per tile.
#define W_MFMA_ACC(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
__device__ __forceinline__ float w_xmax(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }
__device__ __forceinline__ float w_xsum(float x) { return x + __shfl_xor(x, 32, 64); }

__global__ __launch_bounds__(W_NW * 64, 1) void flash_attn_wide32_kernel(
    const bf16* __restrict__ Q, const bf16* __restrict__ Kc, const bf16* __restrict__ Vc, bf16* __restrict__ O, int Lq,
    int H, int ldq, int ldo, int ldk, long long k_batch_stride, int kstart, int nkeys, float c, int nqt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ksm = smem;
  char* const vsm = smem + W_KSTAGES * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware bijective remap: physical id -> logical (head-major) tile
  int nwg = nqt * H, bid = blockIdx.x;
  int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int head = lid / nqt, qt = lid % nqt;
  const int b = blockIdx.y;
  const int q0 = qt * W_ROWS + wave * 64;

  const int nt = (nkeys + KT - 1) / KT;
  const int last_valid = nkeys - (nt - 1) * KT;
  const bf16* kh = Kc + (size_t)b * k_batch_stride + (size_t)kstart * ldk + head * 128;
  const bf16* vh = Vc + (size_t)b * k_batch_stride + (size_t)kstart * ldk + head * 128;

  // Q^T fragments for both q-blocks
  bf16x8 qf[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qr = q0 + 32 * qb + r;
    qr = qr < Lq ? qr : Lq - 1;
    const bf16* qp = Q + ((size_t)b * Lq + qr) * ldq + head * 128 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }

  // staging: 512 16-byte chunks per tile / 256 threads = 2 K + 2 V chunks per thread
  uint4 kr[2], vr[2];
#define W_LOAD(T)                                                                               \
  {                                                                                             \
    int t_ = (T) < nt ? (T) : nt - 1;                                                           \
    int valid_ = (t_ == nt - 1) ? last_valid : KT;                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                          \
      int cid_ = tid + i_ * 256;                                                                \
      int key_ = cid_ >> 4, ch_ = cid_ & 15;                                                    \
      key_ = key_ < valid_ ? key_ : valid_ - 1;                                                 \
      size_t off_ = (size_t)(t_ * KT + key_) * ldk + ch_ * 8;                                   \
      kr[i_] = *reinterpret_cast<const uint4*>(kh + off_);                                      \
      vr[i_] = *reinterpret_cast<const uint4*>(vh + off_);                                      \
    }                                                                                           \
  }
#define W_STORE(KS, VS)                                                                         \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                            \
    int cid_ = tid + i_ * 256;                                                                  \
    int key_ = cid_ >> 4, ch_ = cid_ & 15;                                                      \
    *reinterpret_cast<uint4*>(ksm + (KS) * TILE_B + key_ * 256 + ((ch_ ^ (key_ & 15)) << 4)) = kr[i_];        \
    *reinterpret_cast<uint4*>(vsm + (VS) * TILE_B + key_ * 256 + ((ch_ ^ ((key_ & 3) << 2)) << 4)) = vr[i_];  \
  }

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

  f32x16 o[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qb][d][i] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  float alpha[2] = {0.f, 0.f}, mc[2];

  // ---- prologue --------------------------------------------------------------------------------------------------
  W_LOAD(0);
  W_STORE(0, 0);
  W_LOAD(1);
  W_STORE(1, 1);
  __syncthreads();
  f32x16 sA[2][NKB], sB[2][NKB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) sA[qb][kb][i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(ksm + k_off[ks] + kb * 8192);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
        sA[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], sA[qb][kb], 0, 0, 0);
    }
  __syncthreads();   // K stage 0 is restaged by iteration 0

  // Row max of a freshly computed S tile; sets m_run / alpha / mc for its softmax (stage 1 of the softmax).  Branch-free,
  // and deliberately UNMASKED: rows past the end of a ragged last tile are clamped duplicates of its last valid key,
  // so they cannot raise the maximum; their probabilities are zeroed by the -inf mask applied one iteration later.
#define W_ROWMAX(S)                                                                             \
  _Pragma("unroll") for (int qb_ = 0; qb_ < 2; ++qb_) {                                         \
    float mx_ = S[qb_][0][0];                                                                   \
    _Pragma("unroll") for (int kb_ = 0; kb_ < NKB; ++kb_)                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) mx_ = fmaxf(mx_, S[qb_][kb_][i_]);        \
    mx_ = w_xmax(mx_);                                                                          \
    float mn_ = fmaxf(m_run[qb_], mx_);                                                         \
    alpha[qb_] = __builtin_amdgcn_exp2f((m_run[qb_] - mn_) * c);                                \
    m_run[qb_] = mn_;                                                                           \
    mc[qb_] = mn_ * c;                                                                          \
  }
  W_ROWMAX(sA);   // alpha = 0 on the first tile: O and l start from zero

  int vstage = 0;
  // One iteration = tile T: SC holds S(T), SN receives S(T+1).  KNEXT / KSTORE are the compile-time K-ring stages.
#define W_ITER(SC, SN, T, KNEXT, KSTORE)                                                                                \
  {                                                                                                                     \
    const char* kn = ksm + (KNEXT) * TILE_B;                                                                            \
    const char* vc = vsm + vstage * TILE_B;                                                                             \
    W_LOAD((T) + 2);                                                                                                    \
    if ((T) == nt - 1 && last_valid < KT) { /* ragged last tile: runs once, outside the scheduled region */            \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                                  \
      _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb)                                                                  \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                  \
        int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;                                                             \
        if (key >= last_valid) SC[qb][kb][i] = -INFINITY;                                                               \
      }                                                                                                                 \
    }                                                                                                                   \
    /* region 1: S(T+1) on the matrix pipe || P(T) = exp2(c S(T) - c m), row sums, pack */                              \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                                    \
    _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) SN[qb][kb][i] = 0.f;                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                                    \
    _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb) {                                                                  \
      bf16x8 kf = *reinterpret_cast<const bf16x8*>(kn + k_off[ks] + kb * 8192);                                         \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                                  \
        SN[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], SN[qb][kb], 0, 0, 0);                      \
    }                                                                                                                   \
    uint4 pw[2][NKB][2];                                                                                                  \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                                  \
      float rs = 0.f;                                                                                                   \
      _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb)                                                                  \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                                \
        float p[8];                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                 \
          p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(SC[qb][kb][8 * s2 + j], c, -mc[qb]));                            \
          rs += p[j];                                                                                                   \
        }                                                                                                               \
        pw[qb][kb][s2] = make_uint4(w_pack2(p[0], p[1]), w_pack2(p[2], p[3]), w_pack2(p[4], p[5]), w_pack2(p[6], p[7])); \
      }                                                                                                                 \
      rs = w_xsum(rs);                                                                                                  \
      l_run[qb] = l_run[qb] * alpha[qb] + rs;                                                                           \
    }                                                                                                                   \
    /* region 2: O^T += V(T)^T P(T)^T || row max of S(T+1), staging of tile T+2 */                                      \
    _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb)                                                                    \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                                    \
    _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                                  \
      const char* a0 = vc + v_off[db] + (32 * kb + 16 * s2) * 256;                                                      \
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));                                         \
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 8 * 256));                               \
      s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                                               \
      bf16x8 vf = __builtin_bit_cast(bf16x8, vv);                                                                       \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                                  \
        W_MFMA_ACC(o[qb][db], vf, __builtin_bit_cast(bf16x8, pw[qb][kb][s2]));                                          \
    }                                                                                                                   \
    W_ROWMAX(SN); /* on the last iteration S(T+1) is a dummy tile; its result is never used */                          \
    {                                                                                                                   \
      int vs2 = vstage + 2;                                                                                             \
      vs2 = vs2 >= W_VSTAGES ? vs2 - W_VSTAGES : vs2;                                                                   \
      W_STORE(KSTORE, vs2);                                                                                             \
    }                                                                                                                   \
    /* O (complete for tile T at the old scale) moves to the scale of tile T+1 */                                       \
    if (__any((alpha[0] != 1.0f || alpha[1] != 1.0f) && (T) + 1 < nt)) {                                                \
      _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                                  \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                                     \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) o[qb][d][i] *= alpha[qb];                                          \
    }                                                                                                                   \
    __syncthreads();                                                                                                    \
    vstage = vstage == W_VSTAGES - 1 ? 0 : vstage + 1;                                                                  \
  }

  int t = 0;
  for (; t + 1 < nt; t += 2) {
    W_ITER(sA, sB, t, 1, 0);        // even tile: K(t+1) in stage 1, K(t+2) -> stage 0
    W_ITER(sB, sA, t + 1, 0, 1);    // odd tile
  }
  if (t < nt) W_ITER(sA, sB, t, 1, 0);

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qr = q0 + 32 * qb + r;
    if (qr < Lq) {
      float inv = 1.0f / l_run[qb];
      bf16* op = O + ((size_t)b * Lq + qr) * ldo + head * 128 + 4 * h;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          uint2 w;
          w.x = w_pack2(o[qb][db][4 * g4] * inv, o[qb][db][4 * g4 + 1] * inv);
          w.y = w_pack2(o[qb][db][4 * g4 + 2] * inv, o[qb][db][4 * g4 + 3] * inv);
          *reinterpret_cast<uint2*>(op + 32 * db + 8 * g4) = w;
        }
    }
  }
}

int flash_attn_wide32_launch(const ll_bf16* q, const ll_bf16* k, const ll_bf16* v, ll_bf16* out, int B, int Lq, int H,
                           int ldq, int ldo, int ldk, long long k_batch_stride, int kstart, int nkeys, float c,
                           ll_stream stream) {
  size_t lds = (W_KSTAGES + W_VSTAGES) * TILE_B;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)flash_attn_wide32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  int nqt = (Lq + W_ROWS - 1) / W_ROWS;
  dim3 grid(nqt * H, B), block(W_NW * 64);
  hipLaunchKernelGGL(flash_attn_wide32_kernel, grid, block, lds, (hipStream_t)stream, (const bf16*)q, (const bf16*)k,
                     (const bf16*)v, (bf16*)out, Lq, H, ldq, ldo, ldk, k_batch_stride, kstart, nkeys, c, nqt);
  return ll_check_launch("ll_flash_attn(wide32)");
}
