// GEMM  out[M,N] = epilogue(x[M,K] @ w[N,K]^T + bias)  on gfx950 MFMA, bf16 (v_mfma_f32_16x16x32_bf16) or
// W8A8 int8 (v_mfma_i32_16x16x64_i8, per-row activation scale x per-output-channel weight scale).
//
// Both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]), which is exactly the MFMA fragment
// shape (16 consecutive bytes of k per lane), so there are no transposes anywhere.  Common structure:
//   operands SWAPPED (A := w fragment, B := x fragment) so that each lane ends up with 4 consecutive N of one row M:
//     the bf16 epilogue store is 8 bytes per lane and the per-row gate / per-column bias are cheap to fetch
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction)
//   LDS image: rows of 128 BYTES (64 bf16 or 128 int8 of k); 16-B chunk c of row r sits at chunk position c ^ (r & 7).  The
//     DMA writes LDS linearly, so the XOR is applied to the per-lane SOURCE address and again on the ds_read_b128 side
//     (both-or-neither); every ds_read_b128 lane group then touches 16 distinct 16-B slots (conflict-free, PMC-verified)
//   workgroup id -> tile: XCD-aware (ids b, b+8, ... share an XCD/L2): each XCD gets a contiguous band of tiles, N fastest
// Two tilings behind one entry point (measured in profiles/r01_kbench_gemm_variants.txt):
//   v2: 256(M) x 128(N), 8 waves (4 x 2, 64 x 64 each), 3-stage LDS ring, counted s_waitcnt vmcnt(6) + raw s_barrier
//       -> N = 1536 GEMMs (228 workgroups ~ one per CU).  Its 64 x 64 per-wave tile needs 1/32 B of LDS reads per FLOP
//       (128 B/clk at full MFMA rate) + the DMA fill against 256 B/clk of LDS: LDS-bound around 0.8 PF.
//   v3: 256 x 256, 8 waves (2 x 4, 128(M) x 64(N) each, 128 accumulator registers), 2-stage ring -> wide GEMMs (QKV, FFN1);
//       ~40% fewer LDS bytes per FLOP (4096^3: 1.00-1.05 PF vs 0.90).
// (A first 128 x 128 / 4-wave / 2-barrier kernel, ~0.7 PF, was retired; experiments with up-front double fragment sets and
//  a DMA-issue stagger between the two waves of a SIMD measured 0...-5% and were not kept.)
#include <stdio.h>
#include <string.h>
#include <atomic>

#include "gemm_common.h"

// ---------------------------------------------------------------------------------------------------------------
// v2: 256 x 128 tile, 3-stage ring (3 x 48 KiB).  Two K-tiles stay in flight across the barrier: the only wait in the
// loop is a COUNTED s_waitcnt vmcnt(6) (the 6 DMA instructions of the newest tile may still be outstanding).
#define V2_BM 256
#define V2_STAGE ((V2_BM + BN) * ROWB)   // 48 KiB

template <int EPI, bool I8>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v2(const char* __restrict__ X, const char* __restrict__ Wt,
                                                         bf16* __restrict__ Y, int M, int N, int nk, size_t xrow_bytes,
                                                         size_t wrow_bytes, int ldo, int ntm, int ntn, int gm, int lds_epi, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Ty<I8>::frag frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int lid = xcd_remap(blockIdx.x, ntm * ntn), mt_, nt_;
  tile_of(lid, ntm, ntn, gm, mt_, nt_);
  const int m0 = mt_ * V2_BM, n0 = nt_ * BN;

  typename Ty<I8>::acc acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = acc_zero<I8>();

  auto stage = [&](int kt, int slot) {
    char* base = smem + slot * V2_STAGE;
    stage_rows(X, xrow_bytes, m0, M, kt * ROWB, base, wave * 4, 4, lane);                  // 32 x 1 KiB over 8 waves
    stage_rows(Wt, wrow_bytes, n0, N, kt * ROWB, base + V2_BM * ROWB, wave * 2, 2, lane);  // 16 x 1 KiB over 8 waves
  };
  stage(0, 0);
  if (nk > 1) stage(1, 1);

  const int fr = lane & 15, fg = lane >> 4;
  const bool live = m0 + wm * 64 < M;       // wave-uniform
  int slot = 0;
#define V2_READ(WF, XF, KS)                                                                      \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
    int ch_ = (KS) * 4 + fg;                                                                     \
    int rw = wn * 64 + t * 16 + fr;                                                              \
    WF[t] = *reinterpret_cast<const frag_t*>(ws + rw * ROWB + ((ch_ ^ (rw & 7)) << 4));          \
    int rx = wm * 64 + t * 16 + fr;                                                              \
    XF[t] = *reinterpret_cast<const frag_t*>(xs + rx * ROWB + ((ch_ ^ (rx & 7)) << 4));          \
  }
#define V2_MMA(WF, XF)                                                                           \
  _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                  \
  _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = Ty<I8>::mma(WF[a], XF[b], acc[a][b]);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt landed; tile kt+1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every wave's share of tile kt is in LDS; slot (kt+2)%3 is no longer being read
    const bool do_stage = kt + 2 < nk && !(lds_epi & 0x100);          // (0x100: timing experiment, results invalid)
    int s2 = slot + 2;
    s2 = s2 >= 3 ? s2 - 3 : s2;
    if (do_stage) stage(kt + 2, s2);
    const char* xs = smem + slot * V2_STAGE;
    const char* ws = xs + V2_BM * ROWB;
    if (!live || (lds_epi & 0x200)) {   // rows past M (last m-tile): stage and sync only (idle matrix pipes are speed elsewhere)
      slot = slot == 2 ? 0 : slot + 1;
      continue;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t wf[4], xf[4];
      V2_READ(wf, xf, ks);
      V2_MMA(wf, xf);
    }
    __builtin_amdgcn_s_setprio(0);
    slot = slot == 2 ? 0 : slot + 1;
  }
#undef V2_READ
#undef V2_MMA
  if (lds_epi) {
    __builtin_amdgcn_s_barrier();      // every wave has read its last K-step's fragments: the ring is free
    gemm_epilogue_lds<EPI, I8, 4, 4>(acc, Y, M, N, ldo, m0 + wm * 64, n0 + wn * 64, lane, smem + wave * (64 * EPI_ROW_BYTES(4)), ea);
  } else {
    gemm_epilogue<EPI, I8, 4, 4>(acc, Y, M, N, ldo, m0 + wm * 64, n0 + wn * 64, fr, fg, ea);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v3: 256 x 256 tile, per-wave 128(M) x 64(N), 2-stage ring (2 x 64 KiB), one barrier per K-step; the 64 MFMAs of a
// K-step cover the next tile's DMA latency.
#define V3_BM 256
#define V3_BN 256
#define V3_STAGE ((V3_BM + V3_BN) * ROWB)   // 64 KiB

template <int EPI, bool I8>
__global__ __launch_bounds__(512, 2) void gemm_kernel_v3(const char* __restrict__ X, const char* __restrict__ Wt,
                                                         bf16* __restrict__ Y, int M, int N, int nk, size_t xrow_bytes,
                                                         size_t wrow_bytes, int ldo, int ntm, int ntn, int gm, int lds_epi, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Ty<I8>::frag frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int lid = xcd_remap(blockIdx.x, ntm * ntn), mt_, nt_;
  tile_of(lid, ntm, ntn, gm, mt_, nt_);
  const int m0 = mt_ * V3_BM, n0 = nt_ * V3_BN;

  typename Ty<I8>::acc acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = acc_zero<I8>();

  auto stage = [&](int kt, int slot) {
    char* base = smem + slot * V3_STAGE;
    stage_rows(X, xrow_bytes, m0, M, kt * ROWB, base, wave * 4, 4, lane);
    stage_rows(Wt, wrow_bytes, n0, N, kt * ROWB, base + V3_BM * ROWB, wave * 4, 4, lane);
  };
  stage(0, 0);

  const int fr = lane & 15, fg = lane >> 4;
  const bool live = m0 + wm * 128 < M;      // wave-uniform
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // tile kt is in LDS for every wave; the other stage is no longer being read
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* xs = smem + (kt & 1) * V3_STAGE;
    const char* ws = xs + V3_BM * ROWB;
    if (!live) continue;               // rows past M: stage and sync only (see v2)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t wf[4], xf[8];
      int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int rw = wn * 64 + t * 16 + fr;
        wf[t] = *reinterpret_cast<const frag_t*>(ws + rw * ROWB + ((ch ^ (rw & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        int rx = wm * 128 + t * 16 + fr;
        xf[t] = *reinterpret_cast<const frag_t*>(xs + rx * ROWB + ((ch ^ (rx & 7)) << 4));
      }
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = Ty<I8>::mma(wf[a], xf[b], acc[a][b]);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  gemm_epilogue<EPI, I8, 4, 8>(acc, Y, M, N, ldo, m0 + wm * 128, n0 + wn * 64, fr, fg, ea);
}

// ---------------------------------------------------------------------------------------------------------------
// v5: v3's structure (256 rows, 2-stage ring, one barrier per K-step) with the column width as a parameter, for shapes where
// 256-column tiles quantise badly on 256 CUs: WM x WN waves of (MT x NT) 16 x 16 tiles, BN = WN * NT * 16.
//   <2, 4, 8, 3>: 256 x 192 (QKV, N = 4608: 456 tiles = 2 rounds of 0.75 instead of 342 = 2 rounds of 1.0)
//   <4, 2, 4, 7>: 256 x 224 (FFN1, N = 8960: 760 tiles = 2.97 rounds of 0.875 instead of 665 = 3 rounds of 1.0)
template <int EPI, bool I8, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(512, 1) void gemm_kernel_v5(const char* __restrict__ X, const char* __restrict__ Wt,
                                                         bf16* __restrict__ Y, int M, int N, int nk, size_t xrow_bytes,
                                                         size_t wrow_bytes, int ldo, int ntm, int ntn, int gm, int lds_epi, EpiArgs ea) {
  static_assert(WM * WN == 8 && WM * MT * 16 == 256, "8 waves, 256 rows");
  constexpr int BNv = WN * NT * 16, STAGE = (256 + BNv) * ROWB, NB = BNv / 8;   // NB = B pieces of 8 rows per K-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Ty<I8>::frag frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int lid = xcd_remap(blockIdx.x, ntm * ntn), mt_, nt_;
  tile_of(lid, ntm, ntn, gm, mt_, nt_);
  const int m0 = mt_ * 256, n0 = nt_ * BNv;

  typename Ty<I8>::acc acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = acc_zero<I8>();

  auto stage = [&](int kt, int slot) {
    char* base = smem + slot * STAGE;
    stage_rows(X, xrow_bytes, m0, M, kt * ROWB, base, wave * 4, 4, lane);
    stage_rows(Wt, wrow_bytes, n0, N, kt * ROWB, base + 256 * ROWB, wave * (NB / 8), NB / 8, lane);
    if (NB % 8 != 0 && wave < NB % 8) stage_rows(Wt, wrow_bytes, n0, N, kt * ROWB, base + 256 * ROWB, (NB / 8) * 8 + wave, 1, lane);
  };
  stage(0, 0);

  const int fr = lane & 15, fg = lane >> 4;
  const bool live = m0 + wm * MT * 16 < M;      // wave-uniform
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // tile kt is in LDS for every wave; the other stage is no longer being read
    if (kt + 1 < nk && !(lds_epi & 0x100)) stage(kt + 1, (kt + 1) & 1);      // (0x100 / 0x200: timing experiments, results invalid)
    const char* xs = smem + (kt & 1) * STAGE;
    const char* ws = xs + 256 * ROWB;
    if (!live || (lds_epi & 0x200)) continue;               // rows past M: stage and sync only (see v2)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag_t wf[NT], xf[MT];
      int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int rw = wn * NT * 16 + t * 16 + fr;
        wf[t] = *reinterpret_cast<const frag_t*>(ws + rw * ROWB + ((ch ^ (rw & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        int rx = wm * MT * 16 + t * 16 + fr;
        xf[t] = *reinterpret_cast<const frag_t*>(xs + rx * ROWB + ((ch ^ (rx & 7)) << 4));
      }
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int a = 0; a < NT; ++a) acc[a][b] = Ty<I8>::mma(wf[a], xf[b], acc[a][b]);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  if (lds_epi) {
    __builtin_amdgcn_s_barrier();
    gemm_epilogue_lds<EPI, I8, NT, MT>(acc, Y, M, N, ldo, m0 + wm * MT * 16, n0 + wn * NT * 16, lane,
                                       smem + wave * (MT * 16 * EPI_ROW_BYTES(NT)), ea);
  } else {
    gemm_epilogue<EPI, I8, NT, MT>(acc, Y, M, N, ldo, m0 + wm * MT * 16, n0 + wn * NT * 16, fr, fg, ea);
  }
}

// runtime tuning switches (A/B experiments from tools/kbench; defaults are the shipped configuration)
static int g_gemm_variant = 0;
static int g_gemm_variant_wide = 0;  // like gemm_variant, but only for N >= 4096 (QKV, FFN1): A/B of the wide tilings alone
// 1: epilogue staged through LDS (whole-line residual loads / stores) in the v2 / v5 tilings, except the GELU epilogue, whose
// register form measured 1.6 % faster (FFN1 127.2 vs 129.3 us; everything else 1.5-11 % faster staged); 2: all; 0: none
static int g_gemm_lds_epi = 1;
static int g_gemm_asm = 35;        // generated kernels where they cover the call (bits 0, 1), persistent form for multi-round launches (bit 5)
extern int g_gemm_asm_persistent;
static int g_gemm_group_m = 4;     // m-tiles per group in the workgroup -> tile walk (tile_of); <= 1: N fastest (round 1's order)
void ll_set_attn_variant_internal(int v);
void ll_set_attn_xcd_internal(int v);
void ll_set_attn_pp_min_internal(int v);
void ll_set_attn_asm_min_internal(int v);
void ll_set_attn_asm_internal(int v);
void ll_set_conv_halo_internal(int v);
extern "C" int ll_set_tuning(const char* key, int value) {
  if (!strcmp(key, "gemm_variant")) { g_gemm_variant = value; return LL_OK; }
  if (!strcmp(key, "gemm_group_m")) { g_gemm_group_m = value; return LL_OK; }
  if (!strcmp(key, "gemm_variant_wide")) { g_gemm_variant_wide = value; return LL_OK; }
  if (!strcmp(key, "gemm_lds_epi")) {
#ifndef LL_GEMM_DIAG      // bits 0x100 / 0x200 (K-loop staging / compute switched off: results invalid) exist for timing builds only
    if (value < 0 || value > 2) { ll_set_error("ll_set_tuning: gemm_lds_epi=%d (0, 1 or 2; diagnostic bits need -DLL_GEMM_DIAG)", value); return LL_ERR_INVALID_ARG; }
#endif
    g_gemm_lds_epi = value;
    return LL_OK;
  }
  if (!strcmp(key, "attn_variant")) { ll_set_attn_variant_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_xcd")) { ll_set_attn_xcd_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_pp_min_keys")) { ll_set_attn_pp_min_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_asm_min_keys")) { ll_set_attn_asm_min_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_asm")) { ll_set_attn_asm_internal(value); return LL_OK; }
  if (!strcmp(key, "conv_halo")) { ll_set_conv_halo_internal(value); return LL_OK; }
  if (!strcmp(key, "gemm_asm")) { g_gemm_asm = value; g_gemm_asm_persistent = (value & 32) ? 1 : 0; return LL_OK; }
  ll_set_error("ll_set_tuning: unknown key %s", key);
  return LL_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------------------------------------------
// Small-M linear (time embedding, M = B*F <= 8 rows): one wave per output column, weights streamed once.
__global__ __launch_bounds__(256) void linear_small_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                           const bf16* __restrict__ bias, bf16* __restrict__ out, int M,
                                                           int N, int K, int act_in, int act_out) {
  int lane = threadIdx.x & 63;
  int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + (size_t)n * K + k);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < M) {
        bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + (size_t)m * K + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xj = (float)xv[j];
          if (act_in == 1) xj = rbf(silu(xj));
          acc[m] += xj * (float)wv[j];
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (m < M) {
      float s = wave_sum(acc[m]);
      if (lane == 0) {
        float v = rbf(s + (float)bias[n]);
        if (act_out == 1) v = silu(v);
        out[(size_t)m * N + n] = (bf16)v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Per-row symmetric int8 quantisation: scale[r] = max|x[r,:]| / 127 (1 if the row is all zero), q = rint(x / scale).
// One wave per row, two passes over the row (the second one is served by L2).  Used for activations (per token) and,
// once at load time, for weights (rows of [N,K] = per output channel).
__global__ __launch_bounds__(256) void quantize_rows_kernel(const bf16* __restrict__ x, int8_t* __restrict__ q,
                                                            float* __restrict__ scale, int rows, int K, int ldx) {
  int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16* xr = x + (size_t)row * ldx;
  float mx = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf((float)v[j]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sc = mx > 0.f ? mx / 127.0f : 1.0f;
  float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  int8_t* qr = q + (size_t)row * K;
  for (int k = lane * 8; k < K; k += 512) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + k);
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int a = __float2int_rn((float)v[j] * inv), b = __float2int_rn((float)v[4 + j] * inv);
      a = a < -127 ? -127 : (a > 127 ? 127 : a);
      b = b < -127 ? -127 : (b > 127 ? 127 : b);
      lo |= (unsigned)(a & 0xFF) << (8 * j);
      hi |= (unsigned)(b & 0xFF) << (8 * j);
    }
    *reinterpret_cast<uint2*>(qr + k) = make_uint2(lo, hi);
  }
}

// The same arithmetic with the row RESIDENT IN REGISTERS between the two passes (K <= 512 NCH: the FFN hidden, 8960 = 17.5 x 512, is
// 18 x 16 bytes per lane): one read of the row instead of two.  Every chunk's load is issued before the first maximum is taken.
template <int NCH>
__global__ __launch_bounds__(256) void quantize_rows_reg_kernel(const bf16* __restrict__ x, int8_t* __restrict__ q,
                                                                float* __restrict__ scale, int rows, int K, int ldx) {
  int lane = threadIdx.x & 63;
  int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= rows) return;
  const bf16* xr = x + (size_t)row * ldx;
  bf16x8 v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int k = lane * 8 + 512 * i;
    if (k < K) v[i] = *reinterpret_cast<const bf16x8*>(xr + k);
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = (bf16)0.f;
  }
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf((float)v[i][j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sc = mx > 0.f ? mx / 127.0f : 1.0f;
  float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  int8_t* qr = q + (size_t)row * K;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int k = lane * 8 + 512 * i;
    if (k >= K) continue;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int a = __float2int_rn((float)v[i][j] * inv), b = __float2int_rn((float)v[i][4 + j] * inv);
      a = a < -127 ? -127 : (a > 127 ? 127 : a);
      b = b < -127 ? -127 : (b > 127 ? 127 : b);
      lo |= (unsigned)(a & 0xFF) << (8 * j);
      hi |= (unsigned)(b & 0xFF) << (8 * j);
    }
    *reinterpret_cast<uint2*>(qr + k) = make_uint2(lo, hi);
  }
}

// variant 2 = 256x128 / 3-stage ring, 3 = 256x256 (128x64 per wave), 5 = 256x192, 6 = 256x224;
// 0 = auto: the shape with the smallest   rounds(on 256 CUs) x columns x per-flop cost   (v2's 64x64 wave tile costs ~15 %
// more per flop than the 128-row ones; a tile count below the CU count is one round of whatever fills most CUs).
static int pick_gemm_variant(int M, int N) {
  int variant = g_gemm_variant;
  if (N >= 4096 && g_gemm_variant_wide >= 2 && g_gemm_variant_wide <= 6) variant = g_gemm_variant_wide;
  if (variant < 2 || variant > 6 || variant == 4) {
    const int ntm_ = (M + 255) / 256;
    auto cost = [&](int bn, double eff) {
      long tiles = (long)ntm_ * ((N + bn - 1) / bn);
      double rounds = tiles <= 256 ? 1.0 + (256 - tiles) / 256.0 * 0.6 : (double)((tiles + 255) / 256);   // under-filled: idle CUs
      return rounds * bn * eff;
    };
    double c2 = cost(128, 1.15), c3 = cost(256, 1.0), c5 = cost(192, 1.03), c6 = cost(224, 1.03);
    variant = 2;
    double best = c2;
    if (M >= 2048 && N >= 1024) {
      if (c3 < best) { best = c3; variant = 3; }
      if (c5 < best) { best = c5; variant = 5; }
      if (c6 < best) { best = c6; variant = 6; }
    }
  }
  return variant;
}

// Which kernel instance and tile ll_gemm_bf16 / ll_gemm_w8a8 launch for this shape under the current tuning (host only;
// bench.py's per-kernel table takes its kernel names from here instead of hard-coding them).
extern "C" int ll_gemm_plan(int M, int N, int K, int int8, char* out, int cap) {
  LL_REQUIRE(out != nullptr && cap > 0, "ll_gemm_plan: needs an output buffer");
  const int v = pick_gemm_variant(M, N);
  const int bn = v == 3 ? 256 : v == 5 ? 192 : v == 6 ? 224 : 128;
  char walk[48];
  if (g_gemm_group_m > 1) snprintf(walk, sizeof walk, ", groups of %d m-tiles", g_gemm_group_m);
  else snprintf(walk, sizeof walk, ", N fastest");
  const char* name = v == 2 ? "gemm_kernel_v2" : v == 3 ? "gemm_kernel_v3" : "gemm_kernel_v5";
  int ntm = (M + 255) / 256, ntn = (N + bn - 1) / bn;
  snprintf(out, (size_t)cap, "%s<%s> tile 256x%d, %d workgroups%s", name, int8 ? "i8" : "bf16", bn, ntm * ntn, walk);
  (void)K;
  return LL_OK;
}

// ll_gemm_plan for a call whose epilogue is known: names the generated kernel where ll_gemm_bf16 takes it under the current tuning;
// `plain` = 1 when the call has no V-cache output and no per-batch modulation vector (the block linears of the pipeline except QKV).
extern "C" int ll_gemm_plan_epi(int M, int N, int K, int int8, int epilogue, int plain, char* out, int cap);

// ===============================================================================================================
// tuning key gemm_asm (declared near ll_set_tuning): bit 0 = bf16 block linears on the generated one-wave-per-SIMD kernels (gemm_asm.hip) where a
                                // tile width fits (FFN1: 256 x 224 + GELU; N <= 2048: 256 x 128 with bias / gate-residual / residual);
                                // (bit 1 was "also in place of the split-K kernel": that kernel is gone, experiments/gemm_r02_variants.hip)
int gemm_asm_launch(const bf16* x, const bf16* w, bf16* out, int M, int N, int K, int ldx, int ldo, int epilogue, const EpiArgs& ea,
                    int gm, hipStream_t s);
int gemm_asm_width(int M, int N, int K, int ldx, int epilogue, bool plain, bool has_v, bool v_ok, int frame_len);
int gemm_ksplit_splits(int M, int N, int K, int cus);
int gemm_asm_launch_i8(const int8_t* x, const int8_t* w, bf16* out, int M, int N, int K, int ldo, int epilogue, const EpiArgs& ea,
                       int gm, hipStream_t s);
int gemm_asm_ksplit_launch(const bf16* x, const bf16* w, const bf16* bias, bf16* out, int M, int N, int K, int ldx, int ldo,
                           int epilogue, const bf16* res, float* workspace, int splits, int gm, hipStream_t s, const bf16* norm_w,
                           float eps, bf16* h_out);
extern "C" int ll_t5_rmsnorm(const ll_bf16* x, const ll_bf16* w, ll_bf16* out, int rows, int C, float eps, ll_stream stream);
const char* gemm_asm_plan(int M, int N, int wn, int epilogue, char* out, int cap, bool i8);
int gemm_asm_width_i8(int M, int N, int K, int epilogue, bool plain, bool has_v, bool v_ok, int frame_len);
static bool gemm_asm_wanted(int epilogue) {
  return (g_gemm_asm & 1) && !((g_gemm_asm & 4) && epilogue == LL_EPI_BIAS_GELU) && !((g_gemm_asm & 8) && epilogue != LL_EPI_BIAS_GELU);
}

template <bool I8>
static int launch_gemm(const void* x, const void* w, bf16* out, int M, int N, int K, size_t xrow_bytes, size_t wrow_bytes,
                       int ldo, int epilogue, const EpiArgs& ea, hipStream_t s) {
  // gemm_asm: bit 0 = generated kernels for the shapes they cover; bit 2 / bit 3 leave the GELU (256 x 224) / the 128-wide kernels
  // out (A/B of their share in the pipeline's power budget)
  if (!I8 && gemm_asm_wanted(epilogue) && wrow_bytes == (size_t)K * 2) {
    const int r = gemm_asm_launch((const bf16*)x, (const bf16*)w, out, M, N, K, (int)(xrow_bytes / 2), ldo, epilogue, ea, g_gemm_group_m, s);
    if (r) return r < 0 ? r : 0;                      // launched, or failed (error code); 0 = not covered: the HIP kernels below
  }
  if (I8 && (g_gemm_asm & 16) && gemm_asm_wanted(epilogue) && xrow_bytes == (size_t)K && wrow_bytes == (size_t)K) {      // bit 4: W8A8 on the generated kernels
    const int r = gemm_asm_launch_i8((const int8_t*)x, (const int8_t*)w, out, M, N, K, ldo, epilogue, ea, g_gemm_group_m, s);
    if (r) return r < 0 ? r : 0;
  }
  const int kbytes = I8 ? K : 2 * K;
  const int nk = kbytes / ROWB;
  const int variant = pick_gemm_variant(M, N);
  const bool v3 = (variant == 3);
  const bool v5 = (variant == 5), v6 = (variant == 6);
  int bm = (v3 || v5 || v6) ? 256 : V2_BM, bn = v3 ? V3_BN : v5 ? 192 : v6 ? 224 : BN;
  int ntm = (M + bm - 1) / bm, ntn = (N + bn - 1) / bn;
  dim3 grid(ntm * ntn), block(512);
  const int gm = g_gemm_group_m;
  const int lds_epi = ((g_gemm_lds_epi & 3) == 2 || ((g_gemm_lds_epi & 3) == 1 && epilogue != LL_EPI_BIAS_GELU) ? 1 : 0) |
                      (g_gemm_lds_epi & 0x300);   // 0x100 / 0x200: timing experiments (no in-loop staging / no compute), results invalid
  size_t lds = v3 ? 2 * V3_STAGE : (v5 || v6) ? 2 * (size_t)(256 + bn) * ROWB : 3 * V2_STAGE;
#define LAUNCH(E)                                                                                                      \
  do {                                                                                                                 \
    if (v5 || v6) {                                                                                                    \
      (void)ll_lds_attr((const void*)gemm_kernel_v5<E, I8, 2, 4, 8, 3>, 2 * (256 + 192) * ROWB);                       \
      (void)ll_lds_attr((const void*)gemm_kernel_v5<E, I8, 4, 2, 4, 7>, 2 * (256 + 224) * ROWB);                       \
      if (v5)                                                                                                          \
        hipLaunchKernelGGL((gemm_kernel_v5<E, I8, 2, 4, 8, 3>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                           xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                                 \
      else                                                                                                             \
        hipLaunchKernelGGL((gemm_kernel_v5<E, I8, 4, 2, 4, 7>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                           xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                                 \
    } else if (v3) {                                                                                                   \
      (void)ll_lds_attr((const void*)gemm_kernel_v3<E, I8>, (int)lds);                                                 \
      hipLaunchKernelGGL((gemm_kernel_v3<E, I8>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk,   \
                         xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                      \
    } else {                                                                                                           \
      (void)ll_lds_attr((const void*)gemm_kernel_v2<E, I8>, (int)lds);                                          \
      hipLaunchKernelGGL((gemm_kernel_v2<E, I8>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                         xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                      \
    }                                                                                                                  \
  } while (0)
  switch (epilogue) {
    case LL_EPI_BIAS: LAUNCH(LL_EPI_BIAS); break;
    case LL_EPI_BIAS_GELU: LAUNCH(LL_EPI_BIAS_GELU); break;
    case LL_EPI_BIAS_GATE_RES: LAUNCH(LL_EPI_BIAS_GATE_RES); break;
    default: LAUNCH(LL_EPI_BIAS_RES); break;
  }
#undef LAUNCH
  return LL_OK;
}

static int check_epilogue(const char* fn, int M, int N, int ldo, int epilogue, const void* bias, const void* res,
                          const void* e, const void* mod, int nmod, int gate_idx, int rows_per_batch, int frame_len) {
  LL_REQUIRE(N > 0 && N % 8 == 0, "%s: N=%d must be a positive multiple of 8", fn, N);
  LL_REQUIRE(ldo >= N && ldo % 4 == 0, "%s: ldo=%d must be >= N and a multiple of 4", fn, ldo);
  LL_REQUIRE(bias != nullptr, "%s: bias is required", fn);
  LL_REQUIRE(epilogue >= 0 && epilogue <= 3, "%s: unknown epilogue %d", fn, epilogue);
  if (epilogue == LL_EPI_BIAS_GATE_RES) {
    LL_REQUIRE(res && e, "%s: gate-residual epilogue needs res and e (mod may be NULL: e then holds bf16(mod + e))", fn);
    LL_REQUIRE(frame_len > 0 && rows_per_batch > 0 && rows_per_batch % frame_len == 0 && M % rows_per_batch == 0,
               "%s: rows_per_batch=%d / frame_len=%d do not tile M=%d", fn, rows_per_batch, frame_len, M);
    LL_REQUIRE(gate_idx >= 0 && gate_idx < nmod, "%s: gate_idx %d outside nmod %d", fn, gate_idx, nmod);
  }
  if (epilogue == LL_EPI_BIAS_RES) LL_REQUIRE(res != nullptr, "%s: residual epilogue needs res", fn);
  return LL_OK;
}

extern "C" int ll_gemm_bf16(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K,
                            int ldx, int ldo, int epilogue, const ll_bf16* res, const ll_bf16* e, const ll_bf16* mod,
                            int nmod, int gate_idx, int rows_per_batch, int frame_len, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 64 == 0, "ll_gemm_bf16: K=%d must be a positive multiple of 64", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16: ldx=%d must be >= K and a multiple of 8", ldx);
  int rc = check_epilogue("ll_gemm_bf16", M, N, ldo, epilogue, bias, res, e, mod, nmod, gate_idx, rows_per_batch, frame_len);
  if (rc) return rc;
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, (const bf16*)e, (const bf16*)mod, nullptr, nullptr, nmod, gate_idx,
             rows_per_batch, frame_len, frame_len > 0 && rows_per_batch > 0 ? rows_per_batch / frame_len : 0};
  if (int lrc = launch_gemm<false>(x, w, (bf16*)out, M, N, K, (size_t)ldx * 2, (size_t)K * 2, ldo, epilogue, ea, (hipStream_t)stream)) return lrc;
  return ll_check_launch("ll_gemm_bf16");
}

// Small-M split-K (gemm_asm.hip): how many K-ranges the call would be cut into on this device (0 = the path is not taken: the
// shape already fills half the device, N % 128, K too short, no device, or the generated kernels are switched off).
static int device_cus();
// ll_gemm_bf16 (LL_EPI_BIAS) that also leaves ssq[N / 128][M] (fp32): per row and 128-column n-tile the sum of squares of the bf16
// outputs -- the statistics of the RMSNorm that follows the projection, applied by the consumer (ll_flash_attn_qnorm).
int gemm_asm_ssq_launch(const bf16* x, const bf16* w, const bf16* bias, bf16* out, float* ssq, int M, int N, int K, int ldx, int ldo, int gm,
                        hipStream_t s);
extern "C" int ll_gemm_ssq_planes(int M, int N, int K) {
  if (!gemm_asm_wanted(LL_EPI_BIAS)) return 0;
  return gemm_asm_width(M, N, K, K, LL_EPI_BIAS, true, false, false, 0) == 128 && N / 128 <= 16 ? N / 128 : 0;
}
extern "C" int ll_gemm_bf16_ssq(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, float* ssq, int M, int N, int K,
                                int ldx, int ldo, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 64 == 0, "ll_gemm_bf16_ssq: K=%d must be a positive multiple of 64", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16_ssq: ldx=%d must be >= K and a multiple of 8", ldx);
  int rc = check_epilogue("ll_gemm_bf16_ssq", M, N, ldo, LL_EPI_BIAS, bias, nullptr, nullptr, nullptr, 0, 0, 0, 0);
  if (rc) return rc;
  LL_REQUIRE(ssq != nullptr && ((size_t)ssq & 3) == 0, "ll_gemm_bf16_ssq: ssq is required");
  LL_REQUIRE(ll_gemm_ssq_planes(M, N, K) > 0 && ldx == K, "ll_gemm_bf16_ssq: %d x %d x %d is not covered by the generated kernel under the current "
             "tuning (ask ll_gemm_ssq_planes first and run ll_gemm_bf16 + ll_rmsnorm instead)", M, N, K);
  if (M == 0) return LL_OK;
  const int r = gemm_asm_ssq_launch((const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)out, ssq, M, N, K, ldx, ldo, g_gemm_group_m,
                                    (hipStream_t)stream);
  if (r < 0) return r;
  LL_REQUIRE(r == 1, "ll_gemm_bf16_ssq: the generated kernel refused a shape its plan accepted");
  return ll_check_launch("ll_gemm_bf16_ssq");
}

extern "C" int ll_gemm_ksplit_plan(int M, int N, int K) {
  if (!gemm_asm_wanted(LL_EPI_BIAS)) return 0;        // bit 0 off, or bit 3 ("leave the 128-wide kernels out"): gemm_asm_128_partial is one of them
  return gemm_ksplit_splits(M, N, K, device_cus());
}
extern "C" long long ll_gemm_ksplit_workspace_bytes(int M, int N, int K) {
  const int S = ll_gemm_ksplit_plan(M, N, K);
  return S ? (long long)S * M * N * 4 : 0;
}
extern "C" int ll_gemm_bf16_ksplit(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K,
                                   int ldx, int ldo, int epilogue, const ll_bf16* res, void* workspace,
                                   long long workspace_bytes, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 64 == 0, "ll_gemm_bf16_ksplit: K=%d must be a positive multiple of 64", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16_ksplit: ldx=%d must be >= K and a multiple of 8", ldx);
  LL_REQUIRE(epilogue == LL_EPI_BIAS || epilogue == LL_EPI_BIAS_RES, "ll_gemm_bf16_ksplit: epilogue %d (bias or bias + residual only)", epilogue);
  int rc = check_epilogue("ll_gemm_bf16_ksplit", M, N, ldo, epilogue, bias, res, nullptr, nullptr, 0, 0, 0, 0);
  if (rc) return rc;
  LL_REQUIRE(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0), "ll_gemm_bf16_ksplit: workspace_bytes without a workspace");
  if (M == 0) return LL_OK;
  const int S = ll_gemm_ksplit_plan(M, N, K);
  if (S >= 2 && workspace != nullptr && ldo % 8 == 0) {
    LL_REQUIRE(workspace_bytes >= (long long)S * M * N * 4 && ((size_t)workspace & 15) == 0,
               "ll_gemm_bf16_ksplit: workspace of %lld bytes, need %lld (16-byte aligned)", workspace_bytes, (long long)S * M * N * 4);
    const int r = gemm_asm_ksplit_launch((const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)out, M, N, K, ldx, ldo, epilogue,
                                         (const bf16*)res, (float*)workspace, S, g_gemm_group_m, (hipStream_t)stream, nullptr, 0.f, nullptr);
    if (r) return r < 0 ? r : ll_check_launch("ll_gemm_bf16_ksplit");
  }
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
  if (int lrc = launch_gemm<false>(x, w, (bf16*)out, M, N, K, (size_t)ldx * 2, (size_t)K * 2, ldo, epilogue, ea, (hipStream_t)stream)) return lrc;
  return ll_check_launch("ll_gemm_bf16_ksplit");
}

// ll_gemm_bf16_ksplit with LL_EPI_BIAS_RES followed by the T5 RMSNorm of the new residual stream (wan/modules/t5.py:57-63,119-160:
// x = x + linear(...); h = norm(x)): out = x_new [M, ldo], h_out = T5LayerNorm(x_new) [M, N].  On the small-M path the K-range sum,
// bias, residual and the norm are ONE pass over the row; otherwise it is ll_gemm_bf16 + ll_t5_rmsnorm.  Same bits either way.
extern "C" int ll_gemm_bf16_ksplit_t5norm(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K,
                                          int ldx, int ldo, const ll_bf16* res, const ll_bf16* norm_w, float eps, ll_bf16* h_out,
                                          void* workspace, long long workspace_bytes, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 64 == 0, "ll_gemm_bf16_ksplit_t5norm: K=%d must be a positive multiple of 64", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16_ksplit_t5norm: ldx=%d must be >= K and a multiple of 8", ldx);
  LL_REQUIRE(norm_w != nullptr && h_out != nullptr, "ll_gemm_bf16_ksplit_t5norm: needs the norm weight and an output for the normalised rows");
  LL_REQUIRE(ldo == N, "ll_gemm_bf16_ksplit_t5norm: ldo=%d must equal N=%d (the norm runs over whole rows)", ldo, N);
  int rc = check_epilogue("ll_gemm_bf16_ksplit_t5norm", M, N, ldo, LL_EPI_BIAS_RES, bias, res, nullptr, nullptr, 0, 0, 0, 0);
  if (rc) return rc;
  LL_REQUIRE(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0), "ll_gemm_bf16_ksplit_t5norm: workspace_bytes without a workspace");
  if (M == 0) return LL_OK;
  const int S = ll_gemm_ksplit_plan(M, N, K);
  if (S >= 2 && workspace != nullptr && N <= 4096) {
    LL_REQUIRE(workspace_bytes >= (long long)S * M * N * 4 && ((size_t)workspace & 15) == 0,
               "ll_gemm_bf16_ksplit_t5norm: workspace of %lld bytes, need %lld (16-byte aligned)", workspace_bytes, (long long)S * M * N * 4);
    const int r = gemm_asm_ksplit_launch((const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)out, M, N, K, ldx, ldo, LL_EPI_BIAS_RES,
                                         (const bf16*)res, (float*)workspace, S, g_gemm_group_m, (hipStream_t)stream, (const bf16*)norm_w, eps,
                                         (bf16*)h_out);
    if (r) return r < 0 ? r : ll_check_launch("ll_gemm_bf16_ksplit_t5norm");
  }
  rc = ll_gemm_bf16_ksplit(x, w, bias, out, M, N, K, ldx, ldo, LL_EPI_BIAS_RES, res, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return ll_t5_rmsnorm(out, norm_w, h_out, M, N, eps, stream);
}

// Split-K form of ll_gemm_bf16 (gemm_kernel_v4sk): same arguments plus a workspace.  Taken when N is a multiple of 256, K a
// multiple of 128 and the 2 x (M / 256) x (N / 256) workgroups fit the device in one round; every other shape runs ll_gemm_bf16's
// kernels (the workspace is then unused).  workspace: >= ll_gemm_splitk_workspace_bytes(M, N) bytes, 16-byte aligned, ZEROED
// once by the caller before its first use (only the error word needs it: flags carry a per-launch epoch) and afterwards owned by
// the launches of ONE stream.  ll_gemm_splitk_status() reports a timed-out hand-off.
static int device_cus() {
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

extern "C" int ll_gemm_plan_epi(int M, int N, int K, int int8, int epilogue, int plain, char* out, int cap) {
  LL_REQUIRE(out != nullptr && cap > 0, "ll_gemm_plan_epi: needs an output buffer");
  if (!int8 && gemm_asm_wanted(epilogue)) {
    // plain: 1 = an ordinary call, 0 = per-batch modulation vector (HIP kernels), 2 = the fused QKV call with its V redirect (B = 1)
    const int wn = gemm_asm_width(M, N, K, K, epilogue, plain != 0, plain == 2, plain == 2 && (2 * (N / 3)) % 192 == 0, 1);
    if (wn) { gemm_asm_plan(M, N, wn, epilogue, out, cap, false); ll_plan_append_knobs(out, cap); return LL_OK; }
  }
  if (int8 && (g_gemm_asm & 16) && gemm_asm_wanted(epilogue)) {      // bit 4: W8A8 calls on the generated kernels (launch_gemm<true>)
    const int wn = gemm_asm_width_i8(M, N, K, epilogue, plain != 0, plain == 2, plain == 2 && (2 * (N / 3)) % 192 == 0, 1);
    if (wn) { gemm_asm_plan(M, N, wn, epilogue, out, cap, true); ll_plan_append_knobs(out, cap); return LL_OK; }
  }
  return ll_gemm_plan(M, N, K, int8, out, cap);
}

extern "C" int ll_gemm_w8a8(const int8_t* xq, const float* sx, const int8_t* wq, const float* sw, const ll_bf16* bias,
                            ll_bf16* out, int M, int N, int K, int ldo, int epilogue, const ll_bf16* res,
                            const ll_bf16* e, const ll_bf16* mod, int nmod, int gate_idx, int rows_per_batch,
                            int frame_len, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 128 == 0, "ll_gemm_w8a8: K=%d must be a positive multiple of 128", K);
  LL_REQUIRE(sx && sw, "ll_gemm_w8a8: activation and weight scales are required");
  int rc = check_epilogue("ll_gemm_w8a8", M, N, ldo, epilogue, bias, res, e, mod, nmod, gate_idx, rows_per_batch, frame_len);
  if (rc) return rc;
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, (const bf16*)e, (const bf16*)mod, sx, sw, nmod, gate_idx,
             rows_per_batch, frame_len, frame_len > 0 && rows_per_batch > 0 ? rows_per_batch / frame_len : 0};
  if (int lrc = launch_gemm<true>(xq, wq, (bf16*)out, M, N, K, (size_t)K, (size_t)K, ldo, epilogue, ea, (hipStream_t)stream)) return lrc;
  return ll_check_launch("ll_gemm_w8a8");
}

// Fused QKV projection with the V third written into the KV cache (see EpiArgs::v_out): ll_gemm_bf16 / ll_gemm_w8a8 with
// LL_EPI_BIAS, N = 3 C, plus the cache destination.  The q and k thirds land in `out` [M, ldo] as usual (they still need the
// full-row RMSNorm + RoPE of ll_qk_norm_rope_kv_store, called with cache_v = NULL afterwards); the v third of `out` is left
// unwritten.  M = B * L tokens.
static int check_v_insert(const char* fn, int M, int N, int B, int L, int S, int write_start, int roped_offset, int write_len,
                          const void* cache_v) {
  LL_REQUIRE(cache_v != nullptr, "%s: cache_v is required", fn);
  LL_REQUIRE(N % 3 == 0 && (N / 3) % 8 == 0, "%s: N=%d must be 3 C with C a multiple of 8", fn, N);
  LL_REQUIRE(B > 0 && L > 0 && M == B * L, "%s: M=%d is not B=%d x L=%d", fn, M, B, L);
  LL_REQUIRE(write_len >= 0 && roped_offset >= 0 && (write_len == 0 || roped_offset + write_len <= L), "%s: write window outside the new tokens", fn);
  LL_REQUIRE(write_len == 0 || (write_start >= 0 && write_start + write_len <= S), "%s: write [%d,+%d) outside cache of %d slots", fn, write_start, write_len, S);
  return LL_OK;
}

extern "C" int ll_gemm_bf16_qkv(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K, int ldx,
                                int ldo, ll_bf16* cache_v, int B, int L, int S, int write_start, int roped_offset, int write_len,
                                ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 64 == 0, "ll_gemm_bf16_qkv: K=%d must be a positive multiple of 64", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16_qkv: ldx=%d must be >= K and a multiple of 8", ldx);
  int rc = check_epilogue("ll_gemm_bf16_qkv", M, N, ldo, LL_EPI_BIAS, bias, nullptr, nullptr, nullptr, 0, 0, 0, 0);
  if (rc) return rc;
  rc = check_v_insert("ll_gemm_bf16_qkv", M, N, B, L, S, write_start, roped_offset, write_len, cache_v);
  if (rc) return rc;
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
  ea.v_out = (bf16*)cache_v; ea.v_col0 = 2 * (N / 3); ea.v_C = N / 3; ea.v_L = L; ea.v_S = S;
  ea.v_write_start = write_start; ea.v_roped_offset = roped_offset; ea.v_write_len = write_len;
  if (int lrc = launch_gemm<false>(x, w, (bf16*)out, M, N, K, (size_t)ldx * 2, (size_t)K * 2, ldo, LL_EPI_BIAS, ea, (hipStream_t)stream)) return lrc;
  return ll_check_launch("ll_gemm_bf16_qkv");
}

extern "C" int ll_gemm_w8a8_qkv(const int8_t* xq, const float* sx, const int8_t* wq, const float* sw, const ll_bf16* bias,
                                ll_bf16* out, int M, int N, int K, int ldo, ll_bf16* cache_v, int B, int L, int S, int write_start,
                                int roped_offset, int write_len, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 128 == 0, "ll_gemm_w8a8_qkv: K=%d must be a positive multiple of 128", K);
  LL_REQUIRE(sx && sw, "ll_gemm_w8a8_qkv: activation and weight scales are required");
  int rc = check_epilogue("ll_gemm_w8a8_qkv", M, N, ldo, LL_EPI_BIAS, bias, nullptr, nullptr, nullptr, 0, 0, 0, 0);
  if (rc) return rc;
  rc = check_v_insert("ll_gemm_w8a8_qkv", M, N, B, L, S, write_start, roped_offset, write_len, cache_v);
  if (rc) return rc;
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, nullptr, nullptr, nullptr, sx, sw, 0, 0, 0, 0, 0};
  ea.v_out = (bf16*)cache_v; ea.v_col0 = 2 * (N / 3); ea.v_C = N / 3; ea.v_L = L; ea.v_S = S;
  ea.v_write_start = write_start; ea.v_roped_offset = roped_offset; ea.v_write_len = write_len;
  if (int lrc = launch_gemm<true>(xq, wq, (bf16*)out, M, N, K, (size_t)K, (size_t)K, ldo, LL_EPI_BIAS, ea, (hipStream_t)stream)) return lrc;
  return ll_check_launch("ll_gemm_w8a8_qkv");
}

extern "C" int ll_quantize_rows(const ll_bf16* x, int8_t* q, float* scale, int rows, int K, int ldx, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 8 == 0, "ll_quantize_rows: K=%d must be a positive multiple of 8", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_quantize_rows: ldx=%d must be >= K and a multiple of 8", ldx);
  if (rows == 0) return LL_OK;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  // rows of up to 9216 elements stay in registers between the maximum and the rounding pass (same arithmetic, one read)
  if (K <= 2048) hipLaunchKernelGGL(quantize_rows_reg_kernel<4>, grid, block, 0, s, (const bf16*)x, q, scale, rows, K, ldx);
  else if (K <= 9216) hipLaunchKernelGGL(quantize_rows_reg_kernel<18>, grid, block, 0, s, (const bf16*)x, q, scale, rows, K, ldx);
  else hipLaunchKernelGGL(quantize_rows_kernel, grid, block, 0, s, (const bf16*)x, q, scale, rows, K, ldx);
  return ll_check_launch("ll_quantize_rows");
}

extern "C" int ll_linear_small(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N,
                               int K, int act_in, int act_out, ll_stream stream) {
  LL_REQUIRE(M >= 0 && M <= 8, "ll_linear_small: M=%d must be <= 8", M);
  LL_REQUIRE(K % 8 == 0, "ll_linear_small: K=%d must be a multiple of 8", K);
  if (M == 0 || N == 0) return LL_OK;
  hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                     (const bf16*)w, (const bf16*)bias, (bf16*)out, M, N, K, act_in, act_out);
  return ll_check_launch("ll_linear_small");
}
