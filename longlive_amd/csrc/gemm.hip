// bf16 GEMM  out[M,N] = epilogue(x[M,K] @ w[N,K]^T + bias)  on gfx950 MFMA (v_mfma_f32_16x16x32_bf16).
//
// Both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]), which is exactly the MFMA fragment
// shape (8 consecutive k per lane), so no transposes anywhere.  Tiling:
//   workgroup 256 threads = 4 waves (2 x 2), block tile 128(M) x 128(N) x 64(K); each wave owns 64 x 64 = 4 x 4 MFMA tiles
//   operands SWAPPED (A := w fragment, B := x fragment) so that each lane ends up with 4 consecutive N of one row M:
//     the bf16 epilogue store is 8 bytes per lane and the per-row gate / per-column bias are cheap to fetch
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction), double buffered, ONE barrier per
//     K-step: tile k+1 streams in while tile k is multiplied
//   LDS image: [128 rows][64 k] bf16 = 128-B rows; 16-B chunk c of row r sits at chunk position c ^ (r & 7).  The DMA
//     writes LDS linearly, so the XOR is applied to the per-lane SOURCE address and again on the ds_read_b128 side
//     (both-or-neither); every ds_read_b128 lane group then touches 16 distinct 16-B slots (conflict-free).
//   workgroup id -> tile: XCD-aware (ids b, b+8, ... share an XCD/L2): each XCD gets a contiguous band of tiles, N fastest,
//     so the x row-panel of a band stays in that XCD's L2 while the weight panel streams through.
#include <string.h>

#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * BK * 2)  // 16 KiB per operand tile

struct EpiArgs {
  const bf16* bias;
  const bf16* res;
  const bf16* e;
  const bf16* mod;
  int nmod, gate_idx, rows_per_batch, frame_len, F;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// stage one 128 x 64 operand tile: 16 wave-instructions of 1 KiB, 4 per wave.
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ src, int ld, int row0, int nrows, int k0, char* lds,
                                           int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int inst = wave * 4 + i;
    int r = inst * 8 + (lane >> 3);            // LDS row this lane fills
    int cl = lane & 7;                         // LDS chunk position
    int cg = cl ^ (r & 7);                     // global chunk that belongs there
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;          // clamp: rows past the edge re-read the last row (never stored)
    const bf16* g = src + (size_t)gr * ld + k0 + cg * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + inst * 1024), 16, 0, 0);
  }
}

// Epilogue shared by the GEMM kernels.  acc[a][b] is the 16x16 MFMA tile (n-tile a, m-tile b) of a wave's 64 x 64 block:
// lane holds out[m][n .. n+3] with m = mw + b*16 + (lane & 15), n = nw + a*16 + (lane >> 4)*4.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[4][4], bf16* __restrict__ Y, int M, int N, int ldo, int mw,
                                              int nw, int fr, int fg, const EpiArgs& ea) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int m = mw + b * 16 + fr;
    if (m >= M) continue;
    const bf16* gate_e = nullptr;
    if (EPI == LL_EPI_BIAS_GATE_RES) {
      int bb = m / ea.rows_per_batch, f = (m % ea.rows_per_batch) / ea.frame_len;
      gate_e = ea.e + ((size_t)(bb * ea.F + f) * ea.nmod + ea.gate_idx) * N;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int n = nw + a * 16 + fg * 4;
      if (n >= N) continue;
      bf16x4 bv = *reinterpret_cast<const bf16x4*>(ea.bias + n);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = rbf(acc[a][b][j] + (float)bv[j]);
      bf16x4 o;
      if (EPI == LL_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
      } else if (EPI == LL_EPI_BIAS_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)gelu_tanh(v[j]);
      } else if (EPI == LL_EPI_BIAS_GATE_RES) {
        bf16x4 ge = *reinterpret_cast<const bf16x4*>(gate_e + n);
        bf16x4 gm = *reinterpret_cast<const bf16x4*>(ea.mod + (size_t)ea.gate_idx * N + n);
        bf16x4 rv = *reinterpret_cast<const bf16x4*>(ea.res + (size_t)m * ldo + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float g = rbf((float)gm[j] + (float)ge[j]);
          o[j] = (bf16)((float)rv[j] + rbf(v[j] * g));
        }
      } else {  // LL_EPI_BIAS_RES
        bf16x4 rv = *reinterpret_cast<const bf16x4*>(ea.res + (size_t)m * ldo + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)rv[j] + v[j]);
      }
      *reinterpret_cast<bf16x4*>(Y + (size_t)m * ldo + n) = o;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ Wt,
                                                           bf16* __restrict__ Y, int M, int N, int K, int ldx, int ldo,
                                                           int ntm, int ntn, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (X tile | W tile) = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap of the linear workgroup id
  int nwg = ntm * ntn, bid = blockIdx.x;
  int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;

  f32x4 acc[4][4];  // [n-tile][m-tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage_tile(X, ldx, m0, M, 0, smem, wave, lane);
  stage_tile(Wt, K, n0, N, 0, smem + TILE_BYTES, wave, lane);

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; everyone is done reading the other stage
    char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
    if (kt + 1 < nk) {
      char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      stage_tile(X, ldx, m0, M, (kt + 1) * BK, nxt, wave, lane);
      stage_tile(Wt, K, n0, N, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* xs = cur;
    const char* ws = cur + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], xf[4];
      int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int rw = wn * 64 + t * 16 + fr;
        wf[t] = *reinterpret_cast<const bf16x8*>(ws + rw * 128 + ((ch ^ (rw & 7)) << 4));
        int rx = wm * 64 + t * 16 + fr;
        xf[t] = *reinterpret_cast<const bf16x8*>(xs + rx * 128 + ((ch ^ (rx & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
  }

  gemm_epilogue<EPI>(acc, Y, M, N, ldo, m0 + wm * 64, n0 + wn * 64, fr, fg, ea);
}

// ---------------------------------------------------------------------------------------------------------------
// v2: 256(M) x 128(N) x 64(K) block tile, 8 waves (4 x 2, 64 x 64 each), 3-stage LDS ring (3 x 48 KiB) filled by
// LDS-DMA.  Two K-tiles stay in flight across the barrier: the only wait in the loop is a COUNTED s_waitcnt vmcnt(6)
// (= the 6 DMA instructions of the newest tile may still be outstanding) followed by a raw s_barrier, so HBM/L2
// latency is covered by two tiles of MFMA work instead of one.  One workgroup per CU (144 KiB LDS, 2 waves/SIMD).
#define V2_BM 256
#define V2_STAGE (V2_BM * 128 + BN * 128)   // 48 KiB

__device__ __forceinline__ void stage_rows(const bf16* __restrict__ src, int ld, int row0, int nrows, int k0, char* lds,
                                           int inst0, int ninst, int lane) {
#pragma unroll
  for (int i = 0; i < ninst; ++i) {
    int inst = inst0 + i;
    int r = inst * 8 + (lane >> 3);
    int cg = (lane & 7) ^ (r & 7);
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;
    const bf16* g = src + (size_t)gr * ld + k0 + cg * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + inst * 1024), 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel_v2(const bf16* __restrict__ X, const bf16* __restrict__ Wt,
                                                              bf16* __restrict__ Y, int M, int N, int K, int ldx,
                                                              int ldo, int ntm, int ntn, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x (X 256x64 | W 128x64)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int nwg = ntm * ntn, bid = blockIdx.x;
  int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int m0 = (lid / ntn) * V2_BM, n0 = (lid % ntn) * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  auto stage = [&](int kt, int slot) {
    char* base = smem + slot * V2_STAGE;
    stage_rows(X, ldx, m0, M, kt * BK, base, wave * 4, 4, lane);                   // 32 instructions over 8 waves
    stage_rows(Wt, K, n0, N, kt * BK, base + V2_BM * 128, wave * 2, 2, lane);      // 16 instructions over 8 waves
  };
  stage(0, 0);
  if (nk > 1) stage(1, 1);

  const int fr = lane & 15, fg = lane >> 4;
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt landed; tile kt+1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every wave's share of tile kt is in LDS; slot (kt+2)%3 is no longer being read
    if (kt + 2 < nk) {
      int s2 = slot + 2;
      s2 = s2 >= 3 ? s2 - 3 : s2;
      stage(kt + 2, s2);
    }
    const char* xs = smem + slot * V2_STAGE;
    const char* ws = xs + V2_BM * 128;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], xf[4];
      int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int rw = wn * 64 + t * 16 + fr;
        wf[t] = *reinterpret_cast<const bf16x8*>(ws + rw * 128 + ((ch ^ (rw & 7)) << 4));
        int rx = wm * 64 + t * 16 + fr;
        xf[t] = *reinterpret_cast<const bf16x8*>(xs + rx * 128 + ((ch ^ (rx & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    slot = slot == 2 ? 0 : slot + 1;
  }
  gemm_epilogue<EPI>(acc, Y, M, N, ldo, m0 + wm * 64, n0 + wn * 64, fr, fg, ea);
}

// ---------------------------------------------------------------------------------------------------------------
// v3: 256 x 256 x 64 block tile, 8 waves as 2(M) x 4(N), per-wave tile 128(M) x 64(N) = 8 x 4 MFMA tiles (128 accumulator
// registers).  The 64 x 64 per-wave tile of v1/v2 needs 1/32 B of LDS reads per FLOP = 128 B/clk at full MFMA rate, plus
// the LDS-DMA fill, against 256 B/clk of LDS: those kernels are LDS-bandwidth-bound (~0.7-0.8 PF).  128 x 64 per wave
// and a 256-wide block tile cut LDS bytes per FLOP by ~40%.  2-stage LDS ring (2 x 64 KiB), one workgroup per CU,
// one barrier per K-step; the 64 MFMAs of a K-step (1024 cycles) cover the next tile's DMA latency.
#define V3_BM 256
#define V3_BN 256
#define V3_STAGE ((V3_BM + V3_BN) * 128)   // 64 KiB

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_v3(f32x4 (&acc)[4][8], bf16* __restrict__ Y, int M, int N, int ldo, int mw,
                                                 int nw, int fr, int fg, const EpiArgs& ea) {
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    int m = mw + b * 16 + fr;
    if (m >= M) continue;
    const bf16* gate_e = nullptr;
    if (EPI == LL_EPI_BIAS_GATE_RES) {
      int bb = m / ea.rows_per_batch, f = (m % ea.rows_per_batch) / ea.frame_len;
      gate_e = ea.e + ((size_t)(bb * ea.F + f) * ea.nmod + ea.gate_idx) * N;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int n = nw + a * 16 + fg * 4;
      if (n >= N) continue;
      bf16x4 bv = *reinterpret_cast<const bf16x4*>(ea.bias + n);
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = rbf(acc[a][b][j] + (float)bv[j]);
      bf16x4 o;
      if (EPI == LL_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
      } else if (EPI == LL_EPI_BIAS_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)gelu_tanh(v[j]);
      } else if (EPI == LL_EPI_BIAS_GATE_RES) {
        bf16x4 ge = *reinterpret_cast<const bf16x4*>(gate_e + n);
        bf16x4 gm = *reinterpret_cast<const bf16x4*>(ea.mod + (size_t)ea.gate_idx * N + n);
        bf16x4 rv = *reinterpret_cast<const bf16x4*>(ea.res + (size_t)m * ldo + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float g = rbf((float)gm[j] + (float)ge[j]);
          o[j] = (bf16)((float)rv[j] + rbf(v[j] * g));
        }
      } else {
        bf16x4 rv = *reinterpret_cast<const bf16x4*>(ea.res + (size_t)m * ldo + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)rv[j] + v[j]);
      }
      *reinterpret_cast<bf16x4*>(Y + (size_t)m * ldo + n) = o;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel_v3(const bf16* __restrict__ X, const bf16* __restrict__ Wt,
                                                              bf16* __restrict__ Y, int M, int N, int K, int ldx,
                                                              int ldo, int ntm, int ntn, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (X 256x64 | W 256x64)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int nwg = ntm * ntn, bid = blockIdx.x;
  int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int m0 = (lid / ntn) * V3_BM, n0 = (lid % ntn) * V3_BN;

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  auto stage = [&](int kt, int slot) {
    char* base = smem + slot * V3_STAGE;
    stage_rows(X, ldx, m0, M, kt * BK, base, wave * 4, 4, lane);                   // 32 x 1 KiB over 8 waves
    stage_rows(Wt, K, n0, N, kt * BK, base + V3_BM * 128, wave * 4, 4, lane);      // 32 x 1 KiB over 8 waves
  };
  stage(0, 0);

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // tile kt is in LDS for every wave; the other stage is no longer being read
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* xs = smem + (kt & 1) * V3_STAGE;
    const char* ws = xs + V3_BM * 128;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], xf[8];
      int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int rw = wn * 64 + t * 16 + fr;
        wf[t] = *reinterpret_cast<const bf16x8*>(ws + rw * 128 + ((ch ^ (rw & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        int rx = wm * 128 + t * 16 + fr;
        xf[t] = *reinterpret_cast<const bf16x8*>(xs + rx * 128 + ((ch ^ (rx & 7)) << 4));
      }
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  gemm_epilogue_v3<EPI>(acc, Y, M, N, ldo, m0 + wm * 128, n0 + wn * 64, fr, fg, ea);
}

// runtime tuning switches (A/B experiments from tools/kbench; defaults are the shipped configuration)
static int g_gemm_variant = 0;
void ll_set_attn_variant_internal(int v);
void ll_set_attn_xcd_internal(int v);
extern "C" int ll_set_tuning(const char* key, int value) {
  if (!strcmp(key, "gemm_variant")) { g_gemm_variant = value; return LL_OK; }
  if (!strcmp(key, "attn_variant")) { ll_set_attn_variant_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_xcd")) { ll_set_attn_xcd_internal(value); return LL_OK; }
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

// ===============================================================================================================
extern "C" int ll_gemm_bf16(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K,
                            int ldx, int ldo, int epilogue, const ll_bf16* res, const ll_bf16* e, const ll_bf16* mod,
                            int nmod, int gate_idx, int rows_per_batch, int frame_len, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % BK == 0, "ll_gemm_bf16: K=%d must be a positive multiple of %d", K, BK);
  LL_REQUIRE(N > 0 && N % 8 == 0, "ll_gemm_bf16: N=%d must be a positive multiple of 8", N);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16: ldx=%d must be >= K and a multiple of 8", ldx);
  LL_REQUIRE(ldo >= N && ldo % 4 == 0, "ll_gemm_bf16: ldo=%d must be >= N and a multiple of 4", ldo);
  LL_REQUIRE(bias != nullptr, "ll_gemm_bf16: bias is required");
  LL_REQUIRE(epilogue >= 0 && epilogue <= 3, "ll_gemm_bf16: unknown epilogue %d", epilogue);
  if (epilogue == LL_EPI_BIAS_GATE_RES) {
    LL_REQUIRE(res && e && mod, "ll_gemm_bf16: gate-residual epilogue needs res, e and mod");
    LL_REQUIRE(frame_len > 0 && rows_per_batch > 0 && rows_per_batch % frame_len == 0 && M % rows_per_batch == 0,
               "ll_gemm_bf16: rows_per_batch=%d / frame_len=%d do not tile M=%d", rows_per_batch, frame_len, M);
    LL_REQUIRE(gate_idx >= 0 && gate_idx < nmod, "ll_gemm_bf16: gate_idx %d outside nmod %d", gate_idx, nmod);
  }
  if (epilogue == LL_EPI_BIAS_RES) LL_REQUIRE(res != nullptr, "ll_gemm_bf16: residual epilogue needs res");
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, (const bf16*)e, (const bf16*)mod, nmod, gate_idx, rows_per_batch,
             frame_len, frame_len > 0 && rows_per_batch > 0 ? rows_per_batch / frame_len : 0};
  hipStream_t s = (hipStream_t)stream;
  // variant: 1 = 128x128 / 4 waves, 2 = 256x128 / 8 waves / 3-stage ring, 3 = 256x256 / 8 waves (128x64 per wave).
  // 0 = auto: the 256x256 tile only where it still fills the chip (N >= 4096), else variant 2.
  int variant = g_gemm_variant;
  if (variant == 0) variant = (N >= 4096 && M >= 2048) ? 3 : 2;
  const bool v2 = (variant == 2), v3 = (variant == 3);
  int bm = v3 ? V3_BM : v2 ? V2_BM : BM, bn = v3 ? V3_BN : BN;
  int ntm = (M + bm - 1) / bm, ntn = (N + bn - 1) / bn;
  dim3 grid(ntm * ntn), block((v2 || v3) ? 512 : 256);
  size_t lds = v3 ? 2 * V3_STAGE : v2 ? 3 * V2_STAGE : 4 * TILE_BYTES;
#define LAUNCH(E)                                                                                                      \
  do {                                                                                                                 \
    if (v3) {                                                                                                          \
      static bool attr3 = false;                                                                                       \
      if (!attr3) {                                                                                                    \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel_v3<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        attr3 = true;                                                                                                  \
      }                                                                                                                \
      hipLaunchKernelGGL(gemm_bf16_kernel_v3<E>, grid, block, lds, s, (const bf16*)x, (const bf16*)w, (bf16*)out, M, N, \
                         K, ldx, ldo, ntm, ntn, ea);                                                                   \
    } else if (v2) {                                                                                                   \
      static bool attr_set = false;                                                                                    \
      if (!attr_set) {                                                                                                 \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel_v2<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        attr_set = true;                                                                                               \
      }                                                                                                                \
      hipLaunchKernelGGL(gemm_bf16_kernel_v2<E>, grid, block, lds, s, (const bf16*)x, (const bf16*)w, (bf16*)out, M, N, \
                         K, ldx, ldo, ntm, ntn, ea);                                                                   \
    } else {                                                                                                           \
      hipLaunchKernelGGL(gemm_bf16_kernel<E>, grid, block, lds, s, (const bf16*)x, (const bf16*)w, (bf16*)out, M, N, K, \
                         ldx, ldo, ntm, ntn, ea);                                                                      \
    }                                                                                                                  \
  } while (0)
  switch (epilogue) {
    case LL_EPI_BIAS: LAUNCH(LL_EPI_BIAS); break;
    case LL_EPI_BIAS_GELU: LAUNCH(LL_EPI_BIAS_GELU); break;
    case LL_EPI_BIAS_GATE_RES: LAUNCH(LL_EPI_BIAS_GATE_RES); break;
    default: LAUNCH(LL_EPI_BIAS_RES); break;
  }
#undef LAUNCH
  return ll_check_launch("ll_gemm_bf16");
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
