// Shared pieces of the MFMA GEMM family (gemm.hip, conv.hip): operand-type policy, LDS-DMA staging, fused epilogues.
#pragma once
#include "common.h"

#define BN 128
#define ROWB 128                    // LDS / staging row = 128 bytes of k
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct EpiArgs {
  const bf16* bias;
  const bf16* res;
  const bf16* e;
  const bf16* mod;
  const float* sx;                  // int8: per-row activation scale [M]
  const float* sw;                  // int8: per-output-channel weight scale [N]
  int nmod, gate_idx, rows_per_batch, frame_len, F;
};

template <bool I8>
struct Ty;
template <>
struct Ty<false> {
  typedef bf16x8 frag;
  typedef f32x4 acc;
  static __device__ __forceinline__ acc mma(frag a, frag b, acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Ty<true> {
  // A and B fragments are fetched with the SAME (row, 16-byte chunk) addressing, so whatever order the instruction
  // assigns to the 16 k-values inside a lane's fragment, products pair equal k: the exact int32 sum is layout-agnostic.
  typedef i32x4 frag;
  typedef i32x4 acc;
  static __device__ __forceinline__ acc mma(frag a, frag b, acc c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  }
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of `ninst` x 8 rows x 128 B starting at LDS row inst0*8; row_bytes = global row stride in bytes.
__device__ __forceinline__ void stage_rows(const char* __restrict__ src, size_t row_bytes, int row0, int nrows,
                                           int kbyte0, char* lds, int inst0, int ninst, int lane) {
#pragma unroll
  for (int i = 0; i < ninst; ++i) {
    int inst = inst0 + i;
    int r = inst * 8 + (lane >> 3);            // LDS row this lane fills
    int cg = (lane & 7) ^ (r & 7);             // global chunk that belongs at this lane's (linear) LDS position
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;          // rows past the edge re-read the last row (never stored)
    const char* g = src + (size_t)gr * row_bytes + kbyte0 + cg * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + inst * 1024), 16, 0, 0);
  }
}

// Epilogue: acc[a][b] is the 16x16 MFMA tile (n-tile a, m-tile b); lane holds out[m][n .. n+3] with
// m = mw + b*16 + (lane & 15), n = nw + a*16 + (lane >> 4)*4.  Rounding points follow the reference's bf16 modules.
template <int EPI, bool I8, int NTL, int MT>
__device__ __forceinline__ void gemm_epilogue(typename Ty<I8>::acc (&acc)[NTL][MT], bf16* __restrict__ Y, int M, int N,
                                              int ldo, int mw, int nw, int fr, int fg, const EpiArgs& ea) {
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    int m = mw + b * 16 + fr;
    if (m >= M) continue;
    const bf16* gate_e = nullptr;
    if (EPI == LL_EPI_BIAS_GATE_RES) {
      int bb = m / ea.rows_per_batch, f = (m % ea.rows_per_batch) / ea.frame_len;
      gate_e = ea.e + ((size_t)(bb * ea.F + f) * ea.nmod + ea.gate_idx) * N;
    }
    float sxm = 1.0f;
    if (I8) sxm = ea.sx[m];
#pragma unroll
    for (int a = 0; a < NTL; ++a) {
      int n = nw + a * 16 + fg * 4;
      if (n >= N) continue;
      bf16x4 bv = *reinterpret_cast<const bf16x4*>(ea.bias + n);
      float v[4];
      if (I8) {
        f32x4 swv = *reinterpret_cast<const f32x4*>(ea.sw + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf((float)acc[a][b][j] * (sxm * swv[j]) + (float)bv[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf((float)acc[a][b][j] + (float)bv[j]);
      }
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

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Workgroup id -> (m-tile, n-tile).  The XCD remap hands every XCD (= every L2) a contiguous range of `lid`s; inside it the
// tiles are walked in groups of `gm` m-tiles, m fastest: the ~32 workgroups an XCD runs at a time then cover a compact
// gm x (32 / gm) block of tiles and share gm X panels and 32 / gm W panels through the L2, instead of one X panel and 32
// different W panels (N fastest, gm <= 1).  PMC, FFN1 4680 x 8960 x 1536 with N fastest: 640 MB of L2-miss traffic per launch
// against 126 MB algorithmic -- every W panel was fetched once per m-tile (profiles/r02_pmc_shipped.md).
__device__ __forceinline__ void tile_of(int lid, int ntm, int ntn, int gm, int& mt, int& nt) {
  if (gm <= 1) {
    mt = lid / ntn;
    nt = lid % ntn;
    return;
  }
  int per = gm * ntn;
  int g = lid / per;
  int first = g * gm;
  int gs = ntm - first < gm ? ntm - first : gm;
  int w = lid - g * per;
  mt = first + w % gs;
  nt = w / gs;
}

template <bool I8>
__device__ __forceinline__ typename Ty<I8>::acc acc_zero() {
  typename Ty<I8>::acc z = {0, 0, 0, 0};
  return z;
}

