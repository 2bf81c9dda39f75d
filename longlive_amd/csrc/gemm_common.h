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
  // Fused QKV projection: output columns >= v_col0 (the V third) go straight into the KV cache instead of the output buffer,
  // token t of batch b to cache row v_write_start + (t - v_roped_offset) when that index is in [0, v_write_len)
  // (wan/modules/causal_model.py:264-269,302-311: `temp_v[:, write_start:local_end] = v[:, roped_offset:roped_offset + write_len]`).
  bf16* v_out = nullptr;            // cache_v [B, v_S, C] or null
  int v_col0 = 0, v_C = 0, v_L = 1, v_S = 0, v_write_start = 0, v_roped_offset = 0, v_write_len = 0;   // v_C = cache row width
};

// destination of output element block (row m, first column n) under the V redirect; nullptr = not stored
__device__ __forceinline__ bf16* epi_dest(const EpiArgs& ea, bf16* __restrict__ Y, int ldo, int m, int n) {
  if (ea.v_out == nullptr || n < ea.v_col0) return Y + (size_t)m * ldo + n;
  int bb = m / ea.v_L, wi = m - bb * ea.v_L - ea.v_roped_offset;
  if (wi < 0 || wi >= ea.v_write_len) return nullptr;
  return ea.v_out + ((size_t)bb * ea.v_S + ea.v_write_start + wi) * (size_t)ea.v_C + (n - ea.v_col0);
}

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
// Every global read of the epilogue -- residual rows, gate vectors, bias, int8 scales -- is issued up front, unconditionally,
// from clamped addresses (rows past M re-read row M-1, columns past N re-read the last 4 columns; never stored), so the
// MT x NTL residual loads of a lane are ONE round trip.  Guarded by `if (m >= M) continue` they were MT dependent round
// trips of ~1 us each at the end of every O / cross-O / FFN2 launch.
template <int EPI, bool I8, int NTL, int MT>
__device__ __forceinline__ void gemm_epilogue(typename Ty<I8>::acc (&acc)[NTL][MT], bf16* __restrict__ Y, int M, int N,
                                              int ldo, int mw, int nw, int fr, int fg, const EpiArgs& ea) {
  constexpr bool RES = (EPI == LL_EPI_BIAS_GATE_RES || EPI == LL_EPI_BIAS_RES);
  constexpr bool GATE = (EPI == LL_EPI_BIAS_GATE_RES);
  int nc[NTL];
  bf16x4 bv[NTL], gm[GATE ? NTL : 1];
  f32x4 swv[I8 ? NTL : 1];
#pragma unroll
  for (int a = 0; a < NTL; ++a) {
    int n = nw + a * 16 + fg * 4;
    nc[a] = n < N ? n : N - 4;
    bv[a] = *reinterpret_cast<const bf16x4*>(ea.bias + nc[a]);
    if (GATE && ea.mod) gm[a] = *reinterpret_cast<const bf16x4*>(ea.mod + (size_t)ea.gate_idx * N + nc[a]);
    if (I8) swv[a] = *reinterpret_cast<const f32x4*>(ea.sw + nc[a]);
  }
  // m-subtiles per batch of loads: <= 16 (m, n) subtiles in flight per lane, or the wide tilings' epilogues spill
  constexpr int CH = NTL > 4 ? 2 : (MT > 4 ? 4 : MT);
  static_assert(MT % CH == 0, "epilogue batches must tile MT");
#pragma unroll
  for (int b0 = 0; b0 < MT; b0 += CH) {
    bf16x4 rv[RES ? CH : 1][RES ? NTL : 1], ge[GATE ? CH : 1][GATE ? NTL : 1];
    float sxm[I8 ? CH : 1];
#pragma unroll
    for (int bi = 0; bi < CH; ++bi) {
      int m = mw + (b0 + bi) * 16 + fr;
      int mc = m < M ? m : M - 1;
      if (I8) sxm[bi] = ea.sx[mc];
      const bf16* gate_e = nullptr;
      if (GATE) {
        int bb = mc / ea.rows_per_batch, f = (mc % ea.rows_per_batch) / ea.frame_len;
        gate_e = ea.e + ((size_t)(bb * ea.F + f) * ea.nmod + ea.gate_idx) * N;
      }
#pragma unroll
      for (int a = 0; a < NTL; ++a) {
        if (RES) rv[bi][a] = *reinterpret_cast<const bf16x4*>(ea.res + (size_t)mc * ldo + nc[a]);
        if (GATE) ge[bi][a] = *reinterpret_cast<const bf16x4*>(gate_e + nc[a]);
      }
    }
#pragma unroll
    for (int bi = 0; bi < CH; ++bi) {
      const int b = b0 + bi;
      int m = mw + b * 16 + fr;
#pragma unroll
      for (int a = 0; a < NTL; ++a) {
        int n = nw + a * 16 + fg * 4;
        float v[4];
        if (I8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = rbf((float)acc[a][b][j] * (sxm[bi] * swv[a][j]) + (float)bv[a][j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = rbf((float)acc[a][b][j] + (float)bv[a][j]);
        }
        bf16x4 o;
        if (EPI == LL_EPI_BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
        } else if (EPI == LL_EPI_BIAS_GELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)gelu_tanh(v[j]);
        } else if (EPI == LL_EPI_BIAS_GATE_RES) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float g = ea.mod ? rbf((float)gm[a][j] + (float)ge[bi][a][j]) : (float)ge[bi][a][j];   // mod == NULL: e is bf16(mod + e)
            o[j] = (bf16)((float)rv[bi][a][j] + rbf(v[j] * g));
          }
        } else {  // LL_EPI_BIAS_RES
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)rv[bi][a][j] + v[j]);
        }
        if (m < M && n < N) {
          bf16* dst = epi_dest(ea, Y, ldo, m, n);
          if (dst) *reinterpret_cast<bf16x4*>(dst) = o;
        }
      }
    }
  }
}

// LDS-staged form of the epilogue (the GEMM kernels' K-loop rings are free by then): every wave parks its bias(+GELU)-ed
// bf16 tile in a private LDS region in the MFMA lane layout (8 bytes per lane and subtile), reads it back ROW-contiguous
// (16 bytes per lane, a row's NTL*32 bytes on consecutive lanes) and does the residual / gate arithmetic there, so that
// residual loads and output stores are whole 128-byte lines (1 KiB per wave instruction) instead of 32-byte row slivers over
// 16 rows: half the store instructions, a quarter of the touched lines per instruction.  Same values, same rounding points.
// `ep` = this wave's region, EPI_ROW_BYTES(NTL) bytes per row, MT*16 rows; the caller has passed a workgroup barrier after the
// last K-step's LDS reads.
#define EPI_ROW_BYTES(NTL) ((NTL) * 32 + 16)

// (Loading the residual rows at kernel entry instead, 32 registers held across the K-loop, measured no gain: 72.4-72.7 vs
// 72.4-72.7 frames/s in an interleaved A/B -- the read is short once it is whole lines; not kept.)
template <int EPI, bool I8, int NTL, int MT>
__device__ __forceinline__ void gemm_epilogue_lds(typename Ty<I8>::acc (&acc)[NTL][MT], bf16* __restrict__ Y, int M, int N,
                                                  int ldo, int mw, int nw, int lane, char* __restrict__ ep, const EpiArgs& ea) {
  constexpr bool RES = (EPI == LL_EPI_BIAS_GATE_RES || EPI == LL_EPI_BIAS_RES);
  constexpr bool GATE = (EPI == LL_EPI_BIAS_GATE_RES);
  constexpr int RS = EPI_ROW_BYTES(NTL), CP = NTL * 2, R = MT * 16, NIT = R * CP / 64;
  static_assert(R * CP % 64 == 0, "the wave tile must be a whole number of 64-lane row passes");
  const int fr = lane & 15, fg = lane >> 4;
  // ---- pass 1 (MFMA layout): v = bf16(acc [* scales] + bias) [GELU] -> LDS
#pragma unroll
  for (int a = 0; a < NTL; ++a) {
    int n = nw + a * 16 + fg * 4;
    int ncl = n < N ? n : N - 4;
    bf16x4 bv = *reinterpret_cast<const bf16x4*>(ea.bias + ncl);
    f32x4 swv = {1.f, 1.f, 1.f, 1.f};
    if (I8) swv = *reinterpret_cast<const f32x4*>(ea.sw + ncl);
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      float sxm = 1.0f;
      if (I8) {
        int m = mw + b * 16 + fr;
        sxm = ea.sx[m < M ? m : M - 1];
      }
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = I8 ? rbf((float)acc[a][b][j] * (sxm * swv[j]) + (float)bv[j]) : rbf((float)acc[a][b][j] + (float)bv[j]);
        o[j] = (bf16)(EPI == LL_EPI_BIAS_GELU ? gelu_tanh(v) : v);
      }
      *reinterpret_cast<bf16x4*>(ep + (b * 16 + fr) * RS + (a * 16 + fg * 4) * 2) = o;
    }
  }
  // ---- pass 2 (row layout): chunk q = it*64 + lane -> (row q / CP, 16-byte chunk q % CP); loads of PC iterations up front
  constexpr int PC = GATE ? 4 : 8;
#pragma unroll
  for (int it0 = 0; it0 < NIT; it0 += PC) {
    int row[PC], col[PC];
    bool ok[PC];
    bf16x8 rv[RES ? PC : 1], ge[GATE ? PC : 1], gm[GATE ? PC : 1];
#pragma unroll
    for (int i = 0; i < PC; ++i) {
      if (it0 + i >= NIT) continue;
      int q = (it0 + i) * 64 + lane;
      int rr = q / CP, ch = q - rr * CP;
      row[i] = rr;
      col[i] = ch * 8;
      int m = mw + rr, n = nw + ch * 8;
      ok[i] = m < M && n < N;
      int mc = m < M ? m : M - 1, ncl = n < N ? n : N - 8;
      if (RES) rv[i] = *reinterpret_cast<const bf16x8*>(ea.res + (size_t)mc * ldo + ncl);
      if (GATE) {
        int bb = mc / ea.rows_per_batch, f = (mc % ea.rows_per_batch) / ea.frame_len;
        ge[i] = *reinterpret_cast<const bf16x8*>(ea.e + ((size_t)(bb * ea.F + f) * ea.nmod + ea.gate_idx) * N + ncl);
        if (ea.mod) gm[i] = *reinterpret_cast<const bf16x8*>(ea.mod + (size_t)ea.gate_idx * N + ncl);
      }
    }
#pragma unroll
    for (int i = 0; i < PC; ++i) {
      if (it0 + i >= NIT) continue;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(ep + row[i] * RS + col[i] * 2);
      bf16x8 o;
      if (GATE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float g = ea.mod ? rbf((float)gm[i][j] + (float)ge[i][j]) : (float)ge[i][j];   // mod == NULL: e is bf16(mod + e)
          o[j] = (bf16)((float)rv[i][j] + rbf((float)v[j] * g));
        }
      } else if (RES) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)rv[i][j] + (float)v[j]);
      } else {
        o = v;
      }
      if (ok[i]) {
        bf16* dst = epi_dest(ea, Y, ldo, mw + row[i], nw + col[i]);
        if (dst) *reinterpret_cast<bf16x8*>(dst) = o;
      }
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

