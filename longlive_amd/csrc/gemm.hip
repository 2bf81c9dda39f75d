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

// STAG: waves 4-7 (the SIMD partners of waves 0-3) run HALF A K-STEP behind: they defer the 16 MFMAs of a tile's second k-half
// to the start of the next iteration, their fragments waiting in registers across the barrier.  At every barrier release one
// wave of each SIMD then has matrix work ready while its partner starts with LDS reads, instead of both waiting for reads and
// then contending for the pipe (MI355X_MICROARCH.md, "two waves that run the same program with one barrier per block").
template <int EPI, bool I8, bool STAG>
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
  const bool late = STAG && wave >= 4;      // wave-uniform
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
  frag_t wd[4], xd[4];                      // STAG: the second k-half's fragments (consumed one barrier later by the late group)
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt landed; tile kt+1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the deferred fragments have left the slot restaged below
    __builtin_amdgcn_s_barrier();      // every wave's share of tile kt is in LDS; slot (kt+2)%3 is no longer being read
    const bool do_stage = kt + 2 < nk && !(lds_epi & 0x100);          // (0x100: timing experiment, results invalid)
    int s2 = slot + 2;
    s2 = s2 >= 3 ? s2 - 3 : s2;
    if (do_stage && !late) stage(kt + 2, s2);          // STAG: the late group issues its share mid-iteration (below), so that
                                                        // the two waves of a SIMD are never both busy issuing LDS-DMA pieces
    const char* xs = smem + slot * V2_STAGE;
    const char* ws = xs + V2_BM * ROWB;
    if (!live || (lds_epi & 0x200)) {   // rows past M (last m-tile): stage and sync only -- the chip runs at its power cap,
      if (do_stage && late) stage(kt + 2, s2);                  // idle matrix pipes are speed elsewhere
      slot = slot == 2 ? 0 : slot + 1;
      continue;
    }
    __builtin_amdgcn_s_setprio(1);
    if (STAG) {
      if (late && kt > 0) { V2_MMA(wd, xd); }
      frag_t wf[4], xf[4];
      V2_READ(wf, xf, 0);
      V2_MMA(wf, xf);
      if (late) {
        __builtin_amdgcn_s_setprio(0);
        if (do_stage) stage(kt + 2, s2);
        __builtin_amdgcn_s_setprio(1);
      }
      V2_READ(wd, xd, 1);
      if (!late) { V2_MMA(wd, xd); }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        frag_t wf[4], xf[4];
        V2_READ(wf, xf, ks);
        V2_MMA(wf, xf);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    slot = slot == 2 ? 0 : slot + 1;
  }
  if (STAG && late && live && nk > 0) { V2_MMA(wd, xd); }
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

// ---------------------------------------------------------------------------------------------------------------
// ws ("wave-specialised"): the v2 / v5 tiles with the LDS-DMA staging moved OFF the compute waves.
//
// Why: timing the K-loop with parts switched off (profiles/r02_ab_gemm_epilogue.txt) shows staging alone at 0.42 (256x128) /
// 0.81 us (256x224) per K-step and LDS reads + MFMAs alone at 0.69 / 1.05 us, but the two together at 0.91 / 1.46 us: they do
// not overlap.  The CU's LDS-DMA path moves ~35-50 bytes per clock; the 48-60 one-KiB pieces of a K-step fill its queue, every
// wave that issues one stalls IN ORDER behind it, and after a barrier all eight waves issue theirs at once -- the matrix pipe
// idles for the length of the issue burst.  Here waves 0-7 only read fragments and issue MFMAs; NL extra waves (one or two,
// co-resident on SIMD 0 / 1 as a third wave) issue every piece and absorb the back-pressure.  One workgroup barrier per
// K-step as before: the loader passes it once its pieces of tile kt have landed (counted vmcnt), the compute waves once they
// are done with tile kt-1, and the loader then restages the slot tile kt-1 occupied.
//   STAGES = 3 for the 256x128 tile (144 KiB), 2 for 256x192 / 256x224 (the ring would not fit three).
template <int EPI, bool I8, int WM, int WN, int MT, int NT, int STAGES, int NL>
__global__ __launch_bounds__(512 + 64 * NL, 1) void gemm_kernel_ws(const char* __restrict__ X, const char* __restrict__ Wt,
                                                                     bf16* __restrict__ Y, int M, int N, int nk, size_t xrow_bytes,
                                                                     size_t wrow_bytes, int ldo, int ntm, int ntn, int gm,
                                                                     int lds_epi, EpiArgs ea) {
  static_assert(WM * WN == 8 && WM * MT * 16 == 256, "8 compute waves, 256 rows");
  static_assert(NL == 1 || NL == 2, "one or two loader waves");
  constexpr int BNv = WN * NT * 16, STAGE = (256 + BNv) * ROWB, NB = BNv / 8;   // NB = W pieces (8 rows x 128 B) per K-step
  constexpr int PX = 32 / NL, PW = (NB + NL - 1) / NL;                          // pieces per loader wave
  static_assert((PX + PW) * (STAGES - 2) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Ty<I8>::frag frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid = xcd_remap(blockIdx.x, ntm * ntn), mt_, nt_;
  tile_of(lid, ntm, ntn, gm, mt_, nt_);
  const int m0 = mt_ * 256, n0 = nt_ * BNv;

  if (wave >= 8) {
    // ---- loader wave(s): all LDS-DMA of the workgroup -------------------------------------------------------------------
    const int li = wave - 8;
    auto stage = [&](int kt, int slot) {
      char* base = smem + slot * STAGE;
      stage_rows(X, xrow_bytes, m0, M, kt * ROWB, base, li * PX, PX, lane);
      int w0 = li * PW, wn_ = NB - w0 < PW ? NB - w0 : PW;
#pragma unroll
      for (int i = 0; i < PW; ++i)
        if (i < wn_) stage_rows(Wt, wrow_bytes, n0, N, kt * ROWB, base + 256 * ROWB, w0 + i, 1, lane);
    };
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
      if (t < nk) stage(t, t);
    int slot_new = STAGES - 1;                       // slot the next staged tile goes to
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed when at most the younger tiles' pieces are outstanding (STAGES - 2 tiles of PX + PW pieces; the
      // loader of a 2-stage ring has nothing younger in flight)
      if (STAGES == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PX + PW) * (STAGES - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // tile kt visible to the compute waves; they are done with tile kt - 1
      if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, slot_new);
      slot_new = slot_new == STAGES - 1 ? 0 : slot_new + 1;
    }
    if (lds_epi) __builtin_amdgcn_s_barrier();       // the compute waves' "ring is free" barrier before the staged epilogue
    return;
  }

  // ---- compute waves ------------------------------------------------------------------------------------------------------
  const int wm = wave / WN, wn = wave % WN;
  typename Ty<I8>::acc acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = acc_zero<I8>();
  const int fr = lane & 15, fg = lane >> 4;
  const bool live = m0 + wm * MT * 16 < M;      // wave-uniform
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();
    const char* xs = smem + slot * STAGE;
    const char* ws = xs + 256 * ROWB;
    slot = slot == STAGES - 1 ? 0 : slot + 1;
    if (!live) continue;               // rows past M: sync only
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
    __builtin_amdgcn_s_barrier();      // every wave has read its last K-step's fragments: the ring is free
    gemm_epilogue_lds<EPI, I8, NT, MT>(acc, Y, M, N, ldo, m0 + wm * MT * 16, n0 + wn * NT * 16, lane,
                                       smem + wave * (MT * 16 * EPI_ROW_BYTES(NT)), ea);
  } else {
    gemm_epilogue<EPI, I8, NT, MT>(acc, Y, M, N, ldo, m0 + wm * MT * 16, n0 + wn * NT * 16, fr, fg, ea);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v4: the 256 x 256 tile as a PING-PONG of the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; waves w
// and w + 4 share a SIMD).  A K-step is cut into four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each:
//     [ds_read the quadrant's A (8 x b128) and/or B (4 x b128) fragments; 2 LDS-DMA pieces] s_barrier [16 MFMA] s_barrier
// and the second group runs one barrier behind the first, so on every SIMD one wave is on the matrix pipe while its partner
// fills registers from LDS (in v2/v3 both waves of a SIMD wait for LDS and then compete for the pipe at the same time).
// Staging is spread evenly -- an LDS-DMA piece costs ~100 cycles of issue, 8 of them in one phase would starve the
// partner's MFMA block -- by cutting a K-step's operands into four "half-tiles" along the quadrants:
//     X0 / X1 = the A rows of every wave's first / second 64-row half,  Y0 / Y1 = the B rows of its first / second 32 columns
// read in phases a | c (X0 | X1) and a,d | b (Y0 | Y1).  Each slot is restaged two phases after its last read and every
// phase issues exactly one half-tile (2 pieces per wave):   a: X1(kt+1)   b: Y0(kt+1)   c: X0(kt+2)   d: Y1(kt+2)
// so one counted wait per K-step -- vmcnt(4) in phase d, leaving only X0(kt+2), Y1(kt+2) in flight -- retires everything
// K-step kt+1 reads, one barrier before its first read for either group.  Past the last K-step the source is clamped (the
// piece count per phase must not change or the counted wait would retire the wrong loads); those pieces are never read.
#define V4_LOAD_A(MH)                                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
    int rx = wm * 128 + (MH) * 64 + t * 16 + fr;                                                 \
    xf[t][ks] = *reinterpret_cast<const frag_t*>(xs + rx * ROWB + (((ks * 4 + fg) ^ (rx & 7)) << 4)); \
  }
#define V4_LOAD_B(NH)                                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                \
    int rw = wn * 64 + (NH) * 32 + t * 16 + fr;                                                  \
    wf[t][ks] = *reinterpret_cast<const frag_t*>(ws + rw * ROWB + (((ks * 4 + fg) ^ (rw & 7)) << 4)); \
  }
#define V4_MMA(MH, NH)                                                                           \
  __builtin_amdgcn_s_barrier();                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_setprio(1);                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
  _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                  \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                  \
    acc[(NH) * 2 + a][(MH) * 4 + b] = Ty<I8>::mma(wf[a][ks], xf[b][ks], acc[(NH) * 2 + a][(MH) * 4 + b]); \
  __builtin_amdgcn_s_setprio(0);                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_barrier();

template <int EPI, bool I8>
__global__ __launch_bounds__(512, 1) void gemm_kernel_v4(const char* __restrict__ X, const char* __restrict__ Wt,
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

  // half-tile staging: which 2 of the operand's 32 eight-row pieces this wave copies
  const int xi = (wave < 4 ? 2 * wave : 16 + 2 * (wave - 4));        // X0: rows 0-63 | 128-191 (X1: + 8 pieces)
  const int yi = (wave >> 1) * 8 + (wave & 1) * 2;                   // Y0: rows 0-31 | 64-95 | 128-159 | 192-223 (Y1: + 4)
  auto stage_x = [&](int kt, int half) {
    int kc = kt < nk ? kt : nk - 1;
    stage_rows(X, xrow_bytes, m0, M, kc * ROWB, smem + (kt & 1) * V3_STAGE, xi + half * 8, 2, lane);
  };
  auto stage_y = [&](int kt, int half) {
    int kc = kt < nk ? kt : nk - 1;
    stage_rows(Wt, wrow_bytes, n0, N, kc * ROWB, smem + (kt & 1) * V3_STAGE + V3_BM * ROWB, yi + half * 4, 2, lane);
  };
  stage_x(0, 0); stage_y(0, 1); stage_x(0, 1); stage_y(0, 0);       // K-step 0
  stage_x(1, 0); stage_y(1, 1);                                     // what phases c, d of K-step -1 would have issued
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();      // the second group runs one barrier behind

  const int fr = lane & 15, fg = lane >> 4;
  frag_t xf[4][2], wf[2][2];
  for (int kt = 0; kt < nk; ++kt) {
    const char* xs = smem + (kt & 1) * V3_STAGE;
    const char* ws = xs + V3_BM * ROWB;
    // phase a: rows 0-63 x cols 0-31 of the wave tile
    V4_LOAD_B(0);
    V4_LOAD_A(0);
    stage_x(kt + 1, 1);
    V4_MMA(0, 0);
    // phase b: rows 0-63 x cols 32-63
    V4_LOAD_B(1);
    stage_y(kt + 1, 0);
    V4_MMA(0, 1);
    // phase c: rows 64-127 x cols 32-63
    V4_LOAD_A(1);
    stage_x(kt + 2, 0);
    V4_MMA(1, 1);
    // phase d: rows 64-127 x cols 0-31
    V4_LOAD_B(0);
    stage_y(kt + 2, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    V4_MMA(1, 0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  gemm_epilogue<EPI, I8, 4, 8>(acc, Y, M, N, ldo, m0 + wm * 128, n0 + wn * 64, fr, fg, ea);
}

// ---------------------------------------------------------------------------------------------------------------
// v4sk: the 256 x 256 ping-pong tile with K cut in two (split-K 2) for N = 1536-wide GEMMs with a long K (FFN2:
// 4680 x 1536 x 8960).  256 x 128 tiles are forced there by the tile count (19 x 12 = 228 workgroups = one round); they stage
// (256 + 128) x 128 B per K-step against 1024 MFMA cycles, and the LDS-DMA path is what bounds the K-loop.  The
// two K-halves of a 256 x 256 tile are 2 x 114 = 228 workgroups again, each staging 512 x 128 B per 2048 MFMA cycles: 33 % fewer
// staged bytes per FLOP.  The halves meet in the epilogue: workgroup `split` keeps the rows of its waves' MH = split half
// (64 of every 128), hands the fp32 accumulators of the other half to its partner through a workspace in MFMA register layout
// (128 KiB per workgroup, whole 128-byte lines per store instruction), waits for the partner's half, adds, and runs the usual
// fused epilogue on the half it owns -- each workgroup finishes half a tile, nobody idles.
//   Hand-off (MI355X_MICROARCH.md, hand-offs without an agent release): every partial store and load is `sc1`; each storing wave
//   runs s_waitcnt vmcnt(0), then a workgroup barrier, then ONE lane's agent-scope atomic add on the workgroup's flag; the
//   consumer's one lane polls the partner's flag with an sc1 load, resets it for the next launch, and a workgroup barrier stands
//   between that poll and every load of the bytes.  Both workgroups of a pair must be resident (each waits for the other):
//   the launcher only takes this path when the whole grid fits the device in one round (one workgroup per CU: 128 KiB LDS).
//   Pairs are placed on ONE XCD (block b -> XCD b % 8: pair = (b / 8) / 2), so the hand-off stays in that XCD's L2 / its memory
//   channel neighbourhood; correctness does not depend on that placement.
//   L2 = true (taken only when a probe launch has shown that block b's XCC_ID is a function of b % 8 on this device, i.e. that
//   partners DO share an XCD): the partial tiles use plain stores and loads -- the lines stay in the pair's L2 instead of making
//   a round trip through memory (29 MB written and read back at the very end of the launch cost ~15 us); the flag protocol is
//   unchanged.  The reader has never touched those addresses in this launch, so its L1 cannot hold them.
// fp32 addition commutes: the result does not depend on which half arrives first (deterministic), but it differs in the last
// bits from the unsplit kernels' single accumulation chain.
#define V4SK_PART_FLOATS (8 * 16 * 64 * 4)      // per (tile, split): [wave][a][b'][lane][4]
#define V4SK_FLAG_BYTES 4096
#define V4SK_ERR_WORD (V4SK_FLAG_BYTES / 4 - 1)   // last word of the flag page: epoch of a launch whose hand-off timed out (0 = none)
#define V4SK_POLL_TICKS 5000000ull             // 50 ms of s_memrealtime (100 MHz); a hand-off takes ~2 us

template <bool L2, typename V>
__device__ __forceinline__ void sk_store(float* p, V v) {      // f32x4 partial sums, or i32x4 (W8A8: exact integers)
  if (L2) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int EPI, bool I8, int SPLIT, bool L2>
__device__ __forceinline__ void v4sk_finish(typename Ty<I8>::acc (&acc)[4][8], float* __restrict__ part, unsigned* __restrict__ flags, int tile,
                                            bf16* __restrict__ Y, int M, int N, int ldo, int mw, int nw, int wave, int lane,
                                            const EpiArgs& ea, unsigned epoch) {
  constexpr int GIVE = SPLIT ^ 1;
  float* mine = part + ((size_t)tile * 2 + SPLIT) * V4SK_PART_FLOATS + ((size_t)wave * 16 * 64 + lane) * 4;
  float* theirs = part + ((size_t)tile * 2 + GIVE) * V4SK_PART_FLOATS + ((size_t)wave * 16 * 64 + lane) * 4;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) sk_store<L2>(mine + (a * 4 + b) * 256, acc[a][GIVE * 4 + b]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // The flag carries THIS launch's epoch (a process-wide launch counter, never 0): a word left behind by an earlier or an
    // aborted launch on the same workspace can never match, and nothing has to be reset.  The poll is bounded by the 100 MHz
    // real-time counter: a partner that never arrives (it faulted, its process was killed, the grid was not co-resident after
    // all) costs V4SK_POLL_TICKS, sets the workspace's error word and lets the kernel drain with a wrong tile instead of
    // hanging the wave -- ll_gemm_splitk_status() reports it at the caller's next synchronisation point.
    __hip_atomic_exchange(flags + tile * 2 + SPLIT, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned* pf = flags + tile * 2 + GIVE;
    unsigned v;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(pf) : "memory");
      if (v == epoch) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > V4SK_POLL_TICKS) {
        __hip_atomic_store(flags + V4SK_ERR_WORD, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  typedef typename Ty<I8>::acc acc_t;
  // all 16 loads of the partner's half in flight, ONE wait (four dependent round trips cost ~4 us of every launch)
  acc_t r[4][4];
  const float* q0 = theirs;
  const float* q1 = theirs + 4 * 256;
  const float* q2 = theirs + 8 * 256;
  const float* q3 = theirs + 12 * 256;
  if (L2) {
    asm volatile(
        "global_load_dwordx4 %0, %16, off\n\t"
        "global_load_dwordx4 %1, %16, off offset:1024\n\t"
        "global_load_dwordx4 %2, %16, off offset:2048\n\t"
        "global_load_dwordx4 %3, %16, off offset:3072\n\t"
        "global_load_dwordx4 %4, %17, off\n\t"
        "global_load_dwordx4 %5, %17, off offset:1024\n\t"
        "global_load_dwordx4 %6, %17, off offset:2048\n\t"
        "global_load_dwordx4 %7, %17, off offset:3072\n\t"
        "global_load_dwordx4 %8, %18, off\n\t"
        "global_load_dwordx4 %9, %18, off offset:1024\n\t"
        "global_load_dwordx4 %10, %18, off offset:2048\n\t"
        "global_load_dwordx4 %11, %18, off offset:3072\n\t"
        "global_load_dwordx4 %12, %19, off\n\t"
        "global_load_dwordx4 %13, %19, off offset:1024\n\t"
        "global_load_dwordx4 %14, %19, off offset:2048\n\t"
        "global_load_dwordx4 %15, %19, off offset:3072\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[0][2]), "=&v"(r[0][3]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[1][2]),
          "=&v"(r[1][3]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[2][2]), "=&v"(r[2][3]), "=&v"(r[3][0]), "=&v"(r[3][1]),
          "=&v"(r[3][2]), "=&v"(r[3][3])
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3)
        : "memory");
  } else {
    asm volatile(
        "global_load_dwordx4 %0, %16, off sc1\n\t"
        "global_load_dwordx4 %1, %16, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %16, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %16, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %4, %17, off sc1\n\t"
        "global_load_dwordx4 %5, %17, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %6, %17, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %7, %17, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %8, %18, off sc1\n\t"
        "global_load_dwordx4 %9, %18, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %10, %18, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %11, %18, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %12, %19, off sc1\n\t"
        "global_load_dwordx4 %13, %19, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %14, %19, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %15, %19, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[0][2]), "=&v"(r[0][3]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[1][2]),
          "=&v"(r[1][3]), "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[2][2]), "=&v"(r[2][3]), "=&v"(r[3][0]), "=&v"(r[3][1]),
          "=&v"(r[3][2]), "=&v"(r[3][3])
        : "v"(q0), "v"(q1), "v"(q2), "v"(q3)
        : "memory");
  }
  acc_t hlo[4][2], hhi[4][2];     // two 32-row halves: the gate-residual epilogue of a 64-row batch does not fit the register file
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    hlo[a][0] = acc[a][SPLIT * 4 + 0] + r[a][0];
    hlo[a][1] = acc[a][SPLIT * 4 + 1] + r[a][1];
    hhi[a][0] = acc[a][SPLIT * 4 + 2] + r[a][2];
    hhi[a][1] = acc[a][SPLIT * 4 + 3] + r[a][3];
  }
  gemm_epilogue<EPI, I8, 4, 2>(hlo, Y, M, N, ldo, mw + SPLIT * 64, nw, lane & 15, lane >> 4, ea);
  gemm_epilogue<EPI, I8, 4, 2>(hhi, Y, M, N, ldo, mw + SPLIT * 64 + 32, nw, lane & 15, lane >> 4, ea);
}

template <int EPI, bool I8, bool L2>
__global__ __launch_bounds__(512, 1) void gemm_kernel_v4sk(const char* __restrict__ X, const char* __restrict__ Wt,
                                                           bf16* __restrict__ Y, int M, int N, int nkh, size_t xrow_bytes,
                                                           size_t wrow_bytes, int ldo, int ntiles, int ntn,
                                                           float* __restrict__ part, unsigned* __restrict__ flags, EpiArgs ea, unsigned epoch,
                                                           int fault) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Ty<I8>::frag frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // block b runs on XCD b % 8: both halves of a tile on one XCD, every XCD a contiguous range of tiles
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int pair = idx >> 1, split = idx & 1;
  const int q_ = ntiles >> 3, r_ = ntiles & 7;
  const int cnt = q_ + (xcd < r_ ? 1 : 0), start = xcd * q_ + (xcd < r_ ? xcd : r_);
  if (pair >= cnt) return;                                  // grid padding (whole workgroup, before any barrier)
  if (fault && split == 1) return;                          // test hook (tuning key gemm_splitk_fault): a partner that never arrives
  const int tile = start + pair;
  const int mt_ = tile / ntn, nt_ = tile - mt_ * ntn;
  const int m0 = mt_ * V3_BM, n0 = nt_ * V3_BN;
  const int k0 = split * nkh;

  typename Ty<I8>::acc acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = acc_zero<I8>();

  const int xi = (wave < 4 ? 2 * wave : 16 + 2 * (wave - 4));
  const int yi = (wave >> 1) * 8 + (wave & 1) * 2;
  auto stage_x = [&](int kt, int half) {
    int kc = (kt < nkh ? kt : nkh - 1) + k0;
    stage_rows(X, xrow_bytes, m0, M, kc * ROWB, smem + (kt & 1) * V3_STAGE, xi + half * 8, 2, lane);
  };
  auto stage_y = [&](int kt, int half) {
    int kc = (kt < nkh ? kt : nkh - 1) + k0;
    stage_rows(Wt, wrow_bytes, n0, N, kc * ROWB, smem + (kt & 1) * V3_STAGE + V3_BM * ROWB, yi + half * 4, 2, lane);
  };
  stage_x(0, 0); stage_y(0, 1); stage_x(0, 1); stage_y(0, 0);
  stage_x(1, 0); stage_y(1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();

  const int fr = lane & 15, fg = lane >> 4;
  frag_t xf[4][2], wf[2][2];
  for (int kt = 0; kt < nkh; ++kt) {
    const char* xs = smem + (kt & 1) * V3_STAGE;
    const char* ws = xs + V3_BM * ROWB;
    V4_LOAD_B(0);
    V4_LOAD_A(0);
    stage_x(kt + 1, 1);
    V4_MMA(0, 0);
    V4_LOAD_B(1);
    stage_y(kt + 1, 0);
    V4_MMA(0, 1);
    V4_LOAD_A(1);
    stage_x(kt + 2, 0);
    V4_MMA(1, 1);
    V4_LOAD_B(0);
    stage_y(kt + 2, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    V4_MMA(1, 0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (split == 0) v4sk_finish<EPI, I8, 0, L2>(acc, part, flags, tile, Y, M, N, ldo, m0 + wm * 128, n0 + wn * 64, wave, lane, ea, epoch);
  else v4sk_finish<EPI, I8, 1, L2>(acc, part, flags, tile, Y, M, N, ldo, m0 + wm * 128, n0 + wn * 64, wave, lane, ea, epoch);
}
#undef V4_LOAD_A
#undef V4_LOAD_B
#undef V4_MMA

// runtime tuning switches (A/B experiments from tools/kbench; defaults are the shipped configuration)
static int g_gemm_variant = 0;
static int g_gemm_variant_wide = 0;  // like gemm_variant, but only for N >= 4096 (QKV, FFN1): A/B of the wide tilings alone
// 1: epilogue staged through LDS (whole-line residual loads / stores) in the v2 / v5 tilings, except the GELU epilogue, whose
// register form measured 1.6 % faster (FFN1 127.2 vs 129.3 us; everything else 1.5-11 % faster staged); 2: all; 0: none
static int g_gemm_lds_epi = 1;
static int g_gemm_ws_mask = 15;     // which launches gemm_ws applies to: 1 = 256x128 with K >= 4096 (FFN2), 2 = 256x128 otherwise, 4 = 256x192, 8 = 256x224
static int g_gemm_ws = 0;            // wave-specialised staging (gemm_kernel_ws): 0 off, 1 / 2 loader waves, for variants 2, 5, 6
static int g_gemm_stagger = 0;       // 1: 256x128 tiling with waves 4-7 half a K-step behind (gemm_kernel_v2<.., true>)
static int g_gemm_asm = 3;         // generated kernels where they cover the call, also in place of split-K (see launch_gemm)
static int g_gemm_group_m = 4;     // m-tiles per group in the workgroup -> tile walk (tile_of); <= 1: N fastest (round 1's order)
void ll_set_attn_variant_internal(int v);
void ll_set_attn_xcd_internal(int v);
void ll_set_attn_sk_internal(int v);
void ll_set_attn_pp_min_internal(int v);
void ll_set_attn_asm_min_internal(int v);
void ll_set_attn_mfma16_internal(int v);
void ll_set_attn_asm_internal(int v);
void ll_set_conv_halo_internal(int v);
void ll_set_splitk_l2_internal(int v);
void ll_set_splitk_fault_internal(int v);
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
  if (!strcmp(key, "gemm_stagger")) { g_gemm_stagger = value; return LL_OK; }
  if (!strcmp(key, "gemm_ws")) { g_gemm_ws = value; return LL_OK; }
  if (!strcmp(key, "gemm_ws_mask")) { g_gemm_ws_mask = value; return LL_OK; }
  if (!strcmp(key, "attn_variant")) { ll_set_attn_variant_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_xcd")) { ll_set_attn_xcd_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_sk_wgs")) { ll_set_attn_sk_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_pp_min_keys")) { ll_set_attn_pp_min_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_asm_min_keys")) { ll_set_attn_asm_min_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_mfma16")) { ll_set_attn_mfma16_internal(value); return LL_OK; }
  if (!strcmp(key, "attn_asm")) { ll_set_attn_asm_internal(value); return LL_OK; }
  if (!strcmp(key, "conv_halo")) { ll_set_conv_halo_internal(value); return LL_OK; }
  if (!strcmp(key, "gemm_splitk_l2")) { ll_set_splitk_l2_internal(value); return LL_OK; }
  if (!strcmp(key, "gemm_splitk_fault")) { ll_set_splitk_fault_internal(value); return LL_OK; }
  if (!strcmp(key, "gemm_asm")) { g_gemm_asm = value; return LL_OK; }
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

// variant 2 = 256x128 / 3-stage ring, 3 = 256x256 (128x64 per wave), 4 = 256x256 ping-pong, 5 = 256x192, 6 = 256x224;
// 0 = auto: the shape with the smallest   rounds(on 256 CUs) x columns x per-flop cost   (v2's 64x64 wave tile costs ~15 %
// more per flop than the 128-row ones; a tile count below the CU count is one round of whatever fills most CUs).
static int pick_gemm_variant(int M, int N) {
  int variant = g_gemm_variant;
  if (N >= 4096 && g_gemm_variant_wide >= 2 && g_gemm_variant_wide <= 6) variant = g_gemm_variant_wide;
  if (variant < 2 || variant > 6) {
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
  const int bn = v == 3 || v == 4 ? 256 : v == 5 ? 192 : v == 6 ? 224 : 128;
  char walk[48];
  if (g_gemm_group_m > 1) snprintf(walk, sizeof walk, ", groups of %d m-tiles", g_gemm_group_m);
  else snprintf(walk, sizeof walk, ", N fastest");
  const char* name = v == 2 ? "gemm_kernel_v2" : v == 3 ? "gemm_kernel_v3" : v == 4 ? "gemm_kernel_v4" : "gemm_kernel_v5";
  int ntm = (M + 255) / 256, ntn = (N + bn - 1) / bn;
  snprintf(out, (size_t)cap, "%s<%s> tile 256x%d, %d workgroups%s", name, int8 ? "i8" : "bf16", bn, ntm * ntn, walk);
  (void)K;
  return LL_OK;
}

// ll_gemm_plan for a call whose epilogue is known: names the generated kernel where ll_gemm_bf16 (splitk_call = 0) or
// ll_gemm_bf16_splitk (splitk_call = 1, workspace given) would take it under the current tuning; `plain` = 1 when the call has no
// V-cache output and no per-batch modulation vector (the block linears of the pipeline except QKV).
extern "C" int ll_gemm_plan_epi(int M, int N, int K, int int8, int epilogue, int plain, int splitk_call, char* out, int cap);

// ===============================================================================================================
// tuning key gemm_asm (declared near ll_set_tuning): bit 0 = bf16 block linears on the generated one-wave-per-SIMD kernels (gemm_asm.hip) where a
                                // tile width fits (FFN1: 256 x 224 + GELU; N <= 2048: 256 x 128 with bias / gate-residual / residual);
                                // bit 1 = ll_gemm_bf16_splitk calls (FFN2) take them too instead of the split-K kernel
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
  // gemm_asm: bit 0 = generated kernels for the shapes they cover, bit 1 = also instead of split-K; bit 2 / bit 3 leave the
  // GELU (256 x 224) / the 128-wide kernels out (A/B of their share in the pipeline's power budget)
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
  const bool v4 = (variant == 4);
  const bool v3 = (variant == 3) || v4;      // same tile and LDS footprint
  const bool v5 = (variant == 5), v6 = (variant == 6);
  int bm = (v3 || v5 || v6) ? 256 : V2_BM, bn = v3 ? V3_BN : v5 ? 192 : v6 ? 224 : BN;
  int ntm = (M + bm - 1) / bm, ntn = (N + bn - 1) / bn;
  dim3 grid(ntm * ntn), block(512);
  const int gm = g_gemm_group_m;
  const int lds_epi = ((g_gemm_lds_epi & 3) == 2 || ((g_gemm_lds_epi & 3) == 1 && epilogue != LL_EPI_BIAS_GELU) ? 1 : 0) |
                      (g_gemm_lds_epi & 0x300);   // 0x100 / 0x200: timing experiments (no in-loop staging / no compute), results invalid
  size_t lds = v3 ? 2 * V3_STAGE : (v5 || v6) ? 2 * (size_t)(256 + bn) * ROWB : 3 * V2_STAGE;
#define LAUNCH_WS(E, WM_, WN_, MT_, NT_, ST_, NL_)                                                                     \
  do {                                                                                                                 \
    {                                                                                           \
      (void)ll_lds_attr((const void*)gemm_kernel_ws<E, I8, WM_, WN_, MT_, NT_, ST_, NL_>, \
                                (int)lds);                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_kernel_ws<E, I8, WM_, WN_, MT_, NT_, ST_, NL_>), grid, dim3(512 + 64 * NL_), lds, s,      \
                       (const char*)x, (const char*)w, out, M, N, nk, xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea); \
  } while (0)
#define LAUNCH(E)                                                                                                      \
  do {                                                                                                                 \
    const int wsb = v5 ? 4 : v6 ? 8 : (variant == 2 ? (nk >= 64 ? 1 : 2) : 0);                                         \
    const int wsn = (g_gemm_ws_mask & wsb) ? g_gemm_ws : 0;                                                            \
    if (wsn == 1 && v5) { LAUNCH_WS(E, 2, 4, 8, 3, 2, 1); break; }                                                     \
    if (wsn == 2 && v5) { LAUNCH_WS(E, 2, 4, 8, 3, 2, 2); break; }                                                     \
    if (wsn == 1 && v6) { LAUNCH_WS(E, 4, 2, 4, 7, 2, 1); break; }                                                     \
    if (wsn == 2 && v6) { LAUNCH_WS(E, 4, 2, 4, 7, 2, 2); break; }                                                     \
    if (wsn == 1 && variant == 2) { LAUNCH_WS(E, 4, 2, 4, 4, 3, 1); break; }                                           \
    if (wsn == 2 && variant == 2) { LAUNCH_WS(E, 4, 2, 4, 4, 3, 2); break; }                                           \
    if (v5 || v6) {                                                                                                    \
      {                                                                                          \
        (void)ll_lds_attr((const void*)gemm_kernel_v5<E, I8, 2, 4, 8, 3>, 2 * (256 + 192) * ROWB); \
        (void)ll_lds_attr((const void*)gemm_kernel_v5<E, I8, 4, 2, 4, 7>, 2 * (256 + 224) * ROWB); \
      }                                                                                                                \
      if (v5)                                                                                                          \
        hipLaunchKernelGGL((gemm_kernel_v5<E, I8, 2, 4, 8, 3>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                           xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                                 \
      else                                                                                                             \
        hipLaunchKernelGGL((gemm_kernel_v5<E, I8, 4, 2, 4, 7>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                           xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                                 \
    } else if (v4) {                                                                                                   \
      {                                                                                          \
        (void)ll_lds_attr((const void*)gemm_kernel_v4<E, I8>, (int)lds); \
      }                                                                                                                \
      hipLaunchKernelGGL((gemm_kernel_v4<E, I8>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk,   \
                         xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                                   \
    } else if (v3) {                                                                                                   \
      {                                                                                          \
        (void)ll_lds_attr((const void*)gemm_kernel_v3<E, I8>, (int)lds); \
      }                                                                                                                \
      hipLaunchKernelGGL((gemm_kernel_v3<E, I8>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk,   \
                         xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                                   \
    } else {                                                                                                           \
      {                                                                                          \
        (void)ll_lds_attr((const void*)gemm_kernel_v2<E, I8, false>, (int)lds); \
        (void)ll_lds_attr((const void*)gemm_kernel_v2<E, I8, true>, (int)lds); \
      }                                                                                                                \
      if (g_gemm_stagger)                                                                                              \
        hipLaunchKernelGGL((gemm_kernel_v2<E, I8, true>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                           xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                    \
      else                                                                                                             \
        hipLaunchKernelGGL((gemm_kernel_v2<E, I8, false>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nk, \
                           xrow_bytes, wrow_bytes, ldo, ntm, ntn, gm, lds_epi, ea);                                    \
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
static int splitk_tiles(int M, int N) { return ((M + 255) / 256) * (N / 256); }
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
__global__ void xcc_probe_kernel(unsigned* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 15u;
}
static int g_gemm_splitk_l2 = 0;     // tuning key gemm_splitk_l2: 1 = exchange through the pair's L2 (plain stores / loads) when the
                                     // placement probe allows it; measured 146.8 vs 150.9 us alone, +0.1 % in the pipeline: off
void ll_set_splitk_l2_internal(int v) { g_gemm_splitk_l2 = v; }
static int g_gemm_splitk_fault = 0;  // tuning key gemm_splitk_fault (tests only): 1 = the second workgroup of every pair exits before publishing
void ll_set_splitk_fault_internal(int v) { g_gemm_splitk_fault = v; }
// 1 when blocks b and b + 8k of a launch run on one XCD on this device (probed once with 2048 single-wave blocks), else 0
static int splitk_same_xcd() {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cached[dev] == 0) {
    cached[dev] = -1;
    const int n = 2048;
    unsigned* d = nullptr;
    unsigned h[n];
    if (hipMalloc(&d, n * sizeof(unsigned)) == hipSuccess) {
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(n), dim3(64), 0, 0, d);
      if (hipMemcpy(h, d, n * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) {
        bool ok = true;
        for (int i = 8; i < n && ok; ++i) ok = h[i] == h[i & 7];
        cached[dev] = ok ? 1 : -1;
      }
      (void)hipFree(d);
    }
  }
  return cached[dev] > 0 ? 1 : 0;
}

// Co-residency: the two workgroups of a pair wait for each other, so both must be resident at once.  The launcher guarantees
// it for THIS launch only (grid <= CU count at one 128-KiB-LDS workgroup per CU); a co-running launch on another stream (the VAE
// decoder's 132-KiB-LDS convolutions under overlap_decode, a second split-K FFN2 under overlap_context, another process) can hold
// CUs, and progress then rests on the dispatcher handing freed CUs to the oldest pending workgroups -- observed, not promised.
// That is why the hand-off is fail-safe rather than assumed: epoch-valued flags (a stale or foreign word never matches) and a
// poll bounded by real time (V4SK_POLL_TICKS) that records the failure in the workspace instead of hanging the wave.
static bool splitk_eligible(int M, int N, int kbytes) {      // kbytes = bytes of K per operand row (bf16: 2 K, int8: K)
  if (M <= 0 || N <= 0 || N % 256 != 0 || kbytes % 256 != 0 || kbytes < 2048) return false;
  const int tiles = splitk_tiles(M, N);
  const int grid = 16 * ((tiles + 7) / 8);
  return tiles * 2 <= 1024 && grid <= device_cus();            // every workgroup resident at once: partners wait for each other
}
extern "C" long long ll_gemm_splitk_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0 || N % 256 != 0) return 0;
  return (long long)V4SK_FLAG_BYTES + (long long)splitk_tiles(M, N) * 2 * V4SK_PART_FLOATS * 4;
}
extern "C" int ll_gemm_splitk_plan(int M, int N, int K, int int8) { return splitk_eligible(M, N, int8 ? K : 2 * K) ? 1 : 0; }

extern "C" int ll_gemm_plan_epi(int M, int N, int K, int int8, int epilogue, int plain, int splitk_call, char* out, int cap) {
  LL_REQUIRE(out != nullptr && cap > 0, "ll_gemm_plan_epi: needs an output buffer");
  const bool sk = splitk_call && splitk_eligible(M, N, int8 ? K : 2 * K);
  if (!int8 && gemm_asm_wanted(epilogue) && (!sk || (g_gemm_asm & 2))) {
    // plain: 1 = an ordinary call, 0 = per-batch modulation vector (HIP kernels), 2 = the fused QKV call with its V redirect (B = 1)
    const int wn = gemm_asm_width(M, N, K, K, epilogue, plain != 0, plain == 2, plain == 2 && (2 * (N / 3)) % 192 == 0, 1);
    if (wn) { gemm_asm_plan(M, N, wn, epilogue, out, cap, false); ll_plan_append_knobs(out, cap); return LL_OK; }
  }
  if (int8 && (g_gemm_asm & 16) && gemm_asm_wanted(epilogue) && !sk) {      // bit 4: W8A8 calls on the generated kernels (launch_gemm<true>)
    const int wn = gemm_asm_width_i8(M, N, K, epilogue, plain != 0, plain == 2, plain == 2 && (2 * (N / 3)) % 192 == 0, 1);
    if (wn) { gemm_asm_plan(M, N, wn, epilogue, out, cap, true); ll_plan_append_knobs(out, cap); return LL_OK; }
  }
  if (sk) {
    snprintf(out, (size_t)cap, "gemm_kernel_v4sk<%s> tile 256x256 x split-K 2, %d workgroups, halves reduced in the epilogue",
             int8 ? "i8" : "bf16", 2 * ((M + 255) / 256) * (N / 256));
    return LL_OK;
  }
  return ll_gemm_plan(M, N, K, int8, out, cap);
}

// Reads back the workspace's error word (BLOCKING: synchronises `stream`).  *status = 0: every hand-off of every launch on this
// workspace so far completed; otherwise the epoch of a launch whose partner workgroup did not arrive within the poll budget --
// that launch's output is invalid.  The word is cleared, so the workspace can be used again.
extern "C" int ll_gemm_splitk_status(void* workspace, unsigned* status, ll_stream stream) {
  LL_REQUIRE(workspace != nullptr && status != nullptr, "ll_gemm_splitk_status: needs a workspace and an output word");
  unsigned* w = (unsigned*)workspace + V4SK_ERR_WORD;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(status, w, sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    ll_set_error("ll_gemm_splitk_status: read-back failed: %s", hipGetErrorString(hipGetLastError()));
    return LL_ERR_LAUNCH;
  }
  if (*status != 0 && hipMemsetAsync(w, 0, sizeof(unsigned), s) != hipSuccess) {
    ll_set_error("ll_gemm_splitk_status: could not clear the error word");
    return LL_ERR_LAUNCH;
  }
  return LL_OK;
}

template <bool I8>
static int launch_splitk(const char* fn, const void* x, const void* w, bf16* out, int M, int N, int K, size_t xrow_bytes,
                         size_t wrow_bytes, int ldo, int epilogue, const EpiArgs& ea, void* workspace, long long workspace_bytes,
                         hipStream_t s) {
  const int kbytes = I8 ? K : 2 * K;
  if (!splitk_eligible(M, N, kbytes) || workspace == nullptr) {
    if (int lrc = launch_gemm<I8>(x, w, out, M, N, K, xrow_bytes, wrow_bytes, ldo, epilogue, ea, s)) return lrc;
    return ll_check_launch(fn);
  }
  // gemm_asm bit 1: the generated 256 x 128 kernel where it covers the call (measured: 109 us / 145 mJ against 123 us / 166 mJ
  // for the split-K kernel at FFN2's shape, profiles/r03_kenergy_gemm.txt) -- no partner hand-off on that path
  if (!I8 && (g_gemm_asm & 2) && gemm_asm_wanted(epilogue) && wrow_bytes == (size_t)K * 2) {
    const int r = gemm_asm_launch((const bf16*)x, (const bf16*)w, out, M, N, K, (int)(xrow_bytes / 2), ldo, epilogue, ea, g_gemm_group_m, s);
    if (r) return r < 0 ? r : ll_check_launch(fn);
  }
  LL_REQUIRE(workspace_bytes >= ll_gemm_splitk_workspace_bytes(M, N) && ((size_t)workspace & 15) == 0,
             "%s: workspace of %lld bytes, need %lld (16-byte aligned)", fn, workspace_bytes, ll_gemm_splitk_workspace_bytes(M, N));
  const int tiles = splitk_tiles(M, N), ntn = N / 256, nkh = kbytes / (2 * ROWB);
  dim3 grid(16 * ((tiles + 7) / 8)), block(512);
  unsigned* flags = (unsigned*)workspace;
  float* part = (float*)((char*)workspace + V4SK_FLAG_BYTES);
  const size_t lds = 2 * V3_STAGE;
  const bool l2 = g_gemm_splitk_l2 && splitk_same_xcd();
  static std::atomic<unsigned> launch_counter{0};
  unsigned epoch = launch_counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
  if (epoch == 0) epoch = launch_counter.fetch_add(1u, std::memory_order_relaxed) + 1u;      // 0 = "no flag"
#define SK_LAUNCH(E)                                                                                                   \
  do {                                                                                                                 \
    {                                                                                           \
      (void)ll_lds_attr((const void*)gemm_kernel_v4sk<E, I8, false>, (int)lds); \
      (void)ll_lds_attr((const void*)gemm_kernel_v4sk<E, I8, true>, (int)lds); \
    }                                                                                                                  \
    if (l2)                                                                                                            \
      hipLaunchKernelGGL((gemm_kernel_v4sk<E, I8, true>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nkh, \
                         xrow_bytes, wrow_bytes, ldo, tiles, ntn, part, flags, ea, epoch, g_gemm_splitk_fault);        \
    else                                                                                                               \
      hipLaunchKernelGGL((gemm_kernel_v4sk<E, I8, false>), grid, block, lds, s, (const char*)x, (const char*)w, out, M, N, nkh, \
                         xrow_bytes, wrow_bytes, ldo, tiles, ntn, part, flags, ea, epoch, g_gemm_splitk_fault);        \
  } while (0)
  switch (epilogue) {
    case LL_EPI_BIAS: SK_LAUNCH(LL_EPI_BIAS); break;
    case LL_EPI_BIAS_GELU: SK_LAUNCH(LL_EPI_BIAS_GELU); break;
    case LL_EPI_BIAS_GATE_RES: SK_LAUNCH(LL_EPI_BIAS_GATE_RES); break;
    default: SK_LAUNCH(LL_EPI_BIAS_RES); break;
  }
#undef SK_LAUNCH
  return ll_check_launch(fn);
}

extern "C" int ll_gemm_bf16_splitk(const ll_bf16* x, const ll_bf16* w, const ll_bf16* bias, ll_bf16* out, int M, int N, int K,
                                   int ldx, int ldo, int epilogue, const ll_bf16* res, const ll_bf16* e, const ll_bf16* mod,
                                   int nmod, int gate_idx, int rows_per_batch, int frame_len, void* workspace,
                                   long long workspace_bytes, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 64 == 0, "ll_gemm_bf16_splitk: K=%d must be a positive multiple of 64", K);
  LL_REQUIRE(ldx >= K && ldx % 8 == 0, "ll_gemm_bf16_splitk: ldx=%d must be >= K and a multiple of 8", ldx);
  int rc = check_epilogue("ll_gemm_bf16_splitk", M, N, ldo, epilogue, bias, res, e, mod, nmod, gate_idx, rows_per_batch, frame_len);
  if (rc) return rc;
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, (const bf16*)e, (const bf16*)mod, nullptr, nullptr, nmod, gate_idx,
             rows_per_batch, frame_len, frame_len > 0 && rows_per_batch > 0 ? rows_per_batch / frame_len : 0};
  return launch_splitk<false>("ll_gemm_bf16_splitk", x, w, (bf16*)out, M, N, K, (size_t)ldx * 2, (size_t)K * 2, ldo, epilogue, ea,
                              workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ll_gemm_w8a8_splitk(const int8_t* xq, const float* sx, const int8_t* wq, const float* sw, const ll_bf16* bias,
                                   ll_bf16* out, int M, int N, int K, int ldo, int epilogue, const ll_bf16* res,
                                   const ll_bf16* e, const ll_bf16* mod, int nmod, int gate_idx, int rows_per_batch,
                                   int frame_len, void* workspace, long long workspace_bytes, ll_stream stream) {
  LL_REQUIRE(K > 0 && K % 128 == 0, "ll_gemm_w8a8_splitk: K=%d must be a positive multiple of 128", K);
  LL_REQUIRE(sx && sw, "ll_gemm_w8a8_splitk: activation and weight scales are required");
  int rc = check_epilogue("ll_gemm_w8a8_splitk", M, N, ldo, epilogue, bias, res, e, mod, nmod, gate_idx, rows_per_batch, frame_len);
  if (rc) return rc;
  if (M == 0) return LL_OK;
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, (const bf16*)e, (const bf16*)mod, sx, sw, nmod, gate_idx,
             rows_per_batch, frame_len, frame_len > 0 && rows_per_batch > 0 ? rows_per_batch / frame_len : 0};
  return launch_splitk<true>("ll_gemm_w8a8_splitk", xq, wq, (bf16*)out, M, N, K, (size_t)K, (size_t)K, ldo, epilogue, ea, workspace,
                             workspace_bytes, (hipStream_t)stream);
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
  hipLaunchKernelGGL(quantize_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, q, scale,
                     rows, K, ldx);
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
