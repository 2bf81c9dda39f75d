// Wan VAE decoder kernels: causal 3-D / 2-D convolution as an implicit GEMM on the MFMA
// GEMM structure of gemm.hip, plus the decoder's row kernels.  Activations are CHANNELS-LAST bf16 [T, H, W, C].
//
// Convolution (CausalConv3d 3x3x3 / (3,1,1) / 1x1x1 and Conv2d 3x3 after a nearest x2 upsample, wan/modules/vae.py:17-36,
// 74-84):   out[(t,ho,wo), co] = bias[co] + sum_{tap,ci} in[t + kt - (KT-1), (ho + kh - p) >> up, (wo + kw - p) >> up, ci] * w[co, tap, ci]
//   = a GEMM with M = T*Ho*Wo output pixels, N = Cout, K = taps*Cin ordered (tap, ci).  In channels-last layout every
//   16-byte chunk of a K-row is 8 consecutive channels of ONE shifted input pixel, and the LDS-DMA source address is per
//   lane -- so the A tile is gathered straight into the GEMM's swizzled LDS image: no im2col buffer, padding = a pointer to
//   a zero row, the two frames of temporal context = a pointer into the layer's cache, the nearest-neighbour upsample =
//   a shift of the source coordinates.  Weights are re-ordered once at load to [Cout, Kpad] (Kpad = K rounded up to 64).
// Tile: 256 pixels x 32*NT channels, 8 waves (4 x 2), 3-stage LDS ring (same as gemm v2); NT = 3 for Cout = 96 / 192.
#include "gemm_common.h"

#define CV_BM 256
#define CV_STAGE ((CV_BM + 128) * ROWB)   // 48 KiB

struct ConvGeo {
  const char* x;        // first NEW frame of [2 + T, H, W, Cin] when KT == 3 (two history frames precede it), else [T, H, W, Cin]
  const char* zero;     // >= 16 zero bytes
  int T, H, W, Cin, Ho, Wo, KT, KH, up, cpt, nchunks, taps;
  unsigned inv_cpt;     // ceil(65536 / cpt): g / cpt == (g * inv_cpt) >> 16 for g < 4096
};

// A-tile staging.  LDS row r of the tile holds K-chunks 8s..8s+7 of output pixel r, chunk g at position (g & 7) ^ (r & 7)
// (the GEMM's XOR swizzle); every lane DMA-copies one 16-byte chunk per instruction and (r & 7) == (lane >> 3) & 7 for all
// of a lane's rows, so a lane always handles chunk column c = (lane & 7) ^ ((lane >> 3) & 7) of K-step s.
//
// MODE 1 / 2 (Cin >= 64, i.e. >= 8 chunks per tap): a K-step touches at most two taps, A = floor(8s / cpt) for chunk
// columns < split and A + 1 for the rest.  Both taps are decoded on the SCALAR unit (kt, kh, kw, byte offset); a lane picks
// one with a compare, adds it to its rows' precomputed base pointers and tests one precomputed validity bit per row:
// ~35 VALU per K-step per wave instead of ~170 with per-lane div/mod and bounds checks -- the VALU pipe, shared by the 4
// waves of a SIMD, was what bounded the first version of this kernel (profiles/r01_vae_kernels.md).
// MODE 2 adds the nearest x2 upsample: source row (ho + kh - 1) >> 1 = base + ((kh + parity) >> 1).
// MODE 0: generic per-lane decode for Cin < 64 (conv2, decoder.conv1: 16 input channels, < 0.1 % of the FLOPs).
template <int MODE>
struct ConvRows {
  const char* ptr[4];   // MODE 1/2: address of the (kt, kh, kw) = (0, 0, 0) tap of each row's pixel, channel 0
  unsigned mask[4];     // bit kh*KH + kw: tap inside the image; MODE 2: bits 16 / 17 = parity of (ho - 1) / (wo - 1)
  int rt[4], rh[4], rw[4];   // MODE 0 only
};

template <int MODE>
__device__ __forceinline__ void conv_rows_init(ConvRows<MODE>& R, const ConvGeo& g, int m0, int M, int wave, int lane) {
  const int hw = g.Ho * g.Wo, pad = g.KH >> 1;
  const long long fb = (long long)g.H * g.W * g.Cin * 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (wave * 4 + i) * 8 + (lane >> 3);
    m = m < M ? m : M - 1;
    int t = m / hw;
    int rem = m - t * hw;
    int h = rem / g.Wo;
    int w = rem - h * g.Wo;
    R.rt[i] = t; R.rh[i] = h; R.rw[i] = w;
    unsigned mk = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int hy = h + kh - pad, wx = w + kw - pad;
        bool ok = kh < g.KH && kw < g.KH && hy >= 0 && hy < g.Ho && wx >= 0 && wx < g.Wo;
        mk |= ok ? 1u << (kh * g.KH + kw) : 0u;
      }
    if (MODE == 2) {
      int a = h - 1, b = w - 1;
      mk |= (unsigned)(a & 1) << 16 | (unsigned)(b & 1) << 17;
      R.ptr[i] = g.x + (long long)t * fb + ((long long)(a >> 1) * g.W + (b >> 1)) * g.Cin * 2;
    } else {
      R.ptr[i] = g.x + (long long)(t - (g.KT - 1)) * fb + ((long long)(h - pad) * g.W + (w - pad)) * g.Cin * 2;
    }
    R.mask[i] = mk;
  }
}

struct TapDec { int idx, off, kh, kw; };
__device__ __forceinline__ TapDec decode_tap(const ConvGeo& g, int tap, int fb32) {   // uniform: runs on the scalar unit
  TapDec d;
  int kt = g.KH == 3 ? (tap * 57) >> 9 : tap;            // tap / 9 for tap < 27
  int rem = g.KH == 3 ? tap - 9 * kt : 0;
  d.kh = (rem * 11) >> 5;                                  // rem / 3 for rem < 9
  d.kw = rem - 3 * d.kh;
  bool live = tap < g.taps;
  d.idx = live ? rem : 31;                                 // bit 31 of a row mask is always 0: K padding reads the zero row
  d.off = live ? kt * fb32 + (d.kh * g.W + d.kw) * g.Cin * 2 : 0;
  return d;
}

template <int MODE>
__device__ __forceinline__ void stage_conv_rows(const ConvRows<MODE>& R, const ConvGeo& g, int kstep, char* lds, int wave,
                                                int lane) {
  const int c = (lane & 7) ^ ((lane >> 3) & 7);
  if (MODE == 0) {
    const int pad = g.KH >> 1;
    const long long fb = (long long)g.H * g.W * g.Cin * 2;
    int gch = kstep * 8 + c;
    int tap = (int)(((unsigned)gch * g.inv_cpt) >> 16);
    int ci8 = gch - tap * g.cpt;
    int kt = g.KH == 3 ? tap / 9 : tap;
    int rem = g.KH == 3 ? tap - kt * 9 : 0;
    int kh = rem / 3, kw = rem - kh * 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int inst = wave * 4 + i;
      int ti = R.rt[i] + kt - (g.KT - 1);
      int hy = R.rh[i] + kh - pad, wx = R.rw[i] + kw - pad;
      bool ok = gch < g.nchunks && hy >= 0 && hy < g.Ho && wx >= 0 && wx < g.Wo;
      if (g.up) { hy >>= 1; wx >>= 1; }
      const char* src = g.x + (long long)ti * fb + ((long long)(hy * g.W + wx) * g.Cin + ci8 * 8) * 2;
      src = ok ? src : g.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + inst * 1024), 16, 0, 0);
    }
  } else {
    const int fb32 = g.H * g.W * g.Cin * 2;
    int g0 = kstep * 8;
    int tapA = (int)(((unsigned)g0 * g.inv_cpt) >> 16);
    int ciA = g0 - tapA * g.cpt;
    int split = g.cpt - ciA;                               // chunk columns >= split belong to tap A + 1 (cpt >= 8)
    TapDec A = decode_tap(g, tapA, fb32), B = decode_tap(g, tapA + 1, fb32);
    bool inB = c >= split;
    int ci8 = inB ? c - split : ciA + c;
    int idx = inB ? B.idx : A.idx;
    if (MODE == 1) {
      unsigned off = (unsigned)((inB ? B.off : A.off) + ci8 * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int inst = wave * 4 + i;
        const char* src = ((R.mask[i] >> idx) & 1u) ? R.ptr[i] + off : g.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + inst * 1024), 16, 0, 0);
      }
    } else {
      int kh = inB ? B.kh : A.kh, kw = inB ? B.kw : A.kw;
      const int c2 = g.Cin * 2, wc2 = g.W * c2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int inst = wave * 4 + i;
        int dh = (kh + (int)((R.mask[i] >> 16) & 1u)) >> 1, dw = (kw + (int)((R.mask[i] >> 17) & 1u)) >> 1;
        unsigned off = (unsigned)(dh * wc2 + dw * c2 + ci8 * 16);
        const char* src = ((R.mask[i] >> idx) & 1u) ? R.ptr[i] + off : g.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + inst * 1024), 16, 0, 0);
      }
    }
  }
}

template <int EPI, int NT, int MODE>
__global__ __launch_bounds__(512, 2) void conv_cl_kernel(ConvGeo g, const char* __restrict__ Wt, bf16* __restrict__ Y,
                                                         int M, int N, int nk, size_t wrow_bytes, int ldo, int ntm,
                                                         int ntn, EpiArgs ea) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (lid / ntn) * CV_BM, n0 = (lid % ntn) * (32 * NT);

  ConvRows<MODE> R;
  conv_rows_init<MODE>(R, g, m0, M, wave, lane);

  f32x4 acc[NT][4];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kt, int slot) {
    char* base = smem + slot * CV_STAGE;
    stage_conv_rows<MODE>(R, g, kt, base, wave, lane);
    stage_rows(Wt, wrow_bytes, n0, N, kt * ROWB, base + CV_BM * ROWB, wave * 2, 2, lane);
  };
  stage(0, 0);
  if (nk > 1) stage(1, 1);

  const int fr = lane & 15, fg = lane >> 4;
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) {
      int s2 = slot + 2;
      s2 = s2 >= 3 ? s2 - 3 : s2;
      stage(kt + 2, s2);
    }
    const char* xs = smem + slot * CV_STAGE;
    const char* ws = xs + CV_BM * ROWB;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NT], xf[4];
      int ch = ks * 4 + fg;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        int rwi = wn * (16 * NT) + t * 16 + fr;
        wf[t] = *reinterpret_cast<const bf16x8*>(ws + rwi * ROWB + ((ch ^ (rwi & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int rx = wm * 64 + t * 16 + fr;
        xf[t] = *reinterpret_cast<const bf16x8*>(xs + rx * ROWB + ((ch ^ (rx & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    slot = slot == 2 ? 0 : slot + 1;
  }
  gemm_epilogue<EPI, false, NT, 4>(acc, Y, M, N, ldo, m0 + wm * 64, n0 + wn * (16 * NT), fr, fg, ea);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3x3 causal convolution with the input staged ONCE per (temporal tap, 32-channel slice) as a halo tile.
//
// The implicit GEMM above stages every shifted input pixel again for each of the 27 taps: 44 KiB of LDS-DMA per 64-deep K-step
// against 768 MFMA cycles (N = 96), and the CU's LDS-DMA path (~35-50 B/clk) is what bounds it (31 % of the MFMA peak on the
// six 96 -> 96 convolutions at 480 x 832 that are 70 % of the decoder's FLOPs).  Here a workgroup owns a 16 x 32 pixel tile of
// one output frame and 96 output channels; for each (kt, 32-channel slice) it stages the (16+2) x (32+2) halo of that input
// frame (38.3 KiB) and serves all nine spatial taps from it -- a tap is only a shifted LDS read address -- plus the 9 x 96 x 32
// weights in three 18-KiB units (one per kh): 94 KiB staged per 1728 MFMAs per wave pair instead of 396 KiB.
//   LDS: 2 halo slots x 39 KiB + 3 weight-unit slots x 18 KiB = 132 KiB; unit u = ((kt * nslice) + slice) * 3 + kh, two units
//   staged ahead, one barrier per unit (72 MFMAs per wave).  Halo / weight rows are 64 B (32 channels); 16-byte part q of row j
//   sits at position q ^ (((j >> 2) & 1) << 1): ds_read_b128 serves lanes {fr 0-3, 12-15 | part p} with {fr 4-11 | part p ^ 1}
//   per cycle group, i.e. for every row residue mod 4 the rows r, r + 12 at part p and r + 4, r + 8 at part p ^ 1: with the swizzle
//   on row bit 2 those four land on four different 16-byte slots for ANY start row, so the kw-shifted fragments are conflict-free
//   too (bit 3, the first choice, was conflict-free only for aligned runs: PMC 25 % of the LDS cycles were conflict cycles).
//   Waves: wave w owns image rows 2w, 2w+1 of the tile (4 blocks of 16 pixels) x 6 blocks of 16 channels = 96 accumulator
//   registers; the accumulation order over K is (kt, slice, kh, kw) instead of (tap, channel) -- same products, fp32 sums in
//   another order (parity bar unchanged: <= 2 bf16 ulp against fp32 conv3d).
#define HL_TH 16
#define HL_TW 32
#define HL_LDS(NCB, UP) (2 * hl_halo_pieces(UP) * 1024 + 3 * 3 * (NCB) * 1024)

// UP = 0: 3x3x3 causal convolution, halo (16+2) x (32+2) of input frame t - 2 + kt.
// UP = 1: Conv2d 3x3 on the nearest x2 upsampled frame (Resample, vae.py:74-84): output pixel (h, w), tap (kh, kw) reads source
//         pixel ((h + kh - 1) >> 1, (w + kw - 1) >> 1); the halo is the (8+2) x (16+2) SOURCE region of the tile (h0, w0 even),
//         a fragment's 16 rows are per-lane halo rows (two output pixels share a source pixel).
__host__ __device__ constexpr int hl_halo_w(int up) { return up ? HL_TW / 2 + 2 : HL_TW + 2; }
__host__ __device__ constexpr int hl_halo_px(int up) { return (up ? HL_TH / 2 + 2 : HL_TH + 2) * hl_halo_w(up); }   // 612 / 180
__host__ __device__ constexpr int hl_halo_pieces(int up) { return (hl_halo_px(up) + 15) / 16; }                      // 39 / 12

template <int N>
__device__ __forceinline__ void hl_wait_vm() {
  static_assert(N == 1 || N == 3 || N == 5 || N == 6 || N == 8, "add the literal");
  if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// RMS_norm (+ SiLU) over channels, channels-last rows of C in {96, 192, 384} (wan/modules/vae.py:39-55, 193-197):
//   n = bf16(||x||_2); y = bf16(bf16(bf16(x / max(n, 1e-12)) * sqrt(C)) * gamma); out = bf16(silu(y))
// (the reference's bf16 rounding points).  G lanes per pixel (16 / 32 / 64), 8 channels per lane.
// x / n for a pair with ONE reciprocal per lane: the steps of the IEEE-correct fp32 division the compiler emits (Newton step on the
// reciprocal -- done by the caller, once --, q0 = x r, two residual corrections) without its operand pre-scaling, which only acts when
// an exponent is beyond 2^+-96: the caller takes this path for n in [2^-60, 2^60] (and |x| <= n by construction), else the plain `/`.
__device__ __forceinline__ f32x2 div_by_shared(f32x2 x, float n, float r1) {
  const f32x2 nn = splat2(-n), rr = splat2(r1);
  f32x2 q = x * rr;
  f32x2 rem = __builtin_elementwise_fma(nn, q, x);
  q = __builtin_elementwise_fma(rem, rr, q);
  rem = __builtin_elementwise_fma(nn, q, x);
  return __builtin_elementwise_fma(rem, rr, q);
}


// Fused form for the convolutions whose workgroup holds ALL channels of a pixel (Cout = 16 NCB, one n-tile): the convolution's own
// epilogue -- v = bf16(acc + bias) [bf16(res + v)] -- followed by the RMS_norm (+ SiLU) the decoder applies to that tensor next
// (ResidualBlock: conv -> RMS_norm -> SiLU -> conv, vae.py:193-220), with rms_silu_cl_kernel's rounding points:
//   n = max(bf16(||v||_2), 1e-12); y = bf16(bf16(bf16(v / n) sqrt(C)) gamma); out2 = bf16(silu(y)).
// A lane holds 4 channels of each of the NCB 16-channel blocks of pixel (b, fr); the pixel's other channels sit in the three lanes
// with the same fr (fg = 0..3): the sum of squares is the lane's 4 NCB values in block order, then two xor shuffles -- another
// order of the fp32 sum than the stand-alone kernel's (8 consecutive channels per lane), so n may differ in its last bf16 bit in
// rare cases (tests/test_vae_gpu.py bounds the outputs at 1 ulp).  Y == nullptr: the un-normalised tensor is not needed, only Y2.
template <int EPI, int NCB>
__device__ __forceinline__ void conv_epilogue_rms(f32x4 (&acc)[NCB][2], bf16* __restrict__ Y, bf16* __restrict__ Y2, int M, int ldo,
                                                  int mw, int fr, int fg, const EpiArgs& ea, const bf16* __restrict__ gamma,
                                                  int do_silu, float sqrt_c) {
  static_assert(EPI == LL_EPI_BIAS || EPI == LL_EPI_BIAS_RES, "convolution epilogues");
  bf16x4 bv[NCB], gv[NCB];
#pragma unroll
  for (int a = 0; a < NCB; ++a) {
    bv[a] = *reinterpret_cast<const bf16x4*>(ea.bias + a * 16 + fg * 4);
    gv[a] = *reinterpret_cast<const bf16x4*>(gamma + a * 16 + fg * 4);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int m = mw + b * 16 + fr;
    const int mc = m < M ? m : M - 1;
    bf16x4 rv[NCB];
    if (EPI == LL_EPI_BIAS_RES) {
#pragma unroll
      for (int a = 0; a < NCB; ++a) rv[a] = *reinterpret_cast<const bf16x4*>(ea.res + (size_t)mc * ldo + a * 16 + fg * 4);
    }
    float v[NCB][4];
    float ss = 0.f;
#pragma unroll
    for (int a = 0; a < NCB; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = rbf((float)acc[a][b][j] + (float)bv[a][j]);
        if (EPI == LL_EPI_BIAS_RES) t = rbf((float)rv[a][j] + t);
        v[a][j] = t;
        ss += t * t;
      }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float n = fmaxf(rbf(sqrtf(ss)), 1e-12f);
    const bool tame = n >= 0x1p-60f && n <= 0x1p60f;
    const float r0 = __builtin_amdgcn_rcpf(n);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-n, r0, 1.0f), r0, r0);
    const f32x2 sc = splat2(sqrt_c), one = splat2(1.0f), nl2e = splat2(-1.4426950408889634f);
#pragma unroll
    for (int a = 0; a < NCB; ++a) {
      unsigned ow[2], yw[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x2 x2;
        x2.x = v[a][2 * h], x2.y = v[a][2 * h + 1];
        ow[h] = pack2(x2);
        f32x2 q;
        if (tame) q = div_by_shared(x2, n, r1);
        else q.x = x2.x / n, q.y = x2.y / n;
        f32x2 g2;
        g2.x = (float)gv[a][2 * h], g2.y = (float)gv[a][2 * h + 1];
        f32x2 y = rbf2(rbf2(rbf2(q) * sc) * g2);
        if (do_silu) {
          f32x2 e = nl2e * y;
          e.x = __builtin_amdgcn_exp2f(e.x);
          e.y = __builtin_amdgcn_exp2f(e.y);
          e = one + e;
          e.x = __builtin_amdgcn_rcpf(e.x);
          e.y = __builtin_amdgcn_rcpf(e.y);
          y = y * e;
        }
        yw[h] = pack2(y);
      }
      if (m < M) {
        const size_t off = (size_t)m * ldo + a * 16 + fg * 4;
        if (Y) *reinterpret_cast<uint2*>(Y + off) = make_uint2(ow[0], ow[1]);
        *reinterpret_cast<uint2*>(Y2 + off) = make_uint2(yw[0], yw[1]);
      }
    }
  }
}

// NCB = 16-channel output blocks per workgroup: 6 (96 channels) or 1 (the 3-channel head, weights padded to 8 rows).
template <int EPI, int NCB, int UP, bool RMS = false>
__global__ __launch_bounds__(512, 1) void conv_halo_kernel(const char* __restrict__ x, const char* __restrict__ zero,
                                                           const char* __restrict__ Wt, bf16* __restrict__ Y, int T, int H,
                                                           int W, int Cin, int N, size_t wrow_bytes, int ldo, int tiles_w,
                                                           int tiles_h, int ntn, EpiArgs ea, const bf16* __restrict__ rms_gamma = nullptr,
                                                           bf16* __restrict__ Y2 = nullptr, int rms_silu = 0, float rms_sqrt_c = 0.f) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HW_ = hl_halo_w(UP), HPX = hl_halo_px(UP), HPIECES = hl_halo_pieces(UP);
  constexpr int HALO_B = HPIECES * 1024, WPIECES = 3 * NCB, WUNIT_B = WPIECES * 1024;
  constexpr int NH = (HPIECES + 7) / 8, NWP = (WPIECES + 7) / 8;      // pieces per wave: halo 5 / 2, weights 3 / 1
  char* const hsm = smem;
  char* const wsm = smem + 2 * HALO_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = lid % ntn;
  int tile = lid / ntn;
  const int tw = tile % tiles_w;
  tile /= tiles_w;
  const int th = tile % tiles_h, t = tile / tiles_h;
  const int h0 = th * HL_TH, w0 = tw * HL_TW, n0 = nt * (16 * NCB);
  const int Wo = UP ? 2 * W : W, Ho = UP ? 2 * H : H;
  const long long fb = (long long)H * W * Cin * 2;
  const int nslice = Cin >> 5, nu = (UP ? 3 : 9) * nslice;

  // ---- staging plan (loop invariant per lane).  Every wave issues NWP weight pieces per unit and NH halo pieces per group;
  // a wave whose index runs past the last piece wraps around (same bytes to the same place) so that all vmcnt counts agree.
  int hoff[NH];           // byte offset of the lane's 16-byte part inside the source frame, channel slice 0; < 0: zero row
  int hdst[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    int p = (wave + 8 * k) % HPIECES;
    int j = 16 * p + (lane >> 2);
    int hr = j / HW_, hc = j - hr * HW_;
    int hy = (UP ? (h0 >> 1) : h0) - 1 + hr, wx = (UP ? (w0 >> 1) : w0) - 1 + hc;
    bool ok = j < HPX && hy >= 0 && hy < H && wx >= 0 && wx < W;
    int part = (lane & 3) ^ (((j >> 2) & 1) << 1);
    hoff[k] = ok ? (hy * W + wx) * Cin * 2 + part * 16 : -1;
    hdst[k] = p * 1024;
  }
  const char* wsrc[NWP];
  int wdst[NWP];
#pragma unroll
  for (int k = 0; k < NWP; ++k) {
    int bpc = (wave + 8 * k) % WPIECES;
    int kw = bpc / NCB, pp = bpc - NCB * kw;
    int row = 16 * pp + (lane >> 2);
    int part = (lane & 3) ^ (((row >> 2) & 1) << 1);
    int nrow = n0 + row;
    nrow = nrow < N ? nrow : N - 1;                          // head: 8 weight rows feed a 16-row block (rows >= N are not stored)
    wsrc[k] = Wt + (size_t)nrow * wrow_bytes + (size_t)kw * Cin * 2 + part * 16;
    wdst[k] = bpc * 1024;
  }
  const char* xt = x + (long long)(UP ? t : t - 2) * fb;
  auto issue = [&](int u) {
    int G = u / 3, kh = u - 3 * G;
    int kt = UP ? 0 : G / nslice, sl = G - kt * nslice;
    char* wb = wsm + (u % 3) * WUNIT_B;
    unsigned woff = (unsigned)(((kt * 9 + kh * 3) * Cin + sl * 32) * 2);
#pragma unroll
    for (int k = 0; k < NWP; ++k)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[k] + woff), (lptr_t)(wb + wdst[k]), 16, 0, 0);
    if (kh == 0) {
      char* hb = hsm + (G & 1) * HALO_B;
      const char* xf = xt + (long long)kt * fb + sl * 64;
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const char* src = hoff[k] >= 0 ? xf + hoff[k] : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(hb + hdst[k]), 16, 0, 0);
      }
    }
  };

  f32x4 acc[2][NCB][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int a = 0; a < NCB; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[i][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment rows.  UP = 0: halo row j = (2 wave + i + kh) * 34 + 16 b + kw + fr.  UP = 1: j = (((2 wave + i + kh - 1) >> 1) + 1) * 18
  // + ((16 b + fr + kw - 1) >> 1) + 1.  Weight row = kw * 16 NCB + 16 a + fr.
  const int jbase = (2 * wave) * HW_ + fr;
  const int wswz = ((fr >> 2) & 1) << 1;

  issue(0);
  if (nu > 1) issue(1);
  for (int u = 0; u < nu; ++u) {
    // unit u has landed; unit u + 1's pieces (NWP weight, + NH halo when it opens a group) may stay in flight
    if (u + 1 >= nu) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if ((u + 1) % 3 == 0) hl_wait_vm<NWP + NH>();
    else hl_wait_vm<NWP>();
    __builtin_amdgcn_s_barrier();
    if (u + 2 < nu) issue(u + 2);
    const int G = u / 3, kh = u - 3 * G;
    const char* hb = hsm + (G & 1) * HALO_B;
    const char* wb = wsm + (u % 3) * WUNIT_B;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      bf16x8 wf[NCB], xf[2][2];
#pragma unroll
      for (int a = 0; a < NCB; ++a)
        wf[a] = *reinterpret_cast<const bf16x8*>(wb + (kw * (16 * NCB) + 16 * a + fr) * 64 + ((fg ^ wswz) << 4));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          int j;
          if (UP) j = (((2 * wave + i + kh - 1) >> 1) + 1) * HW_ + ((16 * b + fr + kw - 1) >> 1) + 1;
          else j = jbase + (i + kh) * HW_ + 16 * b + kw;
          xf[i][b] = *reinterpret_cast<const bf16x8*>(hb + j * 64 + ((fg ^ (((j >> 2) & 1) << 1)) << 4));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int a = 0; a < NCB; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[i][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[i][b], acc[i][a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  // Partial tiles at the right / bottom edge: the halo reads outside the image were zero rows; here an image row below the frame
  // is skipped (wave-uniform) and the row limit handed to the epilogue is the END OF THIS IMAGE ROW, so pixels right of the
  // frame fail its m < M test (their loads are clamped to the row's last pixel).
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int hrow = h0 + 2 * wave + i;
    if (hrow >= Ho) continue;
    int m_row = (t * Ho + hrow) * Wo + w0;
    if (RMS) conv_epilogue_rms<EPI, NCB>(acc[i], Y, Y2, (t * Ho + hrow + 1) * Wo, ldo, m_row, fr, fg, ea, rms_gamma, rms_silu, rms_sqrt_c);
    else gemm_epilogue<EPI, false, NCB, 2>(acc[i], Y, (t * Ho + hrow + 1) * Wo, N, ldo, m_row, n0, fr, fg, ea);
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void rms_silu_cl_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gamma,
                                                          bf16* __restrict__ out, long long pixels, int C, float sqrt_c,
                                                          int do_silu) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (G - 1);
  long long pix = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / G) + (lane / G);
  if (pix >= pixels) return;
  const int c = sub * 8;
  // on pairs of channels (common.h): 155 vector instructions per lane instead of 316 -- the kernel was bound by them, not by HBM
  f32x2 v[4];
  unsigned gw[4] = {0u, 0u, 0u, 0u};
  float ss = 0.f;
  if (c < C) {
    uint4 t = *reinterpret_cast<const uint4*>(x + pix * C + c);
    uint4 g4 = *reinterpret_cast<const uint4*>(gamma + c);
    gw[0] = g4.x, gw[1] = g4.y, gw[2] = g4.z, gw[3] = g4.w;
    const unsigned xw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = unpack2(xw[j]);
      f32x2 sq = v[j] * v[j];
      ss += sq.x;            // element order, as the scalar form
      ss += sq.y;
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  float n = fmaxf(rbf(sqrtf(ss)), 1e-12f);
  if (c < C) {
    const bool tame = n >= 0x1p-60f && n <= 0x1p60f;
    float r0 = __builtin_amdgcn_rcpf(n);
    float r1 = __builtin_fmaf(__builtin_fmaf(-n, r0, 1.0f), r0, r0);
    const f32x2 sc = splat2(sqrt_c), one = splat2(1.0f), nl2e = splat2(-1.4426950408889634f);
    unsigned o[4];
    f32x2 q[4];
    if (tame) {
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = div_by_shared(v[j], n, r1);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[j].x = v[j].x / n;
        q[j].y = v[j].y / n;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 y = rbf2(rbf2(rbf2(q[j]) * sc) * unpack2(gw[j]));
      if (do_silu) {           // silu (common.h): y * rcp(1 + exp2(-log2(e) y))
        f32x2 e = nl2e * y;
        e.x = __builtin_amdgcn_exp2f(e.x);
        e.y = __builtin_amdgcn_exp2f(e.y);
        e = one + e;
        e.x = __builtin_amdgcn_rcpf(e.x);
        e.y = __builtin_amdgcn_rcpf(e.y);
        y = y * e;
      }
      o[j] = pack2(y);
    }
    *reinterpret_cast<uint4*>(out + pix * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// Row softmax for the decoder's single-head attention (vae.py:249-254): p = softmax(scale * s) over N columns, bf16 out.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const bf16* __restrict__ s, bf16* __restrict__ p, int rows, int N,
                                                           int ld, float scale_log2e) {
  int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16* sr = s + (size_t)row * ld;
  float mx = -INFINITY;
  for (int k = lane * 8; k < N; k += 512) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(sr + k);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, k + j < N ? (float)v[j] : -INFINITY);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float mc = mx * scale_log2e, sum = 0.f;
  for (int k = lane * 8; k < N; k += 512) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(sr + k);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      sum += k + j < N ? __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[j], scale_log2e, -mc)) : 0.f;
  }
  sum = wave_sum(sum);
  float inv = 1.0f / sum;
  bf16* pr = p + (size_t)row * ld;
  for (int k = lane * 8; k < ld; k += 512) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(sr + k);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)      // padding columns come out as zeros (the P.V contraction runs over ld)
      o[j] = (bf16)(k + j < N ? __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[j], scale_log2e, -mc)) * inv : 0.f);
    *reinterpret_cast<bf16x8*>(pr + k) = o;
  }
}

// WanVAEWrapper.decode_to_pixel's un-scaling (utils/wan_wrapper.py:99-110 -> vae.py:548-550: z / (1/std) + mean in bf16)
// fused with the layout change: latent [T, C, h, w] -> channels-last [T, h, w, C].
__global__ __launch_bounds__(256) void vae_unscale_cl_kernel(const bf16* __restrict__ z, const bf16* __restrict__ mean,
                                                             const bf16* __restrict__ inv_std, bf16* __restrict__ out, int C,
                                                             long long hw, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // index into out
  if (i >= total) return;
  int c = (int)(i % C);
  long long px = i / C, t = px / hw, p = px - t * hw;
  float v = (float)z[(t * C + c) * hw + p];
  out[i] = (bf16)(rbf(v / (float)inv_std[c]) + (float)mean[c]);
}

// Final layout change + clamp (utils/wan_wrapper.py:112-116): channels-last bf16 [T, H, W, ldc] (first 3 channels) ->
// fp32 [T, 3, H, W] clamped to [-1, 1].
__global__ __launch_bounds__(256) void cl_to_tchw_clamp_kernel(const bf16* __restrict__ x, float* __restrict__ out,
                                                               long long pixels_per_frame, long long total, int ldc) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  long long t = i / pixels_per_frame, p = i - t * pixels_per_frame;
  const bf16* px = x + i * ldc;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float v = (float)px[ch];
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    out[(t * 3 + ch) * pixels_per_frame + p] = v;
  }
}

// ===============================================================================================================
static int g_conv_halo = 1;       // tuning key conv_halo: 0 = always the implicit-GEMM kernel
void ll_set_conv_halo_internal(int v) { g_conv_halo = v; }

// the halo-tile kernel takes this convolution (every 3x3x3 / upsampled 1x3x3 layer of the decoder at 60x104 ... 480x832 with
// Cout % 96 == 0 or the <= 16-channel head): partial edge tiles are allowed when they waste < 30 % of the tile grid
static bool conv_halo_takes(int H, int W, int Cin, int Cout, int KT, int KH, int upsample) {
  const bool shape3 = KT == 3 && KH == 3 && !upsample, shape_up = KT == 1 && KH == 3 && upsample == 1;
  const bool head = shape3 && Cout <= 16, wide = Cout % 96 == 0;
  const int Ho = upsample ? 2 * H : H, Wo = upsample ? 2 * W : W;
  const int tiles_w = (Wo + HL_TW - 1) / HL_TW, tiles_h = (Ho + HL_TH - 1) / HL_TH;
  const bool fits = (long long)Ho * Wo * 10 >= 7ll * tiles_w * HL_TW * tiles_h * HL_TH && Wo % 2 == 0 && Ho % 2 == 0;
  return g_conv_halo && (shape3 || shape_up) && (wide || head) && Cin % 32 == 0 && fits;
}

// 1 = ll_conv_cl_rms covers this convolution: the halo-tile kernel with ALL output channels of a pixel in one workgroup (Cout = 96)
extern "C" int ll_conv_cl_rms_ok(int H, int W, int Cin, int Cout, int KT, int KH, int upsample) {
  return (Cout == 96 && conv_halo_takes(H, W, Cin, Cout, KT, KH, upsample)) ? 1 : 0;
}

static int conv_cl_launch(const ll_bf16* x, const ll_bf16* zero16, const ll_bf16* w, const ll_bf16* bias, const ll_bf16* res,
                          ll_bf16* out, int T, int H, int W, int Cin, int Cout, int Kpad, int KT, int KH, int upsample, int ldo,
                          ll_stream stream, const ll_bf16* rms_gamma, ll_bf16* out_rms, int rms_silu);

extern "C" int ll_conv_cl(const ll_bf16* x, const ll_bf16* zero16, const ll_bf16* w, const ll_bf16* bias, const ll_bf16* res,
                          ll_bf16* out, int T, int H, int W, int Cin, int Cout, int Kpad, int KT, int KH, int upsample,
                          int ldo, ll_stream stream) {
  LL_REQUIRE(out != nullptr, "ll_conv_cl: null operand");
  return conv_cl_launch(x, zero16, w, bias, res, out, T, H, W, Cin, Cout, Kpad, KT, KH, upsample, ldo, stream, nullptr, nullptr, 0);
}

extern "C" int ll_conv_cl_rms(const ll_bf16* x, const ll_bf16* zero16, const ll_bf16* w, const ll_bf16* bias, const ll_bf16* res,
                              ll_bf16* out, const ll_bf16* rms_gamma, ll_bf16* out_rms, int rms_silu, int T, int H, int W, int Cin,
                              int Cout, int Kpad, int KT, int KH, int upsample, int ldo, ll_stream stream) {
  LL_REQUIRE(rms_gamma != nullptr && out_rms != nullptr, "ll_conv_cl_rms: rms_gamma and out_rms are required (out may be NULL)");
  LL_REQUIRE(ll_conv_cl_rms_ok(H, W, Cin, Cout, KT, KH, upsample), "ll_conv_cl_rms: this convolution is not covered (ask ll_conv_cl_rms_ok "
             "first and run ll_conv_cl + ll_rms_silu_cl instead)");
  LL_REQUIRE(ldo == Cout, "ll_conv_cl_rms: ldo=%d must equal Cout (both outputs are dense [pixels, Cout])", ldo);
  return conv_cl_launch(x, zero16, w, bias, res, out, T, H, W, Cin, Cout, Kpad, KT, KH, upsample, ldo, stream, rms_gamma, out_rms, rms_silu);
}

static int conv_cl_launch(const ll_bf16* x, const ll_bf16* zero16, const ll_bf16* w, const ll_bf16* bias, const ll_bf16* res,
                          ll_bf16* out, int T, int H, int W, int Cin, int Cout, int Kpad, int KT, int KH, int upsample, int ldo,
                          ll_stream stream, const ll_bf16* rms_gamma, ll_bf16* out_rms, int rms_silu) {
  LL_REQUIRE(Cin > 0 && Cin % 8 == 0, "ll_conv_cl: Cin=%d must be a multiple of 8", Cin);
  LL_REQUIRE(Cout > 0 && Cout % 8 == 0, "ll_conv_cl: Cout=%d must be a multiple of 8 (pad the weights)", Cout);
  LL_REQUIRE((KT == 1 || KT == 3) && (KH == 1 || KH == 3), "ll_conv_cl: taps must be 1 or 3 (got %d x %d x %d)", KT, KH, KH);
  LL_REQUIRE(upsample == 0 || (upsample == 1 && KT == 1 && KH == 3), "ll_conv_cl: upsample is 0, or 1 with a 1x3x3 kernel");
  LL_REQUIRE(x && zero16 && w && bias && (out || out_rms), "ll_conv_cl: null operand");
  const int taps = KT * KH * KH;
  const int nchunks = taps * (Cin / 8);
  LL_REQUIRE(Kpad % 64 == 0 && Kpad >= nchunks * 8 && Kpad < nchunks * 8 + 64, "ll_conv_cl: Kpad=%d does not match taps*Cin=%d", Kpad, nchunks * 8);
  LL_REQUIRE(nchunks < 4096, "ll_conv_cl: K too large for the chunk decoder");
  LL_REQUIRE(ldo >= Cout && ldo % 4 == 0, "ll_conv_cl: ldo=%d must be >= Cout and a multiple of 4", ldo);
  const long long fb = (long long)H * W * Cin * 2;
  LL_REQUIRE(3 * fb + 4ll * (W + 2) * Cin < (1ll << 31), "ll_conv_cl: frame of %lld bytes too large for 32-bit tap offsets", fb);
  const int Ho = upsample ? 2 * H : H, Wo = upsample ? 2 * W : W;
  const long long Mll = (long long)T * Ho * Wo;
  LL_REQUIRE(Mll > 0 && Mll < (1ll << 31), "ll_conv_cl: too many output pixels");
  const int M = (int)Mll;
  ConvGeo g;
  g.x = (const char*)x; g.zero = (const char*)zero16;
  g.T = T; g.H = H; g.W = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.KT = KT; g.KH = KH; g.up = upsample;
  g.cpt = Cin / 8; g.nchunks = nchunks; g.taps = taps; g.inv_cpt = (65536u + g.cpt - 1) / g.cpt;
  EpiArgs ea{(const bf16*)bias, (const bf16*)res, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  {
    // halo-tile kernel: 3x3x3 (96-channel tiles, or the <= 16-channel head) and the upsampled 1x3x3 (96-channel tiles)
    const bool shape3 = KT == 3 && KH == 3 && !upsample, shape_up = KT == 1 && KH == 3 && upsample == 1;
    const bool head = shape3 && Cout <= 16;
    const int tiles_w = (Wo + HL_TW - 1) / HL_TW, tiles_h = (Ho + HL_TH - 1) / HL_TH;
    if (conv_halo_takes(H, W, Cin, Cout, KT, KH, upsample)) {
      const int ntn_h = head ? 1 : Cout / 96;
      const long long nwg = (long long)T * tiles_h * tiles_w * ntn_h;
      LL_REQUIRE(nwg < (1ll << 31), "ll_conv_cl: too many tiles");
      dim3 hgrid((unsigned)nwg), hblock(512);
#define HL_LAUNCH(E, NCBV, UPV)                                                                                        \
      do {                                                                                                             \
        {                                                                                      \
          (void)ll_lds_attr((const void*)conv_halo_kernel<E, NCBV, UPV>, \
                                    (int)HL_LDS(NCBV, UPV));                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((conv_halo_kernel<E, NCBV, UPV>), hgrid, hblock, HL_LDS(NCBV, UPV), s, (const char*)x,      \
                           (const char*)zero16, (const char*)w, (bf16*)out, T, H, W, Cin, Cout, (size_t)Kpad * 2, ldo,  \
                           tiles_w, tiles_h, ntn_h, ea);                                                               \
      } while (0)
      if (rms_gamma != nullptr) {           // Cout == 96: one n-tile holds every channel of a pixel (ll_conv_cl_rms_ok)
        const float sqc = sqrtf((float)Cout);
#define HL_LAUNCH_RMS(E, UPV)                                                                                          \
        do {                                                                                                           \
          (void)ll_lds_attr((const void*)conv_halo_kernel<E, 6, UPV, true>, (int)HL_LDS(6, UPV));                      \
          hipLaunchKernelGGL((conv_halo_kernel<E, 6, UPV, true>), hgrid, hblock, HL_LDS(6, UPV), s, (const char*)x,    \
                             (const char*)zero16, (const char*)w, (bf16*)out, T, H, W, Cin, Cout, (size_t)Kpad * 2, ldo, \
                             tiles_w, tiles_h, ntn_h, ea, (const bf16*)rms_gamma, (bf16*)out_rms, rms_silu, sqc);       \
        } while (0)
        if (shape_up) { if (res) HL_LAUNCH_RMS(LL_EPI_BIAS_RES, 1); else HL_LAUNCH_RMS(LL_EPI_BIAS, 1); }
        else { if (res) HL_LAUNCH_RMS(LL_EPI_BIAS_RES, 0); else HL_LAUNCH_RMS(LL_EPI_BIAS, 0); }
#undef HL_LAUNCH_RMS
        return ll_check_launch("ll_conv_cl_rms(halo)");
      }
      if (shape_up) {
        if (res) HL_LAUNCH(LL_EPI_BIAS_RES, 6, 1); else HL_LAUNCH(LL_EPI_BIAS, 6, 1);
      } else if (head) {
        if (res) HL_LAUNCH(LL_EPI_BIAS_RES, 1, 0); else HL_LAUNCH(LL_EPI_BIAS, 1, 0);
      } else {
        if (res) HL_LAUNCH(LL_EPI_BIAS_RES, 6, 0); else HL_LAUNCH(LL_EPI_BIAS, 6, 0);
      }
#undef HL_LAUNCH
      return ll_check_launch("ll_conv_cl(halo)");
    }
  }
  LL_REQUIRE(rms_gamma == nullptr, "ll_conv_cl_rms: only the halo-tile kernel has the fused RMS_norm epilogue");
  const int nk = Kpad / 64;
  const bool nt3 = (Cout % 96 == 0) && (Cout % 128 != 0), nt1 = Cout <= 32;
  const int bn = nt1 ? 32 : nt3 ? 96 : 128;
  const int mode = Cin < 64 ? 0 : upsample ? 2 : 1;
  int ntm = (M + CV_BM - 1) / CV_BM, ntn = (Cout + bn - 1) / bn;
  dim3 grid(ntm * ntn), block(512);
  size_t lds = 3 * CV_STAGE;
#define CV_LAUNCH(E, NTV, MD)                                                                                          \
  do {                                                                                                                 \
    {                                                                                          \
      (void)ll_lds_attr((const void*)conv_cl_kernel<E, NTV, MD>, (int)lds); \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_cl_kernel<E, NTV, MD>), grid, block, lds, s, g, (const char*)w, (bf16*)out, M, Cout, nk,   \
                       (size_t)Kpad * 2, ldo, ntm, ntn, ea);                                                           \
  } while (0)
#define CV_MODES(E, NTV)                                                                                               \
  do {                                                                                                                 \
    if (mode == 0) CV_LAUNCH(E, NTV, 0); else if (mode == 1) CV_LAUNCH(E, NTV, 1); else CV_LAUNCH(E, NTV, 2);          \
  } while (0)
  if (res) {
    if (nt1) CV_MODES(LL_EPI_BIAS_RES, 1); else if (nt3) CV_MODES(LL_EPI_BIAS_RES, 3); else CV_MODES(LL_EPI_BIAS_RES, 4);
  } else {
    if (nt1) CV_MODES(LL_EPI_BIAS, 1); else if (nt3) CV_MODES(LL_EPI_BIAS, 3); else CV_MODES(LL_EPI_BIAS, 4);
  }
#undef CV_MODES
#undef CV_LAUNCH
  return ll_check_launch("ll_conv_cl");
}

extern "C" int ll_rms_silu_cl(const ll_bf16* x, const ll_bf16* gamma, ll_bf16* out, long long pixels, int C, int do_silu,
                              ll_stream stream) {
  LL_REQUIRE(C > 0 && C % 8 == 0 && C <= 512, "ll_rms_silu_cl: C=%d must be a multiple of 8 and <= 512", C);
  if (pixels == 0) return LL_OK;
  float sc = sqrtf((float)C);
  hipStream_t s = (hipStream_t)stream;
  if (C <= 128) {
    long long per_block = 4 * 4;
    hipLaunchKernelGGL(rms_silu_cl_kernel<16>, dim3((unsigned)((pixels + per_block - 1) / per_block)), dim3(256), 0, s,
                       (const bf16*)x, (const bf16*)gamma, (bf16*)out, pixels, C, sc, do_silu);
  } else if (C <= 256) {
    long long per_block = 4 * 2;
    hipLaunchKernelGGL(rms_silu_cl_kernel<32>, dim3((unsigned)((pixels + per_block - 1) / per_block)), dim3(256), 0, s,
                       (const bf16*)x, (const bf16*)gamma, (bf16*)out, pixels, C, sc, do_silu);
  } else {
    long long per_block = 4;
    hipLaunchKernelGGL(rms_silu_cl_kernel<64>, dim3((unsigned)((pixels + per_block - 1) / per_block)), dim3(256), 0, s,
                       (const bf16*)x, (const bf16*)gamma, (bf16*)out, pixels, C, sc, do_silu);
  }
  return ll_check_launch("ll_rms_silu_cl");
}

extern "C" int ll_softmax_rows(const ll_bf16* s, ll_bf16* p, int rows, int N, int ld, float scale, ll_stream stream) {
  LL_REQUIRE(N > 0 && ld >= N && ld % 8 == 0, "ll_softmax_rows: need 0 < N=%d <= ld=%d and ld a multiple of 8", N, ld);
  if (rows == 0) return LL_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)s, (bf16*)p,
                     rows, N, ld, scale * 1.4426950408889634f);
  return ll_check_launch("ll_softmax_rows");
}

extern "C" int ll_vae_unscale_cl(const ll_bf16* z, const ll_bf16* mean, const ll_bf16* inv_std, ll_bf16* out, int T, int C,
                                 int h, int w, ll_stream stream) {
  long long hw = (long long)h * w, total = hw * T * C;
  if (total == 0) return LL_OK;
  hipLaunchKernelGGL(vae_unscale_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)z, (const bf16*)mean, (const bf16*)inv_std, (bf16*)out, C, hw, total);
  return ll_check_launch("ll_vae_unscale_cl");
}

extern "C" int ll_cl_to_tchw_clamp(const ll_bf16* x, float* out, int T, int H, int W, int ldc, ll_stream stream) {
  LL_REQUIRE(ldc >= 3, "ll_cl_to_tchw_clamp: needs >= 3 channels per pixel");
  long long ppf = (long long)H * W, total = ppf * T;
  if (total == 0) return LL_OK;
  hipLaunchKernelGGL(cl_to_tchw_clamp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, out, ppf, total, ldc);
  return ll_check_launch("ll_cl_to_tchw_clamp");
}
