// included by gemm_asm.hip once per (tile width, epilogue) with GA_NAME / GA_INC defined: the PERSISTENT form of the generated GEMM
// (gen/gemm_asm_gen.py generate(persistent=True)).  The text maps tiles to (m-tile, n-tile) itself and computes every per-tile
// scalar -- incl. the V-cache redirect of the fused QKV projection -- so this wrapper only pins the launch's arguments.
__global__ __launch_bounds__(256, 1) void GA_NAME(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                  const bf16* __restrict__ bias, bf16* __restrict__ Y,
                                                  const bf16* __restrict__ res, const bf16* __restrict__ gate, int M, int N,
                                                  int K, int ldx, int ldo, int frame_len, int gate_stride, int ntm, int ntn,
                                                  int gm, bf16* __restrict__ v_out, int v_col0, int v_C, int v_shift, int v_lo,
                                                  int v_hi) {
  unsigned long long xb = (unsigned long long)X, wb = (unsigned long long)W, yb = (unsigned long long)Y, bb = (unsigned long long)bias;
  unsigned long long rb = (unsigned long long)(res ? res : Y), gb = (unsigned long long)(gate ? gate : bias), vb = (unsigned long long)v_out;
  unsigned ldx_b = (unsigned)ldx * 2u, ldw_b = (unsigned)K * 2u, ldo_b = (unsigned)ldo * 2u, um = (unsigned)M, un = (unsigned)N, nk = (unsigned)(K / 64);
  unsigned flen = (unsigned)(frame_len > 0 ? frame_len : 1), gstride = (unsigned)gate_stride;
  unsigned tile = blockIdx.x, grid = gridDim.x, ntiles = (unsigned)(ntm * ntn), untm = (unsigned)ntm, untn = (unsigned)ntn, ugm = (unsigned)(gm > 1 ? gm : 1);
  unsigned vcol0 = (unsigned)v_col0, vc2 = (unsigned)v_C * 2u, vlo = (unsigned)v_lo, vhi = (unsigned)v_hi, tid = threadIdx.x;
  asm volatile(
#include GA_INC
      :
      : "{s[84:85]}"(xb), "{s[86:87]}"(wb), "{s[88:89]}"(yb), "{s[90:91]}"(bb), "{s[92:93]}"(rb), "{s[94:95]}"(gb), "{s20}"(ldx_b),
        "{s21}"(ldw_b), "{s68}"(ldo_b), "{s69}"(um), "{s70}"(un), "{s25}"(nk), "{s26}"(flen), "{s27}"(gstride), "{s71}"(tile),
        "{s72}"(grid), "{s73}"(ntiles), "{s74}"(untm), "{s75}"(untn), "{s76}"(ugm), "{s[78:79]}"(vb), "{s80}"(vcol0), "{s81}"(vc2),
        "{s82}"(v_shift), "{s83}"(vlo), "{s77}"(vhi), "{v0}"(tid)
      : "memory", "v255", "a255", "s67", "vcc");
  __builtin_unreachable();
}
