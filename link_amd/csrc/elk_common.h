// link_amd/csrc/elk_common.h -- device helpers shared by elk.hip (general path) and dense.hip (dense-cell path).
#pragma once
#include "common.h"

namespace link {

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// theta for one channel of one voxel.  theta = ((x/div)*w0 + (y/div)*w1) + (z/div)*w2 evaluated as an
// fma chain in x,y,z order (nn.Linear(3, cg, bias=False) on float coords, linkunet.py:151), then
// * alpha for cos_x (linkunet.py:165).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float theta_of(float x, float y, float z, float w0, float w1, float w2,
                                          float alpha) {
  float t = fmaf(z, w2, fmaf(y, w1, x * w0));
  return t * alpha;
}

// sin & cos of a moderate argument: 3-term Cody-Waite reduction by pi/2 + degree-7/8 minimax
// polynomials on [-pi/4, pi/4] (<= ~1.5 ulp for |x| < 2^15); larger or non-finite arguments take the
// library path.  Branch-free on the fast path: ~25 VALU ops instead of the library's table walk.
static __device__ __noinline__ void sincos_slow(float x, float *sn, float *cs) { sincosf(x, sn, cs); }

__device__ __forceinline__ void sincos_fast(float x, float &sn, float &cs) {
  if (__builtin_expect(!(fabsf(x) < 32768.0f), 0)) {
    sincos_slow(x, &sn, &cs);     // kept out of line: the fast path is what the i-cache should hold
    return;
  }
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.5703125f, x);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.54978995489188216e-8f, r);
  const float r2 = r * r;
  float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  ps = fmaf(ps * r2, r, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  pc = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)k;
  const float s0 = (q & 1) ? pc : ps;
  const float c0 = (q & 1) ? ps : pc;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}

// Call-free variant used by the dense-cell kernels (a call to the out-of-line libm path costs every kernel
// that contains it the callee's registers and a scratch frame).  |x| < 2^15: the same float path as
// sincos_fast, bit for bit.  Larger arguments: the reduction is done in double (k = rint(x * 2/pi),
// r = x - k * pi/2 with a two-term pi/2 and fma): exact to float precision for |x| < 2^50, far beyond any
// theta = coords . w a voxel grid produces; non-finite x gives NaN like sinf / cosf.
__device__ __forceinline__ void sincos_nocall(float x, float &sn, float &cs) {
  float r;
  int q;
  if (__builtin_expect(fabsf(x) < 32768.0f, 1)) {
    const float k = rintf(x * 0.63661977236758134f);
    r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    q = (int)k;
  } else {
    const double xd = (double)x;
    const double kd = rint(xd * 0.63661977236758134308);
    double rd = fma(-kd, 1.57079632679489655800e+00, xd);
    rd = fma(-kd, 6.12323399573676603587e-17, rd);
    r = (float)rd;
    q = (int)(kd - 4.0 * floor(kd * 0.25));
  }
  const float r2 = r * r;
  float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  ps = fmaf(ps * r2, r, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  pc = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const float s0 = (q & 1) ? pc : ps;
  const float c0 = (q & 1) ? ps : pc;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}

// The |x| < 2^15 body of sincos_nocall alone, branch-free (bit-identical there): for code that has already
// established the range for the whole wave and must stay one basic block.
__device__ __forceinline__ void sincos_hw(float x, float &sn, float &cs);
#ifndef LINK_HW_TRIG
#define LINK_HW_TRIG 1      /* fused dense-cell kernels: hardware trig (sincos_hw below; 4e-7 absolute, inside the path's 1e-4 bar).
                               -DLINK_HW_TRIG=0 restores the libm-rounded polynomial (bit-identical to the general layout) */
#endif
__device__ __forceinline__ void sincos_small(float x, float &sn, float &cs) {
  if (LINK_HW_TRIG) { sincos_hw(x, sn, cs); return; }
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.5703125f, x);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.54978995489188216e-8f, r);
  const int q = (int)k;
  const float r2 = r * r;
  float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  ps = fmaf(ps * r2, r, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  pc = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const float s0 = (q & 1) ? pc : ps;
  const float c0 = (q & 1) ? ps : pc;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}

// Hardware trig (v_sin_f32 / v_cos_f32 take the argument in revolutions): reduction by whole turns -- three-term
// Cody-Waite 2*pi = 6.28125 + 0x1.fb4p-10 + 0x1.4442d2p-22, exact products under fma for |k| < 2^13 -- then one
// multiply by 1/(2 pi) and the two transcendental ops: 8 instructions instead of ~25.  Not the libm rounding:
// max abs error ~3e-7 for |theta| < 3e4 (tools/sintest2.hip), three orders below the 1e-4 relative bar of the
// path; only the dense-cell kernels may use it (-DLINK_HW_TRIG=1), same range contract as sincos_small.
#ifndef LINK_HW_TRIG_REV
#define LINK_HW_TRIG_REV 1  /* round 5: the reduction is done in REVOLUTIONS, where the hardware wants its argument: k = rint(x / 2pi),
                               f = fma(x, hi, -k) (the product is exact inside the fma, so this is x * hi - k rounded once: |f| <= 1/2,
                               error <= 2^-26 turns), f += x * lo with hi + lo = 1 / 2pi to 2^-52 -- 4 instructions in front of
                               v_sin / v_cos instead of 6, the same 3e-8-turn accuracy for |x| < 2^15 (0: the round-3 form) */
#endif
__device__ __forceinline__ void sincos_hw(float x, float &sn, float &cs) {
#if LINK_HW_TRIG_REV
  const float k = rintf(x * 0x1.45f306p-3f);
  float f = fmaf(x, 0x1.45f306p-3f, -k);
  f = fmaf(x, 0x1.b9391p-28f, f);
  sn = __builtin_amdgcn_sinf(f);
  cs = __builtin_amdgcn_cosf(f);
#else
  const float k = rintf(x * 0.15915494309189535f);
  float r = fmaf(-k, 6.28125f, x);
  r = fmaf(-k, 0x1.fb4p-10f, r);
  r = fmaf(-k, 0x1.4442d2p-22f, r);
  const float t = r * 0.15915494309189535f;
  sn = __builtin_amdgcn_sinf(t);
  cs = __builtin_amdgcn_cosf(t);
#endif
}

// cos_x (linkunet.py:164-176) carries a third, LINEAR channel group fin * theta: whatever error fin has is multiplied by theta,
// which is a few hundred to a few thousand radians on LiDAR frames.  The fp16 hi | lo split of the pre_mix contraction
// represents its operands to 2^-22: harmless for cos / sin (|modulation| <= 1), but 5-8 x the fp32 reference's own conditioning
// error once it is scaled by theta (S-kitti stage 1, theta up to 744 / 1489: 1.1e-4 / 1.6e-4 of max|out| against 2.3e-5 / 2.9e-5
// for the reference evaluated in fp32, both measured against a float64 evaluation: tools/lidar_core_parity.py, round 5).  So the
// fused kernels take the exact fp32 matrix instruction for cos_x (0: the split everywhere, the round-3/4 behaviour).
#ifndef LINK_COSX_EXACT
#define LINK_COSX_EXACT 1
#endif
// The tile kernels (tile_common.h: dc_stage_weights / dc_premix_tile) take the exact contraction from an fp32 image of W in LDS;
// LINK_TILES_EXACT_ALL 1 makes them do so for every operator (A/B switch, docs/experiments.md 5e).
#ifndef LINK_TILES_EXACT_ALL
#define LINK_TILES_EXACT_ALL 0
#endif
#define LINK_TILE_EXACT(OP) ((LINK_COSX_EXACT && (OP) == LINK_OP_COSX) || LINK_TILES_EXACT_ALL)

template <int LPR>
__device__ __forceinline__ float grp_sum(float v) {
  // butterfly over the LPR lanes of a group; steps <= 8 stay inside a 16-lane DPP row
  if (LPR >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if (LPR >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if (LPR >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if (LPR >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// value held by the partner lane (li ^ LPR/2) of the same group: the lane owning channel ch +- C/2
template <int LPR>
__device__ __forceinline__ float partner(float v) {
  if (LPR == 16)   // rotate the 16-lane DPP row by 8: swaps its halves, one VALU op
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return __shfl_xor(v, LPR / 2, 64);
}

// Internal op codes of the modulate kernel beyond LINK_OP_*: the BACKWARD forms.  d(new)/d(v) of the
// de-modulation is [cos, sin] (cos), [cos, -sin] (sin), [cos, sin, 1] (cos_x), so the gradient of the
// block table is the same segmented block sum with these factors applied to grad(new).
#define LINK_OPI_SIN_BWD 3
#define LINK_OPI_COSX_BWD 4
template <int OP>
struct op_parts { static constexpr int value = (OP == LINK_OP_COSX || OP == LINK_OPI_COSX_BWD) ? 3 : 2; };

template <int OP>
__device__ __forceinline__ void mod_accum(float &a0, float &a1, float &a2, float f, float sn, float cs, float th) {
  if (OP == LINK_OP_SIN) { a0 += f * sn; a1 += f * cs; }
  else if (OP == LINK_OPI_SIN_BWD) { a0 += f * cs; a1 -= f * sn; }
  else { a0 += f * cs; a1 += f * sn; }
  if (OP == LINK_OP_COSX) a2 += link_mul_rn(f, th);        // a rounded product, as the reference sums it (common.h)
  if (OP == LINK_OPI_COSX_BWD) a2 += f;
}

}  // namespace link
