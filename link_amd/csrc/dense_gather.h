// link_amd/csrc/dense_gather.h -- geometry + LDS reader of the dense-cell box-sum kernels (k_dc_gather in
// dense.hip, k_dc_gather_demod in dense_fused.hip).
#pragma once
#include "dense_common.h"

namespace link {

template <int C, int P, int R>
struct dc_gather_cfg {
  static constexpr int LPR = C / 4;                  // C in {16,32,64,128}: power of two
  static constexpr int NG = 256 / LPR;               // groups per workgroup
  static constexpr int TY = NG >= 64 ? 8 : (NG >= 16 ? 4 : 2);
  static constexpr int TX = NG / TY;
  static constexpr int HLO = (R == 3) ? 1 : 0;
  static constexpr int HX = TX + R - 1, HY = TY + R - 1;
  static constexpr int NCOL = HX * HY;
  static constexpr int RP = P * C / 4;               // 16-byte pieces per row
  static constexpr int NPC = NCOL * RP;              // pieces per plane
  static constexpr int PASSES = (NPC + 255) / 256;
  static constexpr int EPW = 64;                     // count entries per wave-instruction (linear image)
  static constexpr int NI = PASSES + 1;              // DMA instructions per plane per wave
  static constexpr int PLANE_BYTES = PASSES * 256 * 16;
  static constexpr int CNT_BYTES = 4 * 256;
  static constexpr int BUF_BYTES = PLANE_BYTES + CNT_BYTES;
  static constexpr int LDS_BYTES = 3 * BUF_BYTES;
  static_assert(NCOL <= 256, "count image: 4 waves x 64 entries");
  static_assert(NCOL * P * C * 4 + 3 * C * 4 < 65536, "ds_read immediate offsets are 16 bit");
};


// LDS reads of the box sum are issued through inline asm: a plain C++ LDS load makes hipcc wait vmcnt(0)
// for EVERY LDS-DMA in flight (it cannot tell the ring slots apart), which would drain the two planes
// being prefetched.  One statement = the R*P row reads + R count reads of one x-offset + their lgkmcnt(0),
// outputs early-clobber: data is valid when the statement ends (cdna_hip_programming.md 5.7, form i).
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int C, int P, int R, int DX>
__device__ __forceinline__ void dc_read_dx(uint32_t ra, uint32_t ca, float4 (&cur)[P], float &cc) {
  using K = dc_gather_cfg<C, P, R>;
  constexpr int RB = P * C * 4;
#define O_(dy, pp) "i"(((DX * K::HY + (dy)) * RB) + (pp) * C * 4)
#define Q_(dy) "i"((DX * K::HY + (dy)) * 4)
  v4f_t v0, v1, v2, v3, v4, v5, v6, v7, v8;
  int n0, n1, n2 = 0;
  if constexpr (P == 2 && R == 3) {
    asm volatile(
        "ds_read_b128 %0, %9 offset:%c11\n\tds_read_b128 %1, %9 offset:%c12\n\t"
        "ds_read_b128 %2, %9 offset:%c13\n\tds_read_b128 %3, %9 offset:%c14\n\t"
        "ds_read_b128 %4, %9 offset:%c15\n\tds_read_b128 %5, %9 offset:%c16\n\t"
        "ds_read_b32 %6, %10 offset:%c17\n\tds_read_b32 %7, %10 offset:%c18\n\tds_read_b32 %8, %10 offset:%c19\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(n0), "=&v"(n1), "=&v"(n2)
        : "v"(ra), "v"(ca), O_(0, 0), O_(0, 1), O_(1, 0), O_(1, 1), O_(2, 0), O_(2, 1), Q_(0), Q_(1), Q_(2)
        : "memory");
    const v4f_t a = (v0 + v2) + v4, b = (v1 + v3) + v5;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
  } else if constexpr (P == 3 && R == 3) {
    asm volatile(
        "ds_read_b128 %0, %12 offset:%c14\n\tds_read_b128 %1, %12 offset:%c15\n\tds_read_b128 %2, %12 offset:%c16\n\t"
        "ds_read_b128 %3, %12 offset:%c17\n\tds_read_b128 %4, %12 offset:%c18\n\tds_read_b128 %5, %12 offset:%c19\n\t"
        "ds_read_b128 %6, %12 offset:%c20\n\tds_read_b128 %7, %12 offset:%c21\n\tds_read_b128 %8, %12 offset:%c22\n\t"
        "ds_read_b32 %9, %13 offset:%c23\n\tds_read_b32 %10, %13 offset:%c24\n\tds_read_b32 %11, %13 offset:%c25\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7), "=&v"(v8),
          "=&v"(n0), "=&v"(n1), "=&v"(n2)
        : "v"(ra), "v"(ca), O_(0, 0), O_(0, 1), O_(0, 2), O_(1, 0), O_(1, 1), O_(1, 2), O_(2, 0), O_(2, 1), O_(2, 2),
          Q_(0), Q_(1), Q_(2)
        : "memory");
    const v4f_t a = (v0 + v3) + v6, b = (v1 + v4) + v7, d = (v2 + v5) + v8;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
    cur[P - 1].x += d.x; cur[P - 1].y += d.y; cur[P - 1].z += d.z; cur[P - 1].w += d.w;
  } else if constexpr (P == 2 && R == 2) {
    asm volatile(
        "ds_read_b128 %0, %6 offset:%c8\n\tds_read_b128 %1, %6 offset:%c9\n\t"
        "ds_read_b128 %2, %6 offset:%c10\n\tds_read_b128 %3, %6 offset:%c11\n\t"
        "ds_read_b32 %4, %7 offset:%c12\n\tds_read_b32 %5, %7 offset:%c13\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(n0), "=&v"(n1)
        : "v"(ra), "v"(ca), O_(0, 0), O_(0, 1), O_(1, 0), O_(1, 1), Q_(0), Q_(1)
        : "memory");
    const v4f_t a = v0 + v2, b = v1 + v3;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
  } else {
    static_assert(P == 3 || P == 2, "parts");
    asm volatile(
        "ds_read_b128 %0, %8 offset:%c10\n\tds_read_b128 %1, %8 offset:%c11\n\tds_read_b128 %2, %8 offset:%c12\n\t"
        "ds_read_b128 %3, %8 offset:%c13\n\tds_read_b128 %4, %8 offset:%c14\n\tds_read_b128 %5, %8 offset:%c15\n\t"
        "ds_read_b32 %6, %9 offset:%c16\n\tds_read_b32 %7, %9 offset:%c17\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(n0), "=&v"(n1)
        : "v"(ra), "v"(ca), O_(0, 0), O_(0, 1), O_(0, 2), O_(1, 0), O_(1, 1), O_(1, 2), Q_(0), Q_(1)
        : "memory");
    const v4f_t a = v0 + v3, b = v1 + v4, d = v2 + v5;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
    cur[P - 1].x += d.x; cur[P - 1].y += d.y; cur[P - 1].z += d.z; cur[P - 1].w += d.w;
  }
  cc += (float)n0 + (float)n1 + (float)n2;
#undef O_
#undef Q_
}

// Pipelined form of the three dc_read_dx calls of one plane (P = 2, R = 3: the cfg2 / detection case): the rows of x-offset
// 0 and 1 are requested together, offset 2 (into offset 0's registers) while offset 0 is being added, so the plane costs
// ONE LDS latency plus the transfers instead of three dependent round trips.  Guide section 5.7 form (ii): every statement
// that waits names the registers that become valid there as "+v", so nothing consumes them earlier; the sums are formed in
// the order dc_read_dx forms them (bitwise identical results).  Needs 27 more registers than the one-offset-at-a-time form.
template <int C>
__device__ __forceinline__ void dc_read_plane_p2r3(uint32_t ra, uint32_t ca, float4 (&cur)[2], float &cc) {
  using K = dc_gather_cfg<C, 2, 3>;
  constexpr int RB = 2 * C * 4;
#define O_(dx, dy, pp) "i"((((dx) * K::HY + (dy)) * RB) + (pp) * C * 4)
#define Q_(dx, dy) "i"(((dx) * K::HY + (dy)) * 4)
  v4f_t a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5;
  int m0, m1, m2, n0, n1, n2;
  asm volatile(
      "ds_read_b128 %0, %18 offset:%c20\n\tds_read_b128 %1, %18 offset:%c21\n\t"
      "ds_read_b128 %2, %18 offset:%c22\n\tds_read_b128 %3, %18 offset:%c23\n\t"
      "ds_read_b128 %4, %18 offset:%c24\n\tds_read_b128 %5, %18 offset:%c25\n\t"
      "ds_read_b32 %6, %19 offset:%c26\n\tds_read_b32 %7, %19 offset:%c27\n\tds_read_b32 %8, %19 offset:%c28\n\t"
      "ds_read_b128 %9, %18 offset:%c29\n\tds_read_b128 %10, %18 offset:%c30\n\t"
      "ds_read_b128 %11, %18 offset:%c31\n\tds_read_b128 %12, %18 offset:%c32\n\t"
      "ds_read_b128 %13, %18 offset:%c33\n\tds_read_b128 %14, %18 offset:%c34\n\t"
      "ds_read_b32 %15, %19 offset:%c35\n\tds_read_b32 %16, %19 offset:%c36\n\tds_read_b32 %17, %19 offset:%c37\n\t"
      "s_waitcnt lgkmcnt(9)"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(m0), "=&v"(m1), "=&v"(m2),
        "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(n0), "=&v"(n1), "=&v"(n2)
      : "v"(ra), "v"(ca), O_(0, 0, 0), O_(0, 0, 1), O_(0, 1, 0), O_(0, 1, 1), O_(0, 2, 0), O_(0, 2, 1), Q_(0, 0), Q_(0, 1), Q_(0, 2),
        O_(1, 0, 0), O_(1, 0, 1), O_(1, 1, 0), O_(1, 1, 1), O_(1, 2, 0), O_(1, 2, 1), Q_(1, 0), Q_(1, 1), Q_(1, 2)
      : "memory");
  {
    const v4f_t a = (a0 + a2) + a4, b = (a1 + a3) + a5;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
    cc += (float)m0 + (float)m1 + (float)m2;
  }
  asm volatile(
      "ds_read_b128 %0, %18 offset:%c20\n\tds_read_b128 %1, %18 offset:%c21\n\t"
      "ds_read_b128 %2, %18 offset:%c22\n\tds_read_b128 %3, %18 offset:%c23\n\t"
      "ds_read_b128 %4, %18 offset:%c24\n\tds_read_b128 %5, %18 offset:%c25\n\t"
      "ds_read_b32 %6, %19 offset:%c26\n\tds_read_b32 %7, %19 offset:%c27\n\tds_read_b32 %8, %19 offset:%c28\n\t"
      "s_waitcnt lgkmcnt(9)"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(m0), "=&v"(m1), "=&v"(m2),
        "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(n0), "+v"(n1), "+v"(n2)
      : "v"(ra), "v"(ca), O_(2, 0, 0), O_(2, 0, 1), O_(2, 1, 0), O_(2, 1, 1), O_(2, 2, 0), O_(2, 2, 1), Q_(2, 0), Q_(2, 1), Q_(2, 2)
      : "memory");
  {
    const v4f_t a = (b0 + b2) + b4, b = (b1 + b3) + b5;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
    cc += (float)n0 + (float)n1 + (float)n2;
  }
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(m0), "+v"(m1), "+v"(m2)
               :
               : "memory");
  {
    const v4f_t a = (a0 + a2) + a4, b = (a1 + a3) + a5;
    cur[0].x += a.x; cur[0].y += a.y; cur[0].z += a.z; cur[0].w += a.w;
    cur[1].x += b.x; cur[1].y += b.y; cur[1].z += b.z; cur[1].w += b.w;
    cc += (float)m0 + (float)m1 + (float)m2;
  }
#undef O_
#undef Q_
}


}  // namespace link
