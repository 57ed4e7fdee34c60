// link_amd/csrc/elk_lean_dispatch.h -- launch + template dispatch of the lean-form kernels (elk_lean_impl.h) for the I/O type
// of the including translation unit (DC_IO / DC_IO_NS).  Defines DC_IO_NS::run_lean, which the C ABI in elk_lean.hip calls
// after validating the arguments.
namespace DC_IO_NS {
using namespace link;

template <int C, int OP, int NB>
static void launch_lean_a(const lean_args &a, int64_t n_prev, hipStream_t st) {
  const int lds = dc_wimg<C>::W_BYTES;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lean_insert_premix<C, OP, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  // the frame's workgroups, then (indexed steps) 256 item slots of the previous frame per workgroup to clean its counters
  const int64_t wgs = (int64_t)a.nwg + (a.build ? ((int64_t)LEAN_SEGS * a.idx_cap_prev + 255) / 256 : 0);
  if (wgs < 1) return;
  hipLaunchKernelGGL((k_lean_insert_premix<C, OP, NB>), dim3((unsigned)wgs), dim3(256), lds, st, a);
}

// which form of launch 1: the channel-split form (16 voxels per workgroup) where a frame is too small to fill the chip with
// tile-per-wave workgroups; link_elk_desc_t::flags can force either (LINK_ELK_LEAN_CS / LINK_ELK_LEAN_NO_CS)
static bool lean_use_cs(const link_elk_desc_t &d, int64_t n) {
  if ((d.c != 64 && d.c != 128) || (d.flags & LINK_ELK_LEAN_NO_CS)) return false;
  if (d.flags & LINK_ELK_LEAN_CS) return true;
  return d.c == 64 && n <= 4096;
}

template <int C, int OP>
static void lean_nb(const link_elk_desc_t &d, const lean_args &a, int64_t n_prev, hipStream_t st) {
  constexpr int T = C / 16;
  if constexpr (C == 64 || C == 128) {
    if (lean_use_cs(d, a.n)) {
      const int64_t wgs = (int64_t)a.nwg + (a.build ? ((int64_t)LEAN_SEGS * a.idx_cap_prev + 255) / 256 : 0);
      if (wgs >= 1) hipLaunchKernelGGL((k_lean_insert_premix_cs<C, OP>), dim3((unsigned)wgs), dim3(256), 0, st, a);
      return;
    }
  }
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (T >= 2 && nb == T / 2) return launch_lean_a<C, OP, (T >= 2 ? T / 2 : 1)>(a, n_prev, st);
  if (T >= 4 && nb == T / 4) return launch_lean_a<C, OP, (T >= 4 ? T / 4 : 1)>(a, n_prev, st);
  return launch_lean_a<C, OP, T>(a, n_prev, st);
}

// a wave per item slot (16 lists x idx_cap entries), at most 4096 workgroups (then a wave strides through its list); a multiple
// of 4 workgroups so that the stride is a multiple of the 16 lists
static unsigned lean_item_wgs(int idx_cap) {
  int64_t w = ((int64_t)LEAN_SEGS * idx_cap + LEAN_IW - 1) / LEAN_IW;
  if (w > 4096) w = 4096;
  if (w < 4) w = 4;
  return (unsigned)((w + 3) & ~(int64_t)3);
}

// The form without the scratch matrix X (launch 1 = slot insert alone, pre_mix inside launch 2; C <= 64) is selected by
// link_elk_desc_t::flags only (LINK_ELK_LEAN_PM).  Measured on the LiDAR stage frames (tools/lean_pm_ab.sh, one box): with the
// lists reused it is 4-5 us faster on the cos frames (28.9 against 33.8 us at 31k voxels, C = 64), rebuilt it ties there (its
// insert-only launch costs what the X round trip saves), and on the cos_x C = 64 frames it loses 13-30 us: every chunk of ~5
// voxels pays a whole 16-voxel tile of matrix instructions and W reads at two waves per SIMD.
static bool lean_use_pm(const link_elk_desc_t &d, int64_t n) {
  if (d.c > 64 || (d.flags & LINK_ELK_LEAN_NO_PM)) return false;
  if (d.flags & LINK_ELK_LEAN_PM) return true;
  return false;
}

template <int C, int OP, int NB>
static void launch_lean_pm(const lean_args &a, hipStream_t st) {
  const int lds = dc_wimg<C>::W_BYTES;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lean_sums_pm<C, OP, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k_lean_sums_pm<C, OP, NB>), dim3(lean_item_wgs(a.idx_cap)), dim3(256), lds, st, a);
}

template <int C, int OP>
static void lean_pm_nb(const link_elk_desc_t &d, const lean_args &a, hipStream_t st) {
  constexpr int T = C / 16;
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (T >= 2 && nb == T / 2) return launch_lean_pm<C, OP, (T >= 2 ? T / 2 : 1)>(a, st);
  if (T >= 4 && nb == T / 4) return launch_lean_pm<C, OP, (T >= 4 ? T / 4 : 1)>(a, st);
  return launch_lean_pm<C, OP, T>(a, st);
}

template <int C, int OP>
static int lean_c_op(const link_elk_desc_t &d, const lean_args &a, int64_t n_prev, hipStream_t st) {
  constexpr int P = op_parts<OP>::value;
  bool pm = false;
  if constexpr (C <= 64) pm = lean_use_pm(d, a.n);
  if (pm) {
    if constexpr (C <= 64) {
      if (a.build) {
        const int64_t wgs = (int64_t)a.nwg + ((int64_t)LEAN_SEGS * a.idx_cap_prev + 255) / 256;
        if (wgs >= 1) hipLaunchKernelGGL(k_lean_insert, dim3((unsigned)wgs), dim3(256), 0, st, a);
      }
      lean_pm_nb<C, OP>(d, a, st);
    }
  } else {
    lean_nb<C, OP>(d, a, n_prev, st);
    hipLaunchKernelGGL((k_lean_sums<C, P>), dim3(lean_item_wgs(a.idx_cap)), dim3(64 * LEAN_IW), 0, st, a);
  }
  if (d.r == 2) hipLaunchKernelGGL((k_lean_gather<C, OP, 2>), dim3(lean_item_wgs(a.idx_cap)), dim3(64 * LEAN_IW), 0, st, a);
  else hipLaunchKernelGGL((k_lean_gather<C, OP, 3>), dim3(lean_item_wgs(a.idx_cap)), dim3(64 * LEAN_IW), 0, st, a);
  return check_launch("link_elk_core_lean_forward");
}

template <int C>
static int lean_c(const link_elk_desc_t &d, const lean_args &a, int64_t n_prev, hipStream_t st) {
  switch (d.op) {
    case LINK_OP_COS: return lean_c_op<C, LINK_OP_COS>(d, a, n_prev, st);
    case LINK_OP_SIN: return lean_c_op<C, LINK_OP_SIN>(d, a, n_prev, st);
    default: return lean_c_op<C, LINK_OP_COSX>(d, a, n_prev, st);
  }
}

int run_lean(const link_lean_buffers_t &b, const link_grid_t &g, const link_elk_desc_t &d, int64_t n, int64_t n_prev, int build,
             hipStream_t st) {
  lean_args a;
  a.feats = b.feats; a.coords = reinterpret_cast<const int4 *>(b.coords);
  a.w_pre = b.w_pre; a.pre_ln_w = b.pre_ln_w; a.pre_ln_b = b.pre_ln_b; a.w_pos = b.w_pos; a.alpha = b.alpha;
  a.ln_w = b.ln_w; a.ln_b = b.ln_b;
  a.g = g; a.cg = d.cg; a.coord_div = d.coord_div; a.eps = d.eps;
  a.n = (int)n; a.k = b.k; a.kch = (b.k + LEAN_CH - 1) / LEAN_CH; a.build = build;
  // a list receives the items of every 16th workgroup of launch 1, at most one per voxel of the workgroup (64, or 16 in the
  // channel-split form); the previous frame's bound covers either form
  const bool pm = d.c <= 64 && lean_use_pm(d, n);
  const int vpw = pm ? 256 : (lean_use_cs(d, n) ? 16 : 64);      // voxels per workgroup of launch 1
  a.nwg = (int)((n + vpw - 1) / vpw);
  // (the insert-only launch appends per WAVE of 64 voxels, list = wave % 16)
  a.idx_cap = pm ? (int)(((n + 63) / 64 + LEAN_SEGS - 1) / LEAN_SEGS * 64) : (int)(((int64_t)a.nwg + LEAN_SEGS - 1) / LEAN_SEGS * vpw);
  const int parts = d.op == LINK_OP_COSX ? 3 : 2;
  a.xl_stride = pm ? d.c : parts * d.c;
  a.xl_off = pm ? 0 : 2 * d.c;
  a.idx_cap_prev = (int)(((n_prev + 63) / 64 + LEAN_SEGS - 1) / LEAN_SEGS * 64 + 16);
  a.seg_cap = b.seg_cap; a.cshift = b.cnt_shift;
  a.cnt = b.cnt; a.cnt_prev = b.cnt_prev; a.list = b.list; a.occ = b.occ; a.occ_prev = b.occ_prev;
  a.ctrl = b.ctrl; a.ctrl_prev = b.ctrl_prev; a.rec2 = reinterpret_cast<int4 *>(b.rec2);
  a.X = b.X; a.S = b.S; a.hdr = b.hdr; a.out = b.out;
  switch (d.c) {
    case 16: return lean_c<16>(d, a, n_prev, st);
    case 32: return lean_c<32>(d, a, n_prev, st);
    case 64: return lean_c<64>(d, a, n_prev, st);
    default: return lean_c<128>(d, a, n_prev, st);
  }
}

}  // namespace DC_IO_NS
