// link_amd/csrc/elk_lean.hip -- lean form of R_core with the index rebuilt every call (elk_lean_impl.h): C ABI; the kernels for
// fp32 feature rows (fp16 / bf16 rows at the kernel boundary: elk_lean_f16.hip, elk_lean_bf16.hip).
#define DC_IO 0
#define DC_IO_NS elkl_f32
#include "elk_lean_impl.h"
#include "elk_lean_dispatch.h"

using namespace link;

#define ELKL_DECL(NS)                                                                                                       \
  namespace NS {                                                                                                            \
  int run_lean(const link_lean_buffers_t &, const link_grid_t &, const link_elk_desc_t &, int64_t, int64_t, int, hipStream_t); \
  }
ELKL_DECL(elkl_f16)
ELKL_DECL(elkl_bf16)
#undef ELKL_DECL

extern "C" int link_elk_core_lean_forward(const link_lean_buffers_t *b, const link_grid_t *grid, const link_elk_desc_t *d,
                                          int64_t n, int64_t n_prev, int32_t build_index, void *stream) {
  if (!b || !grid || !d || n < 0 || n_prev < 0 || n >= (1LL << 31) || n_prev >= (1LL << 31)) return LINK_ERR_ARG;
  if (!dc_width_ok(d->c) || (d->r != 2 && d->r != 3) || d->cg < 1 || d->c % d->cg != 0) return LINK_ERR_ARG;
  if (d->op != LINK_OP_COS && d->op != LINK_OP_SIN && d->op != LINK_OP_COSX) return LINK_ERR_ARG;
  if (b->io_dtype < LINK_IO_F32 || b->io_dtype > LINK_IO_BF16 || b->k < 1 || b->k > elkl_f32::LEAN_KMAX || b->cnt_shift < 0 || b->cnt_shift > 5)
    return LINK_ERR_ARG;
  int64_t v = 1;
  for (int ax = 0; ax < 4; ax++) {
    if (grid->dim[ax] <= 0) return LINK_ERR_ARG;
    v *= grid->dim[ax];
    if (v >= (1LL << 27)) return LINK_ERR_ARG;              // an item is cell * 16 + chunk in 31 bits
  }
  if (grid->s <= 0 || b->seg_cap < (n / 64 + 16) / 16 * 64 + 64) return LINK_ERR_ARG;
  if (!b->cnt || !b->list || !b->rec2 || !b->occ || !b->ctrl || !b->X || !b->S || !b->hdr) return LINK_ERR_ARG;
  if (build_index && (!b->cnt_prev || !b->occ_prev || !b->ctrl_prev)) return LINK_ERR_ARG;
  if (n > 0 && (!b->feats || !b->coords || !b->out || !b->w_pre || !b->pre_ln_w || !b->pre_ln_b || !b->w_pos || !b->ln_w || !b->ln_b))
    return LINK_ERR_ARG;
  if (n * d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;      // 32-bit byte offsets into the feature rows
  if (n == 0 && !build_index) return LINK_OK;
  hipStream_t st = S(stream);
  const int build = build_index ? 1 : 0;
  switch (b->io_dtype) {
    case LINK_IO_F16: return elkl_f16::run_lean(*b, *grid, *d, n, n_prev, build, st);
    case LINK_IO_BF16: return elkl_bf16::run_lean(*b, *grid, *d, n, n_prev, build, st);
    default: return elkl_f32::run_lean(*b, *grid, *d, n, n_prev, build, st);
  }
}
