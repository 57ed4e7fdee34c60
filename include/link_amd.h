/*
 * include/link_amd.h -- C ABI of the MI355X-native LinK hot path (liblink_amd.so).
 *
 * This is the drop-in boundary: plain pointers + sizes + a HIP stream, no torch types.  All
 * pointers are DEVICE pointers unless a parameter says "host".  Every entry point
 *   - launches asynchronously on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *   - never allocates, never synchronises, never throws; caller owns every buffer,
 *   - returns LINK_OK (0) or a negative LINK_ERR_* code (argument/launch errors only; data-dependent
 *     conditions are reported through the device-side status word documented per call).
 *
 * Section A replaces, one for one, the functions the reference registers in its pybind11 module
 * `torchsparse.backend` (/root/reference/segmentation/torchsparse-u/torchsparse/backend/
 * pybind_cuda.cpp:18-39) that lie on the LinK path.  Section B is the fused form of the Python layer
 * above them (segmentation/core/models/utils.py:44-84 voxel_to_aux / aux_to_voxel,
 * detection/det3d/models/utils/ts_elk.py:68-107).  Section C is the fused R_core of
 * ELKBlock.forward / TSELKBlock.forward_ (segmentation/core/models/semantic_kitti/linkunet.py:124-185,
 * detection/det3d/models/utils/ts_elk.py:144-230).
 *
 * Feature dtype: fp32 (the shipped reference runs this path in fp32; voxelnet.py:55 has autocast
 * commented out).  Integer results are bit-exact with the reference; fp32 within 1e-4 rel.
 */
#ifndef LINK_AMD_H_
#define LINK_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LINK_OK 0
#define LINK_ERR_ARG (-1)       /* null pointer / negative size / unsupported width */
#define LINK_ERR_LAUNCH (-2)    /* hipGetLastError() != hipSuccess after a launch */
#define LINK_ERR_WORKSPACE (-3) /* caller-provided workspace too small */
#define LINK_BATCH_TIMEOUT (-4) /* link_dc_batch_status: a bounded wait inside a persistent batch kernel gave up (section H) */

/* Version of this ABI (bumped on any signature change). */
int link_abi_version(void);
/* sizeof of the structs crossing this boundary (0 link_grid_t, 1 link_elk_desc_t, 2 link_elk_buffers_t, 3 link_dc_grid_t,
 * 4 link_dc_tuning_t, 5 link_dc_buffers_t, 6 link_lean_buffers_t, 7 link_block_args_t; -1 otherwise): what a binding checks its own layout against. */
int32_t link_abi_struct_size(int32_t which);
/* Human-readable last HIP error string of the calling thread ("" if none). Host pointer. */
const char *link_last_error(void);

/* =============================================================================================
 * A. torchsparse.backend drop-ins
 * ============================================================================================= */

/* hash_cuda(idx[N,4] i32) -> i64[N]           backend/hash/hash_cuda.cu:10-23,67-73
 * 64-bit FNV-1a over the four 32-bit words, folded to 60 bits.  Bit-exact. */
int link_hash(const int32_t *coords, int64_t n, int64_t *out, void *stream);

/* kernel_hash_cuda(idx[N,4] i32, off[K,3] i32) -> i64[K,N] (k-major)
 *                                             backend/hash/hash_cuda.cu:27-55,75-84
 * Hash of (x+ox, y+oy, z+oz, b); batch taken from the row itself (CUDA semantics). Bit-exact. */
int link_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets, int64_t k,
                     int64_t *out, void *stream);

/* hash_query_cuda(q i64[n1], tgt i64[n], tgt_idx i64[n]) -> i64[n1]
 *                                             backend/others/query_cuda.cu:9-58,
 *                                             backend/hashmap/hashmap_cuda.cu:9-130
 * out[i] = tgt_idx[j]+1 for the FIRST j with tgt[j]==q[i], else 0 (the Python wrapper subtracts 1,
 * nn/functional/query.py:32).  The reference builds a 3-function cuckoo table with host-side
 * rehash retries and two device syncs per call; here: one open-addressing table in `workspace`
 * (linear probing, wait-free 64-bit CAS insert of the key, atomicMin on the index for first-wins),
 * no sync, no reserved key.  Workspace need not be initialised.  `link_hash_query_workspace_bytes(n)`. */
size_t link_hash_query_workspace_bytes(int64_t n_target);
int link_hash_query(const int64_t *query, int64_t n1, const int64_t *target,
                    const int64_t *target_idx, int64_t n, int64_t *out, void *workspace,
                    size_t workspace_bytes, void *stream);

/* count_cuda(idx i32[N], s) -> i32[s]         backend/others/count_cuda.cu:10-31
 * Histogram; idx<0 (or >= s) ignored.  `out` is zeroed by the call. */
int link_count(const int32_t *idx, int64_t n, int32_t *out, int64_t s, void *stream);

/* calc_ti_weights(coords fp32[P,4] (x,y,z,batch), idx_query i64[8,P], scale) -> fp32[8,P]
 *                                              torchsparse/nn/functional/devoxelize.py:10-48
 * Trilinear weights of a point's 8 corner voxels (corner k = 4 dx + 2 dy + dz), zero where idx_query is -1,
 * renormalised by (their sum + 1e-8).  One kernel, one thread per point. */
int link_ti_weights(const float *coords, const int64_t *idx_query, int64_t p, float scale, float *w, void *stream);

/* voxelize_forward_cuda(in fp[N,c], idx i32[N], counts i32[N1]) -> fp[N1,c]
 *                                             backend/voxelize/voxelize_cuda.cu:12-25,44-61
 * out[idx[i]] += in[i] / (float)counts[idx[i]].  `out` is zeroed by the call.  Generic form for an
 * arbitrary idx (fp32 atomics, like the reference); the indexed, deterministic form is B.2. */
int link_voxelize_forward(const float *in, const int32_t *idx, const int32_t *counts, int64_t n,
                          int64_t c, int64_t n1, float *out, void *stream);

/* voxelize_backward_cuda(top fp[N1,c], idx, counts, N) -> fp[N,c]
 *                                             backend/voxelize/voxelize_cuda.cu:28-42,63-80
 * bottom[i] = top[idx[i]] / counts[idx[i]]  (0 where idx[i] < 0). */
int link_voxelize_backward(const float *top, const int32_t *idx, const int32_t *counts, int64_t n,
                           int64_t c, float *bottom, void *stream);

/* devoxelize_forward_cuda(feat fp[n,c], ind i32[N,K], w fp[N,K], r) -> fp[N,c],  K = r^3
 *                                             backend/devoxelize/devoxelize_cuda.cu:11-34,63-81
 * out[i] = sum_{k<K, ind>=0} w[i,k]*feat[ind[i,k]], k ascending (deterministic). */
int link_devoxelize_forward(const float *feat, const int32_t *ind, const float *w, int64_t nq,
                            int64_t c, int64_t k, float *out, void *stream);

/* devoxelize_backward_cuda(top fp[N,c], ind, w, n, r) -> fp[n,c]
 *                                             backend/devoxelize/devoxelize_cuda.cu:37-59,85-101
 * bottom[ind[i,k]] += w[i,k]*top[i].  `bottom` is zeroed by the call (fp32 atomics). */
int link_devoxelize_backward(const float *top, const int32_t *ind, const float *w, int64_t nq,
                             int64_t n, int64_t c, int64_t k, float *bottom, void *stream);

/* =============================================================================================
 * B. Block index + indexed aggregation (fused voxel_to_aux / aux_to_voxel)
 * ============================================================================================= */

/* Dense block grid covering the frame.  Block coordinate of a voxel = (floor(x/s), floor(y/s),
 * floor(z/s), b) (utils.py:45); `lo` is the smallest block coordinate per axis, `dim` the extent in
 * blocks.  Cells are linearised x-major: cell = (((bx-lo0)*dim1 + (by-lo1))*dim2 + (bz-lo2))*dim3 +
 * (bb-lo3), which is exactly the signed lexicographic row order torch.unique(dim=0) produces
 * (utils.py:47), so the rank of an occupied cell among occupied cells IS the reference's block id. */
typedef struct {
  int32_t s;       /* block edge in voxel-coordinate units (the `s` of voxel_to_aux) */
  int32_t lo[4];   /* block-coordinate lower bound (x,y,z,b) */
  int32_t dim[4];  /* block-grid extent (x,y,z,b); product = number of cells V (< 2^30) */
} link_grid_t;

/* Host helper: grid from inclusive voxel-coordinate bounds lo/hi[4] (x,y,z,b).  Returns V or -1. */
int64_t link_grid_from_bounds(const int32_t lo[4], const int32_t hi[4], int32_t s, link_grid_t *g);

/* Device bounding box of coords (for callers that do not know the bounds): writes
 * bbox[8] = {min x,y,z,b, max x,y,z,b}.  One pass, per-workgroup reduction + 8 atomics per
 * workgroup.  `bbox` must be initialised by the caller to {INT_MAX x4, INT_MIN x4}. */
int link_coords_bbox(const int32_t *coords, int64_t n, int32_t *bbox, void *stream);

/* Index header written by link_index_build (device int32[LINK_HDR_WORDS]). */
#define LINK_HDR_M 0        /* number of occupied blocks M */
#define LINK_HDR_STATUS 1   /* 0 ok; bit0: a voxel lay outside the grid (index invalid) */
#define LINK_HDR_NVALID 2   /* number of voxels indexed (= N when status == 0) */
#define LINK_HDR_WORDS 8

/* Scratch sizes for link_index_build (bytes). `cell_counts` (V u32) MUST be all-zero on entry and is
 * all-zero again on exit (the scan clears what it consumes), so one allocation zeroed once serves
 * every later call.  `scratch` needs no initialisation. */
size_t link_index_scratch_bytes(int64_t n, int64_t v);

/* Build the block index of a frame: replaces sphash + torch.unique + sphash + sphashquery + spcount
 * of voxel_to_aux (utils.py:45-51) by counting into the dense grid + one decoupled-look-back scan.
 * Outputs (caller-allocated; M <= N so N rows always suffice):
 *   cell_blk   i32[V]    block id + 1 of each cell, 0 = empty       (neighbour lookup table)
 *   vox_blk    i32[N]    block id of each voxel                      (== idx_query, utils.py:50)
 *   idx_query  i64[N]    same as int64 (may be NULL)                 (reference dtype)
 *   perm       i32[N]    voxel ids grouped by block, ascending voxel id inside a block
 *   vox_sorted i32[N,4]  (x, y, z, voxel id) of the voxel at each position of perm: the record the
 *                        fused kernels stream instead of chasing perm -> coords (may be NULL)
 *   pos_blk    i32[N]    block id of the voxel at each position of perm (may be NULL)
 *   blk_start  i32[N+1]  start of each block's segment in perm; blk_start[M] = N
 *   blk_coords i32[N,4]  block coordinates, rows [0,M) valid         (== small_x.C, utils.py:47)
 *   counts     i32[N]    voxels per block, rows [0,M) valid          (== spcount, utils.py:51)
 *   hdr        i32[8]    see LINK_HDR_*
 * Bit-exact with the reference for small_x.C / idx_query / counts. */
int link_index_build(const int32_t *coords, int64_t n, const link_grid_t *grid /* host */,
                     uint32_t *cell_counts, void *scratch, size_t scratch_bytes, int32_t *cell_blk,
                     int32_t *vox_blk, int64_t *idx_query, int32_t *perm, int32_t *vox_sorted,
                     int32_t *pos_blk, int32_t *blk_start, int32_t *blk_coords, int32_t *counts,
                     int32_t *hdr, void *stream);
/* link_index_build with the blocks numbered in FIRST-VOXEL order instead of cell order: block b is the b-th voxel, in id order,
 * that has the smallest id of its cell.  For callers that need a numbering, not the reference's (ElkCorePlan: block order never
 * leaves the arena): the scan runs over the N voxels instead of the V cells of the grid -- on LiDAR stage frames (1-2 % of the
 * cells occupied) the cell scan is 10-17 us of a 28 us index.  Same outputs and meanings as link_index_build (replaces
 * utils.py:44-58 up to the order of the rows of small_x.C), except: block order; no cell_counts (cell_pair takes its place);
 * the previous frame's entries of cell_blk are cleared through its block list (hdr[LINK_HDR_M] rows of blk_coords as the previous
 * call on these buffers left them), so cell_blk / hdr / blk_coords must be the ones the previous call wrote (or zero-filled).
 * cell_pair: u64[V], zero-filled once by the caller, zero again after every call.  n < 2^30.  Deterministic: voxel ids, not
 * insertion order, decide. */
int link_index_build_first(const int32_t *coords, int64_t n, const link_grid_t *grid /* host */, uint64_t *cell_pair, void *scratch,
                           size_t scratch_bytes, int32_t *cell_blk, int32_t *vox_blk, int64_t *idx_query, int32_t *perm,
                           int32_t *vox_sorted, int32_t *pos_blk, int32_t *blk_start, int32_t *blk_coords, int32_t *counts,
                           int32_t *hdr, void *stream);
/* The cell half of link_index_build alone (no voxel placement): sorted unique block coordinates blk_coords i32[.,4],
 * counts, cell table and hdr[LINK_HDR_M] of the rows `coords` -- rows outside the grid are dropped (status word).
 * Same scratch / cell_counts contract as link_index_build.  Used for the output-site set of site-creating
 * convolutions, whose candidate rows are mostly duplicates. */
int link_index_cells(const int32_t *coords, int64_t n, const link_grid_t *grid, uint32_t *cell_counts, void *scratch,
                     size_t scratch_bytes, int32_t *cell_blk, int32_t *blk_start, int32_t *blk_coords, int32_t *counts,
                     int32_t *hdr, void *stream);

/* Neighbour map nbr i32[M,K] (K = r^3, offsets in get_kernel_offsets(r) order, nn/utils/kernel.py:
 * 11-32: odd r x fastest, even r z fastest): replaces sphash(C, offsets) + sphash + sphashquery +
 * transpose of aux_to_voxel (utils.py:65-73).  -1 = absent.  `m` may be an upper bound (rows past
 * hdr[M] are not written); offsets are multiplied by `step` (1 for LinK blocks; the tensor stride when
 * the rows are strided voxel coordinates, as sparse-conv kernel maps use them, nn/functional/conv.py:
 * 105-107); transpose != 0 uses negated offsets (the adjoint relation, needed by the backward pass
 * when r is even).  `hdr` may be NULL (then all m rows are valid).  Bit-exact. */
int link_neighbor_map(const int32_t *blk_coords, const int32_t *cell_blk, const link_grid_t *grid,
                      const int32_t *hdr, int64_t m, int32_t r, int32_t step, int32_t transpose,
                      int32_t *nbr, void *stream);

/* Same for arbitrary (foreign) block rows: first scatter row ids into a cell table that the caller
 * zero-initialised (first row wins for duplicates), then look up.  Rows outside the grid -> status. */
int link_cell_table_build(const int32_t *rows, int64_t m, const link_grid_t *grid, int32_t *cell_blk,
                          int32_t *hdr, void *stream);
/* ... and back: zero the cells link_cell_table_build wrote for the same rows (m scattered stores), so that a table kept
 * between calls is all zero again without a memset over every cell of a sparse grid. */
int link_cell_table_clear(const int32_t *rows, int64_t m, const link_grid_t *grid, int32_t *cell_blk, void *stream);

/* Indexed block mean (spvoxelize forward, utils.py:52): out[b] = sum_{i in block b, ascending i}
 * in[i]/counts[b].  Deterministic, no atomics; rows [0,M) of out[.,c] written.  `m_cap` = rows
 * allocated in `out` (>= M). */
int link_block_mean(const float *in, const int32_t *perm, const int32_t *blk_start,
                    const int32_t *hdr, int64_t n, int64_t c, int64_t m_cap, float *out, void *stream);

/* Adjoint of link_block_mean (spvoxelize backward): bottom[i] = top[vox_blk[i]]/counts[vox_blk[i]]
 * is link_voxelize_backward (A). */

/* aux_to_voxel feature half (utils.py:75-82):
 *   new[m] = sum_k small_f[nbr[m,k]]*counts[nbr[m,k]] / sum_k counts[nbr[m,k]] ; out[i] = new[idx[i]].
 * small_f fp[M,c] block means, nbr i32[M,K], idx i64[N].  `new_feat` fp[M,c] and `denom` fp[M]
 * (sum of neighbour counts) are kept for the backward pass.  k ascending, deterministic. */
int link_aux_to_voxel_forward(const float *small_f, const int32_t *counts, const int32_t *nbr,
                              const int64_t *idx, int64_t n, int64_t m, int64_t c, int64_t k,
                              float *new_feat, float *denom, float *out, void *stream);

/* Backward of the above wrt small_f: given g_out fp[N,c]:
 *   g_new[m] = sum_{i: idx[i]==m} g_out[i]            (segmented via perm/blk_start: deterministic)
 *   g_small[j] = counts[j] * sum_{m: j in nbr(m)} g_new[m]/denom[m]   (via the transposed map nbr_t)
 * g_new fp[M,c] is scratch. */
int link_aux_to_voxel_backward(const float *g_out, const int32_t *perm, const int32_t *blk_start,
                               const int32_t *counts, const int32_t *nbr_t, const float *denom,
                               int64_t n, int64_t m, int64_t c, int64_t k, float *g_new,
                               float *g_small, void *stream);
/* aux_to_voxel forward on the dense block grid (same results as link_aux_to_voxel_forward, which takes an
 * explicit neighbour table): means -> sum table, r^3 neighbour sum by the fused path's z-sliding block
 * gather, float4 row gather to voxels.  Needs the index that produced `small_f`'s row order
 * (blk_coords, cell_blk, grid, hdr from link_index_build), w % 8 == 0 or w % 12 == 0, r <= 3
 * (LINK_ERR_ARG otherwise).  S scratch fp[(m+1)*(w+1)]; new_feat fp[m,w]; denom fp[m]; out fp[n,w]. */
int link_aux_to_voxel_forward_grid(const float *small_f, const int32_t *counts, const int32_t *blk_coords,
                                   const int32_t *cell_blk, const link_grid_t *grid /* host */,
                                   const int32_t *hdr, const int64_t *idx, int64_t n, int64_t m, int32_t w,
                                   int32_t r, float *S, float *new_feat, float *denom, float *out,
                                   void *stream);
/* The same result with the rows written by the gather kernel itself: out[perm[p]] = the row of block b for p in
 * [blk_start[b], blk_start[b + 1]) (perm / blk_start of link_index_build).  Two launches; neither the [M, W] table of
 * neighbour means (`new_feat`) nor the row gather exist.  denom f32[M] as above. */
int link_aux_to_voxel_forward_scatter(const float *small_f, const int32_t *counts, const int32_t *blk_coords,
                                      const int32_t *cell_blk, const link_grid_t *grid /* host */, const int32_t *hdr,
                                      const int32_t *blk_start, const int32_t *perm, int64_t n, int64_t m, int32_t w, int32_t r,
                                      float *S /* scratch f32[(M+1)*(W+1)] */, float *denom, float *out, void *stream);

/* =============================================================================================
 * C. Fused ELKBlock core (R_core of SURVEY.md section 8d)
 * ============================================================================================= */

#define LINK_OP_COS 0   /* linkunet.py:150-162 / ts_elk.py:166-177   X = [F cos, F sin]            */
#define LINK_OP_SIN 1   /* linkunet.py:136-148 / ts_elk.py:155-164   X = [F sin, F cos], minus sign */
#define LINK_OP_COSX 2  /* linkunet.py:164-176                       X = [F cos, F sin, F theta]    */

typedef struct {
  int32_t op;            /* LINK_OP_* */
  int32_t c;             /* channels C (inc) */
  int32_t cg;            /* theta channels: channel j uses theta[j % cg]  (C/groups; C for cos_x) */
  int32_t r;             /* neighbourhood edge in blocks */
  float coord_div;       /* theta input = float(coord) / coord_div (float division, as the reference
                            does it): 1.0 everywhere except the encoder cos_x variant, which feeds
                            coords / tensor_stride (linkencoder.py:165) */
  float eps;             /* LayerNorm eps (1e-6, linkunet.py:111,121) */
  int32_t flags;         /* general layout: kernel-path selection of THIS call (0 = defaults; nothing is process-global).  The
                            alternative paths are what the default ones are tested against (tests/test_gpu_split_paths.py) */
} link_elk_desc_t;
#define LINK_ELK_LANE_CHANNEL 1    /* lane = channel kernels instead of the lane-group kernels */
#define LINK_ELK_NO_PAIR 2         /* no voxel-pair sharing of sincos between channels j and j + C/2 */
#define LINK_ELK_FUSED_GATHER 4    /* one fused gather + de-modulate kernel instead of block gather + per-voxel kernel */
#define LINK_ELK_NO_DENSE_GRID 8   /* block gather: column-walking form even on mostly occupied grids */
#define LINK_ELK_LEAN_CS 32        /* link_elk_core_lean_forward: channel-split form of launch 1 whenever C is 64 / 128 */
#define LINK_ELK_LEAN_NO_CS 64     /* ... never (default: by frame size) */
#define LINK_ELK_LEAN_PM 128       /* link_elk_core_lean_forward: the form without the scratch matrix X (pre_mix inside launch 2; C <= 64) */
#define LINK_ELK_LEAN_NO_PM 256    /* ... never (default: by frame size) */
#define LINK_ELK_TILES 16          /* link_elk_core_forward: the tile form (two launches: link_elk_premix_modsum_tiles +
                                      link_elk_gather_demod_tiles); needs link_elk_buffers_t::s_bytes */

/* pre_mix: fin = LayerNorm(F @ Wpre^T) * g + b  (linkunet.py:109-112,132).  F fp[N,C], Wpre fp[C,C]
 * (nn.Linear layout [out,in]).  C <= 256. */
int link_premix_ln(const float *feats, const float *w_pre, const float *ln_w, const float *ln_b,
                   int64_t n, int32_t c, float eps, float *fin, void *stream);

/* Modulate + per-block pre-aggregation (linkunet.py:151-160 + utils.py:52,75-76 fused):
 *   theta = (float(xyz)/coord_div) @ Wpos^T (* alpha), tiled by cg;
 *   S[b] = [ sum_i fin_i*cos(theta_i), sum_i fin_i*sin(theta_i) (, sum_i fin_i*theta_i), count_b ]
 * over the voxels of block b in ascending voxel id.  Block SUMS, not means (mean*count of the
 * reference re-multiplies what it just divided).  S layout: m_cap rows of nparts*C floats (row =
 * 512 B at C=64: four aligned 128-B lines), one extra all-zero row (id m_cap: what absent neighbours
 * point at), then the m_cap+1 counts (fp32) at S + (m_cap+1)*nparts*C, i.e. S has
 * (m_cap+1)*(nparts*C + 1) floats.  The SAME m_cap must be passed to every call that touches S.
 * w_pos fp[cg,3]; alpha fp[cg] or NULL. */
int link_modulate_block_sum(const float *fin, const int32_t *vox_sorted, const float *w_pos,
                            const float *alpha, const int32_t *blk_start,
                            const int32_t *hdr, const link_elk_desc_t *desc /* host */, int64_t n,
                            int64_t m_cap, float *S, void *stream);

/* Neighbour-block sum + normalise + broadcast + de-modulate + LayerNorm(norm)
 * (utils.py:65-82 + linkunet.py:162,178 fused): for every block the r^3 neighbour rows of S are
 * summed in get_kernel_offsets order, divided by the summed count, and every voxel of the block gets
 *   new = A_cos*cos(theta) + A_sin*sin(theta) [+ (A_lin - fin*theta)]   ('sin': A_cos*cos - A_sin*sin
 *   with X=[F sin, F cos]) ; out = LayerNorm(new)*g + b.
 * `fin` is only read for LINK_OP_COSX (may be NULL otherwise).  out fp[N,C]. */
int link_gather_demod_ln(const float *S, const float *fin, const int32_t *vox_sorted,
                         const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                         const int32_t *blk_start, const int32_t *blk_coords,
                         const int32_t *cell_blk, const link_grid_t *grid /* host */,
                         const int32_t *hdr, const link_elk_desc_t *desc /* host */, int64_t n,
                         int64_t m_cap, float *out, void *stream);

/* Split form of link_gather_demod_ln (same result, two simpler kernels; C % 4 == 0, r <= 3):
 *   link_block_gather   A[m] = (sum of the r^3 neighbour rows of S) / (summed count)  -> fp[M, P*C]
 *   link_voxel_demod_ln per voxel (loop-free, one 16-byte-per-lane group per voxel pair): read the
 *                       block's A row, de-modulate, LayerNorm, store out[voxel]. */
int link_block_gather(const float *S, const int32_t *blk_coords, const int32_t *cell_blk,
                      const link_grid_t *grid /* host */, const int32_t *hdr,
                      const link_elk_desc_t *desc /* host */, int64_t m_cap, float *A, void *stream);
int link_voxel_demod_ln(const float *A, const float *fin, const int32_t *vox_sorted,
                        const int32_t *pos_blk, const float *w_pos, const float *alpha, const float *ln_w,
                        const float *ln_b, const int32_t *hdr, const link_elk_desc_t *desc /* host */,
                        int64_t n, float *out, void *stream);

/* One-call form of the whole R_core step (what bench.py times): optional index build (section B)
 * followed by the section-C kernels (pre_mix, modulate+block-sum, block gather, voxel de-modulate),
 * all on `stream`, from caller-owned buffers.  Exists so
 * that a host written in an interpreted language pays ONE FFI crossing per LinK block instead of one
 * per kernel.  All pointers are device pointers with the meanings documented above. */
typedef struct {
  const void *feats;         /* [N,C]  block input (st.F): fp32, or io_dtype rows with LINK_ELK_TILES */
  const int32_t *coords;     /* i32[N,4] (st.C) */
  const float *w_pre, *pre_ln_w, *pre_ln_b;   /* pre_mix.0.weight [C,C], pre_mix.1.{weight,bias} [C] */
  const float *w_pos, *alpha;                 /* pos_weight.0.weight [cg,3]; alpha [cg] or NULL */
  const float *ln_w, *ln_b;                   /* norm.{weight,bias} [C] */
  uint32_t *cell_counts; void *scratch; size_t scratch_bytes;          /* link_index_build scratch */
  int32_t *cell_blk, *vox_blk; int64_t *idx_query; int32_t *perm, *vox_sorted, *pos_blk, *blk_start, *blk_coords, *counts, *hdr;
  float *fin;                /* fp[N,C]            scratch: pre_mix output */
  float *S;                  /* fp[(m_cap+1)*(P*C+1)] scratch: block table rows + zero row + counts */
  float *A;                  /* fp[m_cap, P*C]     scratch: normalised neighbour sums (NULL: fused gather) */
  void *out;                 /* [N,C]              result (new_st_F after self.norm), same element type as feats */
  int64_t s_bytes;           /* bytes available at S (0: the size above); LINK_ELK_TILES needs link_elk_tiles_table_bytes() */
  int32_t io_dtype;          /* LINK_IO_F32 (0) / LINK_IO_F16 / LINK_IO_BF16: element type of feats and out; 16-bit rows need
                                LINK_ELK_TILES (the four-kernel form is fp32) */
  int32_t reserved;
  uint64_t *cell_pair;       /* u64[V] zero-filled once, self-cleaning: link_index_build_first (build_index = 2); NULL otherwise */
} link_elk_buffers_t;

/* Tile form of the section-C kernels on a built index (elk_tiles_impl.h): TWO launches for R_core instead of four, made for
 * the frames the dense-cell layout (section E) does not take: sparse block grids with many voxels per occupied block
 * (LiDAR: linkunet.py:345-363 on SemanticKITTI, scn.py:586-607 on nuScenes), C in {16, 32, 64, 128}, r in {2, 3}.
 *   link_elk_premix_modsum_tiles  = link_premix_ln + link_modulate_block_sum in one pass over the voxels in block order
 *       (pre_mix on the matrix cores; per-block sums by segmented scans in the accumulator layout; blocks of any size are
 *       split over waves and combined in a fixed order: bitwise reproducible, no atomics).  Writes the table S (rows,
 *       zero row, counts -- the layout of link_modulate_block_sum -- followed by scratch rows: S must hold
 *       link_elk_tiles_table_bytes(desc, n, m_cap) bytes) and, for LINK_OP_COSX only, fin.
 *   link_elk_gather_demod_tiles   = link_block_gather + link_voxel_demod_ln in one kernel: a wave per 16-64 sorted positions,
 *       the neighbour sums of the blocks among them formed once in LDS.
 * Same results as the four-kernel form to rounding (sums are formed in a different, fixed order). */
int64_t link_elk_tiles_table_bytes(const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap);
int link_elk_premix_modsum_tiles(const float *feats, const int32_t *vox_sorted, const int32_t *pos_blk,
                                 const int32_t *blk_start, const int32_t *hdr, const float *w_pre,
                                 const float *pre_ln_w, const float *pre_ln_b, const float *w_pos, const float *alpha,
                                 const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap, float *S,
                                 int64_t s_bytes, float *fin, void *stream);
int link_elk_gather_demod_tiles(const float *S, const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                                const int32_t *blk_coords, const int32_t *cell_blk, const link_grid_t *grid /* host */,
                                const int32_t *hdr, const float *w_pos, const float *alpha, const float *ln_w,
                                const float *ln_b, const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap,
                                float *out, void *stream);
/* The two tile-form entries with fp16 / bf16 feature rows at the kernel boundary (io_dtype = LINK_IO_F32 / LINK_IO_F16 /
 * LINK_IO_BF16, section D: feats of the first, out of the second; tables, fin, sums and LayerNorm stay fp32) -- what
 * TSELKBlock runs under autocast on LiDAR-shaped frames, with no cast pass on either side. */
int link_elk_premix_modsum_tiles_io(const void *feats, int32_t io_dtype, const int32_t *vox_sorted, const int32_t *pos_blk,
                                    const int32_t *blk_start, const int32_t *hdr, const float *w_pre, const float *pre_ln_w,
                                    const float *pre_ln_b, const float *w_pos, const float *alpha,
                                    const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap, float *S,
                                    int64_t s_bytes, float *fin, void *stream);
int link_elk_gather_demod_tiles_io(const float *S, const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                                   const int32_t *blk_coords, const int32_t *cell_blk, const link_grid_t *grid /* host */,
                                   const int32_t *hdr, const float *w_pos, const float *alpha, const float *ln_w,
                                   const float *ln_b, const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap,
                                   void *out, int32_t io_dtype, void *stream);

/* build_index: 0 = the index of the previous call on these buffers (same coordinates); 1 = link_index_build first;
 * 2 = link_index_build_first first (buf->cell_pair must be set). */
int link_elk_core_forward(const link_elk_buffers_t *buf /* host */, const link_grid_t *grid /* host */,
                          const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap,
                          int32_t build_index, void *stream);

/* Lean form of R_core with the index REBUILT every call (round 4, csrc/elk_lean_impl.h): THREE launches, none proportional to the
 * grid, for the LiDAR stage frames the reference feeds its blocks (linkunet.py:345-363, scn.py:586-607: a few thousand to a few ten
 * thousand voxels, block grid 1-20 % occupied).  Replaces, per call, what the reference rebuilds per call: sphash + torch.unique +
 * sphashquery (utils.py:44-63, query_cuda.cu:9-58) and the voxelize / devoxelize pair (voxelize_cuda.cu:12-25,
 * devoxelize_cuda.cu:11-34) around linkunet.py:132-178.
 *   launch 1  X rows [F cos | F sin (| F theta)] of the voxels in input order (pre_mix on the matrix cores, LayerNorm, theta,
 *             sincos) into `X`; rank = cnt[cell]++, list[cell][rank] = voxel id; one work item (cell * 16 + chunk) per started
 *             chunk of 32 voxels of a cell, appended to one of 16 item lists (one atomic per workgroup); workgroups behind the
 *             frame's own clear the previous frame's counters through its item lists;
 *   launch 2  a wave per item: the chunk's voxels in ascending id (ranked by counting),
 *             S[cell][chunk] = sum of their X rows, their records (x, y, z, id) in that order -> rec2;
 *   launch 3  a wave per item: chunk rows of the r^3 neighbour cells (present iff cnt > 0; the first chunk of each is requested
 *             together with the counts) summed and divided by the summed count, the chunk's voxels de-modulated and
 *             LayerNorm'ed -> out.  cos_x reads fin * theta back from the voxel's X row.
 * Bitwise reproducible (sums in id order, fixed combination order).  No block numbering exists: tables are addressed by grid
 * cell and touched only where voxels land.  C in {16, 32, 64, 128}, r in {2, 3}, k <= 352, cells < 2^27.
 * The caller alternates (cnt, occ, ctrl) with (cnt_prev, occ_prev, ctrl_prev) between indexed steps (build_index = 1); with
 * build_index = 0 the lists of the previous step on the same buffers are reused (same coordinates) and the *_prev members are
 * not touched.  `n_prev`: voxels of the previous indexed step (0 on the first).
 * hdr: STATUS bit 0 a voxel outside the grid, bit 1 a cell over capacity (such voxels get no output row). */
typedef struct {
  const void *feats;          /* [N,C] io_dtype rows */
  const int32_t *coords;      /* i32[N,4] */
  const float *w_pre, *pre_ln_w, *pre_ln_b, *w_pos, *alpha, *ln_w, *ln_b;   /* as link_elk_buffers_t */
  uint32_t *cnt, *cnt_prev;   /* u32[V << cnt_shift] each, zero-filled once; self-cleaning from then on */
  int32_t *list;              /* i32[V*k]: voxel ids of a cell in arrival order */
  int32_t *rec2;              /* i32[16*seg_cap*32][4]: records (x, y, z, id) of an item's voxels in ascending id */
  int32_t *occ, *occ_prev;    /* i32[16 * seg_cap] each: work items, list l at l * seg_cap */
  uint32_t *ctrl, *ctrl_prev; /* u32[256] each, zero-filled once: item count of list l at word 16 l */
  float *X;                   /* f32[n_cap, P*C] scratch */
  float *S;                   /* f32[V * kch, P*C] chunk rows by cell, kch = ceil(k / 32); needs no initialisation */
  int32_t *hdr;               /* i32[8] zero-filled once */
  void *out;                  /* [N,C] io_dtype rows */
  int64_t seg_cap;            /* items a list holds: >= ceil(ceil(n_cap / 64) / 16) * 64 + 64 */
  int32_t k;                  /* slot capacity of a cell (s^3 clipped to 352) */
  int32_t io_dtype;           /* LINK_IO_* */
  int32_t cnt_shift;          /* the counter of cell c is word c << cnt_shift (0 .. 5): on a small grid every counter gets its own
                                 line, because the atomics of a few hundred cells with hundreds of voxels each would otherwise
                                 serialise on a handful of lines */
  int32_t reserved;
} link_lean_buffers_t;
int link_elk_core_lean_forward(const link_lean_buffers_t *buf /* host */, const link_grid_t *grid /* host */,
                               const link_elk_desc_t *desc /* host */, int64_t n, int64_t n_prev, int32_t build_index,
                               void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training form of R_core (forward with saved state + hand-written backward).
 *
 * The reference differentiates linkunet.py:132-178 through torch autograd: nn.Linear / nn.LayerNorm
 * nodes, the sin/cos/mul/cat graph, and VoxelizeFunction.backward / DevoxelizeFunction.backward
 * (voxelize.py:34-50, devoxelize.py:76-93; voxelize_cuda.cu:28-42, devoxelize_cuda.cu:37-59: fp
 * atomicAdd scatter).  Here the forward is the inference kernels (plus the region counts `den`), and
 * the backward is six kernels, deterministic (no atomics); the only step left to the host's GEMM library
 * is the weight gradient g_pre^T @ F.  Group kernels only: C % 4 == 0 and r <= 3 (pre_mix backward:
 * C % 16 == 0, C <= 128); LINK_ERR_ARG otherwise and the host falls back to its op-by-op composition.
 *
 * All `partials` outputs are fp[link_elk_mid_partial_rows(), Q, C] per-workgroup partial sums which the
 * host adds up over rows (fixed grid -> deterministic).
 *
 * link_elk_mid_forward   fin -> out.  S scratch fp[(m_cap+1)*(P*C+1)]; A fp[m_cap, P*C] and den
 *     fp[m_cap] are SAVED for backward.  ln_w/ln_b NULL: out = new (before self.norm); non-NULL:
 *     out = self.norm(new) (the inference kernels unchanged).
 * link_elk_out_ln_backward   backward of self.norm with its input recomputed from (A, fin, theta):
 *     g_out = d/d(norm output) -> g_new = d/d(new); partials Q=2: [d norm.weight | d norm.bias].
 * link_elk_mid_backward      g_new -> g_fin (d/d(pre_mix output)); S and gS fp[m_cap,P*C] scratch;
 *     partials Q=4: [d alpha (tiled) | d w_pos[:,0] | d w_pos[:,1] | d w_pos[:,2]] per channel; the host
 *     folds channels ch -> ch % cg (theta is tiled, linkunet.py:154).
 * link_premix_ln_backward    backward of pre_mix = LayerNorm(F @ Wpre^T) with the pre-norm activations
 *     recomputed by the forward's MFMA schedule: g_fin -> g_pre fp[N,C] (d/d(F @ Wpre^T), stored for the
 *     weight-gradient GEMM) and g_feats fp[N,C] = g_pre @ Wpre (second MFMA pass); partials Q=2:
 *     [d pre_mix.1.weight | d pre_mix.1.bias]. */
int link_elk_mid_forward(const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                         const int32_t *blk_start, const int32_t *blk_coords, const int32_t *cell_blk,
                         const link_grid_t *grid /* host */, const int32_t *hdr, const float *w_pos,
                         const float *alpha, const link_elk_desc_t *desc /* host */, const float *ln_w,
                         const float *ln_b, int64_t n, int64_t m_cap, float *S, float *A, float *den,
                         float *out, void *stream);
int32_t link_elk_mid_partial_rows(void);
int link_elk_out_ln_backward(const float *g_out, const float *A, const float *fin, const int32_t *vox_sorted,
                             const int32_t *pos_blk, const float *w_pos, const float *alpha,
                             const float *ln_w, const int32_t *hdr, const link_elk_desc_t *desc /* host */,
                             int64_t n, float *g_new, float *partials, void *stream);
int link_elk_mid_backward(const float *g_new, const float *fin, const float *A, const float *den,
                          const int32_t *vox_sorted, const int32_t *pos_blk, const int32_t *blk_start,
                          const int32_t *blk_coords, const int32_t *cell_blk,
                          const link_grid_t *grid /* host */, const int32_t *hdr, const float *w_pos,
                          const float *alpha, const link_elk_desc_t *desc /* host */, int64_t n, int64_t m_cap,
                          float *S, float *gS, float *g_fin, float *partials, void *stream);
int link_premix_ln_backward(const float *feats, const float *w_pre, const float *ln_w, const float *g_fin,
                            int64_t n, int32_t c, float eps, float *g_pre, float *g_feats, float *partials,
                            void *stream);
/* The block's tail for training: y = relu(addend + LayerNorm(x) * ln_w + ln_b)
 * (`st.F = self.activate(new_st_F + self.norm_local(st_local.F))`, linkunet.py:183 / ts_elk.py:228) and its
 * backward: g_addend = g_y * (y > 0); g_x = LayerNorm backward of that (statistics recomputed from x);
 * partials fp[link_elk_mid_partial_rows(), 2, C] = per-workgroup sums of [d ln_w | d ln_b].  C % 4 == 0,
 * C <= 256.  (Inference fuses the tail into the convolution: link_subm_conv_ln_add_relu.) */
int link_ln_add_relu_forward(const float *x, const float *addend, const float *ln_w, const float *ln_b,
                             int64_t n, int32_t c, float eps, float *y, void *stream);
int link_ln_add_relu_backward(const float *g_y, const float *y, const float *x, const float *ln_w, int64_t n,
                              int32_t c, float eps, float *g_addend, float *g_x, float *partials, void *stream);
/* Column sums of up to three partial arrays fp[rows, cols_k] (k = 0..2; cols_k == 0 skips one) into
 * out fp[cols0+cols1+cols2], one launch, fixed summation order. */
int link_sum_partials(const float *p0, int32_t cols0, const float *p1, int32_t cols1, const float *p2,
                      int32_t cols2, int64_t rows, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Section D -- row N1 of SURVEY.md section 8f: stride-1 submanifold sparse convolution (what
 * ELKBlock.local_mix = spnn.Conv3d(inc, inc, 3) runs, linkunet.py:109,125).
 * Replaces torchsparse.backend.convolution_forward_cuda (pybind_cuda.cpp:19;
 * convolution/convolution_cuda.cu:53-165: per kernel offset gather -> cuBLAS mm -> scatter-add) for the
 * in-place-coordinates case with ONE output-stationary kernel (accumulators in registers over all
 * offsets, W_k staged in LDS, f32 MFMA), fed by a per-output neighbour table instead of the
 * reference's (in,out) pair lists:
 *   nbr  i32[N, kvol]   nbr[v,k] = input row at coords[v] + offset_k * tensor_stride, -1 absent; offsets
 *                       in get_kernel_offsets order (nn/utils/kernel.py:9-33) -- link_neighbor_map output
 *   w    fp[kvol, Cin, Cout]   the module's `kernel` parameter as stored (nn/modules/conv.py:34-38)
 *   out  fp[N, Cout] = sum_k feats[nbr[:,k]] @ w[k]
 *   order i32[N] or NULL: a permutation of the voxels (e.g. link_index_build's `perm`): tile t of the MFMA
 *                       kernel computes voxels order[16t..16t+15]; spatially sorted voxels let whole tiles
 *                       skip absent offsets.  Results do not depend on it.
 * The table is per OUTPUT row, so the same entry serves down-sampling (kernel 2 / stride 2: N = coarse rows,
 * entries = fine rows) and transposed convolutions (N = fine rows, one non-absent entry = the parent);
 * feats may have any number of rows.  MFMA path for Cin, Cout multiples of 16 up to 128 (the square
 * widths and the channel changes of the reference encoders are instantiated); other widths <= 256 take a
 * lane=channel kernel.
 * The input gradient of this convolution is the same call on grad_out with w'[k] = w[kvol-1-k]^T
 * (the neighbour relation of an odd kernel at stride 1 is symmetric). */
int link_subm_conv_forward(const float *feats, const int32_t *nbr, const float *w, const int32_t *order,
                           int64_t n, int32_t cin, int32_t cout, int32_t kvol, float *out, void *stream);
/* Row N2 (fused epilogue): out = relu(addend + LayerNorm(conv(feats)) * ln_w + ln_b) in the convolution's
 * store phase -- `st.F = self.activate(new_st_F + self.norm_local(st_local.F))` (linkunet.py:183,
 * ts_elk.py:228) with addend = the R_core output.  addend may be NULL; eps of norm_local.
 * `relu` is a flag word: bit 0 = ReLU; bit 1 = ln_w / ln_b are a plain per-channel affine out = conv * ln_w + ln_b
 * (an inference-mode BatchNorm folded to scale / shift, with the convolution's bias folded into the shift) instead
 * of LayerNorm weights -- the BN (+ residual) (+ ReLU) epilogues of the detection stages (scn.py:83-107,482-489)
 * and of the encoders' conv blocks (linkunet.py:18-92,217-224).  Same flag word in link_conv_pairs_sum and
 * link_conv_centre_sum. */
int link_subm_conv_ln_add_relu(const float *feats, const int32_t *nbr, const float *w, const int32_t *order,
                               int64_t n, int32_t cin, int32_t cout, int32_t kvol, const float *ln_w,
                               const float *ln_b, float eps, const float *addend, int32_t relu, float *out,
                               void *stream);
/* Weight gradient of the convolution above: g_w[k][ci][co] = sum_v feats[nbr[v,k]][ci] * g_out[v][co]
 * (replaces the weight half of convolution_backward_cuda, convolution_cuda.cu:167-278: per offset gather +
 * cuBLAS mm(in^T, grad)).  nbr_t i32[kvol, N] is the TRANSPOSED neighbour table (coalesced column
 * reads); C = Cin = Cout <= 64, C % 4 == 0.  partial fp[link_subm_conv_wgrad_chunks(), kvol, C, C]: the
 * host sums the chunk axis (fixed grid -> deterministic). */
int32_t link_subm_conv_wgrad_chunks(void);
int link_subm_conv_wgrad(const float *feats, const float *gout, const int32_t *nbr_t, int64_t n, int32_t c,
                         int32_t kvol, float *partial, void *stream);
/* Round 5: the split is the caller's -- `chunks` pieces of the voxel range per offset, and `centre_extra` MORE pieces for the
 * centre offset kvol / 2, whose column is dense on a submanifold table (every voxel pairs with itself: with one workgroup per
 * (offset, chunk) its workgroups set the kernel's time).  partial f32[kvol * chunks + centre_extra][c][c]; g_w[k] = sum over
 * chunk of slot [chunk * kvol + k], plus for k = kvol / 2 the slots from kvol * chunks on.  (conv.py:67-101, the reference's
 * convolution backward.) */
int link_subm_conv_wgrad_split(const float *feats, const float *gout, const int32_t *nbr_t, int64_t n, int32_t c, int32_t kvol,
                               int32_t chunks, int32_t centre_extra, float *partial, void *stream);
/* ... and its reduction in one launch: g_w f32[kvol][c][c] = sum over chunk of slot [chunk * kvol + k] (+ the centre_extra slots for
 * k = kvol / 2), slot order (deterministic). */
int link_subm_conv_wgrad_reduce(const float *partial, int32_t c, int32_t kvol, int32_t chunks, int32_t centre_extra, float *gw,
                                void *stream);

/* Pair-list form of the same convolution, for sparse frames (few of the K neighbours present per voxel).
 * The kernel map is the reference's own: per kernel offset the list of (input row, output row) pairs
 * (nbmaps / nbsizes, nn/functional/conv.py:109-122; consumed offset by offset in convolution_cuda.cu:90-165
 * as gather -> cuBLAS -> scatter-add).  Here ONE MFMA launch computes every pair's contribution row and ONE
 * output-stationary launch adds them per voxel in a fixed order (no atomics, deterministic) with the optional
 * bias / LayerNorm + add + ReLU epilogue.
 *   rows_pad     contribution rows, a multiple of 128.  Rows are grouped by kernel offset, each group padded to
 *                a 128-row granule; pair_in i32[rows_pad] = input row of the pair (-1: padding);
 *                wg_k i32[rows_pad/128] = kernel offset of each granule (-1: skip).
 *   contrib      fp32[rows_pad, cout] scratch: contrib[p] = feats[pair_in[p]] . w[wg_k[p/128]].
 *   n_direct     output voxels i < n_direct take contrib[i] as their first term (submanifold: the centre
 *                offset's pairs are the identity and occupy rows [0, n)); 0 for strided / transposed maps.
 *   ext_start i32[n+1], ext_list i32[...]: CSR list of the other contribution rows of every output voxel,
 *                ascending kernel offset.
 * Widths: link_conv_pairs_supported(cin, cout) (multiples of 16 the LinK networks use, <= 128). */
int link_conv_pairs_supported(int32_t cin, int32_t cout);
int link_conv_pairs_gemm(const float *feats, const int32_t *pair_in, const int32_t *wg_k, int64_t rows_pad,
                         const float *w, int32_t cin, int32_t cout, float *contrib, void *stream);
int link_conv_pairs_sum(const float *contrib, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                        int64_t n_direct, int32_t cout, const float *bias, const float *ln_w, const float *ln_b,
                        float eps, const float *addend, int32_t relu, float *out, void *stream);
/* Submanifold tables (odd kernel, identical input and output coordinates: nbr[i, centre] == i): the centre
 * offset's GEMM runs on the output rows themselves and the voxel is finished in the accumulators --
 * out[i] = epilogue(feats[i] . w[centre] + sum of contrib[ext_list[ext_start[i] .. ext_start[i+1])]) with the
 * same epilogue as link_conv_pairs_sum.  contrib (contrib_rows rows, < 4 GiB) then holds only the OTHER offsets' rows (link_conv_pairs_gemm
 * over a pair list without the centre), so the centre term never travels through HBM. */
/* Output sites of a site-creating sparse convolution (spconv's SparseConv3d as the detection backbone uses it,
 * scn.py:496-502: kernel 3 or 1, stride 2 or 1, padding per axis; axes (z, y, x), indices i32[n,4] = (b, z, y, x)):
 * cand i32[n * link_conv_out_candidate_count(kernel, stride), 4] receives every candidate site of every input, rows
 * of -1 where a combination is invalid or outside out_shape.  Sorted unique rows = link_index_build with block
 * edge 1 over `cand` (out-of-range rows are dropped there). */
int32_t link_conv_out_candidate_count(const int32_t *kernel, const int32_t *stride);
int link_conv_out_candidates(const int32_t *indices, int64_t n, const int32_t *kernel, const int32_t *stride,
                             const int32_t *padding, const int32_t *out_shape, int32_t *cand, void *stream);
/* Gather table of such a convolution straight from the (b, z, y, x) rows: link_conv_site_table scatters input row + 1 into a
 * table over (batch, in_shape) that the caller keeps all zero between uses (clear != 0 undoes the scatter of the same rows:
 * no memset over a sparse grid); link_conv_gather_table then writes table i32[m, taps] = the input row at
 * out * stride - padding + tap for the taps (a, b, c) in row-major order over `kernel` (each 1..3), -1 where no site.
 * batch * prod(in_shape) < 2^31. */
int link_conv_site_table(const int32_t *indices, int64_t n, const int32_t *in_shape, int32_t batch, int32_t *table,
                         int32_t clear, void *stream);
int link_conv_gather_table(const int32_t *out_indices, int64_t m, const int32_t *kernel, const int32_t *stride,
                           const int32_t *padding, const int32_t *in_shape, int32_t batch, const int32_t *site_table,
                           int32_t *table, void *stream);
/* Building the pair plan from a per-output neighbour table nbr i32[n, kvol] (-1 absent), kvol <= 64 -- the device
 * half of what nn/functional/conv.py:109-122 does with nonzero / sum on the host side.  G = ceil(n / 256) workgroups:
 *   link_pair_plan_count   wg_counts i32[G, kvol + 1]: per workgroup, pairs of every offset, and (last column) rows
 *                          whose centre entry nbr[i, kvol/2] != i (column sum 0 <=> submanifold table);
 *                          row_info i32[n] = (#valid entries of the row) | (centre entry valid) << 16.
 *   link_pair_plan_fill    after the host laid out base_k i32[kvol] (first contribution row of every offset's
 *                          128-row granules), wg_base i32[G, kvol] (exclusive scan of wg_counts over the workgroups)
 *                          and ext_start i32[n+1] (exclusive scan of the rows' list lengths): pair_in[p] = input row
 *                          of pair p, pair_out[p] = its output row (both pre-filled with -1), ext_list = every row's contribution rows in
 *                          ascending offset order.  No atomics: placement is deterministic.  skip_centre != 0 leaves
 *                          the centre offset out (link_conv_centre_sum computes it). */
int link_pair_plan_count(const int32_t *nbr, int64_t n, int32_t kvol, int32_t *wg_counts, int32_t *row_info, void *stream);
int link_pair_plan_fill(const int32_t *nbr, int64_t n, int32_t kvol, int32_t skip_centre, const int32_t *base_k,
                        const int32_t *wg_base, const int32_t *ext_start, int32_t *pair_in, int32_t *pair_out,
                        int32_t *ext_list, void *stream);
/* The step between the two, on the device (no host round trip; one workgroup): from link_pair_plan_count's per-workgroup
 * counts to base_k i32[kvol] (first contribution row of every offset, 128-row granules), wg_base i32[ceil(n/256), kvol]
 * (pairs of earlier workgroups per offset), gran_start i32[kvol + 1], wg_k i32[gran_cap] (offset of every granule, -1 behind
 * the last: the GEMM kernels return there, so they can be launched over the capacity) and hdr i32[8] = {pairs, rows_pad,
 * granules, rows whose centre neighbour is not the row itself, capacity exceeded, 0, 0, 0} for a later, asynchronous read.
 * skip_centre as in link_pair_plan_fill (the caller knows a submanifold table structurally).  A capacity of
 * ceil((n * kvol + 127 * kvol) / 128) granules is never exceeded.  The reference reads its counts back per layer
 * (nn/functional/conv.py:103-122, `nbsizes.cpu()`). */
int link_pair_plan_layout(const int32_t *wg_counts, int64_t n, int32_t kvol, int32_t skip_centre, int64_t gran_cap,
                          int32_t *base_k, int32_t *wg_base, int32_t *gran_start, int32_t *wg_k, int32_t *hdr, void *stream);
/* All three steps in one call, for capacity-sized lists (what link_amd's device-laid-out plans use): the fill pass computes
 * ext_start i32[n + 1] itself (wg_ext i32[ceil(n/256)] scratch: rows of earlier workgroups) and the layout pass writes -1
 * into the unused tail of every offset's last granule, so neither a prefix-sum pass nor a fill of pair_in / pair_out
 * (i32[gran_cap * 128] each, otherwise uninitialised) is needed.  ext_list i32[>= n * kvol].  gran_cap below
 * ceil((n * (kvol - skip_centre) + 127 * kvol) / 128) -- the capacity that can never be exceeded -- is LINK_ERR_ARG: the
 * lists would be written past it before the header's overflow flag exists. */
int link_pair_plan_build(const int32_t *nbr, int64_t n, int32_t kvol, int32_t skip_centre, int64_t gran_cap, int32_t *wg_counts,
                         int32_t *row_info, int32_t *base_k, int32_t *wg_base, int32_t *gran_start, int32_t *wg_ext,
                         int32_t *wg_k, int32_t *hdr, int32_t *ext_start, int32_t *pair_in, int32_t *pair_out,
                         int32_t *ext_list, void *stream);
/* The same three entries with fp16 / bf16 feature rows at the boundary (io_dtype = LINK_IO_F32 / F16 / BF16: feats, addend
 * and out rows in that type; weights, contribution rows, statistics and accumulation fp32) -- the reference's AMP
 * contract for its convolution (custom_fwd(cast_inputs=torch.half), nn/functional/conv.py:18). */
int link_conv_pairs_gemm_io(const void *feats, int32_t io_dtype, const int32_t *pair_in, const int32_t *wg_k, int64_t rows_pad,
                            const float *w, int32_t cin, int32_t cout, float *contrib, void *stream);
int link_conv_pairs_sum_io(const float *contrib, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                           int64_t n_direct, int32_t cout, const float *bias, const float *ln_w, const float *ln_b, float eps,
                           const void *addend, int32_t relu, void *out, int32_t io_dtype, void *stream);
int link_conv_centre_sum_io(const void *feats, const float *w, int32_t centre, const float *contrib, int64_t contrib_rows,
                            const int32_t *ext_start, const int32_t *ext_list, int64_t n, int32_t cin, int32_t cout,
                            const float *bias, const float *ln_w, const float *ln_b, float eps, const void *addend,
                            int32_t relu, void *out, int32_t io_dtype, void *stream);
/* fp32 rows on the f16 matrix cores: link_conv_pairs_gemm with both operands as fp16 hi + lo pairs (22 mantissa bits,
 * exact products, fp32 accumulation; three 4-pass matrix instructions per 16 input channels instead of four 8-pass).
 * ws = the weights split and transposed per offset, fp16 [kvol][cout][hi(cin) | lo(cin)]; w = the fp32 weights (used when a
 * value lies outside the fp16 range); w_big = device flag, non-zero when some |w| >= 2^15.  Inference form. */
int link_conv_pairs_gemm_split(const float *feats, const int32_t *pair_in, const int32_t *wg_k, int64_t rows_pad, const void *ws,
                               const float *w, const int32_t *w_big, int32_t cin, int32_t cout, float *contrib, void *stream);
/* AMP form of the two MFMA entries: the rows AND the weights are 16-bit (io_dtype = LINK_IO_F16 / LINK_IO_BF16 for both;
 * custom_fwd(cast_inputs=torch.half), nn/functional/conv.py:18, rounds the kernel together with the features), the
 * products run on the f16 / bf16 matrix cores with fp32 accumulation; contribution rows, statistics and epilogue
 * stay fp32.  wt = the weights rounded to the row type and transposed per offset: [kvol][cout][cin].
 * contrib_dtype = LINK_IO_F32, or io_dtype: the contribution rows are stored in the row type too (the reference's half
 * torch.mm output, convolution_cuda.cu:127-140) and summed in fp32 -- half the bytes of the dominant stream. */
int link_conv_pairs_gemm_amp(const void *feats, int32_t io_dtype, const int32_t *pair_in, const int32_t *wg_k, int64_t rows_pad,
                             const void *wt, int32_t cin, int32_t cout, void *contrib, int32_t contrib_dtype, void *stream);
int link_conv_centre_sum_amp(const void *feats, const void *wt, int32_t centre, const void *contrib, int32_t contrib_dtype,
                             int64_t contrib_rows, const int32_t *ext_start, const int32_t *ext_list, int64_t n, int32_t cin,
                             int32_t cout, const float *bias, const float *ln_w, const float *ln_b, float eps,
                             const void *addend, int32_t relu, void *out, int32_t io_dtype, void *stream);
/* Narrow layers in the AMP form without a pair list (conv.hip: every W_k resident in LDS; cin, cout in {16, 32}, kvol <= 27):
 * out = epilogue(sum_k feats[nbr[:, k]] . wt[k]^T) over the per-output neighbour table nbr i32[n, kvol] (as
 * link_subm_conv_forward), rows / addend / out in io_dtype (LINK_IO_F16 or LINK_IO_BF16), wt = [kvol][cout][cin] in the
 * row type (as link_conv_pairs_gemm_amp), fp32 accumulation over all offsets, statistics / affine in fp32.  ln_w NULL: the
 * plain convolution; relu bit 0 ReLU, bit 1 "ln_w / ln_b are a per-channel affine" (folded BatchNorm), as
 * link_subm_conv_ln_add_relu.  Replaces convolution_forward_cuda (pybind_cuda.cpp:19) under autocast for those widths. */
int link_subm_conv_resident_amp(const void *feats, int32_t io_dtype, const int32_t *nbr, const void *wt, const int32_t *order,
                                int64_t n, int32_t cin, int32_t cout, int32_t kvol, const float *ln_w, const float *ln_b,
                                float eps, const void *addend, int32_t relu, void *out, void *stream);
/* Weight gradient over the pair list (the weight half of convolution_backward_cuda, convolution_cuda.cu:167-278):
 * gw[k] = sum over the pairs p of offset k of feats[pair_in[p]]^T . gout[pair_out[p]]   (fp32 [kvol, cin, cout]).
 * One MFMA workgroup per 128-pair granule writes partial fp32[rows_pad/128, cin, cout]; the per-offset sums run in
 * granule order (gran_start i32[kvol+1] = first granule of every offset): deterministic, no atomics.  pair_out
 * i32[rows_pad] as written by link_pair_plan_fill (pre-filled with -1).  n_direct > 0 (a plan built with skip_centre
 * over n_direct voxels): the centre offset's identity pairs i -> i run as ceil(n_direct / 128) further granules, so
 * partial needs rows_pad/128 + ceil(n_direct/128) slots. */
int link_conv_pairs_wgrad(const float *feats, const float *gout, const int32_t *pair_in, const int32_t *pair_out,
                          const int32_t *wg_k, const int32_t *gran_start, int64_t rows_pad, int32_t kvol, int64_t n_direct,
                          int32_t cin, int32_t cout, float *partial, float *gw, void *stream);
int link_conv_centre_sum(const float *feats, const float *w, int32_t centre, const float *contrib,
                         int64_t contrib_rows, const int32_t *ext_start, const int32_t *ext_list, int64_t n, int32_t cin, int32_t cout,
                         const float *bias, const float *ln_w, const float *ln_b, float eps, const float *addend,
                         int32_t relu, float *out, void *stream);

/* =============================================================================================
 * E. Dense-cell form of R_core (the fast path when the block grid is mostly occupied)
 *
 * Same result as section C (linkunet.py:124-185 / ts_elk.py:144-230, R_core of SURVEY.md section 8d), but
 * built for frames whose dense block grid is about as large as the frame itself (V <~ 2 N: cfg1/cfg2
 * S-uniform).  There the reference's hash tables, torch.unique and scatter-add (utils.py:45-52,65-82) and
 * section B's count/scan/place index all disappear: the block table is indexed by GRID CELL, a voxel
 * finds its block by arithmetic, and the only index structure is a per-cell slot list filled by the pre_mix
 * kernel itself (one atomic per voxel).  Three or four launches per step, no scan, no sort, no host sync:
 *   link_dc_premix_insert  fin = LayerNorm(F Wpre^T) (f32 MFMA, software-pipelined) + rank = cnt[cell]++,
 *                          slots[cell][rank] = (x,y,z,id)
 *   link_dc_modsum         per cell: sort the slot list by voxel id (<= 4: network; more: selection), modulate,
 *                          sum -> S[cell] (every interior cell written, empty ones as zero rows); resets cnt
 *   link_dc_gather         r^3 box sum over the padded grid: LDS-DMA plane ring, xy sum from LDS, z ring in
 *                          registers -> A[cell]
 *   link_voxel_demod_ln    (section C kernel, fed with vrec/vcell) per voxel de-modulate + LayerNorm
 * The grid is padded by one empty cell on every spatial side (rows stay zero: the caller zero-fills S, Scnt
 * and A ONCE), so neighbour addressing needs no bounds checks.  Cell id =
 * ((b*pdim0 + x+1)*pdim1 + y+1)*pdim2 + z+1 with (x,y,z,b) the block coordinate minus grid.lo.
 * Slot capacity k = s^3 covers every frame with unique voxel coordinates; a voxel that finds its cell full
 * (duplicate coordinates) or lies outside the grid is dropped and flagged in hdr[LINK_HDR_STATUS]
 * (bit0 outside, bit1 slot overflow) -- callers that cannot rule that out use section C.
 * Supported: C in {16,32,64,128}, r in {2,3}; LINK_ERR_ARG otherwise.
 * ============================================================================================= */
typedef struct {
  int32_t s;        /* block edge in voxel-coordinate units */
  int32_t lo[4];    /* block-coordinate lower bound (x,y,z,b) */
  int32_t dim[4];   /* interior extent in blocks (x,y,z,b) */
  int32_t pdim[3];  /* padded spatial extent = dim + 2 */
  int32_t k;        /* slots per cell */
  int64_t vp;       /* padded cells = pdim0*pdim1*pdim2*dim3 */
} link_dc_grid_t;
/* Host helper: padded grid + slot capacity from a link_grid_t.  k <= 0 selects s^3.  Returns vp or -1
 * (vp*k must stay below 2^31 slots). */
int64_t link_dc_grid_from(const link_grid_t *grid, int32_t k, link_dc_grid_t *out);

#define LINK_HDR_STATUS_ACC 3   /* status bits being collected by the running step (dense-cell path) */

#define LINK_IO_F32 0
#define LINK_IO_F16 1
#define LINK_IO_BF16 2
/* Launch geometry and kernel selection of ONE plan (all zero = defaults).  Part of link_dc_buffers_t, read at every
 * call: nothing about the dense-cell path is process-global, two plans with different settings can run concurrently
 * from different threads / streams (re-entrant, no global state). */
typedef struct {
  int32_t k1_wgs;      /* workgroups (4 waves each) of the fused pre_mix kernel; 0 = 512.  One frame alone wants 2 per CU
                          (512); with several frames in flight 256 co-schedules better */
  int32_t k2_zsplit;   /* z-segments of the fused gather + de-modulate kernel; 0 = auto (enough for ~512 workgroups) */
  int32_t k1_lds_pad;  /* extra dynamic LDS bytes of the fused pre_mix kernel (<= 16384) / the gather kernel (<= 4096): */
  int32_t k2_lds_pad;  /* which workgroups share a CU when several frames are in flight */
  int32_t k1_form;     /* 0 = cell-range form (a wave owns a range of cells: LDS voxel list + X tile; the faster one with frames in
                          flight); 1 = tile form (id slots, in-wave id sort, per-cell sums by segmented DPP scan in the matrix-core
                          accumulator layout; 36 KB of LDS, any cell size up to the slot capacity in one pass structure) */
  int32_t k2_form;     /* 0 = by measurement: producer / consumer form with quad consumers (two-part rows whose channels j and j + C/2
                          share theta), producer / consumer form with pair consumers (other two-part rows), own-cell form (cos_x, r = 3),
                          single-role form (cos_x, r = 2).  bit 3: pair consumers (the round-2 kernel) instead of quad consumers;
                          bit 0: single-role form (workgroup-wide dealing); bit 1: one voxel per lane group instead of a pair (pair
                          forms); bit 2: own-cell form (a lane group de-modulates its own cell from the A row in registers; 2-slot
                          plane ring fed by a dedicated DMA wave; 39 KB of LDS) */
  int32_t mode;        /* 0 = default (7); else bit 0 fused pre_mix+modsum, bit 1 dense-cell demod kernel, bit 2 fused
                          gather + de-modulate (C = 64; the other widths keep box sum and de-modulation as two kernels: a fused form for them
                          measured slower and was removed in round 4) -- the unfused stages are what the fused ones are tested against */
  int32_t reserved0;   /* (was k1_pipe, the software-pipelined tile variant: removed in round 6 -- 35.7 against 34.2 us / frame) */
  int32_t reserved;
  uint64_t *k1_dbg;    /* bench only: device buffer u64[waves*8] for per-wave phase timings of the fused pre_mix kernel; NULL = off */
  uint64_t *k2_dbg;    /* bench only: device buffer u64[workgroups*8*8] for per-wave timings of the producer / consumer gather kernel */
} link_dc_tuning_t;

typedef struct {
  const void *feats;         /* [N,C] in io_dtype */
  const int32_t *coords;     /* i32[N,4] */
  const float *w_pre, *pre_ln_w, *pre_ln_b, *w_pos, *alpha, *ln_w, *ln_b;   /* as link_elk_buffers_t */
  uint32_t *cnt;             /* u32[vp]     voxels per cell; all-zero on entry and on exit (self-cleaning) */
  int32_t *slots;            /* i32[vp*k,4] (x,y,z,id) records; no initialisation needed */
  uint32_t *sid;             /* u32[vp*max(k,8)] voxel ids per cell in arrival order (tile form of the fused pre_mix kernel:
                                8 inline ids per cell = one 32-byte piece, then the overflow region); no initialisation */
  int32_t *vrec;             /* i32[N,4]    (x,y,z,id) per voxel, original order */
  int32_t *vcell;            /* i32[N]      padded cell id per voxel (0 for dropped voxels) */
  int32_t *cell_n;           /* i32[vp]     voxels per cell of the last indexed frame (zero-filled once) */
  int32_t *hdr;              /* i32[8]      LINK_HDR_*; zero-filled once */
  float *fin;                /* fp[N,C] */
  float *S;                  /* fp[(vp+1), P*C] block table, zero-filled once (border rows stay zero) */
  float *A;                  /* fp[(vp+1), P*C] normalised neighbour sums, zero-filled once */
  void *out;                 /* [N,C] in io_dtype */
  int32_t io_dtype;          /* LINK_IO_*: type of feats and out at the kernel boundary.  fp16 / bf16 are honoured
                                by the fused kernels (link_dc_premix_modsum, link_dc_gather_demod, link_dc_demod):
                                rows are converted on load / store, the contraction, theta, the block table and the
                                LayerNorm statistics stay fp32 -- the reference's AMP contract (custom_fwd(cast_inputs
                                = torch.half) on voxelize / devoxelize, nn/functional/voxelize.py:13, devoxelize.py:54,
                                fp32 accumulation) */
  link_dc_tuning_t tune;
} link_dc_buffers_t;

int link_dc_premix_insert(const float *feats, const int32_t *coords, const float *w_pre, const float *ln_w,
                          const float *ln_b, int64_t n, int32_t c, float eps, const link_dc_grid_t *g /* host */,
                          int32_t insert, float *fin, uint32_t *cnt, int32_t *slots, int32_t *vrec,
                          int32_t *vcell, int32_t *hdr, void *stream);
int link_dc_modsum(const float *fin, const int32_t *slots, uint32_t *cnt, int32_t *cell_n, const float *w_pos,
                   const float *alpha, const link_elk_desc_t *desc /* host */, const link_dc_grid_t *g /* host */,
                   int32_t warm, float *S, int32_t *hdr, void *stream);
int link_dc_gather(const float *S, const int32_t *cell_n, const link_elk_desc_t *desc /* host */,
                   const link_dc_grid_t *g /* host */, float *A, void *stream);
/* Fused forms (link_amd/csrc/dense_fused.hip) -- what link_elk_core_dense_forward runs by default:
 *   link_dc_index          the slot insert alone: coords -> cnt / slots / vcell (status bits as above)
 *   link_dc_premix_modsum  pre_mix + LayerNorm + theta + modulate + per-cell sum in ONE kernel: a wave owns a
 *                          range of cells, lays their voxels out id-ordered in LDS, and runs them through MFMA
 *                          tiles of 16 voxels, so `fin` never leaves registers (it is written, for the
 *                          de-modulation, only when op == LINK_OP_COSX); C in {16,32,64}, k <= 352
 *   link_dc_demod          per-voxel de-modulate + LayerNorm in original voxel order from A[vcell] */
int link_dc_index(const int32_t *coords, int64_t n, const link_dc_grid_t *g /* host */, uint32_t *cnt,
                  int32_t *slots, int32_t *vcell, int32_t *hdr, void *stream);
int link_dc_premix_modsum(const link_dc_buffers_t *buf /* host */, const link_dc_grid_t *g /* host */,
                          const link_elk_desc_t *desc /* host */, int64_t n, int32_t warm, void *stream);
int link_dc_demod(const float *A, const float *fin, const int32_t *coords, const int32_t *vcell,
                  const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                  const link_elk_desc_t *desc /* host */, const link_dc_grid_t *g /* host */, int64_t n, void *out,
                  int32_t io_dtype, void *stream);
/*   link_dc_gather_demod   box sum + de-modulate + LayerNorm in ONE kernel (C = 64 only: producer / consumer forms; LINK_ERR_ARG
 *                          at other widths, which run link_dc_gather + link_dc_demod): the normalised neighbour sums
 *                          of a z-plane live in LDS only and the plane's voxels are dealt out to the consumer waves -- a voxel
 *                          per quad of lanes through a map a mapper wave lays out two steps ahead (round 4), or as pairs to 16
 *                          lane groups (round-2 form); neither the A table nor its per-voxel gather exists */
int link_dc_gather_demod(const link_dc_buffers_t *buf /* host */, const link_dc_grid_t *g /* host */,
                         const link_elk_desc_t *desc /* host */, int64_t n, void *stream);
/* One call = one R_core step on the dense-cell path (build_index = 0 reuses slots/cell_n of the previous
 * call on the same coordinates: the "warm" figure; 1 inserts the frame first; 2 = link_dc_index_probe inserted it). */
int link_elk_core_dense_forward(const link_dc_buffers_t *buf /* host */, const link_dc_grid_t *g /* host */,
                                const link_elk_desc_t *desc /* host */, int64_t n, int32_t build_index,
                                void *stream);
/* The slot insert of a step (the form buf->tune picks) that also reports the frame's occupancy on this grid: stats
 * i32[16][16] (device, zeroed by the caller): 16 partial slots on separate cache lines, [k][0] += voxels inside the grid,
 * [k][1] += occupied cells, [k][2] max= fullest cell's count -- the reader sums / sums / maxes over k.  For the
 * first visit of a coordinate set: a caller that accepts the layout continues with build_index = 2, one that does not
 * zero-fills cnt and hdr (nothing else of the frame stays behind). */
int link_dc_index_probe(const link_dc_buffers_t *buf /* host */, const link_dc_grid_t *g /* host */, int64_t n, int32_t *stats,
                        void *stream);
/* The slot insert of the tile form: coords -> cnt / sid / vcell (ids only; the fused pre_mix kernel of the tile form
 * writes the id-ordered (x,y,z,id) records the gather kernel reads into `slots`).  Status bits as link_dc_index. */
int link_dc_index_ids(const int32_t *coords, int64_t n, const link_dc_grid_t *g /* host */, uint32_t *cnt,
                      uint32_t *sid, int32_t *vcell, int32_t *hdr, void *stream);

/* =============================================================================================
 * F. Training-mode BatchNorm statistics over feature rows (row N2; torchsparse/nn/modules/norm.py:10-13 applies
 * nn.BatchNorm1d to the [N, C] feature matrix after every convolution, linkunet.py:18-92).  Two column reductions at
 * memory speed with deterministic (fixed-order) double accumulation; the normalisation itself and the input gradient
 * are one fused multiply-add per element and stay with the caller.  C % 4 == 0, 4 <= C <= 1024.
 *   partial   f64[link_bn_partial_workgroups(n, c) * 2 * c] scratch
 *   forward   mean[c], invstd[c] = 1/sqrt(biased var + eps); running_mean / running_var (may be NULL) updated as
 *             nn.BatchNorm1d does: (1 - momentum) * old + momentum * (mean | unbiased var)
 *             scale = weight * invstd, shift = bias  (y = (x - mean) * scale + shift)
 *   backward  sum_g[c] = sum_i g[i], sum_gx[c] = sum_i g[i] * (x[i] - mean) * invstd  (= grad bias, grad weight);
 *             coef = a | bq | cq with grad_x = a * g + bq * (x - mean) + cq per channel
 * ============================================================================================= */
int32_t link_bn_partial_workgroups(int64_t n, int32_t c);
int link_bn_forward_stats(const float *x, int64_t n, int32_t c, float eps, float momentum, double *partial, float *mean,
                          float *invstd, float *running_mean, float *running_var, const float *weight /* NULL: 1 */,
                          const float *bias /* NULL: 0 */, float *scale /* [c] or NULL */, float *shift, void *stream);
int link_bn_backward_reduce(const float *g, const float *x, const float *mean, const float *invstd, int64_t n, int32_t c,
                            double *partial, float *sum_g, float *sum_gx, const float *weight /* NULL: 1 */,
                            float *coef /* [3c] or NULL */, void *stream);
/* Round 5: the normalisation and the input gradient as ONE pass each, with the ReLU that follows the BatchNorm in the
 * reference's blocks (linkunet.py:23-38: Conv3d -> BatchNorm -> ReLU(True)) folded in -- a training step of the cfg3
 * encoder spent a quarter of its 9 ms in torch's elementwise kernels (x - mean, addcmul, clamp, threshold_backward, ...).
 *   link_bn_apply_forward          y = (x - mean) * scale + shift, relu != 0: y = max(y, 0)
 *   link_bn_backward_reduce_relu   link_bn_backward_reduce on g' = g masked by y > 0 (y recomputed from x, scale, shift)
 *   link_bn_apply_backward         gx = a * g' + bq * (x - mean) + cq; scale / shift NULL: g' = g (no relu in the forward) */
int link_bn_apply_forward(const float *x, const float *mean, const float *scale, const float *shift, int64_t n, int32_t c,
                          int32_t relu, float *y, void *stream);
int link_bn_backward_reduce_relu(const float *g, const float *x, const float *mean, const float *invstd, const float *scale,
                                 const float *shift, int64_t n, int32_t c, double *partial, float *sum_g, float *sum_gx,
                                 const float *weight /* NULL: 1 */, float *coef /* [3c] */, void *stream);
int link_bn_apply_backward(const float *g, const float *x, const float *mean, const float *coef, const float *scale /* NULL: no relu */,
                           const float *shift, int64_t n, int32_t c, float *gx, void *stream);

/* =============================================================================================
 * G. One host call per LinK block on a NEW coordinate set (round 5)
 *
 * ELKBlock.forward / TSELKBlock.forward_ of the reference is one Python call that rebuilds every map of its coordinate set
 * (linkunet.py:124-185: voxel_to_aux's hash / unique / query chain, utils.py:44-52; the 3x3x3 convolution's kernel map with
 * `nbsizes.cpu()`, nn/functional/conv.py:103-122).  link_elk_block_forward is that call for the inference forward on the
 * dense-cell layout: bounding box + slot insert with occupancy counters -> ONE host round trip -> R_core (section E) on the
 * caller's stream, behind the 27-neighbour table read off the frame's slot lists (link_dc_neighbor_map), while the context's side
 * stream lays the pair plan out on the device and runs the pair GEMM (section D) -> the convolution's finish: centre offset + pair sums + LayerNorm + add(R_core) + ReLU.
 * The frame is TRIED on the plan the caller hands over (the module's last plan: frames of a stream share their grid): when a voxel
 * lies outside that grid or the occupancy is not what the dense-cell kernels are built for (more than mean_max voxels per occupied
 * cell on average / cell_max in the fullest), the call returns
 * LINK_BLOCK_MISS with bbox / stats filled in and nothing of the frame left in the plan (counters and status word zeroed) --
 * the caller continues on its per-stage entry points without measuring the bounds again.
 * fp32 rows, C = cin = cout with link_conv_pairs_supported(C, C), a submanifold table (subm != 0: unique coordinates, so the centre
 * column is the identity -- verified on the device, pair header word 3).  LINK_ERR_WORKSPACE when pair_arena / contrib are smaller
 * than link_pair_plan_arena says.  The context owns a non-blocking side stream, two events and 2 x 1 KB of scratch (device +
 * pinned host); create one per device and host thread.
 * ============================================================================================= */
typedef struct link_block_ctx link_block_ctx_t;
int link_block_ctx_create(link_block_ctx_t **out);     /* on the current device */
int link_block_ctx_destroy(link_block_ctx_t *ctx);
/* Word offsets (16-byte aligned pieces) of the arena link_pair_plan_build works in with capacity-sized lists:
 * offs[0..9] = wg_counts | row_info | base_k + wg_base + gran_start | wg_ext | wg_k | hdr | ext_start | pair_in | pair_out |
 * ext_list, offs[10] = total words.  Returns the granule capacity (contribution rows = 128 x that) or -1. */
int64_t link_pair_plan_arena(int64_t n, int32_t kvol, int32_t skip_centre, int64_t offs[11]);

/* The 3x3x3 neighbour table of a frame from its freshly inserted slot lists (between link_dc_index / link_dc_index_probe and the
 * pre_mix kernel, which re-orders the lists and resets the counters): nbr i32[n, 27], entry [i][k] = row of the voxel at
 * coords[i] + offset_k * step in get_kernel_offsets(3) order, -1 absent -- the table of link_cell_table_build + link_neighbor_map
 * (conv.py:103-113) without the voxel-resolution cell table.  Voxels the insert dropped (outside the grid) are absent. */
int link_dc_neighbor_map(const int32_t *coords, int64_t n, const link_dc_grid_t *g /* host */, const uint32_t *cnt,
                         const int32_t *slots, int32_t step, int32_t *nbr, void *stream);

#define LINK_BLOCK_DONE 0
#define LINK_BLOCK_MISS 1
typedef struct {
  const link_dc_buffers_t *buf;   /* the plan the frame is tried on; feats / coords = the frame, out = R_core's rows [n, C]
                                     (the convolution's addend), parameters bound */
  const link_dc_grid_t *g;
  const link_elk_desc_t *desc;
  int64_t n;
  int32_t mean_max, cell_max;     /* occupancy the dense-cell kernels take: voxels per occupied cell, mean and maximum */
  int32_t ts;                     /* tensor stride: neighbour of voxel i at coords_i + offset * ts (conv.py:105-113) */
  int32_t subm;                   /* the coordinates are unique (must be non-zero) */
  int32_t *nbr;                   /* out: i32[n, 27] neighbour table (kept by the caller for later layers on these coordinates) */
  int32_t *pair_arena;            /* out: the pair plan (link_pair_plan_arena layout) */
  int64_t pair_arena_words;
  float *contrib;                 /* scratch: contribution rows f32[contrib_rows, C] */
  int64_t contrib_rows;
  const float *w;                 /* local_mix kernel f32[27, C, C] */
  const void *ws;                 /* its fp16 hi | lo split (link_conv_pairs_gemm_split), or NULL: link_conv_pairs_gemm_io */
  const int32_t *w_big;
  const float *nl_w, *nl_b;       /* norm_local */
  float nl_eps;
  int32_t flags;                  /* bit 0: ReLU */
  void *out;                      /* [n, C]: relu(R_core + LayerNorm(conv(feats))) */
  int32_t bbox[8];                /* results (host): min x,y,z,b, max x,y,z,b */
  int32_t stats[4];               /* voxels inside the plan's grid, occupied cells, fullest cell, miss reasons (1 outside, 2 occupancy) */
  int32_t verdict;                /* LINK_BLOCK_DONE / LINK_BLOCK_MISS */
  int32_t reserved;
} link_block_args_t;
int link_elk_block_forward(link_block_ctx_t *ctx, link_block_args_t *args /* host, in/out */, void *stream);

/* =============================================================================================
 * H. R_core of a BATCH of independent frames in one call (round 6; csrc/dense_batch.hip)
 *
 * The reference batches frames by the batch column of its coordinates (the batch index is part of every block key,
 * segmentation/core/models/utils.py:45; linkunet.py:132,151-162,178 run one ELKBlock over the whole collated batch) and shards
 * independent frames over GPUs (BASELINE.json configs[3]: "a batch of 8 independent frames").  link_elk_core_dense_forward_batch
 * is the dense-cell R_core (section E) of `nframes` frames -- each with its own link_dc_buffers_t, all on one grid, one block's
 * parameters, one descriptor -- as THREE launches, two of them persistent: the slot insert of every frame (one ordinary grid), a
 * K1-role kernel (one workgroup per CU; parameters staged once; every wave walks the frames: its range of cells of frame f as soon
 * as the insert of f has arrived) and a K2-role kernel (one workgroup per CU pulling (frame, tile) items off per-XCD cursors: a tile
 * of frame f as soon as K1 of f has arrived).  The stages of different frames overlap by construction -- per-frame arrival counters
 * with write-through stores / one agent-scope acquire per item replace the stream-ordered launch boundaries of the per-frame calls.
 * Results: those of link_elk_core_dense_forward(build_index = 1) per frame, bit for bit.
 *
 * Contract: C = 64, cg = 32 (two-part rows whose channels j and j + 32 share theta), op cos / sin, r in {2, 3}, coord_div = 1, no
 * alpha, fp32 / fp16 / bf16 rows (io_dtype: one type for all frames of a call), slot capacity <= 352, every frame its own cnt / slots / vcell / cell_n / S / hdr / out -- LINK_ERR_ARG otherwise
 * (nothing launched; the caller runs the frames through section E one by one).  frames[i].tune is not read (the geometry is the
 * roles': workgroups = CUs, whole columns per gather tile).  The call returns when everything is ENQUEUED; the results are complete in `stream`
 * order.  Calls whose frames share no buffers overlap on the device (K1 of the next batch starts under K2 of the previous one):
 * a caller that keeps two batches in flight alternates two sets of frame buffers and submits call s + 1 before it joins call s
 * (link_dc_batch_submit / link_dc_batch_join below).  Up to 48 frames form one launch set; a longer call runs as consecutive sets.  The
 * context owns up to five non-blocking streams (drawn at creation so that the roles sit on hardware queues of their own), 21 events
 * and 29 KB of counters; create one per device (and per host thread).
 * link_dc_batch_status synchronises the context's streams and returns LINK_BATCH_TIMEOUT if a bounded wait inside a kernel gave
 * up (the frames' rows are then undefined), LINK_OK otherwise; out[0] = the first error word, out[1] = launch sets so far.
 * ============================================================================================= */
typedef struct link_dc_batch link_dc_batch_t;
int link_dc_batch_create(link_dc_batch_t **out);       /* on the current device */
int link_dc_batch_destroy(link_dc_batch_t *ctx);
int link_elk_core_dense_forward_batch(link_dc_batch_t *ctx, const link_dc_buffers_t *frames /* host [nframes] */,
                                      const int64_t *n /* host [nframes] */, int32_t nframes, const link_dc_grid_t *g /* host */,
                                      const link_elk_desc_t *desc /* host */, void *stream);
/* The same call in two halves, for a caller that keeps several calls in flight from ONE stream: link_dc_batch_submit enqueues the call
 * behind what `stream` holds so far and hands out a ticket; link_dc_batch_join makes `stream` wait for that call's rows.  Submitting
 * call s + 1 BEFORE joining call s lets the pre_mix role of s + 1 start under the gather role of s without a second caller stream --
 * two caller streams that the runtime happens to multiplex onto one hardware queue serialise the calls (the wait of one stream's
 * join sits in front of the other stream's submit: 50 against 37 us / frame).
 * link_elk_core_dense_forward_batch == submit + join. */
int link_dc_batch_submit(link_dc_batch_t *ctx, const link_dc_buffers_t *frames /* host [nframes] */, const int64_t *n /* host [nframes] */,
                         int32_t nframes, const link_dc_grid_t *g /* host */, const link_elk_desc_t *desc /* host */, void *stream,
                         int64_t *ticket /* host, out */);
int link_dc_batch_join(link_dc_batch_t *ctx, int64_t ticket, void *stream);
int link_dc_batch_status(link_dc_batch_t *ctx, int32_t *out /* host [2] */);
/* Measurement hook: with timing on, the three launches of every launch set are bracketed by timed events on their own streams;
 * link_dc_batch_kernel_times waits for the set `ticket` names (one of the last four) and returns the brackets of its insert, pre_mix and
 * gather kernels in ms. */
int link_dc_batch_set_timing(link_dc_batch_t *ctx, int32_t on);
int link_dc_batch_kernel_times(link_dc_batch_t *ctx, int64_t ticket, float *ms /* host [3] */);
/* Profiling hook (tools/batch_timeline.py): device buffers of 8 x u64 rows the K1 / K2 items append 100 MHz timestamps to (word 0 =
 * rows so far, zeroed by the caller; word 1 = capacity in rows).  Honoured by a -DDC_BT_PROF=1 build of csrc/dense_batch.hip (returns
 * LINK_OK), ignored by the default build (returns 1). */
int link_dc_batch_set_debug(link_dc_batch_t *ctx, uint64_t *k1_rows, uint64_t *k2_rows);
/* Do two streams of the caller sit on ONE hardware queue?  *delay_us = how long a kernel on `b` is held up by a 150 us kernel + event
 * record on `a`: ~5 = queues of their own, >= 150 = one queue (frames kept in flight on two such streams run one after the other).
 * Synchronises both streams. */
int link_streams_share_queue(void *stream_a, void *stream_b, double *delay_us /* host, out */);
/* Diagnostic: does a pair of streams sit on ONE hardware queue (GPU_MAX_HW_QUEUES; an event record on one stream then holds up the
 * other's kernels)?  A 150 us spin kernel + an event record on the first stream, a stamp kernel on the second; delays_us[12] = the
 * second's start behind the first's for (pre_mix -> gather), (pre_mix -> insert), (gather -> insert), (stream -> pre_mix),
 * (stream -> gather), (stream -> insert) and the six reverse pairs: ~5 = separate queues, >= 150 = one queue.  Synchronises the
 * streams involved.  (link_dc_batch_create runs the same test to put its role streams on queues of their own.) */
int link_dc_batch_probe_streams(link_dc_batch_t *ctx, hipStream_t stream, double *delays_us /* host [12] */);

#ifdef __cplusplus
}
#endif
#endif /* LINK_AMD_H_ */
