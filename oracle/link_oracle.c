/*
 * oracle/link_oracle.c -- CPU restatement of the LinK hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (link_amd/) is HIP-only and
 * fails loudly when its extension is missing.
 *
 * Every function restates one reference routine in plain scalar C and cites the reference
 * file:line it follows (paths relative to /root/reference/segmentation/torchsparse-u/torchsparse/
 * unless they start with segmentation/ or detection/).
 *
 * Pinning: the reference ships no tests / golden vectors for this path, so
 * the oracle is pinned against (a) the reference's own C++ CPU ops compiled from where they lie
 * (oracle/build_ref.py -> oracle/_ref/) and (b) golden fixtures produced by importing the
 * reference's Python (tests/golden/make_golden.py), see tests/test_oracle_golden.py.
 *
 * Build: gcc -O2 -fPIC -shared -o oracle/liblink_oracle.so oracle/link_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * sphash: 64-bit FNV-1a over the four 32-bit words of a coordinate row, folded to 60 bits.
 * Follows backend/hash/hash_cpu.cpp:7-18 (== backend/hash/hash_cuda.cu:10-23).
 * ------------------------------------------------------------------------------------------- */
/* OpenMP build (-fopenmp -DLINK_ORACLE_OMP -> liblink_oracle_omp.so): the pragmas sit exactly where the
 * reference's CPU ops have them (voxelize: inner channel loop; devoxelize forward: outer row loop), so that
 * the all-cores CPU baseline of bench.py times the reference's parallel structure, not a better one. */
#ifdef LINK_ORACLE_OMP
#define LINK_OMP_FOR _Pragma("omp parallel for")
#else
#define LINK_OMP_FOR
#endif

static inline int64_t fnv_row(const int32_t c[4]) {
  uint64_t h = 14695981039346656037ULL;
  for (int j = 0; j < 4; j++) {
    h ^= (uint32_t)c[j];
    h *= 1099511628211ULL;
  }
  h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
  return (int64_t)h;
}

void oracle_hash(int64_t n, const int32_t *coords, int64_t *out) {
  for (int64_t i = 0; i < n; i++) out[i] = fnv_row(coords + 4 * i);
}

/* ---------------------------------------------------------------------------------------------
 * sphash(coords, offsets): hash of (x+ox, y+oy, z+oz, b) for each of K offsets, K-major [K,N].
 * Follows backend/hash/hash_cuda.cu:27-55 (the CUDA kernel takes the batch index from the row
 * itself, :42-46).  The CPU twin backend/hash/hash_cpu.cpp:20-39 has a defect: it reads
 * data[3] (batch index of row 0) for every row (:29).  `cpu_batch_bug` != 0 reproduces that
 * defect so the restatement can also be pinned bit-exactly against the reference CPU build.
 * ------------------------------------------------------------------------------------------- */
void oracle_kernel_hash(int64_t n, int64_t k, const int32_t *coords, const int32_t *offsets,
                        int64_t *out, int cpu_batch_bug) {
  for (int64_t kk = 0; kk < k; kk++) {
    for (int64_t i = 0; i < n; i++) {
      int32_t c[4];
      for (int j = 0; j < 3; j++) c[j] = coords[4 * i + j] + offsets[3 * kk + j];
      c[3] = cpu_batch_bug ? coords[3] : coords[4 * i + 3];
      out[kk * n + i] = fnv_row(c);
    }
  }
}

/* ---------------------------------------------------------------------------------------------
 * sphashquery backend: map hash -> (index+1) built from the references with insert-if-absent
 * (first duplicate wins), looked up for each query; 0 = miss.  The Python wrapper subtracts 1
 * (nn/functional/query.py:32).  Follows backend/others/query_cpu.cpp:12-37 (dense_hash_map
 * insert/find semantics; google sparsehash is not vendored -- libsparsehash-dev, unpinned,
 * segmentation/INSTALL.md:42 -- so only its published insert-if-absent/find contract is restated)
 * and backend/others/query_cuda.cu:9-58 (same contract on the cuckoo table).  No reserved key.
 * Implementation: open addressing with linear probing on a power-of-two table.
 * ------------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

int oracle_hash_query(int64_t n1, const int64_t *query, int64_t n, const int64_t *target,
                      const int64_t *target_idx, int64_t *out) {
  uint64_t cap = 16;
  while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
  int64_t *keys = (int64_t *)malloc(cap * sizeof(int64_t));
  int64_t *vals = (int64_t *)calloc(cap, sizeof(int64_t)); /* 0 = empty slot (vals are idx+1 >= 1) */
  if (!keys || !vals) { free(keys); free(vals); return -1; }
  for (int64_t i = 0; i < n; i++) {
    uint64_t s = mix64((uint64_t)target[i]) & (cap - 1);
    for (;;) {
      if (vals[s] == 0) { keys[s] = target[i]; vals[s] = target_idx[i] + 1; break; }
      if (keys[s] == target[i]) break; /* first wins */
      s = (s + 1) & (cap - 1);
    }
  }
  for (int64_t i = 0; i < n1; i++) {
    uint64_t s = mix64((uint64_t)query[i]) & (cap - 1);
    out[i] = 0;
    for (;;) {
      if (vals[s] == 0) break;
      if (keys[s] == query[i]) { out[i] = vals[s]; break; }
      s = (s + 1) & (cap - 1);
    }
  }
  free(keys); free(vals);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * spcount: histogram of idx, negative indices ignored.  Follows backend/others/count_cpu.cpp:7-23.
 * ------------------------------------------------------------------------------------------- */
void oracle_count(int64_t n, const int32_t *idx, int32_t s, int32_t *out) {
  memset(out, 0, (size_t)s * sizeof(int32_t));
  for (int64_t i = 0; i < n; i++)
    if (idx[i] >= 0) out[idx[i]]++;
}

/* ---------------------------------------------------------------------------------------------
 * spvoxelize forward: scatter-mean, out[idx[i]] += in[i] / (float)counts[idx[i]], voxels visited
 * in ascending i.  Follows backend/voxelize/voxelize_cpu.cpp:7-25 (serial outer loop; the CUDA
 * twin voxelize_cuda.cu:12-25 does the same with fp atomics in arbitrary order).
 * ------------------------------------------------------------------------------------------- */
void oracle_voxelize_fwd(int64_t n, int64_t c, const float *in, const int32_t *idx,
                         const int32_t *counts, int64_t n1, float *out) {
  memset(out, 0, (size_t)(n1 * c) * sizeof(float));
  for (int64_t i = 0; i < n; i++) {
    int32_t pos = idx[i];
    if (pos < 0 || counts[pos] == 0) continue;
    float cnt = (float)counts[pos];
    LINK_OMP_FOR                                   /* voxelize_cpu.cpp:17: the INNER loop is the parallel one */
    for (int64_t j = 0; j < c; j++) out[pos * c + j] += in[i * c + j] / cnt;
  }
}

/* spvoxelize backward: bottom[i] = top[idx[i]] / counts[idx[i]].  voxelize_cpu.cpp:27-43. */
void oracle_voxelize_bwd(int64_t n, int64_t c, const float *top, const int32_t *idx,
                         const int32_t *counts, float *bottom) {
  memset(bottom, 0, (size_t)(n * c) * sizeof(float));
  for (int64_t i = 0; i < n; i++) {
    int32_t pos = idx[i];
    if (pos < 0 || counts[pos] == 0) continue;
    float cnt = (float)counts[pos];
    LINK_OMP_FOR                                   /* voxelize_cpu.cpp:36 */
    for (int64_t j = 0; j < c; j++) bottom[i * c + j] = top[pos * c + j] / cnt;
  }
}

/* ---------------------------------------------------------------------------------------------
 * spdevoxelize forward with LinK's neighbourhood size: out[i] = sum_{k<K} w[i,k]*feat[ind[i,k]],
 * ind<0 skipped, k ascending, K = r^3.  Follows backend/devoxelize/devoxelize_cuda.cu:11-34.
 * (The CPU twin devoxelize_cpu.cpp:19-24 hard-wires K=8 and is therefore only valid for r=2;
 *  with K=8 this restatement is bit-identical to it -- tests/test_oracle_golden.py.)
 * ------------------------------------------------------------------------------------------- */
void oracle_devoxelize_fwd(int64_t nq, int64_t c, int64_t K, const int32_t *ind, const float *w,
                           const float *feat, float *out) {
  LINK_OMP_FOR                                     /* devoxelize_cpu.cpp:16: the OUTER loop is the parallel one */
  for (int64_t i = 0; i < nq; i++) {
    for (int64_t j = 0; j < c; j++) {
      float acc = 0.f;
      for (int64_t k = 0; k < K; k++) {
        int32_t q = ind[i * K + k];
        float cur = (q >= 0) ? feat[(int64_t)q * c + j] : 0.f;
        acc += w[i * K + k] * cur;
      }
      out[i * c + j] = acc;
    }
  }
}

/* spdevoxelize backward: bottom[ind[i,k]] += w[i,k]*top[i], ind<0 skipped.
 * Follows backend/devoxelize/devoxelize_cuda.cu:37-59 (the CPU twin devoxelize_cpu.cpp:43-55 is
 * defective: it reads top_grad[ind] and writes at ind==-1; SURVEY.md section 8c defect 2). */
void oracle_devoxelize_bwd(int64_t nq, int64_t n, int64_t c, int64_t K, const int32_t *ind,
                           const float *w, const float *top, float *bottom) {
  memset(bottom, 0, (size_t)(n * c) * sizeof(float));
  for (int64_t i = 0; i < nq; i++)
    for (int64_t k = 0; k < K; k++) {
      int32_t q = ind[i * K + k];
      if (q < 0) continue;
      LINK_OMP_FOR                                 /* devoxelize_cpu.cpp:49 */
      for (int64_t j = 0; j < c; j++) bottom[(int64_t)q * c + j] += w[i * K + k] * top[i * c + j];
    }
}

/* ---------------------------------------------------------------------------------------------
 * Block coordinates: floor division of x,y,z by s, batch column untouched.
 * Follows segmentation/core/models/utils.py:45 (torch.div(..., rounding_mode='floor').int()).
 * ------------------------------------------------------------------------------------------- */
static inline int32_t floordiv(int32_t a, int32_t b) {
  int32_t q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

void oracle_block_coords(int64_t n, const int32_t *coords, int32_t s, int32_t *out) {
  for (int64_t i = 0; i < n; i++) {
    for (int j = 0; j < 3; j++) out[4 * i + j] = floordiv(coords[4 * i + j], s);
    out[4 * i + 3] = coords[4 * i + 3];
  }
}

/* ---------------------------------------------------------------------------------------------
 * torch.unique(rows, dim=0): sorted (signed lexicographic over columns 0..3) unique rows.
 * Follows the call site segmentation/core/models/utils.py:47 (PyTorch semantics: ATen
 * unique_dim sorts rows lexicographically, then drops consecutive duplicates).  Returns M.
 * ------------------------------------------------------------------------------------------- */
static int cmp_row4(const void *a, const void *b) {
  const int32_t *x = (const int32_t *)a, *y = (const int32_t *)b;
  for (int j = 0; j < 4; j++) {
    if (x[j] < y[j]) return -1;
    if (x[j] > y[j]) return 1;
  }
  return 0;
}

int64_t oracle_unique_rows(int64_t n, const int32_t *rows, int32_t *out) {
  if (n == 0) return 0;
  int32_t *tmp = (int32_t *)malloc((size_t)n * 4 * sizeof(int32_t));
  memcpy(tmp, rows, (size_t)n * 4 * sizeof(int32_t));
  qsort(tmp, (size_t)n, 4 * sizeof(int32_t), cmp_row4);
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    if (i == 0 || cmp_row4(tmp + 4 * i, tmp + 4 * (i - 1)) != 0) {
      memcpy(out + 4 * m, tmp + 4 * i, 4 * sizeof(int32_t));
      m++;
    }
  }
  free(tmp);
  return m;
}

/* ---------------------------------------------------------------------------------------------
 * voxel_to_aux (index half): block coords -> unique -> hash -> query -> count.
 * Follows segmentation/core/models/utils.py:44-51 (== detection/det3d/models/utils/ts_elk.py:68-75).
 * Outputs: small_c[M,4] (caller allocates n rows), idx_query[n] (int64), counts (caller allocates n).
 * Returns M (or -1 on allocation failure).
 * ------------------------------------------------------------------------------------------- */
int64_t oracle_voxel_to_aux_index(int64_t n, const int32_t *coords, int32_t s, int32_t *small_c,
                                  int64_t *idx_query, int32_t *counts) {
  int32_t *xc = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(int32_t));
  int64_t *lh = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  int64_t *sh = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  int64_t *ar = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  if (!xc || !lh || !sh || !ar) { free(xc); free(lh); free(sh); free(ar); return -1; }
  oracle_block_coords(n, coords, s, xc);                 /* utils.py:45 */
  oracle_hash(n, xc, lh);                                /* utils.py:46 */
  int64_t m = oracle_unique_rows(n, xc, small_c);        /* utils.py:47 */
  oracle_hash(m, small_c, sh);                           /* utils.py:48 */
  for (int64_t i = 0; i < m; i++) ar[i] = i;             /* query.py:16-18 arange */
  oracle_hash_query(n, lh, m, sh, ar, idx_query);        /* utils.py:50 */
  for (int64_t i = 0; i < n; i++) idx_query[i] -= 1;     /* query.py:32 */
  memset(counts, 0, (size_t)m * sizeof(int32_t));
  for (int64_t i = 0; i < n; i++)                        /* utils.py:51, count_cpu.cpp:7-23 */
    if (idx_query[i] >= 0) counts[idx_query[i]]++;
  free(xc); free(lh); free(sh); free(ar);
  return m;
}

/* ---------------------------------------------------------------------------------------------
 * aux_to_voxel (index half): neighbour map idx_query[M,K] of the r^3 neighbour blocks.
 * Follows segmentation/core/models/utils.py:65-73: kernel-offset hashes [K,M] -> query against
 * the block hashes -> transpose to [M,K].  `offsets` is get_kernel_offsets(r) (int32[K,3]).
 * ------------------------------------------------------------------------------------------- */
int oracle_neighbor_index(int64_t m, const int32_t *small_c, int64_t K, const int32_t *offsets,
                          int32_t *nbr /* [M,K] */) {
  int64_t *nh = (int64_t *)malloc((size_t)(m * K > 0 ? m * K : 1) * sizeof(int64_t));
  int64_t *sh = (int64_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int64_t));
  int64_t *ar = (int64_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int64_t));
  int64_t *res = (int64_t *)malloc((size_t)(m * K > 0 ? m * K : 1) * sizeof(int64_t));
  if (!nh || !sh || !ar || !res) { free(nh); free(sh); free(ar); free(res); return -1; }
  oracle_kernel_hash(m, K, small_c, offsets, nh, 0);     /* utils.py:66-68 */
  oracle_hash(m, small_c, sh);                           /* utils.py:70 */
  for (int64_t i = 0; i < m; i++) ar[i] = i;
  oracle_hash_query(m * K, nh, m, sh, ar, res);          /* utils.py:72 */
  for (int64_t k = 0; k < K; k++)                        /* utils.py:73 transpose */
    for (int64_t i = 0; i < m; i++) nbr[i * K + k] = (int32_t)(res[k * m + i] - 1);
  free(nh); free(sh); free(ar); free(res);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * aux_to_voxel (feature half).  Follows segmentation/core/models/utils.py:75-82:
 *   f = cat[F, 1] * counts ; weights = (nbr != -1) ; new = devoxelize(f, nbr, weights, r) ;
 *   new = new[:, :-1] / new[:, -1:] ; out = new[idx].
 * small_f[M,W] are the block means produced by voxel_to_aux.  out[N,W].
 * ------------------------------------------------------------------------------------------- */
int oracle_aux_to_voxel_feats(int64_t n, int64_t m, int64_t W, int64_t K, const float *small_f,
                              const int32_t *counts, const int32_t *nbr, const int64_t *idx,
                              float *out) {
  int64_t c = W + 1;
  float *f = (float *)malloc((size_t)(m * c > 0 ? m * c : 1) * sizeof(float));
  float *w = (float *)malloc((size_t)(m * K > 0 ? m * K : 1) * sizeof(float));
  float *nf = (float *)malloc((size_t)(m * c > 0 ? m * c : 1) * sizeof(float));
  if (!f || !w || !nf) { free(f); free(w); free(nf); return -1; }
  for (int64_t i = 0; i < m; i++) {
    float cnt = (float)counts[i];
    for (int64_t j = 0; j < W; j++) f[i * c + j] = small_f[i * W + j] * cnt; /* utils.py:75-76 */
    f[i * c + W] = 1.0f * cnt;
    for (int64_t k = 0; k < K; k++) w[i * K + k] = (nbr[i * K + k] == -1) ? 0.f : 1.f; /* :77-78 */
  }
  oracle_devoxelize_fwd(m, c, K, nbr, w, f, nf);          /* utils.py:79 */
  for (int64_t i = 0; i < n; i++) {                       /* utils.py:80,82 */
    int64_t b = idx[i];
    for (int64_t j = 0; j < W; j++) out[i * W + j] = nf[b * c + j] / nf[b * c + W];
  }
  free(f); free(w); free(nf);
  return 0;
}

#ifdef __cplusplus
}
#endif
