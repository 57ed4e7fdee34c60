// oracle/ref_bind.cpp -- binding shim for oracle/_ref/ (TEST INFRASTRUCTURE ONLY).
//
// Exposes the reference's own torchsparse-u CPU ops -- compiled by oracle/build_ref.py from the
// sources WHERE THEY LIE under /root/reference/segmentation/torchsparse-u/torchsparse/backend
// (hash/hash_cpu.cpp, others/count_cpu.cpp, voxelize/voxelize_cpu.cpp,
// devoxelize/devoxelize_cpu.cpp) -- under the names the reference's pybind_cpu.cpp:12-23 uses.
// Nothing from the reference is copied: this file only includes the reference headers by path.
//
// NOT built from the reference: others/query_cpu.cpp and hashmap/hashmap_cpu.* need Google
// sparsehash (<google/dense_hash_map>, query_cpu.cpp:6), which this image lacks -> unbuildable
// here.  `hash_query_cpu` below is therefore the ORACLE's restatement (oracle/link_oracle.c
// oracle_hash_query: insert-if-absent / find, value idx+1, 0 = miss), exported under the
// reference's name only so that the reference's Python layer can be imported for generating
// golden vectors.  Every fixture that went through it says so in its metadata.
#include <torch/extension.h>

#include "hash/hash_cpu.h"
#include "others/count_cpu.h"
#include "voxelize/voxelize_cpu.h"
#include "devoxelize/devoxelize_cpu.h"

extern "C" int oracle_hash_query(int64_t n1, const int64_t *query, int64_t n, const int64_t *target,
                                 const int64_t *target_idx, int64_t *out);

static at::Tensor hash_query_oracle(const at::Tensor hash_query, const at::Tensor hash_target,
                                    const at::Tensor idx_target) {
  auto q = hash_query.contiguous();
  auto t = hash_target.contiguous();
  auto ti = idx_target.contiguous();
  at::Tensor out = torch::zeros({q.size(0)}, at::device(q.device()).dtype(at::ScalarType::Long));
  oracle_hash_query(q.size(0), q.data_ptr<int64_t>(), t.size(0), t.data_ptr<int64_t>(),
                    ti.data_ptr<int64_t>(), out.data_ptr<int64_t>());
  return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("hash_cpu", &hash_cpu);                                  // reference
  m.def("kernel_hash_cpu", &kernel_hash_cpu);                    // reference
  m.def("count_cpu", &count_cpu);                                // reference
  m.def("voxelize_forward_cpu", &voxelize_forward_cpu);          // reference
  m.def("voxelize_backward_cpu", &voxelize_backward_cpu);        // reference
  m.def("devoxelize_forward_cpu", &devoxelize_forward_cpu);      // reference (K hard-wired to 8)
  m.def("devoxelize_backward_cpu", &devoxelize_backward_cpu);    // reference (defective, unused)
  m.def("hash_query_cpu", &hash_query_oracle);                   // ORACLE restatement (see header)
  m.attr("query_is_oracle_restatement") = true;
}
