// link_amd/csrc/dense_fused_bf16.hip -- the fused dense-cell kernels with bf16 feature rows at the kernel
// boundary (dense_fused_impl.h; fp32 everywhere inside).
#define DC_IO 2
#define DC_IO_NS dcio_bf16
#include "dense_fused_impl.h"
