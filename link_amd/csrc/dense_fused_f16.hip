// link_amd/csrc/dense_fused_f16.hip -- the fused dense-cell kernels with fp16 feature rows at the kernel
// boundary (dense_fused_impl.h; fp32 everywhere inside).
#define DC_IO 1
#define DC_IO_NS dcio_f16
#include "dense_fused_impl.h"
