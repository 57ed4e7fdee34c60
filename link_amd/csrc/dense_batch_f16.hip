// link_amd/csrc/dense_batch_f16.hip -- the batch entry point's kernels with fp16 feature rows at the kernel boundary
// (dense_batch_impl.h; fp32 everywhere inside).  The host side is dense_batch.hip.
#define DC_IO 1
#define DC_IO_NS dcb_f16
#include "dense_batch_impl.h"
