// link_amd/csrc/dense_tiles_bf16.hip -- tile form of the fused pre_mix kernel with bf16 feature rows at the kernel boundary
// (dense_tiles_impl.h; fp32 everywhere inside).
#define DC_IO 2
#define DC_IO_NS dcio_bf16
#include "dense_tiles_impl.h"
