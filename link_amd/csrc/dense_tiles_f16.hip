// link_amd/csrc/dense_tiles_f16.hip -- tile form of the fused pre_mix kernel with fp16 feature rows at the kernel boundary
// (dense_tiles_impl.h; fp32 everywhere inside).
#define DC_IO 1
#define DC_IO_NS dcio_f16
#include "dense_tiles_impl.h"
