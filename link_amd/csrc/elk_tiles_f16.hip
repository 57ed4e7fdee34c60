// link_amd/csrc/elk_tiles_f16.hip -- tile form of R_core on the general layout with fp16 feature rows at the kernel boundary
// (elk_tiles_impl.h; fp32 everywhere inside: tables, sums, LayerNorm).
#define DC_IO 1
#define DC_IO_NS elkt_f16
#include "elk_tiles_impl.h"
#include "elk_tiles_dispatch.h"
