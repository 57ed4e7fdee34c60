// link_amd/csrc/elk_tiles_bf16.hip -- tile form of R_core on the general layout with bf16 feature rows at the kernel boundary
// (elk_tiles_impl.h; fp32 everywhere inside: tables, sums, LayerNorm).
#define DC_IO 2
#define DC_IO_NS elkt_bf16
#include "elk_tiles_impl.h"
#include "elk_tiles_dispatch.h"
