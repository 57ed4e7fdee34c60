// link_amd/csrc/elk_lean_bf16.hip -- lean form of R_core with bf16 feature rows at the kernel boundary (elk_lean_impl.h; fp32
// everywhere inside: X rows, chunk sums, LayerNorm).
#define DC_IO 2
#define DC_IO_NS elkl_bf16
#include "elk_lean_impl.h"
#include "elk_lean_dispatch.h"
