// The f16-operand instantiation of the fused attention kernels (SG_PREC_F16): attention.hip compiled with the operand kind
// switched, as its own translation unit so the two builds run in parallel.  Defines sg::attention_f16_impl.
#define SG_ATTN_F16 1
#include "attention.hip"
