// Second re-compilation of the fused attention kernels: two-plane f16 operands (SG_PREC_F16X2, common.h h2_t) -- every score and every
// context element is three f16 MFMAs (hi.hi + lo.hi + hi.lo) into the f32 accumulators of the same kernel bodies.  Defines attention_h2_impl.
#define SG_ATTN_H2 1
#include "attention.hip"
