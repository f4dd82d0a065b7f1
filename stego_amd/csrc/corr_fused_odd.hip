// The odd-K instantiations of corr_fused_kernel (C = 384 / 768) and their launch function: corr_fused.hip compiled as part 1.
#define STEGO_FUSED_PART 1
#include "corr_fused.hip"
