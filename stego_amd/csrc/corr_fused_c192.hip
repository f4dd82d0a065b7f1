// The C = 192 instantiations of corr_fused_kernel (vit_tiny; even and odd K) and their launch function: corr_fused.hip compiled as part 2.
#define STEGO_FUSED_PART 2
#include "corr_fused.hip"
