/* Fast-arithmetic kernels (FMD_MATH_FAST): explicit FMAs, PCM within +-1 LSB.
 * Also holds the launcher and the tiling helpers shared by both builds. */
#define FMD_BUILD_EXACT 0
#include "fmd_kernels.inc"
