/* Fast-arithmetic kernels with the matrix pipe (FMD_MATH_FAST_MFMA): PCM within +-1 LSB; the /8 decimator runs as
 * int8 MFMAs beside the vector ALU (fmd_kernels.inc, decimate_mfma). */
#define FMD_BUILD_EXACT 0
#define FMD_BUILD_MFMA 1
#include "fmd_kernels.inc"
