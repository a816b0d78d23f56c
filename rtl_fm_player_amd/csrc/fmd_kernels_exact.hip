/* Exact-arithmetic kernels (FMD_MATH_EXACT): bit-identical PCM.  Built with
 * -ffp-contract=off -fno-slp-vectorize (see Makefile). */
#define FMD_BUILD_EXACT 1
#include "fmd_kernels.inc"
