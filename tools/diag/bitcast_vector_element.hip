// hipcc 7.2 (ROCm 7.2.0), gfx950: __builtin_bit_cast applied DIRECTLY to an element of an ext_vector_type value - `__builtin_bit_cast(float, v[r])` -
// reads element 0 for every r: 192 of the 256 values below come out wrong (form 0).  Copy the element to a scalar first and it is right (form 1).
// Met twice in round 4 inside mpx_tile_i8 (csrc/fmd_kernels.inc): the DPP / readlane exchange of the pilot outputs, and the accumulators read as floats.
//   hipcc --offload-arch=gfx950 -O3 -o bitcast_vector_element bitcast_vector_element.hip && ./bitcast_vector_element
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
template <int FORM> __global__ void k(float *out, const int *in) {
  i4 a;
#pragma unroll
  for (int r = 0; r < 4; r++) a[r] = in[r * 64 + threadIdx.x];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float f;
    if constexpr (FORM == 0) f = __builtin_bit_cast(float, a[r]);
    else { const int t = a[r]; f = __builtin_bit_cast(float, t); }
    out[r * 64 + threadIdx.x] = f * 2.0f;
  }
}
int main() {
  float *d; int *din; hipMalloc(&d, 1024); hipMalloc(&din, 1024);
  int hin[256]; float h[256];
  for (int i = 0; i < 256; i++) { float v = 1.0f + i; hin[i] = *(int *)&v; }
  hipMemcpy(din, hin, 1024, hipMemcpyHostToDevice);
  for (int form = 0; form < 2; form++) {
    if (form == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, din); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, din);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; i++) if (h[i] != 2.0f * (1.0f + i)) bad++;
    printf("form %d: %d of 256 wrong\n", form, bad);
  }
}
