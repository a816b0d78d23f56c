// Are 16-byte LDS reads at addresses that are not 16-byte aligned correct on gfx950, and what do they cost?
//   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k(int *out, int shift_mul, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = (unsigned char)(i * 7 + 3);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned addr = (unsigned)(size_t)(&buf[0]) + 32 * lane + shift_mul * (lane & 15);   // byte shift = shift_mul * (lane mod 16)
  i4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    i4 v;
    asm volatile("ds_read_b128 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + 0 * it));
    acc += v;
  }
  out[threadIdx.x * 4 + 0] = acc.x; out[threadIdx.x * 4 + 1] = acc.y; out[threadIdx.x * 4 + 2] = acc.z; out[threadIdx.x * 4 + 3] = acc.w;
}
int main() {
  int *d; CHECK(hipMalloc(&d, 64 * 16));
  for (int sm : {0, 4, 1, 2}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sm, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<int> h(256);
    CHECK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int b = 0; b < 16; b++) {
        const int a = 32 * l + sm * (l & 15) + b;
        const unsigned char want = (unsigned char)(a * 7 + 3), got = ((unsigned char *)&h[l * 4])[b];
        bad += want != got;
      }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, d, sm, 20000);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("byte shift %d x (lane mod 16): %d wrong bytes of 1024; 20000 dependent reads per wave, 8 waves per SIMD... %.2f ms\n", sm, bad, ms);
  }
  return 0;
}
