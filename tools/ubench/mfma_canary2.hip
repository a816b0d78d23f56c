// Follow-up of mfma_canary.hip: the canary registers ARE former MFMA destinations.  One asm block per iteration:
//   a burst of NB back-to-back bf16 MFMAs accumulating into v[52:55] and v[56:59]; their results are consumed (added
//   into v60); then v[52:59] are re-used as canaries: set to a lane pattern, left alone for a while (s_sleep / VALU on
//   other registers) while the other waves of the SIMD run their own bursts, and compared.  Any mismatch = a register that
//   changed under the wave's feet.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#ifndef MNEM
#define MNEM "v_mfma_f32_16x16x32_bf16"
#endif
template <int NB>
__global__ __launch_bounds__(256, 3) void k(unsigned *bad, int iters, int idle) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned errs = 0, lanebits = 0;
  for (int s = 0; s < (int)((wave * 7 + blockIdx.x * 3) % 13) * 20; s++) __builtin_amdgcn_s_sleep(1);
  const float pat = (float)(lane + 1);
  for (int it = 0; it < iters; it++) {
    unsigned mism;
    asm volatile(
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"
        "v_mov_b32 v44, 0x3c003c00\n\tv_mov_b32 v45, 0x3c003c00\n\tv_mov_b32 v46, 0x3c003c00\n\tv_mov_b32 v47, 0x3c003c00\n\t"
        "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
        "v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"
        "s_mov_b32 s20, %3\n\t"
        "s_nop 4\n\t"
        "1:\n\t"
        MNEM " v[52:55], v[40:43], v[44:47], v[52:55]\n\t"
        MNEM " v[56:59], v[40:43], v[44:47], v[56:59]\n\t"
        "s_sub_u32 s20, s20, 2\n\t"
        "s_cmp_lg_u32 s20, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        "v_add_f32 v60, v52, v56\n\t"          /* consume */
        "v_add_f32 v60, v60, v55\n\t"
        "v_add_f32 v60, v60, v59\n\t"
        "v_mov_b32 v52, %1\n\tv_mov_b32 v53, %1\n\tv_mov_b32 v54, %1\n\tv_mov_b32 v55, %1\n\t"
        "v_mov_b32 v56, %1\n\tv_mov_b32 v57, %1\n\tv_mov_b32 v58, %1\n\tv_mov_b32 v59, %1\n\t"
        "s_mov_b32 s20, %2\n\t"
        "2:\n\t"
        "v_add_f32 v61, v60, v60\n\t"          /* VALU on other registers */
        "v_add_f32 v62, v61, v60\n\t"
        "s_sleep 1\n\t"
        "s_sub_u32 s20, s20, 1\n\t"
        "s_cmp_lg_u32 s20, 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "v_mov_b32 %0, 0\n\t"
        "v_cmp_neq_f32 vcc, v52, %1\n\tv_cndmask_b32 v61, 0, 1, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v53, %1\n\tv_cndmask_b32 v61, 0, 2, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v54, %1\n\tv_cndmask_b32 v61, 0, 4, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v55, %1\n\tv_cndmask_b32 v61, 0, 8, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v56, %1\n\tv_cndmask_b32 v61, 0, 16, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v57, %1\n\tv_cndmask_b32 v61, 0, 32, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v58, %1\n\tv_cndmask_b32 v61, 0, 64, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        "v_cmp_neq_f32 vcc, v59, %1\n\tv_cndmask_b32 v61, 0, -1, vcc\n\tv_or_b32 %0, %0, v61\n\t"
        : "=&v"(mism)
        : "v"(pat), "s"(idle), "s"(NB)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "s20", "vcc", "scc");
    if (mism) { errs++; lanebits |= mism; }
  }
  if (errs) { atomicAdd(&bad[0], errs); atomicOr(&bad[1 + (lane >> 4)], lanebits); }
  if (errs == 0xffffffffu) lds[threadIdx.x] = 1.f;
}

int main() {
  unsigned *d_bad; CHECK(hipMalloc(&d_bad, 64));
  for (int idle = 4; idle <= 256; idle *= 4) {
    CHECK(hipMemset(d_bad, 0, 64));
    CHECK(hipFuncSetAttribute((const void *)k<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 50 * 1024));
    hipLaunchKernelGGL((k<48>), dim3(768), dim3(256), 50 * 1024, 0, d_bad, 3000, idle);
    CHECK(hipDeviceSynchronize());
    unsigned h[16]; CHECK(hipMemcpy(h, d_bad, 64, hipMemcpyDeviceToHost));
    printf(MNEM " idle %3d: iterations with a changed canary: %u; register bits by lane group: %02x %02x %02x %02x\n", idle, h[0], h[1], h[2], h[3], h[4]);
  }
  return 0;
}
