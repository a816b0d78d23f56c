// What an instruction costs at the package power cap: every SIMD of the chip streams ONE instruction class (W waves per SIMD), the host counts
// wave-instructions per second; tools/power_price.sh samples rocm-smi beside it.  energy per wave-instruction <= package power / rate (an upper
// bound: the package draws ~280 W doing nothing).   hipcc --offload-arch=gfx950 -O2 -o power_price power_price.hip ; ./power_price <kind> [waves per SIMD] [seconds]
//   kinds: mfma (v_mfma_i32_16x16x64_i8), fma (v_fma_f32), pkfma (v_pk_fma_f32), mix (1 MFMA : 5 v_fma_f32 : the stereo kernel's ratio), nop (s_nop),
//          lds128 / lds64 / lds32 (ds_read_b128 / _b64 / _b32 of the lane's own aligned word: what an MFMA operand read costs), ldsw128 (ds_write_b128),
//          perm (v_perm_b32), cvt (v_cvt_rpi_i32_f32), dpp (v_mov_b32 row_shr:1)
//   round 6: mfmar (the same MFMA on LIVE operands: four different random register quads per side, rotating), mfma32 / mfma32r (v_mfma_i32_32x32x32_i8: the same
//          multiply-adds per 1024 outputs with half the operand fetches per multiply-add; constant / live operands), lds2r32 (ds_read2_b32 at a 4-byte-aligned,
//          not 8-byte-aligned address: half of what a 4-aligned 16-byte operand costs), lds128u4 (ds_read_b128 at such an address), addpp (v_add_u32 with a DPP source)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int UNROLL = 16;
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float *sink) {
  i4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
  i4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
  float f[8];
  f2 p[4];
  for (int i = 0; i < 8; i++) f[i] = (float)(threadIdx.x + i) * 1e-3f;
  for (int i = 0; i < 4; i++) p[i] = f2{f[i], f[i + 4]};
  __shared__ i4 lbuf[256 * 4];
  for (int i = threadIdx.x; i < 256 * 4; i += 256) lbuf[i] = i4{i, i + 1, i + 2, i + 3};
  __syncthreads();
  typedef __attribute__((address_space(3))) i4 lds_i4;
  lds_i4 *lp = (lds_i4 *)&lbuf[threadIdx.x];
  i4 ld[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  int pi[4] = {(int)threadIdx.x, 77, 99, 1234567};
  typedef int i16v __attribute__((ext_vector_type(16)));
  i16v big[2] = {};
  /* live operands: eight quads of hashed bytes (different per lane and quad) */
  i4 ra[4], rb[4];
  {
    unsigned h = 2654435761u * (threadIdx.x + 977u * blockIdx.x + 1u);
    for (int i = 0; i < 4; i++) {
      int t[8];
      for (int j = 0; j < 8; j++) { h = h * 1664525u + 1013904223u; t[j] = (int)(h ^ (h >> 13)); }
      ra[i] = i4{t[0], t[1], t[2], t[3]};
      rb[i] = i4{t[4], t[5], t[6], t[7]};
    }
  }
  typedef __attribute__((address_space(3))) char lds_c;
  lds_c *lp4 = (lds_c *)&lbuf[threadIdx.x] + 4 * (1 + (threadIdx.x & 2));     /* 4 or 12 bytes into the lane's word: 4-aligned, never 8- or 16-aligned */
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if constexpr (KIND == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[u & 3]) : "v"(lp), "n"((u & 3) * 4096));
      if constexpr (KIND == 6) { long long t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(lp), "n"((u & 3) * 4096)); ld[u & 3].x ^= (int)t; }
      if constexpr (KIND == 7) { int t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(lp), "n"((u & 3) * 4096)); ld[u & 3].x ^= t; }
      if constexpr (KIND == 8) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(lp), "v"(ld[u & 3]), "n"((u & 3) * 4096) : "memory");
      if constexpr (KIND == 9) pi[u & 3] = __builtin_amdgcn_perm(pi[u & 3], pi[(u + 1) & 3], 0x05010602u);
      if constexpr (KIND == 10) asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(pi[u & 3]) : "v"(f[u & 7]));
      if constexpr (KIND == 11) pi[u & 3] = __builtin_amdgcn_update_dpp(0, pi[(u + 1) & 3], 0x111, 0xf, 0xf, true);
      if constexpr (KIND == 0) acc[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[u & 3], 0, 0, 0);
      if constexpr (KIND == 1) f[u & 7] = __builtin_fmaf(f[u & 7], 1.0001f, 0.5f);
      if constexpr (KIND == 2) p[u & 3] = __builtin_elementwise_fma(p[u & 3], f2{1.0001f, 0.9999f}, f2{0.5f, 0.25f});
      if constexpr (KIND == 3) {
        if ((u % 6) == 0) acc[(u / 6) & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[(u / 6) & 3], 0, 0, 0);
        else f[u & 7] = __builtin_fmaf(f[u & 7], 1.0001f, 0.5f);
      }
      if constexpr (KIND == 4) asm volatile("s_nop 0");
      if constexpr (KIND == 12) acc[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ra[u & 3], rb[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
      if constexpr (KIND == 13) big[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, big[u & 1], 0, 0, 0);
      if constexpr (KIND == 14) big[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[u & 3], rb[(u >> 2) & 3], big[u & 1], 0, 0, 0);
      if constexpr (KIND == 15) { long long t; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(lp4), "n"((u & 3) * 32), "n"((u & 3) * 32 + 1)); ld[u & 3].x ^= (int)t; }
      if constexpr (KIND == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[u & 3]) : "v"(lp4), "n"((u & 3) * 2048));
      if constexpr (KIND == 17) pi[u & 3] += __builtin_amdgcn_update_dpp(0, pi[(u + 1) & 3], 0x101, 0xf, 0xf, true);
    }
  }
  if constexpr ((KIND >= 5 && KIND <= 8) || KIND == 15 || KIND == 16) asm volatile("s_waitcnt lgkmcnt(0)");
  for (int i = 0; i < 16; i++) pi[i & 3] ^= big[0][i] + big[1][i];
  float s = 0.f;
  for (int i = 0; i < 4; i++) s += (float)(ld[i].x + ld[i].y + ld[i].z + ld[i].w + pi[i]);
  for (int i = 0; i < 8; i++) s += f[i];
  for (int i = 0; i < 4; i++) s += p[i].x + p[i].y + (float)(acc[i].x + acc[i].y + acc[i].z + acc[i].w);
  if (s == 12345.678f) sink[0] = s;
}
int main(int argc, char **argv) {
  const char *kind = argc > 1 ? argv[1] : "mfma";
  const int W = argc > 2 ? atoi(argv[2]) : 2;
  const double seconds = argc > 3 ? atof(argv[3]) : 6.0;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * W;                 /* 256 threads = 4 waves = one per SIMD; W blocks per CU */
  float *sink;
  (void)hipMalloc(&sink, 4);
  const int iters = 20000;
  auto launch = [&]() {
    if (!strcmp(kind, "mfma")) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "fma")) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "pkfma")) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "mix")) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "lds128")) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "lds64")) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "lds32")) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "ldsw128")) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "perm")) hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "cvt")) hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "dpp")) hipLaunchKernelGGL(k<11>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "mfmar")) hipLaunchKernelGGL(k<12>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "mfma32")) hipLaunchKernelGGL(k<13>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "mfma32r")) hipLaunchKernelGGL(k<14>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "lds2r32")) hipLaunchKernelGGL(k<15>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "lds128u4")) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else if (!strcmp(kind, "addpp")) hipLaunchKernelGGL(k<17>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, iters, sink);
  };
  for (int i = 0; i < 3; i++) launch();
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < seconds) {
    for (int i = 0; i < 4; i++) launch();
    (void)hipDeviceSynchronize();
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double winstr = (double)launches * blocks * 4 * (double)iters * UNROLL;      /* wave-instructions of the class */
  printf("kind %s W %d: %.3e wave-instructions/s over %.1f s (%d CUs); cycles per instruction and SIMD at 2.4 GHz: %.2f\n", kind, W, winstr / el, el, cus,
         2.4e9 * el / (winstr / (cus * 4.0)));
  return 0;
}
