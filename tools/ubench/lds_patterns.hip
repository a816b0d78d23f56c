// What does SQ_LDS_BANK_CONFLICT count on gfx950?  One kernel per LDS read pattern of the fused kernel's stages, the
// same number of read instructions each, so that the counter (and the time) can be read per pattern:
//   hipcc --offload-arch=gfx950 -O3 -o lds_patterns lds_patterns.hip
//   ./lds_patterns                                             (time per pattern)
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d out -- ./lds_patterns
// Patterns (byte address of lane l, read k):
//   0  b32   4 l + 4 k                     ideal 4-byte reads
//   1  b32   8 l + 4 k                     two lanes per bank
//   2  b64   8 l + 8 k                     ideal 8-byte reads
//   3  b128  16 l + 16 k                   ideal 16-byte reads
//   4  b128  32 l + 16 k                   stage C's window / the NFM resampler's window: a lane owns 8 floats
//   5  b64   8 (i_q + k), q = 4 (l & 15) + (l >> 4), i_q = floor(6.25 q)      stereo stage D, {L+R, L-R} pairs, frame per lane
//   6  b32   4 (i_f + 8 (l & 3) + k), f = l >> 2, i_f = floor(6.25 f)         mono stage D, four lanes per frame (resample_mono_oct)
//   7  b32   4 (i_l + k), i_l = floor(6.25 l)                                  mono stage D, frame per lane (round 2)
//   8  2xb64 pattern 5 with ds_read2_b64 (two consecutive pairs per instruction: what the stereo stage D issues)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int PAT>
__global__ void __launch_bounds__(256) lds_pattern(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[4 * 4096];                     // 16 KB per wave
  for (int i = threadIdx.x; i < 4 * 4096; i += blockDim.x) buf[i] = (float)(i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned base = (unsigned)(size_t)(&buf[4096 * wave]);
  int q;
  switch (PAT) {
    case 0: base += 4 * lane; break;
    case 1: base += 8 * lane; break;
    case 2: base += 8 * lane; break;
    case 3: base += 16 * lane; break;
    case 4: base += 32 * lane; break;
    case 5: case 8: q = 4 * (lane & 15) + (lane >> 4); base += 8 * ((25 * q) >> 2); break;
    case 6: q = lane >> 2; base += 4 * (((25 * q) >> 2) + 8 * (lane & 3)); break;
    case 7: base += 4 * ((25 * lane) >> 2); break;
    case 9: q = lane >> 2; base += 4 * (((25 * q + 3) >> 2) + 160 - 8 * (lane & 3)); break;              /* high side of the mono window: x[i - 8 o - k] */
    case 10: q = lane >> 2; base += (q < 2) ? 4 * (((25 * q) >> 2) + 8 * (lane & 3)) : 0; break;        /* last pass: 2 frames live, the other lanes read one address */
    case 12: base += 16 * lane; break;                                                               /* ds_read2_b64, ideal */
    case 13: q = (lane >> 5) + 4 * ((lane >> 2) & 7); base += 4 * (((25 * q + 1) >> 2) + 8 * (lane & 3)); break;   /* mono stage D, every 4th frame per cycle, other phase */
    case 14: q = (lane >> 5) + 4 * ((lane >> 2) & 7); base += 4 * (((25 * q + 2) >> 2) + 160 - 8 * (lane & 3)); break;   /* ... high side, third phase */
    case 11: q = 16 + (lane >> 2); base += 4 * (((25 * q + 1) >> 2) + 8 * (lane & 3)); break;           /* second pass of an iteration (frames 16 .. 31), other phase */
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; it++) {
    // eight reads per iteration at immediate offsets, like the unrolled stages
    if constexpr (PAT == 0 || PAT == 1 || PAT == 6 || PAT == 7 || (PAT >= 9 && PAT != 12)) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if constexpr (PAT == 9 || PAT == 14) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(base), "n"(4 * (7 - k)));
        else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(base), "n"(4 * k));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; k++) acc.x += v[k];
    } else if constexpr (PAT == 2 || PAT == 5) {
      f2 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[k]) : "v"(base), "n"(8 * k));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; k++) { acc.x += v[k].x; acc.y += v[k].y; }
    } else {
      f4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if constexpr (PAT == 8 || PAT == 12) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v[k]) : "v"(base), "n"(2 * k), "n"(2 * k + 1));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(base), "n"(16 * k));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; k++) acc += v[k];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int PAT>
static void run(float *d, const char *what, int bytes) {
  const int iters = 2000, grid = 256 * 3;
  hipLaunchKernelGGL(lds_pattern<PAT>, dim3(grid), dim3(256), 0, 0, d, 10);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(lds_pattern<PAT>, dim3(grid), dim3(256), 0, 0, d, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double insts = (double)grid * 4 * iters * 8;                    // wave instructions
  printf("pattern %d %-62s %7.3f ms  %6.2f ns per wave-read  %6.1f B/clk/CU at 2.4 GHz\n", PAT, what, ms,
         ms * 1e6 / (insts / (256.0 * 4)) , insts * 64 * bytes / 256.0 / (ms * 1e-3 * 2.4e9));
}

int main() {
  float *d; CHECK(hipMalloc(&d, 256 * 3 * 256 * sizeof(float)));
  run<0>(d, "b32 ideal", 4);
  run<1>(d, "b32 stride 8 B", 4);
  run<2>(d, "b64 ideal", 8);
  run<3>(d, "b128 ideal", 16);
  run<4>(d, "b128 stride 32 B (stage C / NFM window)", 16);
  run<5>(d, "b64 stereo stage D (every 4th frame, 25 samples apart)", 8);
  run<6>(d, "b32 mono stage D, four lanes per frame", 4);
  run<7>(d, "b32 mono stage D, frame per lane", 4);
  run<8>(d, "ds_read2_b64 stereo stage D", 16);
  run<9>(d, "b32 mono stage D, four lanes per frame, high side (descending)", 4);
  run<10>(d, "b32 mono stage D, last pass (2 of 16 frames live)", 4);
  run<11>(d, "b32 mono stage D, frames 16..31", 4);
  run<12>(d, "ds_read2_b64 ideal", 16);
  run<13>(d, "b32 mono stage D, every 4th frame per LDS cycle, low side", 4);
  run<14>(d, "b32 mono stage D, every 4th frame per LDS cycle, high side", 4);
  return 0;
}
