// Typed buffer loads (MTBUF, 8_8_8_8 USCALED / SSCALED) as the u8 -> f32 converter of stage A:
// do they convert on gfx950, and what does a stage-A-shaped loop cost with them against
// raw 16-byte loads + one v_cvt per byte?
//   hipcc --offload-arch=gfx950 -O3 -o fmtload fmtload.hip && ./fmtload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// the LLVM intrinsic itself (clang has no builtin for it); format = dfmt | nfmt << 4, dfmt 10 = 8_8_8_8, nfmt 2 = USCALED, 3 = SSCALED
__device__ f4 tbuf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff, int fmt, int aux) __asm("llvm.amdgcn.raw.ptr.tbuffer.load.v4f32");
constexpr int FMT_U = 10 | (2 << 4), FMT_S = 10 | (3 << 4);
constexpr uint32_t RSRC_RAW = 0x00020000u, RSRC_FMT = 0x00020FACu;   // dst_sel x, y, z, w = R, G, B, A

__global__ void k_check(const uint8_t *p, float *out, int n) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, n, RSRC_FMT);
  const int i = threadIdx.x + blockIdx.x * blockDim.x;         // one dword each, the last few past the end
  const f4 u = tbuf_load4(r, 4 * i, 0, FMT_U, 0), s = tbuf_load4(r, 4 * i, 0, FMT_S, 0);
  for (int c = 0; c < 4; c++) { out[8 * i + c] = u[c]; out[8 * i + 4 + c] = s[c]; }
}

template <int MODE>   // 0: raw b128 loads + signed-byte conversions, 1: typed loads (unsigned), 2: typed + packed subtract of 128,
                      // 3: the raw loads alone (words xor-ed together: the HBM side of stage A), 4: mode 0 with the next tile's words
                      //    requested before this tile's arithmetic (the kernel's one-tile-ahead prefetch)
__global__ __launch_bounds__(256, 3) void k_stage_a(const uint8_t *p, float *out, int n_words, int tiles, const float *gt) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float g[16];
  for (int k = 0; k < 16; k++) g[k] = __builtin_amdgcn_readfirstlane(gt[k]);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, n_words * 16, RSRC_RAW);
  __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p), 0, n_words * 16, RSRC_FMT);
  f2 tot = {0.f, 0.f};
  u4v qn[11] = {};
  for (int t = 0; t < tiles; t++) {
    const int m0 = (wave * tiles + t) * 512 + lane * 8;
    const int byte0 = (m0 - 3) * 16;
    f2 pa[8];
#pragma unroll
    for (int r = 0; r < 8; r++) pa[r] = f2{0.f, 0.f};
    u4v q[11];
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < 11; i++) q[i] = __builtin_amdgcn_raw_buffer_load_b128(rr, byte0 + 16 * i, 0, 0);
    }
    if (MODE == 4) {
      if (t == 0) {
#pragma unroll
        for (int i = 0; i < 11; i++) qn[i] = __builtin_amdgcn_raw_buffer_load_b128(rr, byte0 + 16 * i, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 11; i++) q[i] = qn[i];
      if (t + 1 < tiles) {
#pragma unroll
        for (int i = 0; i < 11; i++) qn[i] = __builtin_amdgcn_raw_buffer_load_b128(rr, byte0 + 512 * 16 + 16 * i, 0, 0);
      }
    }
    if (MODE == 3) {
      u4v x = q[0];
#pragma unroll
      for (int i = 1; i < 11; i++) x ^= q[i];
      tot += f2{__builtin_bit_cast(float, x.x ^ x.y), __builtin_bit_cast(float, x.z ^ x.w)};
      continue;
    }
#pragma unroll
    for (int wd = 0; wd < 11; wd++) {
      f4 f[4];
      if (MODE == 0 || MODE == 4) {
        uint32_t kx = 0x80808080u;
        asm volatile("" : "+s"(kx));
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const uint32_t d = q[wd][c] ^ kx;
          f[c] = f4{(float)(int8_t)(d & 0xff), (float)(int8_t)((d >> 8) & 0xff), (float)(int8_t)((d >> 16) & 0xff), (float)(int8_t)(d >> 24)};
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          f[c] = tbuf_load4(rf, byte0 + 16 * wd + 4 * c, 0, FMT_U, 0);
          if (MODE == 2) {
            f2 lo = {f[c].x, f[c].y}, hi = {f[c].z, f[c].w};
            lo = lo - f2{128.f, 128.f}; hi = hi - f2{128.f, 128.f};
            f[c] = f4{lo.x, lo.y, hi.x, hi.y};
          }
        }
      }
#pragma unroll
      for (int tt = 0; tt < 8; tt++) {
        const f2 uiq = (tt & 1) ? f2{f[tt >> 1].z, f[tt >> 1].w} : f2{f[tt >> 1].x, f[tt >> 1].y};
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const int j = 8 * wd + tt - 8 * r;
          if (j >= 0 && j < 32) {
            const float gk = g[j < 16 ? j : 31 - j];
            pa[r] = __builtin_elementwise_fma(uiq, f2{gk, gk}, pa[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) tot += pa[r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot.x + tot.y;
}

int main() {
  // 1. conversion check
  {
    const int n = 1024;
    std::vector<uint8_t> h(n);
    for (int i = 0; i < n; i++) h[i] = (uint8_t)(i * 37 + (i >> 8));
    uint8_t *d; float *o;
    CHECK(hipMalloc(&d, n)); CHECK(hipMalloc(&o, (n / 4 + 64) * 8 * sizeof(float)));
    CHECK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
    const int threads = n / 4 + 64;   // 64 dwords past the end
    k_check<<<(threads + 63) / 64, 64>>>(d, o, n);
    std::vector<float> r((size_t)((threads + 63) / 64 * 64) * 8);
    CHECK(hipMemcpy(r.data(), o, (size_t)threads * 8 * sizeof(float), hipMemcpyDeviceToHost));
    int bad_u = 0, bad_s = 0, bad_oob = 0;
    for (int i = 0; i < threads; i++)
      for (int c = 0; c < 4; c++) {
        const int b = 4 * i + c;
        const float u = r[8 * i + c], s = r[8 * i + 4 + c];
        if (b < n) { bad_u += u != (float)h[b]; bad_s += s != (float)(int8_t)h[b]; }
        else bad_oob += (u != 0.f) || (s != 0.f);
      }
    printf("typed loads: USCALED mismatches %d, SSCALED mismatches %d, past-the-end nonzero %d (u[5]=%g s[5]=%g want %d %d)\n", bad_u, bad_s, bad_oob,
           r[8 * 1 + 1], r[8 * 1 + 5], h[5], (int8_t)h[5]);
  }
  // 2. stage-A-shaped loop
  int cus = 256; hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount;
  const int blocks = cus * 3, waves = blocks * 4, tiles = 40;
  const size_t n_words = (size_t)waves * tiles * 512 + 16;
  uint8_t *d; float *o, *gt;
  CHECK(hipMalloc(&d, n_words * 16)); CHECK(hipMalloc(&o, (size_t)blocks * 256 * 4)); CHECK(hipMalloc(&gt, 64));
  std::vector<uint8_t> h(n_words * 16);
  uint32_t s = 12345; for (auto &b : h) { s = s * 1664525u + 1013904223u; b = s >> 24; }
  CHECK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
  float g[16]; for (int k = 0; k < 16; k++) g[k] = 0.001f * (k + 1);
  CHECK(hipMemcpy(gt, g, 64, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> res[5];
  for (int mode = 0; mode < 5; mode++) {
    auto launch = [&]() {
      if (mode == 0) k_stage_a<0><<<blocks, 256>>>(d, o, (int)n_words, tiles, gt);
      if (mode == 1) k_stage_a<1><<<blocks, 256>>>(d, o, (int)n_words, tiles, gt);
      if (mode == 2) k_stage_a<2><<<blocks, 256>>>(d, o, (int)n_words, tiles, gt);
      if (mode == 3) k_stage_a<3><<<blocks, 256>>>(d, o, (int)n_words, tiles, gt);
      if (mode == 4) k_stage_a<4><<<blocks, 256>>>(d, o, (int)n_words, tiles, gt);
    };
    for (int i = 0; i < 60; i++) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    res[mode].resize((size_t)blocks * 256);
    CHECK(hipMemcpy(res[mode].data(), o, res[mode].size() * 4, hipMemcpyDeviceToHost));
    const double bytes = (double)waves * tiles * 512 * 16;
    printf("stage-A loop mode %d (%s): %.4f ms, %.1f GB/s of IQ, %.2f us per wave-tile\n", mode,
           mode == 0 ? "b128 + cvt" : mode == 1 ? "typed loads" : mode == 2 ? "typed loads + pk_sub" : mode == 3 ? "b128 loads only" : "b128 + cvt, one tile ahead", ms, bytes / ms / 1e6, ms * 1e3 / tiles);
  }
  // mode 2 computes the same sums as mode 0 (u - 128); mode 1 differs by the offset
  double dmax = 0; for (size_t i = 0; i < res[0].size(); i++) { double e = fabs((double)res[0][i] - res[2][i]); if (e > dmax) dmax = e; }
  printf("max |mode0 - mode2| = %g (sum magnitude ~%g)\n", dmax, fabs((double)res[0][0]));
  return 0;
}
