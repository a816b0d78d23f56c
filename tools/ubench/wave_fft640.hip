// Check and time tools/ubench/fmd_fft320.inc on its own: three 90-tap FIRs on one input as an overlap-save
// convolution inside ONE wavefront, N = 640 real samples per block (128 of history + 512 new = one tile of the kernel),
// handled as a 320-point complex FFT of (even, odd) pairs:
//   Z = FFT320(z),  Z'_f[k] = A_f[k] Z[k] + B_f[k] conj(Z[320 - k]),  z'_f = IFFT320(Z'_f),  y_f[2m] = Re z'_f[m], y_f[2m+1] = Im z'_f[m].
//   hipcc --offload-arch=gfx950 -O3 -o wave_fft640 wave_fft640.hip && ./wave_fft640
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmd_fft320.inc"   /* tools/ubench/ (moved from csrc/ in round 3) */

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 640, M = 320, TAPS = 90, NEW = 512, WAVES = 4;

__global__ __launch_bounds__(64 * WAVES, 3) void k_conv(const float *xg, float *yg, const f2 *twg, const f4 *abg, int blocks_per_wave,
                                                       int store_all) {
  __shared__ f2 exs[WAVES][fft320::EXN];
  __shared__ float xs[WAVES][N];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * WAVES + wv;
  f2 *ex = exs[wv];
  float *xl = xs[wv];
  fft320::Twiddles tw;
#pragma unroll
  for (int k = 0; k < 5; k++) tw.t1[k] = twg[lane * 13 + k];
#pragma unroll
  for (int k = 0; k < 8; k++) tw.t2[k] = twg[lane * 13 + 5 + k];
  // the lane that holds the mirrored bins 320 - k of this lane's bins k = k1 + 5 (e + 8 g): register 7 - g there
  const int k1 = lane >> 3, e = lane & 7;
  const int src = k1 >= 1 ? (5 - k1) * 8 + (7 - e) : (e == 0 ? 0 : 8 - e);
  const int from = (lane < 40 ? src : lane) << 2;
  f2 sum = {0.f, 0.f};
  for (int t = 0; t < blocks_per_wave; t++) {
    const float *x = xg + ((size_t)wave * blocks_per_wave + t) * N;
    float *y = yg + ((size_t)wave * blocks_per_wave + t) * 3 * N;
    for (int i = lane; i < N / 4; i += 64) reinterpret_cast<f4 *>(xl)[i] = reinterpret_cast<const f4 *>(x)[i];
    fft320::wave_sync();
    f2 z[5], Z[8];
#pragma unroll
    for (int j = 0; j < 5; j++) z[j] = reinterpret_cast<const f2 *>(xl)[64 * j + lane];
    fft320::forward(z, Z, lane, tw, ex);
    f2 Zc[8];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      float ox = Z[7 - g].x, oy = Z[7 - g].y;         // (opaque scalars: see the note on paired cross-lane operations in DESIGN.md)
      asm volatile("" : "+v"(ox));
      asm volatile("" : "+v"(oy));
      float mx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, ox)));
      float my = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, oy)));
      const f2 own = Z[(8 - g) & 7];
      if (lane == 0) { mx = own.x; my = own.y; }       // k1 = 0, e = 0: bins 40 g mirror onto 40 (8 - g) of the same lane
      Zc[g] = f2{mx, -my};
    }
#pragma unroll 1
    for (int f = 0; f < 3; f++) {
      f2 w[8], o[5];
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const f4 ab = abg[(f * 8 + g) * 64 + lane];
        w[g] = fft320::cmul(Z[g], f2{ab.x, ab.y}) + fft320::cmul(Zc[g], f2{ab.z, ab.w});
      }
      fft320::inverse(w, o, lane, tw, ex);
      if (store_all) {
#pragma unroll
        for (int j = 0; j < 5; j++) reinterpret_cast<f2 *>(y + f * N)[64 * j + lane] = o[j];
      } else {
#pragma unroll
        for (int j = 1; j < 5; j++) sum += o[j];
      }
    }
  }
  if (!store_all) reinterpret_cast<f2 *>(yg)[(size_t)wave * 64 + lane] = sum;
}

int main() {
  std::vector<double> F(3 * TAPS);
  srand(7);
  for (int f = 0; f < 3; f++)
    for (int j = 0; j < 45; j++) { const double v = (rand() / (double)RAND_MAX - 0.5) * 0.1; F[f * TAPS + j] = v; F[f * TAPS + 89 - j] = v; }
  typedef std::complex<double> cd;
  const double PI = 3.14159265358979323846;
  std::vector<f4> ab(3 * 8 * 64, f4{0, 0, 0, 0});
  for (int f = 0; f < 3; f++) {
    std::vector<cd> H(N);
    for (int k = 0; k < N; k++) { cd s = 0; for (int j = 0; j < TAPS; j++) s += F[f * TAPS + j] * std::polar(1.0, -2 * PI * k * j / N); H[k] = s; }
    for (int k = 0; k < M; k++) {
      const cd W = std::polar(1.0, -2 * PI * k / N), I(0, 1);
      const cd P = 0.5 * (H[k] + H[k + M]) + 0.5 * I * std::conj(W) * (H[k] - H[k + M]);
      const cd Q = 0.5 * W * (H[k] - H[k + M]) + 0.5 * I * (H[k] + H[k + M]);
      const cd A = (P - I * Q) / 2.0 / (double)M, B = (P + I * Q) / 2.0 / (double)M;
      const int k1 = k % 5, k2 = k / 5, e = k2 & 7, g = k2 >> 3;        // k = k1 + 5 (e + 8 g) lives in lane k1 8 + e, register g
      ab[(f * 8 + g) * 64 + k1 * 8 + e] = f4{(float)A.real(), (float)A.imag(), (float)B.real(), (float)B.imag()};
    }
  }
  std::vector<f2> tw(64 * 13);
  for (int l = 0; l < 64; l++) {
    for (int k = 0; k < 5; k++) { const cd a = std::polar(1.0, -2 * PI * l * k / 320.0); tw[l * 13 + k] = f2{(float)a.real(), (float)a.imag()}; }
    for (int k = 0; k < 8; k++) { const cd b = std::polar(1.0, -2 * PI * (l & 7) * k / 64.0); tw[l * 13 + 5 + k] = f2{(float)b.real(), (float)b.imag()}; }
  }
  int cus = 256; hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount;
  const int blocks = cus * 3, waves = blocks * WAVES, T = 42;
  const size_t nx = (size_t)waves * T * N;
  std::vector<float> hx(nx);
  uint32_t s = 12345; for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
  float *dx, *dy; f2 *dt; f4 *dab;
  CHECK(hipMalloc(&dx, nx * 4)); CHECK(hipMalloc(&dy, nx * 3 * 4));
  CHECK(hipMalloc(&dt, tw.size() * 8)); CHECK(hipMalloc(&dab, ab.size() * 16));
  CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dt, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dab, ab.data(), ab.size() * 16, hipMemcpyHostToDevice));
  k_conv<<<blocks, 64 * WAVES>>>(dx, dy, dt, dab, T, 1);
  CHECK(hipDeviceSynchronize());
  {
    const size_t blk = (size_t)7 * T + 5;
    std::vector<float> hy(3 * N);
    CHECK(hipMemcpy(hy.data(), dy + blk * 3 * N, 3 * N * 4, hipMemcpyDeviceToHost));
    double worst = 0, rms = 0;
    for (int f = 0; f < 3; f++)
      for (int n = N - NEW; n < N; n++) {
        double r = 0;
        for (int j = 0; j < TAPS; j++) r += F[f * TAPS + j] * (double)hx[blk * N + n - j];
        worst = fmax(worst, fabs(r - (double)hy[f * N + n]));
        rms += r * r;
      }
    printf("wave FFT-640 convolution vs double: max |err| %.3g (rms of the outputs %.3g)\n", worst, sqrt(rms / (3.0 * NEW)));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 40; i++) k_conv<<<blocks, 64 * WAVES>>>(dx, dy, dt, dab, T, 0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 30;
  for (int i = 0; i < reps; i++) k_conv<<<blocks, 64 * WAVES>>>(dx, dy, dt, dab, T, 0);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double outs = (double)waves * T * NEW;
  printf("%d waves x %d tiles: %.4f ms per launch, %.1f G samples/s through three 90-tap filters (%.2f us per tile and wave)\n", waves, T, ms,
         outs / ms / 1e6, ms * 1e3 / T);
  printf("for scale: the direct stage C filters 67.1 M samples per launch in about 0.27 ms = 249 G samples/s\n");
  return 0;
}
