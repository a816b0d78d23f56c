// Design study for the next round: stage C (three 90-tap FIRs on one input) as an overlap-save FFT inside ONE wavefront.
// A block is N = 1024 real samples (89 of history + 935 new) handled as a 512-point complex FFT of (even, odd) pairs:
//   Z = FFT512(z),  Z'_f[k] = A_f[k] Z[k] + B_f[k] conj(Z[512 - k])   (split, filter spectrum and merge in two tables),
//   z'_f = IFFT512(Z'_f)  ->  y_f[2m] = Re z'_f[m],  y_f[2m+1] = Im z'_f[m],  valid for n >= 89.
// FFT512 = three radix-8 passes on 8 complex values per lane with two exchanges through LDS (see fft512).
// The program checks the three outputs of one block against a double-precision convolution and times the loop.
//   hipcc --offload-arch=gfx950 -O3 -o wave_fft wave_fft.hip && ./wave_fft
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 1024, M = 512, TAPS = 90, NEW = N - (TAPS - 1);
constexpr int WAVES = 4;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// complex helpers on (re, im) register pairs; swaps and signs are VOP3P operand modifiers (hipcc would spend moves on them)
__device__ __forceinline__ f2 cmul(f2 a, f2 b) {      // a * b: a (b.x, b.x), then + (a.y, a.x) (-b.y, b.y)
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
__device__ __forceinline__ f2 cmulc(f2 a, f2 b) {     // a * conj(b)
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
// a + i^ROT b  (ROT 0..3): one v_pk_add_f32
template <int ROT>
__device__ __forceinline__ f2 add_rot(f2 a, f2 b) {
  f2 r;
  if constexpr (ROT == 0) r = a + b;
  else if constexpr (ROT == 2) r = a - b;
  else if constexpr (ROT == 1) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));   // (-b.y, b.x)
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));                       // (b.y, -b.x)
  return r;
}
// q * W8^1, W8^3 (forward) and their conjugates, times sqrt 2: one v_pk_add_f32 of q with its own swapped halves
template <int K, bool INV>
__device__ __forceinline__ f2 w8_unscaled(f2 q) {
  f2 r;
  if constexpr (K == 1 && !INV) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(q));            // (x + y, y - x)
  else if constexpr (K == 1 && INV) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(q));        // (x - y, y + x)
  else if constexpr (K == 3 && !INV) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(r) : "v"(q));   // (-x + y, -y - x)
  else asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(r) : "v"(q));                        // (-x - y, -y + x)
  return r;
}

// d[k] = sum_j p[j] (-+i)^(jk), the odd input p[2] optionally still to be turned by -+i (ROT2: folded into the additions)
template <bool INV, bool ROT2>
__device__ __forceinline__ void dft4(const f2 (&p)[4], f2 &d0, f2 &d1, f2 &d2, f2 &d3) {
  constexpr int R = INV ? 1 : 3;                      // i^R = +-i
  f2 e0, o0;
  if constexpr (ROT2) { e0 = add_rot<R>(p[0], p[2]); o0 = add_rot<(R + 2) & 3>(p[0], p[2]); }
  else { e0 = p[0] + p[2]; o0 = p[0] - p[2]; }
  const f2 e1 = p[1] + p[3], t = p[1] - p[3];
  d0 = e0 + e1; d2 = e0 - e1;
  d1 = add_rot<R>(o0, t); d3 = add_rot<(R + 2) & 3>(o0, t);
}
// a[k] <- sum_j a[j] W8^(jk), W8 = exp(-+ 2 pi i / 8)
template <bool INV>
__device__ __forceinline__ void dft8(f2 (&a)[8]) {
  constexpr float R = 0.70710678118654752f;
  f2 p[4], q[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { p[j] = a[j] + a[j + 4]; q[j] = a[j] - a[j + 4]; }
  q[1] = w8_unscaled<1, INV>(q[1]) * f2{R, R};
  q[3] = w8_unscaled<3, INV>(q[3]) * f2{R, R};
  dft4<INV, false>(p, a[0], a[2], a[4], a[6]);
  dft4<INV, true>(q, a[1], a[3], a[5], a[7]);
}

// r[j] = z[64 j + lane]  ->  r[g] = Z[lane + 64 g].  Index m = 64 j + l, l = 8 c + d; bin k = k1 + 8 (e + 8 g):
//   pass 1: radix 8 over j in registers, twiddle W512^(l k1); exchange to lane (k1, d) holding c = 0..7;
//   pass 2: radix 8 over c, twiddle W64^(d e); exchange to lane k1 + 8 e holding d = 0..7;
//   pass 3: radix 8 over d.
// The two exchanges go through a padded LDS buffer (rows of 8 complex values 9 apart, the second one 76 per k1) so that
// the 8-byte writes of 16 neighbouring lanes and the 8-byte reads of 32 lanes spread over the banks.
constexpr int EXN = 8 * 76 + 8;
template <bool INV>
__device__ __forceinline__ void fft512(f2 (&r)[8], int lane, const f2 (&tw1)[8], const f2 (&tw2)[8], f2 *ex) {
  dft8<INV>(r);
#pragma unroll
  for (int k = 1; k < 8; k++) r[k] = INV ? cmulc(r[k], tw1[k]) : cmul(r[k], tw1[k]);
  {
    const int c = lane >> 3, d = lane & 7;
    f2 *wr = ex + d * 9 + c;
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) wr[k1 * 72] = r[k1];
    wave_sync();
    const f2 *rd = ex + lane * 9;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = rd[i];
    wave_sync();
  }
  dft8<INV>(r);
#pragma unroll
  for (int e = 1; e < 8; e++) r[e] = INV ? cmulc(r[e], tw2[e]) : cmul(r[e], tw2[e]);
  {
    const int k1 = lane >> 3, d = lane & 7;
    f2 *wr = ex + k1 * 76 + d;
#pragma unroll
    for (int e = 0; e < 8; e++) wr[e * 9] = r[e];
    wave_sync();
    const f2 *rd = ex + (lane & 7) * 76 + (lane >> 3) * 9;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = rd[i];
    wave_sync();
  }
  dft8<INV>(r);
}

__global__ __launch_bounds__(64 * WAVES, 3) void k_conv(const float *xg, float *yg, const f2 *tw1g, const f2 *tw2g, const f4 *abg,
                                                       int blocks_per_wave, int store_all) {
  __shared__ f2 exs[WAVES][EXN];
  __shared__ float xs[WAVES][N];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * WAVES + wv;
  f2 *ex = exs[wv];
  float *xl = xs[wv];
  f2 tw1[8], tw2[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { tw1[k] = tw1g[lane * 8 + k]; tw2[k] = tw2g[lane * 8 + k]; }
  f2 sum = {0.f, 0.f};
  for (int t = 0; t < blocks_per_wave; t++) {
    const float *x = xg + ((size_t)wave * blocks_per_wave + t) * N;
    float *y = yg + ((size_t)wave * blocks_per_wave + t) * 3 * N;
    // the block into LDS (in the fused kernel stage B has left it there)
#pragma unroll
    for (int i = 0; i < 4; i++) reinterpret_cast<f4 *>(xl)[lane + 64 * i] = reinterpret_cast<const f4 *>(x)[lane + 64 * i];
    wave_sync();
    f2 z[8];
#pragma unroll
    for (int j = 0; j < 8; j++) z[j] = reinterpret_cast<const f2 *>(xl)[64 * j + lane];
    fft512<false>(z, lane, tw1, tw2, ex);
    // conj(Z[512 - k]) for k = lane + 64 g: lane 64 - lane, register 7 - g (lane 0: its own register (8 - g) mod 8)
    f2 zc[8];
    const int from = ((64 - lane) & 63) << 2;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      /* the two halves go through opaque scalars: hipcc 7.2 pairs two 32-bit cross-lane operations on the halves of a
       * 2-vector into one and drops the second (seen with DPP moves and with ds_bpermute) */
      float ox = z[7 - g].x, oy = z[7 - g].y;
      asm volatile("" : "+v"(ox));
      asm volatile("" : "+v"(oy));
      float mx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, ox)));
      float my = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(from, __builtin_bit_cast(int, oy)));
      const f2 own = z[(8 - g) & 7];
      if (lane == 0) { mx = own.x; my = own.y; }
      zc[g] = f2{mx, -my};
    }
#pragma unroll 1
    for (int f = 0; f < 3; f++) {
      f2 w[8];
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const f4 ab = abg[(f * 8 + g) * 64 + lane];
        w[g] = cmul(f2{ab.x, ab.y}, z[g]) + cmul(f2{ab.z, ab.w}, zc[g]);
      }
      fft512<true>(w, lane, tw1, tw2, ex);
      if (store_all) {
#pragma unroll
        for (int j = 0; j < 8; j++) reinterpret_cast<f2 *>(y + f * N)[64 * j + lane] = w[j];
      } else {                                     /* timing: the fused kernel keeps the outputs on chip */
#pragma unroll
        for (int j = 0; j < 8; j++) sum += w[j];
      }
    }
  }
  if (!store_all) reinterpret_cast<f2 *>(yg)[(size_t)wave * 64 + lane] = sum;
}

int main() {
  // taps: three random symmetric 90-tap filters (like fm / fp / fs); tables in double
  std::vector<double> F(3 * TAPS);
  srand(7);
  for (int f = 0; f < 3; f++)
    for (int j = 0; j < 45; j++) { const double v = (rand() / (double)RAND_MAX - 0.5) * 0.1; F[f * TAPS + j] = v; F[f * TAPS + 89 - j] = v; }
  typedef std::complex<double> cd;
  const double PI = 3.14159265358979323846;
  std::vector<f4> ab(3 * 8 * 64);
  for (int f = 0; f < 3; f++) {
    std::vector<cd> H(N);
    for (int k = 0; k < N; k++) { cd s = 0; for (int j = 0; j < TAPS; j++) s += F[f * TAPS + j] * std::polar(1.0, -2 * PI * k * j / N); H[k] = s; }
    for (int k = 0; k < M; k++) {
      const cd W = std::polar(1.0, -2 * PI * k / N), I(0, 1);
      const cd P = 0.5 * (H[k] + H[k + M]) + 0.5 * I * std::conj(W) * (H[k] - H[k + M]);
      const cd Q = 0.5 * W * (H[k] - H[k + M]) + 0.5 * I * (H[k] + H[k + M]);
      const cd A = (P - I * Q) / 2.0 / (double)M, B = (P + I * Q) / 2.0 / (double)M;
      ab[(f * 8 + (k >> 6)) * 64 + (k & 63)] = f4{(float)A.real(), (float)A.imag(), (float)B.real(), (float)B.imag()};
    }
  }
  std::vector<f2> tw1(64 * 8), tw2(64 * 8);
  for (int l = 0; l < 64; l++)
    for (int k = 0; k < 8; k++) {
      const cd a = std::polar(1.0, -2 * PI * l * k / 512.0), b = std::polar(1.0, -2 * PI * (l & 7) * k / 64.0);
      tw1[l * 8 + k] = f2{(float)a.real(), (float)a.imag()};
      tw2[l * 8 + k] = f2{(float)b.real(), (float)b.imag()};
    }
  int cus = 256; hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount;
  const int blocks = cus * 3, waves = blocks * WAVES, T = 24;
  const size_t nx = (size_t)waves * T * N;
  std::vector<float> hx(nx);
  uint32_t s = 12345; for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
  float *dx, *dy; f2 *d1, *d2; f4 *dab;
  CHECK(hipMalloc(&dx, nx * 4)); CHECK(hipMalloc(&dy, nx * 3 * 4));
  CHECK(hipMalloc(&d1, tw1.size() * 8)); CHECK(hipMalloc(&d2, tw2.size() * 8)); CHECK(hipMalloc(&dab, ab.size() * 16));
  CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d1, tw1.data(), tw1.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d2, tw2.data(), tw2.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dab, ab.data(), ab.size() * 16, hipMemcpyHostToDevice));
  k_conv<<<blocks, 64 * WAVES>>>(dx, dy, d1, d2, dab, T, 1);
  CHECK(hipDeviceSynchronize());
  // check block 5 of wave 7
  {
    const size_t blk = (size_t)7 * T + 5;
    std::vector<float> hy(3 * N);
    CHECK(hipMemcpy(hy.data(), dy + blk * 3 * N, 3 * N * 4, hipMemcpyDeviceToHost));
    double worst = 0, rms = 0;
    for (int f = 0; f < 3; f++)
      for (int n = TAPS - 1; n < N; n++) {
        double r = 0;
        for (int j = 0; j < TAPS; j++) r += F[f * TAPS + j] * (double)hx[blk * N + n - j];
        worst = fmax(worst, fabs(r - (double)hy[f * N + n]));
        rms += r * r;
      }
    printf("wave FFT convolution vs double: max |err| %.3g (rms of the outputs %.3g)\n", worst, sqrt(rms / (3.0 * NEW)));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 40; i++) k_conv<<<blocks, 64 * WAVES>>>(dx, dy, d1, d2, dab, T, 0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 30;
  for (int i = 0; i < reps; i++) k_conv<<<blocks, 64 * WAVES>>>(dx, dy, d1, d2, dab, T, 0);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double outs = (double)waves * T * NEW;
  printf("%d waves x %d blocks: %.4f ms per launch, %.1f G useful samples/s through three 90-tap filters (%.2f us per block and wave)\n", waves, T, ms,
         outs / ms / 1e6, ms * 1e3 / T);
  printf("for scale: the shipped stage C filters 67.1 M samples per launch in about 0.27 ms = 249 G samples/s\n");
  return 0;
}
