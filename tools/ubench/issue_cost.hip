// Issue cost of the instruction classes the fused kernels are made of, at THEIR occupancy (gfx950 / MI355X):
// W waves per SIMD each run a long stream of one instruction class (eight independent chains, inline asm, no memory), every
// wave times itself with s_memtime, and  cycles per wave-instruction and SIMD = cycles / (instructions per wave x W).
// A second table runs a class in one wave of a SIMD beside v_mfma_i32_16x16x64_i8 in the other: what an MFMA costs its neighbour.
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost [waves per SIMD: 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s, line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr int ITER = 2000, UNROLL = 8;

// one class = one asm template on registers %0 (float / int chain value), %1, %2 (operands)
#define CLASSES(X)                                                                   \
  X(0, "v_fma_f32 %0, %0, %1, %2")                                                   \
  X(1, "v_add_f32 %0, %0, %1")                                                       \
  X(2, "v_mul_f32 %0, %0, %1")                                                       \
  X(3, "v_fmac_f32 %0, %1, %2")                                                      \
  X(4, "v_cndmask_b32 %0, %0, %1, vcc")                                              \
  X(5, "v_bfi_b32 %0, %1, %0, %2")                                                   \
  X(6, "v_xor_b32 %0, %0, %1")                                                       \
  X(7, "v_perm_b32 %0, %0, %1, %2")                                                  \
  X(8, "v_cvt_f32_i32 %0, %0")                                                       \
  X(9, "v_cvt_f32_ubyte1 %0, %0")                                                    \
  X(10, "v_cvt_rpi_i32_f32 %0, %0")                                                  \
  X(11, "v_rcp_f32 %0, %0")                                                          \
  X(12, "v_max3_f32 %0, %0, %1, %2")                                                 \
  X(13, "v_min_f32 %0, %0, %1")                                                      \
  X(14, "v_lshl_add_u32 %0, %0, 8, %1")                                              \
  X(15, "v_mul_hi_u32 %0, %0, %1")                                                   \
  X(16, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")                 \
  X(17, "v_add_u32 %0, %0, %1")                                                      \
  X(18, "v_med3_f32 %0, %0, %1, %2")                                                 \
  X(19, "v_cmp_lt_f32 vcc, %0, %1")                                                  \
  X(20, "v_add_f32 %0, |%0|, |%1|")                                                  \
  X(21, "v_fma_f32 %0, -%0, %1, %2")
constexpr int N_SCALAR = 22;
static const char *NAMES[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_fmac_f32", "v_cndmask_b32", "v_bfi_b32", "v_xor_b32", "v_perm_b32",
                              "v_cvt_f32_i32", "v_cvt_f32_ubyte1", "v_cvt_rpi_i32_f32", "v_rcp_f32", "v_max3_f32", "v_min_f32", "v_lshl_add_u32",
                              "v_mul_hi_u32", "v_mov_b32_dpp row_shr:1", "v_add_u32", "v_med3_f32", "v_cmp_lt_f32 (vcc)", "v_add_f32 |a|,|b|",
                              "v_fma_f32 -a,b,c", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
                              "v_mfma_i32_16x16x64_i8", "v_mfma_i32_16x16x64_i8 (one accumulator)", "v_readlane_b32 + s_nop", "ds_bpermute_b32",
                              "v_cndmask_b32_e64 (sgpr pair)", "v_add_f32 + s_nop 0 (per pair)", "v_cmp_lt_f32 + v_cndmask vcc (per pair)",
                              "v_add_f32 + s_mul_i32 (per pair)", "MFMA + 1 v_fma (per group)", "MFMA + 2 v_fma (per group)", "MFMA + 3 v_fma (per group)",
                              "MFMA + 4 v_fma (per group)", "s_nop 0", "ds_read_b128 x1 + wait (per read)",
                              "v_fma_f32, ONE dependent chain", "v_fma_f32, two chains", "v_pk_fma_f32, ONE dependent chain", "v_cvt_f32_i32, ONE dependent chain",
                              "v_fma_f32 -> v_cndmask_e64 -> v_bfi (one chain, per instr)", "MFMA -> v_cvt_f32_i32 of its result (per pair)"};
constexpr int C_PKFMA = 22, C_PKADD = 23, C_PKMUL = 24, C_PKFMA_SEL = 25, C_MFMA = 26, C_MFMA1 = 27, C_READLANE = 28, C_BPERM = 29, C_CND64 = 30, C_ADDNOP = 31, C_CMPCND = 32, C_ADDSMUL = 33, C_M1 = 34, C_M2 = 35, C_M3 = 36, C_M4 = 37, C_SNOP = 38, C_LDS128 = 39, C_DEP1 = 40, C_DEP2 = 41, C_PKDEP1 = 42, C_CVTDEP1 = 43, C_MIXDEP = 44, C_MFMACVT = 45, N_CLASSES = 46;

template <int C>
__device__ __forceinline__ void body(float (&x)[UNROLL], f2 (&p)[UNROLL], i4 (&acc)[UNROLL], const i4 &ma, const i4 &mb, float a, float b) {
#pragma unroll
  for (int u = 0; u < UNROLL; u++) {
#define X(ID, TXT) if constexpr (C == ID) asm volatile(TXT : "+v"(x[u]) : "v"(a), "v"(b) : "vcc");
    CLASSES(X)
#undef X
    if constexpr (C == C_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(f2{a, a}), "v"(f2{b, b}));
    if constexpr (C == C_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u]) : "v"(f2{a, a}));
    if constexpr (C == C_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u]) : "v"(f2{a, a}));
    if constexpr (C == C_PKFMA_SEL) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[u]) : "v"(f2{a, a}), "v"(f2{b, b}));
    if constexpr (C == C_MFMA) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(ma), "v"(mb));
    if constexpr (C == C_MFMA1) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(ma), "v"(mb));
    if constexpr (C == C_READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 5\n\ts_nop 0" : "=s"(s) : "v"(x[u])); asm volatile("" :: "s"(s)); }
    if constexpr (C == C_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(x[u]) : "v"(__builtin_bit_cast(float, (int)(threadIdx.x * 4 + 4) & 255)));
    if constexpr (C == C_CND64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[u]) : "v"(a), "s"(0x5555aaaa5555aaaaull));
    if constexpr (C == C_ADDNOP) asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(x[u]) : "v"(a));
    if constexpr (C == C_CMPCND) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[u]) : "v"(a), "v"(b) : "vcc");
    if constexpr (C == C_ADDSMUL) { int sd; asm volatile("v_add_f32 %0, %0, %2\n\ts_mul_i32 %1, %3, %3" : "+v"(x[u]), "=s"(sd) : "v"(a), "s"(u + 3)); asm volatile("" :: "s"(sd)); }
    if constexpr (C >= C_M1 && C <= C_M4) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(ma), "v"(mb));
#pragma unroll
      for (int n = 0; n < C - C_M1 + 1; n++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(u + n) % UNROLL]) : "v"(a), "v"(b));
    }
    if constexpr (C == C_SNOP) asm volatile("s_nop 0");
    if constexpr (C == C_DEP1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
    if constexpr (C == C_DEP2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u & 1]) : "v"(a), "v"(b));
    if constexpr (C == C_PKDEP1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(f2{a, a}), "v"(f2{b, b}));
    if constexpr (C == C_CVTDEP1) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[0]));
    if constexpr (C == C_MIXDEP) {
      if (u % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
      if (u % 3 == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "s"(0x5555aaaa5555aaaaull));
      if (u % 3 == 2) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[0]) : "v"(a), "v"(b));
    }
    if constexpr (C == C_MFMACVT) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(ma), "v"(mb));
      asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(x[u]) : "v"(acc[u].x));
    }
    if constexpr (C == C_LDS128) { i4 t; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((int)(threadIdx.x * 16) & 0x3ff0) : "memory"); acc[u] = t; }
  }
  if constexpr (C == C_BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ONE workgroup of 4 W waves per CU: waves w and w + 4 share a SIMD (the dispatcher deals a workgroup's waves round the four SIMDs),
// wave >> 2 decides the role when two classes share a SIMD; every wave reports its own cycles for ITER x UNROLL instructions
template <int CA, int CB>
__global__ __launch_bounds__(1024) void run(long long *cycles, float *sink, float a, float b, int waves_per_simd) {
  __shared__ int lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  const bool second = ((threadIdx.x >> 8) & 1) != 0;
  float x[UNROLL]; f2 p[UNROLL]; i4 acc[UNROLL];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int u = 0; u < UNROLL; u++) { x[u] = 0.5f + 0.001f * (lane + u); p[u] = f2{x[u], x[u] + 1.f}; acc[u] = i4{0, 0, 0, 0}; }
  const i4 ma = {lane, lane * 3, lane * 5, lane * 7}, mb = {lane * 11, lane * 13, lane * 17, lane * 19};
  __syncthreads();
  const long long t0 = clock64();
  if (!second) { for (int it = 0; it < ITER; it++) body<CA>(x, p, acc, ma, mb, a, b); }
  else { for (int it = 0; it < ITER; it++) body<CB>(x, p, acc, ma, mb, a, b); }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < UNROLL; u++) s += x[u] + p[u].x + p[u].y + (float)(acc[u].x + acc[u].y + acc[u].z + acc[u].w);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)lds[threadIdx.x];
  if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(long long *, float *, float, float, int);
template <int C> kern_t same() { return run<C, C>; }
template <int C> kern_t beside_mfma() { return run<C_MFMA, C>; }

template <int... I> void fill(kern_t (&s)[N_CLASSES], kern_t (&m)[N_CLASSES], std::integer_sequence<int, I...>) {
  ((s[I] = same<I>()), ...);
  ((m[I] = beside_mfma<I>()), ...);
}

int main(int argc, char **argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 2;
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus, wpb = 4 * W;
  long long *cyc; float *sink;
  CHECK(hipMalloc(&cyc, blocks * wpb * sizeof(long long))); CHECK(hipMalloc(&sink, blocks * 64 * wpb * sizeof(float)));
  kern_t s[N_CLASSES], m[N_CLASSES];
  fill(s, m, std::make_integer_sequence<int, N_CLASSES>{});
  std::vector<long long> h(blocks * wpb);
  printf("device %s, %d CUs, %d waves per SIMD, %d instructions per wave\n", prop.name, cus, W, ITER * UNROLL);
  printf("%-42s %14s | beside v_mfma_i32_16x16x64_i8 in the SIMD's other wave: %10s %10s\n", "class (all waves of the SIMD)", "cycles/instr", "this class", "the MFMA");
  for (int c = 0; c < N_CLASSES; c++) {
    double alone = 0, me = 0, mf = 0;
    for (int rep = 0; rep < 3; rep++) {                      // the last of three launches counts (clocks settled)
      hipLaunchKernelGGL(s[c], dim3(blocks), dim3(64 * wpb), 0, 0, cyc, sink, 1.0001f, 0.25f, W);
      CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (long long v : h) alone += (double)v;
    alone = alone / h.size() / (ITER * UNROLL) / W;          // per wave-instruction and SIMD
    if (W == 2) {
      for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(m[c], dim3(blocks), dim3(64 * wpb), 0, 0, cyc, sink, 1.0001f, 0.25f, W);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      int n0 = 0, n1 = 0;
      for (int bl = 0; bl < blocks; bl++)
        for (int w = 0; w < wpb; w++) { if ((w >> 2) & 1) { me += (double)h[bl * wpb + w]; n1++; } else { mf += (double)h[bl * wpb + w]; n0++; } }
      me = me / n1 / (ITER * UNROLL); mf = mf / n0 / (ITER * UNROLL);   // cycles per own instruction while sharing the SIMD
      printf("%-42s %14.2f | %45.2f %10.2f\n", NAMES[c], alone, me, mf);
    } else {
      printf("%-42s %14.2f\n", NAMES[c], alone);
    }
  }
  if (W == 2) {
    printf("wave A: one MFMA then n v_fma_f32, repeated; wave B of the same SIMD: a stream of v_fma_f32 / v_pk_fma_f32\n");
    kern_t mix[4][2] = {{run<C_M1, 0>, run<C_M1, C_PKFMA>}, {run<C_M2, 0>, run<C_M2, C_PKFMA>}, {run<C_M3, 0>, run<C_M3, C_PKFMA>}, {run<C_M4, 0>, run<C_M4, C_PKFMA>}};
    for (int n = 0; n < 4; n++)
      for (int k = 0; k < 2; k++) {
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(mix[n][k], dim3(blocks), dim3(64 * wpb), 0, 0, cyc, sink, 1.0001f, 0.25f, W); CHECK(hipDeviceSynchronize()); }
        CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double ta = 0, tb = 0; int na = 0, nb = 0;
        for (int bl = 0; bl < blocks; bl++)
          for (int w = 0; w < wpb; w++) { if ((w >> 2) & 1) { tb += (double)h[bl * wpb + w]; nb++; } else { ta += (double)h[bl * wpb + w]; na++; } }
        printf("  A = MFMA + %d v_fma: %.2f cycles per group | B = %s: %.2f cycles per instruction\n", n + 1, ta / na / (ITER * UNROLL), k ? "v_pk_fma_f32" : "v_fma_f32", tb / nb / (ITER * UNROLL));
      }
  }
  return 0;
}
