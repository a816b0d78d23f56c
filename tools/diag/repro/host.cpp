// Host side of the neighbour experiment: for every victim code object on the command line, one clean launch (the
// reference: all waves must agree with wave 0's tiles), then launches with a neighbour kernel alive on a second stream;
// a device-side compare counts wrong values by lane quarter and by output index and keeps the first records.
//   ./host <neighbour.hsaco> <kinds e.g. 0,3> <launches> <grid blocks> <tiles> <prio> <victim.hsaco> ...
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
struct vparams { float g[16]; float c_i, c_q; };

int main(int argc, char **argv) {
  if (argc < 8) { printf("usage\n"); return 1; }
  const char *nb_path = argv[1];
  std::vector<int> kinds;
  for (char *t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) kinds.push_back(atoi(t));
  const int launches = atoi(argv[3]), grid = atoi(argv[4]), tiles = atoi(argv[5]), prio = atoi(argv[6]);
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipModule_t nb;
  CHECK(hipModuleLoad(&nb, nb_path));
  int *stop; CHECK(hipHostMalloc((void **)&stop, 64, hipHostMallocMapped));
  int *dstop; CHECK(hipHostGetDevicePointer((void **)&dstop, stop, 0));
  float *sink; CHECK(hipMalloc(&sink, 256 * 256 * sizeof(float)));
  /* IQ bytes: an LCG, the same for every wave */
  std::vector<unsigned char> hiq((size_t)tiles * 8192);
  unsigned s = 12345u;
  /* PATTERN: 0 LCG noise on I and Q, 1 noise on I with Q = 0 (byte 0x80), 2 noise on Q with I = 0, 3 I = Q = +64 constant */
  const int pattern = getenv("PATTERN") ? atoi(getenv("PATTERN")) : 0, pow2 = getenv("TAPS") ? atoi(getenv("TAPS")) : 0;
  for (size_t i = 0; i < hiq.size(); i++) {
    s = s * 1664525u + 1013904223u;
    unsigned char b = (unsigned char)(s >> 24);
    if (pattern == 1 && (i & 1)) b = 0x80;
    if (pattern == 2 && !(i & 1)) b = 0x80;
    if (pattern == 3) b = 0xC0;
    hiq[i] = b;
  }
  unsigned char *iq; CHECK(hipMalloc(&iq, hiq.size()));
  CHECK(hipMemcpy(iq, hiq.data(), hiq.size(), hipMemcpyHostToDevice));
  vparams P;
  for (int i = 0; i < 16; i++) {          /* the reference's 32-tap low-pass (half), / 128 */
    const float j = (float)i - 15.5f;
    P.g[i] = sinf(0.125f * 3.14159265f * j) / (3.14159265f * j) * (0.54f - 0.46f * cosf(3.14159265f * (float)i / 15.5f)) / 128.f;
  }
  if (pow2) for (int i = 0; i < 16; i++) P.g[i] = ldexpf(1.f, -(8 + i));      /* TAPS=1: every tap its own power of two (a wrong term names itself) */
  P.c_i = pow2 ? 0.f : 0.001f; P.c_q = pow2 ? 0.f : -0.002f;
  printf("pattern %d taps %s\n", pattern, pow2 ? "2^-(8+k)" : "low-pass");
  const size_t units = (size_t)grid * 4, n_per_unit = (size_t)tiles * 512, n_total = units * n_per_unit;
  float2 *out, *ref; unsigned *counts, *rec;
  CHECK(hipMalloc(&out, n_total * sizeof(float2)));
  CHECK(hipMalloc(&ref, n_per_unit * sizeof(float2)));
  CHECK(hipMalloc(&counts, 64 * 4)); CHECK(hipMalloc(&rec, 256 * 8 * 4));
  for (int vi = 7; vi < argc; vi++) {
    hipModule_t vm;
    if (hipModuleLoad(&vm, argv[vi]) != hipSuccess) { printf("%s: cannot load\n", argv[vi]); continue; }
    hipFunction_t victim, compare;
    CHECK(hipModuleGetFunction(&victim, vm, "victim"));
    CHECK(hipModuleGetFunction(&compare, vm, "compare"));
    int tl = tiles, pr = prio;
    void *vargs[] = {&P, &iq, &out, &tl, &pr};
    auto run_victim = [&]() { CHECK(hipModuleLaunchKernel(victim, grid, 1, 1, 256, 1, 1, 0, s1, vargs, nullptr)); };
    unsigned hc[64]; std::vector<unsigned> hrec(256 * 8);
    auto run_compare = [&]() {
      CHECK(hipMemsetAsync(counts, 0, 64 * 4, s1));
      unsigned long long npu = n_per_unit, nt = n_total;
      void *cargs[] = {&out, &ref, &npu, &nt, &counts, &rec};
      CHECK(hipModuleLaunchKernel(compare, 2048, 1, 1, 256, 1, 1, 0, s1, cargs, nullptr));
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipMemcpy(hc, counts, 64 * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hrec.data(), rec, 256 * 8 * 4, hipMemcpyDeviceToHost));
    };
    /* clean launch: wave 0's tiles are the reference, every other wave must agree */
    CHECK(hipMemsetAsync(out, 0xff, n_total * sizeof(float2), s1));
    run_victim(); CHECK(hipStreamSynchronize(s1));
    CHECK(hipMemcpyAsync(ref, out, n_per_unit * sizeof(float2), hipMemcpyDeviceToDevice, s1));
    CHECK(hipStreamSynchronize(s1));
    run_compare();
    printf("== %s | grid %d x 256, %d tiles, prio %d | clean launch: %u wrong values\n", argv[vi], grid, tiles, prio, hc[12]);
    for (int kind : kinds) {
      char name[16]; snprintf(name, sizeof name, "burst%d", kind);
      hipFunction_t bf; CHECK(hipModuleGetFunction(&bf, nb, name));
      unsigned tot[14] = {0}; int alive = 0; float ms_sum = 0.f;
      std::vector<unsigned> first_rec;
      *stop = 0;
      int nprio = 0, loops = 500000;
      void *bargs[] = {&dstop, &sink, &nprio, &loops};
      CHECK(hipModuleLaunchKernel(bf, 256, 1, 1, 256, 1, 1, 0, s2, bargs, nullptr));
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      for (int l = 0; l < launches; l++) {
        CHECK(hipEventRecord(e0, s1));
        run_victim();
        CHECK(hipEventRecord(e1, s1));
        CHECK(hipStreamSynchronize(s1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms;
        alive += hipStreamQuery(s2) == hipErrorNotReady;
        run_compare();                       /* (runs beside the neighbour too: integer compares; cross-checked by the clean launch above) */
        for (int i = 0; i < 14; i++) tot[i] += hc[i];
        if (first_rec.empty() && hc[12]) first_rec.assign(hrec.begin(), hrec.begin() + 8 * (hc[13] < 24 ? hc[13] : 24));
      }
      *stop = 1;
      CHECK(hipStreamSynchronize(s2));
      printf("  neighbour kind %d: %u wrong values in %d launches (%.3f ms each; neighbour alive after %d) | lanes 0-15 %u 16-31 %u 32-47 %u 48-63 %u | by r %u %u %u %u %u %u %u %u\n",
             kind, tot[12], launches, ms_sum / launches, alive, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], tot[6], tot[7], tot[8], tot[9], tot[10], tot[11]);
      for (size_t k = 0; k + 8 <= first_rec.size(); k += 8) {
        float g0, g1, w0, w1;
        memcpy(&g0, &first_rec[k + 4], 4); memcpy(&g1, &first_rec[k + 5], 4); memcpy(&w0, &first_rec[k + 6], 4); memcpy(&w1, &first_rec[k + 7], 4);
        printf("    unit %u tile %u lane %u r %u got (%.9g, %.9g) want (%.9g, %.9g) diff (%.6g, %.6g) sum (%.6g, %.6g)\n", first_rec[k], first_rec[k + 1], first_rec[k + 2], first_rec[k + 3],
               g0, g1, w0, w1, g0 - w0, g1 - w1, g0 + w0, g1 + w1);
      }
      fflush(stdout);
    }
    CHECK(hipModuleUnload(vm));
  }
  return 0;
}
