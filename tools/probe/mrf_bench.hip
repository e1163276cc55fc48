// Micro-benchmark of mrf_small_kernel alone (tools/gpu scripts build it with hipcc on the GPU box or here):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRF_C=16 -DMRF_T=256 -DMRF_NW=4 tools/probe/mrf_bench.hip -o /tmp/mrf_bench
// Times one launch over BASELINE config 4's ragged batch (8 rows, 2200 frames) with random weights; prints us per launch and
// the rate in useful TFLOP/s.  Numerics are checked by the test-suite, not here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "../../larynx_amd/csrc/mrf_small.h"
using namespace mi355tts;
#ifndef MRF_C
#define MRF_C 16
#endif
#ifndef MRF_T
#define MRF_T 256
#endif
#ifndef MRF_NW
#define MRF_NW 4
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int C = MRF_C, mul = C == 16 ? 128 : 256;
  int frames[8] = {108, 106, 120, 174, 254, 172, 502, 764};
  int B = 8;
  if (argc > 1) { B = 1; frames[0] = atoi(argv[1]); }
  int Lmax = 0; long long Lsum = 0;
  for (int b = 0; b < B; ++b) { Lmax = frames[b] * mul > Lmax ? frames[b] * mul : Lmax; Lsum += (long long)frames[b] * mul; }
  const int ld = (Lmax + 3) & ~3;
  const int Ks[3] = {3, 7, 11}, dil[3] = {1, 3, 5};
  std::vector<float> w, bias(3 * 3 * 2 * 16, 0.01f);
  int tab[MRF_TAB_INTS] = {};
  srand(1);
  for (int j = 0; j < 3; ++j)
    for (int d = 0; d < 3; ++d) {
      for (int cv = 0; cv < 2; ++cv) {
        tab[(j * 3 + d) * 2 + cv] = (int)w.size();
#ifdef MRF_K8  // the 4x4x1 packing of mrf8_kernel: [tap][64]
        for (int i = 0; i < Ks[j] * 64; ++i) w.push_back((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
#else
        for (int i = 0; i < Ks[j] * (C / 4) * 64; ++i) w.push_back(((i & 15) < C) ? (rand() / (float)RAND_MAX - 0.5f) * 0.1f : 0.f);
#endif
      }
      tab[MRF_TAB_DIL + j * 3 + d] = dil[d];
    }
  for (int i = 0; i < 1024; ++i) w.push_back(0.f);
  std::vector<float> x((size_t)B * C * ld);
  for (auto& v : x) v = rand() / (float)RAND_MAX - 0.5f;
  float *dx, *dy, *dy2, *dw, *db; int *dt, *dl;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dy, x.size() * 4)); CK(hipMalloc(&dy2, x.size() * 4));
  CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&db, bias.size() * 4)); CK(hipMalloc(&dt, sizeof(tab))); CK(hipMalloc(&dl, 32));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tab, sizeof(tab), hipMemcpyHostToDevice));
  CK(hipMemcpy(dl, frames, 32, hipMemcpyHostToDevice));
  MrfArgs a{};
  a.x = dx; a.y = dy; a.y2 = dy2; a.bs = (long long)C * ld; a.ld = ld; a.len = B > 1 ? dl : nullptr; a.len_mul = mul; a.len_const = frames[0] * mul;
  a.w = dw; a.bias = db; a.tab = dt; a.nsteps = 3; a.slope = 0.1f;
  dim3 grid(2 * ((Lmax + MRF_T - 1) / MRF_T), 1, B);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef MRF_K8
#define MRF_LAUNCH hipLaunchKernelGGL(HIP_KERNEL_NAME(mrf8_kernel<MRF_T, 3, 7, 11>), grid, dim3(128), 0, 0, a)
#else
#define MRF_LAUNCH hipLaunchKernelGGL(HIP_KERNEL_NAME(mrf_small_kernel<MRF_C, MRF_T, MRF_NW, 3, 7, 11>), grid, dim3(64 * MRF_NW), 0, 0, a)
#endif
  for (int i = 0; i < 3; ++i) MRF_LAUNCH;
  CK(hipDeviceSynchronize());
  const int N = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < N; ++i) MRF_LAUNCH;
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / N, flop = 2.0 * 2.0 * C * C * 21 * 3 * (double)Lsum;
  printf("mrf_small<%d,%d,%d> %s: B=%d columns=%lld  %.1f us/launch  %.1f useful TFLOP/s (%.3f of 157.3)\n", MRF_C, MRF_T, MRF_NW,
#ifdef MRF_TAG
         MRF_TAG,
#else
         "",
#endif
         B, Lsum, us, flop / us / 1e6, flop / us / 1e6 / 157.3);
  return 0;
}
