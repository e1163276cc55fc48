// Micro-benchmark of conv_group_kernel (the grouped ResBlock-conv launch of the wide HiFi-GAN stages) alone:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCG_C=128 -DCG_L=39936 -DCG_CI=16 -DCG_MB=1 -DCG_NB=2 -DCG_WN=1 -DCG_KS=1 -DCG_WM=4]
//         tools/probe/conv_group_bench.hip -o /tmp/cgb
// Three members (k = 11, 7, 3; dilation 1) of C x C convs over one row of L columns, random weights and data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../larynx_amd/csrc/conv_mfma.h"
#include "../../larynx_amd/csrc/weights_pack.h"
using namespace mi355tts;
#ifndef CG_C
#define CG_C 128
#endif
#ifndef CG_L
#define CG_L 39936
#endif
#ifndef CG_CI
#define CG_CI 16
#endif
#ifndef CG_MB
#define CG_MB 1
#endif
#ifndef CG_NB
#define CG_NB 2
#endif
#ifndef CG_WN
#define CG_WN 1
#endif
#ifndef CG_KS
#define CG_KS 1
#endif
#ifndef CG_WM
#define CG_WM 4
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int C = CG_C, L = CG_L;
  const int Ks[3] = {11, 7, 3};
  srand(2);
  std::vector<float> x((size_t)C * L);
  for (auto& v : x) v = rand() / (float)RAND_MAX - 0.5f;
  float *dx, *dy[3];
  CK(hipMalloc(&dx, x.size() * 4));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  ConvGroupArgs g;
  memset(&g, 0, sizeof(g));
  constexpr int T_T = CG_WN * CG_NB * 32;
  const int ytiles = C / (32 * CG_MB * CG_WM);
  int off = 0;
  double flop = 0;
  for (int m = 0; m < 3; ++m) {
    const int K = Ks[m];
    std::vector<float> w((size_t)C * C * K), b(C);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (auto& v : b) v = 0.01f;
    PackedConv p = pack_conv(C, CG_MB * CG_WM, C, K, [&](int v) { return v; }, [&](int co, int ci, int k) { return w[((size_t)co * C + ci) * K + k]; },
                             [&](int co) { return b[co]; }, true, 8);
    float *dw, *db;
    CK(hipMalloc(&dw, p.w.size() * 4)); CK(hipMalloc(&db, p.bias.size() * 4)); CK(hipMalloc(&dy[m], x.size() * 4));
    CK(hipMemcpy(dw, p.w.data(), p.w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, p.bias.data(), p.bias.size() * 4, hipMemcpyHostToDevice));
    ConvArgs& a = g.c[m];
    a.x = dx; a.x_bs = (long long)C * L; a.x_ld = L; a.in_const = L; a.in_mul = 1;
    a.w = dw; a.bias = db; a.noct = p.noct; a.Cin = C; a.rows = C; a.dil = 1; a.pad = (K - 1) / 2; a.in_slope = 0.1f;
    a.y = dy[m]; a.y_bs = (long long)C * L; a.y_ld = L; a.split = 1 << 30; a.alpha = 1.f; a.out_const = L; a.out_mul = 1; a.res = dx;
    g.gx[m] = (L + T_T - 1) / T_T; g.gy[m] = ytiles; g.off[m] = off;
    off += (g.gx[m] * g.gy[m] + 7) & ~7;
    flop += 2.0 * C * C * K * (double)L;
  }
  g.off[3] = off;
  dim3 grid(off, 1, 1);
#define LAUNCH hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_group_kernel<11, 7, 3, CG_CI, CG_MB, CG_NB, CG_WN, CG_KS, 56, 76, 16, CG_WM>), grid, dim3(64 * CG_WM * CG_WN * CG_KS), 0, 0, g)
  for (int i = 0; i < 3; ++i) LAUNCH;
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < N; ++i) LAUNCH;
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / N;
  printf("conv_group<CI=%d MB=%d NB=%d WN=%d KS=%d WM=%d> C=%d L=%d wgs=%d: %.1f us/launch  %.1f TFLOP/s (%.3f of 157.3)\n", CG_CI, CG_MB, CG_NB, CG_WN,
         CG_KS, CG_WM, C, L, off, us, flop / us / 1e6, flop / us / 1e6 / 157.3);
  return 0;
}
