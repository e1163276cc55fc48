// Standalone timing + CPU check of attention_mfma_kernel on one encoder layer's shapes
// (H = 192, 2 heads, window 4, P ids).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 att_bench.hip -o att_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ long long att_stamps[16];
#define ATT_STAMP(n) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) att_stamps[n] = wall_clock64(); } while (0)
#include "../../larynx_amd/csrc/small_kernels.h"
using namespace mi355tts;
#ifndef ATT_EXACT
#define ATT_EXACT true
#endif
#ifndef ATT_P
#define ATT_P 120
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int H = 192, nh = 2, dk = H / nh, win = 4, nrel = 2 * win + 1, P = ATT_P, B = 1;
  std::vector<float> qkv((size_t)3 * H * P), ek(nrel * dk), ev(nrel * dk), ref((size_t)H * P), got((size_t)H * P);
  srand(7);
  auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
  for (auto& x : qkv) x = rnd();
  for (auto& x : ek) x = rnd();
  for (auto& x : ev) x = rnd();
  for (int h = 0; h < nh; ++h)
    for (int i = 0; i < P; ++i) {
      std::vector<double> s(P);
      double mx = -1e30;
      for (int j = 0; j < P; ++j) {
        double a = 0;
        for (int c = 0; c < dk; ++c) {
          const double qq = qkv[(size_t)(h * dk + c) * P + i];
          a += qq * qkv[(size_t)(H + h * dk + c) * P + j];
          if (abs(j - i) <= win) a += qq * ek[(j - i + win) * dk + c];
        }
        s[j] = a / sqrt((double)dk);
        mx = fmax(mx, s[j]);
      }
      double den = 0;
      for (int j = 0; j < P; ++j) { s[j] = exp(s[j] - mx); den += s[j]; }
      for (int c = 0; c < dk; ++c) {
        double o = 0;
        for (int j = 0; j < P; ++j) {
          o += s[j] / den * qkv[(size_t)(2 * H + h * dk + c) * P + j];
          if (abs(j - i) <= win) o += s[j] / den * ev[(j - i + win) * dk + c];
        }
        ref[(size_t)(h * dk + c) * P + i] = (float)o;
      }
    }
  float *dq, *dek, *dev, *dout;
  int* dlen;
  CK(hipMalloc(&dq, qkv.size() * 4)); CK(hipMalloc(&dek, ek.size() * 4)); CK(hipMalloc(&dev, ev.size() * 4));
  CK(hipMalloc(&dout, got.size() * 4)); CK(hipMalloc(&dlen, 4));
  CK(hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dek, ek.data(), ek.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dev, ev.data(), ev.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlen, &P, 4, hipMemcpyHostToDevice));
  const dim3 ag((P + 31) / 32, nh, B);
  auto launch = [&] {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_mfma_kernel<48, ATT_EXACT>), ag, dim3(512), 0, 0,
 dq, (long long)3 * H * P, P, dlen, H, nh, win,
                       dek, dev, dout, (long long)H * P, P);
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
  double md = 0;
  for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs((double)got[i] - ref[i]));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipEventRecord(e0, 0));
  const int N = 500;
  for (int i = 0; i < N; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  long long st[16];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(att_stamps), sizeof(st)));
  printf("stamps (10 ns ticks; 1 = loads issued, 3 = QK^T done, 4 = V parked, 5 = band, 6 = softmax, 7 = PV done, 8 = stored):");
  for (int i = 3; i <= 8; ++i) printf(" %lld", st[i] - st[1]);
  printf("\n");
  printf("P=%d max|diff|=%.3g  %.2f us per launch (back to back)\n", P, md, ms * 1000.0 / N);
  return md < 1e-4 ? 0 : 2;
}
