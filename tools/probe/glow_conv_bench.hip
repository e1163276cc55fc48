// Phase stamps + back-to-back timing of the k-split conv tile on the GlowTTS decoder's shapes (312 columns):
//   gate  : k = 5, 192 -> 2 x 192 rows, tanh * sigmoid epilogue   conv_mfma_kernel<5,64,1,1,1,8,28,EPI_GATE>
//   1 x 1 : 192 -> 384 rows, linear                               conv_mfma_kernel<1,64,1,1,1,8,0,EPI_LINEAR>
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 glow_conv_bench.hip -o glow_conv_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
__device__ long long conv_stamps[8];
#define CONV_STAMP(n) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) conv_stamps[n] = wall_clock64(); } while (0)
#define GATE_STAMP(n) CONV_STAMP(n)
#include "../../larynx_amd/csrc/conv_mfma.h"
#include "../../larynx_amd/csrc/gate16.h"
#include "../../larynx_amd/csrc/weights_pack.h"
using namespace mi355tts;
#ifndef GC_T
#define GC_T 312
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <class F>
static float time_us(F launch, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / n;
}
static void show(const char* name, float us) {
  long long st[8];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(conv_stamps), sizeof(st));
  printf("%-28s %.2f us per launch; stamps (10 ns ticks): issue %lld, staged+barrier %lld, main loop %lld, reduce %lld, epilogue %lld\n", name, us,
         st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4]);
}
int main() {
  const int H = 192, T = GC_T, Tld = (T + 3) & ~3;
  srand(3);
  std::vector<float> x((size_t)H * Tld);
  for (auto& v : x) v = rand() / (float)RAND_MAX - 0.5f;
  float *dx, *dy, *dy2;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dy, 2 * x.size() * 4)); CK(hipMalloc(&dy2, 2 * x.size() * 4));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  auto make = [&](int K, int vrows, auto rowmap, ConvArgs& a) -> int {
    std::vector<float> w((size_t)2 * H * H * K), b(2 * H, 0.01f);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    PackedConv p = pack_conv(vrows, 1, H, K, rowmap, [&](int co, int ci, int k) { return w[((size_t)co * H + ci) * K + k]; }, [&](int co) { return b[co]; }, true, 8);
    float *dw, *db;
    CK(hipMalloc(&dw, p.w.size() * 4)); CK(hipMalloc(&db, p.bias.size() * 4));
    CK(hipMemcpy(dw, p.w.data(), p.w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, p.bias.data(), p.bias.size() * 4, hipMemcpyHostToDevice));
    memset(&a, 0, sizeof(a));
    a.x = dx; a.x_bs = (long long)H * Tld; a.x_ld = Tld; a.in_const = T; a.in_mul = 1;
    a.w = dw; a.bias = db; a.noct = p.noct; a.Cin = H; a.rows = vrows; a.dil = 1; a.pad = (K - 1) / 2; a.in_slope = 1.f;
    a.y_bs = (long long)2 * H * Tld; a.y_ld = Tld; a.split = 1 << 30; a.alpha = 1.f; a.out_const = T; a.out_mul = 1; a.half = H;
    return 0;
  };
  ConvArgs ag, al;
  if (make(5, 2 * H, [&](int v) { const int p = v / 32, i = v % 32; return i < 16 ? 16 * p + i : H + 16 * p + (i - 16); }, ag)) return 1;
  ag.y = dy;
  if (make(1, 2 * H, [&](int v) { return v; }, al)) return 1;
  al.y = dy2;
  // the same gate conv for gate16_kernel (same w / bias values: make() draws them first, so redraw identically)
  Gate16Args g16;
  {
    srand(3);
    for (size_t i = 0; i < x.size(); ++i) (void)rand();
    std::vector<float> w((size_t)2 * H * H * 5), b(2 * H, 0.01f);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    PackedGate16 p = pack_gate16(H, H, 5, [&](int co, int ci, int k) { return w[((size_t)co * H + ci) * 5 + k]; }, [&](int co) { return b[co]; }, true);
    float *dw, *db;
    CK(hipMalloc(&dw, p.w.size() * 4)); CK(hipMalloc(&db, p.bias.size() * 4));
    CK(hipMemcpy(dw, p.w.data(), p.w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, p.bias.data(), p.bias.size() * 4, hipMemcpyHostToDevice));
    memset(&g16, 0, sizeof(g16));
    g16.x = dx; g16.x_bs = (long long)H * Tld; g16.x_ld = Tld; g16.len_const = T; g16.len_mul = 1;
    g16.w = dw; g16.bias = db; g16.Cin = H; g16.half = H; g16.dil = 1; g16.pad = 2;
    CK(hipMalloc(&g16.y, x.size() * 4)); g16.y_bs = (long long)H * Tld; g16.y_ld = Tld;
    CK(hipMemset(g16.y, 0, x.size() * 4));
  }
  const dim3 grid16((T + 31) / 32, H / 8, 1);
  auto gate16 = [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(gate16_kernel<5, 6>), grid16, dim3(512), 0, 0, g16); };
  const dim3 grid((T + 31) / 32, 2 * H / 32, 1);
  auto gate = [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_kernel<5, 64, 1, 1, 1, 8, 28, EPI_GATE, 1>), grid, dim3(512), 0, 0, ag); };
  auto lin = [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_kernel<1, 64, 1, 1, 1, 8, 0, EPI_LINEAR, 1>), grid, dim3(512), 0, 0, al); };
  for (int r = 0; r < 2; ++r) {
    show("gate k=5 192->2x192", time_us(gate, 500));
    show("gate16 k=5 192->2x192", time_us(gate16, 500));
    show("1x1 192->384", time_us(lin, 500));
    int f = 0;
    printf("gate, 1x1 alternating        %.2f us per launch\n", time_us([&] { if (f ^= 1) gate(); else lin(); }, 1000));
  }
  CK(hipDeviceSynchronize());
  {  // gate16 against the 32-row tile
    std::vector<float> y0((size_t)H * Tld), y1((size_t)H * Tld);
    CK(hipMemset(dy, 0, y0.size() * 4));
    gate(); gate16();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y0.data(), dy, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), g16.y, y1.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (int c = 0; c < H; ++c)
      for (int t = 0; t < T; ++t) {
        md = fmax(md, fabs((double)y0[(size_t)c * Tld + t] - y1[(size_t)c * Tld + t]));
        mx = fmax(mx, fabs((double)y0[(size_t)c * Tld + t]));
      }
    printf("gate16 vs 32-row tile: max|diff| = %.3g (max|y| = %.3g)\n", md, mx);
    // host reference
    srand(3);
    for (size_t i = 0; i < x.size(); ++i) (void)rand();
    std::vector<float> w((size_t)2 * H * H * 5);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    double d0 = 0, d1 = 0;
    for (int c = 0; c < H; ++c)
      for (int t = 0; t < T; ++t) {
        double s0 = 0.01, s1 = 0.01;
        for (int ci = 0; ci < H; ++ci)
          for (int k = 0; k < 5; ++k) {
            const int tt = t + k - 2;
            if (tt < 0 || tt >= T) continue;
            s0 += (double)w[((size_t)c * H + ci) * 5 + k] * x[(size_t)ci * Tld + tt];
            s1 += (double)w[((size_t)(H + c) * H + ci) * 5 + k] * x[(size_t)ci * Tld + tt];
          }
        const double ref = tanh(s0) / (1.0 + exp(-s1));
        d0 = fmax(d0, fabs(ref - y0[(size_t)c * Tld + t]));
        d1 = fmax(d1, fabs(ref - y1[(size_t)c * Tld + t]));
      }
    printf("against the host reference: 32-row tile %.3g, gate16 %.3g\n", d0, d1);
  }
  return 0;
}
