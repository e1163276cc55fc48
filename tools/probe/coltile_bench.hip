// Phase stamps + back-to-back timing of the column-owner launches (csrc/coltile.h) on the GlowTTS shapes:
//   glow_tail_kernel : H = 192, half = 80, 312 decoder columns (20 workgroups)
//   oproj_ln_kernel  : H = 192, 120 encoder columns (8 workgroups)
// The launches rotate through NSETS weight sets (default 48: 16 MB, four L2s' worth) so that every launch finds its
// weights cold in L2, as in the pipeline (114 MB of GlowTTS weights cycle through per utterance).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 coltile_bench.hip -o coltile_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__device__ long long col_stamps[8];
#define COL_STAMP(n) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) col_stamps[n] = wall_clock64(); } while (0)
#include "../../larynx_amd/csrc/coltile.h"
#include "../../larynx_amd/csrc/weights_pack.h"
using namespace mi355tts;
#ifndef NSETS
#define NSETS 48
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <class F>
static float time_us(F launch, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch(i);
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) launch(i);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / n;
}
static float frand() { return rand() / (float)RAND_MAX - 0.5f; }
static float* upload(const std::vector<float>& v) {
  float* d = nullptr;
  if (hipMalloc(&d, v.size() * 4) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  return d;
}
int main() {
  const int H = 192, half = 80, C = 2 * half, T = 312, Tld = (T + 3) & ~3, P = 120, Pld = (P + 3) & ~3;
  srand(5);
  std::vector<float> acts((size_t)H * Tld), skip((size_t)H * Tld), z((size_t)C * Tld);
  for (auto& v : acts) v = frand();
  for (auto& v : skip) v = frand();
  for (auto& v : z) v = frand();
  float *d_acts = upload(acts), *d_skip = upload(skip), *d_z = upload(z), *d_h = upload(acts), *d_z0 = upload(z);
  std::vector<float> w_rs((size_t)H * H), b_rs(H), w_end((size_t)C * H), b_end(C), w_st((size_t)H * half), b_st(H), mixw(16), mixb(C), mixs(C);
  std::vector<GlowTailArgs> sets(NSETS);
  for (int s = 0; s < NSETS; ++s) {
    for (auto& v : w_rs) v = frand() * 0.1f;
    for (auto& v : w_end) v = frand() * 0.1f;
    for (auto& v : w_st) v = frand() * 0.1f;
    for (auto& v : b_rs) v = frand() * 0.1f;
    for (auto& v : b_end) v = frand() * 0.1f;
    for (auto& v : b_st) v = frand() * 0.1f;
    for (auto& v : mixw) v = frand();
    for (auto& v : mixb) v = frand() * 0.1f;
    for (auto& v : mixs) v = 1.f + frand() * 0.1f;
    PackedCol16 p_rs = pack_col16(H, H, [&](int r, int k) { return w_rs[(size_t)r * H + k]; }, [&](int r) { return b_rs[r]; }, true);
    PackedCol16 p_end = pack_col16(C, H, [&](int r, int k) { return w_end[(size_t)r * H + k]; }, [&](int r) { return b_end[r]; }, true);
    PackedCol16 p_st = pack_col16(H, half, [&](int r, int k) { return w_st[(size_t)r * half + k]; }, [&](int r) { return b_st[r]; }, true);
    GlowTailArgs& a = sets[s];
    memset(&a, 0, sizeof(a));
    a.acts = d_acts; a.skip = d_skip; a.hnext = d_h; a.h_bs = (long long)H * Tld; a.h_ld = Tld;
    a.z = d_z; a.z_bs = (long long)C * Tld; a.z_ld = Tld; a.len_const = T; a.len_mul = 1;
    a.w_rs = upload(p_rs.w); a.b_rs = upload(p_rs.bias); a.w_end = upload(p_end.w); a.b_end = upload(p_end.bias);
    a.w_st = upload(p_st.w); a.b_st = upload(p_st.bias); a.mix_w = upload(mixw); a.mix_bias = upload(mixb); a.mix_scale = upload(mixs);
    a.H = H; a.half = half;
  }
  // ---- correctness of the last set against a float64 host reference (z restored first)
  {
    CK(hipMemcpy(d_z, d_z0, z.size() * 4, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(glow_tail_kernel, dim3((T + COL_T - 1) / COL_T, 1), dim3(512), 0, 0, sets[NSETS - 1]);
    CK(hipDeviceSynchronize());
    std::vector<float> zo(z.size()), ho(acts.size());
    CK(hipMemcpy(zo.data(), d_z, zo.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ho.data(), d_h, ho.size() * 4, hipMemcpyDeviceToHost));
    double dz = 0, dh = 0;
    std::vector<double> s(H), e(C), zn(C);
    for (int t = 0; t < T; ++t) {
      for (int r = 0; r < H; ++r) {
        double v = b_rs[r];
        for (int k = 0; k < H; ++k) v += (double)w_rs[(size_t)r * H + k] * acts[(size_t)k * Tld + t];
        s[r] = v + skip[(size_t)r * Tld + t];
      }
      for (int r = 0; r < C; ++r) {
        double v = b_end[r];
        for (int k = 0; k < H; ++k) v += (double)w_end[(size_t)r * H + k] * s[k];
        e[r] = v;
      }
      for (int k = 0; k < half / 2; ++k) {
        const int c0 = 2 * k;
        const double in[4] = {z[(size_t)c0 * Tld + t], z[(size_t)(c0 + 1) * Tld + t], (z[(size_t)(half + c0) * Tld + t] - e[c0]) * exp(-e[half + c0]),
                              (z[(size_t)(half + c0 + 1) * Tld + t] - e[c0 + 1]) * exp(-e[half + c0 + 1])};
        const int ch[4] = {c0, c0 + 1, half + c0, half + c0 + 1};
        for (int m = 0; m < 4; ++m) {
          double o = 0;
          for (int i = 0; i < 4; ++i) o += (double)mixw[m * 4 + i] * in[i];
          zn[ch[m]] = (o - mixb[ch[m]]) * mixs[ch[m]];
        }
      }
      for (int c = 0; c < C; ++c) dz = fmax(dz, fabs(zn[c] - zo[(size_t)c * Tld + t]));
      for (int r = 0; r < H; ++r) {
        double v = b_st[r];
        for (int k = 0; k < half; ++k) v += (double)w_st[(size_t)r * half + k] * zn[k];
        dh = fmax(dh, fabs(v - ho[(size_t)r * Tld + t]));
      }
    }
    printf("glow_tail vs float64 host reference: max|dz| = %.3g, max|dh| = %.3g\n", dz, dh);
  }
  // ---- oproj_ln sets
  std::vector<float> x((size_t)H * Pld), res((size_t)H * Pld), gamma(H), beta(H), wo((size_t)H * H), bo(H);
  for (auto& v : x) v = frand();
  for (auto& v : res) v = frand();
  float *d_x = upload(x), *d_res = upload(res), *d_y = upload(res);
  std::vector<OprojLnArgs> osets(NSETS);
  for (int s = 0; s < NSETS; ++s) {
    for (auto& v : wo) v = frand() * 0.1f;
    for (auto& v : bo) v = frand() * 0.1f;
    for (auto& v : gamma) v = 1.f + frand() * 0.1f;
    for (auto& v : beta) v = frand() * 0.1f;
    PackedCol16 p = pack_col16(H, H, [&](int r, int k) { return wo[(size_t)r * H + k]; }, [&](int r) { return bo[r]; }, true);
    OprojLnArgs& a = osets[s];
    memset(&a, 0, sizeof(a));
    a.x = d_x; a.res = d_res; a.y = d_y; a.bs = (long long)H * Pld; a.ld = Pld; a.len_const = P; a.len_mul = 1;
    a.w = upload(p.w); a.b = upload(p.bias); a.gamma = upload(gamma); a.beta = upload(beta); a.H = H; a.eps = 1e-4f;
  }
  const dim3 gt((T + COL_T - 1) / COL_T, 1), go((P + COL_T - 1) / COL_T, 1);
  auto tail = [&](int i) { hipLaunchKernelGGL(glow_tail_kernel, gt, dim3(512), 0, 0, sets[i % NSETS]); };
  auto tail_hot = [&](int) { hipLaunchKernelGGL(glow_tail_kernel, gt, dim3(512), 0, 0, sets[0]); };
  auto oln = [&](int i) { hipLaunchKernelGGL(oproj_ln_kernel, go, dim3(512), 0, 0, osets[i % NSETS]); };
  auto oln_hot = [&](int) { hipLaunchKernelGGL(oproj_ln_kernel, go, dim3(512), 0, 0, osets[0]); };
  for (int r = 0; r < 2; ++r) {
    const float t_cold = time_us(tail, 960);
    long long st[8];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(col_stamps), sizeof(st));
    printf("glow_tail  rotating %d weight sets: %.2f us per launch; stamps (10 ns ticks): loads issued %lld, landed %lld, barrier %lld, rs %lld, end %lld, coupling %lld, start %lld\n",
           NSETS, t_cold, st[6] - st[0], st[7] - st[6], st[1] - st[7], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4]);
    printf("glow_tail  one weight set (L2-hot): %.2f us per launch\n", time_us(tail_hot, 960));
    printf("oproj_ln   rotating: %.2f us per launch;  one set: %.2f\n", time_us(oln, 960), time_us(oln_hot, 960));
  }
  CK(hipDeviceSynchronize());
  return 0;
}
