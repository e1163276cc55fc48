// Harness of conv_f16.h (the native 16-bit vocoder tile): one conv / upsampler launch on random data, checked against a CPU
// reference in double (fp16-rounded operands), then timed.  NOT part of the product library.
//   GPU:      hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/f16_bench.hip -o build/f16_bench
//   emulator: clang++ -x c++ -std=c++17 -O2 -Itests/hipemu/include tools/probe/f16_bench.hip tests/hipemu/hipemu_runtime.cpp -lpthread
// usage: f16_bench <Cin> <Cout> <K> <dil> <L> <cfg> [iters] [up]     (up > 0: polyphase upsampler with K = 2 taps)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../larynx_amd/csrc/pair_f16.h"
#include "../../larynx_amd/csrc/weights_pack.h"

using namespace mi355tts;

#define HC(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <int K, int EPI, bool MRF>
static void launch(int cfg, dim3& grid_out, const HConvArgs& a, int n_len, hipStream_t s, bool really) {
  constexpr int HALO = K == 2 ? 4 : ConvHalo<K>::v;
#define CFG(ID, MB, NB, WM, WN, CH)                                                                                          \
  if (cfg == ID) {                                                                                                           \
    grid_out = dim3((n_len + 32 * NB * WN - 1) / (32 * NB * WN), (a.rows + 32 * MB * WM - 1) / (32 * MB * WM), 1);            \
    if (really) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_kernel<K, MB, NB, WM, WN, HALO, CH, EPI, MRF>), grid_out, dim3(64 * WM * WN), 0, s, a); \
    return;                                                                                                                  \
  }
  CFG(0, 2, 2, 2, 2, 64)  // 128 rows x 128 columns, wave 64 x 64
  CFG(1, 2, 4, 2, 2, 64)  // 128 x 256, wave 64 x 128
  CFG(2, 2, 2, 1, 4, 64)  // 64 x 256
  CFG(3, 1, 2, 1, 4, 32)  // 32 x 256
  CFG(4, 2, 2, 2, 2, 32)
  CFG(5, 2, 4, 1, 4, 32)  // 64 x 512
  CFG(6, 1, 4, 1, 4, 32)  // 32 x 512
#undef CFG
  std::printf("unknown cfg %d\n", cfg);
}

// grouped launch of the three MRF members (K = 11, 7, 3), timing only: f16_bench Cin Cout 0 dil L cfg iters
template <int MB, int NB, int WM, int WN, int CH>
static void launch_group(const HConvGroupArgs& g, dim3 grid) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_group_kernel<11, 7, 3, MB, NB, WM, WN, ConvHalo<11>::v, ConvHalo<7>::v, ConvHalo<3>::v, CH>), grid,
                     dim3(64 * WM * WN), 0, nullptr, g);
}
static int run_group(int C, int dil, int L, int cfg, int iters) {
  const int Ks[3] = {11, 7, 3};
  const int noct = C / 8, ld = L + 3;
  uint4 *dx, *dy[3], *dw[3];
  float* db;
  HC(hipMalloc(&dx, (size_t)noct * ld * 16));
  HC(hipMemset(dx, 0, (size_t)noct * ld * 16));
  HC(hipMalloc(&db, 4 * 1024));
  HC(hipMemset(db, 0, 4 * 1024));
  HConvGroupArgs g;
  std::memset(&g, 0, sizeof(g));
  int MBr = 2, NBc = 2, WMr = 2, WNc = 2;
  if (cfg == 1) NBc = 4;
  if (cfg == 2) { WMr = 1; WNc = 4; }
  if (cfg == 3) { MBr = 1; WMr = 1; WNc = 4; }
  if (cfg == 5) { NBc = 4; WMr = 1; WNc = 4; }
  if (cfg == 6) { MBr = 1; NBc = 4; WMr = 1; WNc = 4; }
  const int tcols = 32 * NBc * WNc, trows = 32 * MBr * WMr;
  int off = 0;
  double flop = 0;
  for (int m = 0; m < 3; ++m) {
    const int K = Ks[m];
    std::vector<uint16_t> w((size_t)((C + 31) / 32) * 4 * (C / 16 + 4) * K * 64 * 8, f16_rne(0.01f));
    HC(hipMalloc(&dw[m], w.size() * 2));
    HC(hipMemcpy(dw[m], w.data(), w.size() * 2, hipMemcpyHostToDevice));
    HC(hipMalloc(&dy[m], (size_t)noct * ld * 16));
    HConvArgs& a = g.c[m];
    a.x = dx; a.x_bs = (long long)noct * ld; a.x_ld = ld; a.in_const = L; a.w = dw[m]; a.bias = db;
    a.nslab = 4 * ((C + 63) / 64); a.Cin = C; a.rows = C; a.dil = dil; a.pad = (K * dil - dil) / 2; a.in_slope = 0.1f; a.out_slope = 1.0f;
    a.y = dy[m]; a.y_bs = (long long)noct * ld; a.y_ld = ld; a.res = dx; a.out_const = L; a.cout = C;
    g.gx[m] = (L + tcols - 1) / tcols;
    g.gy[m] = (C + trows - 1) / trows;
    g.off[m] = off;
    off += (g.gx[m] * g.gy[m] + 7) & ~7;
    flop += 2.0 * C * C * K * (double)L;
  }
  g.off[3] = off;
  const dim3 grid(off, 1, 1);
  auto go = [&]() {
    if (cfg == 0) launch_group<2, 2, 2, 2, 64>(g, grid);
    else if (cfg == 1) launch_group<2, 4, 2, 2, 64>(g, grid);
    else if (cfg == 2) launch_group<2, 2, 1, 4, 64>(g, grid);
    else if (cfg == 3) launch_group<1, 2, 1, 4, 32>(g, grid);
    else if (cfg == 4) launch_group<2, 2, 2, 2, 32>(g, grid);
    else if (cfg == 5) launch_group<2, 4, 1, 4, 32>(g, grid);
    else launch_group<1, 4, 1, 4, 32>(g, grid);
  };
  for (int i = 0; i < 3; ++i) go();
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  HC(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) go();
  HC(hipEventRecord(e1, nullptr));
  HC(hipEventSynchronize(e1));
  float ms = 0;
  HC(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  std::printf("GROUP C %d dil %d L %d cfg %d (%d workgroups): %.2f us per launch, %.1f TFLOP/s (%.3f of 2500)\n", C, dil, L, cfg, off, us, flop / us * 1e-6,
              flop / us * 1e-6 / 2500.0);
  return 0;
}

// fused conv1 + conv2 steps of the three MRF members (pair_f16.h), timing only: f16_bench C C -1 dil L tile iters   (tile 0 WIDE, 1 MID, 2 SLIM)
#ifndef PROBE_PAIR_MINW
#define PROBE_PAIR_MINW 3
#endif
static int run_pair_group(int C, int dil, int L, int tile, int iters) {
  const int Ks[3] = {11, 7, 3};
  const int noct = C / 8, ld = L + 3;
  uint4 *dx, *dy[3], *dw[3][2];
  float* db;
  HC(hipMalloc(&dx, (size_t)noct * ld * 16));
  {
    std::vector<uint16_t> hx((size_t)noct * ld * 8);
    uint32_t st = 777u;
    for (auto& v : hx) {
      st = st * 1664525u + 1013904223u;
      v = f16_rne(((st >> 8) * (1.0f / 16777216.0f)) * 4.0f - 2.0f);
    }
    HC(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  }
  HC(hipMalloc(&db, 4 * 1024));
  HC(hipMemset(db, 0, 4 * 1024));
  HPairGroupArgs g;
  std::memset(&g, 0, sizeof(g));
  g.n = 3;
  const int tcols = tile == 0 ? 128 : tile == 5 ? 512 : 256;  // tile 4: 128 rows x 256 columns (waves of 64 x 128)
  int off = 0;
  double flop = 0;
  for (int m = 0; m < 3; ++m) {
    const int K = Ks[m];
    std::vector<uint16_t> w((size_t)4 * (4 * ((C + 63) / 64)) * K * 64 * 8 + 4096);
    {
      uint32_t st = 4242u + m;
      for (auto& v : w) {
        st = st * 1664525u + 1013904223u;
        v = f16_rne((((st >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f) * 0.03f);
      }
    }
    for (int c = 0; c < 2; ++c) {
      HC(hipMalloc(&dw[m][c], w.size() * 2));
      HC(hipMemcpy(dw[m][c], w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    HC(hipMalloc(&dy[m], (size_t)noct * ld * 16));
    HPairArgs& a = g.p[m];
    a.x = dx; a.y = dy[m]; a.bs = (long long)noct * ld; a.ld = ld; a.len_const = L;
    a.w1 = dw[m][0]; a.b1 = db; a.nslab1 = 4 * ((C + 63) / 64);
    a.w2 = dw[m][1]; a.b2 = db; a.nslab2 = a.nslab1;
    a.C = C; a.dil = dil; a.slope = 0.1f;
    const int to = tcols - (K - 1);
    g.gx[m] = (L + to - 1) / to;
    g.off[m] = off;
    off += (g.gx[m] + 7) & ~7;
    flop += 4.0 * C * C * K * (double)L;
  }
  g.off[3] = off;
  const dim3 grid(off, 1, 1);
  auto go = [&]() {
    constexpr int H0 = ConvHalo<11>::v, H1 = ConvHalo<7>::v, H2 = ConvHalo<3>::v;
    if (tile == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<11, 7, 3, 2, 2, 2, 2, H0, H1, H2, 32, 3, PROBE_PAIR_MINW>), grid, dim3(256), 0, nullptr, g);
    else if (tile == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<11, 7, 3, 2, 2, 1, 4, H0, H1, H2, 32, 2, PROBE_PAIR_MINW>), grid, dim3(256), 0, nullptr, g);
    else if (tile == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<11, 7, 3, 2, 4, 2, 2, H0, H1, H2, 32, 3, 1>), grid, dim3(256), 0, nullptr, g);
    else if (tile == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<11, 7, 3, 2, 4, 1, 4, H0, H1, H2, 32, 2, 1>), grid, dim3(256), 0, nullptr, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<11, 7, 3, 1, 2, 1, 4, H0, H1, H2, 32, 1, 4>), grid, dim3(256), 0, nullptr, g);
  };
  for (int i = 0; i < 3; ++i) go();
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  HC(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) go();
  HC(hipEventRecord(e1, nullptr));
  HC(hipEventSynchronize(e1));
  float ms = 0;
  HC(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters;
  std::printf("PAIR GROUP C %d dil %d L %d tile %d (%d workgroups): %.2f us per launch, %.1f TFLOP/s (%.3f of 2500)\n", C, dil, L, tile, off, us, flop / us * 1e-6,
              flop / us * 1e-6 / 2500.0);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    std::printf("usage: f16_bench Cin Cout K dil L cfg [iters] [up]\n");
    return 2;
  }
  const int Cin = std::atoi(argv[1]), Cout = std::atoi(argv[2]), K = std::atoi(argv[3]), dil = std::atoi(argv[4]), L = std::atoi(argv[5]);
  const int cfg = std::atoi(argv[6]);
  const int iters = argc > 7 ? std::atoi(argv[7]) : 20;
  const int up = argc > 8 ? std::atoi(argv[8]) : 0;
  const bool mrf = argc > 9 && std::atoi(argv[9]) != 0;
  if (K == 0) return run_group(Cin, dil, L, cfg, iters);
  if (K < 0) return run_pair_group(Cin, dil, L, cfg, iters);
  uint32_t st = 12345u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    return ((st >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f;
  };
  const int noct = (Cin + 7) / 8, ld = L + 3;
  const int npl = mrf ? 3 : 1;
  std::vector<uint16_t> hx((size_t)npl * noct * ld * 8, 0);
  std::vector<float> fx((size_t)npl * Cin * L);
  for (int p = 0; p < npl; ++p)
    for (int c = 0; c < Cin; ++c)
      for (int t = 0; t < L; ++t) {
        const uint16_t h = f16_rne(rnd() * 2.0f);
        hx[(size_t)p * noct * ld * 8 + ((size_t)(c / 8) * ld + t) * 8 + (c % 8)] = h;
        fx[((size_t)p * Cin + c) * L + t] = f16_to_float(h);
      }
  const int Kt = up ? 2 : K;
  const int rows = up ? Cout * up : Cout;
  // logical weights: conv w[Cout][Cin][K]; transposed conv wt[Cin][Cout][2 up]
  std::vector<float> w((size_t)Cout * Cin * (up ? 2 * up : K)), bias(Cout);
  const float sc = 1.0f / std::sqrt((float)Cin * Kt);
  for (auto& v : w) v = f16_to_float(f16_rne(rnd() * sc));
  for (auto& v : bias) v = rnd();
  PackedConvH pk;
  if (up) {
    pk = pack_conv_f16(
        rows, 4, Cin, 2, 64,
        [&](int v, int ci, int k) {
          const int r = v / Cout, co = v % Cout, m = 1 - k;
          return w[((size_t)ci * Cout + co) * (2 * up) + m * up + r];
        },
        [&](int v) { return bias[v % Cout]; }, true);
  } else {
    pk = pack_conv_f16(
        rows, 4, Cin, K, 64, [&](int v, int ci, int k) { return w[((size_t)v * Cin + ci) * K + k]; }, [&](int v) { return bias[v]; }, true);
  }
  const int Lout = up ? L * up : L;
  const int yoct = Cout / 8, yld = Lout + 5;
  std::vector<uint16_t> hres((size_t)yoct * yld * 8, 0);
  for (auto& v : hres) v = f16_rne(rnd());
  uint4 *dx, *dw, *dy, *dres;
  float* db;
  HC(hipMalloc(&dx, hx.size() * 2));
  HC(hipMalloc(&dw, pk.w.size() * 2));
  HC(hipMalloc(&db, pk.bias.size() * 4));
  HC(hipMalloc(&dy, (size_t)yoct * yld * 16));
  HC(hipMalloc(&dres, (size_t)yoct * yld * 16));
  HC(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  HC(hipMemcpy(dw, pk.w.data(), pk.w.size() * 2, hipMemcpyHostToDevice));
  HC(hipMemcpy(db, pk.bias.data(), pk.bias.size() * 4, hipMemcpyHostToDevice));
  HC(hipMemcpy(dres, hres.data(), hres.size() * 2, hipMemcpyHostToDevice));
  HC(hipMemset(dy, 0, (size_t)yoct * yld * 16));
  HConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = dx;
  if (mrf) {
    a.x2 = dx + (size_t)noct * ld;
    a.x3 = dx + (size_t)2 * noct * ld;
  }
  a.in_div = 3.0f;
  a.x_bs = (long long)noct * ld;
  a.x_ld = ld;
  a.in_const = L;
  a.w = dw;
  a.bias = db;
  a.nslab = pk.nslab;
  a.Cin = Cin;
  a.rows = rows;
  a.dil = up ? 1 : dil;
  a.pad = up ? 1 : (K * dil - dil) / 2;
  a.in_slope = 0.1f;
  a.out_slope = up ? 1.0f : 0.1f;
  a.y = dy;
  a.y_bs = (long long)yoct * yld;
  a.y_ld = yld;
  a.res = up ? nullptr : dres;
  a.out_const = Lout;
  a.up = up;
  a.up_pad = up / 2;
  a.cout = Cout;
  const int n_len = up ? L + 1 : L;
  dim3 grid;
  auto go = [&](bool really) {
    if (up) {
      if (mrf) launch<2, EPI_UPSAMPLE, true>(cfg, grid, a, n_len, nullptr, really);
      else launch<2, EPI_UPSAMPLE, false>(cfg, grid, a, n_len, nullptr, really);
    } else if (K == 3) launch<3, EPI_LINEAR, false>(cfg, grid, a, n_len, nullptr, really);
    else if (K == 7) launch<7, EPI_LINEAR, false>(cfg, grid, a, n_len, nullptr, really);
    else if (K == 11) launch<11, EPI_LINEAR, false>(cfg, grid, a, n_len, nullptr, really);
    else std::printf("K = %d not instantiated\n", K);
  };
  go(true);
  HC(hipDeviceSynchronize());
  std::vector<uint16_t> hy((size_t)yoct * yld * 8);
  HC(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost));
  // reference on a sample of outputs (all of them when the problem is small)
  auto xin = [&](int c, int t) -> double {
    if (t < 0 || t >= L) return 0.0;
    double v = fx[(size_t)c * L + t];
    if (mrf) {  // f32 sum, one rounding
      const float s3 = ((fx[(size_t)c * L + t] + fx[((size_t)Cin + c) * L + t]) + fx[((size_t)2 * Cin + c) * L + t]) * (1.0f / 3.0f);
      v = f16_to_float(f16_rne(s3));
    }
    const float h = (float)v;
    const float act = f16_to_float(f16_rne(h > 0.f ? h : f16_to_float(f16_rne(h * f16_to_float(f16_rne(0.1f))))));
    return act;
  };
  const long long total = (long long)Cout * Lout;
  const long long stride = total > 400000 ? total / 200000 : 1;
  double max_err = 0, max_ref = 0;
  long long checked = 0, bad = 0;
  for (long long idx = 0; idx < total; idx += stride) {
    const int co = (int)(idx / Lout), n = (int)(idx % Lout);
    double acc = bias[co];
    if (up) {
      // out[co][n] = sum_ci sum_k x[ci][q - m] wt[ci][co][m up + r], n = q up + r - up/2
      const int np = n + up / 2;
      for (int kk = 0; kk < 2 * up; ++kk) {
        // transposed conv: out[n] += x[i] wt[kk] where n = i up + kk - up/2
        const int num = np - kk;
        if (num % up) continue;
        const int i = num / up;
        if (i < 0 || i >= L) continue;
        for (int ci = 0; ci < Cin; ++ci) acc += xin(ci, i) * w[((size_t)ci * Cout + co) * (2 * up) + kk];
      }
    } else {
      for (int ci = 0; ci < Cin; ++ci)
        for (int k = 0; k < K; ++k) acc += xin(ci, n + k * dil - a.pad) * w[((size_t)co * Cin + ci) * K + k];
      acc += f16_to_float(hres[((size_t)(co / 8) * yld + n) * 8 + (co % 8)]);
      acc = acc > 0 ? acc : acc * 0.1f;
    }
    const double got = f16_to_float(hy[((size_t)(co / 8) * yld + n) * 8 + (co % 8)]);
    const double err = std::fabs(got - acc);
    max_err = std::fmax(max_err, err);
    max_ref = std::fmax(max_ref, std::fabs(acc));
    if (err > 2e-3 * std::fmax(1.0, std::fabs(acc))) ++bad;
    ++checked;
  }
  std::printf("check: %lld outputs, max |ref| %.3f, max err %.3e, bad %lld  grid %u x %u\n", checked, max_ref, max_err, bad, grid.x, grid.y);
  if (iters > 0) {
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) go(true);
    HC(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) go(true);
    HC(hipEventRecord(e1, nullptr));
    HC(hipEventSynchronize(e1));
    float ms = 0;
    HC(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    const double flop = 2.0 * Cout * Cin * (up ? 2.0 * up : (double)K) * (double)L;
    std::printf("Cin %d Cout %d K %d dil %d L %d cfg %d up %d: %.2f us per launch, %.1f TFLOP/s (%.3f of 2500)\n", Cin, Cout, K, dil, L, cfg, up, us,
                flop / us * 1e-6, flop / us * 1e-6 / 2500.0);
  }
#if F16_STAMPS
  {  // one more launch with the phase stamps read back: per workgroup start offset, prologue, main loop, epilogue, store tail
    const size_t nwg = (size_t)grid.x * grid.y;
    unsigned long long* ds;
    HC(hipMalloc(&ds, nwg * 64));
    HC(hipMemset(ds, 0, nwg * 64));
    HC(hipMemcpyToSymbol(HIP_SYMBOL(g_f16_stamps), &ds, sizeof(ds)));
    go(true);
    HC(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nwg * 8);
    HC(hipMemcpy(st.data(), ds, nwg * 64, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0;
    for (size_t i = 0; i < nwg; ++i)
      if (st[i * 8]) { t0 = std::min(t0, st[i * 8]); tend = std::max(tend, st[i * 8 + 4]); }
    double sum[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
    size_t n = 0;
    for (size_t i = 0; i < nwg; ++i) {
      if (!st[i * 8]) continue;
      const double v[5] = {(double)(st[i * 8] - t0), (double)(st[i * 8 + 1] - st[i * 8]), (double)(st[i * 8 + 2] - st[i * 8 + 1]),
                           (double)(st[i * 8 + 3] - st[i * 8 + 2]), (double)(st[i * 8 + 4] - st[i * 8 + 3])};
      for (int k = 0; k < 5; ++k) { sum[k] += v[k]; mx[k] = std::max(mx[k], v[k]); }
      ++n;
    }
    {
      double c1 = 0, c2 = 0;
      for (size_t i = 0; i < nwg; ++i) { c1 += (double)(st[i * 8 + 6] - st[i * 8 + 5]); c2 += (double)(st[i * 8 + 7] - st[i * 8 + 6]); }
      std::printf("chunk 2: top -> before barrier avg %.0f, barrier avg %.0f\n", c1 / nwg, c2 / nwg);
    }
    std::printf("stamps (ticks; %zu workgroups; launch span %llu): start offset avg %.0f max %.0f | prologue avg %.0f max %.0f | main loop avg %.0f max %.0f | epilogue avg %.0f max %.0f | store tail avg %.0f max %.0f\n",
                n, tend - t0, sum[0] / n, mx[0], sum[1] / n, mx[1], sum[2] / n, mx[2], sum[3] / n, mx[3], sum[4] / n, mx[4]);
  }
#endif
  return bad ? 1 : 0;
}
