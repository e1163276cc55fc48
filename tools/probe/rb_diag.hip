// Diagnostics of the grouped ResBlock-conv launch (conv_group_kernel) on a standalone harness:
//   * back-to-back timing at L, 2L, 4L, 8L columns (what the ramp and the tail of a batch-1 launch cost),
//   * S concurrent streams of the same launch (does the hardware fill one launch's tail with the next launch's head?),
//   * a per-workgroup timeline (begin / end in s_memrealtime and s_memtime ticks, HW_ID, XCC_ID) dumped to a file,
//   * per-chunk phase stamps of sampled workgroups (-DRB_CHUNK_STAMPS),
//   * dispatch-order variants: -DRB_ORDER=1 interleaves the three members in runs of 8 workgroups, -DRB_ORDER=2 is the
//     product's snake order for all-resident launches (group_snake_order), -DRB_PERSIST=1 runs
//     the tiles from an atomic ticket counter on a grid of RB_PERSIST_WGS workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCG_C=128 -DCG_L=39488 -DCG_CI=16 ... ] tools/probe/rb_diag.hip -o /tmp/rbd
// Run:   /tmp/rbd [dump-file]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
struct WgRec { long long real0, clk0, real1, clk1; unsigned hwid, xcc; };
__device__ WgRec* rb_wg_rec;
__device__ long long* rb_chunk_rec;  // [sample][wave 0..15][chunk 0..15][4]
__device__ long long* rb_phase_rec;  // [workgroup][4]: rb_tile's RB_STAMP(0..3) = begin, staged, main loop done, end (s_memtime)
#define RB_STAMP(n) do { if (threadIdx.x == 0 && rb_phase_rec) rb_phase_rec[(size_t)blockIdx.x * 4 + (n)] = clock64(); } while (0)
#ifndef RB_CHUNK_EVERY
#define RB_CHUNK_EVERY 61
#endif
#define CONV_WG_STAMP(lin, which)                                                                 \
  do {                                                                                            \
    if (threadIdx.x == 0 && rb_wg_rec) {                                                          \
      WgRec* r = rb_wg_rec + (lin);                                                               \
      if ((which) == 0) {                                                                         \
        r->real0 = wall_clock64(); r->clk0 = clock64();                                           \
        r->hwid = __builtin_amdgcn_s_getreg(63492); r->xcc = __builtin_amdgcn_s_getreg(63508);    \
      } else { r->real1 = wall_clock64(); r->clk1 = clock64(); }                                  \
    }                                                                                             \
  } while (0)
#ifdef RB_CHUNK_STAMPS
#define CONV_CHUNK_STAMP(chunk, which)                                                            \
  do {                                                                                            \
    if ((threadIdx.x & 63) == 0 && rb_chunk_rec && (blockIdx.x % RB_CHUNK_EVERY) == 3 && (chunk) < 16) \
      rb_chunk_rec[((((size_t)(blockIdx.x / RB_CHUNK_EVERY)) * 16 + (threadIdx.x >> 6)) * 16 + (chunk)) * 4 + (which)] = clock64(); \
  } while (0)
#endif
#include "../../larynx_amd/csrc/conv_mfma.h"
#include "../../larynx_amd/csrc/rb_conv.h"
#include "../../larynx_amd/csrc/weights_pack.h"
using namespace mi355tts;
#ifndef CG_C
#define CG_C 128
#endif
#ifndef CG_L
#define CG_L 39488
#endif
#ifndef CG_CI
#define CG_CI 16
#endif
#ifndef CG_CI0  // staged channels per chunk, per member (k = 11, 7, 3)
#define CG_CI0 CG_CI
#endif
#ifndef CG_CI1
#define CG_CI1 CG_CI
#endif
#ifndef CG_CI2
#define CG_CI2 CG_CI
#endif
#ifndef RB_ONLY  // >= 0: every workgroup runs member RB_ONLY's conv (per-tap-count efficiency)
#define RB_ONLY -1
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
#ifndef CG_DIL
#define CG_DIL 1
#endif
#ifndef CG_LB   // second launch bound (waves per SIMD)
#define CG_LB 4
#endif
#ifndef RB_ORDER
#define RB_ORDER 0
#endif
#ifndef RB_PERSIST
#define RB_PERSIST 0
#endif
#ifndef RB_NEW  // 1: rb_group_kernel (rb_conv.h, continuous matrix stream) instead of the chunked tile
#define RB_NEW 0
#endif
#ifndef RB_PERSIST_WGS
#define RB_PERSIST_WGS 1024
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NTHREADS = 64 * CG_WM * CG_WN * CG_KS;
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int LDSF = cmax(cmax(conv_lds_floats<11, CG_CI0, CG_MB, CG_NB, CG_WN, CG_KS, 56, EPI_LINEAR, CG_WM>(), conv_lds_floats<7, CG_CI1, CG_MB, CG_NB, CG_WN, CG_KS, 76, EPI_LINEAR, CG_WM>()),
                          conv_lds_floats<3, CG_CI2, CG_MB, CG_NB, CG_WN, CG_KS, 16, EPI_LINEAR, CG_WM>());

#ifndef RB_LB
#define RB_LB 4
#endif
constexpr int LDSF_NEW = cmax(cmax(rb_lds_floats<RbCfg<11>::HALO, CG_NB>(), rb_lds_floats<RbCfg<7>::HALO, CG_NB>()), rb_lds_floats<RbCfg<3>::HALO, CG_NB>());
template <bool NEW>
__device__ __forceinline__ void run_tile(const ConvGroupArgs& g, int m, int l, float* xs) {
  int tx, ty;
  if (l >= g.gx[m] * g.gy[m]) return;
  xcd_tile_lin(l, g.gx[m], g.gy[m], tx, ty);
#if RB_ONLY >= 0
  m = RB_ONLY;
#endif
  if constexpr (NEW) {
    if (m == 0) rb_tile<11, RbCfg<11>::HALO, EPI_LINEAR, false, CG_NB>(g.c[0], tx, ty, 0, xs);
    else if (m == 1) rb_tile<7, RbCfg<7>::HALO, EPI_LINEAR, false, CG_NB>(g.c[1], tx, ty, 0, xs);
    else rb_tile<3, RbCfg<3>::HALO, EPI_LINEAR, false, CG_NB>(g.c[2], tx, ty, 0, xs);
    return;
  }
  if (m == 0) conv_tile<11, CG_CI0, CG_MB, CG_NB, CG_WN, CG_KS, 56, EPI_LINEAR, CG_WM>(g.c[0], tx, ty, 0, xs);
  else if (m == 1) conv_tile<7, CG_CI1, CG_MB, CG_NB, CG_WN, CG_KS, 76, EPI_LINEAR, CG_WM>(g.c[1], tx, ty, 0, xs);
  else conv_tile<3, CG_CI2, CG_MB, CG_NB, CG_WN, CG_KS, 16, EPI_LINEAR, CG_WM>(g.c[2], tx, ty, 0, xs);
}

// a launch that does (next to) nothing: what do kernel BOUNDARIES of other streams cost the big launches?
__global__ void tiny_kernel(float* p, int n) {
  extern __shared__ float dyn[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
  if (n < 0) dyn[threadIdx.x] = p[0];  // (keeps the dynamic LDS allocation alive)
}
// a small launch that also WORKS for a few microseconds (a dependent FMA chain per thread), like a GlowTTS conv
__global__ void small_work_kernel(float* p, int n, int iters) {
  extern __shared__ float dyn[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? p[i] : 0.f;
  for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
  if (i < n) p[i] = v;
  if (n < 0) dyn[threadIdx.x] = v;
}

// dispatch-order variants around the unchanged tile code
template <bool NEW>
__global__ __launch_bounds__(NEW ? 256 : NTHREADS, NEW ? RB_LB : CG_LB) void diag_group_kernel(const ConvGroupArgs g, int* ticket) {
  __shared__ float xs[NEW ? LDSF_NEW : LDSF];
#if RB_PERSIST
  __shared__ int s_t;
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1);
    __syncthreads();
    const int lin = s_t;
    __syncthreads();
    if (lin >= g.off[3]) return;
    const int m = lin < g.off[1] ? 0 : lin < g.off[2] ? 1 : 2;
    CONV_WG_STAMP(lin, 0);
    run_tile<NEW>(g, m, lin - g.off[m], xs);
    CONV_WG_STAMP(lin, 1);
    __syncthreads();
  }
#else
  const int lin = blockIdx.x;
  CONV_WG_STAMP(lin, 0);
#if RB_ORDER == 1
  // members interleaved in runs of 8 workgroups (a tile's XCD stays lin % 8): 11, 7, 3, 11, 7, 3, ... (equal tile counts)
  const int run = lin >> 3, m = run % 3;
  run_tile<NEW>(g, m, (run / 3) * 8 + (lin & 7), xs);
#else
  int m, l;
  if (g.nseg) {  // -DRB_ORDER=2: the product's snake order (group_snake_order, conv_mfma.h)
    int sg = 0;
    while (sg + 1 < g.nseg && lin >= g.seg_off[sg + 1]) ++sg;
    m = g.seg_m[sg];
    l = g.seg_first[sg] + (lin - g.seg_off[sg]);
  } else {
    m = lin < g.off[1] ? 0 : lin < g.off[2] ? 1 : 2;
    l = lin - g.off[m];
  }
  run_tile<NEW>(g, m, l, xs);
#endif
  CONV_WG_STAMP(lin, 1);
#endif
}

int main(int argc, char** argv) {
  const int C = CG_C;
  const int Ks[3] = {11, 7, 3};
  const int LMAX = CG_L * 8;
  // (a generator of our own: rand() is shared with the HIP runtime's threads, which made the data differ from run to run)
  unsigned long long lcg = 2;
  auto rnd = [&]() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((lcg >> 40) & 0xFFFFFF) / 16777216.0f; };
  std::vector<float> x((size_t)C * LMAX);
  // activations shaped like the pipeline's (post leaky-ReLU residual stream): zero-mean, a few tenths
  for (auto& v : x) v = (rnd() - 0.5f) * 0.6f;
  for (size_t i = 0; i < x.size(); i += 7) x[i] = -x[i] * 3.f;  // (both signs of the leaky-ReLU)
  float* dx;
  CK(hipMalloc(&dx, x.size() * 4));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  const int NSTREAM = 4;
  float* dy[NSTREAM][3];
  float *dw[3], *db[3];
  int noct[3];
  for (int m = 0; m < 3; ++m) {
    const int K = Ks[m];
    std::vector<float> w((size_t)C * C * K), b(C);
    for (auto& v : w) v = (rnd() - 0.5f) * 0.05f;
    for (auto& v : b) v = 0.01f;
    PackedConv p = pack_conv(C, CG_MB * CG_WM, C, K, [&](int v) { return v; }, [&](int co, int ci, int k) { return w[((size_t)co * C + ci) * K + k]; },
                             [&](int co) { return b[co]; }, true, 8);
    noct[m] = p.noct;
    CK(hipMalloc(&dw[m], p.w.size() * 4)); CK(hipMalloc(&db[m], p.bias.size() * 4));
    CK(hipMemcpy(dw[m], p.w.data(), p.w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db[m], p.bias.data(), p.bias.size() * 4, hipMemcpyHostToDevice));
    for (int s = 0; s < NSTREAM; ++s) CK(hipMalloc(&dy[s][m], (size_t)C * LMAX * 4 / (s == 0 ? 1 : 8)));
  }
  constexpr int T_T = CG_WN * CG_NB * 32;
  const int ytiles = C / (32 * CG_MB * CG_WM);
  auto make = [&](int L, int s, ConvGroupArgs& g, double& flop) {
    memset(&g, 0, sizeof(g));
    int off = 0;
    flop = 0;
    for (int m = 0; m < 3; ++m) {
      const int K = Ks[m];
      ConvArgs& a = g.c[m];
      a.x = dx; a.x_bs = (long long)C * L; a.x_ld = L; a.in_const = L; a.in_mul = 1;
      a.w = dw[m]; a.bias = db[m]; a.noct = noct[m]; a.Cin = C; a.rows = C; a.dil = CG_DIL; a.pad = CG_DIL * (K - 1) / 2; a.in_slope = 0.1f;
#ifdef CG_ABLATE  // with -DMI355TTS_ABLATION: conv_tile's ablation bits (1 = no staging after chunk 0, 2 = no A loads, 4 = no barrier)
      a.ablate = CG_ABLATE;
#endif
      a.y = dy[s][m]; a.y_bs = (long long)C * L; a.y_ld = L; a.split = 1 << 30; a.alpha = 1.f; a.out_const = L; a.out_mul = 1; a.res = dx;
      g.gx[m] = (L + T_T - 1) / T_T; g.gy[m] = ytiles; g.off[m] = off;
      off += (g.gx[m] * g.gy[m] + 7) & ~7;
      flop += 2.0 * C * C * (RB_ONLY >= 0 ? Ks[RB_ONLY] : K) * (double)L;
    }
    g.off[3] = off;
#if RB_ORDER == 2
    group_snake_order(g, 256, 1024);
#endif
  };
  int* d_ticket;
  CK(hipMalloc(&d_ticket, 4 * 64));
  hipStream_t st[NSTREAM];
  for (int s = 0; s < NSTREAM; ++s) CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
  auto launch = [&](const ConvGroupArgs& g, hipStream_t s, int slot) {
#if RB_PERSIST
    hipMemsetAsync(d_ticket + 16 * slot, 0, 4, s);
    hipLaunchKernelGGL(diag_group_kernel<RB_NEW != 0>, dim3(RB_PERSIST_WGS), dim3(RB_NEW ? 256 : NTHREADS), 0, s, g, d_ticket + 16 * slot);
#else
    hipLaunchKernelGGL(diag_group_kernel<RB_NEW != 0>, dim3(g.off[3]), dim3(RB_NEW ? 256 : NTHREADS), 0, s, g, d_ticket);
#endif
  };
#if RB_NEW
  {  // the new tile against the chunked one, bit for bit
    ConvGroupArgs g0, g1; double flop;
    make(CG_L, 0, g0, flop); make(CG_L, 1, g1, flop);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_group_kernel<11, 7, 3, CG_NB>), dim3(g0.off[3]), dim3(256), 0, st[0], g0);
    hipLaunchKernelGGL(diag_group_kernel<false>, dim3(g1.off[3]), dim3(NTHREADS), 0, st[0], g1, d_ticket);
    CK(hipStreamSynchronize(st[0]));
    std::vector<float> y0((size_t)C * CG_L), y1((size_t)C * CG_L);
    for (int m = 0; m < 3; ++m) {
      CK(hipMemcpy(y0.data(), dy[0][m], y0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(y1.data(), dy[1][m], y1.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; double mx = 0;
      for (size_t i = 0; i < y0.size(); ++i) { if (memcmp(&y0[i], &y1[i], 4)) ++bad; mx = fmax(mx, fabs((double)y0[i] - y1[i])); }
      printf("member k=%d: new vs chunked tile: %zu of %zu words differ, max |diff| %.3g, y[123] = %g\n", Ks[m], bad, y0.size(), mx, y0[123]);
    }
  }
#endif
  printf("== rb_diag C=%d L=%d dil=%d only=%d tile<CI=%d/%d/%d MB=%d NB=%d WN=%d KS=%d WM=%d LB=%d> order=%d persist=%d(%d) threads=%d lds=%d B\n", C, CG_L, CG_DIL, RB_ONLY, CG_CI0, CG_CI1, CG_CI2, CG_MB,
         CG_NB, CG_WN, CG_KS, CG_WM, CG_LB, RB_ORDER, RB_PERSIST, RB_PERSIST_WGS, NTHREADS, LDSF * 4);
  {  // clock warm-up: the first ~10 ms after idle run at a low shader clock
    ConvGroupArgs g; double flop;
    make(CG_L, 0, g, flop);
    for (int i = 0; i < 100; ++i) launch(g, st[0], 0);
    CK(hipStreamSynchronize(st[0]));
  }
  if (argc > 2 && !strcmp(argv[2], "pmc")) {  // counter runs: 20 launches of the L x4 problem, nothing else
    ConvGroupArgs g; double flop;
    make(CG_L * 4, 0, g, flop);
    for (int i = 0; i < 20; ++i) launch(g, st[0], 0);
    CK(hipStreamSynchronize(st[0]));
    return 0;
  }
  // ---- 1. back-to-back on one stream at L, 2L, 4L, 8L
  for (int mul = 1; mul <= 8; mul *= 2) {
    ConvGroupArgs g; double flop;
    make(CG_L * mul, 0, g, flop);
    for (int i = 0; i < 3; ++i) launch(g, st[0], 0);
    CK(hipStreamSynchronize(st[0]));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = mul == 1 ? 30 : 10;
    CK(hipEventRecord(e0, st[0]));
    for (int i = 0; i < N; ++i) launch(g, st[0], 0);
    CK(hipEventRecord(e1, st[0])); CK(hipStreamSynchronize(st[0]));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / N;
    printf("L x%d wgs=%d: %.1f us/launch  %.1f TFLOP/s (%.3f of 157.3)\n", mul, g.off[3], us, flop / us / 1e6, flop / us / 1e6 / 157.3);
  }
  // ---- 1b. pipeline conditions: every member on its OWN planes (as the three MRF chains are), six launches chained
  // x -> t -> x' ... (the input of a launch is what the previous launch wrote), a different weight set per launch
  {
    const int NW = 6, L = CG_L;
    std::vector<ConvGroupArgs> gs(NW);
    float* pl[3][2];
    for (int m = 0; m < 3; ++m)
      for (int q = 0; q < 2; ++q) { CK(hipMalloc(&pl[m][q], (size_t)C * L * 4)); CK(hipMemcpy(pl[m][q], dx, (size_t)C * L * 4, hipMemcpyDeviceToDevice)); }
    double flop = 0;
    for (int i = 0; i < NW; ++i) {
      make(L, 0, gs[i], flop);
      for (int m = 0; m < 3; ++m) {
        ConvArgs& a = gs[i].c[m];
        float* dwi; const size_t wn = (size_t)(C / 32) * noct[m] * Ks[m] * 256;
        CK(hipMalloc(&dwi, wn * 4)); CK(hipMemcpy(dwi, dw[m], wn * 4, hipMemcpyDeviceToDevice));
        a.w = dwi; a.x = pl[m][i & 1]; a.y = pl[m][(i & 1) ^ 1]; a.res = (i & 1) ? pl[m][(i & 1) ^ 1] : nullptr;
        if (i & 1) a.res = nullptr;  // (in place: conv2 adds x = the plane it overwrites; timing only — keep the values bounded instead)
        a.bias = db[m];
      }
    }
    for (int r = 0; r < 3; ++r) for (int i = 0; i < NW; ++i) launch(gs[i], st[0], 0);
    CK(hipStreamSynchronize(st[0]));
    for (int m = 0; m < 3; ++m)
      for (int q = 0; q < 2; ++q) CK(hipMemcpy(pl[m][q], dx, (size_t)C * L * 4, hipMemcpyDeviceToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2;
    CK(hipEventRecord(e0, st[0]));
    for (int r = 0; r < N; ++r) for (int i = 0; i < NW; ++i) launch(gs[i], st[0], 0);
    CK(hipEventRecord(e1, st[0])); CK(hipStreamSynchronize(st[0]));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / (N * NW);
    printf("chained, own planes, rotating weights: %.1f us/launch  %.1f TFLOP/s (%.3f of 157.3)\n", us, flop / us / 1e6, flop / us / 1e6 / 157.3);
  }
  // ---- 2. S streams, each launching the L-column problem N times
  for (int S = 1; S <= NSTREAM; S *= 2) {
    ConvGroupArgs g[NSTREAM]; double flop = 0;
    for (int s = 0; s < S; ++s) make(CG_L, s, g[s], flop);
    for (int s = 0; s < S; ++s) launch(g[s], st[s], s);
    CK(hipDeviceSynchronize());
    const int N = 30;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i)
      for (int s = 0; s < S; ++s) launch(g[s], st[s], s);
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (N * S);
    printf("%d stream(s): %.1f us per launch (host clock)  %.1f TFLOP/s (%.3f)\n", S, us, flop / us / 1e6, flop / us / 1e6 / 157.3);
  }
  // ---- 2b. four streams of the big launch + ONE stream of dependent small launches of a given per-workgroup footprint
  // (threads, LDS bytes): how long does a small launch take next to the big ones?  (240 workgroups each, like a GlowTTS conv)
  {
    ConvGroupArgs g[NSTREAM]; double flop = 0;
    for (int s = 0; s < NSTREAM; ++s) make(CG_L, s, g[s], flop);
    hipStream_t es; float* ep;
    CK(hipStreamCreateWithFlags(&es, hipStreamNonBlocking)); CK(hipMalloc(&ep, 1 << 20)); CK(hipMemset(ep, 0, 1 << 20));
    const int fp[][2] = {{64, 0}, {256, 16 << 10}, {256, 30 << 10}, {512, 30 << 10}, {256, 60 << 10}, {512, 60 << 10}, {512, 96 << 10}};
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
    for (auto& f : fp) {
      CK(hipDeviceSynchronize());
      const int N = 12, SM = 100;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int s = 0; s < NSTREAM; ++s) for (int i = 0; i < 3; ++i) launch(g[s], st[s], s);  // the load is up before the small stream starts
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i)
        for (int s = 0; s < NSTREAM; ++s) launch(g[s], st[s], s);
      CK(hipEventRecord(e0, es));
      for (int k = 0; k < SM; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(240), dim3(f[0]), f[1], es, ep, 240 * f[0]);
      CK(hipEventRecord(e1, es));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      for (int s = 0; s < NSTREAM; ++s) CK(hipStreamSynchronize(st[s]));
      const double us_big = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (N * NSTREAM);
      printf("small launches of 240 x (%d threads, %d KB LDS) next to 4 big streams: %.1f us per small launch; big launches %.1f us each\n", f[0], f[1] >> 10,
             1e3 * ms / SM, us_big);
    }
  }
  // ---- 2c. four big streams + K streams of dependent small WORKING launches (240 x 512 threads, 30 KB LDS, ~5 us alone):
  // does the aggregate rate of small launches grow with K, or is it capped?
  {
    ConvGroupArgs g[NSTREAM]; double flop = 0;
    for (int s = 0; s < NSTREAM; ++s) make(CG_L, s, g[s], flop);
    const int KMAX = 8;
    hipStream_t es[KMAX]; float* ep[KMAX];
    for (int k = 0; k < KMAX; ++k) { CK(hipStreamCreateWithFlags(&es[k], hipStreamNonBlocking)); CK(hipMalloc(&ep[k], 1 << 20)); CK(hipMemset(ep[k], 0, 1 << 20)); }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(small_work_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
    {  // alone
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, es[0]));
      for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(small_work_kernel, dim3(240), dim3(512), 30 << 10, es[0], ep[0], 240 * 512, 600);
      CK(hipEventRecord(e1, es[0])); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("small working launch alone: %.1f us each\n", 1e3 * ms / 200);
    }
    for (int K : {1, 2, 4, 8}) {
      CK(hipDeviceSynchronize());
      const int N = 10, SM = 60;
      for (int s = 0; s < NSTREAM; ++s) for (int i = 0; i < 3; ++i) launch(g[s], st[s], s);
      hipEvent_t e0[KMAX], e1[KMAX];
      for (int k = 0; k < K; ++k) { CK(hipEventCreate(&e0[k])); CK(hipEventCreate(&e1[k])); }
      for (int i = 0; i < N; ++i)
        for (int s = 0; s < NSTREAM; ++s) launch(g[s], st[s], s);
      for (int k = 0; k < K; ++k) CK(hipEventRecord(e0[k], es[k]));
      for (int j = 0; j < SM; ++j)
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(small_work_kernel, dim3(240), dim3(512), 30 << 10, es[k], ep[k], 240 * 512, 600);
      for (int k = 0; k < K; ++k) CK(hipEventRecord(e1[k], es[k]));
      double tot = 0;
      for (int k = 0; k < K; ++k) { CK(hipEventSynchronize(e1[k])); float ms; CK(hipEventElapsedTime(&ms, e0[k], e1[k])); tot += ms; }
      CK(hipDeviceSynchronize());
      printf("%d chain(s) of small working launches next to 4 big streams: %.1f us per launch per chain -> aggregate %.0f launches/ms\n", K, 1e3 * tot / K / SM,
             K * SM / (tot / K));
    }
  }
  // ---- 3. timeline of one launch in steady state (the 3rd of 4 back-to-back launches)
  {
    ConvGroupArgs g; double flop;
    make(CG_L, 0, g, flop);
    const int nwg = g.off[3];
    WgRec* d_rec; long long* d_chunk; long long* d_phase;
    CK(hipMalloc(&d_phase, 32 * (size_t)nwg)); CK(hipMemset(d_phase, 0, 32 * (size_t)nwg));
    const size_t chunk_n = ((size_t)(nwg / RB_CHUNK_EVERY + 1)) * 16 * 16 * 4;
    CK(hipMalloc(&d_rec, sizeof(WgRec) * nwg)); CK(hipMemset(d_rec, 0, sizeof(WgRec) * nwg));
    CK(hipMalloc(&d_chunk, 8 * chunk_n)); CK(hipMemset(d_chunk, 0, 8 * chunk_n));
    WgRec* null_rec = nullptr;
    for (int i = 0; i < 4; ++i) {
      if (i == 2) { CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(rb_phase_rec), &d_phase, 8, 0, hipMemcpyHostToDevice, st[0])); CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(rb_wg_rec), &d_rec, 8, 0, hipMemcpyHostToDevice, st[0])); CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(rb_chunk_rec), &d_chunk, 8, 0, hipMemcpyHostToDevice, st[0])); }
      if (i == 3) { CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(rb_phase_rec), &null_rec, 8, 0, hipMemcpyHostToDevice, st[0])); CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(rb_wg_rec), &null_rec, 8, 0, hipMemcpyHostToDevice, st[0])); CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(rb_chunk_rec), &null_rec, 8, 0, hipMemcpyHostToDevice, st[0])); }
      launch(g, st[0], 0);
    }
    CK(hipStreamSynchronize(st[0]));
    std::vector<WgRec> rec(nwg);
    std::vector<long long> ch(chunk_n);
    CK(hipMemcpy(rec.data(), d_rec, sizeof(WgRec) * nwg, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ch.data(), d_chunk, 8 * chunk_n, hipMemcpyDeviceToHost));
    long long t0 = 1LL << 62, t1 = 0;
    double clk = 0; int nclk = 0;
    for (auto& r : rec) if (r.real1) { t0 = r.real0 < t0 ? r.real0 : t0; t1 = r.real1 > t1 ? r.real1 : t1; if (r.real1 - r.real0 > 500) { clk += (double)(r.clk1 - r.clk0) / (r.real1 - r.real0); ++nclk; } }
#if RB_NEW
    {  // phases of the new tile per member: prologue, main loop, epilogue (shader cycles, medians)
      std::vector<long long> ph((size_t)nwg * 4);
      CK(hipMemcpy(ph.data(), d_phase, 32 * (size_t)nwg, hipMemcpyDeviceToHost));
      for (int m = 0; m < 3; ++m) {
        std::vector<long long> a, b, c;
        for (int i = g.off[m]; i < g.off[m + 1]; ++i) {
          const long long* q = &ph[(size_t)i * 4];
          if (q[0] && q[3]) { a.push_back(q[1] - q[0]); b.push_back(q[2] - q[1]); c.push_back(q[3] - q[2]); }
        }
        if (a.empty()) continue;
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
        printf("member %d phases (cycles, median [p10 p90]): prologue %lld [%lld %lld]  main loop %lld [%lld %lld]  epilogue %lld [%lld %lld]\n", m, a[a.size() / 2], a[a.size() / 10],
               a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() / 10], b[b.size() * 9 / 10], c[c.size() / 2], c[c.size() / 10], c[c.size() * 9 / 10]);
      }
    }
#endif
    printf("timeline: span %.1f us (s_memrealtime, 100 MHz), shader clock %.3f GHz (s_memtime / s_memrealtime over %d workgroups)\n", (t1 - t0) / 100.0, nclk ? clk / nclk / 10.0 : 0, nclk);
    if (argc > 1) {
      FILE* f = fopen(argv[1], "w");
      if (f) {
        fprintf(f, "# lin member real0 real1 clk0 clk1 hwid xcc   (real in 10 ns ticks from the first begin)\n");
        for (int i = 0; i < nwg; ++i) {
          const WgRec& r = rec[i];
          if (!r.real1) continue;
          int m = i < g.off[1] ? 0 : i < g.off[2] ? 1 : 2;
          if (g.nseg) { int sg = 0; while (sg + 1 < g.nseg && i >= g.seg_off[sg + 1]) ++sg; m = g.seg_m[sg]; }
#if RB_ORDER == 1
          m = (i >> 3) % 3;
#endif
          fprintf(f, "%d %d %lld %lld %lld %lld %u %u\n", i, m, r.real0 - t0, r.real1 - t0, r.clk0, r.clk1, r.hwid, r.xcc);
        }
#ifdef RB_CHUNK_STAMPS
        fprintf(f, "# chunk stamps: sample wave chunk s0 s1 s2 s3 (s_memtime)\n");
        for (size_t s = 0; s < chunk_n / (16 * 16 * 4); ++s)
          for (int w = 0; w < 16; ++w)
            for (int c = 0; c < 16; ++c) {
              const long long* p = &ch[((s * 16 + w) * 16 + c) * 4];
              if (p[0]) fprintf(f, "C %zu %d %d %lld %lld %lld %lld\n", s, w, c, p[0], p[1], p[2], p[3]);
            }
#endif
        fclose(f);
      }
    }
  }
  return 0;
}
