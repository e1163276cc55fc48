// What a dependent launch costs on this stack: back-to-back launches on one stream of (a) an empty kernel,
// (b) one workgroup doing one dependent load -> store, (c) 120 workgroups x 512 threads doing the same,
// (d) as (c) with 64 KB of LDS declared.  Build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_touch(const float* x, float* y) { y[blockIdx.x * blockDim.x + threadIdx.x] = x[blockIdx.x * blockDim.x + threadIdx.x] + 1.f; }
__global__ void k_touch_lds(const float* x, float* y) {
  __shared__ float s[16384];
  s[threadIdx.x] = x[blockIdx.x * blockDim.x + threadIdx.x];
  __syncthreads();
  y[blockIdx.x * blockDim.x + threadIdx.x] = s[threadIdx.x ^ 1] + 1.f;
}
template <class F>
static float time_us(F launch, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / n;
}
int main() {
  float *x, *y;
  CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 1 << 20));
  CK(hipMemset(x, 0, 1 << 20));
  const int N = 2000;
  printf("empty kernel, 1 workgroup          : %.2f us per launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, N));
  printf("empty kernel, 120 x 512            : %.2f us per launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(120), dim3(512), 0, 0); }, N));
  printf("load -> store, 1 x 64              : %.2f us per launch\n", time_us([&] { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0, x, y); }, N));
  printf("load -> store, 120 x 512           : %.2f us per launch\n", time_us([&] { hipLaunchKernelGGL(k_touch, dim3(120), dim3(512), 0, 0, x, y); }, N));
  printf("load -> LDS -> store, 120 x 512    : %.2f us per launch\n", time_us([&] { hipLaunchKernelGGL(k_touch_lds, dim3(120), dim3(512), 0, 0, x, y); }, N));
  // ping-pong x -> y -> x: a true dependent chain (each launch reads what the previous one wrote)
  int flip = 0;
  printf("dependent chain (x->y->x), 120x512 : %.2f us per launch\n",
         time_us([&] { if (flip ^= 1) hipLaunchKernelGGL(k_touch, dim3(120), dim3(512), 0, 0, x, y); else hipLaunchKernelGGL(k_touch, dim3(120), dim3(512), 0, 0, y, x); }, N));
  return 0;
}
