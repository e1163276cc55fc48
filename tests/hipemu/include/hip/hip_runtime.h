// TEST INFRASTRUCTURE — a CPU stand-in for <hip/hip_runtime.h>.
//
// `tests/hipemu` compiles the UNMODIFIED product sources under
// larynx_amd/csrc/ with the host clang++ (`-x c++ -Itests/hipemu/include`), so
// that the kernels' index arithmetic, masking, LDS staging and MFMA fragment
// bookkeeping can be exercised in the CPU-only CI (`pytest -m "not gpu"`),
// where no GPU exists.  Every workgroup thread is a fiber; `__syncthreads()`,
// wave shuffles and the f32 MFMA builtins are rendezvous points between
// fibers with the gfx950 lane->element maps
// (/opt/skills/guides/cdna_hip_programming.md §3).
//
// This is NOT a fallback: the product (`larynx_amd/ffi.py`) only ever loads the
// hipcc-built `libmi355tts.so`; nothing outside `tests/` references this
// directory, and nothing here is timed or shipped.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// ---------------------------------------------------------------- host types
typedef int hipError_t;
enum {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNotReady = 600,
  hipErrorUnknown = 999
};
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum hipMemcpyKind {
  hipMemcpyHostToHost = 0,
  hipMemcpyHostToDevice = 1,
  hipMemcpyDeviceToHost = 2,
  hipMemcpyDeviceToDevice = 3,
  hipMemcpyDefault = 4
};
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000, hipEventDisableTiming = 2, hipEventBlockingSync = 1 };

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  int multiProcessorCount;
  size_t totalGlobalMem;
  int clockRate;
};

// ------------------------------------------------------------- kernel language
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_KERNEL_NAME(...) __VA_ARGS__

namespace hipemu {

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  dim3 tid;
  int linear = 0;
  int wave = 0;
  int lane = 0;
  bool done = false;
  unsigned block_gen_seen = 0;
  unsigned wave_gen_seen = 0;
};

struct WaveState {
  int waiting = 0;
  int live = 0;
  unsigned gen = 0;
  // exchange buffers for cross-lane ops
  uint32_t xchg[64];
  float A[32 * 2];
  float B[2 * 32];
  float A16[16 * 4];
  float B16[4 * 16];
  float Ab[32 * 16];
  float Bb[16 * 32];
};

struct BlockCtx {
  dim3 bid, bdim, gdim;
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  int waiting = 0;
  int live = 0;
  unsigned gen = 0;
  void* sched_sp = nullptr;
  const std::function<void()>* body = nullptr;
  Fiber* cur = nullptr;
  bool progress = false;
};

BlockCtx*& ctx();
void yield_to_scheduler();
void block_barrier();
void wave_barrier();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
[[noreturn]] void die(const char* msg);

inline Fiber* cur() { return ctx()->cur; }

template <typename T>
inline T shfl_generic(T v, int src_lane) {
  static_assert(sizeof(T) == 4, "shuffle emulation supports 32-bit types");
  BlockCtx* c = ctx();
  Fiber* f = c->cur;
  WaveState& w = c->waves[f->wave];
  uint32_t bits;
  std::memcpy(&bits, &v, 4);
  w.xchg[f->lane] = bits;
  wave_barrier();
  uint32_t r = w.xchg[src_lane & 63];
  wave_barrier();
  T out;
  std::memcpy(&out, &r, 4);
  return out;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5); k-ordered fmaf chain.
inline f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
  BlockCtx* cx = ctx();
  Fiber* f = cx->cur;
  WaveState& w = cx->waves[f->wave];
  const int l = f->lane;
  w.A[(l & 31) * 2 + (l >> 5)] = a;
  w.B[(l >> 5) * 32 + (l & 31)] = b;
  wave_barrier();
  const int col = l & 31;
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    float acc = c[reg];
    acc = std::fmaf(w.A[row * 2 + 0], w.B[0 * 32 + col], acc);
    acc = std::fmaf(w.A[row * 2 + 1], w.B[1 * 32 + col], acc);
    c[reg] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D col = l&15, row = (l>>4)*4 + reg.
inline f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  BlockCtx* cx = ctx();
  Fiber* f = cx->cur;
  WaveState& w = cx->waves[f->wave];
  const int l = f->lane;
  w.A16[(l & 15) * 4 + (l >> 4)] = a;
  w.B16[(l >> 4) * 16 + (l & 15)] = b;
  wave_barrier();
  const int col = l & 15;
  for (int reg = 0; reg < 4; ++reg) {
    const int row = (l >> 4) * 4 + reg;
    float acc = c[reg];
    for (int k = 0; k < 4; ++k) acc = std::fmaf(w.A16[row * 4 + k], w.B16[k * 16 + col], acc);
    c[reg] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 x 4 x 1 — block b takes A[i] from lane 4b + i and B[j] from lane 4b + j and
// holds D_b[i][j] in register i of lane 4b + j.  CBSZ = n broadcasts A inside sets of 2^n blocks: every block of a set reads
// block (set base + ABID)'s A.  (CBSZ = 4, ABID = k: all 16 blocks use the A of lanes 4k .. 4k + 3 — a 4-row x 64-column
// rank-1 update, how composable_kernel drives it.)
inline f32x4 mfma_f32_4x4x1(float a, float b, f32x4 c, int cbsz, int abid) {
  BlockCtx* cx = ctx();
  Fiber* f = cx->cur;
  WaveState& w = cx->waves[f->wave];
  const int l = f->lane;
  w.A[l] = a;
  wave_barrier();
  const int blk = l >> 2;
  const int set = cbsz > 0 ? (blk >> cbsz) << cbsz : blk;
  const int src = cbsz > 0 ? set + (abid & ((1 << cbsz) - 1)) : blk;
  for (int i = 0; i < 4; ++i) c[i] = std::fmaf(w.A[4 * src + i], b, c[i]);
  wave_barrier();
  return c;
}

// v_mfma_f32_32x32x16_bf16: lane l supplies 8 consecutive k of A row l&31 and of B column l&31,
// k = 8*(l>>5) .. +7; D as for the 32x32 f32 MFMA.  Products of bf16 values are exact in f32;
// accumulated here in k order (the hardware's internal order is not specified — tests allow for it).
typedef __bf16 bf16x8_emu __attribute__((ext_vector_type(8)));
inline f32x16 mfma_f32_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16 c) {
  BlockCtx* cx = ctx();
  Fiber* f = cx->cur;
  WaveState& w = cx->waves[f->wave];
  const int l = f->lane;
  for (int i = 0; i < 8; ++i) {
    w.Ab[(l & 31) * 16 + 8 * (l >> 5) + i] = (float)a[i];
    w.Bb[(8 * (l >> 5) + i) * 32 + (l & 31)] = (float)b[i];
  }
  wave_barrier();
  const int col = l & 31;
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    float acc = c[reg];
    for (int k = 0; k < 16; ++k) acc = std::fmaf(w.Ab[row * 16 + k], w.Bb[k * 32 + col], acc);
    c[reg] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_32x32x16_f16: the same lane maps with IEEE half operands (products of two halves are exact in f32)
typedef _Float16 f16x8_emu __attribute__((ext_vector_type(8)));
inline f32x16 mfma_f32_32x32x16_f16(f16x8_emu a, f16x8_emu b, f32x16 c) {
  BlockCtx* cx = ctx();
  Fiber* f = cx->cur;
  WaveState& w = cx->waves[f->wave];
  const int l = f->lane;
  for (int i = 0; i < 8; ++i) {
    w.Ab[(l & 31) * 16 + 8 * (l >> 5) + i] = (float)a[i];
    w.Bb[(8 * (l >> 5) + i) * 32 + (l & 31)] = (float)b[i];
  }
  wave_barrier();
  const int col = l & 31;
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
    float acc = c[reg];
    for (int k = 0; k < 16; ++k) acc = std::fmaf(w.Ab[row * 16 + k], w.Bb[k * 32 + col], acc);
    c[reg] = acc;
  }
  wave_barrier();
  return c;
}

// v_permlane32_swap_b32 (gfx950): lanes 32 .. 63 of `old` change places with lanes 0 .. 31 of `src`; returns {old', src'}
struct u32x2_emu {
  uint32_t v[2];
  uint32_t operator[](int i) const { return v[i]; }
};
u32x2_emu permlane32_swap(uint32_t old_v, uint32_t src_v);

}  // namespace hipemu

#define threadIdx (hipemu::cur()->tid)
#define blockIdx (hipemu::ctx()->bid)
#define blockDim (hipemu::ctx()->bdim)
#define gridDim (hipemu::ctx()->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::block_barrier(); }
template <typename T> static inline T __shfl(T v, int lane, int width = 64) {
  int base = (hipemu::cur()->lane / width) * width;
  return hipemu::shfl_generic(v, base + (lane % width));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return hipemu::shfl_generic(v, hipemu::cur()->lane ^ mask);
}
template <typename T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int l = hipemu::cur()->lane;
  int src = ((l % width) + (int)delta < width) ? l + (int)delta : l;
  return hipemu::shfl_generic(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  int l = hipemu::cur()->lane;
  int src = ((l % width) >= (int)delta) ? l - (int)delta : l;
  return hipemu::shfl_generic(v, src);
}

#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_f32_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_f32_16x16x4((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, cbsz, abid, blgp) hipemu::mfma_f32_4x4x1((a), (b), (c), (cbsz), (abid))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_f32_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu::mfma_f32_32x32x16_f16((a), (b), (c))
namespace hipemu {
inline u32x2_emu permlane32_swap(uint32_t old_v, uint32_t src_v) {
  const int l = cur()->lane;
  const uint32_t old_o = shfl_generic(old_v, l ^ 32), src_o = shfl_generic(src_v, l ^ 32);
  u32x2_emu r;
  r.v[0] = l < 32 ? old_v : src_o;  // upper lanes of old' = lower lanes of src
  r.v[1] = l < 32 ? old_o : src_v;  // lower lanes of src' = upper lanes of old
  return r;
}
}  // namespace hipemu
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu::permlane32_swap((a), (b))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// lanes are fibers here: a wave's lock step (an LDS write another lane of the SAME wave then reads) needs a real rendezvous
#define __builtin_amdgcn_wave_barrier() hipemu::wave_barrier()
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
// only ever applied to wave-uniform values in csrc/ (it tells the compiler they ARE uniform)
#define __builtin_amdgcn_readfirstlane(x) (x)
// launches run to completion inside the launch call here: a kernel that waits for the host (mi355tts.hip, queue_wait_kernel) must
// fall through at once — the clock leaps past any bound
static inline long long wall_clock64() { static std::atomic<long long> t{0}; return t.fetch_add(1LL << 40); }
#define __HIP_MEMORY_SCOPE_SYSTEM 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float cospif(float x) { return (float)std::cos(3.14159265358979323846 * (double)x); }
static inline float sinpif(float x) { return (float)std::sin(3.14159265358979323846 * (double)x); }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float exp10f_emu(float x) { return std::pow(10.0f, x); }
static inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; std::memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(unsigned v) { float f; std::memcpy(&f, &v, 4); return f; }

static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  float f;
  do {
    std::memcpy(&f, &old, 4);
    f += v;
    std::memcpy(&nw, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  std::memcpy(&f, &old, 4);
  return f;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (shmem), [&]() { kernel(__VA_ARGS__); })

// ------------------------------------------------------------------ runtime API
hipError_t hipMalloc(void** p, size_t n);
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
// test hook: verify the NaN guard zones around every live allocation
extern "C" int hipemu_check_guards();
