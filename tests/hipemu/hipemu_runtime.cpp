// TEST INFRASTRUCTURE — runtime of the CPU HIP emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <chrono>
#include <map>
#include <sys/mman.h>

namespace hipemu {

// ---------------------------------------------------------------- context switch
// Minimal x86-64 SysV stack switch: saves callee-saved registers on the current
// stack, stores rsp to *save, loads rsp from `load`, restores and returns.
extern "C" void hipemu_switch(void** save, void* load);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

static thread_local BlockCtx* g_ctx = nullptr;
BlockCtx*& ctx() { return g_ctx; }

[[noreturn]] void die(const char* msg) {
  std::fprintf(stderr, "hipemu: fatal: %s\n", msg);
  std::abort();
}

void yield_to_scheduler() {
  BlockCtx* c = g_ctx;
  Fiber* f = c->cur;
  hipemu_switch(&f->sp, c->sched_sp);
}

void block_barrier() {
  BlockCtx* c = g_ctx;
  Fiber* f = c->cur;
  const unsigned gen = c->gen;
  c->waiting++;
  while (c->gen == gen) yield_to_scheduler();
  (void)f;
}

void wave_barrier() {
  BlockCtx* c = g_ctx;
  Fiber* f = c->cur;
  WaveState& w = c->waves[f->wave];
  const unsigned gen = w.gen;
  w.waiting++;
  while (w.gen == gen) yield_to_scheduler();
}

static void fiber_entry() {
  BlockCtx* c = g_ctx;
  Fiber* f = c->cur;
  (*c->body)();
  f->done = true;
  c->live--;
  c->waves[f->wave].live--;
  c->progress = true;
  hipemu_switch(&f->sp, c->sched_sp);
  die("resumed a finished fiber");
}

static constexpr size_t kStackBytes = 256 * 1024;

static void run_block(BlockCtx& c, char* stacks) {
  const int n = (int)c.fibers.size();
  g_ctx = &c;
  for (int i = 0; i < n; ++i) {
    Fiber& f = c.fibers[i];
    f.stack = stacks + (size_t)i * kStackBytes;
    f.done = false;
    uintptr_t top = (uintptr_t)(f.stack + kStackBytes);
    top &= ~(uintptr_t)15;
    void** slot = (void**)(top - 16);  // return address slot, 16-byte aligned
    *slot = (void*)&fiber_entry;
    void** sp = slot - 6;  // six callee-saved registers
    for (int r = 0; r < 6; ++r) sp[r] = nullptr;
    f.sp = (void*)sp;
  }
  c.live = n;
  c.waiting = 0;
  c.gen = 0;
  for (auto& w : c.waves) {
    w.waiting = 0;
    w.gen = 0;
    w.live = 0;
  }
  for (auto& f : c.fibers) c.waves[f.wave].live++;

  while (c.live > 0) {
    c.progress = false;
    for (int i = 0; i < n; ++i) {
      Fiber& f = c.fibers[i];
      if (f.done) continue;
      c.cur = &f;
      const int w0 = c.waiting;
      const int ww0 = c.waves[f.wave].waiting;
      hipemu_switch(&c.sched_sp, f.sp);
      if (c.waiting != w0 || c.waves[f.wave].waiting != ww0) c.progress = true;
      // release barriers whose every live participant has arrived
      WaveState& w = c.waves[f.wave];
      if (w.waiting > 0 && w.waiting >= w.live) {
        w.waiting = 0;
        w.gen++;
        c.progress = true;
      }
      if (c.waiting > 0 && c.waiting >= c.live) {
        c.waiting = 0;
        c.gen++;
        c.progress = true;
      }
    }
    if (!c.progress && c.live > 0) die("deadlock: divergent __syncthreads()/wave op");
  }
  g_ctx = nullptr;
}

static int worker_count() {
  const char* e = std::getenv("HIPEMU_THREADS");
  int n = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(n, 64));
}

void launch(dim3 grid, dim3 block, size_t /*shmem*/, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nblocks == 0 || nthreads == 0) return;
  if (nthreads > 1024) die("block too large");
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    char* stacks = (char*)mmap(nullptr, kStackBytes * (size_t)nthreads, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char*)MAP_FAILED) die("mmap fiber stacks");
    BlockCtx c;
    c.bdim = block;
    c.gdim = grid;
    c.body = &body;
    c.fibers.resize(nthreads);
    c.waves.resize((nthreads + 63) / 64);
    for (int i = 0; i < nthreads; ++i) {
      Fiber& f = c.fibers[i];
      f.linear = i;
      f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
      f.wave = i / 64;
      f.lane = i % 64;
    }
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      c.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
      run_block(c, stacks);
    }
    munmap(stacks, kStackBytes * (size_t)nthreads);
  };
  int nw = (int)std::min<size_t>(worker_count(), nblocks);
  if (nw <= 1) {
    worker();
  } else {
    std::vector<std::thread> ts;
    for (int i = 0; i < nw; ++i) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
  }
}

}  // namespace hipemu

// ------------------------------------------------------------------ runtime API
namespace {
constexpr size_t kGuard = 256;  // bytes of NaN canary on each side
struct Alloc { size_t n; };
std::mutex g_mu;
std::map<void*, Alloc> g_allocs;
const uint32_t kCanary = 0x7fc0beefu;  // a quiet NaN: OOB reads poison results

void fill_guard(char* p) {
  for (size_t i = 0; i < kGuard; i += 4) std::memcpy(p + i, &kCanary, 4);
}
bool guard_ok(const char* p) {
  for (size_t i = 0; i < kGuard; i += 4) {
    uint32_t v;
    std::memcpy(&v, p + i, 4);
    if (v != kCanary) return false;
  }
  return true;
}
}  // namespace

hipError_t hipMalloc(void** p, size_t n) {
  size_t padded = (n + 15) & ~(size_t)15;
  char* raw = (char*)std::malloc(padded + 2 * kGuard);
  if (!raw) return hipErrorOutOfMemory;
  fill_guard(raw);
  // fresh device memory is garbage: make accidental reads of it visible
  for (size_t i = 0; i < padded; i += 4) std::memcpy(raw + kGuard + i, &kCanary, 4);
  fill_guard(raw + kGuard + padded);
  *p = raw + kGuard;
  std::lock_guard<std::mutex> lk(g_mu);
  g_allocs[*p] = Alloc{padded};
  return hipSuccess;
}

extern "C" int hipemu_check_guards() {
  std::lock_guard<std::mutex> lk(g_mu);
  int bad = 0;
  for (auto& kv : g_allocs) {
    const char* base = (const char*)kv.first;
    if (!guard_ok(base - kGuard) || !guard_ok(base + kv.second.n)) {
      std::fprintf(stderr, "hipemu: out-of-bounds WRITE around allocation %p (%zu bytes)\n", kv.first, kv.second.n);
      bad++;
    }
  }
  return bad;
}

hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.find(p);
  if (it == g_allocs.end()) return hipErrorInvalidValue;
  const char* base = (const char*)p;
  if (!guard_ok(base - kGuard) || !guard_ok(base + it->second.n)) {
    std::fprintf(stderr, "hipemu: out-of-bounds WRITE detected at hipFree(%p)\n", p);
    std::abort();
  }
  std::free((char*)p - kGuard);
  g_allocs.erase(it);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
struct hipemuStream { int id; };
struct hipemuEvent { std::chrono::steady_clock::time_point t; };
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipemuStream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
  if (least) *least = 0;
  if (greatest) *greatest = -1;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipemu_check_guards() ? hipErrorUnknown : hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipemu_check_guards() ? hipErrorUnknown : hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }  // kernels run synchronously at launch
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  std::snprintf(p->name, sizeof(p->name), "hipemu (CPU test emulator)");
  std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "hipemu");
  p->multiProcessorCount = 256;
  p->clockRate = 2400000;
  return hipSuccess;
}
