// mi355tts host runtime — context, per-call workers (stream + workspace + pinned staging), device-block pool
// (one translation unit: included once by mi355tts.hip, after the kernel headers)
#pragma once

// ------------------------------------------------------------------ context
struct ProfEvent {
  hipEvent_t a, b;
  int cls;
  double flop;
  int kn, sub;  // kernel name (KName, -1 = a kernel without one) and a sub-key of the launch (its output rows / channels)
};
enum KClass { KC_RESBLOCK = 0, KC_UPSAMPLE, KC_VOC_IO, KC_GLOW_ENC_CONV, KC_GLOW_DEC_CONV, KC_SMALL, KC_MRF_NARROW, KC_COUNT };
static const char* kclass_name[KC_COUNT] = {"conv_mfma.hifigan_resblock", "conv_mfma.hifigan_upsample",
                                            "conv_mfma.hifigan_pre_post", "conv_mfma.glow_encoder",
                                            "conv_mfma.glow_decoder",     "elementwise",
                                            "mrf_small.hifigan_narrow_stage"};

// Launches per kernel NAME since the last mi355tts_profile_reset (always counted: one relaxed atomic add per launch) —
// mi355tts_kernel_counts_json.  The class counters above cannot tell a kernel from the fallback that would take its place
// (rb_group_kernel -> conv_group_kernel, rb_pair_group_kernel -> pair_group_kernel:
// same launch counts per class, same bits by design), so the device tests assert on these.
enum KName {
  KN_CONV_MFMA = 0, KN_CONV_M128, KN_CONV_GROUP, KN_RB_CONV, KN_RB_GROUP, KN_RB_GROUP_SNAKE, KN_PAIR, KN_PAIR_GROUP, KN_RB_PAIR,
  KN_RB_PAIR_GROUP, KN_CONV_BF16, KN_CONV_BF16_GROUP, KN_PAIR_BF16, KN_PAIR_BF16_GROUP, KN_MRF_SMALL, KN_MRF8, KN_GATE16, KN_GATE16_WIDE, KN_LIN16,
  KN_LIN16_LN, KN_LIN16_WIDE, KN_GLOW_TAIL, KN_OPROJ_LN, KN_POST_CONV, KN_WAVE_OUT, KN_ATTENTION, KN_CONV_F16, KN_CONV_F16_GROUP, KN_POST_F16,
  KN_PACK_OCTETS, KN_PAIR_F16_GROUP, KN_WN_F16, KN_RB_GROUP_NB4, KN_COUNT
};
static const char* kname_name[KN_COUNT] = {
    "conv_mfma_kernel", "conv_mfma_kernel.m128", "conv_group_kernel", "rb_conv_kernel", "rb_group_kernel", "rb_group_kernel.snake",
    "resblock_pair_kernel", "pair_group_kernel", "rb_pair_kernel", "rb_pair_group_kernel", "conv_bf16_kernel", "conv_bf16_group_kernel",
    "pair_bf16_kernel", "pair_bf16_group_kernel", "mrf_small_kernel", "mrf8_kernel", "gate16_kernel", "gate16_kernel.wide", "lin16_kernel", "lin16_kernel.ln", "lin16_kernel.wide",
    "glow_tail_kernel", "oproj_ln_kernel", "post_conv_kernel", "wave_out_kernel", "attention_mfma_kernel", "conv_f16_kernel", "conv_f16_group_kernel",
    "post_f16_kernel", "pack_octets_kernel", "pair_f16_group_kernel", "wn_f16_kernel", "rb_group_kernel.nb4"};
// the launch helpers without a context argument (launch_conv_k, launch_group_k) count through this: set by run_plan / run_group
static thread_local std::atomic<long long>* g_kn = nullptr;
// the kernel name (and launch sub-key: output rows / channels) of the launch inside the running ProfScope: the scope's
// destructor files its event pair under them (mi355tts_profile_kernels_json: durations per kernel NAME, not only per class)
static thread_local int g_last_kn = -1, g_last_sub = 0;
static inline void kn_add(int k) {
  g_last_kn = k;
  if (g_kn) g_kn[k].fetch_add(1, std::memory_order_relaxed);
}

// Host wait for a stream.  hipStreamSynchronize SPINS: a caller thread burns a core for the ~4 ms its kernels run (28 ms of CPU
// per utterance with eight callers, measured), and the reference's calling pattern is a ThreadPoolExecutor of up to 32 such
// threads per process (larynx/__init__.py:66-67, :146) — times 8 ranks on a node.  Default (mode 3, adaptive): poll
// hipStreamQuery without sleeping for the first 60 us (short waits: the frame-count read-back on an idle GPU, a warm vocoder tail),
// then poll every ~20 us from nanosleep, with the calling thread's timer slack lowered to 1 us for the duration of the wait
// (the default slack of 50 us would add that much to every wake-up) and restored afterwards.
// MI355TTS_SYNC_MODE: 0 = hipStreamSynchronize, 1 = blocking event, 2 = query + 20 us sleep (no spin phase, default slack),
// 3 = adaptive.  Read once.
// Option "sync_mode" (mi355tts_set_option; process-wide, like the environment variable that seeds it) selects the mode at run time.
static std::atomic<int> g_sync_mode{[] { const char* e = std::getenv("MI355TTS_SYNC_MODE"); return e ? std::atoi(e) : 3; }()};
static hipError_t mi355_sync(hipStream_t s) {
  const int mode = g_sync_mode.load(std::memory_order_relaxed);
  if (mode == 1) {
    // one blocking event per (thread, device): an event belongs to the device that was current when it was created
    constexpr int MAXDEV = 16;
    thread_local hipEvent_t evs[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return hipStreamSynchronize(s);
    if (!evs[dev] && hipEventCreateWithFlags(&evs[dev], hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
      evs[dev] = nullptr;
      return hipStreamSynchronize(s);
    }
    hipError_t e = hipEventRecord(evs[dev], s);
    return e == hipSuccess ? hipEventSynchronize(evs[dev]) : e;
  }
  if (mode == 2) {
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
    }
  }
  if (mode == 3) {
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {  // spin phase
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec) > 60000) break;
    }
    // the calling thread's timer slack: lowered for the duration of this wait only and restored — and left alone where the
    // process may not change it (a seccomp filter that denies prctl: the sleeps then keep the default slack)
    const int slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
    const bool lowered = slack > 1000 && prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0) == 0;
    hipError_t e;
    for (;;) {
      e = hipStreamQuery(s);
      if (e != hipErrorNotReady) break;
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
    }
    if (lowered) prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
    return e;
  }
  return hipStreamSynchronize(s);
}

#ifndef MI355TTS_CALL_COALESCE_DEFAULT
#define MI355TTS_CALL_COALESCE_DEFAULT 0  // lanes of host_join.h's whole-call coalescing (0 = off)
#endif

struct Worker {
  hipStream_t stream = nullptr;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  size_t arena_pos = 0;
  int* pinned = nullptr;  // pinned host staging for frame counts
  size_t pinned_ints = 0;
  char* pinned_out = nullptr;  // pinned host staging for waveform outputs (grow-only)
  size_t pinned_out_bytes = 0;
  std::vector<ProfEvent> events;
  // profiled FLOP of the launches that follow = the padded-batch figure x this (sum of the rows' real lengths / (B x longest))
  double flop_scale = 1.0;
  // this worker's launches are neither profiled nor counted per kernel name (the dispatch self-check's own launches are not a
  // caller's: a worker-local switch, so concurrent calls on the context keep their samples and counts)
  bool quiet = false;
  // hardware-queue group of `stream` (streams of one group run their kernels one after the other): learnt by
  // probe_queue_groups at mi355tts_reserve, -1 = not probed (a worker created on demand)
  int qgroup = -1;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  // side streams for the independent MRF branches of a HiFi-GAN stage
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join[2] = {nullptr, nullptr};
  // the context's kernel-selection options as THIS call saw them at its start (glow_run / hifigan_run snapshot them once, so
  // a mi355tts_set_option from another thread never changes a call's schedule half way through)
  bool o_glow_fuse = true, o_gate16 = true, o_rb_conv = true, o_rb_pair = true, o_group_promote = true;
  bool o_snake = true;  // grouped launches whose workgroups are all resident go out in the snake order (group_snake_order)
  int o_gate16_wide = 512;  // wide passes (at least this many 16-row tiles; 0 = never): two row tiles per gate16 workgroup
};

struct mi355tts_ctx {
  int device = 0;
  int ncu = 256;  // compute units (hipGetDeviceProperties at create): the dispatch-order logic of grouped launches
  std::mutex mu;
  std::map<int, std::shared_ptr<GlowModel>> glow;
  std::map<int, std::shared_ptr<HifiModel>> hifi;
  int next_id = 1;
  std::vector<Worker*> free_workers;
  std::vector<Worker*> all_workers;
  // option flags: written by mi355tts_set_option / _set_profiling while calls are in flight on other threads -> atomics;
  // a call reads each flag ONCE at its start (glow_run / hifigan_run copy them: locals, and Worker::o_* for the launch
  // helpers) so one call never mixes schedules
  std::atomic<bool> profiling{false};
  std::atomic<bool> serial_branches{false};
  // calls currently holding a worker; with "adaptive_schedule" on and more than one in flight the vocoder
  // launches the members of a grouped step one by one (and never forks its MRF chains)
  std::atomic<int> active_calls{0};
  // calls in flight per hardware-queue group (index = Worker::qgroup; guarded by `mu`): acquire_worker hands out the free worker
  // whose group is the least busy
  std::vector<int> qgroup_busy;
  std::atomic<bool> adaptive_schedule{false};
  std::atomic<int> gate16_wide{512};  // ... with two row tiles per workgroup in passes of at least this many 16-row tiles (0 = never; same bits)
  std::atomic<bool> gate16{true};     // GlowTTS WaveNet gate convs on 16-row tiles (gate16.h) when the launch is small
  // GlowTTS column-owner launches (coltile.h: block tails, conv_o + LayerNorm) AND the whole-tile-in-LDS convs of
  // gate16.h's lin16_kernel (FFN / duration predictor / prenet / 1 x 1 convs, LayerNorm prologues): 0 = the generic tiles
  std::atomic<bool> glow_fuse{true};
  std::atomic<bool> voc_out{true};    // conv_post + peak and the delivery of the rows as two dedicated launches (voc_out.h); 0 = round 4's ten
  std::atomic<bool> mrf_small{true};  // narrow stages (C = 8 / 16) as one fused launch per stage (mrf_small.h)
  std::atomic<bool> mrf_group{true};  // grouped launches of the MRF chains' same-geometry convs (hifigan_forward.h)
  std::atomic<bool> rb_conv{true};    // grouped 128-row launches on the continuous-stream tile (rb_conv.h; same bits)
  std::atomic<bool> group_promote{true};  // batch-1 ResBlock steps move to the 128-row tile when the snake deal is balanced (promote_group_plans)
  // Grouped launches whose workgroups are all resident at once are laid out as a snake over the dispatcher's rounds
  // (group_snake_order).  That order encodes an OBSERVED dispatcher rule (workgroup i -> CU i mod #CUs);
  // mi355tts_dispatch_selfcheck times it against the plain order on this device (first 'high'-class vocoder load) and turns
  // it off where it does not win (a partitioned GPU, another CU count, a firmware that deals differently).  The ORDER of a
  // launch's workgroups never changes a result; the promotion rule (which picks the TILE, i.e. the summation order) is
  // decided from the CU count and the geometry alone and is never touched by a timing.
  std::atomic<bool> group_snake{true};
  std::atomic<int> selfcheck_state{0};  // 0 = not run, 4 = running, 1 = snake kept, 2 = snake order disabled, 3 = skipped / failed
  float selfcheck_plain_us = 0.f, selfcheck_snake_us = 0.f;
  std::atomic<bool> rb_pair{true};    // fused ResBlock steps (64 / 32 channels) on the 4-wave tile without a k-split (rb_pair.h)
  // Whole-call coalescing (host_join.h): concurrent batch-1 mi355tts_synthesize calls become the rows of fused padded calls,
  // at most `call_coalesce` of them in flight (0 = off).  A caller that finds a lane free while other passes are in flight
  // gathers for up to `call_coalesce_window_us`; a lone caller never waits.
  std::atomic<int> call_coalesce{MI355TTS_CALL_COALESCE_DEFAULT};
  std::atomic<int> call_coalesce_window_us{300};
  std::mutex join_mu;
  std::condition_variable join_cv;
  std::vector<struct CallReq*> join_q;  // waiting requests in arrival order (under join_mu, like everything below)
  int join_inflight = 0, join_rows_inflight = 0;  // fused passes in flight and the rows they carry
  bool join_gathering = false;                    // a leader-to-be is inside its gather window
  long long join_arrivals = 0;
  long long join_passes = 0, join_rows = 0;  // counters since the context was created (mi355tts_coalesce_stats)
  // recycled device blocks for the mel result objects: hipMalloc/hipFree synchronise
  // the whole device, which would serialise the concurrent per-utterance streams
  std::vector<std::pair<void*, size_t>> mel_pool;
  size_t mel_pool_cap = 256;  // raised by mi355tts_reserve to 3 x workers + slack
  std::map<void*, size_t> mel_sizes;  // true size of every block the pool has ever handed out
  struct Acc {
    long long launches = 0;
    double ms = 0, flop = 0;
  } prof[KC_COUNT];
  std::map<std::pair<int, int>, Acc> prof_kn[KC_COUNT];  // per class: (kernel name, sub-key) -> the same sums
  std::atomic<long long> kn[KN_COUNT] = {};  // launches per kernel name (KName)
};

struct mi355tts_mel {
  mi355tts_ctx* ctx;
  int B, M, ld;
  float* raw = nullptr;   // [B][M][ld]
  float* voc = nullptr;   // [B][M][ld]
  int* frames_dev = nullptr;
  std::vector<int32_t> frames;
  int max_frames = 0;
  size_t raw_bytes = 0;  // allocation size of raw / voc (pool bookkeeping)
};

static inline void kn_hit(mi355tts_ctx* ctx, int k) {
  g_last_kn = k;
  ctx->kn[k].fetch_add(1, std::memory_order_relaxed);
}

// The kernel-selection options as a call sees them: taken when the call checks its worker out, so EVERY entry point (the op /
// bench entry points and the denoiser bias too, not only glow_run / hifigan_run) launches under the context's current options
// and never under what the worker's previous call left behind.
static void snapshot_options(mi355tts_ctx* ctx, Worker* w) {
  w->o_glow_fuse = ctx->glow_fuse.load();
  w->o_gate16 = ctx->gate16.load();
  w->o_rb_conv = ctx->rb_conv.load();
  w->o_rb_pair = ctx->rb_pair.load();
  w->o_group_promote = ctx->group_promote.load();
  w->o_snake = ctx->group_snake.load();
  w->o_gate16_wide = ctx->gate16_wide.load();
}

static int acquire_worker(mi355tts_ctx* ctx, Worker** out) {
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->free_workers.empty()) {
      // the free worker whose hardware queue carries the fewest calls right now (ties, and workers that were never probed: the
      // most recently released one, as before)
      size_t pick = ctx->free_workers.size() - 1;
      if (!ctx->qgroup_busy.empty()) {
        // (MI355TTS_QUEUE_POLICY = 1, probe: an idle queue first, otherwise the BUSIEST one — exclusive queues for as many calls as
        // there are queues, the rest piled on one)
        static const int policy = [] { const char* e = std::getenv("MI355TTS_QUEUE_POLICY"); return e ? std::atoi(e) : 0; }();
        int best = 1 << 30;
        for (size_t i = ctx->free_workers.size(); i-- > 0;) {
          const int g = ctx->free_workers[i]->qgroup;
          int busy = (g >= 0 && g < (int)ctx->qgroup_busy.size()) ? ctx->qgroup_busy[g] : 0;
          if (policy == 1 && busy > 0) busy = 1000 - busy;
          if (busy < best) {
            best = busy;
            pick = i;
          }
        }
      }
      *out = ctx->free_workers[pick];
      ctx->free_workers.erase(ctx->free_workers.begin() + (long)pick);
      if ((*out)->qgroup >= 0 && (*out)->qgroup < (int)ctx->qgroup_busy.size()) ctx->qgroup_busy[(*out)->qgroup] += 1;
      (*out)->arena_pos = 0;
      (*out)->flop_scale = 1.0;
      snapshot_options(ctx, *out);
      ctx->active_calls.fetch_add(1, std::memory_order_relaxed);
      return 0;
    }
  }
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = new Worker();
  hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete w;
    return fail(MI355TTS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  w->pinned_ints = 4096;
  e = hipHostMalloc(&w->pinned, w->pinned_ints * sizeof(int), hipHostMallocDefault);
  if (e != hipSuccess) {
    hipStreamDestroy(w->stream);
    delete w;
    return fail(MI355TTS_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e));
  }
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->all_workers.push_back(w);
  }
  ctx->active_calls.fetch_add(1, std::memory_order_relaxed);
  snapshot_options(ctx, w);
  *out = w;
  return 0;
}

static void drain_profile(mi355tts_ctx* ctx, Worker* w) {
  if (w->events.empty()) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto& ev : w->events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
      ctx->prof[ev.cls].launches++;
      ctx->prof[ev.cls].ms += ms;
      ctx->prof[ev.cls].flop += ev.flop;
      mi355tts_ctx::Acc& k = ctx->prof_kn[ev.cls][std::make_pair(ev.kn, ev.sub)];
      k.launches++;
      k.ms += ms;
      k.flop += ev.flop;
    }
    w->event_pool.emplace_back(ev.a, ev.b);
  }
  w->events.clear();
}

static void release_worker(mi355tts_ctx* ctx, Worker* w) {
  ctx->active_calls.fetch_sub(1, std::memory_order_relaxed);
  drain_profile(ctx, w);
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (w->qgroup >= 0 && w->qgroup < (int)ctx->qgroup_busy.size() && ctx->qgroup_busy[w->qgroup] > 0) ctx->qgroup_busy[w->qgroup] -= 1;
  ctx->free_workers.push_back(w);
}

struct WorkerGuard {
  mi355tts_ctx* ctx;
  Worker* w;
  ~WorkerGuard() {
    if (w) release_worker(ctx, w);
  }
};

// grow-only workspace: a call computes its total need, then carves.
static int reserve(Worker* w, size_t bytes) {
  if (bytes <= w->arena_bytes) return 0;
  if (w->arena) {
    HIPCHECK(hipStreamSynchronize(w->stream));
    HIPCHECK(hipFree(w->arena));
    w->arena = nullptr;
    w->arena_bytes = 0;
  }
  size_t want = bytes + bytes / 8 + (1 << 20);
  hipError_t e = hipMalloc(&w->arena, want);
  if (e != hipSuccess) return fail(MI355TTS_ERR_NOMEM, "hipMalloc(%zu) for workspace: %s", want, hipGetErrorString(e));
  w->arena_bytes = want;
  return 0;
}
static int reserve_pinned_out(Worker* w, size_t bytes) {
  if (bytes <= w->pinned_out_bytes) return 0;
  if (w->pinned_out) {
    HIPCHECK(hipHostFree(w->pinned_out));
    w->pinned_out = nullptr;
    w->pinned_out_bytes = 0;
  }
  const size_t want = bytes + bytes / 4 + (1 << 16);
  hipError_t e = hipHostMalloc(&w->pinned_out, want, hipHostMallocDefault);
  if (e != hipSuccess) return fail(MI355TTS_ERR_NOMEM, "hipHostMalloc(%zu) for output staging: %s", want, hipGetErrorString(e));
  w->pinned_out_bytes = want;
  return 0;
}
struct Carver {
  size_t pos = 0;
  size_t take(size_t bytes) {
    size_t off = (pos + 255) & ~(size_t)255;
    pos = off + bytes;
    return off;
  }
};
