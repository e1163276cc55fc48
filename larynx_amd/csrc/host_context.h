// mi355tts host runtime — context, per-call workers (stream + workspace + pinned staging), device-block pool
// (one translation unit: included once by mi355tts.hip, after the kernel headers)
#pragma once

// ------------------------------------------------------------------ context
struct ProfEvent {
  hipEvent_t a, b;
  int cls;
  double flop;
};
enum KClass { KC_RESBLOCK = 0, KC_UPSAMPLE, KC_VOC_IO, KC_GLOW_ENC_CONV, KC_GLOW_DEC_CONV, KC_SMALL, KC_MRF_NARROW, KC_COUNT };
static const char* kclass_name[KC_COUNT] = {"conv_mfma.hifigan_resblock", "conv_mfma.hifigan_upsample",
                                            "conv_mfma.hifigan_pre_post", "conv_mfma.glow_encoder",
                                            "conv_mfma.glow_decoder",     "elementwise",
                                            "mrf_small.hifigan_narrow_stage"};

// Host wait for a stream.  hipStreamSynchronize SPINS (a caller thread burns a core while its ~4 ms of kernels run, and 8 of
// them contend with the launching threads for the runtime's locks); MI355TTS_SYNC_MODE=1 waits on a blocking event
// (interrupt), =2 polls hipStreamQuery with a short sleep.  Read once.
static hipError_t mi355_sync(hipStream_t s) {
  static const int mode = [] { const char* e = std::getenv("MI355TTS_SYNC_MODE"); return e ? std::atoi(e) : 0; }();
  if (mode == 1) {
    thread_local hipEvent_t ev = nullptr;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) return hipStreamSynchronize(s);
    hipError_t e = hipEventRecord(ev, s);
    return e == hipSuccess ? hipEventSynchronize(ev) : e;
  }
  if (mode == 2) {
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      struct timespec ts = {0, 20000};
      nanosleep(&ts, nullptr);
    }
  }
  return hipStreamSynchronize(s);
}

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
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  // side streams for the independent MRF branches of a HiFi-GAN stage
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join[2] = {nullptr, nullptr};
  // the context's kernel-selection options as THIS call saw them at its start (glow_run / hifigan_run snapshot them once, so
  // a mi355tts_set_option from another thread never changes a call's schedule half way through)
  bool o_glow_fuse = true, o_gate16 = true, o_rb_conv = true, o_rb_pair = true, o_group_promote = true;
  // option "glow_priority": the acoustic model's ~140 small launches of a fused call go out on a HIGH-priority stream of
  // their own (created on first use), the vocoder follows on `stream` behind `ev_glow`
  hipStream_t gstream = nullptr;
  hipEvent_t ev_glow = nullptr;
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
  std::atomic<bool> adaptive_schedule{false};
  std::atomic<bool> gate16{true};     // GlowTTS WaveNet gate convs on 16-row tiles (gate16.h) when the launch is small
  // GlowTTS column-owner launches (coltile.h: block tails, conv_o + LayerNorm) AND the whole-tile-in-LDS convs of
  // gate16.h's lin16_kernel (FFN / duration predictor / prenet / 1 x 1 convs, LayerNorm prologues): 0 = the generic tiles
  std::atomic<bool> glow_fuse{true};
  std::atomic<bool> mrf_small{true};  // narrow stages (C = 8 / 16) as one fused launch per stage (mrf_small.h)
  std::atomic<bool> mrf_group{true};  // grouped launches of the MRF chains' same-geometry convs (hifigan_forward.h)
  std::atomic<bool> rb_conv{true};    // grouped 128-row launches on the continuous-stream tile (rb_conv.h; same bits)
  std::atomic<bool> group_promote{true};  // batch-1 ResBlock steps move to the 128-row tile when the snake deal is balanced (promote_group_plans)
  std::atomic<bool> rb_pair{true};    // fused ResBlock steps (64 / 32 channels) on the 4-wave tile without a k-split (rb_pair.h)
  // mi355tts_synthesize: GlowTTS on a high-priority stream of the call's worker (see Worker::gstream).  The hardware queues
  // run one kernel at a time each and the runtime maps all bulk streams onto 4 of them: a call's chain of ~140 small
  // dependent launches otherwise advances one launch per 200 us ResBlock kernel of the stream it shares a queue with
  std::atomic<int> glow_priority{0};
  // concurrent batch-1 mi355tts_synthesize calls share ONE GlowTTS pass (host_join.h): the callers waiting when a pass
  // starts become its rows.  Off by default: measured neutral to -1 % on the 'high' vocoder (profiles/NOTES.md)
  std::atomic<bool> glow_coalesce{false};
  std::mutex join_mu;
  std::condition_variable join_cv;
  std::vector<struct GlowJoinReq*> join_q;
  std::vector<const struct GlowJoinReq*> join_leaders;  // the leaders of the passes in flight (under join_mu)
  long long join_passes = 0, join_rows = 0;  // under join_mu
  // recycled device blocks for the mel result objects: hipMalloc/hipFree synchronise
  // the whole device, which would serialise the concurrent per-utterance streams
  std::vector<std::pair<void*, size_t>> mel_pool;
  size_t mel_pool_cap = 256;  // raised by mi355tts_reserve to 3 x workers + slack
  std::map<void*, size_t> mel_sizes;  // true size of every block the pool has ever handed out
  struct Acc {
    long long launches = 0;
    double ms = 0, flop = 0;
  } prof[KC_COUNT];
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

static int acquire_worker(mi355tts_ctx* ctx, Worker** out) {
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->free_workers.empty()) {
      *out = ctx->free_workers.back();
      ctx->free_workers.pop_back();
      (*out)->arena_pos = 0;
      (*out)->flop_scale = 1.0;
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
    }
    w->event_pool.emplace_back(ev.a, ev.b);
  }
  w->events.clear();
}

static void release_worker(mi355tts_ctx* ctx, Worker* w) {
  ctx->active_calls.fetch_sub(1, std::memory_order_relaxed);
  drain_profile(ctx, w);
  std::lock_guard<std::mutex> lk(ctx->mu);
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
