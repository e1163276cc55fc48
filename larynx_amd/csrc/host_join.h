// mi355tts host runtime — whole-call coalescing of concurrent batch-1 mi355tts_synthesize calls.
// (one translation unit: included once by mi355tts.hip, after glow_forward.h / hifigan_forward.h)
//
// The reference fans a text's sentences out over a ThreadPoolExecutor (larynx/__init__.py:146-157, 187-190): N host
// threads each run GlowTTS -> HiFi-GAN for ONE sentence.  On the GPU a batch-1 call is ~140 GlowTTS launches of 20-240
// workgroups (latency chains that cost the same for 1 row or 8) and 25 vocoder launches whose tiles, at batch 1, fill the
// chip less evenly than those of a padded batch (ResBlock class 0.70 of peak at batch 1, 0.75 at 4 rows, 0.78 at 8:
// bench.py --batch).  So the callers that are waiting when a "lane" frees become the rows of ONE fused padded call —
// acoustic pass AND vocoder — on the leader's worker: each row with its own ids, its own seed's noise stream
// (mi355tts_glow_infer_rows semantics), its own pause padding, its own output buffers and strides.
//
//   * option "call_coalesce" = L: at most L fused passes in flight per context (0 = off: every call runs alone).  A
//     caller that finds a lane free leads at once — a LONE caller never waits: nothing in flight, no gather window —; a
//     caller that finds a lane free while other passes are in flight first gathers for "call_coalesce_window_us" (it ends
//     early once no further request has arrived for a third of it): callers come back from a finished pass within a few
//     tens of microseconds of each other, and the pass in flight keeps the GPU busy meanwhile.  A pass takes at most
//     ceil((rows in flight + rows waiting) / L) rows, so that the lanes stay balanced (8 callers, 2 lanes: 4 + 4).
//   * compatible = same models, scales, audio settings, denoiser strength, residency flags; anything else (explicit noise,
//     speaker ids, batches, more than ATTM_MAXP ids) runs alone as before.
//   * results: a row of a padded batch is computed by other tiles than its solitary call (the launches are sized for the
//     batch), i.e. another f32 summation order: equal to the solitary call within f32 round-off (tests: waveform RMS <= 1e-5,
//     int16 within 1 LSB, frames identical) and within the golden tolerances; NOT bit-identical — set the option to 0 where
//     bit-reproducibility across loads matters more than throughput.
//   * a failure of a SHARED pass says nothing about a rider's own request: every rider (the leader included) then runs its
//     solitary call and reports what that returns.
#pragma once

struct CallReq {
  // request: the arguments of one batch-1 mi355tts_synthesize call
  const GlowModel* gm = nullptr;
  HifiModel* hm = nullptr;
  int vocoder = 0;
  const int64_t* ids = nullptr;
  int32_t len = 0;
  float noise_scale = 0.f, length_scale = 1.f;
  uint64_t seed = 0;
  const mi355tts_audio_settings* audio = nullptr;
  uint32_t flags = 0;  // MI355TTS_IN_DEVICE | MI355TTS_OUT_DEVICE
  float denoiser_strength = 0.f;
  VocRow out;
  // result
  int32_t frames = 0;
  int rows_in_pass = 0;
  int rc = 0;
  std::string err;
  bool taken = false;  // a leader has made this request a row of its pass (under join_mu): it only waits for `done` now
  bool done = false;
  bool solo_retry = false;
};

constexpr int CALL_JOIN_MAX_ROWS = VOC_MAX_ROWS;

static bool call_join_compatible(const CallReq& a, const CallReq& b) {
  if (a.gm != b.gm || a.hm != b.hm || a.noise_scale != b.noise_scale || a.length_scale != b.length_scale || a.flags != b.flags ||
      a.denoiser_strength != b.denoiser_strength)
    return false;
  if ((a.audio == nullptr) != (b.audio == nullptr)) return false;
  return !a.audio || std::memcmp(a.audio, b.audio, sizeof(mi355tts_audio_settings)) == 0;
}

// the fused pass of `rows` (>= 1 requests, rows[0] = the leader's) on a worker of its own; fills frames / rc of every row
static void call_join_run(mi355tts_ctx* ctx, std::vector<CallReq*>& rows) {
  const int n = (int)rows.size();
  std::vector<const int64_t*> ptrs(n);
  std::vector<int32_t> lens(n);
  std::vector<uint64_t> seeds(n);
  std::vector<VocRow> outs(n);
  int ld = 1;
  for (int b = 0; b < n; ++b) {
    ptrs[b] = rows[b]->ids;
    lens[b] = rows[b]->len;
    seeds[b] = rows[b]->seed;
    outs[b] = rows[b]->out;
    ld = std::max(ld, (int)rows[b]->len);
  }
  const CallReq& lead = *rows[0];
  GlowCall c;
  c.ids = ptrs[0];
  c.id_lens = lens.data();
  c.B = n;
  c.ids_ld = ld;
  c.noise_scale = lead.noise_scale;
  c.length_scale = lead.length_scale;
  c.audio = lead.audio;
  c.flags = lead.flags & MI355TTS_IN_DEVICE;
  if (n > 1) {
    c.row_ids = ptrs.data();
    c.row_seeds = seeds.data();
  } else {
    c.seed = seeds[0];  // a lone caller's pass: exactly the launches (and copies) of its solitary call
  }
  VocCall v;
  v.denoiser_strength = lead.denoiser_strength;
  v.flags = lead.flags & MI355TTS_OUT_DEVICE;
  v.rows = outs.data();
  int rc = 0;
  auto run = [&]() -> int {
    int Pmax = 0;
    CHECK(glow_precheck(lead.gm, c, &Pmax));
    HIPCHECK(hipSetDevice(ctx->device));
    Worker* w = nullptr;
    CHECK(acquire_worker(ctx, &w));
    WorkerGuard guard{ctx, w};
    mi355tts_mel* mel = nullptr;
    struct MelDrop {
      Worker* w;
      mi355tts_mel* m;
      ~MelDrop() {
        if (!m) return;
        mi355_sync(w->stream);  // its blocks go back to the pool: nothing queued may still read them
        mel_destroy(m);
      }
    } drop{w, nullptr};
    CHECK(glow_run(ctx, w, lead.gm, c, Pmax, false, &mel));
    drop.m = mel;
    for (int b = 0; b < n; ++b) rows[b]->frames = mel->frames[b];  // (also on the error returns below: TOO_SMALL reports the real counts)
    CHECK(hifigan_precheck(ctx, lead.hm, lead.vocoder, mel->frames.data(), n, mel->M, mel->max_frames, v));
    return hifigan_run(ctx, w, lead.hm, mel, v);
  };
  rc = run();
  const std::string msg = rc ? g_err : std::string();
  for (int b = 0; b < n; ++b) {
    rows[b]->rc = rc;
    rows[b]->err = msg;
    rows[b]->rows_in_pass = n;
    rows[b]->solo_retry = rc != 0 && n > 1;
  }
}

// Submit this call; returns when a pass that contained it has finished (led by this thread or by another).  req.solo_retry:
// the shared pass failed and the caller runs its solitary call.
static int call_join(mi355tts_ctx* ctx, CallReq& req, int lanes) {
  using clock = std::chrono::steady_clock;
  std::unique_lock<std::mutex> lk(ctx->join_mu);
  ctx->join_q.push_back(&req);
  ctx->join_arrivals += 1;
  ctx->join_cv.notify_all();  // a gathering leader counts arrivals
  while (!req.done) {
    if (!req.taken && !ctx->join_gathering && ctx->join_inflight < lanes) {
      // lead the next pass
      if (ctx->join_inflight > 0) {
        // other passes keep the GPU busy: gather the callers that are on their way back from the pass that just finished
        const int window_us = ctx->call_coalesce_window_us.load();
        if (window_us > 0) {
          ctx->join_gathering = true;
          const auto deadline = clock::now() + std::chrono::microseconds(window_us);
          const auto quiet = std::chrono::microseconds(std::max(1, window_us / 3));
          long long seen = ctx->join_arrivals;
          auto last = clock::now();
          for (;;) {
            int waiting = 0;
            for (const CallReq* r : ctx->join_q) waiting += call_join_compatible(req, *r) ? 1 : 0;
            if (waiting >= CALL_JOIN_MAX_ROWS) break;
            const auto until = std::min(deadline, last + quiet);
            if (clock::now() >= until) break;
            ctx->join_cv.wait_until(lk, until);
            if (ctx->join_arrivals != seen) {
              seen = ctx->join_arrivals;
              last = clock::now();
            }
          }
          ctx->join_gathering = false;
        }
      }
      int waiting = 0;
      for (const CallReq* r : ctx->join_q) waiting += call_join_compatible(req, *r) ? 1 : 0;
      const int cap = std::min(CALL_JOIN_MAX_ROWS, std::max(1, (ctx->join_rows_inflight + waiting + lanes - 1) / lanes));
      std::vector<CallReq*> rows{&req};
      std::vector<CallReq*> rest;
      for (CallReq* r : ctx->join_q) {
        if (r == &req) continue;
        if ((int)rows.size() < cap && call_join_compatible(req, *r)) rows.push_back(r);
        else rest.push_back(r);
      }
      ctx->join_q.swap(rest);
      for (CallReq* r : rows) r->taken = true;
      const int n = (int)rows.size();
      ctx->join_inflight += 1;
      ctx->join_rows_inflight += n;
      ctx->join_cv.notify_all();  // a waiting caller may lead the next lane
      lk.unlock();
      call_join_run(ctx, rows);
      lk.lock();
      ctx->join_inflight -= 1;
      ctx->join_rows_inflight -= n;
      for (CallReq* r : rows) r->done = true;
      ctx->join_passes += 1;
      ctx->join_rows += n;
      ctx->join_cv.notify_all();
      break;
    }
    ctx->join_cv.wait(lk);
  }
  lk.unlock();
  if (req.solo_retry) return 1;
  if (req.rc != 0) return fail(req.rc, "%s", req.err.c_str());
  return 0;
}
