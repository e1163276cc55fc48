// mi355tts host runtime — coalescing of concurrent batch-1 calls' GlowTTS passes.
// (one translation unit: included once by mi355tts.hip, after glow_forward.h / hifigan_forward.h)
//
// The reference fans a text's sentences out over a ThreadPoolExecutor (larynx/__init__.py:146-157, 187-190): N host
// threads each run GlowTTS -> HiFi-GAN for one sentence.  On the GPU the GlowTTS pass of ONE utterance is ~150 launches of
// 20-240 workgroups each — latency chains that keep a hardware queue busy for ~1.2 ms while using a fraction of the chip
// — and costs the same for 1 row or 8 (measured: tools/voc_only_probe.py; profiles/NOTES.md).  So the batch-1 calls that
// are waiting at the moment a pass starts share it: the first caller that finds no pass in flight becomes the leader,
// takes every compatible waiting request (same model, scales, audio settings, id residency) as the rows of one padded
// batch, runs mi355tts_glow_infer_rows' path on its own stream (each row draws the noise of ITS OWN seed, and every launch
// uses the tile a batch-1 call uses — GlowCall::solo_tiles — so a row equals its batch-1 result bit for bit), records an
// event, and hands each caller a one-row view of the result.  Every caller then runs its vocoder pass on its own stream
// behind that event.  A lone caller is a leader with one row: the same launches as before.
//
// Option "glow_coalesce", OFF by default: on the headline load (8 batch-1 calls in flight, 'high' vocoder) passes carried
// 2.9 rows on average and throughput was 261.4 vs 262.8 utterances/s, the half mode 568 vs 586 (profiles/NOTES.md): callers
// wait for the pass in flight (whose ~100 small launches queue behind other calls' vocoder kernels), which costs what the
// saved launches gain.  Kept for GlowTTS-dominated loads; results are bit-identical either way.
#pragma once

struct GlowBatch {
  mi355tts_mel* mel = nullptr;  // B rows, owned
  hipEvent_t ready = nullptr;   // recorded on the leader's stream behind the pass (nullptr for a one-row pass)
  int device = 0;
  ~GlowBatch() {
    DeviceScope ds(device);
    if (ready) {
      hipEventSynchronize(ready);  // nothing may still write the blocks that go back to the pool
      hipEventDestroy(ready);
    }
    mel_destroy(mel);
  }
};

struct GlowJoinReq {
  // request
  const GlowModel* gm = nullptr;
  const int64_t* ids = nullptr;
  int32_t len = 0;
  float noise_scale = 0.f, length_scale = 1.f;
  uint64_t seed = 0;
  const mi355tts_audio_settings* audio = nullptr;
  uint32_t flags = 0;  // MI355TTS_IN_DEVICE or 0
  // result
  std::shared_ptr<GlowBatch> batch;
  int row = -1;
  int rc = 0;
  std::string err;
  bool done = false;
  bool solo_retry = false;  // the shared pass failed (e.g. ANOTHER row hit the frame cap): this caller runs a pass of its own
};

constexpr int GLOW_JOIN_MAX_ROWS = 16;

static bool glow_join_compatible(const GlowJoinReq& a, const GlowJoinReq& b) {
  if (a.gm != b.gm || a.noise_scale != b.noise_scale || a.length_scale != b.length_scale || a.flags != b.flags) return false;
  if ((a.audio == nullptr) != (b.audio == nullptr)) return false;
  return !a.audio || std::memcmp(a.audio, b.audio, sizeof(mi355tts_audio_settings)) == 0;
}

// the pass of `rows` (>= 1 requests, rows[0] = the leader's) on the leader's worker; fills batch / row / rc of every row
static void glow_join_run(mi355tts_ctx* ctx, Worker* w, std::vector<GlowJoinReq*>& rows) {
  const int n = (int)rows.size();
  std::vector<const int64_t*> ptrs(n);
  std::vector<int32_t> lens(n);
  std::vector<uint64_t> seeds(n);
  int ld = 1;
  for (int b = 0; b < n; ++b) {
    ptrs[b] = rows[b]->ids;
    lens[b] = rows[b]->len;
    seeds[b] = rows[b]->seed;
    ld = std::max(ld, (int)rows[b]->len);
  }
  GlowCall c;
  c.ids = ptrs[0];
  c.id_lens = lens.data();
  c.B = n;
  c.ids_ld = ld;
  c.noise_scale = rows[0]->noise_scale;
  c.length_scale = rows[0]->length_scale;
  c.audio = rows[0]->audio;
  c.flags = rows[0]->flags;
  if (n > 1) {
    c.row_ids = ptrs.data();
    c.row_seeds = seeds.data();
    c.solo_tiles = true;
  } else {
    c.seed = seeds[0];
  }
  int rc = 0, Pmax = 0;
  mi355tts_mel* mel = nullptr;
  auto batch = std::make_shared<GlowBatch>();
  batch->device = ctx->device;
  rc = glow_precheck(rows[0]->gm, c, &Pmax);
  if (rc == 0) rc = glow_run(ctx, w, rows[0]->gm, c, Pmax, false, &mel);
  if (rc == 0) {
    batch->mel = mel;
    if (n > 1) {
      hipError_t e = hipEventCreateWithFlags(&batch->ready, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(batch->ready, w->stream);
      if (e != hipSuccess) rc = fail(MI355TTS_ERR_HIP, "event behind a coalesced GlowTTS pass: %s", hipGetErrorString(e));
    }
  }
  const std::string msg = rc ? g_err : std::string();
  for (int b = 0; b < n; ++b) {
    rows[b]->rc = rc;
    rows[b]->err = msg;
    rows[b]->row = b;
    if (rc == 0) rows[b]->batch = batch;
    // a resource failure of a SHARED pass says nothing about a rider's own request: each rider (the leader included)
    // then runs a solitary pass and reports what that one returns
    rows[b]->solo_retry = rc != 0 && n > 1;
  }
  // on failure `batch` dies here: its destructor waits for whatever the pass queued before the blocks go back to the pool
  if (rc != 0 && w->stream) hipStreamSynchronize(w->stream);
}

// Submit this call's GlowTTS request; returns when a pass that contains it has been queued (by this thread as the leader,
// or by another).  `w` = the caller's own worker: the leader's pass runs on it.
static int glow_join(mi355tts_ctx* ctx, Worker* w, GlowJoinReq& req) {
  std::unique_lock<std::mutex> lk(ctx->join_mu);
  ctx->join_q.push_back(&req);
  while (!req.done) {
    // "busy" is per compatibility class (model + scales + audio settings + id residency): a caller whose request no pass
    // in flight could have taken leads its own pass at once instead of sleeping until an unrelated leader returns
    bool busy = false;
    for (const GlowJoinReq* l : ctx->join_leaders) busy = busy || glow_join_compatible(*l, req);
    if (!busy) {
      // lead: this request plus every compatible one that is waiting, in arrival order
      ctx->join_leaders.push_back(&req);
      std::vector<GlowJoinReq*> rows{&req};
      std::vector<GlowJoinReq*> rest;
      for (GlowJoinReq* r : ctx->join_q) {
        if (r == &req) continue;
        if ((int)rows.size() < GLOW_JOIN_MAX_ROWS && glow_join_compatible(req, *r)) rows.push_back(r);
        else rest.push_back(r);
      }
      ctx->join_q.swap(rest);
      lk.unlock();
      glow_join_run(ctx, w, rows);
      lk.lock();
      for (GlowJoinReq* r : rows) r->done = true;
      ctx->join_passes += 1;
      ctx->join_rows += (long long)rows.size();
      ctx->join_leaders.erase(std::find(ctx->join_leaders.begin(), ctx->join_leaders.end(), &req));
      ctx->join_cv.notify_all();
      break;
    }
    ctx->join_cv.wait(lk);
  }
  lk.unlock();
  if (req.solo_retry) {  // outside the queue: a pass of this request alone, on this caller's own worker
    std::vector<GlowJoinReq*> self{&req};
    glow_join_run(ctx, w, self);
  }
  if (req.rc != 0) return fail(req.rc, "%s", req.err.c_str());
  return 0;
}

// one row of a coalesced pass as a mel object the vocoder can read (not owned: never passed to mel_destroy)
static void glow_join_view(const GlowJoinReq& req, mi355tts_mel* view) {
  const mi355tts_mel* m = req.batch->mel;
  view->ctx = m->ctx;
  view->B = 1;
  view->M = m->M;
  view->ld = m->ld;
  const size_t row_floats = (size_t)m->M * (size_t)m->ld;
  view->raw = m->raw ? m->raw + (size_t)req.row * row_floats : nullptr;
  view->voc = m->voc ? m->voc + (size_t)req.row * row_floats : nullptr;
  view->frames_dev = m->frames_dev + req.row;
  view->frames.assign(1, m->frames[req.row]);
  view->max_frames = m->frames[req.row];
  view->raw_bytes = 0;
}
