// mi355tts host runtime — mi355tts_hifigan_infer: the HiFi-GAN layer schedule (hifi_gan/models.py:186-202) and the denoiser
// (one translation unit: included once by mi355tts.hip, after the kernel headers)
#pragma once

// ------------------------------------------------------------------ HiFi-GAN forward
extern "C" int mi355tts_hifigan_hop(mi355tts_ctx* ctx, int vocoder) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->hifi.find(vocoder);
  if (it == ctx->hifi.end()) return fail(MI355TTS_ERR_NO_MODEL, "no HiFi-GAN model %d", vocoder);
  return it->second->hop;
}

static int ensure_denoiser_bias(mi355tts_ctx* ctx, HifiModel* hm, int vocoder) {
  std::lock_guard<std::mutex> lk(hm->bias_mu);
  const int bi = hm->precision.load() == MI355TTS_PRECISION_F16 ? 1 : 0;
  if (hm->bias_ready[bi]) return 0;
  const int M = hm->hp.num_mels, hop = hm->hop;
  const int zf = 88;  // the reference's all-zero mel has 88 frames (hifi_gan.py:187,198)
  const long long N = (long long)zf * hop;
  if (N <= DN_FFT) return fail(MI355TTS_ERR_INVALID, "vocoder hop %d too small for the 1024-point denoiser STFT", hop);
  HIPCHECK(hipSetDevice(ctx->device));
  std::vector<float> zeros((size_t)M * zf, 0.f);
  int32_t fr = zf;
  mi355tts_mel* zm = nullptr;
  CHECK(mi355tts_mel_from_buffer(ctx, zeros.data(), &fr, 1, M, zf, nullptr, 0, &zm));
  float* dwav = nullptr;
  float* bias = nullptr;
  int rc = 0;
  if (hipMalloc(&dwav, sizeof(float) * (size_t)N) != hipSuccess || hipMalloc(&bias, sizeof(float) * (DN_FFT / 2 + 1)) != hipSuccess)
    rc = fail(MI355TTS_ERR_NOMEM, "hipMalloc denoiser bias");
  if (!rc) rc = mi355tts_hifigan_infer(ctx, vocoder, zm, 0.f, dwav, nullptr, N, MI355TTS_OUT_DEVICE);
  if (!rc) {
    Worker* w = nullptr;
    rc = acquire_worker(ctx, &w);
    if (!rc) {
      WorkerGuard guard{ctx, w};
      hipLaunchKernelGGL(stft_denoise_kernel, dim3(1, 1), dim3(256), 0, w->stream, dwav, (long long)N, zm->frames_dev, hop,
                         (const float*)nullptr, 0.f, (float*)nullptr, 1, bias);
      if (mi355_sync(w->stream) != hipSuccess) rc = fail(MI355TTS_ERR_HIP, "denoiser bias kernel failed");
    }
  }
  mel_destroy(zm);
  if (dwav) hipFree(dwav);
  if (rc) {
    if (bias) hipFree(bias);
    return rc;
  }
  hm->bias_spec[bi] = bias;
  hm->bias_ready[bi] = true;
  return 0;
}

// One vocoder call's outputs: per row [pad_before zeros][frames[b]*hop samples][zeros up to wav_ld]
// (the pads are the SSML pauses `_sentence_task` adds with np.pad, larynx/__init__.py:277-283).
struct VocRow {  // one row's destinations when the rows of a call belong to different callers (host_join.h)
  float* wav_f32 = nullptr;
  int16_t* wav_i16 = nullptr;
  int64_t wav_ld = 0;
  int pad_before = 0, pad_after = 0;
};
struct VocCall {
  float denoiser_strength = 0.f;
  float* wav_f32 = nullptr;
  int16_t* wav_i16 = nullptr;
  int64_t wav_ld = 0;
  uint32_t flags = 0;
  int pad_before = 0, pad_after = 0;
  // optional, [B] (B <= VOC_MAX_ROWS): per-row destinations, strides and pauses; the five fields above are then unused
  const VocRow* rows = nullptr;
};

// pins the model (see find_glow)
static int find_hifi(mi355tts_ctx* ctx, int vocoder, std::shared_ptr<HifiModel>* out) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->hifi.find(vocoder);
  if (it == ctx->hifi.end()) return fail(MI355TTS_ERR_NO_MODEL, "no HiFi-GAN model %d", vocoder);
  *out = it->second;
  return 0;
}

// Checks that need no worker (and may run the one-time bias-spectrum pass on a worker of their own).
static int hifigan_precheck(mi355tts_ctx* ctx, HifiModel* hm, int vocoder, const int32_t* frames, int B, int M, int Fmax,
                            const VocCall& c) {
  if (M != hm->hp.num_mels) return fail(MI355TTS_ERR_INVALID, "mel has %d channels, vocoder expects %d", M, hm->hp.num_mels);
  if (c.rows) {  // per-row destinations: every row against its own frame count
    if (B > VOC_MAX_ROWS) return fail(MI355TTS_ERR_INVALID, "internal: %d rows with per-row outputs", B);
    for (int b = 0; b < B; ++b) {
      const VocRow& r = c.rows[b];
      if (r.pad_before < 0 || r.pad_after < 0) return fail(MI355TTS_ERR_INVALID, "negative pause padding");
      const long long need = (long long)(frames ? frames[b] : 0) * hm->hop + r.pad_before + r.pad_after;
      if (frames && r.wav_ld < need) return fail(MI355TTS_ERR_TOO_SMALL, "wav_ld %lld < %lld samples", (long long)r.wav_ld, need);
    }
  } else {
    if (c.pad_before < 0 || c.pad_after < 0) return fail(MI355TTS_ERR_INVALID, "negative pause padding");
    const long long need = (long long)Fmax * hm->hop + c.pad_before + c.pad_after;
    if (Fmax >= 0 && c.wav_ld < need) return fail(MI355TTS_ERR_TOO_SMALL, "wav_ld %lld < %lld samples", (long long)c.wav_ld, need);
  }
  if (c.denoiser_strength > 0.f && Fmax != 0) {
    // the reference's STFT needs more than one 1024-sample frame per utterance
    // (larynx/audio.py:232-249 raises on shorter input)
    if (frames)
      for (int b = 0; b < B; ++b)
        if ((long long)frames[b] * hm->hop <= DN_FFT)
          return fail(MI355TTS_ERR_INVALID, "utterance %d has %d frames: too short for the denoiser", b, frames[b]);
    CHECK(ensure_denoiser_bias(ctx, hm, vocoder));
  }
  return 0;
}

// Workspace of one vocoder call (bytes, and where each piece sits).  ONE definition, used by the
// forward pass and by mi355tts_reserve.
struct HifiLayout {
  size_t plane = 0, Nld = 0, total = 0;
  int nbuf = 0, Tmax = 0;
  size_t o_buf[2 + 4 * 3], o_wav, o_i16, o_peak, o_wav2, o_fbuf;
};
static HifiLayout hifi_layout(const mi355tts_hifigan_hparams& h, int hop, int B, int F, bool denoise, bool split_out, int pads) {
  HifiLayout L;
  const int C0 = h.upsample_initial_channel;
  L.plane = (size_t)C0 * (size_t)((F + 3) & ~3);  // row strides are multiples of 4 floats (16-byte staging loads)
  long long len = F;
  for (int i = 0; i < h.num_upsamples; ++i) {
    len *= h.upsample_rates[i];
    L.plane = std::max(L.plane, (size_t)(C0 >> (i + 1)) * (size_t)((len + 3) & ~3LL));
  }
  const long long N = (long long)F * hop;
  L.Nld = (size_t)((N + 3) & ~3LL);
  L.nbuf = split_out ? 2 + 4 * h.num_kernels : 6;
  Carver cv;
  for (int i = 0; i < L.nbuf; ++i) L.o_buf[i] = cv.take(sizeof(float) * (size_t)B * L.plane);
  L.o_wav = cv.take(sizeof(float) * (size_t)B * L.Nld);
  L.o_i16 = cv.take(sizeof(short) * (size_t)B * (L.Nld + (size_t)pads));
  L.o_peak = cv.take(sizeof(float) * (size_t)B * (L.Nld / POST_TW + 2));  // post_conv_kernel's per-workgroup maxima of every row
  L.Tmax = denoise ? (int)((N - DN_FFT + DN_HOP - 1) / DN_HOP) : 0;
  L.o_wav2 = cv.take(denoise ? sizeof(float) * (size_t)B * L.Nld : 0);
  L.o_fbuf = cv.take(denoise ? sizeof(float) * (size_t)B * L.Tmax * DN_FFT : 0);
  L.total = cv.pos;
  return L;
}
static bool hifi_split_out(bool serial_branches, const mi355tts_hifigan_hparams& h) {
  return !serial_branches && h.num_kernels >= 2 && h.num_kernels <= 3;
}

// The forward pass proper on worker `w` (already checked by hifigan_precheck); ends with the
// stream synchronised and the outputs delivered.
static int hifigan_run(mi355tts_ctx* ctx, Worker* w, HifiModel* hm, const mi355tts_mel* mel, const VocCall& call) {
  const mi355tts_hifigan_hparams& h = hm->hp;
  const int B = mel->B, F = mel->max_frames, hop = hm->hop;
  const long long N = (long long)F * hop;
  const float denoiser_strength = call.denoiser_strength;
  float* const wav_f32 = call.wav_f32;
  int16_t* const wav_i16 = call.wav_i16;
  const int64_t wav_ld = call.wav_ld;
  const bool denoise = denoiser_strength > 0.f && F > 0;
  const bool out_dev = (call.flags & MI355TTS_OUT_DEVICE) != 0;
  const int pad0 = call.pad_before;
  hipStream_t s = w->stream;
  const VocRow* const prow = call.rows;
  if (prow && B > VOC_MAX_ROWS) return fail(MI355TTS_ERR_INVALID, "internal: %d rows with per-row outputs", B);
  if (F == 0 && prow) {
    for (int b = 0; b < B; ++b) {
      const VocRow& r = prow[b];
      if (out_dev) {
        if (r.wav_f32) HIPCHECK(hipMemsetAsync(r.wav_f32, 0, sizeof(float) * (size_t)r.wav_ld, s));
        if (r.wav_i16) HIPCHECK(hipMemsetAsync(r.wav_i16, 0, sizeof(int16_t) * (size_t)r.wav_ld, s));
      } else {
        if (r.wav_f32) std::memset(r.wav_f32, 0, sizeof(float) * (size_t)r.wav_ld);
        if (r.wav_i16) std::memset(r.wav_i16, 0, sizeof(int16_t) * (size_t)r.wav_ld);
      }
    }
    if (out_dev) HIPCHECK(mi355_sync(s));
    return 0;
  }
  if (F == 0) {
    if (out_dev) {
      if (wav_f32) HIPCHECK(hipMemsetAsync(wav_f32, 0, sizeof(float) * (size_t)B * wav_ld, s));
      if (wav_i16) HIPCHECK(hipMemsetAsync(wav_i16, 0, sizeof(int16_t) * (size_t)B * wav_ld, s));
      HIPCHECK(mi355_sync(s));
    } else {
      if (wav_f32) std::memset(wav_f32, 0, sizeof(float) * (size_t)B * wav_ld);
      if (wav_i16) std::memset(wav_i16, 0, sizeof(int16_t) * (size_t)B * wav_ld);
    }
    return 0;
  }
  // an error return after kernels were queued must not hand the worker (its arena!) to the
  // next call while they still run — possibly on the side streams
  struct DrainOnError {
    Worker* w;
    bool ok = false;
    ~DrainOnError() {
      if (ok) return;
      mi355_sync(w->stream);
      for (int i = 0; i < 2; ++i)
        if (w->aux[i]) mi355_sync(w->aux[i]);
    }
  } drain{w};
  const int C0 = h.upsample_initial_channel;
  const int Fp = (F + 3) & ~3;
  const int nk = h.num_kernels;
  struct FlopScale {  // profiled FLOP of a ragged batch count the rows' real frames, not B x the longest row
    Worker* w;
    ~FlopScale() { w->flop_scale = 1.0; }
  } fscale{w};
  {
    long long sum = 0;
    for (int b = 0; b < B; ++b) sum += mel->frames[b];
    w->flop_scale = (double)sum / ((double)B * F);
  }
  // The nk ResBlock chains of a stage are independent (MRF).  Each chain writes its own
  // output and the average is taken by the consumer's staging load (`split_out`).  A call
  // that has the GPU to itself also runs the chains on separate streams so their workgroups
  // interleave — at batch 1 one conv launch has fewer tiles than the chip has SIMDs
  // (`concurrent`); when other calls are in flight they fill the chip, and forking would only
  // make 3 x calls streams contend for the runtime's 4 hardware queues, so the call stays on
  // one stream (measured: 3-6 calls in flight, +8 % utterances/s).  Both forms compute the
  // same values in the same order: results do not depend on the load.
  // `serial_branches` (profiling / tests) additionally folds the average into the chains'
  // last epilogues (in-place accumulation, one output buffer).
  // the option flags, read once: a call never mixes schedules if an option changes while it runs
  const bool opt_serial = ctx->serial_branches.load(), opt_group = ctx->mrf_group.load(), opt_small = ctx->mrf_small.load();
  w->o_rb_conv = ctx->rb_conv.load();
  w->o_rb_pair = ctx->rb_pair.load();
  w->o_group_promote = ctx->group_promote.load();
  const int prec = hm->precision.load();
  const bool f16 = prec == MI355TTS_PRECISION_F16;  // the native fp16 generator (hifigan_f16.h): its own schedule, chains always write their own planes
  if (f16 && !hm->f16_ok) return fail(MI355TTS_ERR_INVALID, "internal: fp16 mode on a vocoder it does not cover");
  const bool split_out = f16 || hifi_split_out(opt_serial, h);
  // grouped (default): the chains stay on ONE stream and the same-geometry launches of a step go out as
  // one grouped launch (conv_group_kernel / pair_group_kernel) — the chip is filled from one launch, with
  // no stream fork/join and independently of what else is in flight.  "mrf_group" = 0 restores the
  // round-1 schedule (fork onto three streams while the call has the GPU to itself).
  // "adaptive_schedule" = 1 sends the members out one by one while other calls are in flight.  That was worth
  // +3 % utterances/s at 6 calls in flight with the round-2 mid-way kernels; with the final tiles the grouped
  // launch wins under load too (f32 +1.3 %, split-bf16 +5 %, same box), so the option is off by default.
  // Every form runs the same tiles with the same code: results do not depend on the load.
  const bool busy = ctx->adaptive_schedule.load() && ctx->active_calls.load(std::memory_order_relaxed) > 1;
  const bool grouped = split_out && nk == 3 && opt_group && !busy;
  const bool concurrent = split_out && !opt_group && !busy;
  if (concurrent && !w->aux[0]) {
    for (int i = 0; i < 2; ++i) {
      HIPCHECK(hipStreamCreateWithFlags(&w->aux[i], hipStreamNonBlocking));
      HIPCHECK(hipEventCreateWithFlags(&w->ev_join[i], hipEventDisableTiming));
    }
    HIPCHECK(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
  }
  // One workgroup target for the ResBlock launches in every schedule: the tile shape fixes the
  // k-split and with it the summation order, so a load-dependent choice would make results
  // depend on the load.  (With the final tile set 300, 700 and 1024 measured the same within
  // noise for the forked schedule.)  Tuning knob: MI355TTS_RB_TILES.
  static const int rb_env = [] { const char* e = std::getenv("MI355TTS_RB_TILES"); return e ? std::atoi(e) : 0; }();
  const int rb_tiles = rb_env > 0 ? rb_env : 1024;
  const int voc_host_len = B == 1 ? mel->frames[0] : -1;
  int pads = call.pad_before + call.pad_after;
  bool any_f32 = wav_f32 != nullptr, any_i16 = wav_i16 != nullptr;
  if (prow) {
    pads = 0;
    any_f32 = any_i16 = false;
    for (int b = 0; b < B; ++b) {
      pads = std::max(pads, prow[b].pad_before + prow[b].pad_after);
      any_f32 = any_f32 || prow[b].wav_f32;
      any_i16 = any_i16 || prow[b].wav_i16;
    }
  }
  const HifiLayout lay = hifi_layout(h, hop, B, F, denoise, split_out, pads);
  const size_t Nld = lay.Nld;
  const int nbuf = lay.nbuf, Tmax = lay.Tmax;
  CHECK(reserve(w, lay.total));
  char* base = w->arena;
  float* buf[2 + 4 * 3];
  for (int i = 0; i < nbuf; ++i) buf[i] = (float*)(base + lay.o_buf[i]);
  float* wav = (float*)(base + lay.o_wav);
  short* i16 = (short*)(base + lay.o_i16);
  unsigned* peak = (unsigned*)(base + lay.o_peak);
  const size_t o_wav2 = lay.o_wav2, o_fbuf = lay.o_fbuf;
  const int* d_frames = mel->frames_dev;

  static const bool voc_out_off = [] { const char* e = std::getenv("MI355TTS_NO_VOC_OUT"); return e && std::atoi(e) != 0; }();
  const long long peak_ld = (long long)(Nld / POST_TW + 2);
  bool peak_parts_ready = false;  // post_conv_kernel left the per-workgroup maxima of the FINAL waveform
  bool vo = true;
  if (f16) {
    if ((size_t)((mel->M + 7) / 8) * F * 16 > lay.plane * sizeof(float)) return fail(MI355TTS_ERR_INVALID, "internal: mel octets exceed a plane buffer");
    peak_parts_ready = any_i16 && !denoise;
    CHECK(hifigan_body_f16(ctx, w, hm, mel, buf, wav, Nld, peak_parts_ready ? reinterpret_cast<float*>(peak) : nullptr, peak_ld, voc_host_len,
                           opt_group, opt_group && w->o_rb_pair, s));
  } else {
  // stage input: `cur[0]` alone, or the nk chain outputs cur[0..nk) still to be averaged
  float* cur[3] = {buf[0], nullptr, nullptr};
  int ncur = 1;
  float* xu = buf[1];
  {  // conv_pre (models.py:187)
    ConvArgs a = base_args(mel->voc, (long long)mel->M * mel->ld, mel->ld, d_frames, 1, cur[0], (long long)C0 * Fp, Fp, d_frames, 1, 1, 3);
    CHECK(launch_conv(ctx, w, hm->pre, a, EPI_LINEAR, B, F, KC_VOC_IO, nullptr, 1024, voc_host_len));
  }
  float cur_div = 1.0f;  // the stage input is (cur[0] + ... + cur[ncur-1]) / cur_div
  auto set_inputs = [&](ConvArgs& a) {
    if (ncur > 1) {
      a.x2 = cur[1];
      a.x3 = ncur > 2 ? cur[2] : nullptr;
      a.in_div = cur_div;
    }
  };
  int mul = 1;
  int Lin = F;
  int ldin = Fp;
  int ch = C0;
  int flip = 0;  // which half of the chain-output buffers this stage writes
  for (int i = 0; i < h.num_upsamples; ++i) {
    const int u = h.upsample_rates[i], ku = h.upsample_kernel_sizes[i];
    const int cout = C0 >> (i + 1);
    const int Lout = Lin * u;
    const int ldo = (Lout + 3) & ~3;  // row stride of this stage's planes (16-byte staging loads)
    {  // x = ups[i](leaky_relu(x, 0.1))  (models.py:189-190)
      ConvArgs a = base_args(cur[0], (long long)ch * ldin, ldin, d_frames, mul, xu, (long long)cout * ldo, ldo, d_frames, mul * u, 1, ku / u - 1);
      set_inputs(a);
      a.in_slope = 0.1f;
      a.up = u;
      a.up_pad = (ku - u) / 2;
      CHECK(launch_conv(ctx, w, hm->ups[i], a, EPI_UPSAMPLE, B, Lin + ku / u - 1, KC_UPSAMPLE, nullptr, 1024, voc_host_len, prec));
    }
    mul *= u;
    ch = cout;
    const long long bs = (long long)ch * ldo;
    const float inv_nk = 1.0f / (float)nk;
    if (opt_small && !opt_serial && i < (int)hm->mrf.size() && hm->mrf[i].ok) {
      // narrow stage (C = 8 / 16): the three chains in ONE launch on LDS-resident tiles, written as two sums
      // (k = 3 + k = 7, and k = 11) that the consumer adds and divides by nk on load.  The stage-input plane and
      // the previous stage's chain outputs are dead once the upsampler has read them.
      CHECK(run_mrf_small(ctx, w, hm->mrf[i], hm->arena, xu, buf[0], buf[2], bs, ldo, d_frames, mul, B, Lout, voc_host_len, s));
      cur[0] = buf[0];
      cur[1] = buf[2];
      ncur = 2;
      cur_div = (float)nk;
      Lin = Lout;
      ldin = ldo;
      continue;
    }
    if (concurrent) {
      HIPCHECK(hipEventRecord(w->ev_fork, s));
      for (int j = 1; j < nk; ++j) HIPCHECK(hipStreamWaitEvent(w->aux[j - 1], w->ev_fork, 0));
    }
    // per-chain buffers and running input (MRF: resblocks on the same input, models.py:191-197)
    float* outs[MI355TTS_MAX_STAGES] = {nullptr};
    struct Chain {
      float *tb, *pa, *pb, *dst_last;
      const float* rin;
      hipStream_t st;
    } chn[MI355TTS_MAX_STAGES];
    for (int j = 0; j < nk; ++j) {
      Chain& c = chn[j];
      if (split_out) {
        // per-chain scratch: buf[2 + 4j .. 2 + 4j + 3] = {t, ping, out(flip 0), out(flip 1)}
        c.tb = buf[2 + 4 * j];
        c.pa = buf[2 + 4 * j + 1];
        c.pb = buf[2 + 4 * j + 2 + (flip ^ 1)];  // last stage's output: dead once the upsampler (before the fork) has read it
        c.dst_last = buf[2 + 4 * j + 2 + flip];
      } else {
        c.tb = buf[2];
        c.pa = buf[3];
        c.pb = buf[4];
        c.dst_last = buf[5];
      }
      c.rin = xu;
      c.st = (concurrent && j > 0) ? w->aux[j - 1] : s;
      outs[j] = c.dst_last;
    }
    const int nd = h.num_dilations;
    // One dilation step of chain j, planned: the fused pair if its geometry is covered, else conv1 (+ conv2)
    struct Step {
      PairPlan pair;
      ConvPlan c1, c2;
      float* dst;
    };
    auto plan_step = [&](int j, int d, Step& sp) -> int {
      const HifiResConv& rc = hm->rb[i][j][d];
      const int kk = h.resblock_kernel_sizes[j];
      Chain& c = chn[j];
      const bool last = d == nd - 1;
      sp.dst = last ? c.dst_last : ((d & 1) ? c.pb : c.pa);
      if (!sp.dst) return fail(MI355TTS_ERR_INVALID, "internal: resblock scratch aliasing");
      const bool fold = last && !split_out;  // serial form: the MRF average is folded into the chains' last epilogues
      sp.pair.ok = false;
      if (h.resblock_type == 1) {  // ResBlock1.forward, models.py:91-98
        plan_pair(rc.c1, rc.c2, c.rin, sp.dst, bs, ldo, d_frames, mul, rc.dil, fold ? inv_nk : 1.0f, fold ? (j > 0) : 0, B, Lout,
                  voc_host_len, &sp.pair, prec);
        if (sp.pair.ok) return 0;
        ConvArgs a = base_args(c.rin, bs, ldo, d_frames, mul, c.tb, bs, ldo, d_frames, mul, rc.dil, (kk * rc.dil - rc.dil) / 2);
        a.in_slope = 0.1f;
        CHECK(plan_conv(rc.c1, a, EPI_LINEAR, B, Lout, KC_RESBLOCK, rb_tiles, voc_host_len, &sp.c1, prec));
        ConvArgs c2 = base_args(c.tb, bs, ldo, d_frames, mul, sp.dst, bs, ldo, d_frames, mul, 1, (kk - 1) / 2);
        c2.in_slope = 0.1f;
        c2.res = c.rin;
        if (fold) {
          c2.alpha = inv_nk;
          c2.accum = j > 0;
        }
        CHECK(plan_conv(rc.c2, c2, EPI_LINEAR, B, Lout, KC_RESBLOCK, rb_tiles, voc_host_len, &sp.c2, prec));
      } else {  // ResBlock2.forward, models.py:136-141
        ConvArgs a = base_args(c.rin, bs, ldo, d_frames, mul, sp.dst, bs, ldo, d_frames, mul, rc.dil, (kk * rc.dil - rc.dil) / 2);
        a.in_slope = 0.1f;
        a.res = c.rin;
        if (fold) {
          a.alpha = inv_nk;
          a.accum = j > 0;
        }
        CHECK(plan_conv(rc.c1, a, EPI_LINEAR, B, Lout, KC_RESBLOCK, rb_tiles, voc_host_len, &sp.c1, prec));
        sp.c2.empty = true;
      }
      return 0;
    };
    auto run_step_alone = [&](int j, const Step& sp) -> int {
      if (sp.pair.ok) return run_pair(ctx, w, sp.pair, chn[j].st);
      CHECK(run_plan(ctx, w, sp.c1, chn[j].st));
      return run_plan(ctx, w, sp.c2, chn[j].st);
    };
    if (split_out) {
      // dilation-major: step d of every chain before step d + 1 of any — the chains have their own
      // buffers, and the same-geometry launches of a step can go out as ONE grouped launch
      for (int d = 0; d < nd; ++d) {
        Step sp[3];
        for (int j = 0; j < nk; ++j) CHECK(plan_step(j, d, sp[j]));
        if (nk == 3 && !sp[0].pair.ok && !sp[1].pair.ok && !sp[2].pair.ok) {  // tile shape of the step: before the schedule is chosen
          ConvPlan* c1s[3] = {&sp[0].c1, &sp[1].c1, &sp[2].c1};
          ConvPlan* c2s[3] = {&sp[0].c2, &sp[1].c2, &sp[2].c2};
          promote_group_plans(ctx, w, c1s, nk);
          promote_group_plans(ctx, w, c2s, nk);
        }
        bool done = false;
        if (grouped) {
          bool all_pair = true, none_pair = true;
          for (int j = 0; j < nk; ++j) {
            all_pair = all_pair && sp[j].pair.ok;
            none_pair = none_pair && !sp[j].pair.ok;
          }
          if (all_pair) {
            PairPlan pp[3] = {sp[0].pair, sp[1].pair, sp[2].pair};
            const int rc = run_pair_group(ctx, w, pp, nk, s);
            if (rc < 0) return rc;
            done = rc == 0;
          } else if (none_pair) {
            ConvPlan a[3] = {sp[0].c1, sp[1].c1, sp[2].c1};
            const int rc = run_group(ctx, w, a, nk, s);
            if (rc < 0) return rc;
            if (rc == 0) {
              if (!sp[0].c2.empty) {
                ConvPlan b2[3] = {sp[0].c2, sp[1].c2, sp[2].c2};
                const int rc2 = run_group(ctx, w, b2, nk, s);
                if (rc2 < 0) return rc2;
                if (rc2 != 0)
                  for (int j = 0; j < nk; ++j) CHECK(run_plan(ctx, w, sp[j].c2, chn[j].st));
              }
              done = true;
            }
          }
        }
        if (!done)
          for (int j = 0; j < nk; ++j) CHECK(run_step_alone(j, sp[j]));
        for (int j = 0; j < nk; ++j) chn[j].rin = sp[j].dst;
      }
    } else {
      // chain-major (the chains share their scratch planes and accumulate into one output)
      for (int j = 0; j < nk; ++j)
        for (int d = 0; d < nd; ++d) {
          Step sp;
          CHECK(plan_step(j, d, sp));
          CHECK(run_step_alone(j, sp));
          chn[j].rin = sp.dst;
        }
    }
    if (concurrent) {
      for (int j = 1; j < nk; ++j) {
        HIPCHECK(hipEventRecord(w->ev_join[j - 1], w->aux[j - 1]));
        HIPCHECK(hipStreamWaitEvent(s, w->ev_join[j - 1], 0));
      }
    }
    if (split_out) {
      for (int j = 0; j < nk; ++j) cur[j] = outs[j];
      ncur = nk;
      cur_div = (float)nk;
      flip ^= 1;
    } else {
      // serial: buf[5] holds the averaged sum; rotate it with the stage-input buffer
      std::swap(buf[5], buf[0]);
      cur[0] = buf[0];
      ncur = 1;
      cur_div = 1.0f;
    }
    Lin = Lout;
    ldin = ldo;
  }
  // Option "voc_out" (default 1): conv_post + tanh + the rows' peaks in ONE dedicated launch and the delivery of the rows in one
  // more (voc_out.h); 0 = the generic conv tile, zero_tail, absmax, to_int16 and a copy / fill per piece of every row.
  vo = !voc_out_off && ctx->voc_out.load() && hm->post_C == ch && hm->post.K == 7 && ldin % 4 == 0;
  if (prow && !vo) return fail(MI355TTS_ERR_INVALID, "internal: per-row outputs need the voc_out tail");
  if (vo) {  // x = tanh(conv_post(leaky_relu(x)))  — default slope 0.01 (models.py:198-200)
    PostArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = cur[0];
    if (ncur > 1) {
      a.x2 = cur[1];
      a.x3 = ncur > 2 ? cur[2] : nullptr;
    }
    a.in_div = cur_div;
    a.slope = 0.01f;
    a.x_bs = (long long)ch * ldin;
    a.x_ld = ldin;
    if (B == 1 && voc_host_len >= 0) { a.len = nullptr; a.len_const = voc_host_len * mul; } else { a.len = d_frames; }
    a.len_mul = mul;
    a.w = hm->arena + hm->post_w_off;
    a.bias = hm->arena + hm->post_b_off;
    a.C = ch;
    a.y = wav;
    a.y_bs = (long long)Nld;
    if (any_i16 && !denoise) {
      a.peak = reinterpret_cast<float*>(peak);
      a.peak_ld = peak_ld;
      peak_parts_ready = true;
    }
    ProfScope ps(ctx, w, KC_VOC_IO, 2.0 * (double)ch * 7 * (double)Lin * B);
    kn_hit(ctx, KN_POST_CONV);
    const dim3 pg((Lin + POST_TW - 1) / POST_TW, B);
    if (a.x3) hipLaunchKernelGGL(HIP_KERNEL_NAME(post_conv_kernel<7, 3>), pg, dim3(256), 0, s, a);
    else if (a.x2) hipLaunchKernelGGL(HIP_KERNEL_NAME(post_conv_kernel<7, 2>), pg, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(post_conv_kernel<7, 1>), pg, dim3(256), 0, s, a);
  } else {
    ConvArgs a = base_args(cur[0], (long long)ch * ldin, ldin, d_frames, mul, wav, (long long)Nld, (int)Nld, d_frames, mul, 1, 3);
    set_inputs(a);
    a.in_slope = 0.01f;
    a.out_act = ACT_TANH;
    CHECK(launch_conv(ctx, w, hm->post, a, EPI_LINEAR, B, Lin, KC_VOC_IO, nullptr, 1024, voc_host_len));
  }
  }  // !f16
  if (denoise) {  // HiFiGanVocoder.denoise (larynx/hifi_gan.py:171-179)
    ProfScope ps(ctx, w, KC_SMALL, 0);
    float* wav2 = (float*)(base + o_wav2);
    float* fbuf = (float*)(base + o_fbuf);
    hipLaunchKernelGGL(stft_denoise_kernel, dim3(Tmax, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, hm->bias_spec[f16 ? 1 : 0],
                       denoiser_strength, fbuf, Tmax, (float*)nullptr);
    hipLaunchKernelGGL(overlap_add_kernel, dim3(256, B), dim3(256), 0, s, fbuf, Tmax, d_frames, hop, wav2, (long long)Nld,
                       (long long)Nld);
    wav = wav2;
  }
  const size_t ild = Nld + (size_t)pads;  // row stride of the int16 staging buffer
  const long long rowlen = (long long)pad0 + N + call.pad_after;  // samples delivered per row (then zeros up to wav_ld)
  if (vo) {
    // ONE launch delivers the rows: to the caller's device buffers (pause before | samples | zeros up to the row stride), or to
    // the staging buffers the host copy reads (the float rows in place: zero tails behind a short row's samples)
    ProfScope ps(ctx, w, KC_SMALL, 0);
    if (any_i16 && !peak_parts_ready) {  // behind the denoiser: one peak per row, from the denoised rows
      HIPCHECK(hipMemsetAsync(peak, 0, sizeof(unsigned) * B, s));
      hipLaunchKernelGGL(absmax_kernel, dim3(128, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, peak);
    }
    WaveOutArgs o;
    std::memset(&o, 0, sizeof(o));
    o.wav = wav; o.bs = (long long)Nld; o.frames = d_frames; o.hop = hop;
    o.peak = reinterpret_cast<const float*>(peak);
    o.peak_ld = peak_parts_ready ? peak_ld : 1;
    o.peak_parts = peak_parts_ready ? 0 : 1;
    o.pad_before = pad0;
    bool any_out = false;
    if (prow) {
      o.per_row = 1;
      for (int b = 0; b < B; ++b) {
        const VocRow& r = prow[b];
        o.pad_rows[b] = r.pad_before;
        if (out_dev) {
          o.f32_rows[b] = r.wav_f32; o.f_ld_rows[b] = r.wav_ld;
          o.i16_rows[b] = r.wav_i16; o.i_ld_rows[b] = r.wav_ld;
        } else if (r.wav_i16) {
          o.i16_rows[b] = i16 + (size_t)b * ild; o.i_ld_rows[b] = (long long)ild;
        }
        any_out = any_out || o.f32_rows[b] || o.i16_rows[b];
      }
    } else if (out_dev) {
      if (wav_f32) { o.f32 = wav_f32; o.f_bs = wav_ld; o.f_ld = wav_ld; }
      if (wav_i16) { o.i16 = wav_i16; o.i_bs = wav_ld; o.i_ld = wav_ld; }
    } else {
      // (the float rows stay where they are: only a short row's tail up to the longest row is zeroed, in place)
      if (wav_i16) { o.i16 = i16; o.i_bs = (long long)ild; o.i_ld = (long long)ild; }
    }
    if (o.f32 || o.i16 || any_out) {
      kn_hit(ctx, KN_WAVE_OUT);
      hipLaunchKernelGGL(wave_out_kernel, dim3(128, B), dim3(256), 0, s, o);
    }
    if (!out_dev && any_f32 && B > 1) hipLaunchKernelGGL(zero_tail_kernel, dim3(64, B), dim3(256), 0, s, wav, (long long)Nld, (long long)Nld, d_frames, hop);
    if (out_dev) {
      HIPCHECK(mi355_sync(s));
      HIPCHECK(hipGetLastError());
      drain.ok = true;
      return 0;
    }
  } else {
    {
      ProfScope ps(ctx, w, KC_SMALL, 0);
      hipLaunchKernelGGL(zero_tail_kernel, dim3(64, B), dim3(256), 0, s, wav, (long long)Nld, (long long)Nld, d_frames, hop);
      if (wav_i16) {
        HIPCHECK(hipMemsetAsync(peak, 0, sizeof(unsigned) * B, s));
        hipLaunchKernelGGL(absmax_kernel, dim3(128, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, peak);
        hipLaunchKernelGGL(to_int16_kernel, dim3(128, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, peak, i16,
                           (long long)ild, (long long)ild, pad0);
      }
    }
    if (out_dev) {
      for (int b = 0; b < B; ++b) {
        if (wav_f32) {
          float* dst = wav_f32 + (size_t)b * wav_ld;
          if (pad0) HIPCHECK(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)pad0, s));
          HIPCHECK(hipMemcpyAsync(dst + pad0, wav + (size_t)b * Nld, sizeof(float) * (size_t)N, hipMemcpyDeviceToDevice, s));
          if (wav_ld > pad0 + N) HIPCHECK(hipMemsetAsync(dst + pad0 + N, 0, sizeof(float) * (size_t)(wav_ld - pad0 - N), s));
        }
        if (wav_i16) {
          int16_t* dst = wav_i16 + (size_t)b * wav_ld;
          HIPCHECK(hipMemcpyAsync(dst, i16 + (size_t)b * ild, sizeof(short) * (size_t)rowlen, hipMemcpyDeviceToDevice, s));
          if (wav_ld > rowlen) HIPCHECK(hipMemsetAsync(dst + rowlen, 0, sizeof(short) * (size_t)(wav_ld - rowlen), s));
        }
      }
      HIPCHECK(mi355_sync(s));
      HIPCHECK(hipGetLastError());
      drain.ok = true;
      return 0;
    }
  }
  // host outputs: device -> the worker's pinned staging (async DMA) -> the caller's (pageable)
  // buffers; a pageable destination would make every hipMemcpyAsync a blocking staged copy
  if (prow) {
    // per-row destinations: a row travels with its OWN length (its caller's buffer is sized for its own frame count)
    const size_t prl = (size_t)N + (size_t)pads;  // staging stride in samples
    const size_t f32_b = any_f32 ? sizeof(float) * (size_t)B * (size_t)N : 0;
    const size_t i16_b = any_i16 ? sizeof(short) * (size_t)B * prl : 0;
    CHECK(reserve_pinned_out(w, f32_b + i16_b));
    float* pf = (float*)w->pinned_out;
    short* pi = (short*)(w->pinned_out + f32_b);
    for (int b = 0; b < B; ++b) {
      const VocRow& r = prow[b];
      const size_t n = (size_t)mel->frames[b] * hop;
      if (r.wav_f32 && n) HIPCHECK(hipMemcpyAsync(pf + (size_t)b * N, wav + (size_t)b * Nld, sizeof(float) * n, hipMemcpyDeviceToHost, s));
      if (r.wav_i16) HIPCHECK(hipMemcpyAsync(pi + (size_t)b * prl, i16 + (size_t)b * ild, sizeof(short) * (n + r.pad_before + r.pad_after), hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(mi355_sync(s));
    HIPCHECK(hipGetLastError());
    drain.ok = true;
    for (int b = 0; b < B; ++b) {
      const VocRow& r = prow[b];
      const size_t n = (size_t)mel->frames[b] * hop, p0 = (size_t)r.pad_before, rl = n + p0 + (size_t)r.pad_after;
      if (r.wav_f32) {
        std::memset(r.wav_f32, 0, sizeof(float) * p0);
        std::memcpy(r.wav_f32 + p0, pf + (size_t)b * N, sizeof(float) * n);
        std::memset(r.wav_f32 + p0 + n, 0, sizeof(float) * ((size_t)r.wav_ld - p0 - n));
      }
      if (r.wav_i16) {
        std::memcpy(r.wav_i16, pi + (size_t)b * prl, sizeof(short) * rl);
        std::memset(r.wav_i16 + rl, 0, sizeof(int16_t) * ((size_t)r.wav_ld - rl));
      }
    }
    return 0;
  }
  const size_t f32_bytes = wav_f32 ? sizeof(float) * (size_t)B * (size_t)N : 0;
  const size_t i16_bytes = wav_i16 ? sizeof(short) * (size_t)B * (size_t)rowlen : 0;
  CHECK(reserve_pinned_out(w, f32_bytes + i16_bytes));
  float* pf = (float*)w->pinned_out;
  short* pi = (short*)(w->pinned_out + f32_bytes);
  for (int b = 0; b < B; ++b) {
    if (wav_f32) HIPCHECK(hipMemcpyAsync(pf + (size_t)b * N, wav + (size_t)b * Nld, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, s));
    if (wav_i16) HIPCHECK(hipMemcpyAsync(pi + (size_t)b * rowlen, i16 + (size_t)b * ild, sizeof(short) * (size_t)rowlen, hipMemcpyDeviceToHost, s));
  }
  HIPCHECK(mi355_sync(s));
  HIPCHECK(hipGetLastError());
  drain.ok = true;
  for (int b = 0; b < B; ++b) {
    if (wav_f32) {
      float* dst = wav_f32 + (size_t)b * wav_ld;
      std::memset(dst, 0, sizeof(float) * (size_t)pad0);
      std::memcpy(dst + pad0, pf + (size_t)b * N, sizeof(float) * (size_t)N);
      std::memset(dst + pad0 + N, 0, sizeof(float) * (size_t)(wav_ld - pad0 - N));
    }
    if (wav_i16) {
      int16_t* dst = wav_i16 + (size_t)b * wav_ld;
      std::memcpy(dst, pi + (size_t)b * rowlen, sizeof(short) * (size_t)rowlen);
      std::memset(dst + rowlen, 0, sizeof(int16_t) * (size_t)(wav_ld - rowlen));
    }
  }
  return 0;
}

static int hifigan_call(mi355tts_ctx* ctx, int vocoder, const mi355tts_mel* mel, const VocCall& call) {
  if (!ctx || !mel) return fail(MI355TTS_ERR_INVALID, "null argument");
  std::shared_ptr<HifiModel> vpin;
  CHECK(find_hifi(ctx, vocoder, &vpin));
  HifiModel* hm = vpin.get();
  CHECK(hifigan_precheck(ctx, hm, vocoder, mel->frames.data(), mel->B, mel->M, mel->max_frames, call));
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  return hifigan_run(ctx, w, hm, mel, call);
}

extern "C" int mi355tts_hifigan_infer(mi355tts_ctx* ctx, int vocoder, const mi355tts_mel* mel, float denoiser_strength,
                                      float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags) {
  VocCall c;
  c.denoiser_strength = denoiser_strength;
  c.wav_f32 = wav_f32;
  c.wav_i16 = wav_i16;
  c.wav_ld = wav_ld;
  c.flags = flags;
  return hifigan_call(ctx, vocoder, mel, c);
}

extern "C" int mi355tts_hifigan_infer_padded(mi355tts_ctx* ctx, int vocoder, const mi355tts_mel* mel, float denoiser_strength,
                                             float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags,
                                             int32_t pad_before, int32_t pad_after) {
  VocCall c;
  c.denoiser_strength = denoiser_strength;
  c.wav_f32 = wav_f32;
  c.wav_i16 = wav_i16;
  c.wav_ld = wav_ld;
  c.flags = flags;
  c.pad_before = pad_before;
  c.pad_after = pad_after;
  return hifigan_call(ctx, vocoder, mel, c);
}
