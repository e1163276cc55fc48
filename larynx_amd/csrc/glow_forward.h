// mi355tts host runtime — mi355tts_glow_infer: the GlowTTS layer schedule (glow_tts/models.py:118-140, :191-209, :308-354)
// (one translation unit: included once by mi355tts.hip, after the kernel headers)
#pragma once

// ------------------------------------------------------------------ GlowTTS forward
static int run_layernorm(Worker* w, const float* x, const float* res, const float* g, const float* b, float* y, int C,
                         long long bs, int ld, const int* len, int B, int Pmax, int post_relu) {
  if (C <= 256)
    hipLaunchKernelGGL(layernorm16_kernel, dim3((Pmax + 15) / 16, B), dim3(256), 0, w->stream, x, res, g, b, y, C, bs, ld, len, 0,
                       post_relu, 1e-4f);
  else
    hipLaunchKernelGGL(layernorm_kernel, dim3((Pmax + 63) / 64, B), dim3(256), 0, w->stream, x, res, g, b, y, C, bs, ld, len, 0,
                       post_relu, 1e-4f);
  return 0;
}

// ---- column-owner launches (coltile.h).  Each returns 1 when the shape is not one the kernel takes (the caller then runs
// the separate launches), 0 when launched.
static bool glow_fuse_on(const Worker* w) {  // the option as the call saw it at its start (glow_run)
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_GLOW_FUSE"); return e && std::atoi(e) != 0; }();
  return !off && w->o_glow_fuse;
}
static int run_oproj_ln(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const GlowLayer& L, const float* att, float* x, int H,
                        long long bs, int ld, const int* d_len, int host_len, int B, int Pmax) {
  if (!glow_fuse_on(w) || !L.o16.ok || H > COL_MAXROWS || ld % 4 || Pmax <= 0) return 1;
  const float* A = gm->arena;
  OprojLnArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = att; a.res = x; a.y = x; a.bs = bs; a.ld = ld;
  if (B == 1 && host_len >= 0) { a.len = nullptr; a.len_const = host_len; } else { a.len = d_len; }
  a.len_mul = 1;
  a.w = A + L.o16.w_off; a.b = A + L.o16.b_off; a.gamma = A + L.g1; a.beta = A + L.b1;
  a.H = H; a.eps = 1e-4f;
  ProfScope ps(ctx, w, KC_GLOW_ENC_CONV, 2.0 * (double)H * H * (double)Pmax * B);
  kn_hit(ctx, KN_OPROJ_LN);
  hipLaunchKernelGGL(oproj_ln_kernel, dim3((Pmax + COL_T - 1) / COL_T, B), dim3(512), 0, w->stream, a);
  return 0;
}
// the tail of block `Bk` and the start of `next` (nullptr after the last block in reverse order)
static int run_glow_tail(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const GlowBlock& Bk, const GlowBlock* next, const float* acts,
                         const float* skip, float* hbuf, long long bsD, float* z, long long bsZ, int F2, const int* d_f2, int host_len,
                         int B, int F2max) {
  const mi355tts_glow_hparams& h = gm->hp;
  const int H = h.hidden_channels, half = h.mel_channels * h.n_sqz / 2;
  if (!glow_fuse_on(w) || !Bk.t_rs.ok || !Bk.t_end.ok || !Bk.t_st.ok || (next && !next->t_st.ok) || h.n_split != 4 || (half % 2) || F2 % 4 ||
      F2max <= 0)
    return 1;
  const float* A = gm->arena;
  GlowTailArgs a;
  std::memset(&a, 0, sizeof(a));
  a.acts = acts; a.skip = h.n_block_layers > 1 ? skip : nullptr; a.hnext = next ? hbuf : nullptr; a.h_bs = bsD; a.h_ld = F2;
  a.z = z; a.z_bs = bsZ; a.z_ld = F2;
  if (B == 1 && host_len >= 0) { a.len = nullptr; a.len_const = host_len; } else { a.len = d_f2; }
  a.len_mul = 1;
  a.w_rs = A + Bk.t_rs.w_off; a.b_rs = A + Bk.t_rs.b_off;
  a.w_end = A + Bk.t_end.w_off; a.b_end = A + Bk.t_end.b_off;
  const GlowBlock& stb = next ? *next : Bk;  // the last block has no successor: its own start stands in (loaded, never used)
  a.w_st = A + stb.t_st.w_off; a.b_st = A + stb.t_st.b_off;
  a.mix_w = A + Bk.winv; a.mix_bias = A + Bk.an_bias; a.mix_scale = A + Bk.an_scale;
  a.H = H; a.half = half;
  const double mac = (double)H * H + 2.0 * half * H + (next ? (double)H * half : 0.0);
  ProfScope ps(ctx, w, KC_GLOW_DEC_CONV, 2.0 * mac * (double)F2max * B);
  kn_hit(ctx, KN_GLOW_TAIL);
  hipLaunchKernelGGL(glow_tail_kernel, dim3((F2max + COL_T - 1) / COL_T, B), dim3(512), 0, w->stream, a);
  return 0;
}

// the fp16 mode's WaveNet of block `Bk` (wn_f16.h): every layer's gate conv and res_skip but the last res_skip in ONE launch;
// leaves `acts` (last layer's gated activations) and `skip` (layers 0 .. n - 2) as the f32 chain would.  1 = not taken.
static int run_wn_f16(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const GlowBlock& Bk, const float* hcur, float* acts, float* skip,
                      long long bsD, int F2, const int* d_f2, int host_len, int B, int F2max) {
  const mi355tts_glow_hparams& h = gm->hp;
  const int H = h.hidden_channels, n = h.n_block_layers;
  if (!gm->f16_ok || (int)Bk.h_in.size() != n || (int)Bk.h_rs.size() != n - 1 || F2max <= 0 || (H != 192 && H != 32)) return 1;
  WnF16Args a;
  std::memset(&a, 0, sizeof(a));
  a.h = hcur; a.bs = bsD; a.ld = F2;
  if (B == 1 && host_len >= 0) { a.len = nullptr; a.len_const = host_len; } else { a.len = d_f2; }
  for (int j = 0; j < n; ++j) {
    a.w_in[j] = Bk.h_in[j].w; a.b_in[j] = Bk.h_in[j].bias;
    if (j < n - 1) { a.w_rs[j] = Bk.h_rs[j].w; a.b_rs[j] = Bk.h_rs[j].bias; }
  }
  a.n_layers = n;
  a.margin = (h.kernel_size_dec - 1) / 2 * n;
  a.acts = acts; a.skip = skip;
  const int to = WN_W - 2 * a.margin;
  const dim3 grid((F2max + to - 1) / to, B);
  const double mac = (double)n * 2.0 * H * H * h.kernel_size_dec + (double)(n - 1) * 2.0 * H * H;
  ProfScope ps(ctx, w, KC_GLOW_DEC_CONV, 2.0 * mac * (double)F2max * B);
  // MI355TTS_WN_REPEAT (probe): the launch N times — it is idempotent; run 2 .. N find the block's weights in L2
  static const int repeat = [] { const char* e = std::getenv("MI355TTS_WN_REPEAT"); return e ? std::max(1, std::atoi(e)) : 1; }();
  for (int r = 0; r < repeat; ++r) {
    if (H == 192)
      hipLaunchKernelGGL((wn_f16_kernel<5, 24, 3, 10>), grid, dim3(256), 0, w->stream, a);
    else
      hipLaunchKernelGGL((wn_f16_kernel<5, 4, 1, 6>), grid, dim3(256), 0, w->stream, a);
  }
  kn_hit(ctx, KN_WN_F16);
  return 0;
}

// LayerNorm -> [ReLU] -> conv with the norm inside the conv launch (lin16_kernel's LN prologue) when the shape has one;
// otherwise the LayerNorm launch (raw -> normed) and the conv on its output.  `normed` may be nullptr when nothing else
// reads the normalised tensor AND the fused form is taken; the fallback needs a buffer: `scratch`.
static int launch_ln_conv(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const DevConv& c, ConvArgs a, const float* raw, float* normed,
                          float* scratch, const float* gamma, const float* beta, int relu, int C, long long bs, int ld, const int* d_len,
                          int B, int Pmax, int glow_tiles, int host_len, bool solo_tiles);

// an encoder conv: the 16-row tile with the input staged once when the shape has one (lin16_kernel), else the generic tile
static int launch_enc_conv(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const DevConv& c, const ConvArgs& a, int B, int Pmax,
                           int glow_tiles, int host_len, bool solo_tiles = false) {
  if (run_lin16(ctx, w, c, a, gm->arena, B, Pmax, KC_GLOW_ENC_CONV, host_len, solo_tiles) == 0) return 0;
  return launch_conv(ctx, w, c, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, glow_tiles, host_len);
}

static int launch_ln_conv(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const DevConv& c, ConvArgs a, const float* raw, float* normed,
                          float* scratch, const float* gamma, const float* beta, int relu, int C, long long bs, int ld, const int* d_len,
                          int B, int Pmax, int glow_tiles, int host_len, bool solo_tiles) {
  static const bool no_ln = [] { const char* e = std::getenv("MI355TTS_LIN16_NO_LN"); return e && std::atoi(e) != 0; }();
  if (!no_ln && glow_fuse_on(w)) {
    Lin16Ln ln{gamma, beta, relu, normed};
    a.x = raw;
    if (run_lin16(ctx, w, c, a, gm->arena, B, Pmax, KC_GLOW_ENC_CONV, host_len, solo_tiles, &ln) == 0) return 0;
  }
  float* dst = normed ? normed : scratch;
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    run_layernorm(w, raw, nullptr, gamma, beta, dst, C, bs, ld, d_len, B, Pmax, relu);
  }
  a.x = dst;
  return launch_enc_conv(ctx, w, gm, c, a, B, Pmax, glow_tiles, host_len, solo_tiles);
}

struct GlowCall {
  const int64_t* ids = nullptr;
  const int32_t* id_lens = nullptr;
  int B = 0, ids_ld = 0;
  float noise_scale = 0.667f, length_scale = 1.0f;
  const float* noise = nullptr;
  int noise_ld = 0;
  uint64_t seed = 0;
  const uint64_t* row_seeds = nullptr;  // optional, host, [B]: the noise stream of row b (default seed + b)
  const int32_t* speaker_ids = nullptr;  // host, [B]: multi-speaker voices only (required there, forbidden otherwise)
  // optional, host, [B]: row b's ids live at row_ids[b] (id_lens[b] of them; host or device per `flags`) instead of
  // ids + b * ids_ld — the rows of a coalesced pass come from different callers (host_join.h)
  const int64_t* const* row_ids = nullptr;
  // every launch on the smallest tile whatever the batch size: what a batch-1 call uses up to ~5000 decoder columns, so
  // a row of a coalesced pass is computed by exactly the launches (and summation orders) of its own batch-1 call
  bool solo_tiles = false;
  const mi355tts_audio_settings* audio = nullptr;
  uint32_t flags = 0;
};

// pins the model: the caller's shared_ptr keeps it alive until the call returns, whatever mi355tts_unload does meanwhile
static int find_glow(mi355tts_ctx* ctx, int glow, std::shared_ptr<GlowModel>* out) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->glow.find(glow);
  if (it == ctx->glow.end()) return fail(MI355TTS_ERR_NO_MODEL, "no GlowTTS model %d", glow);
  *out = it->second;
  return 0;
}

static int glow_precheck(const GlowModel* gm, const GlowCall& c, int* Pmax_out) {
  if ((!c.ids && !c.row_ids) || !c.id_lens) return fail(MI355TTS_ERR_INVALID, "null argument");
  if (c.B <= 0 || c.ids_ld <= 0) return fail(MI355TTS_ERR_INVALID, "empty batch");
  // the reference fails on either mismatch (no emb_g to index / a duration predictor built for hidden + gin channels)
  if (gm->gin() && !c.speaker_ids)
    return fail(MI355TTS_ERR_INVALID, "this voice has %d speakers: pass speaker ids (mi355tts_glow_infer_speakers / mi355tts_synthesize_speakers)",
                gm->hp.n_speakers);
  if (!gm->gin() && c.speaker_ids) return fail(MI355TTS_ERR_INVALID, "speaker ids given to a single-speaker voice");
  if (c.speaker_ids)
    for (int b = 0; b < c.B; ++b)
      if (c.speaker_ids[b] < 0 || c.speaker_ids[b] >= gm->hp.n_speakers)  // nn.Embedding raises on an out-of-range index
        return fail(MI355TTS_ERR_INVALID, "speaker_ids[%d]=%d outside [0,%d)", b, c.speaker_ids[b], gm->hp.n_speakers);
  int Pmax = 0;
  for (int b = 0; b < c.B; ++b) {
    if (c.id_lens[b] < 1 || c.id_lens[b] > c.ids_ld)
      return fail(MI355TTS_ERR_INVALID, "id_lens[%d]=%d outside [1,%d]", b, c.id_lens[b], c.ids_ld);
    Pmax = std::max(Pmax, c.id_lens[b]);
  }
  if (!(c.flags & MI355TTS_IN_DEVICE)) {
    // the reference's embedding lookup raises on an out-of-range id (glow_tts/models.py:119);
    // device-resident ids cannot be checked without a sync and are clamped by the kernel instead
    for (int b = 0; b < c.B; ++b)
      for (int t = 0; t < c.id_lens[b]; ++t) {
        const int64_t id = c.row_ids ? c.row_ids[b][t] : c.ids[(size_t)b * c.ids_ld + t];
        if (id < 0 || id >= gm->hp.num_symbols)
          return fail(MI355TTS_ERR_INVALID, "phoneme id %lld at [%d][%d] outside [0,%d)", (long long)id, b, t, gm->hp.num_symbols);
      }
  }
  *Pmax_out = Pmax;
  return 0;
}

// Encoder workspace of one call: ONE definition for the forward pass and mi355tts_reserve.
struct GlowEncLayout {
  size_t o_len, o_seed, o_ids, o_x, o_t1, o_t2, o_qkv, o_ffn, o_xm, o_logw, o_cum, o_sc, o_spk, o_cond, o_dps, total;
  int P, att_rows;
};
static GlowEncLayout glow_enc_layout(const mi355tts_glow_hparams& h, int B, int ids_ld, int Pmax) {
  GlowEncLayout L;
  const int H = h.hidden_channels, Fc = h.filter_channels, Fd = h.filter_channels_dp, M = h.mel_channels;
  L.P = (Pmax + 3) & ~3;
  const int P = L.P;
  Carver cv;
  L.o_len = cv.take(sizeof(int) * B);
  L.o_seed = cv.take(sizeof(unsigned long long) * B);
  L.o_ids = cv.take(sizeof(long long) * (size_t)B * ids_ld);
  L.o_x = cv.take(sizeof(float) * (size_t)B * H * P);
  L.o_t1 = cv.take(sizeof(float) * (size_t)B * H * P);
  L.o_t2 = cv.take(sizeof(float) * (size_t)B * H * P);
  L.o_qkv = cv.take(sizeof(float) * (size_t)B * 3 * H * P);
  L.o_ffn = cv.take(sizeof(float) * (size_t)B * std::max(Fc, 3 * Fd) * P);
  L.o_xm = cv.take(sizeof(float) * (size_t)B * M * P);
  L.o_logw = cv.take(sizeof(float) * (size_t)B * P);
  L.o_cum = cv.take(sizeof(int) * (size_t)B * P);
  L.att_rows = ((Pmax + ATT_ROWS - 1) / ATT_ROWS) * ATT_ROWS;
  // score scratch: only the VALU attention fallback (P > ATTM_MAXP) uses it
  L.o_sc = cv.take(Pmax > ATTM_MAXP ? sizeof(float) * (size_t)B * h.n_heads * L.att_rows * P : 0);
  // multi-speaker voices: speaker ids, the decoder's gate offsets [B][n_blocks][2H n_layers] (live until the last flow block:
  // the encoder region is kept when the decoder's is appended) and the duration predictor's per-tap speaker sums [B][Fd][k]
  const bool spk = h.n_speakers > 1;
  L.o_spk = cv.take(spk ? sizeof(int) * B : 0);
  L.o_cond = cv.take(spk ? sizeof(float) * (size_t)B * h.n_blocks_dec * 2 * H * h.n_block_layers : 0);
  L.o_dps = cv.take(spk ? sizeof(float) * (size_t)B * Fd * h.kernel_size : 0);
  L.total = cv.pos;
  return L;
}
struct GlowDecLayout {
  size_t o_z, o_h, o_ac, o_sk, o_nz, total;
};
static GlowDecLayout glow_dec_layout(const mi355tts_glow_hparams& h, size_t enc_bytes, int B, int Fmax, size_t host_noise_floats) {
  GlowDecLayout L;
  const int H = h.hidden_channels, C = h.mel_channels * h.n_sqz;
  const int F2 = (Fmax / h.n_sqz + 3) & ~3;
  Carver dv;
  dv.pos = enc_bytes;
  L.o_z = dv.take(sizeof(float) * (size_t)B * C * F2);
  L.o_h = dv.take(sizeof(float) * (size_t)B * H * F2);
  L.o_ac = dv.take(sizeof(float) * (size_t)B * H * F2);
  L.o_sk = dv.take(sizeof(float) * (size_t)B * H * F2);
  L.o_nz = dv.take(sizeof(float) * host_noise_floats);
  L.total = dv.pos;
  return L;
}

// The forward pass on worker `w`.  With `final_sync` false the mel object is returned while
// its last kernels are still queued on w->stream (the fused synthesize path launches the
// vocoder behind them on the same stream).
static int glow_run(mi355tts_ctx* ctx, Worker* w, const GlowModel* gm, const GlowCall& call, int Pmax, bool final_sync,
                    mi355tts_mel** out) {
  const mi355tts_glow_hparams& h = gm->hp;
  const int64_t* ids = call.ids;
  const int32_t* id_lens = call.id_lens;
  const int B = call.B, ids_ld = call.ids_ld;
  const float noise_scale = call.noise_scale, length_scale = call.length_scale;
  const float* noise = call.noise;
  const int noise_ld = call.noise_ld;
  const uint64_t seed = call.seed;
  const mi355tts_audio_settings* audio = call.audio;
  const uint32_t flags = call.flags;
  hipStream_t s = w->stream;
  w->o_glow_fuse = ctx->glow_fuse.load();  // one read per call: the launch helpers below use the snapshot
  w->o_gate16 = ctx->gate16.load();
  w->o_gate16_wide = ctx->gate16_wide.load();
  // the `half` switch as this call saw it at its start: the decoder's WaveNets in fp16 (wn_f16.h)
  const bool glow_f16 = gm->f16_ok && gm->precision.load() == MI355TTS_PRECISION_F16;
  const float* A = gm->arena;
  const int H = h.hidden_channels, Fc = h.filter_channels, Fd = h.filter_channels_dp, M = h.mel_channels;
  const int k = h.kernel_size, nh = h.n_heads;
  const bool in_dev = (flags & MI355TTS_IN_DEVICE) != 0;
  const int enc_host_len = B == 1 ? id_lens[0] : -1;
  // workgroup target per GlowTTS conv launch (tile-shape choice; tuning knob MI355TTS_GLOW_TILES)
  static const int glow_tiles_env = [] { const char* e = std::getenv("MI355TTS_GLOW_TILES"); return e ? std::atoi(e) : 0; }();
  const int glow_tiles = call.solo_tiles ? (1 << 30) : glow_tiles_env > 0 ? glow_tiles_env : 1024;
  {
    long long sum = 0;
    for (int b = 0; b < B; ++b) sum += id_lens[b];
    w->flop_scale = Pmax > 0 ? (double)sum / ((double)B * Pmax) : 1.0;  // ragged batch: count the rows' real ids
  }

  // ---- encoder workspace
  const GlowEncLayout el = glow_enc_layout(h, B, ids_ld, Pmax);
  const int P = el.P;  // row stride
  const int att_rows = el.att_rows;
  const size_t enc_bytes = el.total;
  const size_t o_len = el.o_len, o_xm = el.o_xm, o_cum = el.o_cum;
  CHECK(reserve(w, enc_bytes));
  char* base = w->arena;
  int* d_len = (int*)(base + el.o_len);
  long long* d_ids = (long long*)(base + el.o_ids);
  float* x = (float*)(base + el.o_x);
  float* t1 = (float*)(base + el.o_t1);
  float* t2 = (float*)(base + el.o_t2);
  float* qkv = (float*)(base + el.o_qkv);
  float* ffn = (float*)(base + el.o_ffn);
  float* xm = (float*)(base + el.o_xm);
  float* logw = (float*)(base + el.o_logw);
  int* cum = (int*)(base + el.o_cum);
  float* sc = (float*)(base + el.o_sc);

  HIPCHECK(hipMemcpyAsync(d_len, id_lens, sizeof(int) * B, hipMemcpyHostToDevice, s));
  unsigned long long* d_seeds = nullptr;
  if (call.row_seeds) {
    d_seeds = (unsigned long long*)(base + el.o_seed);
    HIPCHECK(hipMemcpyAsync(d_seeds, call.row_seeds, sizeof(unsigned long long) * B, hipMemcpyHostToDevice, s));
  }
  if (call.row_ids) {
    HIPCHECK(hipMemsetAsync(d_ids, 0, sizeof(long long) * (size_t)B * ids_ld, s));
    for (int b = 0; b < B; ++b)
      HIPCHECK(hipMemcpyAsync(d_ids + (size_t)b * ids_ld, call.row_ids[b], sizeof(long long) * (size_t)id_lens[b],
                              in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  } else {
    HIPCHECK(hipMemcpyAsync(d_ids, ids, sizeof(long long) * (size_t)B * ids_ld, in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  }

  // multi-speaker voices: everything the speaker vector feeds, once per call (small_kernels.h: speaker_cond_kernel)
  const int gin = gm->gin();
  const int n2 = 2 * H * h.n_block_layers;  // gate offsets per flow block
  float* spk_cond = gin ? (float*)(base + el.o_cond) : nullptr;
  float* spk_dps = gin ? (float*)(base + el.o_dps) : nullptr;
  if (gin) {
    int* d_spk = (int*)(base + el.o_spk);
    HIPCHECK(hipMemcpyAsync(d_spk, call.speaker_ids, sizeof(int) * B, hipMemcpyHostToDevice, s));
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(speaker_cond_kernel, dim3(h.n_blocks_dec + 1, B), dim3(256), 0, s, A + gm->emb_g, h.n_speakers, gin, d_spk,
                       A + gm->cond_w, A + gm->cond_b, h.n_blocks_dec, n2, spk_cond, A + gm->dp_wg, Fd * k, k, spk_dps);
  }
  const long long bsH = (long long)H * P;
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(embed_kernel, dim3((Pmax + 63) / 64, 8, B), dim3(256), 0, s, d_ids, ids_ld, d_len, A + gm->emb,
                       h.num_symbols, H, std::sqrt((float)H), x, bsH, P);
  }
  if (h.prenet) {
    // ConvReluNorm: conv -> LayerNorm -> ReLU (x3), then x + proj(.)  (layers.py:73-80)
    // Each LayerNorm -> ReLU runs inside the conv that consumes it (launch_ln_conv): conv_0 writes its raw output, conv_i
    // normalises conv_{i-1}'s, proj the last one's.  Raw outputs alternate between t1 and t2; qkv is the fallback's scratch.
    float* raw = t1;
    {
      ConvArgs a = base_args(x, bsH, P, d_len, 1, raw, bsH, P, d_len, 1, 1, h.prenet_kernel_size / 2);
      CHECK(launch_enc_conv(ctx, w, gm, gm->pre_conv[0], a, B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
    }
    for (int i = 1; i < h.prenet_layers; ++i) {
      float* out = raw == t1 ? t2 : t1;
      ConvArgs a = base_args(raw, bsH, P, d_len, 1, out, bsH, P, d_len, 1, 1, h.prenet_kernel_size / 2);
      CHECK(launch_ln_conv(ctx, w, gm, gm->pre_conv[i], a, raw, nullptr, qkv, A + gm->pre_g[i - 1], A + gm->pre_b[i - 1], 1, H, bsH, P, d_len,
                           B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
      raw = out;
    }
    ConvArgs a = base_args(raw, bsH, P, d_len, 1, x, bsH, P, d_len, 1, 1, 0);
    a.res = x;
    const int last = h.prenet_layers - 1;
    CHECK(launch_ln_conv(ctx, w, gm, gm->pre_proj, a, raw, nullptr, qkv, A + gm->pre_g[last], A + gm->pre_b[last], 1, H, bsH, P, d_len, B, Pmax,
                         glow_tiles, enc_host_len, call.solo_tiles));
  }
  bool ln2_pending = false;  // the previous layer's norm_layers_2 is still to be applied to t1 (the next qkv conv does it)
  for (int l = 0; l < h.n_layers_enc; ++l) {  // Encoder.forward, attentions.py:62-74
    const GlowLayer& L = gm->layers[l];
    {
      ConvArgs a = base_args(x, bsH, P, d_len, 1, qkv, 3 * bsH, P, d_len, 1, 1, 0);
      if (ln2_pending) {  // x = LayerNorm(t1) on the way in; stored too: it is the residual of this layer's conv_o and FFN
        const GlowLayer& Lp = gm->layers[l - 1];
        CHECK(launch_ln_conv(ctx, w, gm, L.qkv, a, t1, x, nullptr, A + Lp.g2, A + Lp.b2, 0, H, bsH, P, d_len, B, Pmax, glow_tiles,
                             enc_host_len, call.solo_tiles));
        ln2_pending = false;
      } else {
        CHECK(launch_enc_conv(ctx, w, gm, L.qkv, a, B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
      }
    }
    {
      ProfScope ps(ctx, w, KC_SMALL, 0);
      kn_hit(ctx, KN_ATTENTION);
      const dim3 ag((Pmax + 31) / 32, nh, B);
      const int dkh = H / nh;
      static const bool att_big = [] { const char* e = std::getenv("MI355TTS_ATT_BIG_LDS"); return e && std::atoi(e) != 0; }();  // (A/B runs)
#define ATT_LAUNCH_P(NK, X, PM)                                                                                             \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_mfma_kernel<NK, X, PM>), ag, dim3(512), 0, s, qkv, 3 * bsH, P, d_len, H, nh, \
                     h.window_size, A + L.ek, A + L.ev, t2, bsH, P)
#define ATT_LAUNCH_X(NK, X)                        \
  do {                                             \
    if (Pmax <= 256 && !att_big) ATT_LAUNCH_P(NK, X, 256); \
    else ATT_LAUNCH_P(NK, X, ATTM_MAXP);           \
  } while (0)
#define ATT_LAUNCH(NK)                                                                                                   \
  do {                                                                                                                   \
    if (dkh == 2 * NK) ATT_LAUNCH_X(NK, true);                                                                           \
    else ATT_LAUNCH_X(NK, false);                                                                                        \
  } while (0)
      if (Pmax <= ATTM_MAXP && dkh <= 32) ATT_LAUNCH(16);
      else if (Pmax <= ATTM_MAXP && dkh <= 64) ATT_LAUNCH(32);
      else if (Pmax <= ATTM_MAXP && dkh <= 96) ATT_LAUNCH(48);
      else if (Pmax <= ATTM_MAXP) ATT_LAUNCH(64);
#undef ATT_LAUNCH
#undef ATT_LAUNCH_X
#undef ATT_LAUNCH_P
      else
        hipLaunchKernelGGL(attention_kernel, dim3(att_rows / ATT_ROWS, nh, B), dim3(256), 0, s, qkv, 3 * bsH, P, d_len, H, nh,
                           h.window_size, A + L.ek, A + L.ev, t2, bsH, P, sc, P);
    }
    if (run_oproj_ln(ctx, w, gm, L, t2, x, H, bsH, P, d_len, enc_host_len, B, Pmax)) {
      ConvArgs a = base_args(t2, bsH, P, d_len, 1, t1, bsH, P, d_len, 1, 1, 0);
      a.res = x;
      CHECK(launch_conv(ctx, w, L.o, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, glow_tiles, enc_host_len));
      ProfScope ps(ctx, w, KC_SMALL, 0);
      run_layernorm(w, t1, nullptr, A + L.g1, A + L.b1, x, H, bsH, P, d_len, B, Pmax, 0);
    }
    {  // FFN, attentions.py:375-383
      ConvArgs a = base_args(x, bsH, P, d_len, 1, ffn, (long long)Fc * P, P, d_len, 1, 1, k / 2);
      a.out_act = ACT_RELU;
      CHECK(launch_enc_conv(ctx, w, gm, L.ffn1, a, B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
      ConvArgs c = base_args(ffn, (long long)Fc * P, P, d_len, 1, t1, bsH, P, d_len, 1, 1, k / 2);
      c.res = x;
      CHECK(launch_enc_conv(ctx, w, gm, L.ffn2, c, B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
      if (l + 1 < h.n_layers_enc) {
        ln2_pending = true;  // norm_layers_2 rides in the next layer's qkv conv
      } else {
        ProfScope ps(ctx, w, KC_SMALL, 0);
        run_layernorm(w, t1, nullptr, A + L.g2, A + L.b2, x, H, bsH, P, d_len, B, Pmax, 0);
      }
    }
  }
  {  // proj_m and the duration predictor (models.py:133-139, 39-49)
    ConvArgs a = base_args(x, bsH, P, d_len, 1, xm, (long long)M * P, P, d_len, 1, 1, 0);
    CHECK(launch_enc_conv(ctx, w, gm, gm->proj_m, a, B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
    float* d1 = ffn;
    float* d2 = ffn + (size_t)B * Fd * P;
    float* d3 = ffn + (size_t)2 * B * Fd * P;
    const long long bsD = (long long)Fd * P;
    ConvArgs c1 = base_args(x, bsH, P, d_len, 1, d1, bsD, P, d_len, 1, 1, k / 2);
    c1.out_act = ACT_RELU;
    if (gin) {
      // conv_1 over [x ; g repeated along time] (models.py:128-132) = conv_1's encoder half over x + a plane that depends only
      // on the speaker and on where the row's zero padding starts: d3 is free until conv_2's fallback may use it
      ProfScope ps(ctx, w, KC_SMALL, 0);
      hipLaunchKernelGGL(speaker_dp_plane_kernel, dim3((P + 255) / 256, Fd, B), dim3(256), 0, s, spk_dps, Fd, k, k / 2, d_len, d3, bsD, P);
      c1.res = d3;
    }
    CHECK(launch_enc_conv(ctx, w, gm, gm->dp1, c1, B, Pmax, glow_tiles, enc_host_len, call.solo_tiles));
    // norm_1 inside conv_2 (launch_ln_conv): d1 -> d2; d3 is the fallback's scratch
    ConvArgs c2 = base_args(d1, bsD, P, d_len, 1, d2, bsD, P, d_len, 1, 1, k / 2);
    c2.out_act = ACT_RELU;
    CHECK(launch_ln_conv(ctx, w, gm, gm->dp2, c2, d1, nullptr, d3, A + gm->dg1, A + gm->db1, 0, Fd, bsD, P, d_len, B, Pmax, glow_tiles,
                         enc_host_len, call.solo_tiles));
    if (glow_fuse_on(w) && Fd <= 256) {  // norm_2 and proj (1 x 1, Fd -> 1) in one launch
      ProfScope ps(ctx, w, KC_SMALL, 0);
      hipLaunchKernelGGL(layernorm16_kernel, dim3((Pmax + 15) / 16, B), dim3(256), 0, w->stream, d2, (const float*)nullptr,
                         A + gm->dg2, A + gm->db2, d1, Fd, bsD, P, d_len, 0, 0, 1e-4f, A + gm->dpp_w, A + gm->dpp_b, logw, (long long)P);
    } else {
      {
        ProfScope ps(ctx, w, KC_SMALL, 0);
        run_layernorm(w, d2, nullptr, A + gm->dg2, A + gm->db2, d1, Fd, bsD, P, d_len, B, Pmax, 0);
      }
      ConvArgs c3 = base_args(d1, bsD, P, d_len, 1, logw, P, P, d_len, 1, 1, 0);
      CHECK(launch_conv(ctx, w, gm->dpp, c3, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, glow_tiles, enc_host_len));
    }
  }

  // ---- durations -> frame counts (the one host sync of the path)
  mi355tts_mel* mel = nullptr;
  {
    // frames live with the result object
    auto* m = new mi355tts_mel();
    m->ctx = ctx;
    m->B = B;
    m->M = M;
    m->ld = 0;
    m->frames.assign(B, 0);
    m->frames_dev = (int*)pool_alloc(ctx, sizeof(int) * B);
    if (!m->frames_dev) {
      delete m;
      return fail(MI355TTS_ERR_NOMEM, "hipMalloc frames");
    }
    mel = m;
  }
  struct MelGuard {  // error exits: nothing queued on the stream may still write the blocks that go back to the pool
    mi355tts_mel* m;
    hipStream_t st;
    ~MelGuard() {
      if (!m) return;
      mi355_sync(st);
      mel_destroy(m);
    }
  } mguard{mel, s};
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(duration_kernel, dim3(B), dim3(64), 0, s, logw, (long long)P, d_len, length_scale, h.n_sqz, cum, P,
                       mel->frames_dev, 1 << 28);
  }
  if ((size_t)B > w->pinned_ints) return fail(MI355TTS_ERR_INVALID, "batch too large");
  HIPCHECK(hipMemcpyAsync(w->pinned, mel->frames_dev, sizeof(int) * B, hipMemcpyDeviceToHost, s));
  HIPCHECK(mi355_sync(s));
  int Fmax = 0;
  for (int b = 0; b < B; ++b) {
    mel->frames[b] = w->pinned[b];
    Fmax = std::max(Fmax, w->pinned[b]);
  }
  if (noise && noise_scale != 0.f && noise_ld < Fmax)
    return fail(MI355TTS_ERR_TOO_SMALL, "noise has %d columns but the utterance needs %d frames", noise_ld, Fmax);
  mel->max_frames = Fmax;
  {
    long long sum = 0;
    for (int b = 0; b < B; ++b) sum += mel->frames[b];
    w->flop_scale = Fmax > 0 ? (double)sum / ((double)B * Fmax) : 1.0;
  }
  const int Fld = (Fmax + 3) & ~3;
  mel->ld = Fld;
  if (Fmax == 0) {
    mguard.m = nullptr;
    *out = mel;
    return 0;
  }
  {
    const size_t n = (size_t)B * M * Fld * sizeof(float);
    mel->raw_bytes = n;
    mel->raw = (float*)pool_alloc(ctx, n);
    mel->voc = (float*)pool_alloc(ctx, n);
    if (!mel->raw || !mel->voc) return fail(MI355TTS_ERR_NOMEM, "hipMalloc mel");
  }

  // ---- decoder workspace (appended after the encoder's, which stays live)
  const int nsq = h.n_sqz;
  const int C = M * nsq, half = C / 2;
  const int F2max = Fmax / nsq;
  const int F2 = (F2max + 3) & ~3;
  const GlowDecLayout dl = glow_dec_layout(h, enc_bytes, B, Fmax, (noise && !in_dev) ? (size_t)B * M * noise_ld : 0);
  const size_t o_z = dl.o_z, o_h = dl.o_h, o_ac = dl.o_ac, o_sk = dl.o_sk, o_nz = dl.o_nz;
  if (dl.total > w->arena_bytes) {
    // growing would move the encoder buffers: stage the three still-live encoder
    // outputs (x_m, cum, len) through a fresh arena instead
    std::vector<char> keep(enc_bytes);
    HIPCHECK(hipMemcpy(keep.data(), w->arena, enc_bytes, hipMemcpyDeviceToHost));
    CHECK(reserve(w, dl.total));
    HIPCHECK(hipMemcpy(w->arena, keep.data(), enc_bytes, hipMemcpyHostToDevice));
    base = w->arena;
    d_len = (int*)(base + o_len);
    if (d_seeds) d_seeds = (unsigned long long*)(base + el.o_seed);
    xm = (float*)(base + o_xm);
    cum = (int*)(base + o_cum);
    if (gin) spk_cond = (float*)(base + el.o_cond);
  }
  float* z = (float*)(base + o_z);
  float* hbuf = (float*)(base + o_h);
  float* acts = (float*)(base + o_ac);
  float* skip = (float*)(base + o_sk);
  const float* d_noise = noise;
  if (noise && !in_dev) {
    float* nz = (float*)(base + o_nz);
    HIPCHECK(hipMemcpyAsync(nz, noise, sizeof(float) * (size_t)B * M * noise_ld, hipMemcpyHostToDevice, s));
    d_noise = nz;
  }
  const int* d_frames = mel->frames_dev;
  const long long bsZ = (long long)C * F2, bsD = (long long)H * F2;
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(expand_noise_squeeze_kernel, dim3((Fmax + 255) / 256, 8, B), dim3(256), 0, s, xm, (long long)M * P, P,
                       d_len, cum, P, d_frames, d_noise, (long long)M * noise_ld, noise_ld, noise_scale, seed, d_seeds, M, nsq,
                       z, bsZ, F2);
  }
  // frames/n_sqz is the decoder's time axis: len = frames[b] / nsq  -> use out_mul trick via a scaled length array
  // (frames are multiples of n_sqz; kernels take frames with a divisor where needed)
  const int dec_host_len = B == 1 ? mel->frames[0] / nsq : -1;
  int* d_f2 = (int*)(base + o_len);  // reuse: id lengths are no longer needed after expansion
  {
    // d_f2[b] = frames[b] / nsq, computed on the host side of the sync above
    for (int b = 0; b < B; ++b) w->pinned[b] = mel->frames[b] / nsq;
    HIPCHECK(hipMemcpyAsync(d_f2, w->pinned, sizeof(int) * B, hipMemcpyHostToDevice, s));
  }
  float* const hcur = hbuf;  // the WaveNet's hidden state
  bool start_done = false;  // the previous block's tail launch already ran this block's start conv
  for (int blk = h.n_blocks_dec - 1; blk >= 0; --blk) {  // models.py:195-206, reversed flows
    const GlowBlock& Bk = gm->blocks[blk];
    if (!start_done) {  // CouplingBlock reverse (attentions.py:119-142): h = start(x0)
      ConvArgs a = base_args(z, bsZ, F2, d_f2, 1, hcur, bsD, F2, d_f2, 1, 1, 0);
      CHECK(launch_conv(ctx, w, Bk.start, a, EPI_LINEAR, B, F2max, KC_GLOW_DEC_CONV, nullptr, glow_tiles, dec_host_len));
    }
    int dil = 1;
    bool tail_done = false;
    // the fp16 mode: layers 0 .. n - 1 up to the last gated tile in ONE launch (wn_f16.h); the loop then runs the last layer's tail
    const bool wn16 = glow_f16 && run_wn_f16(ctx, w, gm, Bk, hcur, acts, skip, bsD, F2, d_f2, dec_host_len, B, F2max) == 0;
    for (int j = wn16 ? h.n_block_layers - 1 : 0; j < h.n_block_layers; ++j) {  // WN.forward, layers.py:138-162
      const int kd = h.kernel_size_dec;
      const bool last = j == h.n_block_layers - 1;
      ConvArgs a = base_args(hcur, bsD, F2, d_f2, 1, acts, bsD, F2, d_f2, 1, dil, (kd * dil - dil) / 2);
      a.half = H;
      if (gin) {  // x_in + g_l (layers.py:144-154): this block's, this layer's [2H] slice of cond_layer(g), per batch row
        a.cond = spk_cond + (size_t)blk * n2 + (size_t)j * 2 * H;
        a.cond_bs = (long long)h.n_blocks_dec * n2;
      }
      if (B == 1 && dec_host_len >= 0) {  // the length is known on the host: no device length array to chase
        a.in_len = a.out_len = nullptr;
        a.in_const = a.out_const = dec_host_len;
      }
      if (!wn16) {
        const int g16 = run_gate16(ctx, w, Bk.in[j], a, B, F2max, KC_GLOW_DEC_CONV, s);
        if (g16 < 0) return g16;
        if (g16 == 1) CHECK(launch_conv(ctx, w, Bk.in[j], a, EPI_GATE, B, F2max, KC_GLOW_DEC_CONV, nullptr, glow_tiles, dec_host_len));
      }
      if (last) {
        // last layer: res_skip (all skip) + end + coupling + InvConvNear/ActNorm + the next block's start in ONE launch
        const GlowBlock* next = blk > 0 ? &gm->blocks[blk - 1] : nullptr;
        tail_done = run_glow_tail(ctx, w, gm, Bk, next, acts, skip, hcur, bsD, z, bsZ, F2, d_f2, dec_host_len, B, F2max) == 0;
        if (tail_done) break;
      }
      ConvArgs r = base_args(acts, bsD, F2, d_f2, 1, hcur, bsD, F2, d_f2, 1, 1, 0);
      if (j < h.n_block_layers - 1) {
        r.res = hcur;  // x = x + res_skip[:H]
        r.split = H;
      } else {
        r.split = 0;  // last layer: everything is skip
      }
      r.y2 = skip;
      r.y2_bs = bsD;
      r.y2_ld = F2;
      r.accum2 = j > 0;
      if (run_lin16(ctx, w, Bk.rs[j], r, A, B, F2max, KC_GLOW_DEC_CONV, dec_host_len, call.solo_tiles) != 0)
        CHECK(launch_conv(ctx, w, Bk.rs[j], r, EPI_LINEAR, B, F2max, KC_GLOW_DEC_CONV, nullptr, glow_tiles, dec_host_len));
      dil *= h.dilation_rate;
    }
    start_done = tail_done;
    if (tail_done) continue;
    {  // m, logs = end(wn_out);  z1 = (x1 - m) * exp(-logs)
      ConvArgs a = base_args(skip, bsD, F2, d_f2, 1, z + (size_t)half * F2, bsZ, F2, d_f2, 1, 1, 0);
      a.res = z + (size_t)half * F2;
      a.half = half;
      const bool fuse_mix = h.n_split == 4 && (half % 2) == 0;
      if (fuse_mix) {  // InvConvNear + ActNorm ride in the coupling conv's epilogue
        a.mix_x0 = z;
        a.mix_w = A + Bk.winv;
        a.mix_bias = A + Bk.an_bias;
        a.mix_scale = A + Bk.an_scale;
      }
      CHECK(launch_conv(ctx, w, Bk.end, a, EPI_COUPLING, B, F2max, KC_GLOW_DEC_CONV, nullptr, glow_tiles, dec_host_len));
      if (fuse_mix) continue;
    }
    {
      ProfScope ps(ctx, w, KC_SMALL, 0);
      hipLaunchKernelGGL(invconv_actnorm_kernel, dim3((F2max + 255) / 256, std::min(C / h.n_split, 16), B), dim3(256), 0, s, z,
                         bsZ, F2, d_f2, 1, C, h.n_split, A + Bk.winv, A + Bk.an_bias, A + Bk.an_scale);
    }
  }
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(mel_finalize_kernel, dim3((Fld + 255) / 256, std::min(M, 16), B), dim3(256), 0, s, z, bsZ, F2, d_frames, M,
                       nsq, mel->raw, mel->voc, (long long)M * Fld, Fld, to_mt(audio), audio ? 1 : 0);
  }
  if (final_sync) {
    HIPCHECK(mi355_sync(s));
    HIPCHECK(hipGetLastError());
  }
  mguard.m = nullptr;
  w->flop_scale = 1.0;
  *out = mel;
  return 0;
}

static int glow_infer_impl(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                           float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                           const uint64_t* row_seeds, const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out,
                           const int32_t* speaker_ids = nullptr) {
  if (!ctx || !out) return fail(MI355TTS_ERR_INVALID, "null argument");
  std::shared_ptr<GlowModel> gpin;
  CHECK(find_glow(ctx, glow, &gpin));
  const GlowModel* gm = gpin.get();
  GlowCall c;
  c.ids = ids;
  c.id_lens = id_lens;
  c.B = B;
  c.ids_ld = ids_ld;
  c.noise_scale = noise_scale;
  c.length_scale = length_scale;
  c.noise = noise;
  c.noise_ld = noise_ld;
  c.seed = seed;
  c.row_seeds = row_seeds;
  c.speaker_ids = speaker_ids;
  c.audio = audio;
  c.flags = flags;
  int Pmax = 0;
  CHECK(glow_precheck(gm, c, &Pmax));
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  return glow_run(ctx, w, gm, c, Pmax, true, out);
}

extern "C" int mi355tts_glow_infer(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                                   float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                                   const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out) {
  return glow_infer_impl(ctx, glow, ids, id_lens, B, ids_ld, noise_scale, length_scale, noise, noise_ld, seed, nullptr, audio, flags, out);
}
extern "C" int mi355tts_glow_infer_speakers(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                                            float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                                            const uint64_t* row_seeds, const int32_t* speaker_ids, const mi355tts_audio_settings* audio,
                                            uint32_t flags, mi355tts_mel** out) {
  if (!speaker_ids) return fail(MI355TTS_ERR_INVALID, "speaker_ids null");
  return glow_infer_impl(ctx, glow, ids, id_lens, B, ids_ld, noise_scale, length_scale, noise, noise_ld, seed, row_seeds, audio, flags, out,
                         speaker_ids);
}
extern "C" int mi355tts_glow_infer_rows(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                                        float noise_scale, float length_scale, const uint64_t* row_seeds,
                                        const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out) {
  if (!row_seeds) return fail(MI355TTS_ERR_INVALID, "row_seeds null");
  return glow_infer_impl(ctx, glow, ids, id_lens, B, ids_ld, noise_scale, length_scale, nullptr, 0, 0, row_seeds, audio, flags, out);
}
