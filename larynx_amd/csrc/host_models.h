// mi355tts host runtime — error reporting, model arenas (weights packed into MFMA fragment order), tensor manifests
// (one translation unit: included once by mi355tts.hip, after the kernel headers)
#pragma once

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHECK(expr)                                                                               \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return fail(MI355TTS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define CHECK(expr)            \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != 0) return rc_;  \
  } while (0)

// ------------------------------------------------------------------ models
struct DevConv {
  size_t w_off = 0, b_off = 0, t_off = 0;  // offsets (floats) into the model arena: fragments, biases, the int table
  size_t w8_off = 0, t8_off = 0;           // C = 8: the 4x4x1 packing (mrf8_kernel) and its table
  const float* w = nullptr;
  const float* bias = nullptr;
  int mtiles = 0, noct = 0, K = 0, rows = 0, Cin = 0, Cout = 0, MB = 1;
  bool has_bias = false;
  // split-bf16 copy of the weights (conv_bf16.h), present for the convs that mode covers
  size_t w16_off = 0;  // offset (uint16 elements) into the model's bf16 arena
  const void* w16 = nullptr;
  int mtiles16 = 0, nslab16 = 0;
  // the 16-row-tile packing of a WaveNet gate conv (gate16.h), present when its shape is one the kernel is built for
  size_t g16_w_off = 0, g16_b_off = 0;
  const float* g16_w = nullptr;
  const float* g16_b = nullptr;
  int g16_J = 0;  // 4-channel groups per k-group; 0 = no such packing
  // the 16-row-tile packing of a plain encoder conv (lin16_kernel, gate16.h), present for the shapes the kernel is built for
  size_t l16_w_off = 0, l16_b_off = 0;
  int l16_J = 0;
};

// a dense 1 x 1 conv packed for the column-owner launches (pack_col16); w_off == 0 and ok == false when absent
struct DevCol {
  size_t w_off = 0, b_off = 0;
  bool ok = false;
};

struct ArenaBuilder {
  std::vector<float> host;
  size_t add(const float* p, size_t n) {
    size_t off = (host.size() + 63) & ~(size_t)63;  // 256-byte alignment
    host.resize(off + n);
    if (n) std::memcpy(host.data() + off, p, n * sizeof(float));
    return off;
  }
  size_t add(const std::vector<float>& v) { return add(v.data(), v.size()); }
};

enum RowLayout { ROWS_PLAIN, ROWS_PAIR, ROWS_UPSAMPLE };

// Pack a logical conv weight w[Cout][Cin][K] (or, for ROWS_UPSAMPLE, the
// transposed-conv weight w[Cin][Cout][Ku]) into the arena.
static DevConv add_conv(ArenaBuilder& ab, const float* w, const float* bias, int Cout, int Cin, int K, RowLayout layout,
                        int half_or_up = 0) {
  DevConv d;
  d.Cin = Cin;
  d.Cout = Cout;
  d.has_bias = bias != nullptr;
  PackedConv p;
  if (layout == ROWS_PLAIN) {
    d.MB = Cout <= 32 ? 1 : 2;
    p = pack_conv(
        Cout, d.MB, Cin, K, [&](int v) { return v; },
        [&](int co, int ci, int k) { return w[((size_t)co * Cin + ci) * K + k]; }, [&](int co) { return bias[co]; },
        d.has_bias, 8);
  } else if (layout == ROWS_PAIR) {
    // virtual 32-row tile p = rows [16p, 16p+16) of the first half followed by the
    // same rows of the second half (see the paired epilogues of conv_mfma_kernel)
    const int half = half_or_up;
    const int ptiles = (half + 15) / 16;
    d.MB = 1;
    p = pack_conv(
        ptiles * 32, 1, Cin, K,
        [&](int v) {
          const int tile = v / 32, i = v % 32;
          const int c = tile * 16 + (i & 15);
          if (c >= half) return -1;
          return (i >> 4) * half + c;
        },
        [&](int co, int ci, int k) { return w[((size_t)co * Cin + ci) * K + k]; }, [&](int co) { return bias[co]; },
        d.has_bias, 8);
  } else {
    // ConvTranspose1d(Cin, Cout, Ku, stride u, padding (Ku-u)/2) as a Kt = Ku/u tap
    // conv over q with virtual rows v = co*u + r:
    //   out[co][q*u + r - p] = sum_ci sum_m x[ci][q - m] * Wt[ci][co][m*u + r]
    // tap k reads x[q + k - (Kt-1)], i.e. m = Kt-1-k.
    const int u = half_or_up;
    const int Ku = K;  // caller passes the transposed kernel size in K
    const int Kt = Ku / u;
    d.MB = (Cout * u) <= 32 ? 1 : 2;
    p = pack_conv(
        Cout * u, d.MB, Cin, Kt, [&](int v) { return v; },
        [&](int v, int ci, int k) {
          const int co = v / u, r = v % u;
          const int m = Kt - 1 - k;
          return w[((size_t)ci * Cout + co) * Ku + m * u + r];
        },
        [&](int v) { return bias[v / u]; }, d.has_bias, 8);
  }
  d.mtiles = p.mtiles;
  d.noct = p.noct;
  d.K = p.K;
  d.rows = p.rows;
  d.w_off = ab.add(p.w);
  if (d.has_bias) d.b_off = ab.add(p.bias);
  return d;
}

// gate16.h instantiations: taps x channel groups per k-group (Cin <= 32 J)
static bool gate16_shape_ok(int K, int Cin) {
  const int J = (Cin + 31) / 32;
  return (K == 3 || K == 5) && (J == 1 || J == 2 || J == 3 || J == 4 || J == 6 || J == 8);
}
// the second packing of a WaveNet gate conv w[2*half][Cin][K] (add_conv(..., ROWS_PAIR, half) made the first)
static void add_gate16(ArenaBuilder& ab, DevConv& d, const float* w, const float* bias, int half, int Cin, int K) {
  if (!gate16_shape_ok(K, Cin)) return;
  PackedGate16 p = pack_gate16(
      half, Cin, K, [&](int co, int ci, int k) { return w[((size_t)co * Cin + ci) * K + k]; }, [&](int co) { return bias[co]; },
      bias != nullptr);
  d.g16_J = p.J;
  d.g16_w_off = ab.add(p.w);
  d.g16_b_off = ab.add(p.bias);
}

// the column-owner packing of a 1 x 1 conv w[rows][K] (coltile.h takes up to COL_MAXROWS rows / K-depth)
static DevCol add_col16(ArenaBuilder& ab, const float* w, const float* bias, int rows, int K) {
  DevCol d;
  if (rows > COL_MAXROWS || K > COL_MAXROWS) return d;
  PackedCol16 p = pack_col16(
      rows, K, [&](int r, int k) { return w[(size_t)r * K + k]; }, [&](int r) { return bias[r]; }, bias != nullptr);
  d.w_off = ab.add(p.w);
  d.b_off = ab.add(p.bias);
  d.ok = true;
  return d;
}

// lin16_kernel instantiations: (taps, channel groups per k-group)
static bool lin16_shape_ok(int K, int Cin) {
  const int J = (Cin + 31) / 32;
  return (K == 3 && (J == 6 || J == 8 || J == 24)) || (K == 5 && J == 6) || (K == 1 && J == 6);
}
// the second packing of a plain conv w[Cout][Cin][K] (add_conv(..., ROWS_PLAIN) made the first)
static void add_lin16(ArenaBuilder& ab, DevConv& d, const float* w, const float* bias, int Cout, int Cin, int K) {
  if (!lin16_shape_ok(K, Cin)) return;
  PackedGate16 p = pack_lin16(
      Cout, Cin, K, [&](int co, int ci, int k) { return w[((size_t)co * Cin + ci) * K + k]; }, [&](int co) { return bias[co]; },
      bias != nullptr);
  d.l16_J = p.J;
  d.l16_w_off = ab.add(p.w);
  d.l16_b_off = ab.add(p.bias);
}

struct Blob {
  const float* p;
  int64_t n;
  int64_t pos = 0;
  std::vector<std::pair<std::string, int64_t>> manifest;
  size_t idx = 0;
  const float* take(const char* name, int64_t numel) {
    if (idx >= manifest.size() || manifest[idx].first != name || manifest[idx].second != numel || pos + numel > n) {
      fail(MI355TTS_ERR_INVALID, "weight blob does not match manifest at '%s'", name);
      return nullptr;
    }
    const float* r = p + pos;
    pos += numel;
    idx++;
    return r;
  }
};

// one conv of the fp16 modes (conv_f16.h, wn_f16.h): fp16 A fragments in the model's fp16 arena, f32 bias in the float arena
struct HConvW {
  size_t w_off = 0, b_off = 0;  // uint16 elements into arenaH; floats into the model arena
  const uint4* w = nullptr;
  const float* bias = nullptr;
  int mtiles = 0, nslab = 0, K = 0, rows = 0, Cin = 0;
};
struct HResConv {
  HConvW c1, c2;
};
struct GlowLayer {
  DevConv qkv, o, ffn1, ffn2;
  size_t ek, ev, g1, b1, g2, b2;
  DevCol o16;  // conv_o once more, packed for oproj_ln_kernel (coltile.h)
};
struct GlowBlock {
  DevConv start, end;
  std::vector<DevConv> in, rs;
  size_t winv, an_bias, an_scale;
  // the column-owner packings of glow_tail_kernel (coltile.h): res_skip_layers[last], end (rows in natural order), start
  DevCol t_rs, t_end, t_st;
  // fp16 packings of the WaveNet for wn_f16_kernel: in_layers (rows paired per 32-row tile), res_skip_layers[0 .. n - 2]
  std::vector<HConvW> h_in, h_rs;
};
// Models are handed out as shared_ptr pins: a call keeps its models alive for its whole duration, mi355tts_unload only
// drops the context's reference, and the device memory goes when the last call that uses the model has returned.
// RAII: make `device` current for a scope, then put the caller's device back (a destructor that runs on whichever thread
// drops the last reference must not change that thread's current device)
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) hipSetDevice(device);
    else prev = -1;
  }
  ~DeviceScope() {
    if (prev >= 0) hipSetDevice(prev);
  }
};

struct GlowModel {
  mi355tts_glow_hparams hp;
  int device = 0;
  float* arena = nullptr;
  // the `half` switch: MI355TTS_PRECISION_F16 = the decoder's WaveNets in fp16 (wn_f16.h) when the geometry is covered
  // (f16_ok; f16_why says what is not); everything else of the acoustic model computes in f32 in every mode
  std::atomic<int> precision{0};
  uint16_t* arenaH = nullptr;
  bool f16_ok = false;
  std::string f16_why;
  ~GlowModel() {
    DeviceScope ds(device);
    if (arena) hipFree(arena);
    if (arenaH) hipFree(arenaH);
  }
  size_t emb;
  std::vector<DevConv> pre_conv;
  std::vector<size_t> pre_g, pre_b;
  DevConv pre_proj;
  std::vector<GlowLayer> layers;
  DevConv proj_m, dp1, dp2, dpp;
  size_t dpp_w = 0, dpp_b = 0;  // proj's plain weight row [Fd] and bias, for the LayerNorm kernel's fused projection
  size_t dg1, db1, dg2, db2;
  std::vector<GlowBlock> blocks;
  // multi-speaker voices (hp.n_speakers > 1): emb_g [n_speakers][gin]; the speaker half of proj_w.conv_1's weight
  // [Fd][gin][k]; the flow blocks' cond_layer weights [n_blocks][2H n_layers][gin] and biases [n_blocks][2H n_layers]
  size_t emb_g = 0, dp_wg = 0, cond_w = 0, cond_b = 0;
  int gin() const { return hp.n_speakers > 1 ? hp.gin_channels : 0; }
};
struct HifiResConv {
  DevConv c1, c2;
  int dil;
};
// A narrow stage (C = 8 / 16) packed for the one-launch MRF kernel (mrf_small.h)
struct MrfStage {
  bool ok = false;
  int C = 0, nsteps = 0;
  size_t w_off = 0, b_off = 0, t_off = 0;  // offsets (floats) into the model arena: fragments, biases, the int table
  size_t w8_off = 0, t8_off = 0;           // C = 8: the 4x4x1 packing (mrf8_kernel) and its table
  int woff[3][MRF_MAX_STEPS][2] = {};
  int dil[3][MRF_MAX_STEPS] = {};
  double mac_per_col = 0;  // algorithmic MACs per output column (all 18 convs)
};
struct HifiModel {
  mi355tts_hifigan_hparams hp;
  int device = 0;
  float* arena = nullptr;
  uint16_t* arena16 = nullptr;  // split-bf16 weight fragments of the ResBlock convs
  // the native fp16 mode (MI355TTS_PRECISION_F16): every conv of the generator packed for conv_f16.h; f16_ok = the model's
  // geometry is one the fp16 tiles cover (f16_why says what is not)
  uint16_t* arenaH = nullptr;
  bool f16_ok = false;
  std::string f16_why;
  HConvW h_pre;
  std::vector<HConvW> h_ups;
  std::vector<std::vector<std::vector<HResConv>>> h_rb;  // [stage][kernel][dilation index]
  // 0 = exact f32 MFMA, 1 = split-bf16 (3 x bf16 MFMA) for the wide ResBlock convs, 2 = the same with one bf16 MFMA,
  // 3 = native fp16 (fp16 planes, one fp16 MFMA per product, the whole generator)
  std::atomic<int> precision{0};
  DevConv pre, post;
  size_t post_w_off = 0, post_b_off = 0;  // conv_post's raw [C][7] weight and bias (post_conv_kernel)
  int post_C = 0;
  std::vector<DevConv> ups;
  // [stage][kernel][dilation index]
  std::vector<std::vector<std::vector<HifiResConv>>> rb;
  std::vector<MrfStage> mrf;  // per stage
  int hop = 1;
  // denoiser bias spectrum |STFT(generator(zeros))|[:, 0] (larynx/hifi_gan.py:181-203), built on first use
  // one per arithmetic: [0] the f32-plane modes, [1] the fp16 mode (the reference derives the bias from the model it runs,
  // half or not: larynx/hifi_gan.py:181-203)
  std::mutex bias_mu;
  float* bias_spec[2] = {nullptr, nullptr};
  bool bias_ready[2] = {false, false};
  ~HifiModel() {
    DeviceScope ds(device);
    if (arena) hipFree(arena);
    if (arena16) hipFree(arena16);
    if (arenaH) hipFree(arenaH);
    for (float* b : bias_spec)
      if (b) hipFree(b);
  }
};

static std::vector<std::pair<std::string, int64_t>> glow_manifest(const mi355tts_glow_hparams& h) {
  std::vector<std::pair<std::string, int64_t>> m;
  auto add = [&](const std::string& n, int64_t e) { m.emplace_back(n, e); };
  const int64_t H = h.hidden_channels, Fc = h.filter_channels, Fd = h.filter_channels_dp, M = h.mel_channels;
  const int64_t k = h.kernel_size, dk = H / std::max(1, h.n_heads), nrel = 2 * h.window_size + 1;
  const int64_t gin = h.n_speakers > 1 ? h.gin_channels : 0;
  add("encoder.emb.weight", (int64_t)h.num_symbols * H);
  if (gin) add("emb_g.weight", (int64_t)h.n_speakers * gin);
  if (h.prenet) {
    for (int i = 0; i < h.prenet_layers; ++i) {
      std::string p = "encoder.pre.conv_layers." + std::to_string(i);
      add(p + ".weight", H * H * h.prenet_kernel_size);
      add(p + ".bias", H);
      std::string q = "encoder.pre.norm_layers." + std::to_string(i);
      add(q + ".gamma", H);
      add(q + ".beta", H);
    }
    add("encoder.pre.proj.weight", H * H);
    add("encoder.pre.proj.bias", H);
  }
  for (int l = 0; l < h.n_layers_enc; ++l) {
    std::string a = "encoder.encoder.attn_layers." + std::to_string(l);
    add(a + ".emb_rel_k", nrel * dk);
    add(a + ".emb_rel_v", nrel * dk);
    for (const char* c : {"conv_q", "conv_k", "conv_v", "conv_o"}) {
      add(a + "." + c + ".weight", H * H);
      add(a + "." + c + ".bias", H);
    }
    add("encoder.encoder.norm_layers_1." + std::to_string(l) + ".gamma", H);
    add("encoder.encoder.norm_layers_1." + std::to_string(l) + ".beta", H);
    std::string f = "encoder.encoder.ffn_layers." + std::to_string(l);
    add(f + ".conv_1.weight", Fc * H * k);
    add(f + ".conv_1.bias", Fc);
    add(f + ".conv_2.weight", H * Fc * k);
    add(f + ".conv_2.bias", H);
    add("encoder.encoder.norm_layers_2." + std::to_string(l) + ".gamma", H);
    add("encoder.encoder.norm_layers_2." + std::to_string(l) + ".beta", H);
  }
  add("encoder.proj_m.weight", M * H);
  add("encoder.proj_m.bias", M);
  add("encoder.proj_w.conv_1.weight", Fd * (H + gin) * k);  // input = [encoder output ; speaker vector] (models.py:114-116)
  add("encoder.proj_w.conv_1.bias", Fd);
  add("encoder.proj_w.norm_1.gamma", Fd);
  add("encoder.proj_w.norm_1.beta", Fd);
  add("encoder.proj_w.conv_2.weight", Fd * Fd * k);
  add("encoder.proj_w.conv_2.bias", Fd);
  add("encoder.proj_w.norm_2.gamma", Fd);
  add("encoder.proj_w.norm_2.beta", Fd);
  add("encoder.proj_w.proj.weight", Fd);
  add("encoder.proj_w.proj.bias", 1);
  const int64_t C = M * h.n_sqz, half = C / 2;
  for (int b = 0; b < h.n_blocks_dec; ++b) {
    std::string an = "decoder.flows." + std::to_string(3 * b);
    std::string ic = "decoder.flows." + std::to_string(3 * b + 1);
    std::string cp = "decoder.flows." + std::to_string(3 * b + 2);
    add(an + ".logs", C);
    add(an + ".bias", C);
    add(ic + ".weight_inv", (int64_t)h.n_split * h.n_split);
    add(cp + ".start.weight", H * half);
    add(cp + ".start.bias", H);
    if (gin) {
      add(cp + ".wn.cond_layer.weight", 2 * H * h.n_block_layers * gin);
      add(cp + ".wn.cond_layer.bias", 2 * H * h.n_block_layers);
    }
    for (int j = 0; j < h.n_block_layers; ++j) {
      std::string il = cp + ".wn.in_layers." + std::to_string(j);
      add(il + ".weight", 2 * H * H * h.kernel_size_dec);
      add(il + ".bias", 2 * H);
      std::string rl = cp + ".wn.res_skip_layers." + std::to_string(j);
      const int64_t rsn = (j < h.n_block_layers - 1) ? 2 * H : H;
      add(rl + ".weight", rsn * H);
      add(rl + ".bias", rsn);
    }
    add(cp + ".end.weight", C * H);
    add(cp + ".end.bias", C);
  }
  return m;
}

static std::vector<std::pair<std::string, int64_t>> hifigan_manifest(const mi355tts_hifigan_hparams& h) {
  std::vector<std::pair<std::string, int64_t>> m;
  auto add = [&](const std::string& n, int64_t e) { m.emplace_back(n, e); };
  const int64_t C0 = h.upsample_initial_channel;
  add("conv_pre.weight", C0 * h.num_mels * 7);
  add("conv_pre.bias", C0);
  int64_t ch = C0;
  for (int i = 0; i < h.num_upsamples; ++i) {
    const int64_t cin = C0 >> i, cout = C0 >> (i + 1);
    add("ups." + std::to_string(i) + ".weight", cin * cout * h.upsample_kernel_sizes[i]);
    add("ups." + std::to_string(i) + ".bias", cout);
    ch = cout;
    for (int j = 0; j < h.num_kernels; ++j) {
      const int n = i * h.num_kernels + j;
      const int64_t k = h.resblock_kernel_sizes[j];
      for (int d = 0; d < h.num_dilations; ++d) {
        std::string rb = "resblocks." + std::to_string(n);
        if (h.resblock_type == 1) {
          add(rb + ".convs1." + std::to_string(d) + ".weight", ch * ch * k);
          add(rb + ".convs1." + std::to_string(d) + ".bias", ch);
          add(rb + ".convs2." + std::to_string(d) + ".weight", ch * ch * k);
          add(rb + ".convs2." + std::to_string(d) + ".bias", ch);
        } else {
          add(rb + ".convs." + std::to_string(d) + ".weight", ch * ch * k);
          add(rb + ".convs." + std::to_string(d) + ".bias", ch);
        }
      }
    }
  }
  add("conv_post.weight", ch * 7);
  add("conv_post.bias", 1);
  return m;
}
