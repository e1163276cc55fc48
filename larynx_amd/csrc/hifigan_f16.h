// mi355tts host runtime — the native fp16 vocoder (MI355TTS_PRECISION_F16): what the reference's `half` switch is, `.half()` on
// the whole HiFi-GAN generator (larynx/hifi_gan.py:96-97; hifi_gan/models.py:91-98, 136-141, 186-202).  Weight packing at load,
// the tile choice and the layer schedule over conv_f16.h's kernels.  Every layer of the generator runs in this mode: conv_pre,
// the upsamplers, every ResBlock conv of every stage (wide and narrow) and conv_post read and write fp16 planes; the
// waveform leaves conv_post's tanh in f32 (the reference casts its half output to float there, larynx/hifi_gan.py:160-166).
// (one translation unit: included once by mi355tts.hip, after host_launch.h)
#pragma once

// ------------------------------------------------------------------ geometry the fp16 tiles cover
static int f16_halo(int K) {
  switch (K) {
    case 3: return ConvHalo<3>::v;
    case 5: return ConvHalo<5>::v;
    case 7: return ConvHalo<7>::v;
    case 11: return ConvHalo<11>::v;
    default: return -1;
  }
}
// "" when the model can run in fp16, else what stands in the way (mi355tts_model_set_precision reports it)
static std::string hifi_f16_unsupported(const mi355tts_hifigan_hparams& h) {
  const int C0 = h.upsample_initial_channel;
  if (h.num_kernels < 2 || h.num_kernels > 3) return "the fp16 schedule needs 2 or 3 ResBlock chains per stage";
  for (int i = 0; i < h.num_upsamples; ++i) {
    const int cin = C0 >> i, cout = C0 >> (i + 1);
    if ((cin % 8) || (cout % 8) || cout < 8) return "channel counts must be multiples of 8 (16-byte octet units)";
    if (h.upsample_kernel_sizes[i] != 2 * h.upsample_rates[i]) return "upsampler kernel must be twice its stride (two polyphase taps)";
  }
  if ((C0 >> h.num_upsamples) > 64) return "more than 64 channels into conv_post";
  for (int j = 0; j < h.num_kernels; ++j) {
    const int K = h.resblock_kernel_sizes[j];
    const int halo = f16_halo(K);
    if (halo < 0) return "ResBlock kernel size not one of 3, 5, 7, 11";
    for (int d = 0; d < h.num_dilations; ++d)
      if (h.resblock_dilations[j][d] < 1 || h.resblock_dilations[j][d] * (K - 1) > halo) return "ResBlock dilation beyond the staged halo";
  }
  return "";
}

struct HPackSink {
  std::vector<uint16_t> w;  // the fp16 arena (host)
  ArenaBuilder* ab;         // biases ride in the model's float arena
  HConvW add(const PackedConvH& p, int Cin) {
    HConvW d;
    d.w_off = (w.size() + 127) & ~(size_t)127;  // 256-byte alignment
    w.resize(d.w_off + p.w.size());
    std::memcpy(w.data() + d.w_off, p.w.data(), p.w.size() * sizeof(uint16_t));
    d.b_off = ab->add(p.bias);
    d.mtiles = p.mtiles;
    d.nslab = p.nslab;
    d.K = p.K;
    d.rows = p.rows;
    d.Cin = Cin;
    return d;
  }
};
// a plain conv w[Cout][Cin][K]
static HConvW add_conv_h(HPackSink& sk, const float* w, const float* bias, int Cout, int Cin, int K) {
  return sk.add(pack_conv_f16(
                    Cout, 4, Cin, K, 64, [&](int v, int ci, int k) { return w[((size_t)v * Cin + ci) * K + k]; }, [&](int v) { return bias[v]; },
                    bias != nullptr),
                Cin);
}
// ConvTranspose1d(Cin, Cout, 2 u, stride u, padding u / 2) in polyphase form: virtual row v = r * Cout + co (phase-major), two
// taps over q, tap k reads x[q + k - 1] and carries Wt[ci][co][(1 - k) u + r]  (see add_conv's ROWS_UPSAMPLE for the derivation)
static HConvW add_ups_h(HPackSink& sk, const float* wt, const float* bias, int Cout, int Cin, int u) {
  const int Ku = 2 * u;
  return sk.add(pack_conv_f16(
                    Cout * u, 4, Cin, 2, 64,
                    [&](int v, int ci, int k) {
                      const int r = v / Cout, co = v % Cout, m = 1 - k;
                      return wt[((size_t)ci * Cout + co) * Ku + m * u + r];
                    },
                    [&](int v) { return bias[v % Cout]; }, bias != nullptr),
                Cin);
}
// ---- the WaveNets of GlowTTS' coupling blocks for wn_f16_kernel (wn_f16.h)
static std::string glow_f16_unsupported(const mi355tts_glow_hparams& h) {
  if (h.hidden_channels != 192 && h.hidden_channels != 32) return "hidden_channels other than 192 (32)";
  if (h.kernel_size_dec != 5) return "kernel_size_dec other than 5";
  if (h.dilation_rate != 1) return "dilated WaveNet layers";
  if (h.n_speakers > 1) return "speaker conditioning";
  if (h.n_block_layers < 1 || h.n_block_layers > WN_MAX_LAYERS || 4 * h.n_block_layers >= WN_W) return "n_block_layers";
  return "";
}
// in_layers[j]: w [2H][H][K]; virtual 32-row tile p = the tanh rows of channels 16 p .. 16 p + 15, then their sigmoid rows
static HConvW add_wn_gate_h(HPackSink& sk, const float* w, const float* bias, int H, int K) {
  auto row_of = [H](int v) {
    const int p = v / 32, i = v % 32, c = 16 * p + (i & 15);
    return i < 16 ? c : H + c;
  };
  return sk.add(pack_conv_f16(
                    2 * H, 1, H, K, 32, [&](int v, int ci, int k) { return w[((size_t)row_of(v) * H + ci) * K + k]; },
                    [&](int v) { return bias[row_of(v)]; }, true),
                H);
}
// res_skip_layers[j] (j < n - 1): w [2H][H][1], rows in natural order [res | skip]
static HConvW add_wn_rs_h(HPackSink& sk, const float* w, const float* bias, int H) {
  return sk.add(pack_conv_f16(
                    2 * H, 1, H, 1, 32, [&](int v, int ci, int) { return w[(size_t)v * H + ci]; }, [&](int v) { return bias[v]; }, true),
                H);
}
static void fix_h(HConvW& c, const uint16_t* arenaH, const float* arena) {
  c.w = reinterpret_cast<const uint4*>(arenaH + c.w_off);
  c.bias = arena + c.b_off;
}

// ------------------------------------------------------------------ tiles
// Three tile shapes (all 256 threads, 32-channel staged chunks), chosen by the conv's output rows and input channels:
//   WIDE : 2 x 2 waves of 64 rows x 64 columns — 128 rows x 128 columns per workgroup, the three-chunk ring (any Cin)
//   MID  : 1 x 4 waves of 64 x 64              —  64 rows x 256 columns, Cin <= 64: both chunks staged in the prologue
//   SLIM : 1 x 4 waves of 32 x 64              —  32 rows x 256 columns, Cin <= 32: the one chunk
enum HTile { HT_WIDE = 0, HT_MID, HT_SLIM };
static int h_tile_for(int rows, int Cin) { return (rows <= 32 && Cin <= 32) ? HT_SLIM : (rows < 128 && Cin <= 64) ? HT_MID : HT_WIDE; }
static void h_tile_dims(int t, int& trows, int& tcols) {
  trows = t == HT_WIDE ? 128 : t == HT_MID ? 64 : 32;
  tcols = t == HT_WIDE ? 128 : 256;
}
// template arguments MB, NB, WM, WN (then HALO ..., the chunk size) and the ring depth of each shape
#define H_TILE_PARAMS_WIDE 2, 2, 2, 2
#define H_TILE_PARAMS_MID 2, 2, 1, 4
#define H_TILE_PARAMS_SLIM 1, 2, 1, 4
constexpr int H_CH = 32;
constexpr int H_RING_WIDE = 3, H_RING_MID = 2, H_RING_SLIM = 1;

struct HPlan {
  HConvArgs a;
  int K = 0, tile = 0, epi = EPI_LINEAR;
  bool mrf = false;
  dim3 grid;
  int gx = 0, gy = 0;
  double flop = 0;
};

template <int K, int EPI, bool MRF>
static int launch_f16_k(const HPlan& p, hipStream_t s) {
  constexpr int HALO = ConvHalo<K>::v;
  switch (p.tile) {
    case HT_WIDE: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_kernel<K, H_TILE_PARAMS_WIDE, HALO, H_CH, EPI, MRF, H_RING_WIDE>), p.grid, dim3(256), 0, s, p.a); return 0;
    case HT_MID: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_kernel<K, H_TILE_PARAMS_MID, HALO, H_CH, EPI, MRF, H_RING_MID>), p.grid, dim3(256), 0, s, p.a); return 0;
    case HT_SLIM: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_kernel<K, H_TILE_PARAMS_SLIM, HALO, H_CH, EPI, MRF, H_RING_SLIM>), p.grid, dim3(256), 0, s, p.a); return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "internal: fp16 tile %d", p.tile);
}
static int run_plan_f16(mi355tts_ctx* ctx, Worker* w, const HPlan& p, int cls, hipStream_t s) {
  ProfScope ps(ctx, w, cls, p.flop, s);
  kn_hit(ctx, KN_CONV_F16);
  g_last_sub = p.a.rows;
  if (p.epi == EPI_UPSAMPLE) return p.mrf ? launch_f16_k<2, EPI_UPSAMPLE, true>(p, s) : launch_f16_k<2, EPI_UPSAMPLE, false>(p, s);
  switch (p.K) {
    case 3: return launch_f16_k<3, EPI_LINEAR, false>(p, s);
    case 5: return launch_f16_k<5, EPI_LINEAR, false>(p, s);
    case 7: return launch_f16_k<7, EPI_LINEAR, false>(p, s);
    case 11: return launch_f16_k<11, EPI_LINEAR, false>(p, s);
  }
  return fail(MI355TTS_ERR_INVALID, "internal: fp16 conv with %d taps", p.K);
}

template <int K0, int K1, int K2>
static int launch_f16_group_k(int tile, dim3 grid, const HConvGroupArgs& g, hipStream_t s) {
  constexpr int H0 = ConvHalo<K0>::v, H1 = ConvHalo<K1>::v, H2 = ConvHalo<K2>::v;
  switch (tile) {
    case HT_WIDE: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_group_kernel<K0, K1, K2, H_TILE_PARAMS_WIDE, H0, H1, H2, H_CH, H_RING_WIDE>), grid, dim3(256), 0, s, g); return 0;
    case HT_MID: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_group_kernel<K0, K1, K2, H_TILE_PARAMS_MID, H0, H1, H2, H_CH, H_RING_MID>), grid, dim3(256), 0, s, g); return 0;
    case HT_SLIM: hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_f16_group_kernel<K0, K1, K2, H_TILE_PARAMS_SLIM, H0, H1, H2, H_CH, H_RING_SLIM>), grid, dim3(256), 0, s, g); return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "internal: fp16 tile %d", tile);
}
// The same-geometry convs of a step's three chains in ONE launch (members longest first).  Returns 0 = launched, 1 = this tap
// set has no grouped kernel (the caller launches the members one by one), < 0 = error.
static int run_group_f16(mi355tts_ctx* ctx, Worker* w, const HPlan* p, int n, int B, hipStream_t s) {
  if (n != 3) return 1;
  int ord[3] = {0, 1, 2};
  std::sort(ord, ord + 3, [&](int x, int y) { return p[x].K > p[y].K; });
  const int K0 = p[ord[0]].K, K1 = p[ord[1]].K, K2 = p[ord[2]].K;
  const bool k1173 = K0 == 11 && K1 == 7 && K2 == 3, k753 = K0 == 7 && K1 == 5 && K2 == 3;
  if (!k1173 && !k753) return 1;
  if (p[0].tile != p[1].tile || p[0].tile != p[2].tile) return 1;
  HConvGroupArgs g;
  std::memset(&g, 0, sizeof(g));
  int off = 0;
  double flop = 0;
  for (int m = 0; m < 3; ++m) {
    const HPlan& q = p[ord[m]];
    g.c[m] = q.a;
    g.gx[m] = q.gx;
    g.gy[m] = q.gy;
    g.off[m] = off;
    off += (q.gx * q.gy + 7) & ~7;
    flop += q.flop;
  }
  g.off[3] = off;
  ProfScope ps(ctx, w, KC_RESBLOCK, flop, s);
  kn_hit(ctx, KN_CONV_F16_GROUP);
  g_last_sub = p[0].a.rows;
  const dim3 grid(off, 1, B);
  return k1173 ? launch_f16_group_k<11, 7, 3>(p[0].tile, grid, g, s) : launch_f16_group_k<7, 5, 3>(p[0].tile, grid, g, s);
}

// ---- fused ResBlock1 steps (pair_f16.h): conv1 + conv2 of a dilation step of the three chains in ONE launch.  A workgroup owns
// all channels of its columns, so the tile is chosen by the channel count: C <= 32 SLIM, C <= 64 MID, C <= 128 WIDE.
struct HPairPlan {
  HPairArgs a;
  int K = 0;
  double flop = 0;
};
static int h_pair_tile(int C) { return C <= 32 ? HT_SLIM : C <= 64 ? HT_MID : C <= 128 ? HT_WIDE : -1; }
template <int K0, int K1, int K2>
static int launch_pair_group_k(int tile, dim3 grid, const HPairGroupArgs& g, hipStream_t s) {
  constexpr int H0 = ConvHalo<K0>::v, H1 = ConvHalo<K1>::v, H2 = ConvHalo<K2>::v;
  switch (tile) {
    case HT_WIDE: hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<K0, K1, K2, H_TILE_PARAMS_WIDE, H0, H1, H2, H_CH, H_RING_WIDE, 3>), grid, dim3(256), 0, s, g); return 0;
    case HT_MID: hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<K0, K1, K2, H_TILE_PARAMS_MID, H0, H1, H2, H_CH, H_RING_MID, 3>), grid, dim3(256), 0, s, g); return 0;
    case HT_SLIM: hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_f16_group_kernel<K0, K1, K2, H_TILE_PARAMS_SLIM, H0, H1, H2, H_CH, H_RING_SLIM, 4>), grid, dim3(256), 0, s, g); return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "internal: fp16 pair tile %d", tile);
}
// Returns 0 = launched, 1 = not covered (the caller runs conv1 and conv2 as two grouped launches), < 0 = error.
static int run_pair_group_f16(mi355tts_ctx* ctx, Worker* w, const HPairPlan* p, int n, int C, int B, int Lmax, hipStream_t s) {
  if (n != 3) return 1;
  const int tile = h_pair_tile(C);
  if (tile < 0) return 1;
  int ord[3] = {0, 1, 2};
  std::sort(ord, ord + 3, [&](int x, int y) { return p[x].K > p[y].K; });
  if (!(p[ord[0]].K == 11 && p[ord[1]].K == 7 && p[ord[2]].K == 3)) return 1;
  int tr, tc;
  h_tile_dims(tile, tr, tc);
  HPairGroupArgs g;
  std::memset(&g, 0, sizeof(g));
  g.n = 3;
  int off = 0;
  double flop = 0;
  for (int m = 0; m < 3; ++m) {
    const HPairPlan& q = p[ord[m]];
    const int to = tc - (q.K - 1);
    g.p[m] = q.a;
    g.gx[m] = (Lmax + to - 1) / to;
    g.off[m] = off;
    off += (g.gx[m] + 7) & ~7;
    flop += q.flop;
  }
  g.off[3] = off;
  ProfScope ps(ctx, w, KC_RESBLOCK, flop, s);
  kn_hit(ctx, KN_PAIR_F16_GROUP);
  g_last_sub = C;
  return launch_pair_group_k<11, 7, 3>(tile, dim3(off, 1, B), g, s);
}

// lengths: one row with a host-known length takes it as a launch constant (no dependent load in every workgroup's prologue)
static void h_set_lengths(HConvArgs& a, const int* d_frames, int host_len, int in_mul, int out_mul) {
  if (host_len >= 0) {
    a.in_len = nullptr;
    a.out_len = nullptr;
    a.in_const = host_len * in_mul;
    a.out_const = host_len * out_mul;
  } else {
    a.in_len = d_frames;
    a.out_len = d_frames;
  }
  a.in_mul = in_mul;
  a.out_mul = out_mul;
}
static HPlan plan_f16(const HConvW& c, HConvArgs a, int epi, int B, int n_max, double flop) {
  HPlan p;
  a.w = c.w;
  a.bias = c.bias;
  a.nslab = c.nslab;
  a.Cin = c.Cin;
  a.rows = c.rows;
  p.a = a;
  p.K = c.K;
  p.epi = epi;
  p.tile = h_tile_for(c.rows, c.Cin);
  int tr, tc;
  h_tile_dims(p.tile, tr, tc);
  p.gx = (n_max + tc - 1) / tc;
  p.gy = (c.rows + tr - 1) / tr;
  p.grid = dim3(p.gx, p.gy, B);
  p.flop = flop;
  return p;
}

// ------------------------------------------------------------------ the generator, conv_pre .. conv_post + tanh
// planes: `buf[i]` are the worker's plane buffers (hifi_layout: B x `plane` floats each, 2 + 4 nk of them) — an fp16 plane of the
// same channels and row stride takes half of one.  Leaves the f32 waveform rows in `wav` ([B][Nld]) and, when asked, the
// |max| of every 256-sample tile in `peak` (voc_out.h's wave_out_kernel reads both).
static int hifigan_body_f16(mi355tts_ctx* ctx, Worker* w, HifiModel* hm, const mi355tts_mel* mel, float* const* buf, float* wav, size_t Nld,
                            float* peak, long long peak_ld, int voc_host_len, bool use_group, bool use_pairs, hipStream_t s) {
  const mi355tts_hifigan_hparams& h = hm->hp;
  const int B = mel->B, F = mel->max_frames;
  const int C0 = h.upsample_initial_channel, nk = h.num_kernels, nd = h.num_dilations;
  const int* d_frames = mel->frames_dev;
  auto plane = [&](int i) { return reinterpret_cast<uint4*>(buf[i]); };

  // mel [B][M][ld] f32 -> octet planes (buf[1]: free until the first upsampler writes it)
  const int M = mel->M, moct = (M + 7) / 8;
  uint4* melh = plane(1);
  {
    ProfScope ps(ctx, w, KC_SMALL, 0, s);
    kn_hit(ctx, KN_PACK_OCTETS);
    hipLaunchKernelGGL(pack_octets_kernel, dim3((F + 255) / 256, moct, B), dim3(256), 0, s, mel->voc, (long long)mel->M * mel->ld, mel->ld, M, d_frames, 1,
                       melh, (long long)moct * F, F);
  }
  uint4* cur[3] = {plane(0), nullptr, nullptr};
  int ncur = 1;
  float cur_div = 1.0f;
  {  // conv_pre (models.py:187)
    HConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = melh;
    a.x_bs = (long long)moct * F;
    a.x_ld = F;
    a.in_div = 1.0f;
    a.dil = 1;
    a.pad = 3;
    a.in_slope = 1.0f;
    a.out_slope = 1.0f;
    a.y = cur[0];
    a.y_bs = (long long)(C0 / 8) * F;
    a.y_ld = F;
    a.cout = C0;
    h_set_lengths(a, d_frames, voc_host_len, 1, 1);
    const HPlan p = plan_f16(hm->h_pre, a, EPI_LINEAR, B, F, 2.0 * C0 * M * 7 * (double)F * B);
    CHECK(run_plan_f16(ctx, w, p, KC_VOC_IO, s));
  }
  uint4* xu = plane(1);
  int mul = 1, Lin = F, ldin = F, ch = C0, flip = 0;
  for (int i = 0; i < h.num_upsamples; ++i) {
    const int u = h.upsample_rates[i];
    const int cout = C0 >> (i + 1);
    const int Lout = Lin * u, ldo = Lout;
    {  // x = ups[i](leaky_relu(x, 0.1))  (models.py:189-190); the MRF average of the previous stage is taken on load
      HConvArgs a;
      std::memset(&a, 0, sizeof(a));
      a.x = cur[0];
      a.x2 = ncur > 1 ? cur[1] : nullptr;
      a.x3 = ncur > 2 ? cur[2] : nullptr;
      a.in_div = cur_div;
      a.x_bs = (long long)(ch / 8) * ldin;
      a.x_ld = ldin;
      a.dil = 1;
      a.pad = 1;
      a.in_slope = 0.1f;
      a.out_slope = 1.0f;
      a.y = xu;
      a.y_bs = (long long)(cout / 8) * ldo;
      a.y_ld = ldo;
      a.up = u;
      a.up_pad = u / 2;
      a.cout = cout;
      h_set_lengths(a, d_frames, voc_host_len, mul, mul * u);
      HPlan p = plan_f16(hm->h_ups[i], a, EPI_UPSAMPLE, B, Lin + 1, 2.0 * ch * cout * (2.0 * u) * (double)Lin * B);
      p.mrf = ncur > 1;
      CHECK(run_plan_f16(ctx, w, p, KC_UPSAMPLE, s));
    }
    mul *= u;
    ch = cout;
    const long long bs = (long long)(ch / 8) * ldo;
    // per-chain planes: buf[2 + 4 j ..] = {t, ping, out (flip 0), out (flip 1)}; last stage's outputs are dead once the upsampler has read them
    uint4 *tb[3], *pa[3], *pb[3], *dst_last[3];
    const uint4* rin[3];
    for (int j = 0; j < nk; ++j) {
      tb[j] = plane(2 + 4 * j);
      pa[j] = plane(2 + 4 * j + 1);
      pb[j] = plane(2 + 4 * j + 2 + (flip ^ 1));
      dst_last[j] = plane(2 + 4 * j + 2 + flip);
      rin[j] = xu;
    }
    auto conv_args = [&](const uint4* x, uint4* y, const uint4* res, int K, int dil, float in_slope, float out_slope) {
      HConvArgs a;
      std::memset(&a, 0, sizeof(a));
      a.x = x;
      a.in_div = 1.0f;
      a.x_bs = bs;
      a.x_ld = ldo;
      a.dil = dil;
      a.pad = (K * dil - dil) / 2;
      a.in_slope = in_slope;
      a.out_slope = out_slope;
      a.y = y;
      a.y_bs = bs;
      a.y_ld = ldo;
      a.res = res;
      a.cout = ch;
      h_set_lengths(a, d_frames, voc_host_len, mul, mul);
      return a;
    };
    const bool use_pair = h.resblock_type == 1 && use_pairs;  // (the options as the call saw them at its start)
    for (int d = 0; d < nd; ++d) {
      HPlan c1[3], c2[3];
      HPairPlan pp[3];
      uint4* dst[3];
      for (int j = 0; j < nk; ++j) {
        const HResConv& rc = hm->h_rb[i][j][d];
        const int K = h.resblock_kernel_sizes[j], dil = h.resblock_dilations[j][d];
        const double flop = 2.0 * ch * ch * K * (double)Lout * B;
        dst[j] = d == nd - 1 ? dst_last[j] : ((d & 1) ? pb[j] : pa[j]);
        if (h.resblock_type == 1) {
          // ResBlock1.forward (models.py:91-98): xt = c2(lrelu(c1(lrelu(x)))); x = xt + x.  conv1 stores lrelu(c1(.)) — its only
          // consumer is conv2, which would apply it on load
          c1[j] = plan_f16(rc.c1, conv_args(rin[j], tb[j], nullptr, K, dil, 0.1f, 0.1f), EPI_LINEAR, B, Lout, flop);
          c2[j] = plan_f16(rc.c2, conv_args(tb[j], dst[j], rin[j], K, 1, 1.0f, 1.0f), EPI_LINEAR, B, Lout, flop);
          HPairArgs& a = pp[j].a;
          std::memset(&a, 0, sizeof(a));
          a.x = rin[j];
          a.y = dst[j];
          a.bs = bs;
          a.ld = ldo;
          if (voc_host_len >= 0) {
            a.len = nullptr;
            a.len_const = voc_host_len * mul;
          } else {
            a.len = d_frames;
          }
          a.len_mul = mul;
          a.w1 = rc.c1.w;
          a.b1 = rc.c1.bias;
          a.nslab1 = rc.c1.nslab;
          a.w2 = rc.c2.w;
          a.b2 = rc.c2.bias;
          a.nslab2 = rc.c2.nslab;
          a.C = ch;
          a.dil = dil;
          a.slope = 0.1f;
          pp[j].K = K;
          pp[j].flop = 2.0 * flop;
        } else {
          // ResBlock2.forward (models.py:136-141): x = c(lrelu(x)) + x
          c1[j] = plan_f16(rc.c1, conv_args(rin[j], dst[j], rin[j], K, dil, 0.1f, 1.0f), EPI_LINEAR, B, Lout, flop);
        }
      }
      int fused = 1;
      if (use_pair) {
        fused = run_pair_group_f16(ctx, w, pp, nk, ch, B, Lout, s);
        if (fused < 0) return fused;
      }
      for (int pass = 0; fused != 0 && pass < (h.resblock_type == 1 ? 2 : 1); ++pass) {
        const HPlan* pl = pass ? c2 : c1;
        int rc = use_group ? run_group_f16(ctx, w, pl, nk, B, s) : 1;
        if (rc < 0) return rc;
        if (rc == 1)
          for (int j = 0; j < nk; ++j) CHECK(run_plan_f16(ctx, w, pl[j], KC_RESBLOCK, s));
      }
      for (int j = 0; j < nk; ++j) rin[j] = dst[j];
    }
    for (int j = 0; j < nk; ++j) cur[j] = dst_last[j];
    ncur = nk;
    cur_div = (float)nk;
    flip ^= 1;
    Lin = Lout;
    ldin = ldo;
  }
  {  // x = tanh(conv_post(leaky_relu(x)))  — default slope 0.01 (models.py:198-201)
    HPostArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = cur[0];
    a.x2 = ncur > 1 ? cur[1] : nullptr;
    a.x3 = ncur > 2 ? cur[2] : nullptr;
    a.in_div = cur_div;
    a.slope = 0.01f;
    a.x_bs = (long long)(ch / 8) * ldin;
    a.x_ld = ldin;
    if (voc_host_len >= 0) {
      a.len = nullptr;
      a.len_const = voc_host_len * mul;
    } else {
      a.len = d_frames;
    }
    a.len_mul = mul;
    a.w = hm->arena + hm->post_w_off;
    a.bias = hm->arena + hm->post_b_off;
    a.C = ch;
    a.y = wav;
    a.y_bs = (long long)Nld;
    a.peak = peak;
    a.peak_ld = peak_ld;
    ProfScope ps(ctx, w, KC_VOC_IO, 2.0 * (double)ch * 7 * (double)Lin * B, s);
    kn_hit(ctx, KN_POST_F16);
    const dim3 pg((Lin + HPOST_TW - 1) / HPOST_TW, B);
    if (ncur > 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(post_f16_kernel<7, 3>), pg, dim3(256), 0, s, a);
    else if (ncur > 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(post_f16_kernel<7, 2>), pg, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(post_f16_kernel<7, 1>), pg, dim3(256), 0, s, a);
  }
  return 0;
}
