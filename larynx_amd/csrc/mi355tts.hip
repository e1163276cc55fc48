// mi355tts — host runtime + C ABI of the MI355X-native Larynx hot path.
//
// Owns: model arenas in HBM (weights re-laid-out once at load into MFMA fragment
// order), a pool of per-call workers (HIP stream + grow-only workspace + pinned
// staging), the layer schedule of the two networks, and the profiling hooks.
// The schedule follows the reference's module graph:
//   glow_tts/models.py:118-140 (TextEncoder), :191-209 (FlowSpecDecoder reverse),
//   :308-354 (FlowGenerator), hifi_gan/models.py:186-202 (Generator).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355tts.h"
#include "conv_mfma.h"
#include "resblock_pair.h"
#include "small_kernels.h"
#include "weights_pack.h"

using namespace mi355tts;

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
  size_t w_off = 0, b_off = 0;  // offsets (floats) into the model arena
  const float* w = nullptr;
  const float* bias = nullptr;
  int mtiles = 0, noct = 0, K = 0, rows = 0, Cin = 0, Cout = 0, MB = 1;
  bool has_bias = false;
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

struct GlowLayer {
  DevConv qkv, o, ffn1, ffn2;
  size_t ek, ev, g1, b1, g2, b2;
};
struct GlowBlock {
  DevConv start, end;
  std::vector<DevConv> in, rs;
  size_t winv, an_bias, an_scale;
};
struct GlowModel {
  mi355tts_glow_hparams hp;
  float* arena = nullptr;
  size_t emb;
  std::vector<DevConv> pre_conv;
  std::vector<size_t> pre_g, pre_b;
  DevConv pre_proj;
  std::vector<GlowLayer> layers;
  DevConv proj_m, dp1, dp2, dpp;
  size_t dg1, db1, dg2, db2;
  std::vector<GlowBlock> blocks;
};
struct HifiResConv {
  DevConv c1, c2;
  int dil;
};
struct HifiModel {
  mi355tts_hifigan_hparams hp;
  float* arena = nullptr;
  DevConv pre, post;
  std::vector<DevConv> ups;
  // [stage][kernel][dilation index]
  std::vector<std::vector<std::vector<HifiResConv>>> rb;
  int hop = 1;
  // denoiser bias spectrum |STFT(generator(zeros))|[:, 0] (larynx/hifi_gan.py:181-203), built on first use
  std::mutex bias_mu;
  float* bias_spec = nullptr;
  bool bias_ready = false;
};

static std::vector<std::pair<std::string, int64_t>> glow_manifest(const mi355tts_glow_hparams& h) {
  std::vector<std::pair<std::string, int64_t>> m;
  auto add = [&](const std::string& n, int64_t e) { m.emplace_back(n, e); };
  const int64_t H = h.hidden_channels, Fc = h.filter_channels, Fd = h.filter_channels_dp, M = h.mel_channels;
  const int64_t k = h.kernel_size, dk = H / std::max(1, h.n_heads), nrel = 2 * h.window_size + 1;
  add("encoder.emb.weight", (int64_t)h.num_symbols * H);
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
  add("encoder.proj_w.conv_1.weight", Fd * H * k);
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

// ------------------------------------------------------------------ context
struct ProfEvent {
  hipEvent_t a, b;
  int cls;
  double flop;
};
enum KClass { KC_RESBLOCK = 0, KC_UPSAMPLE, KC_VOC_IO, KC_GLOW_ENC_CONV, KC_GLOW_DEC_CONV, KC_SMALL, KC_COUNT };
static const char* kclass_name[KC_COUNT] = {"conv_mfma.hifigan_resblock", "conv_mfma.hifigan_upsample",
                                            "conv_mfma.hifigan_pre_post", "conv_mfma.glow_encoder",
                                            "conv_mfma.glow_decoder",     "elementwise"};

struct Worker {
  hipStream_t stream = nullptr;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  size_t arena_pos = 0;
  int* pinned = nullptr;  // pinned host staging for frame counts
  size_t pinned_ints = 0;
  std::vector<ProfEvent> events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  // side streams for the independent MRF branches of a HiFi-GAN stage
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_join[2] = {nullptr, nullptr};
};

struct mi355tts_ctx {
  int device = 0;
  std::mutex mu;
  std::map<int, std::unique_ptr<GlowModel>> glow;
  std::map<int, std::unique_ptr<HifiModel>> hifi;
  int next_id = 1;
  std::vector<Worker*> free_workers;
  std::vector<Worker*> all_workers;
  bool profiling = false;
  bool serial_branches = false;
  // recycled device blocks for the mel result objects: hipMalloc/hipFree synchronise
  // the whole device, which would serialise the concurrent per-utterance streams
  std::vector<std::pair<void*, size_t>> mel_pool;
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
struct Carver {
  size_t pos = 0;
  size_t take(size_t bytes) {
    size_t off = (pos + 255) & ~(size_t)255;
    pos = off + bytes;
    return off;
  }
};

// ------------------------------------------------------------------ launch helpers
struct ProfScope {
  mi355tts_ctx* ctx;
  Worker* w;
  bool on;
  ProfEvent ev;
  hipStream_t st;
  ProfScope(mi355tts_ctx* c, Worker* wk, int cls, double flop, hipStream_t stream = nullptr)
      : ctx(c), w(wk), on(c->profiling), st(stream ? stream : wk->stream) {
    if (!on) return;
    if (!w->event_pool.empty()) {
      ev.a = w->event_pool.back().first;
      ev.b = w->event_pool.back().second;
      w->event_pool.pop_back();
    } else {
      if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) {
        on = false;
        return;
      }
    }
    ev.cls = cls;
    ev.flop = flop;
    hipEventRecord(ev.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(ev.b, st);
    w->events.push_back(ev);
  }
};

template <int K, int CI_C, int MB, int NB, int WN, int KS, int HALO, int EPI>
static void launch_conv_inst(hipStream_t s, dim3 grid, const ConvArgs& a) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_kernel<K, CI_C, MB, NB, WN, KS, HALO, EPI>), grid, dim3(64 * WN * KS), 0, s, a);
}

// LDS halo capacity per tap count (max (K-1)*dilation the reference configs need)
template <int K> struct ConvCfg;
template <> struct ConvCfg<1> { static constexpr int HALO = 0; };
template <> struct ConvCfg<2> { static constexpr int HALO = 4; };
template <> struct ConvCfg<3> { static constexpr int HALO = 16; };
template <> struct ConvCfg<5> { static constexpr int HALO = 28; };
template <> struct ConvCfg<7> { static constexpr int HALO = 76; };
template <> struct ConvCfg<11> { static constexpr int HALO = 56; };

// Tile shapes (all 512 threads):
// (2-column-block-per-wave variants at 64/128 columns, a 256-thread variant without
//  k-split, and one-m-tile "wide" tiles with 2 or 4 column blocks per wave were measured
//  in round 1 and did not win overall; see profiles/r01_conv_sweep*.txt)
//   TINY  : 1 time-wave  x 8 k-groups, 32 columns  — launches with only a handful of tiles (GlowTTS at batch 1)
//   SMALL : 2 time-waves x 4 k-groups, 64 columns  — few-tile launches (stage 0 at batch 1)
//   NB1   : 4 time-waves x 2 k-groups, 128 columns
//   NB2   : 4 time-waves x 2 k-groups, 256 columns (64x64 outputs per wave)
enum TileShape { TILE_SMALL = 0, TILE_NB1 = 1, TILE_NB2 = 2, TILE_TINY = 3, TILE_LAST = 3 };
static thread_local int g_pin_tile = -1;  // set by mi355tts_bench_conv1d only

template <int K, int EPI>
static int launch_conv_k(hipStream_t s, int MB, int shape, dim3 grid, const ConvArgs& a) {
  constexpr int HALO = ConvCfg<K>::HALO;
  constexpr int CI_BIG = (K == 1) ? 64 : (K <= 5) ? 32 : 16;
  constexpr int CI_SMALL = (K == 1) ? 64 : 32;
  constexpr bool PAIRED = (EPI == EPI_GATE || EPI == EPI_COUPLING);
  // the staged tile starts at the 4-aligned column t0 - roundup(pad, 4)
  if ((K - 1) * a.dil + ((4 - a.pad % 4) % 4) > HALO)
    return fail(MI355TTS_ERR_INVALID, "conv K=%d dilation=%d exceeds the staged halo", K, a.dil);
  if (a.x_ld % 4) return fail(MI355TTS_ERR_INVALID, "internal: activation row stride %d is not a multiple of 4", a.x_ld);
  if (MB == 1) {
    if (shape == TILE_TINY) launch_conv_inst<K, 64, 1, 1, 1, 8, HALO, EPI>(s, grid, a);
    else if (shape == TILE_SMALL) launch_conv_inst<K, CI_SMALL, 1, 1, 2, 4, HALO, EPI>(s, grid, a);
    else if (shape == TILE_NB1) launch_conv_inst<K, CI_BIG, 1, 1, 4, 2, HALO, EPI>(s, grid, a);
    else launch_conv_inst<K, 16, 1, 2, 4, 2, HALO, EPI>(s, grid, a);
    return 0;
  }
  if constexpr (!PAIRED) {
    if (shape == TILE_TINY) launch_conv_inst<K, 64, 2, 1, 1, 8, HALO, EPI>(s, grid, a);
    else if (shape == TILE_SMALL) launch_conv_inst<K, CI_SMALL, 2, 1, 2, 4, HALO, EPI>(s, grid, a);
    else if (shape == TILE_NB1) launch_conv_inst<K, CI_BIG, 2, 1, 4, 2, HALO, EPI>(s, grid, a);
    else launch_conv_inst<K, 16, 2, 2, 4, 2, HALO, EPI>(s, grid, a);
    return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "paired epilogues run on 32-row tiles (MB == 1)");
}

// `a` arrives with every tensor/epilogue field filled; this picks the tile and
// template instance.  n_max = largest GEMM-N extent over the batch rows.
static int launch_conv(mi355tts_ctx* ctx, Worker* w, const DevConv& c, ConvArgs a, int epi, int B, int n_max, int cls,
                       hipStream_t stream = nullptr, int min_tiles = 1024, int host_len = -1) {
  if (n_max <= 0 || B <= 0) return 0;
  if (B == 1 && host_len >= 0) {
    // single utterance: the host already knows the row length, so the kernel need not
    // start with a dependent global load of len[b]
    if (a.in_len) {
      a.in_const = host_len * a.in_mul;
      a.in_len = nullptr;
    }
    if (a.out_len) {
      a.out_const = host_len * a.out_mul;
      a.out_len = nullptr;
    }
  }
  if (epi == EPI_LINEAR && a.split > 0 && a.split < c.rows && (a.split % 32))
    return fail(MI355TTS_ERR_INVALID, "row split %d must be a multiple of 32", a.split);
  a.w = c.w;
  a.bias = c.has_bias ? c.bias : nullptr;
  a.noct = c.noct;
  a.Cin = c.Cin;
  a.rows = c.rows;
  int MB = c.MB;
  int ytiles = c.mtiles / MB;
  // Tile shape: the largest tile that still yields >= min_tiles workgroups, otherwise the
  // smallest tile.  1024 (4 per CU) is the measured sweet spot for a kernel that has the
  // chip to itself (tools/conv_sweep.py); the three concurrent MRF chains ask for 300
  // each — together they fill the chip, and the bigger tiles run closer to the MFMA rate.
  auto tiles = [&](int width) { return (long long)((n_max + width - 1) / width) * ytiles * B; };
  const long long want = min_tiles;
  int shape = TILE_TINY;
  if (tiles(256) >= want) shape = TILE_NB2;
  else if (tiles(128) >= want) shape = TILE_NB1;
  else if (tiles(64) >= want) shape = TILE_SMALL;
  {  // tuning / test knob: MI355TTS_FORCE_TILE=0|1|2 pins the tile shape
    static const int forced = [] {
      const char* e = std::getenv("MI355TTS_FORCE_TILE");
      return e ? std::atoi(e) : -1;
    }();
    int f = forced;
    if (const char* dyn = std::getenv("MI355TTS_FORCE_TILE_DYNAMIC")) f = std::atoi(dyn);
    if (g_pin_tile >= 0) f = g_pin_tile;
    if (f >= TILE_SMALL && f <= TILE_LAST) shape = f;
  }
  // a launch that cannot even give every CU one workgroup: halve the row tile too
  // (32-row m-tiles are independent in the packed weights; paired epilogues need both)
  if (shape == TILE_TINY && MB == 2 && (epi == EPI_LINEAR || epi == EPI_UPSAMPLE) && tiles(32) < 256) {
    MB = 1;
    ytiles = (c.rows + 31) / 32;
  }
  const int T_T = shape == TILE_TINY ? 32 : shape == TILE_SMALL ? 64 : (shape == TILE_NB2 ? 256 : 128);
  dim3 grid((n_max + T_T - 1) / T_T, ytiles, B);
  const double flop = 2.0 * (double)c.Cout * c.Cin * (epi == EPI_UPSAMPLE ? c.K * a.up : c.K) * (double)n_max * B;
  hipStream_t s = stream ? stream : w->stream;
  ProfScope ps(ctx, w, cls, flop, s);
  int rc = 0;
  if (epi == EPI_LINEAR) {
    switch (c.K) {
      case 1: rc = launch_conv_k<1, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 3: rc = launch_conv_k<3, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 5: rc = launch_conv_k<5, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 7: rc = launch_conv_k<7, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 11: rc = launch_conv_k<11, EPI_LINEAR>(s, MB, shape, grid, a); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported conv kernel size %d", c.K);
    }
  } else if (epi == EPI_GATE) {
    switch (c.K) {
      case 3: rc = launch_conv_k<3, EPI_GATE>(s, MB, shape, grid, a); break;
      case 5: rc = launch_conv_k<5, EPI_GATE>(s, MB, shape, grid, a); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported WaveNet kernel size %d", c.K);
    }
  } else if (epi == EPI_COUPLING) {
    if (c.K == 1) rc = launch_conv_k<1, EPI_COUPLING>(s, MB, shape, grid, a);
    else rc = fail(MI355TTS_ERR_INVALID, "coupling conv must be 1x1");
  } else {
    switch (c.K) {
      case 1: rc = launch_conv_k<1, EPI_UPSAMPLE>(s, MB, shape, grid, a); break;
      case 2: rc = launch_conv_k<2, EPI_UPSAMPLE>(s, MB, shape, grid, a); break;
      case 3: rc = launch_conv_k<3, EPI_UPSAMPLE>(s, MB, shape, grid, a); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported upsample taps %d", c.K);
    }
  }
  return rc;
}

// Fused ResBlock1 step (conv1 -> lrelu -> conv2 -> + x) for the 32/64-channel stages.
// Returns 1 if the geometry is not covered (caller falls back to two conv launches).
static int launch_pair(mi355tts_ctx* ctx, Worker* w, const DevConv& c1, const DevConv& c2, const float* x, float* y, long long bs,
                       int ld, const int* len, int len_mul, int dil, float alpha, int accum, int B, int Lmax, hipStream_t s,
                       int host_len = -1) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_PAIR_FUSION"); return e && std::atoi(e) != 0; }();
  const int nb64 = 1;  // measured: 128-column tiles beat 256 at C = 64 (163 vs 197 us for the k = 11 pair)
  const int C = c1.Cout, K = c1.K;
  if (off || (C != 32 && C != 64) || c1.Cin != C || c2.Cin != C || c2.Cout != C || c2.K != K || dil > PAIR_DMAX || dil < 1 ||
      (K != 3 && K != 7 && K != 11) || c1.noct != c2.noct || !c1.has_bias || !c2.has_bias || (ld % 4) || x == y)
    return 1;
  PairArgs a;
  a.x = x;
  a.y = y;
  a.bs = bs;
  a.ld = ld;
  a.len = (B == 1 && host_len >= 0) ? nullptr : len;
  a.len_mul = len_mul;
  a.len_const = host_len * len_mul;
  a.w1 = c1.w;
  a.b1 = c1.bias;
  a.w2 = c2.w;
  a.b2 = c2.bias;
  a.noct = c1.noct;
  a.C = C;
  a.dil = dil;
  a.slope = 0.1f;
  a.alpha = alpha;
  a.accum = accum;
  const int NB = (C == 32) ? 2 : nb64;
  const int T2 = 128 * NB - (K - 1);
  dim3 grid((Lmax + T2 - 1) / T2, 1, B);
  const double flop = 2.0 * 2.0 * (double)C * C * K * (double)Lmax * B;
  ProfScope ps(ctx, w, KC_RESBLOCK, flop, s);
#define PAIR_LAUNCH(KK, CB, NBB) hipLaunchKernelGGL(HIP_KERNEL_NAME(resblock_pair_kernel<KK, CB, NBB>), grid, dim3(512), 0, s, a)
#define PAIR_K(KK)                                  \
  if (C == 32) PAIR_LAUNCH(KK, 1, 2);               \
  else if (NB == 2) PAIR_LAUNCH(KK, 2, 2);          \
  else PAIR_LAUNCH(KK, 2, 1)
  if (K == 3) { PAIR_K(3); }
  else if (K == 7) { PAIR_K(7); }
  else { PAIR_K(11); }
#undef PAIR_K
#undef PAIR_LAUNCH
  return 0;
}

static ConvArgs base_args(const float* x, long long x_bs, int x_ld, const int* in_len, int in_mul, float* y, long long y_bs,
                          int y_ld, const int* out_len, int out_mul, int dil, int pad) {
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x;
  a.x_bs = x_bs;
  a.x_ld = x_ld;
  a.in_len = in_len;
  a.in_mul = in_mul;
  a.y = y;
  a.y_bs = y_bs;
  a.y_ld = y_ld;
  a.out_len = out_len;
  a.out_mul = out_mul;
  a.dil = dil;
  a.pad = pad;
  a.in_slope = 1.0f;
  a.alpha = 1.0f;
  a.split = 1 << 30;
  a.out_act = ACT_NONE;
  return a;
}

static MelTransform to_mt(const mi355tts_audio_settings* s) {
  MelTransform m;
  std::memset(&m, 0, sizeof(m));
  if (!s) return m;
  m.signal_norm = s->signal_norm;
  m.symmetric_norm = s->symmetric_norm;
  m.clip_norm = s->clip_norm;
  m.convert_db_to_amp = s->convert_db_to_amp;
  m.do_drc = s->do_dynamic_range_compression;
  m.min_level_db = s->min_level_db;
  m.max_norm = s->max_norm;
  m.ref_level_db = s->ref_level_db;
  m.spec_gain = s->spec_gain;
  return m;
}

// ------------------------------------------------------------------ C ABI: basics
extern "C" int mi355tts_abi_version(void) { return MI355TTS_ABI_VERSION; }
extern "C" const char* mi355tts_last_error(void) { return g_err.c_str(); }

extern "C" int mi355tts_create(int device, mi355tts_ctx** out) {
  if (!out) return fail(MI355TTS_ERR_INVALID, "out is null");
  int n = 0;
  HIPCHECK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(MI355TTS_ERR_INVALID, "device %d out of range (%d visible)", device, n);
  HIPCHECK(hipSetDevice(device));
  mi355tts_ctx* c = new mi355tts_ctx();
  c->device = device;
  *out = c;
  return 0;
}

extern "C" void mi355tts_destroy(mi355tts_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  for (Worker* w : ctx->all_workers) {
    for (auto& ev : w->events) {
      hipEventDestroy(ev.a);
      hipEventDestroy(ev.b);
    }
    for (auto& p : w->event_pool) {
      hipEventDestroy(p.first);
      hipEventDestroy(p.second);
    }
    for (int i = 0; i < 2; ++i) {
      if (w->aux[i]) hipStreamDestroy(w->aux[i]);
      if (w->ev_join[i]) hipEventDestroy(w->ev_join[i]);
    }
    if (w->ev_fork) hipEventDestroy(w->ev_fork);
    if (w->arena) hipFree(w->arena);
    if (w->pinned) hipHostFree(w->pinned);
    if (w->stream) hipStreamDestroy(w->stream);
    delete w;
  }
  for (auto& pe : ctx->mel_pool) hipFree(pe.first);
  for (auto& kv : ctx->glow)
    if (kv.second->arena) hipFree(kv.second->arena);
  for (auto& kv : ctx->hifi) {
    if (kv.second->arena) hipFree(kv.second->arena);
    if (kv.second->bias_spec) hipFree(kv.second->bias_spec);
  }
  delete ctx;
}

static int copy_name(const std::string& s, char* name, int cap) {
  if (!name || cap <= 0) return 0;
  std::snprintf(name, (size_t)cap, "%s", s.c_str());
  return 0;
}

static int check_glow_hp(const mi355tts_glow_hparams* h) {
  if (!h) return fail(MI355TTS_ERR_INVALID, "hparams null");
  if (h->num_symbols <= 0 || h->hidden_channels <= 0 || h->n_heads <= 0 || h->hidden_channels % h->n_heads)
    return fail(MI355TTS_ERR_INVALID, "bad GlowTTS hparams");
  if (h->hidden_channels % 32)  // WaveNet res/skip rows are split on a 32-row tile boundary (all 51 shipped voices: 192)
    return fail(MI355TTS_ERR_INVALID, "hidden_channels %d must be a multiple of 32", h->hidden_channels);
  if (h->hidden_channels / h->n_heads > ATT_MAXDK) return fail(MI355TTS_ERR_INVALID, "head dim > %d unsupported", ATT_MAXDK);
  if (2 * h->window_size + 1 > ATT_MAXW) return fail(MI355TTS_ERR_INVALID, "window_size too large");
  if (h->n_split > 8 || h->n_split % 2 || (h->mel_channels * h->n_sqz) % h->n_split)
    return fail(MI355TTS_ERR_INVALID, "bad n_split");
  if (h->n_sqz < 1 || ((h->mel_channels * h->n_sqz) & 1)) return fail(MI355TTS_ERR_INVALID, "bad n_sqz");
  if (h->kernel_size != 1 && h->kernel_size != 3 && h->kernel_size != 5) return fail(MI355TTS_ERR_INVALID, "bad kernel_size");
  if (h->kernel_size_dec != 3 && h->kernel_size_dec != 5) return fail(MI355TTS_ERR_INVALID, "bad kernel_size_dec");
  return 0;
}
static int check_hifi_hp(const mi355tts_hifigan_hparams* h) {
  if (!h) return fail(MI355TTS_ERR_INVALID, "hparams null");
  if (h->num_upsamples < 1 || h->num_upsamples > MI355TTS_MAX_STAGES || h->num_kernels < 1 ||
      h->num_kernels > MI355TTS_MAX_STAGES || h->num_dilations < 1 || h->num_dilations > MI355TTS_MAX_STAGES)
    return fail(MI355TTS_ERR_INVALID, "bad HiFi-GAN hparams");
  if (h->resblock_type != 1 && h->resblock_type != 2) return fail(MI355TTS_ERR_INVALID, "resblock must be 1 or 2");
  for (int i = 0; i < h->num_upsamples; ++i) {
    const int u = h->upsample_rates[i], k = h->upsample_kernel_sizes[i];
    if (u < 1 || k % u || (k - u) % 2 || k / u > 3) return fail(MI355TTS_ERR_INVALID, "unsupported upsample (%d,%d)", u, k);
    if ((h->upsample_initial_channel >> (i + 1)) < 1) return fail(MI355TTS_ERR_INVALID, "too many upsample stages");
  }
  return 0;
}

extern "C" int mi355tts_glow_manifest(const mi355tts_glow_hparams* hp, int index, char* name, int cap, int64_t* numel) {
  CHECK(check_glow_hp(hp));
  auto m = glow_manifest(*hp);
  if (index < 0) return fail(MI355TTS_ERR_INVALID, "negative index");
  if ((size_t)index >= m.size()) return 1;
  copy_name(m[index].first, name, cap);
  if (numel) *numel = m[index].second;
  return 0;
}
extern "C" int mi355tts_hifigan_manifest(const mi355tts_hifigan_hparams* hp, int index, char* name, int cap, int64_t* numel) {
  CHECK(check_hifi_hp(hp));
  auto m = hifigan_manifest(*hp);
  if (index < 0) return fail(MI355TTS_ERR_INVALID, "negative index");
  if ((size_t)index >= m.size()) return 1;
  copy_name(m[index].first, name, cap);
  if (numel) *numel = m[index].second;
  return 0;
}

static int fetch_blob(mi355tts_ctx* ctx, const float* blob, int64_t numel, int on_device, std::vector<float>& host,
                      const float** p) {
  if (!on_device) {
    *p = blob;
    return 0;
  }
  HIPCHECK(hipSetDevice(ctx->device));
  host.resize((size_t)numel);
  HIPCHECK(hipMemcpy(host.data(), blob, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost));
  *p = host.data();
  return 0;
}

static int upload_arena(mi355tts_ctx* ctx, ArenaBuilder& ab, float** dev) {
  HIPCHECK(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dev, ab.host.size() * sizeof(float) + 256);
  if (e != hipSuccess) return fail(MI355TTS_ERR_NOMEM, "hipMalloc model arena: %s", hipGetErrorString(e));
  HIPCHECK(hipMemcpy(*dev, ab.host.data(), ab.host.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}
static void fix(DevConv& c, const float* arena) {
  c.w = arena + c.w_off;
  c.bias = c.has_bias ? arena + c.b_off : nullptr;
}

extern "C" int mi355tts_load_glow(mi355tts_ctx* ctx, const mi355tts_glow_hparams* hp, const float* blob, int64_t numel,
                                  int on_device, int* model_out) {
  if (!ctx || !blob || !model_out) return fail(MI355TTS_ERR_INVALID, "null argument");
  CHECK(check_glow_hp(hp));
  const mi355tts_glow_hparams& h = *hp;
  std::vector<float> tmp;
  Blob bl;
  CHECK(fetch_blob(ctx, blob, numel, on_device, tmp, &bl.p));
  bl.n = numel;
  bl.manifest = glow_manifest(h);
  int64_t total = 0;
  for (auto& kv : bl.manifest) total += kv.second;
  if (total != numel) return fail(MI355TTS_ERR_INVALID, "GlowTTS blob has %lld floats, manifest needs %lld", (long long)numel, (long long)total);

  auto gm = std::make_unique<GlowModel>();
  gm->hp = h;
  ArenaBuilder ab;
  const int H = h.hidden_channels, Fc = h.filter_channels, Fd = h.filter_channels_dp, M = h.mel_channels;
  const int k = h.kernel_size, dk = H / h.n_heads, nrel = 2 * h.window_size + 1;
#define TAKE(var, name, n)                         \
  const float* var = bl.take((name).c_str(), (n)); \
  if (!var) return MI355TTS_ERR_INVALID;
  {
    TAKE(emb, std::string("encoder.emb.weight"), (int64_t)h.num_symbols * H);
    gm->emb = ab.add(emb, (size_t)h.num_symbols * H);
  }
  if (h.prenet) {
    for (int i = 0; i < h.prenet_layers; ++i) {
      std::string p = "encoder.pre.conv_layers." + std::to_string(i);
      std::string q = "encoder.pre.norm_layers." + std::to_string(i);
      TAKE(w, p + ".weight", (int64_t)H * H * h.prenet_kernel_size);
      TAKE(b, p + ".bias", H);
      TAKE(g, q + ".gamma", H);
      TAKE(be, q + ".beta", H);
      gm->pre_conv.push_back(add_conv(ab, w, b, H, H, h.prenet_kernel_size, ROWS_PLAIN));
      gm->pre_g.push_back(ab.add(g, H));
      gm->pre_b.push_back(ab.add(be, H));
    }
    TAKE(w, std::string("encoder.pre.proj.weight"), (int64_t)H * H);
    TAKE(b, std::string("encoder.pre.proj.bias"), H);
    gm->pre_proj = add_conv(ab, w, b, H, H, 1, ROWS_PLAIN);
  }
  for (int l = 0; l < h.n_layers_enc; ++l) {
    GlowLayer L;
    std::string a = "encoder.encoder.attn_layers." + std::to_string(l);
    TAKE(ek, a + ".emb_rel_k", (int64_t)nrel * dk);
    TAKE(ev, a + ".emb_rel_v", (int64_t)nrel * dk);
    TAKE(wq, a + ".conv_q.weight", (int64_t)H * H);
    TAKE(bq, a + ".conv_q.bias", H);
    TAKE(wk, a + ".conv_k.weight", (int64_t)H * H);
    TAKE(bk, a + ".conv_k.bias", H);
    TAKE(wv, a + ".conv_v.weight", (int64_t)H * H);
    TAKE(bv, a + ".conv_v.bias", H);
    TAKE(wo, a + ".conv_o.weight", (int64_t)H * H);
    TAKE(bo, a + ".conv_o.bias", H);
    // q, k, v share their input: one GEMM with 3H output rows (attentions.py:205-207)
    std::vector<float> wqkv((size_t)3 * H * H), bqkv((size_t)3 * H);
    std::memcpy(wqkv.data(), wq, sizeof(float) * H * H);
    std::memcpy(wqkv.data() + (size_t)H * H, wk, sizeof(float) * H * H);
    std::memcpy(wqkv.data() + (size_t)2 * H * H, wv, sizeof(float) * H * H);
    std::memcpy(bqkv.data(), bq, sizeof(float) * H);
    std::memcpy(bqkv.data() + H, bk, sizeof(float) * H);
    std::memcpy(bqkv.data() + 2 * H, bv, sizeof(float) * H);
    L.qkv = add_conv(ab, wqkv.data(), bqkv.data(), 3 * H, H, 1, ROWS_PLAIN);
    L.o = add_conv(ab, wo, bo, H, H, 1, ROWS_PLAIN);
    L.ek = ab.add(ek, (size_t)nrel * dk);
    L.ev = ab.add(ev, (size_t)nrel * dk);
    TAKE(g1, "encoder.encoder.norm_layers_1." + std::to_string(l) + ".gamma", H);
    TAKE(b1, "encoder.encoder.norm_layers_1." + std::to_string(l) + ".beta", H);
    L.g1 = ab.add(g1, H);
    L.b1 = ab.add(b1, H);
    std::string f = "encoder.encoder.ffn_layers." + std::to_string(l);
    TAKE(w1, f + ".conv_1.weight", (int64_t)Fc * H * k);
    TAKE(c1, f + ".conv_1.bias", Fc);
    TAKE(w2, f + ".conv_2.weight", (int64_t)H * Fc * k);
    TAKE(c2, f + ".conv_2.bias", H);
    L.ffn1 = add_conv(ab, w1, c1, Fc, H, k, ROWS_PLAIN);
    L.ffn2 = add_conv(ab, w2, c2, H, Fc, k, ROWS_PLAIN);
    TAKE(g2, "encoder.encoder.norm_layers_2." + std::to_string(l) + ".gamma", H);
    TAKE(b2, "encoder.encoder.norm_layers_2." + std::to_string(l) + ".beta", H);
    L.g2 = ab.add(g2, H);
    L.b2 = ab.add(b2, H);
    gm->layers.push_back(L);
  }
  {
    TAKE(w, std::string("encoder.proj_m.weight"), (int64_t)M * H);
    TAKE(b, std::string("encoder.proj_m.bias"), M);
    gm->proj_m = add_conv(ab, w, b, M, H, 1, ROWS_PLAIN);
    TAKE(w1, std::string("encoder.proj_w.conv_1.weight"), (int64_t)Fd * H * k);
    TAKE(b1, std::string("encoder.proj_w.conv_1.bias"), Fd);
    TAKE(g1, std::string("encoder.proj_w.norm_1.gamma"), Fd);
    TAKE(e1, std::string("encoder.proj_w.norm_1.beta"), Fd);
    TAKE(w2, std::string("encoder.proj_w.conv_2.weight"), (int64_t)Fd * Fd * k);
    TAKE(b2, std::string("encoder.proj_w.conv_2.bias"), Fd);
    TAKE(g2, std::string("encoder.proj_w.norm_2.gamma"), Fd);
    TAKE(e2, std::string("encoder.proj_w.norm_2.beta"), Fd);
    TAKE(wp, std::string("encoder.proj_w.proj.weight"), Fd);
    TAKE(bp, std::string("encoder.proj_w.proj.bias"), 1);
    gm->dp1 = add_conv(ab, w1, b1, Fd, H, k, ROWS_PLAIN);
    gm->dp2 = add_conv(ab, w2, b2, Fd, Fd, k, ROWS_PLAIN);
    gm->dpp = add_conv(ab, wp, bp, 1, Fd, 1, ROWS_PLAIN);
    gm->dg1 = ab.add(g1, Fd);
    gm->db1 = ab.add(e1, Fd);
    gm->dg2 = ab.add(g2, Fd);
    gm->db2 = ab.add(e2, Fd);
  }
  const int C = M * h.n_sqz, half = C / 2;
  for (int b = 0; b < h.n_blocks_dec; ++b) {
    GlowBlock B;
    std::string an = "decoder.flows." + std::to_string(3 * b);
    std::string ic = "decoder.flows." + std::to_string(3 * b + 1);
    std::string cp = "decoder.flows." + std::to_string(3 * b + 2);
    TAKE(logs, an + ".logs", C);
    TAKE(abias, an + ".bias", C);
    TAKE(winv, ic + ".weight_inv", (int64_t)h.n_split * h.n_split);
    std::vector<float> scale(C);
    for (int c = 0; c < C; ++c) scale[c] = std::exp(-logs[c]);  // ActNorm reverse, layers.py:192-194
    B.an_bias = ab.add(abias, C);
    B.an_scale = ab.add(scale);
    B.winv = ab.add(winv, (size_t)h.n_split * h.n_split);
    TAKE(ws, cp + ".start.weight", (int64_t)H * half);
    TAKE(bs, cp + ".start.bias", H);
    B.start = add_conv(ab, ws, bs, H, half, 1, ROWS_PLAIN);
    for (int j = 0; j < h.n_block_layers; ++j) {
      std::string il = cp + ".wn.in_layers." + std::to_string(j);
      std::string rl = cp + ".wn.res_skip_layers." + std::to_string(j);
      const int rsn = (j < h.n_block_layers - 1) ? 2 * H : H;
      TAKE(wi, il + ".weight", (int64_t)2 * H * H * h.kernel_size_dec);
      TAKE(bi, il + ".bias", 2 * H);
      TAKE(wr, rl + ".weight", (int64_t)rsn * H);
      TAKE(br, rl + ".bias", rsn);
      B.in.push_back(add_conv(ab, wi, bi, 2 * H, H, h.kernel_size_dec, ROWS_PAIR, H));
      B.rs.push_back(add_conv(ab, wr, br, rsn, H, 1, ROWS_PLAIN));
    }
    TAKE(we, cp + ".end.weight", (int64_t)C * H);
    TAKE(be, cp + ".end.bias", C);
    B.end = add_conv(ab, we, be, C, H, 1, ROWS_PAIR, half);
    gm->blocks.push_back(std::move(B));
  }
#undef TAKE
  CHECK(upload_arena(ctx, ab, &gm->arena));
  const float* A = gm->arena;
  for (auto& c : gm->pre_conv) fix(c, A);
  if (h.prenet) fix(gm->pre_proj, A);
  for (auto& L : gm->layers) {
    fix(L.qkv, A);
    fix(L.o, A);
    fix(L.ffn1, A);
    fix(L.ffn2, A);
  }
  fix(gm->proj_m, A);
  fix(gm->dp1, A);
  fix(gm->dp2, A);
  fix(gm->dpp, A);
  for (auto& B : gm->blocks) {
    fix(B.start, A);
    fix(B.end, A);
    for (auto& c : B.in) fix(c, A);
    for (auto& c : B.rs) fix(c, A);
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int id = ctx->next_id++;
  ctx->glow[id] = std::move(gm);
  *model_out = id;
  return 0;
}

extern "C" int mi355tts_load_hifigan(mi355tts_ctx* ctx, const mi355tts_hifigan_hparams* hp, const float* blob,
                                     int64_t numel, int on_device, int* model_out) {
  if (!ctx || !blob || !model_out) return fail(MI355TTS_ERR_INVALID, "null argument");
  CHECK(check_hifi_hp(hp));
  const mi355tts_hifigan_hparams& h = *hp;
  std::vector<float> tmp;
  Blob bl;
  CHECK(fetch_blob(ctx, blob, numel, on_device, tmp, &bl.p));
  bl.n = numel;
  bl.manifest = hifigan_manifest(h);
  int64_t total = 0;
  for (auto& kv : bl.manifest) total += kv.second;
  if (total != numel) return fail(MI355TTS_ERR_INVALID, "HiFi-GAN blob has %lld floats, manifest needs %lld", (long long)numel, (long long)total);
  auto hm = std::make_unique<HifiModel>();
  hm->hp = h;
  ArenaBuilder ab;
  const int C0 = h.upsample_initial_channel;
#define TAKE(var, name, n)                         \
  const float* var = bl.take((name).c_str(), (n)); \
  if (!var) return MI355TTS_ERR_INVALID;
  {
    TAKE(w, std::string("conv_pre.weight"), (int64_t)C0 * h.num_mels * 7);
    TAKE(b, std::string("conv_pre.bias"), C0);
    hm->pre = add_conv(ab, w, b, C0, h.num_mels, 7, ROWS_PLAIN);
  }
  int ch = C0;
  hm->hop = 1;
  hm->rb.resize(h.num_upsamples);
  for (int i = 0; i < h.num_upsamples; ++i) {
    const int cin = C0 >> i, cout = C0 >> (i + 1);
    const int u = h.upsample_rates[i], ku = h.upsample_kernel_sizes[i];
    hm->hop *= u;
    TAKE(w, "ups." + std::to_string(i) + ".weight", (int64_t)cin * cout * ku);
    TAKE(b, "ups." + std::to_string(i) + ".bias", cout);
    hm->ups.push_back(add_conv(ab, w, b, cout, cin, ku, ROWS_UPSAMPLE, u));
    ch = cout;
    hm->rb[i].resize(h.num_kernels);
    for (int j = 0; j < h.num_kernels; ++j) {
      const int n = i * h.num_kernels + j;
      const int k = h.resblock_kernel_sizes[j];
      std::string rb = "resblocks." + std::to_string(n);
      for (int d = 0; d < h.num_dilations; ++d) {
        HifiResConv rc;
        rc.dil = h.resblock_dilations[j][d];
        if (h.resblock_type == 1) {
          TAKE(w1, rb + ".convs1." + std::to_string(d) + ".weight", (int64_t)ch * ch * k);
          TAKE(b1, rb + ".convs1." + std::to_string(d) + ".bias", ch);
          TAKE(w2, rb + ".convs2." + std::to_string(d) + ".weight", (int64_t)ch * ch * k);
          TAKE(b2, rb + ".convs2." + std::to_string(d) + ".bias", ch);
          rc.c1 = add_conv(ab, w1, b1, ch, ch, k, ROWS_PLAIN);
          rc.c2 = add_conv(ab, w2, b2, ch, ch, k, ROWS_PLAIN);
        } else {
          TAKE(w1, rb + ".convs." + std::to_string(d) + ".weight", (int64_t)ch * ch * k);
          TAKE(b1, rb + ".convs." + std::to_string(d) + ".bias", ch);
          rc.c1 = add_conv(ab, w1, b1, ch, ch, k, ROWS_PLAIN);
        }
        hm->rb[i][j].push_back(rc);
      }
    }
  }
  {
    TAKE(w, std::string("conv_post.weight"), (int64_t)ch * 7);
    TAKE(b, std::string("conv_post.bias"), 1);
    hm->post = add_conv(ab, w, b, 1, ch, 7, ROWS_PLAIN);
  }
#undef TAKE
  CHECK(upload_arena(ctx, ab, &hm->arena));
  const float* A = hm->arena;
  fix(hm->pre, A);
  fix(hm->post, A);
  for (auto& c : hm->ups) fix(c, A);
  for (auto& st : hm->rb)
    for (auto& kk : st)
      for (auto& rc : kk) {
        fix(rc.c1, A);
        if (h.resblock_type == 1) fix(rc.c2, A);
      }
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int id = ctx->next_id++;
  ctx->hifi[id] = std::move(hm);
  *model_out = id;
  return 0;
}

extern "C" int mi355tts_unload(mi355tts_ctx* ctx, int model) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto g = ctx->glow.find(model);
  if (g != ctx->glow.end()) {
    hipFree(g->second->arena);
    ctx->glow.erase(g);
    return 0;
  }
  auto v = ctx->hifi.find(model);
  if (v != ctx->hifi.end()) {
    hipFree(v->second->arena);
    if (v->second->bias_spec) hipFree(v->second->bias_spec);
    ctx->hifi.erase(v);
    return 0;
  }
  return fail(MI355TTS_ERR_NO_MODEL, "no model %d", model);
}

// ------------------------------------------------------------------ mel objects
static void* pool_alloc(mi355tts_ctx* ctx, size_t bytes) {
  bytes = (bytes + 4095) & ~(size_t)4095;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    int best = -1;
    for (int i = 0; i < (int)ctx->mel_pool.size(); ++i)
      if (ctx->mel_pool[i].second >= bytes && ctx->mel_pool[i].second <= 2 * bytes + (1 << 16) &&
          (best < 0 || ctx->mel_pool[i].second < ctx->mel_pool[best].second))
        best = i;
    if (best >= 0) {
      void* p = ctx->mel_pool[best].first;
      ctx->mel_pool.erase(ctx->mel_pool.begin() + best);
      return p;
    }
  }
  void* p = nullptr;
  // the block remembers its size in the pool entry when it comes back; keep a header-free
  // scheme by rounding deterministically (see pool_free)
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}
static void pool_free(mi355tts_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  bytes = (bytes + 4095) & ~(size_t)4095;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (ctx->mel_pool.size() < 96) {
    ctx->mel_pool.emplace_back(p, bytes);
    return;
  }
  hipFree(p);
}
static size_t mel_bytes(const mi355tts_mel* m) { return (size_t)m->B * m->M * (size_t)std::max(m->ld, 1) * sizeof(float); }
static void mel_destroy(mi355tts_mel* m) {
  if (!m) return;
  pool_free(m->ctx, m->raw, m->raw_bytes);
  pool_free(m->ctx, m->voc, m->raw_bytes);
  pool_free(m->ctx, m->frames_dev, sizeof(int) * (size_t)m->B);
  delete m;
}
extern "C" void mi355tts_mel_free(mi355tts_mel* m) { mel_destroy(m); }
extern "C" int mi355tts_mel_batch(const mi355tts_mel* m) { return m ? m->B : fail(MI355TTS_ERR_INVALID, "mel null"); }
extern "C" int mi355tts_mel_channels(const mi355tts_mel* m) { return m ? m->M : fail(MI355TTS_ERR_INVALID, "mel null"); }
extern "C" int mi355tts_mel_max_frames(const mi355tts_mel* m) { return m ? m->max_frames : fail(MI355TTS_ERR_INVALID, "mel null"); }
extern "C" int mi355tts_mel_frames(const mi355tts_mel* m, int32_t* frames) {
  if (!m || !frames) return fail(MI355TTS_ERR_INVALID, "null argument");
  for (int b = 0; b < m->B; ++b) frames[b] = m->frames[b];
  return 0;
}
extern "C" int mi355tts_mel_copy(const mi355tts_mel* m, int which, float* dst, int ld) {
  if (!m || !dst) return fail(MI355TTS_ERR_INVALID, "null argument");
  if (ld < m->max_frames) return fail(MI355TTS_ERR_TOO_SMALL, "ld %d < max_frames %d", ld, m->max_frames);
  if (m->max_frames == 0) return 0;
  HIPCHECK(hipSetDevice(m->ctx->device));
  const float* src = which == 0 ? m->raw : m->voc;
  std::vector<float> tmp((size_t)m->B * m->M * m->ld);
  HIPCHECK(hipMemcpy(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int r = 0; r < m->B * m->M; ++r) {
    std::memcpy(dst + (size_t)r * ld, tmp.data() + (size_t)r * m->ld, sizeof(float) * m->max_frames);
    for (int t = m->max_frames; t < ld; ++t) dst[(size_t)r * ld + t] = 0.f;
  }
  return 0;
}

static int mel_alloc(mi355tts_ctx* ctx, int B, int M, int ld, mi355tts_mel** out) {
  auto* m = new mi355tts_mel();
  m->ctx = ctx;
  m->B = B;
  m->M = M;
  m->ld = ld;
  m->frames.assign(B, 0);
  const size_t n = (size_t)B * M * std::max(ld, 1) * sizeof(float);
  m->raw_bytes = n;
  m->raw = (float*)pool_alloc(ctx, n);
  m->voc = (float*)pool_alloc(ctx, n);
  m->frames_dev = (int*)pool_alloc(ctx, sizeof(int) * B);
  if (!m->raw || !m->voc || !m->frames_dev) {
    mel_destroy(m);
    return fail(MI355TTS_ERR_NOMEM, "hipMalloc mel");
  }
  *out = m;
  return 0;
}

extern "C" int mi355tts_mel_from_buffer(mi355tts_ctx* ctx, const float* mel, const int32_t* frames, int B, int M, int ld,
                                        const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out) {
  if (!ctx || !mel || !frames || !out || B <= 0 || M <= 0 || ld < 0) return fail(MI355TTS_ERR_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  int mx = 0;
  for (int b = 0; b < B; ++b) {
    if (frames[b] < 0 || frames[b] > ld) return fail(MI355TTS_ERR_INVALID, "frames[%d]=%d outside [0,%d]", b, frames[b], ld);
    mx = std::max(mx, frames[b]);
  }
  mi355tts_mel* m = nullptr;
  const int ldp = (ld + 3) & ~3;
  CHECK(mel_alloc(ctx, B, M, ldp, &m));
  m->max_frames = mx;
  for (int b = 0; b < B; ++b) m->frames[b] = frames[b];
  Worker* w = nullptr;
  int rc = acquire_worker(ctx, &w);
  if (rc) {
    mel_destroy(m);
    return rc;
  }
  WorkerGuard guard{ctx, w};
  const size_t n = (size_t)B * M * ldp;
  hipError_t e = hipMemcpyAsync(m->frames_dev, frames, sizeof(int) * B, hipMemcpyHostToDevice, w->stream);
  if (e == hipSuccess && n) e = hipMemsetAsync(m->raw, 0, n * sizeof(float), w->stream);
  if (e == hipSuccess && n)
    e = hipMemcpy2DAsync(m->raw, sizeof(float) * ldp, mel, sizeof(float) * ld, sizeof(float) * ld, (size_t)B * M,
                         (flags & MI355TTS_IN_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, w->stream);
  if (e == hipSuccess && n) {
    if (audio) {
      hipLaunchKernelGGL(mel_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w->stream, m->raw, m->voc,
                         (long long)n, to_mt(audio));
    } else {
      e = hipMemcpyAsync(m->voc, m->raw, n * sizeof(float), hipMemcpyDeviceToDevice, w->stream);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
  if (e != hipSuccess) {
    mel_destroy(m);
    return fail(MI355TTS_ERR_HIP, "mel_from_buffer: %s", hipGetErrorString(e));
  }
  *out = m;
  return 0;
}

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

extern "C" int mi355tts_glow_infer(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                                   float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                                   const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out) {
  if (!ctx || !ids || !id_lens || !out) return fail(MI355TTS_ERR_INVALID, "null argument");
  if (B <= 0 || ids_ld <= 0) return fail(MI355TTS_ERR_INVALID, "empty batch");
  const GlowModel* gm;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->glow.find(glow);
    if (it == ctx->glow.end()) return fail(MI355TTS_ERR_NO_MODEL, "no GlowTTS model %d", glow);
    gm = it->second.get();
  }
  const mi355tts_glow_hparams& h = gm->hp;
  int Pmax = 0;
  for (int b = 0; b < B; ++b) {
    if (id_lens[b] < 1 || id_lens[b] > ids_ld) return fail(MI355TTS_ERR_INVALID, "id_lens[%d]=%d outside [1,%d]", b, id_lens[b], ids_ld);
    Pmax = std::max(Pmax, id_lens[b]);
  }
  const bool in_dev_ids = (flags & MI355TTS_IN_DEVICE) != 0;
  if (!in_dev_ids) {
    // the reference's embedding lookup raises on an out-of-range id (glow_tts/models.py:119);
    // device-resident ids cannot be checked without a sync and are clamped by the kernel instead
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < id_lens[b]; ++t) {
        const int64_t id = ids[(size_t)b * ids_ld + t];
        if (id < 0 || id >= h.num_symbols)
          return fail(MI355TTS_ERR_INVALID, "phoneme id %lld at [%d][%d] outside [0,%d)", (long long)id, b, t, h.num_symbols);
      }
  }
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  hipStream_t s = w->stream;
  const float* A = gm->arena;
  const int H = h.hidden_channels, Fc = h.filter_channels, Fd = h.filter_channels_dp, M = h.mel_channels;
  const int k = h.kernel_size, nh = h.n_heads;
  const int P = (Pmax + 3) & ~3;  // row stride
  const bool in_dev = (flags & MI355TTS_IN_DEVICE) != 0;
  const int enc_host_len = B == 1 ? id_lens[0] : -1;

  // ---- encoder workspace
  Carver cv;
  const size_t o_len = cv.take(sizeof(int) * B);
  const size_t o_ids = cv.take(sizeof(long long) * (size_t)B * ids_ld);
  const size_t o_x = cv.take(sizeof(float) * (size_t)B * H * P);
  const size_t o_t1 = cv.take(sizeof(float) * (size_t)B * H * P);
  const size_t o_t2 = cv.take(sizeof(float) * (size_t)B * H * P);
  const size_t o_qkv = cv.take(sizeof(float) * (size_t)B * 3 * H * P);
  const size_t o_ffn = cv.take(sizeof(float) * (size_t)B * std::max(Fc, 2 * Fd) * P);
  const size_t o_xm = cv.take(sizeof(float) * (size_t)B * M * P);
  const size_t o_logw = cv.take(sizeof(float) * (size_t)B * P);
  const size_t o_cum = cv.take(sizeof(int) * (size_t)B * P);
  const int att_rows = ((Pmax + ATT_ROWS - 1) / ATT_ROWS) * ATT_ROWS;
  const size_t o_sc = cv.take(sizeof(float) * (size_t)B * nh * att_rows * P);
  const size_t enc_bytes = cv.pos;
  CHECK(reserve(w, enc_bytes));
  char* base = w->arena;
  int* d_len = (int*)(base + o_len);
  long long* d_ids = (long long*)(base + o_ids);
  float* x = (float*)(base + o_x);
  float* t1 = (float*)(base + o_t1);
  float* t2 = (float*)(base + o_t2);
  float* qkv = (float*)(base + o_qkv);
  float* ffn = (float*)(base + o_ffn);
  float* xm = (float*)(base + o_xm);
  float* logw = (float*)(base + o_logw);
  int* cum = (int*)(base + o_cum);
  float* sc = (float*)(base + o_sc);

  HIPCHECK(hipMemcpyAsync(d_len, id_lens, sizeof(int) * B, hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(d_ids, ids, sizeof(long long) * (size_t)B * ids_ld, in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));

  const long long bsH = (long long)H * P;
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(embed_kernel, dim3((Pmax + 63) / 64, 8, B), dim3(256), 0, s, d_ids, ids_ld, d_len, A + gm->emb,
                       h.num_symbols, H, std::sqrt((float)H), x, bsH, P);
  }
  if (h.prenet) {
    // ConvReluNorm: conv -> LayerNorm -> ReLU (x3), then x + proj(.)  (layers.py:73-80)
    const float* cur = x;
    for (int i = 0; i < h.prenet_layers; ++i) {
      ConvArgs a = base_args(cur, bsH, P, d_len, 1, t1, bsH, P, d_len, 1, 1, h.prenet_kernel_size / 2);
      CHECK(launch_conv(ctx, w, gm->pre_conv[i], a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
      ProfScope ps(ctx, w, KC_SMALL, 0);
      run_layernorm(w, t1, nullptr, A + gm->pre_g[i], A + gm->pre_b[i], t2, H, bsH, P, d_len, B, Pmax, 1);
      cur = t2;
    }
    ConvArgs a = base_args(cur, bsH, P, d_len, 1, x, bsH, P, d_len, 1, 1, 0);
    a.res = x;
    CHECK(launch_conv(ctx, w, gm->pre_proj, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
  }
  for (int l = 0; l < h.n_layers_enc; ++l) {  // Encoder.forward, attentions.py:62-74
    const GlowLayer& L = gm->layers[l];
    {
      ConvArgs a = base_args(x, bsH, P, d_len, 1, qkv, 3 * bsH, P, d_len, 1, 1, 0);
      CHECK(launch_conv(ctx, w, L.qkv, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
    }
    {
      ProfScope ps(ctx, w, KC_SMALL, 0);
      const dim3 ag((Pmax + 31) / 32, nh, B);
      const int dkh = H / nh;
#define ATT_LAUNCH(NK)                                                                                                   \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(attention_mfma_kernel<NK>), ag, dim3(256), 0, s, qkv, 3 * bsH, P, d_len, H, nh,     \
                     h.window_size, A + L.ek, A + L.ev, t2, bsH, P)
      if (Pmax <= ATTM_MAXP && dkh <= 32) ATT_LAUNCH(16);
      else if (Pmax <= ATTM_MAXP && dkh <= 64) ATT_LAUNCH(32);
      else if (Pmax <= ATTM_MAXP && dkh <= 96) ATT_LAUNCH(48);
      else if (Pmax <= ATTM_MAXP) ATT_LAUNCH(64);
#undef ATT_LAUNCH
      else
        hipLaunchKernelGGL(attention_kernel, dim3(att_rows / ATT_ROWS, nh, B), dim3(256), 0, s, qkv, 3 * bsH, P, d_len, H, nh,
                           h.window_size, A + L.ek, A + L.ev, t2, bsH, P, sc, P);
    }
    {
      ConvArgs a = base_args(t2, bsH, P, d_len, 1, t1, bsH, P, d_len, 1, 1, 0);
      a.res = x;
      CHECK(launch_conv(ctx, w, L.o, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
      ProfScope ps(ctx, w, KC_SMALL, 0);
      run_layernorm(w, t1, nullptr, A + L.g1, A + L.b1, x, H, bsH, P, d_len, B, Pmax, 0);
    }
    {  // FFN, attentions.py:375-383
      ConvArgs a = base_args(x, bsH, P, d_len, 1, ffn, (long long)Fc * P, P, d_len, 1, 1, k / 2);
      a.out_act = ACT_RELU;
      CHECK(launch_conv(ctx, w, L.ffn1, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
      ConvArgs c = base_args(ffn, (long long)Fc * P, P, d_len, 1, t1, bsH, P, d_len, 1, 1, k / 2);
      c.res = x;
      CHECK(launch_conv(ctx, w, L.ffn2, c, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
      ProfScope ps(ctx, w, KC_SMALL, 0);
      run_layernorm(w, t1, nullptr, A + L.g2, A + L.b2, x, H, bsH, P, d_len, B, Pmax, 0);
    }
  }
  {  // proj_m and the duration predictor (models.py:133-139, 39-49)
    ConvArgs a = base_args(x, bsH, P, d_len, 1, xm, (long long)M * P, P, d_len, 1, 1, 0);
    CHECK(launch_conv(ctx, w, gm->proj_m, a, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
    float* d1 = ffn;
    float* d2 = ffn + (size_t)B * Fd * P;
    const long long bsD = (long long)Fd * P;
    ConvArgs c1 = base_args(x, bsH, P, d_len, 1, d1, bsD, P, d_len, 1, 1, k / 2);
    c1.out_act = ACT_RELU;
    CHECK(launch_conv(ctx, w, gm->dp1, c1, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
    {
      ProfScope ps(ctx, w, KC_SMALL, 0);
      run_layernorm(w, d1, nullptr, A + gm->dg1, A + gm->db1, d2, Fd, bsD, P, d_len, B, Pmax, 0);
    }
    ConvArgs c2 = base_args(d2, bsD, P, d_len, 1, d1, bsD, P, d_len, 1, 1, k / 2);
    c2.out_act = ACT_RELU;
    CHECK(launch_conv(ctx, w, gm->dp2, c2, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
    {
      ProfScope ps(ctx, w, KC_SMALL, 0);
      run_layernorm(w, d1, nullptr, A + gm->dg2, A + gm->db2, d2, Fd, bsD, P, d_len, B, Pmax, 0);
    }
    ConvArgs c3 = base_args(d2, bsD, P, d_len, 1, logw, P, P, d_len, 1, 1, 0);
    CHECK(launch_conv(ctx, w, gm->dpp, c3, EPI_LINEAR, B, Pmax, KC_GLOW_ENC_CONV, nullptr, 1024, enc_host_len));
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
  struct MelGuard {
    mi355tts_mel* m;
    ~MelGuard() { mel_destroy(m); }
  } mguard{mel};
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(duration_kernel, dim3(B), dim3(64), 0, s, logw, (long long)P, d_len, length_scale, h.n_sqz, cum, P,
                       mel->frames_dev, 1 << 28);
  }
  if ((size_t)B > w->pinned_ints) return fail(MI355TTS_ERR_INVALID, "batch too large");
  HIPCHECK(hipMemcpyAsync(w->pinned, mel->frames_dev, sizeof(int) * B, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  int Fmax = 0;
  for (int b = 0; b < B; ++b) {
    mel->frames[b] = w->pinned[b];
    Fmax = std::max(Fmax, w->pinned[b]);
  }
  if (noise && noise_scale != 0.f && noise_ld < Fmax)
    return fail(MI355TTS_ERR_TOO_SMALL, "noise has %d columns but the utterance needs %d frames", noise_ld, Fmax);
  mel->max_frames = Fmax;
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
  Carver dv;
  dv.pos = enc_bytes;
  const size_t o_z = dv.take(sizeof(float) * (size_t)B * C * F2);
  const size_t o_h = dv.take(sizeof(float) * (size_t)B * H * F2);
  const size_t o_ac = dv.take(sizeof(float) * (size_t)B * H * F2);
  const size_t o_sk = dv.take(sizeof(float) * (size_t)B * H * F2);
  const size_t o_nz = dv.take((noise && !in_dev) ? sizeof(float) * (size_t)B * M * noise_ld : 0);
  if (dv.pos > w->arena_bytes) {
    // growing would move the encoder buffers: stage the three still-live encoder
    // outputs (x_m, cum, len) through a fresh arena instead
    std::vector<char> keep(enc_bytes);
    HIPCHECK(hipMemcpy(keep.data(), w->arena, enc_bytes, hipMemcpyDeviceToHost));
    CHECK(reserve(w, dv.pos));
    HIPCHECK(hipMemcpy(w->arena, keep.data(), enc_bytes, hipMemcpyHostToDevice));
    base = w->arena;
    d_len = (int*)(base + o_len);
    xm = (float*)(base + o_xm);
    cum = (int*)(base + o_cum);
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
                       d_len, cum, P, d_frames, d_noise, (long long)M * noise_ld, noise_ld, noise_scale, seed, M, nsq, z,
                       bsZ, F2);
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
  for (int blk = h.n_blocks_dec - 1; blk >= 0; --blk) {  // models.py:195-206, reversed flows
    const GlowBlock& Bk = gm->blocks[blk];
    {  // CouplingBlock reverse (attentions.py:119-142): h = start(x0)
      ConvArgs a = base_args(z, bsZ, F2, d_f2, 1, hbuf, bsD, F2, d_f2, 1, 1, 0);
      CHECK(launch_conv(ctx, w, Bk.start, a, EPI_LINEAR, B, F2max, KC_GLOW_DEC_CONV, nullptr, 1024, dec_host_len));
    }
    int dil = 1;
    for (int j = 0; j < h.n_block_layers; ++j) {  // WN.forward, layers.py:138-162
      const int kd = h.kernel_size_dec;
      ConvArgs a = base_args(hbuf, bsD, F2, d_f2, 1, acts, bsD, F2, d_f2, 1, dil, (kd * dil - dil) / 2);
      a.half = H;
      CHECK(launch_conv(ctx, w, Bk.in[j], a, EPI_GATE, B, F2max, KC_GLOW_DEC_CONV, nullptr, 1024, dec_host_len));
      ConvArgs r = base_args(acts, bsD, F2, d_f2, 1, hbuf, bsD, F2, d_f2, 1, 1, 0);
      if (j < h.n_block_layers - 1) {
        r.res = hbuf;  // x = x + res_skip[:H]
        r.split = H;
      } else {
        r.split = 0;  // last layer: everything is skip
      }
      r.y2 = skip;
      r.y2_bs = bsD;
      r.y2_ld = F2;
      r.accum2 = j > 0;
      CHECK(launch_conv(ctx, w, Bk.rs[j], r, EPI_LINEAR, B, F2max, KC_GLOW_DEC_CONV, nullptr, 1024, dec_host_len));
      dil *= h.dilation_rate;
    }
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
      CHECK(launch_conv(ctx, w, Bk.end, a, EPI_COUPLING, B, F2max, KC_GLOW_DEC_CONV, nullptr, 1024, dec_host_len));
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
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  mguard.m = nullptr;
  *out = mel;
  return 0;
}

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
  if (hm->bias_ready) return 0;
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
      if (hipStreamSynchronize(w->stream) != hipSuccess) rc = fail(MI355TTS_ERR_HIP, "denoiser bias kernel failed");
    }
  }
  mel_destroy(zm);
  if (dwav) hipFree(dwav);
  if (rc) {
    if (bias) hipFree(bias);
    return rc;
  }
  hm->bias_spec = bias;
  hm->bias_ready = true;
  return 0;
}

extern "C" int mi355tts_hifigan_infer(mi355tts_ctx* ctx, int vocoder, const mi355tts_mel* mel, float denoiser_strength,
                                      float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags) {
  if (!ctx || !mel) return fail(MI355TTS_ERR_INVALID, "null argument");
  HifiModel* hm;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->hifi.find(vocoder);
    if (it == ctx->hifi.end()) return fail(MI355TTS_ERR_NO_MODEL, "no HiFi-GAN model %d", vocoder);
    hm = it->second.get();
  }
  const mi355tts_hifigan_hparams& h = hm->hp;
  if (mel->M != h.num_mels) return fail(MI355TTS_ERR_INVALID, "mel has %d channels, vocoder expects %d", mel->M, h.num_mels);
  const int B = mel->B, F = mel->max_frames, hop = hm->hop;
  const long long N = (long long)F * hop;
  if (wav_ld < N) return fail(MI355TTS_ERR_TOO_SMALL, "wav_ld %lld < %lld samples", (long long)wav_ld, N);
  const bool denoise = denoiser_strength > 0.f && F > 0;
  if (denoise) {
    // the reference's STFT needs more than one 1024-sample frame per utterance
    // (larynx/audio.py:232-249 raises on shorter input)
    for (int b = 0; b < B; ++b)
      if ((long long)mel->frames[b] * hop <= DN_FFT)
        return fail(MI355TTS_ERR_INVALID, "utterance %d has %d frames: too short for the denoiser", b, mel->frames[b]);
    CHECK(ensure_denoiser_bias(ctx, hm, vocoder));
  }
  const bool out_dev = (flags & MI355TTS_OUT_DEVICE) != 0;
  if (F == 0) {
    if (!out_dev) {
      if (wav_f32) std::memset(wav_f32, 0, sizeof(float) * (size_t)B * wav_ld);
      if (wav_i16) std::memset(wav_i16, 0, sizeof(int16_t) * (size_t)B * wav_ld);
    }
    return 0;
  }
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  hipStream_t s = w->stream;
  const int C0 = h.upsample_initial_channel;
  // largest [C][L] plane over conv_pre and the stages
  const int Fp = (F + 3) & ~3;  // row strides are multiples of 4 floats (16-byte staging loads)
  size_t plane = (size_t)C0 * Fp;
  {
    long long L = F;
    for (int i = 0; i < h.num_upsamples; ++i) {
      L *= h.upsample_rates[i];
      plane = std::max(plane, (size_t)(C0 >> (i + 1)) * (size_t)L);
    }
  }
  const size_t Nld = (size_t)((N + 3) & ~3LL);
  const int nk = h.num_kernels;
  // The nk ResBlock chains of a stage are independent (MRF): run them on separate
  // streams so their workgroups interleave — at batch 1 one conv launch has fewer
  // tiles than the chip has SIMDs.  Each chain writes its own output; the average
  // is taken by the consumer's staging load.
  const bool concurrent = !ctx->serial_branches && nk >= 2 && nk <= 3;
  if (concurrent && !w->aux[0]) {
    for (int i = 0; i < 2; ++i) {
      HIPCHECK(hipStreamCreateWithFlags(&w->aux[i], hipStreamNonBlocking));
      HIPCHECK(hipEventCreateWithFlags(&w->ev_join[i], hipEventDisableTiming));
    }
    HIPCHECK(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
  }
  // concurrent chains share the chip: 300 tiles per launch measured best (sweeps of 80..1024,
  // also per-chain values, in round 1: 6.6 ms vs 7.05 ms per utterance at 1024)
  // tuning knob: MI355TTS_RB_TILES overrides the per-chain workgroup target of the concurrent schedule
  static const int rb_env = [] { const char* e = std::getenv("MI355TTS_RB_TILES"); return e ? std::atoi(e) : 0; }();
  const int rb_tiles = concurrent ? (rb_env > 0 ? rb_env : 300) : 1024;
  const int voc_host_len = B == 1 ? mel->frames[0] : -1;
  const int nbuf = concurrent ? 2 + 4 * nk : 6;
  Carver cv;
  size_t o_buf[16];
  for (int i = 0; i < nbuf; ++i) o_buf[i] = cv.take(sizeof(float) * (size_t)B * plane);
  const size_t o_wav = cv.take(sizeof(float) * (size_t)B * Nld);
  const size_t o_i16 = cv.take(sizeof(short) * (size_t)B * Nld);
  const size_t o_peak = cv.take(sizeof(unsigned) * B);
  const int Tmax = denoise ? (int)((N - DN_FFT + DN_HOP - 1) / DN_HOP) : 0;
  const size_t o_wav2 = cv.take(denoise ? sizeof(float) * (size_t)B * Nld : 0);
  const size_t o_fbuf = cv.take(denoise ? sizeof(float) * (size_t)B * Tmax * DN_FFT : 0);
  CHECK(reserve(w, cv.pos));
  char* base = w->arena;
  float* buf[16];
  for (int i = 0; i < nbuf; ++i) buf[i] = (float*)(base + o_buf[i]);
  float* wav = (float*)(base + o_wav);
  short* i16 = (short*)(base + o_i16);
  unsigned* peak = (unsigned*)(base + o_peak);
  const int* d_frames = mel->frames_dev;

  // stage input: `cur[0]` alone, or the nk chain outputs cur[0..nk) still to be averaged
  float* cur[3] = {buf[0], nullptr, nullptr};
  int ncur = 1;
  float* xu = buf[1];
  {  // conv_pre (models.py:187)
    ConvArgs a = base_args(mel->voc, (long long)mel->M * mel->ld, mel->ld, d_frames, 1, cur[0], (long long)C0 * Fp, Fp, d_frames, 1, 1, 3);
    CHECK(launch_conv(ctx, w, hm->pre, a, EPI_LINEAR, B, F, KC_VOC_IO, nullptr, 1024, voc_host_len));
  }
  auto set_inputs = [&](ConvArgs& a) {
    if (ncur > 1) {
      a.x2 = cur[1];
      a.x3 = ncur > 2 ? cur[2] : nullptr;
      a.in_div = (float)ncur;
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
    {  // x = ups[i](leaky_relu(x, 0.1))  (models.py:189-190)
      ConvArgs a = base_args(cur[0], (long long)ch * ldin, ldin, d_frames, mul, xu, (long long)cout * Lout, Lout, d_frames, mul * u, 1, ku / u - 1);
      set_inputs(a);
      a.in_slope = 0.1f;
      a.up = u;
      a.up_pad = (ku - u) / 2;
      CHECK(launch_conv(ctx, w, hm->ups[i], a, EPI_UPSAMPLE, B, Lin + ku / u - 1, KC_UPSAMPLE, nullptr, 1024, voc_host_len));
    }
    mul *= u;
    ch = cout;
    const long long bs = (long long)ch * Lout;
    const float inv_nk = 1.0f / (float)nk;
    if (concurrent) {
      HIPCHECK(hipEventRecord(w->ev_fork, s));
      for (int j = 1; j < nk; ++j) HIPCHECK(hipStreamWaitEvent(w->aux[j - 1], w->ev_fork, 0));
    }
    float* outs[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < nk; ++j) {  // MRF: resblocks on the same input (models.py:191-197)
      const int kk = h.resblock_kernel_sizes[j];
      hipStream_t sj = (concurrent && j > 0) ? w->aux[j - 1] : s;
      float *tb, *pa, *pb, *dst_last;
      if (concurrent) {
        // per-chain scratch: buf[2 + 4j .. 2 + 4j + 3] = {t, ping, out(flip 0), out(flip 1)}
        tb = buf[2 + 4 * j];
        pa = buf[2 + 4 * j + 1];
        pb = buf[2 + 4 * j + 2 + (flip ^ 1)];  // last stage's output: dead once the upsampler (before the fork) has read it
        dst_last = buf[2 + 4 * j + 2 + flip];
      } else {
        tb = buf[2];
        pa = buf[3];
        pb = buf[4];
        dst_last = buf[5];
      }
      outs[j] = dst_last;
      const float* rin = xu;
      for (int d = 0; d < h.num_dilations; ++d) {
        const HifiResConv& rc = hm->rb[i][j][d];
        const bool last = d == h.num_dilations - 1;
        float* dst = last ? dst_last : ((d & 1) ? pb : pa);
        if (!dst) return fail(MI355TTS_ERR_INVALID, "internal: resblock scratch aliasing");
        if (h.resblock_type == 1) {  // ResBlock1.forward, models.py:91-98
          {
            const float pa_alpha = (last && !concurrent) ? inv_nk : 1.0f;
            const int pa_accum = (last && !concurrent) ? (j > 0) : 0;
            const int fr = launch_pair(ctx, w, rc.c1, rc.c2, rin, dst, bs, Lout, d_frames, mul, rc.dil, pa_alpha, pa_accum, B, Lout, sj, voc_host_len);
            if (fr < 0) return fr;
            if (fr == 0) {
              rin = dst;
              continue;
            }
          }
          ConvArgs a = base_args(rin, bs, Lout, d_frames, mul, tb, bs, Lout, d_frames, mul, rc.dil, (kk * rc.dil - rc.dil) / 2);
          a.in_slope = 0.1f;
          CHECK(launch_conv(ctx, w, rc.c1, a, EPI_LINEAR, B, Lout, KC_RESBLOCK, sj, rb_tiles, voc_host_len));
          ConvArgs c = base_args(tb, bs, Lout, d_frames, mul, dst, bs, Lout, d_frames, mul, 1, (kk - 1) / 2);
          c.in_slope = 0.1f;
          c.res = rin;
          if (last && !concurrent) {
            c.alpha = inv_nk;
            c.accum = j > 0;
          }
          CHECK(launch_conv(ctx, w, rc.c2, c, EPI_LINEAR, B, Lout, KC_RESBLOCK, sj, rb_tiles, voc_host_len));
        } else {  // ResBlock2.forward, models.py:136-141
          ConvArgs a = base_args(rin, bs, Lout, d_frames, mul, dst, bs, Lout, d_frames, mul, rc.dil, (kk * rc.dil - rc.dil) / 2);
          a.in_slope = 0.1f;
          a.res = rin;
          if (last && !concurrent) {
            a.alpha = inv_nk;
            a.accum = j > 0;
          }
          CHECK(launch_conv(ctx, w, rc.c1, a, EPI_LINEAR, B, Lout, KC_RESBLOCK, sj, rb_tiles, voc_host_len));
        }
        rin = dst;
      }
    }
    if (concurrent) {
      for (int j = 1; j < nk; ++j) {
        HIPCHECK(hipEventRecord(w->ev_join[j - 1], w->aux[j - 1]));
        HIPCHECK(hipStreamWaitEvent(s, w->ev_join[j - 1], 0));
      }
      for (int j = 0; j < nk; ++j) cur[j] = outs[j];
      ncur = nk;
      flip ^= 1;
    } else {
      // serial: buf[5] holds the averaged sum; rotate it with the stage-input buffer
      std::swap(buf[5], buf[0]);
      cur[0] = buf[0];
      ncur = 1;
    }
    Lin = Lout;
    ldin = Lout;
  }
  {  // x = tanh(conv_post(leaky_relu(x)))  — default slope 0.01 (models.py:198-200)
    ConvArgs a = base_args(cur[0], (long long)ch * ldin, ldin, d_frames, mul, wav, (long long)Nld, (int)Nld, d_frames, mul, 1, 3);
    set_inputs(a);
    a.in_slope = 0.01f;
    a.out_act = ACT_TANH;
    CHECK(launch_conv(ctx, w, hm->post, a, EPI_LINEAR, B, Lin, KC_VOC_IO, nullptr, 1024, voc_host_len));
  }
  if (denoise) {  // HiFiGanVocoder.denoise (larynx/hifi_gan.py:171-179)
    ProfScope ps(ctx, w, KC_SMALL, 0);
    float* wav2 = (float*)(base + o_wav2);
    float* fbuf = (float*)(base + o_fbuf);
    hipLaunchKernelGGL(stft_denoise_kernel, dim3(Tmax, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, hm->bias_spec,
                       denoiser_strength, fbuf, Tmax, (float*)nullptr);
    hipLaunchKernelGGL(overlap_add_kernel, dim3(256, B), dim3(256), 0, s, fbuf, Tmax, d_frames, hop, wav2, (long long)Nld,
                       (long long)Nld);
    wav = wav2;
  }
  {
    ProfScope ps(ctx, w, KC_SMALL, 0);
    hipLaunchKernelGGL(zero_tail_kernel, dim3(64, B), dim3(256), 0, s, wav, (long long)Nld, (long long)Nld, d_frames, hop);
    if (wav_i16) {
      HIPCHECK(hipMemsetAsync(peak, 0, sizeof(unsigned) * B, s));
      hipLaunchKernelGGL(absmax_kernel, dim3(128, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, peak);
      hipLaunchKernelGGL(to_int16_kernel, dim3(128, B), dim3(256), 0, s, wav, (long long)Nld, d_frames, hop, peak, i16,
                         (long long)Nld, (long long)Nld);
    }
  }
  const hipMemcpyKind kind = out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  for (int b = 0; b < B; ++b) {
    if (wav_f32) {
      HIPCHECK(hipMemcpyAsync(wav_f32 + (size_t)b * wav_ld, wav + (size_t)b * Nld, sizeof(float) * (size_t)N, kind, s));
      if (wav_ld > N) {
        if (out_dev) HIPCHECK(hipMemsetAsync(wav_f32 + (size_t)b * wav_ld + N, 0, sizeof(float) * (size_t)(wav_ld - N), s));
      }
    }
    if (wav_i16) {
      HIPCHECK(hipMemcpyAsync(wav_i16 + (size_t)b * wav_ld, i16 + (size_t)b * Nld, sizeof(short) * (size_t)N, kind, s));
      if (wav_ld > N) {
        if (out_dev) HIPCHECK(hipMemsetAsync(wav_i16 + (size_t)b * wav_ld + N, 0, sizeof(short) * (size_t)(wav_ld - N), s));
      }
    }
  }
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  if (!out_dev && wav_ld > N) {
    for (int b = 0; b < B; ++b) {
      if (wav_f32) std::memset(wav_f32 + (size_t)b * wav_ld + N, 0, sizeof(float) * (size_t)(wav_ld - N));
      if (wav_i16) std::memset(wav_i16 + (size_t)b * wav_ld + N, 0, sizeof(int16_t) * (size_t)(wav_ld - N));
    }
  }
  return 0;
}

// ------------------------------------------------------------------ single operators
static int op_conv_common(mi355tts_ctx* ctx, const float* x, int B, int Cin, int L, const int32_t* lens, const float* wt,
                          const float* bias, int Cout, int K, int dil_or_stride, float in_slope, int out_act, float* y,
                          bool transposed) {
  if (!ctx || !x || !wt || !y || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0 || K <= 0)
    return fail(MI355TTS_ERR_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  ArenaBuilder ab;
  DevConv c;
  int Lout = L;
  if (transposed) {
    const int u = dil_or_stride;
    if (u < 1 || K % u || (K - u) % 2 || K / u > 3) return fail(MI355TTS_ERR_INVALID, "unsupported transposed conv (K=%d, stride=%d)", K, u);
    c = add_conv(ab, wt, bias, Cout, Cin, K, ROWS_UPSAMPLE, u);
    Lout = L * u;
  } else {
    if (!(K % 2)) return fail(MI355TTS_ERR_INVALID, "conv1d needs an odd kernel size");
    c = add_conv(ab, wt, bias, Cout, Cin, K, ROWS_PLAIN);
  }
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  Carver cv;
  const size_t o_w = cv.take(ab.host.size() * sizeof(float));
  const int Lp = (L + 3) & ~3;
  const size_t o_x = cv.take(sizeof(float) * (size_t)B * Cin * Lp);
  const size_t o_y = cv.take(sizeof(float) * (size_t)B * Cout * Lout);
  const size_t o_l = cv.take(sizeof(int) * B);
  CHECK(reserve(w, cv.pos));
  char* base = w->arena;
  float* dw = (float*)(base + o_w);
  float* dx = (float*)(base + o_x);
  float* dy = (float*)(base + o_y);
  int* dl = (int*)(base + o_l);
  hipStream_t s = w->stream;
  HIPCHECK(hipMemcpyAsync(dw, ab.host.data(), ab.host.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * Cin * Lp, s));
  HIPCHECK(hipMemcpy2DAsync(dx, sizeof(float) * Lp, x, sizeof(float) * L, sizeof(float) * L, (size_t)B * Cin, hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemsetAsync(dy, 0, sizeof(float) * (size_t)B * Cout * Lout, s));
  std::vector<int> hl(B, L);
  if (lens)
    for (int b = 0; b < B; ++b) {
      if (lens[b] < 0 || lens[b] > L) return fail(MI355TTS_ERR_INVALID, "lens[%d] out of range", b);
      hl[b] = lens[b];
    }
  HIPCHECK(hipMemcpyAsync(dl, hl.data(), sizeof(int) * B, hipMemcpyHostToDevice, s));
  fix(c, dw);
  int rc;
  if (transposed) {
    const int u = dil_or_stride;
    ConvArgs a = base_args(dx, (long long)Cin * Lp, Lp, dl, 1, dy, (long long)Cout * Lout, Lout, dl, u, 1, K / u - 1);
    a.in_slope = in_slope;
    a.up = u;
    a.up_pad = (K - u) / 2;
    rc = launch_conv(ctx, w, c, a, EPI_UPSAMPLE, B, L + K / u - 1, KC_UPSAMPLE);
  } else {
    const int dil = dil_or_stride;
    ConvArgs a = base_args(dx, (long long)Cin * Lp, Lp, dl, 1, dy, (long long)Cout * Lout, Lout, dl, 1, dil, (K * dil - dil) / 2);
    a.in_slope = in_slope;
    a.out_act = out_act;
    rc = launch_conv(ctx, w, c, a, EPI_LINEAR, B, L, KC_RESBLOCK);
  }
  if (rc) return rc;
  HIPCHECK(hipMemcpyAsync(y, dy, sizeof(float) * (size_t)B * Cout * Lout, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int mi355tts_op_conv1d(mi355tts_ctx* ctx, const float* x, int B, int Cin, int L, const int32_t* lens,
                                  const float* w, const float* bias, int Cout, int K, int dilation, float in_slope,
                                  int out_act, float* y) {
  return op_conv_common(ctx, x, B, Cin, L, lens, w, bias, Cout, K, dilation, in_slope, out_act, y, false);
}
extern "C" int mi355tts_op_conv_transpose1d(mi355tts_ctx* ctx, const float* x, int B, int Cin, int L, const float* w,
                                            const float* bias, int Cout, int K, int stride, float in_slope, float* y) {
  return op_conv_common(ctx, x, B, Cin, L, nullptr, w, bias, Cout, K, stride, in_slope, 0, y, true);
}

extern "C" int mi355tts_op_denoise(mi355tts_ctx* ctx, const float* wav, int B, int64_t N, const float* bias_spec,
                                   float strength, float* out) {
  if (!ctx || !wav || !bias_spec || !out || B <= 0 || N <= DN_FFT || (N % DN_HOP))
    return fail(MI355TTS_ERR_INVALID, "bad argument (N must be a multiple of 256 and > 1024)");
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  const int T = (int)((N - DN_FFT + DN_HOP - 1) / DN_HOP);
  Carver cv;
  const size_t o_in = cv.take(sizeof(float) * (size_t)B * N);
  const size_t o_out = cv.take(sizeof(float) * (size_t)B * N);
  const size_t o_f = cv.take(sizeof(float) * (size_t)B * T * DN_FFT);
  const size_t o_b = cv.take(sizeof(float) * (DN_FFT / 2 + 1));
  const size_t o_fr = cv.take(sizeof(int) * B);
  CHECK(reserve(w, cv.pos));
  char* base = w->arena;
  float* din = (float*)(base + o_in);
  float* dout = (float*)(base + o_out);
  float* fb = (float*)(base + o_f);
  float* db = (float*)(base + o_b);
  int* dfr = (int*)(base + o_fr);
  hipStream_t s = w->stream;
  std::vector<int> fr(B, (int)(N / DN_HOP));
  HIPCHECK(hipMemcpyAsync(din, wav, sizeof(float) * (size_t)B * N, hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(db, bias_spec, sizeof(float) * (DN_FFT / 2 + 1), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(dfr, fr.data(), sizeof(int) * B, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(stft_denoise_kernel, dim3(T, B), dim3(256), 0, s, din, (long long)N, dfr, DN_HOP, db, strength, fb, T,
                     (float*)nullptr);
  hipLaunchKernelGGL(overlap_add_kernel, dim3(256, B), dim3(256), 0, s, fb, T, dfr, DN_HOP, dout, (long long)N, (long long)N);
  HIPCHECK(hipMemcpyAsync(out, dout, sizeof(float) * (size_t)B * N, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int mi355tts_bench_conv1d(mi355tts_ctx* ctx, int B, int Cin, int Cout, int K, int dilation, int L,
                                     int tile_shape, int iters, float* ms_per_launch) {
  if (!ctx || !ms_per_launch || B <= 0 || Cin <= 0 || Cout <= 0 || L <= 0 || iters <= 0 || !(K % 2) || (L % 4))
    return fail(MI355TTS_ERR_INVALID, "bad argument (L must be a multiple of 4)");
  HIPCHECK(hipSetDevice(ctx->device));
  std::vector<float> wh((size_t)Cout * Cin * K), bh(Cout), xh((size_t)B * Cin * L);
  uint32_t st = 12345u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    return ((st >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f;
  };
  const float sc = 1.0f / std::sqrt((float)Cin * K);
  for (auto& v : wh) v = rnd() * sc;
  for (auto& v : bh) v = rnd();
  for (auto& v : xh) v = rnd();
  ArenaBuilder ab;
  DevConv c = add_conv(ab, wh.data(), bh.data(), Cout, Cin, K, ROWS_PLAIN);
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  Carver cv;
  const size_t o_w = cv.take(ab.host.size() * sizeof(float));
  const size_t o_x = cv.take(sizeof(float) * xh.size());
  const size_t o_y = cv.take(sizeof(float) * (size_t)B * Cout * L);
  CHECK(reserve(w, cv.pos));
  char* base = w->arena;
  float* dw = (float*)(base + o_w);
  float* dx = (float*)(base + o_x);
  float* dy = (float*)(base + o_y);
  hipStream_t s = w->stream;
  HIPCHECK(hipMemcpyAsync(dw, ab.host.data(), ab.host.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemcpyAsync(dx, xh.data(), sizeof(float) * xh.size(), hipMemcpyHostToDevice, s));
  fix(c, dw);
  ConvArgs a = base_args(dx, (long long)Cin * L, L, nullptr, 1, dy, (long long)Cout * L, L, nullptr, 1, dilation,
                         (K * dilation - dilation) / 2);
  a.in_const = L;
  a.out_const = L;
  a.in_slope = 0.1f;
  if (const char* ab = std::getenv("MI355TTS_BENCH_ABLATE")) a.ablate = std::atoi(ab);
  const bool prof = ctx->profiling;
  ctx->profiling = false;
  g_pin_tile = tile_shape;
  int rc = 0;
  for (int i = 0; i < 3 && !rc; ++i) rc = launch_conv(ctx, w, c, a, EPI_LINEAR, B, L, KC_RESBLOCK);
  hipEvent_t e0, e1;
  if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = fail(MI355TTS_ERR_HIP, "hipEventCreate");
  if (!rc) {
    hipEventRecord(e0, s);
    for (int i = 0; i < iters && !rc; ++i) rc = launch_conv(ctx, w, c, a, EPI_LINEAR, B, L, KC_RESBLOCK);
    hipEventRecord(e1, s);
    hipError_t e = hipStreamSynchronize(s);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) rc = fail(MI355TTS_ERR_HIP, "bench_conv1d: %s", hipGetErrorString(e));
    *ms_per_launch = ms / (float)iters;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  g_pin_tile = -1;
  ctx->profiling = prof;
  return rc;
}

// ------------------------------------------------------------------ measurement
extern "C" int mi355tts_set_profiling(mi355tts_ctx* ctx, int enabled) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->profiling = enabled != 0;
  return 0;
}
extern "C" int mi355tts_set_option(mi355tts_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return fail(MI355TTS_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (std::strcmp(name, "serial_branches") == 0) {
    ctx->serial_branches = value != 0;
    return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "unknown option '%s'", name);
}
extern "C" int mi355tts_profile_reset(mi355tts_ctx* ctx) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto& a : ctx->prof) a = mi355tts_ctx::Acc();
  return 0;
}
extern "C" int mi355tts_profile_json(mi355tts_ctx* ctx, char* buf, int cap) {
  if (!ctx || !buf || cap <= 2) return fail(MI355TTS_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  std::string s = "{";
  for (int i = 0; i < KC_COUNT; ++i) {
    char tmp[256];
    std::snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"launches\": %lld, \"ms\": %.6f, \"flop\": %.6e}", i ? ", " : "",
                  kclass_name[i], ctx->prof[i].launches, ctx->prof[i].ms, ctx->prof[i].flop);
    s += tmp;
  }
  s += "}";
  if ((int)s.size() + 1 > cap) return fail(MI355TTS_ERR_TOO_SMALL, "profile buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}
