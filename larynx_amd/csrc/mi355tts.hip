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
#include <time.h>
#include <sys/prctl.h>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355tts.h"
#include "conv_mfma.h"
#include "rb_conv.h"
#include "resblock_pair.h"
#include "rb_pair.h"
#include "conv_bf16.h"
#include "resblock_pair_bf16.h"
#include "mrf_small.h"
#include "gate16.h"
#include "coltile.h"
#include "voc_out.h"
#include "small_kernels.h"
#include "conv_f16.h"
#include "pair_f16.h"
#include "wn_f16.h"
#include "weights_pack.h"

using namespace mi355tts;
#include "host_models.h"
#include "host_context.h"
#include "host_launch.h"
#include "hifigan_f16.h"

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
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->ncu = prop.multiProcessorCount;
  }
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
    if (w->pinned_out) hipHostFree(w->pinned_out);
    if (w->stream) hipStreamDestroy(w->stream);
    delete w;
  }
  for (auto& pe : ctx->mel_pool) hipFree(pe.first);
  delete ctx;  // the models free their device memory in their destructors
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
  if (h->mel_channels < 1 || h->n_sqz < 1 || ((h->mel_channels * h->n_sqz) & 1)) return fail(MI355TTS_ERR_INVALID, "bad n_sqz");
  if (h->n_split < 2 || h->n_split > 8 || h->n_split % 2 || (h->mel_channels * h->n_sqz) % h->n_split)
    return fail(MI355TTS_ERR_INVALID, "bad n_split");
  if (h->n_blocks_dec < 0 || h->n_layers_enc < 0 || h->n_block_layers < 1 || h->filter_channels < 1 || h->filter_channels_dp < 1 ||
      h->dilation_rate < 1 || h->window_size < 0)
    return fail(MI355TTS_ERR_INVALID, "bad GlowTTS hparams");
  if (h->kernel_size != 1 && h->kernel_size != 3 && h->kernel_size != 5) return fail(MI355TTS_ERR_INVALID, "bad kernel_size");
  if (h->kernel_size_dec != 3 && h->kernel_size_dec != 5) return fail(MI355TTS_ERR_INVALID, "bad kernel_size_dec");
  if (h->n_speakers > 1) {
    if (h->gin_channels < 1 || h->gin_channels > SPEAKER_MAX_GIN)
      return fail(MI355TTS_ERR_INVALID, "a multi-speaker voice needs gin_channels in [1, %d] (got %d)", SPEAKER_MAX_GIN, h->gin_channels);
  } else if (h->n_speakers < 0 || h->gin_channels != 0) {
    // (the reference builds cond layers from gin_channels alone, models.py:287-301, but without emb_g nothing can feed them)
    return fail(MI355TTS_ERR_INVALID, "gin_channels = %d without n_speakers > 1 is not supported", h->gin_channels);
  }
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
  if (c.g16_J) {
    c.g16_w = arena + c.g16_w_off;
    c.g16_b = arena + c.g16_b_off;
  }
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

  auto gm = std::make_shared<GlowModel>();
  gm->hp = h;
  gm->device = ctx->device;
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
  if (gm->gin()) {
    TAKE(eg, std::string("emb_g.weight"), (int64_t)h.n_speakers * gm->gin());
    gm->emb_g = ab.add(eg, (size_t)h.n_speakers * gm->gin());
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
      add_lin16(ab, gm->pre_conv.back(), w, b, H, H, h.prenet_kernel_size);
      gm->pre_g.push_back(ab.add(g, H));
      gm->pre_b.push_back(ab.add(be, H));
    }
    TAKE(w, std::string("encoder.pre.proj.weight"), (int64_t)H * H);
    TAKE(b, std::string("encoder.pre.proj.bias"), H);
    gm->pre_proj = add_conv(ab, w, b, H, H, 1, ROWS_PLAIN);
    add_lin16(ab, gm->pre_proj, w, b, H, H, 1);
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
    add_lin16(ab, L.qkv, wqkv.data(), bqkv.data(), 3 * H, H, 1);
    L.o = add_conv(ab, wo, bo, H, H, 1, ROWS_PLAIN);
    L.o16 = add_col16(ab, wo, bo, H, H);
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
    add_lin16(ab, L.ffn1, w1, c1, Fc, H, k);
    add_lin16(ab, L.ffn2, w2, c2, H, Fc, k);
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
    add_lin16(ab, gm->proj_m, w, b, M, H, 1);
    TAKE(w1, std::string("encoder.proj_w.conv_1.weight"), (int64_t)Fd * (H + gm->gin()) * k);
    TAKE(b1, std::string("encoder.proj_w.conv_1.bias"), Fd);
    TAKE(g1, std::string("encoder.proj_w.norm_1.gamma"), Fd);
    TAKE(e1, std::string("encoder.proj_w.norm_1.beta"), Fd);
    TAKE(w2, std::string("encoder.proj_w.conv_2.weight"), (int64_t)Fd * Fd * k);
    TAKE(b2, std::string("encoder.proj_w.conv_2.bias"), Fd);
    TAKE(g2, std::string("encoder.proj_w.norm_2.gamma"), Fd);
    TAKE(e2, std::string("encoder.proj_w.norm_2.beta"), Fd);
    TAKE(wp, std::string("encoder.proj_w.proj.weight"), Fd);
    TAKE(bp, std::string("encoder.proj_w.proj.bias"), 1);
    const int gin = gm->gin();
    std::vector<float> w1x;  // multi-speaker: conv_1's weight is [Fd][H + gin][k]; the encoder half goes to the conv kernels
    if (gin) {
      std::vector<float> wg((size_t)Fd * gin * k);
      w1x.resize((size_t)Fd * H * k);
      for (int co = 0; co < Fd; ++co) {
        std::memcpy(&w1x[(size_t)co * H * k], w1 + (size_t)co * (H + gin) * k, sizeof(float) * (size_t)H * k);
        std::memcpy(&wg[(size_t)co * gin * k], w1 + ((size_t)co * (H + gin) + H) * k, sizeof(float) * (size_t)gin * k);
      }
      gm->dp_wg = ab.add(wg);
      w1 = w1x.data();
    }
    gm->dp1 = add_conv(ab, w1, b1, Fd, H, k, ROWS_PLAIN);
    gm->dp2 = add_conv(ab, w2, b2, Fd, Fd, k, ROWS_PLAIN);
    add_lin16(ab, gm->dp1, w1, b1, Fd, H, k);
    add_lin16(ab, gm->dp2, w2, b2, Fd, Fd, k);
    gm->dpp = add_conv(ab, wp, bp, 1, Fd, 1, ROWS_PLAIN);
    gm->dpp_w = ab.add(wp, Fd);
    gm->dpp_b = ab.add(bp, 1);
    gm->dg1 = ab.add(g1, Fd);
    gm->db1 = ab.add(e1, Fd);
    gm->dg2 = ab.add(g2, Fd);
    gm->db2 = ab.add(e2, Fd);
  }
  const int C = M * h.n_sqz, half = C / 2;
  std::vector<float> cond_w_all, cond_b_all;
  // the fp16 form of the decoder's WaveNets (wn_f16.h), when the geometry is one its kernel is built for
  gm->f16_why = glow_f16_unsupported(h);
  gm->f16_ok = gm->f16_why.empty();
  HPackSink gsink;
  gsink.ab = &ab;
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
    B.t_st = add_col16(ab, ws, bs, H, half);
    if (gm->gin()) {  // WN.cond_layer (layers.py:109-113): all blocks' weights side by side for speaker_cond_kernel
      const int64_t n2 = (int64_t)2 * H * h.n_block_layers;
      TAKE(wc, cp + ".wn.cond_layer.weight", n2 * gm->gin());
      TAKE(bc, cp + ".wn.cond_layer.bias", n2);
      cond_w_all.insert(cond_w_all.end(), wc, wc + n2 * gm->gin());
      cond_b_all.insert(cond_b_all.end(), bc, bc + n2);
    }
    for (int j = 0; j < h.n_block_layers; ++j) {
      std::string il = cp + ".wn.in_layers." + std::to_string(j);
      std::string rl = cp + ".wn.res_skip_layers." + std::to_string(j);
      const int rsn = (j < h.n_block_layers - 1) ? 2 * H : H;
      TAKE(wi, il + ".weight", (int64_t)2 * H * H * h.kernel_size_dec);
      TAKE(bi, il + ".bias", 2 * H);
      TAKE(wr, rl + ".weight", (int64_t)rsn * H);
      TAKE(br, rl + ".bias", rsn);
      B.in.push_back(add_conv(ab, wi, bi, 2 * H, H, h.kernel_size_dec, ROWS_PAIR, H));
      add_gate16(ab, B.in.back(), wi, bi, H, H, h.kernel_size_dec);
      B.rs.push_back(add_conv(ab, wr, br, rsn, H, 1, ROWS_PLAIN));
      add_lin16(ab, B.rs.back(), wr, br, rsn, H, 1);
      if (j == h.n_block_layers - 1) B.t_rs = add_col16(ab, wr, br, H, H);
      if (gm->f16_ok) {
        B.h_in.push_back(add_wn_gate_h(gsink, wi, bi, H, h.kernel_size_dec));
        if (j < h.n_block_layers - 1) B.h_rs.push_back(add_wn_rs_h(gsink, wr, br, H));
      }
    }
    TAKE(we, cp + ".end.weight", (int64_t)C * H);
    TAKE(be, cp + ".end.bias", C);
    B.end = add_conv(ab, we, be, C, H, 1, ROWS_PAIR, half);
    B.t_end = add_col16(ab, we, be, C, H);
    gm->blocks.push_back(std::move(B));
  }
#undef TAKE
  if (gm->gin()) {
    gm->cond_w = ab.add(cond_w_all);
    gm->cond_b = ab.add(cond_b_all);
  }
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
  if (gm->f16_ok) {
    hipError_t e = hipMalloc(&gm->arenaH, gsink.w.size() * sizeof(uint16_t) + 256);
    if (e != hipSuccess) return fail(MI355TTS_ERR_NOMEM, "hipMalloc fp16 weight arena: %s", hipGetErrorString(e));
    HIPCHECK(hipMemcpy(gm->arenaH, gsink.w.data(), gsink.w.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    for (auto& B : gm->blocks) {
      for (auto& c : B.h_in) fix_h(c, gm->arenaH, A);
      for (auto& c : B.h_rs) fix_h(c, gm->arenaH, A);
    }
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int id = ctx->next_id++;
  ctx->glow[id] = std::move(gm);
  *model_out = id;
  return 0;
}

static int dispatch_selfcheck_run(mi355tts_ctx* ctx);
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
  auto hm = std::make_shared<HifiModel>();
  hm->hp = h;
  hm->device = ctx->device;
  ArenaBuilder ab;
  std::vector<uint16_t> ab16;  // split-bf16 fragments (conv_bf16.h) of the ResBlock convs
  auto add16 = [&](DevConv& d, const float* w, int ch, int k) {
    if (ch < 32 || (ch % 32)) return;
    PackedConv16 p = pack_conv_bf16(ch, ch >= 128 ? 4 : ch / 32, ch, k, [&](int co, int ci, int kk) { return w[((size_t)co * ch + ci) * k + kk]; });
    d.w16_off = (ab16.size() + 127) & ~(size_t)127;  // 256-byte alignment
    ab16.resize(d.w16_off + p.w.size());
    std::memcpy(ab16.data() + d.w16_off, p.w.data(), p.w.size() * sizeof(uint16_t));
    d.mtiles16 = p.mtiles;
    d.nslab16 = p.nslab;
  };
  // the native fp16 mode's packing of every conv (hifigan_f16.h), when the geometry is one its tiles cover
  hm->f16_why = hifi_f16_unsupported(h);
  hm->f16_ok = hm->f16_why.empty();
  HPackSink hsink;
  hsink.ab = &ab;
  const int C0 = h.upsample_initial_channel;
#define TAKE(var, name, n)                         \
  const float* var = bl.take((name).c_str(), (n)); \
  if (!var) return MI355TTS_ERR_INVALID;
  {
    TAKE(w, std::string("conv_pre.weight"), (int64_t)C0 * h.num_mels * 7);
    TAKE(b, std::string("conv_pre.bias"), C0);
    hm->pre = add_conv(ab, w, b, C0, h.num_mels, 7, ROWS_PLAIN);
    if (hm->f16_ok) hm->h_pre = add_conv_h(hsink, w, b, C0, h.num_mels, 7);
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
    if (hm->f16_ok) hm->h_ups.push_back(add_ups_h(hsink, w, b, cout, cin, u));
    if (cin % 32 == 0 && (cout * u) % 32 == 0 && ku / u == 2) {
      // split-bf16 fragments of the polyphase form (virtual rows v = co * u + r, taps k = 0, 1 <-> m = 1, 0; see add_conv)
      DevConv& d = hm->ups.back();
      const int rows = cout * u, Kt = ku / u;
      PackedConv16 p16 = pack_conv_bf16(rows, rows >= 128 ? 4 : rows / 32, cin, Kt, [&](int v, int ci, int k) {
        const int co = v / u, r = v % u, m = Kt - 1 - k;
        return w[((size_t)ci * cout + co) * ku + m * u + r];
      });
      d.w16_off = (ab16.size() + 127) & ~(size_t)127;
      ab16.resize(d.w16_off + p16.w.size());
      std::memcpy(ab16.data() + d.w16_off, p16.w.data(), p16.w.size() * sizeof(uint16_t));
      d.mtiles16 = p16.mtiles;
      d.nslab16 = p16.nslab;
    }
    ch = cout;
    hm->rb[i].resize(h.num_kernels);
    if (hm->f16_ok) {
      hm->h_rb.resize(h.num_upsamples);
      hm->h_rb[i].resize(h.num_kernels);
    }
    // narrow stages additionally get the packing of the one-launch MRF kernel (mrf_small.h): ResBlock1 chains
    // with taps (3, 7, 11), <= 3 dilation steps and a receptive half-width within the staged halo
    MrfStage ms;
    std::vector<float> mrf_w, mrf_b, mrf_w8;
    int woff8[3][MRF_MAX_STEPS][2] = {};
    ms.ok = h.resblock_type == 1 && h.num_kernels == 3 && (ch == 8 || ch == 16) && h.num_dilations <= MRF_MAX_STEPS &&
            h.resblock_kernel_sizes[0] == 3 && h.resblock_kernel_sizes[1] == 7 && h.resblock_kernel_sizes[2] == 11;
    for (int j = 0; ms.ok && j < h.num_kernels; ++j) {
      int need = 0;
      for (int d = 0; d < h.num_dilations; ++d) {
        if (h.resblock_dilations[j][d] < 1) ms.ok = false;
        need += (h.resblock_kernel_sizes[j] - 1) / 2 * (h.resblock_dilations[j][d] + 1);
      }
      if (need > MRF_HALO) ms.ok = false;
    }
    ms.C = ch;
    ms.nsteps = h.num_dilations;
    if (ms.ok) mrf_b.assign((size_t)3 * MRF_MAX_STEPS * 2 * 16, 0.f);
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
          add16(rc.c1, w1, ch, k);
          add16(rc.c2, w2, ch, k);
          if (hm->f16_ok) {
            HResConv hr;
            hr.c1 = add_conv_h(hsink, w1, b1, ch, ch, k);
            hr.c2 = add_conv_h(hsink, w2, b2, ch, ch, k);
            hm->h_rb[i][j].push_back(hr);
          }
          if (ms.ok) {
            const float* ws[2] = {w1, w2};
            const float* bs[2] = {b1, b2};
            ms.dil[j][d] = rc.dil;
            for (int cv = 0; cv < 2; ++cv) {
              const float* wsrc = ws[cv];
              std::vector<float> pk = pack_mrf_conv(ch, k, [&](int co, int ci, int kk) { return wsrc[((size_t)co * ch + ci) * k + kk]; });
              ms.woff[j][d][cv] = (int)mrf_w.size();
              mrf_w.insert(mrf_w.end(), pk.begin(), pk.end());
              if (ch == 8) {
                std::vector<float> p8 = pack_mrf8_conv(k, [&](int co, int ci, int kk) { return wsrc[((size_t)co * ch + ci) * k + kk]; });
                woff8[j][d][cv] = (int)mrf_w8.size();
                mrf_w8.insert(mrf_w8.end(), p8.begin(), p8.end());
              }
              std::memcpy(&mrf_b[(((size_t)j * MRF_MAX_STEPS + d) * 2 + cv) * 16], bs[cv], sizeof(float) * ch);
            }
            ms.mac_per_col += 2.0 * ch * ch * k;
          }
        } else {
          TAKE(w1, rb + ".convs." + std::to_string(d) + ".weight", (int64_t)ch * ch * k);
          TAKE(b1, rb + ".convs." + std::to_string(d) + ".bias", ch);
          rc.c1 = add_conv(ab, w1, b1, ch, ch, k, ROWS_PLAIN);
          add16(rc.c1, w1, ch, k);
          if (hm->f16_ok) {
            HResConv hr;
            hr.c1 = add_conv_h(hsink, w1, b1, ch, ch, k);
            hm->h_rb[i][j].push_back(hr);
          }
        }
        hm->rb[i][j].push_back(rc);
      }
    }
    if (ms.ok) {
      ms.w_off = ab.add(mrf_w);
      ms.b_off = ab.add(mrf_b);
      int tab[MRF_TAB_INTS] = {};
      for (int j = 0; j < 3; ++j)
        for (int d = 0; d < MRF_MAX_STEPS; ++d) {
          tab[(j * MRF_MAX_STEPS + d) * 2 + 0] = ms.woff[j][d][0];
          tab[(j * MRF_MAX_STEPS + d) * 2 + 1] = ms.woff[j][d][1];
          tab[MRF_TAB_DIL + j * MRF_MAX_STEPS + d] = ms.dil[j][d];
        }
      static_assert(sizeof(int) == sizeof(float), "the table rides in the float arena");
      ms.t_off = ab.add(reinterpret_cast<const float*>(tab), MRF_TAB_INTS);
      if (ch == 8) {
        mrf_w8.resize(mrf_w8.size() + 64, 0.f);  // the tap loop's prefetch reads one fragment past the last conv
        ms.w8_off = ab.add(mrf_w8);
        for (int j = 0; j < 3; ++j)
          for (int d = 0; d < MRF_MAX_STEPS; ++d) {
            tab[(j * MRF_MAX_STEPS + d) * 2 + 0] = woff8[j][d][0];
            tab[(j * MRF_MAX_STEPS + d) * 2 + 1] = woff8[j][d][1];
          }
        ms.t8_off = ab.add(reinterpret_cast<const float*>(tab), MRF_TAB_INTS);
      }
    }
    hm->mrf.push_back(ms);
  }
  {
    TAKE(w, std::string("conv_post.weight"), (int64_t)ch * 7);
    TAKE(b, std::string("conv_post.bias"), 1);
    hm->post = add_conv(ab, w, b, 1, ch, 7, ROWS_PLAIN);
    hm->post_w_off = ab.add(w, (size_t)ch * 7);  // raw [C][7] + bias for post_conv_kernel (voc_out.h)
    hm->post_b_off = ab.add(b, 1);
    hm->post_C = ch;
  }
#undef TAKE
  CHECK(upload_arena(ctx, ab, &hm->arena));
  if (!ab16.empty()) {
    hipError_t e = hipMalloc(&hm->arena16, ab16.size() * sizeof(uint16_t) + 256);
    if (e != hipSuccess) return fail(MI355TTS_ERR_NOMEM, "hipMalloc bf16 weight arena: %s", hipGetErrorString(e));
    HIPCHECK(hipMemcpy(hm->arena16, ab16.data(), ab16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  auto fix16 = [&](DevConv& c) { c.w16 = c.mtiles16 ? (const void*)(hm->arena16 + c.w16_off) : nullptr; };
  const float* A = hm->arena;
  if (hm->f16_ok) {
    hipError_t e = hipMalloc(&hm->arenaH, hsink.w.size() * sizeof(uint16_t) + 256);
    if (e != hipSuccess) return fail(MI355TTS_ERR_NOMEM, "hipMalloc fp16 weight arena: %s", hipGetErrorString(e));
    HIPCHECK(hipMemcpy(hm->arenaH, hsink.w.data(), hsink.w.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    fix_h(hm->h_pre, hm->arenaH, A);
    for (auto& c : hm->h_ups) fix_h(c, hm->arenaH, A);
    for (auto& st : hm->h_rb)
      for (auto& kk : st)
        for (auto& rc : kk) {
          fix_h(rc.c1, hm->arenaH, A);
          if (h.resblock_type == 1) fix_h(rc.c2, hm->arenaH, A);
        }
  }
  fix(hm->pre, A);
  fix(hm->post, A);
  for (auto& c : hm->ups) {
    fix(c, A);
    fix16(c);
  }
  for (auto& st : hm->rb)
    for (auto& kk : st)
      for (auto& rc : kk) {
        fix(rc.c1, A);
        fix16(rc.c1);
        if (h.resblock_type == 1) {
          fix(rc.c2, A);
          fix16(rc.c2);
        }
      }
  // the first vocoder with a 256-channel stage: check the dispatcher rule its batch-1 schedule relies on (once per context)
  if ((C0 >> 1) >= 256 && h.resblock_type == 1) dispatch_selfcheck_run(ctx);  // (a failed check leaves the defaults)
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int id = ctx->next_id++;
  ctx->hifi[id] = std::move(hm);
  *model_out = id;
  return 0;
}

// Safe while other threads are synthesising with the model: every call pins its models (find_glow / find_hifi hand out
// shared_ptr copies), so this only drops the context's reference and the id; the arenas are freed by whoever lets go
// last — here, or the last call still using the model when it returns.  The reference frees a voice by dropping its
// Python object the same way (larynx/__init__.py:290-300 keeps models in a dict).
extern "C" int mi355tts_unload(mi355tts_ctx* ctx, int model) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  std::shared_ptr<GlowModel> g;
  std::shared_ptr<HifiModel> v;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto gi = ctx->glow.find(model);
    if (gi != ctx->glow.end()) {
      g = std::move(gi->second);
      ctx->glow.erase(gi);
    } else {
      auto vi = ctx->hifi.find(model);
      if (vi == ctx->hifi.end()) return fail(MI355TTS_ERR_NO_MODEL, "no model %d", model);
      v = std::move(vi->second);
      ctx->hifi.erase(vi);
    }
  }
  return 0;  // g / v released outside the lock (hipFree synchronises the device)
}

// §8(e): the one collective of the path.  The library does not link RCCL: the entry point is resolved from the
// RCCL library the CALLER's communicator belongs to (already loaded in its process).
extern "C" int mi355tts_broadcast_weights(mi355tts_ctx* ctx, void* nccl_comm, int root, float* device_blob, int64_t numel,
                                          const char* rccl_library) {
  if (!ctx || !nccl_comm || !device_blob || numel <= 0 || root < 0) return fail(MI355TTS_ERR_INVALID, "bad argument");
  typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  typedef const char* (*errstr_fn)(int);
  // Only the instance ALREADY loaded in this process can own the caller's communicator: RTLD_NOLOAD never maps a
  // second copy of RCCL (handing an ncclComm_t to another instance's ncclBroadcast is undefined behaviour).
  const char* names[] = {rccl_library, "librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (!n) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (lib) break;
  }
  if (!lib)
    return fail(MI355TTS_ERR_INVALID, "no RCCL library is loaded in this process under the names tried (%s, librccl.so.1, librccl.so): "
                                      "pass the path the communicator's library was loaded from", rccl_library ? rccl_library : "-");
  struct Close {  // RTLD_NOLOAD still takes a reference
    void* h;
    ~Close() { dlclose(h); }
  } closer{lib};
  bcast_fn bcast = (bcast_fn)dlsym(lib, "ncclBroadcast");
  errstr_fn errstr = (errstr_fn)dlsym(lib, "ncclGetErrorString");
  if (!bcast) return fail(MI355TTS_ERR_INVALID, "ncclBroadcast not found in the RCCL library");
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  const int rc = bcast(device_blob, device_blob, (size_t)numel, /*ncclFloat32*/ 7, root, nccl_comm, w->stream);
  if (rc != 0) return fail(MI355TTS_ERR_HIP, "ncclBroadcast failed: %s", errstr ? errstr(rc) : "?");
  HIPCHECK(mi355_sync(w->stream));
  return 0;
}

extern "C" int mi355tts_model_set_precision(mi355tts_ctx* ctx, int model, int precision) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  if (precision != MI355TTS_PRECISION_F32 && precision != MI355TTS_PRECISION_BF16X3 && precision != MI355TTS_PRECISION_BF16 &&
      precision != MI355TTS_PRECISION_F16)
    return fail(MI355TTS_ERR_INVALID, "unknown precision %d", precision);
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto v = ctx->hifi.find(model);
  if (v != ctx->hifi.end()) {
    if (precision == MI355TTS_PRECISION_F16 && !v->second->f16_ok)
      return fail(MI355TTS_ERR_INVALID, "fp16 mode is not available for this vocoder: %s", v->second->f16_why.c_str());
    v->second->precision.store(precision);
    return 0;
  }
  // GlowTTS: MI355TTS_PRECISION_F16 puts the decoder's WaveNets — 84 % of the acoustic model's FLOPs, 97 of its ~140 launches —
  // on the fp16 matrix cores (wn_f16.h) where the geometry is covered; the split-bf16 requests, and F16 on a geometry the
  // kernel is not built for, are REPORTED as having no effect (a distinct positive status: not an error, not a silent success)
  auto gi = ctx->glow.find(model);
  if (gi != ctx->glow.end()) {
    if (precision == MI355TTS_PRECISION_F32 || (precision == MI355TTS_PRECISION_F16 && gi->second->f16_ok)) {
      gi->second->precision.store(precision);
      return 0;
    }
    return MI355TTS_PRECISION_NOOP;
  }
  return fail(MI355TTS_ERR_NO_MODEL, "no model %d", model);
}

// ------------------------------------------------------------------ mel objects
// Recycled device blocks for the result objects.  Small blocks (frame counts) are one 4 KB
// class; larger ones are handed out best-fit with no upper bound on the slack (HBM is not the
// scarce resource here — a hipMalloc, which synchronises the device under concurrent calls, is).
static size_t pool_round(size_t bytes) { return bytes <= 4096 ? 4096 : ((bytes + 65535) & ~(size_t)65535); }
static void* pool_alloc(mi355tts_ctx* ctx, size_t bytes) {
  bytes = pool_round(bytes);
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    int best = -1;
    for (int i = 0; i < (int)ctx->mel_pool.size(); ++i) {
      const size_t have = ctx->mel_pool[i].second;
      if (have < bytes || (bytes == 4096) != (have == 4096)) continue;
      if (best < 0 || have < ctx->mel_pool[best].second) best = i;
    }
    if (best >= 0) {
      void* p = ctx->mel_pool[best].first;
      ctx->mel_pool.erase(ctx->mel_pool.begin() + best);
      return p;
    }
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->mel_sizes[p] = bytes;
  return p;
}
static void pool_free(mi355tts_ctx* ctx, void* p, size_t /*requested*/) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->mel_sizes.find(p);  // the block's true size, not the size last asked for
    if (it != ctx->mel_sizes.end() && ctx->mel_pool.size() < ctx->mel_pool_cap) {
      ctx->mel_pool.emplace_back(p, it->second);
      return;
    }
    if (it != ctx->mel_sizes.end()) ctx->mel_sizes.erase(it);
  }
  hipFree(p);  // over the cap, or a pointer the pool never handed out
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
  if (e == hipSuccess) e = mi355_sync(w->stream);
  if (e != hipSuccess) {
    mel_destroy(m);
    return fail(MI355TTS_ERR_HIP, "mel_from_buffer: %s", hipGetErrorString(e));
  }
  *out = m;
  return 0;
}

#include "glow_forward.h"
#include "hifigan_forward.h"
#include "host_join.h"

// ------------------------------------------------------------------ fused call + reservation
// ids -> int16/f32 waveform in ONE call on ONE worker: GlowTTS and the vocoder are queued
// back to back on the same stream, so the only host synchronisations are the frame-count
// read-back and the final one (the two-call form adds a sync, a worker hand-over and a mel
// object round trip through the caller).  Replaces the body of `_sentence_task`
// (larynx/__init__.py:229-283) between the two log lines, pause padding included.
static int synthesize_impl(mi355tts_ctx* ctx, int glow, int vocoder, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                           float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                           const int32_t* speaker_ids, const mi355tts_audio_settings* audio, float denoiser_strength,
                           int32_t pad_before, int32_t pad_after, int32_t* frames_out, float* wav_f32, int16_t* wav_i16,
                           int64_t wav_ld, uint32_t flags) {
  if (!ctx || !frames_out) return fail(MI355TTS_ERR_INVALID, "null argument");
  std::shared_ptr<GlowModel> gpin;
  std::shared_ptr<HifiModel> vpin;
  CHECK(find_glow(ctx, glow, &gpin));
  CHECK(find_hifi(ctx, vocoder, &vpin));
  const GlowModel* gm = gpin.get();
  HifiModel* hm = vpin.get();
  GlowCall g;
  g.ids = ids;
  g.id_lens = id_lens;
  g.B = B;
  g.ids_ld = ids_ld;
  g.noise_scale = noise_scale;
  g.length_scale = length_scale;
  g.noise = noise;
  g.noise_ld = noise_ld;
  g.seed = seed;
  g.speaker_ids = speaker_ids;
  g.audio = audio;
  g.flags = flags & MI355TTS_IN_DEVICE;
  VocCall v;
  v.denoiser_strength = denoiser_strength;
  v.wav_f32 = wav_f32;
  v.wav_i16 = wav_i16;
  v.wav_ld = wav_ld;
  v.flags = flags & MI355TTS_OUT_DEVICE;
  v.pad_before = pad_before;
  v.pad_after = pad_after;
  int Pmax = 0;
  CHECK(glow_precheck(gm, g, &Pmax));
  CHECK(hifigan_precheck(ctx, hm, vocoder, nullptr, B, gm->hp.mel_channels, -1, v));  // incl. the one-time denoiser bias
  HIPCHECK(hipSetDevice(ctx->device));
  static const bool no_coalesce = [] { const char* e = std::getenv("MI355TTS_NO_CALL_COALESCE"); return e && std::atoi(e) != 0; }();
  const int lanes = no_coalesce ? 0 : ctx->call_coalesce.load();
  if (lanes > 0 && B == 1 && !noise && !speaker_ids && id_lens[0] <= ATTM_MAXP && ctx->voc_out.load() && !ctx->serial_branches.load()) {
    // a batch-1 call: it rides a fused padded call with whichever other batch-1 calls are waiting right now (host_join.h)
    CallReq req;
    req.gm = gm;
    req.hm = hm;
    req.vocoder = vocoder;
    req.ids = ids;
    req.len = id_lens[0];
    req.noise_scale = noise_scale;
    req.length_scale = length_scale;
    req.seed = seed;
    req.audio = audio;
    req.flags = flags & (MI355TTS_IN_DEVICE | MI355TTS_OUT_DEVICE);
    req.denoiser_strength = denoiser_strength;
    req.out.wav_f32 = wav_f32;
    req.out.wav_i16 = wav_i16;
    req.out.wav_ld = wav_ld;
    req.out.pad_before = pad_before;
    req.out.pad_after = pad_after;
    const int jr = call_join(ctx, req, lanes);
    if (jr <= 0) {
      frames_out[0] = req.frames;
      return jr;
    }
    // jr == 1: the shared pass failed; this call runs alone below and reports its own result
  }
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
  CHECK(glow_run(ctx, w, gm, g, Pmax, false, &mel));
  drop.m = mel;
  for (int b = 0; b < B; ++b) frames_out[b] = mel->frames[b];
  CHECK(hifigan_precheck(ctx, hm, vocoder, mel->frames.data(), B, mel->M, mel->max_frames, v));
  return hifigan_run(ctx, w, hm, mel, v);
}
extern "C" int mi355tts_synthesize(mi355tts_ctx* ctx, int glow, int vocoder, const int64_t* ids, const int32_t* id_lens, int B,
                                   int ids_ld, float noise_scale, float length_scale, const float* noise, int noise_ld,
                                   uint64_t seed, const mi355tts_audio_settings* audio, float denoiser_strength,
                                   int32_t pad_before, int32_t pad_after, int32_t* frames_out, float* wav_f32, int16_t* wav_i16,
                                   int64_t wav_ld, uint32_t flags) {
  return synthesize_impl(ctx, glow, vocoder, ids, id_lens, B, ids_ld, noise_scale, length_scale, noise, noise_ld, seed, nullptr, audio,
                         denoiser_strength, pad_before, pad_after, frames_out, wav_f32, wav_i16, wav_ld, flags);
}
// the same for a multi-speaker voice (larynx/glow_tts.py:116-130: the `speaker_id` setting)
extern "C" int mi355tts_synthesize_speakers(mi355tts_ctx* ctx, int glow, int vocoder, const int64_t* ids, const int32_t* id_lens, int B,
                                            int ids_ld, float noise_scale, float length_scale, const float* noise, int noise_ld,
                                            uint64_t seed, const int32_t* speaker_ids, const mi355tts_audio_settings* audio,
                                            float denoiser_strength, int32_t pad_before, int32_t pad_after, int32_t* frames_out,
                                            float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags) {
  if (!speaker_ids) return fail(MI355TTS_ERR_INVALID, "speaker_ids null");
  return synthesize_impl(ctx, glow, vocoder, ids, id_lens, B, ids_ld, noise_scale, length_scale, noise, noise_ld, seed, speaker_ids, audio,
                         denoiser_strength, pad_before, pad_after, frames_out, wav_f32, wav_i16, wav_ld, flags);
}

// Pre-create `workers` workers (streams, pinned staging, side streams) and size their
// workspaces and the result-block pool for calls of up to `max_batch` rows x `max_ids`
// ids x `max_frames` frames, so that steady-state calls never hipMalloc / hipFree (both
// synchronise the whole device and stall every other in-flight call).
// ---- which worker streams share a hardware queue (mi355tts_reserve)
// The runtime deals its few hardware queues (4: GPU_MAX_HW_QUEUES) to streams when a stream is first used; two streams on one
// queue run their kernels one after the other.  How it deals is not ours to know (measured: the nine workers of an 8-caller
// context land 3 / 2 / 2 / 1 + spare on the four queues, or 2 / 2 / 2 / 2, depending on how many streams the host process used
// before — rocprofv3 kernel trace, tools/gpu/queue_map.sh — and the driver's 20-utterance regions read 287.5 against 293
// utterances/s: profiles/r06_queue_map.txt).  So the context MEASURES it once: stream A runs a kernel that waits for a host flag,
// stream B a kernel that raises another; if B's flag does not come up while A waits, B sits behind A in A's queue.  acquire_worker
// then hands out the free worker whose queue carries the fewest calls.  Which worker a call gets never changes its result.
__global__ void queue_bind_kernel() {}
__global__ void queue_wait_kernel(int* go, long long max_ticks) {
  const long long t0 = wall_clock64();  // 100 MHz: the bound keeps a lost flag from hanging the queue
  while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void queue_mark_kernel(int* done) { __hip_atomic_store(done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// flags: two ints of host memory the device can reach ([0] = go, [1] = done)
static bool streams_share_queue(hipStream_t a, hipStream_t b, int* flags) {
  __atomic_store_n(&flags[0], 0, __ATOMIC_RELEASE);
  __atomic_store_n(&flags[1], 0, __ATOMIC_RELEASE);
  hipLaunchKernelGGL(queue_wait_kernel, dim3(1), dim3(1), 0, a, flags, 400000LL);  // at most 4 ms
  hipLaunchKernelGGL(queue_mark_kernel, dim3(1), dim3(1), 0, b, flags + 1);
  const auto t0 = std::chrono::steady_clock::now();
  bool seen = false;
  while (!(seen = __atomic_load_n(&flags[1], __ATOMIC_ACQUIRE) != 0) && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(500)) {
  }
  __atomic_store_n(&flags[0], 1, __ATOMIC_RELEASE);
  hipStreamSynchronize(a);
  hipStreamSynchronize(b);
  return !seen;
}
// groups the workers by hardware queue (Worker::qgroup) and sizes ctx->qgroup_busy; every worker in `ws` is held by the caller
static void probe_queue_groups(mi355tts_ctx* ctx, const std::vector<Worker*>& ws) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_QUEUE_PROBE"); return e && std::atoi(e) != 0; }();
  if (off || ws.size() < 2) return;
  int* flags = nullptr;
  if (hipHostMalloc(&flags, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return;  // fine-grained: the device sees the host's store while its kernel runs
  // every main stream used once, back to back, before anything is measured: bound to its queue
  for (Worker* w : ws) hipLaunchKernelGGL(queue_bind_kernel, dim3(1), dim3(64), 0, w->stream);
  for (Worker* w : ws) hipStreamSynchronize(w->stream);
  std::vector<Worker*> reps;
  for (Worker* w : ws) {
    int g = -1;
    for (size_t r = 0; r < reps.size() && g < 0; ++r)
      if (streams_share_queue(reps[r]->stream, w->stream, flags)) g = (int)r;
    if (g < 0) {
      g = (int)reps.size();
      reps.push_back(w);
    }
    w->qgroup = g;
  }
  hipHostFree(flags);
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (Worker* w : ctx->all_workers)  // a worker some other call holds right now was not probed: its old group number means nothing any more
    if (std::find(ws.begin(), ws.end(), w) == ws.end()) w->qgroup = -1;
  // (held workers were counted busy under their OLD groups, if any: the caller releases them after this, so start from zero
  // and let release_worker's floor at 0 absorb them)
  ctx->qgroup_busy.assign(reps.size(), 0);
  for (Worker* w : ws) ctx->qgroup_busy[w->qgroup] += 1;  // they are held right now
}

extern "C" int mi355tts_worker_queue_groups(mi355tts_ctx* ctx, int32_t* groups, int capacity) {
  if (!ctx || (capacity > 0 && !groups)) return fail(MI355TTS_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int n = (int)ctx->all_workers.size();
  for (int i = 0; i < n && i < capacity; ++i) groups[i] = ctx->all_workers[i]->qgroup;
  return n;
}

extern "C" int mi355tts_reserve(mi355tts_ctx* ctx, int workers, int glow, int vocoder, int max_batch, int max_ids,
                                int max_frames, int denoiser, int max_pad_samples) {
  if (!ctx || workers < 1 || workers > 256 || max_batch < 1 || max_ids < 1 || max_frames < 1 || max_pad_samples < 0)
    return fail(MI355TTS_ERR_INVALID, "bad argument");
  std::shared_ptr<GlowModel> gpin;
  std::shared_ptr<HifiModel> vpin;
  if (glow > 0) CHECK(find_glow(ctx, glow, &gpin));
  if (vocoder > 0) CHECK(find_hifi(ctx, vocoder, &vpin));
  const GlowModel* gm = gpin.get();
  HifiModel* hm = vpin.get();
  if (gm && max_frames % gm->hp.n_sqz) max_frames += gm->hp.n_sqz - max_frames % gm->hp.n_sqz;
  size_t need = 0, mel_bytes = 0;
  int M = 0;
  if (gm) {
    const GlowEncLayout el = glow_enc_layout(gm->hp, max_batch, max_ids, max_ids);
    const GlowDecLayout dl = glow_dec_layout(gm->hp, el.total, max_batch, max_frames, 0);
    need = std::max(need, dl.total);
    M = gm->hp.mel_channels;
  }
  size_t out_bytes = 0;
  if (hm) {
    const bool dn = denoiser != 0 && (long long)max_frames * hm->hop > DN_FFT;
    // both vocoder schedules (forked MRF chains / one stream) carve the same planes unless serial_branches is set
    const HifiLayout a = hifi_layout(hm->hp, hm->hop, max_batch, max_frames, dn, true && hm->hp.num_kernels >= 2 && hm->hp.num_kernels <= 3, max_pad_samples);
    const HifiLayout b = hifi_layout(hm->hp, hm->hop, max_batch, max_frames, dn, false, max_pad_samples);
    need = std::max(need, std::max(a.total, b.total));
    M = std::max(M, (int)hm->hp.num_mels);
    out_bytes = (size_t)max_batch * ((size_t)max_frames * hm->hop + max_pad_samples) * (sizeof(float) + sizeof(short));
    if (dn) CHECK(ensure_denoiser_bias(ctx, hm, vocoder));
  }
  mel_bytes = (size_t)max_batch * M * (size_t)((max_frames + 3) & ~3) * sizeof(float);
  HIPCHECK(hipSetDevice(ctx->device));
  std::vector<Worker*> held;
  int rc = 0;
  for (int i = 0; i < workers && !rc; ++i) {
    Worker* w = nullptr;
    rc = acquire_worker(ctx, &w);  // holding them all forces `workers` distinct objects
    if (rc) break;
    held.push_back(w);
    rc = reserve(w, need);
    if (!rc) rc = reserve_pinned_out(w, out_bytes);
    if (!rc && hm && !w->aux[0]) {
      for (int k = 0; k < 2 && !rc; ++k) {
        if (hipStreamCreateWithFlags(&w->aux[k], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&w->ev_join[k], hipEventDisableTiming) != hipSuccess)
          rc = fail(MI355TTS_ERR_HIP, "side stream creation failed");
      }
      if (!rc && hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) != hipSuccess) rc = fail(MI355TTS_ERR_HIP, "event creation failed");
    }
  }
  if (!rc) probe_queue_groups(ctx, held);  // which workers share a hardware queue: acquire_worker spreads the calls over the queues
  for (Worker* w : held) release_worker(ctx, w);
  if (rc) return rc;
  // result blocks: per in-flight call two mel planes + one frame-count block (the pool must be able to hold them all)
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->mel_pool_cap = std::max(ctx->mel_pool_cap, (size_t)3 * workers + 64);
  }
  if (mel_bytes) {
    std::vector<void*> blocks;
    for (int i = 0; i < workers && !rc; ++i) {
      for (int k = 0; k < 2; ++k) {
        void* p = pool_alloc(ctx, mel_bytes);
        if (!p) rc = fail(MI355TTS_ERR_NOMEM, "hipMalloc result block");
        else blocks.push_back(p);
      }
      void* q = pool_alloc(ctx, sizeof(int) * (size_t)max_batch);
      if (!q) rc = fail(MI355TTS_ERR_NOMEM, "hipMalloc result block");
      else blocks.push_back(q);
    }
    for (void* p : blocks) pool_free(ctx, p, 0);
  }
  return rc;
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
  HIPCHECK(mi355_sync(s));
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
  HIPCHECK(mi355_sync(s));
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int mi355tts_op_gauss_noise(mi355tts_ctx* ctx, uint64_t seed, int B, int C, int T, float* out) {
  if (!ctx || !out || B <= 0 || C <= 0 || T <= 0 || C > 65535 || B > 65535) return fail(MI355TTS_ERR_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  const size_t n = (size_t)B * C * T;
  CHECK(reserve(w, n * sizeof(float)));
  float* d = (float*)w->arena;
  hipLaunchKernelGGL(noise_fill_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, w->stream, d, C, T, seed);
  HIPCHECK(hipMemcpyAsync(out, d, n * sizeof(float), hipMemcpyDeviceToHost, w->stream));
  HIPCHECK(mi355_sync(w->stream));
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
  const bool prof = ctx->profiling.load();
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
    hipError_t e = mi355_sync(s);
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

// One-off check of the dispatcher rule behind group_snake_order / promote_group_plans (host_launch.h) on THIS device: the
// grouped launch of a 256-channel ResBlock step at the standard utterance's length (3 x 156 tiles of the 128-row tile: all
// resident at once), eight launches in the plain longest-first order against eight in the snake order, interleaved.  Where the
// snake is not at least as fast (2 % margin) the order and the promotion that relies on it are switched off for the context.
// Runs once per context, on the first load of a vocoder with a >= 256-channel stage (or on request); ~15 ms.
static int dispatch_selfcheck_run(mi355tts_ctx* ctx) {
  int expected = 0;
  if (!ctx->selfcheck_state.compare_exchange_strong(expected, 4)) return 0;  // someone ran (or is running: state 4) it
  struct Skipped {  // every early exit below leaves "skipped" behind
    std::atomic<int>& st;
    ~Skipped() {
      int running = 4;
      st.compare_exchange_strong(running, 3);
    }
  } skipped{ctx->selfcheck_state};
  {
    const char* e = std::getenv("MI355TTS_NO_SELFCHECK");
    hipDeviceProp_t prop;
    const bool emu = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && std::strncmp(prop.gcnArchName, "hipemu", 6) == 0;
    if ((e && std::atoi(e) != 0) || emu) return 0;  // (the CPU emulator build of the tests: nothing to measure)
  }
  HIPCHECK(hipSetDevice(ctx->device));
  const int C = 256, L = 78 * 64;
  const int Ks[3] = {11, 7, 3};
  ArenaBuilder ab;
  DevConv cv[3];
  uint32_t st = 2463534242u;
  auto rnd = [&]() {
    st = st * 1664525u + 1013904223u;
    return ((st >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f;
  };
  for (int m = 0; m < 3; ++m) {
    std::vector<float> wh((size_t)C * C * Ks[m]), bh(C, 0.f);
    const float sc = 1.0f / std::sqrt((float)C * Ks[m]);
    for (auto& v : wh) v = rnd() * sc;
    cv[m] = add_conv(ab, wh.data(), bh.data(), C, C, Ks[m], ROWS_PLAIN);
  }
  Worker* w = nullptr;
  CHECK(acquire_worker(ctx, &w));
  WorkerGuard guard{ctx, w};
  Carver cvr;
  const size_t o_w = cvr.take(ab.host.size() * sizeof(float));
  const size_t o_x = cvr.take(sizeof(float) * (size_t)C * L);
  size_t o_y[3];
  for (int m = 0; m < 3; ++m) o_y[m] = cvr.take(sizeof(float) * (size_t)C * L);
  CHECK(reserve(w, cvr.pos));
  char* base = w->arena;
  float* dw = (float*)(base + o_w);
  float* dx = (float*)(base + o_x);
  hipStream_t s = w->stream;
  HIPCHECK(hipMemcpyAsync(dw, ab.host.data(), ab.host.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)C * L, s));
  ConvPlan plans[3];
  ConvPlan* pp[3] = {&plans[0], &plans[1], &plans[2]};
  for (int m = 0; m < 3; ++m) {
    fix(cv[m], dw);
    ConvArgs a = base_args(dx, (long long)C * L, L, nullptr, 1, (float*)(base + o_y[m]), (long long)C * L, L, nullptr, 1, 1, (Ks[m] - 1) / 2);
    a.in_const = a.out_const = L;
    a.in_slope = 0.1f;
    CHECK(plan_conv(cv[m], a, EPI_LINEAR, 1, L, KC_RESBLOCK, 1024, L, &plans[m], 0));
  }
  struct Quiet {  // the check's launches are not a caller's: neither profiled nor counted (worker-local)
    Worker* w;
    ~Quiet() { w->quiet = false; }
  } quiet{w};
  w->quiet = true;
  w->o_group_promote = true;
  w->o_rb_conv = true;
  promote_group_plans(ctx, w, pp, 3);
  int rc = 0;
  float us[2] = {0.f, 0.f};
  if (plans[0].shape != TILE_M128) {
    rc = 1;  // the promotion rule itself declined this geometry on this device (CU count): nothing to check
  } else {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = fail(MI355TTS_ERR_HIP, "hipEventCreate");
    for (int warm = 0; warm < 2 && !rc; ++warm)
      for (int order = 0; order < 2 && !rc; ++order) {
        w->o_snake = order == 1;
        if (run_group(ctx, w, plans, 3, s) != 0) rc = 1;
      }
    for (int rep = 0; rep < 4 && !rc; ++rep)
      for (int order = 0; order < 2 && !rc; ++order) {
        w->o_snake = order == 1;
        hipEventRecord(e0, s);
        for (int i = 0; i < 2 && !rc; ++i)
          if (run_group(ctx, w, plans, 3, s) != 0) rc = 1;
        hipEventRecord(e1, s);
        float ms = 0.f;
        if (mi355_sync(s) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = 1;
        us[order] += 1e3f * ms / 8.0f;
      }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
  }
  if (rc != 0) return rc < 0 ? rc : 0;  // state becomes 3: skipped
  ctx->selfcheck_plain_us = us[0];
  ctx->selfcheck_snake_us = us[1];
  if (us[1] > 1.02f * us[0]) {
    // only the ORDER option: it cannot change a result.  The promotion rule picks the tile (= the summation order) and stays
    // a function of the CU count and the geometry, whatever a timing on a busy device says
    ctx->group_snake = false;
    ctx->selfcheck_state = 2;
  } else {
    ctx->selfcheck_state = 1;
  }
  return 0;
}
// state: 0 = not run, 4 = running on another thread, 1 = the snake order is kept, 2 = the snake order was switched off on this
// device (option "group_snake"; results are the same bits either way), 3 = skipped (MI355TTS_NO_SELFCHECK, the promotion rule declines the geometry on this CU count, emulator); the two times are
// microseconds per grouped launch (0 when skipped)
extern "C" int mi355tts_dispatch_selfcheck(mi355tts_ctx* ctx, int* state, float* plain_us, float* snake_us) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  CHECK(dispatch_selfcheck_run(ctx));
  if (state) *state = ctx->selfcheck_state.load();
  if (plain_us) *plain_us = ctx->selfcheck_plain_us;
  if (snake_us) *snake_us = ctx->selfcheck_snake_us;
  return 0;
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
  if (std::strcmp(name, "adaptive_schedule") == 0) {
    ctx->adaptive_schedule = value != 0;
    return 0;
  }
  if (std::strcmp(name, "mrf_small") == 0) {
    ctx->mrf_small = value != 0;
    return 0;
  }
  if (std::strcmp(name, "gate16") == 0) {
    ctx->gate16 = value != 0;
    return 0;
  }
  if (std::strcmp(name, "call_coalesce") == 0) {
    ctx->call_coalesce = value < 0 ? 0 : value;
    return 0;
  }
  if (std::strcmp(name, "call_coalesce_window_us") == 0) {
    ctx->call_coalesce_window_us = value < 0 ? 0 : value;
    return 0;
  }
  if (std::strcmp(name, "glow_fuse") == 0) {
    ctx->glow_fuse = value != 0;
    return 0;
  }
  if (std::strcmp(name, "mrf_group") == 0) {
    ctx->mrf_group = value != 0;
    return 0;
  }
  if (std::strcmp(name, "rb_conv") == 0) {
    ctx->rb_conv = value != 0;
    return 0;
  }
  if (std::strcmp(name, "group_snake") == 0) {
    ctx->group_snake = value != 0;
    return 0;
  }
  if (std::strcmp(name, "group_promote") == 0) {
    ctx->group_promote = value != 0;
    return 0;
  }
  if (std::strcmp(name, "rb_pair") == 0) {
    ctx->rb_pair = value != 0;
    return 0;
  }
  if (std::strcmp(name, "gate16_wide") == 0) {
    ctx->gate16_wide = value < 0 ? 0 : value;
    return 0;
  }
  if (std::strcmp(name, "voc_out") == 0) {
    ctx->voc_out = value != 0;
    return 0;
  }
  if (std::strcmp(name, "serial_branches") == 0) {
    ctx->serial_branches = value != 0;
    return 0;
  }
  if (std::strcmp(name, "sync_mode") == 0) {  // process-wide: how a caller thread waits for its stream (host_context.h, mi355_sync)
    if (value < 0 || value > 3) return fail(MI355TTS_ERR_INVALID, "sync_mode %d outside [0, 3]", value);
    g_sync_mode.store(value, std::memory_order_relaxed);
    return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "unknown option '%s'", name);
}
extern "C" int mi355tts_call_coalesce_default(void) { return MI355TTS_CALL_COALESCE_DEFAULT; }
extern "C" int mi355tts_coalesce_stats(mi355tts_ctx* ctx, int64_t* passes, int64_t* rows) {
  if (!ctx || !passes || !rows) return fail(MI355TTS_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ctx->join_mu);
  *passes = ctx->join_passes;
  *rows = ctx->join_rows;
  return 0;
}
extern "C" int mi355tts_profile_reset(mi355tts_ctx* ctx) {
  if (!ctx) return fail(MI355TTS_ERR_INVALID, "ctx null");
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto& a : ctx->prof) a = mi355tts_ctx::Acc();
  for (auto& m : ctx->prof_kn) m.clear();
  for (auto& k : ctx->kn) k.store(0, std::memory_order_relaxed);
  return 0;
}
// {"kernel name": launches, ...} since the last mi355tts_profile_reset — counted whether profiling is on or not
extern "C" int mi355tts_kernel_counts_json(mi355tts_ctx* ctx, char* buf, int cap) {
  if (!ctx || !buf || cap <= 2) return fail(MI355TTS_ERR_INVALID, "bad argument");
  std::string s = "{";
  for (int i = 0; i < KN_COUNT; ++i) {
    char tmp[128];
    std::snprintf(tmp, sizeof(tmp), "%s\"%s\": %lld", i ? ", " : "", kname_name[i], ctx->kn[i].load(std::memory_order_relaxed));
    s += tmp;
  }
  s += "}";
  if ((int)s.size() + 1 > cap) return fail(MI355TTS_ERR_TOO_SMALL, "kernel-count buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}
// What the two event records of a ProfScope cost by themselves: `pairs` empty pairs (hipEventRecord a, hipEventRecord b,
// nothing between) on an idle stream of this context's device, median elapsed in microseconds.  A profiled launch's event time
// is its kernel's duration plus at least this (4.5 us on MI355X; rocprofv3's kernel durations do not contain it), so bench.py
// reports its event-timed launch durations with and without it.
extern "C" int mi355tts_profile_event_overhead(mi355tts_ctx* ctx, int pairs, double* us_out) {
  if (!ctx || !us_out || pairs < 1 || pairs > 4096) return fail(MI355TTS_ERR_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  hipStream_t st = nullptr;
  HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<float> el;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  for (int i = 0; i < pairs + 3 && e == hipSuccess; ++i) {
    hipEventRecord(a, st);
    hipEventRecord(b, st);
    e = hipStreamSynchronize(st);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (i >= 3) el.push_back(ms);  // (the first records of a new stream are slower)
  }
  if (a) hipEventDestroy(a);
  if (b) hipEventDestroy(b);
  hipStreamDestroy(st);
  if (e != hipSuccess) return fail(MI355TTS_ERR_HIP, "event overhead: %s", hipGetErrorString(e));
  std::sort(el.begin(), el.end());
  *us_out = 1000.0 * (double)el[el.size() / 2];
  return 0;
}
// The same sums per kernel NAME and launch sub-key (output rows of a conv launch / channels of a fused pair): {"class": {"name/sub":
// {"launches": n, "ms": t, "flop": f}, ...}, ...}; launches of kernels without a counted name are filed under "-".  bench.py's
// `roofline.by_kernel` (a driver record on an unknown box can then be compared with the builder's kernel by kernel).
extern "C" int mi355tts_profile_kernels_json(mi355tts_ctx* ctx, char* buf, int cap) {
  if (!ctx || !buf || cap <= 2) return fail(MI355TTS_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  std::string s = "{";
  bool first_cls = true;
  for (int i = 0; i < KC_COUNT; ++i) {
    if (ctx->prof_kn[i].empty()) continue;
    s += first_cls ? "\"" : ", \"";
    first_cls = false;
    s += kclass_name[i];
    s += "\": {";
    bool first = true;
    for (const auto& kv : ctx->prof_kn[i]) {
      char tmp[256];
      const int kn = kv.first.first;
      std::snprintf(tmp, sizeof(tmp), "%s\"%s/%d\": {\"launches\": %lld, \"ms\": %.6f, \"flop\": %.6e}", first ? "" : ", ",
                    kn >= 0 && kn < KN_COUNT ? kname_name[kn] : "-", kv.first.second, kv.second.launches, kv.second.ms, kv.second.flop);
      s += tmp;
      first = false;
    }
    s += "}";
  }
  s += "}";
  if ((int)s.size() + 1 > cap) return fail(MI355TTS_ERR_TOO_SMALL, "profile buffer too small");
  std::memcpy(buf, s.c_str(), s.size() + 1);
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
