/* TEST INFRASTRUCTURE — a plain C99 caller of include/mi355tts.h: what a cgo / JNI /
 * ctypes binding does, without Python in the way.  Sizes the weight blobs from the
 * manifests, fills them with a small LCG, loads a tiny GlowTTS + HiFi-GAN, synthesises
 * two sentences in one batch and prints a checksum line the pytest wrapper compares with
 * the same calls made through larynx_amd.ffi.
 *
 *   usage: caller <device>        (links against libmi355tts*.so)            */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355tts.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != 0) {                                                                      \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, mi355tts_last_error());        \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

static uint32_t lcg_state = 12345u;
static float lcg_unit(void) { /* uniform in [-1, 1) */
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)(lcg_state >> 8) / 8388608.0f - 1.0f;
}

static float* make_blob(int is_glow, const void* hp, int64_t* total_out) {
  char name[256];
  int64_t numel, total = 0;
  int i;
  for (i = 0;; ++i) {
    int rc = is_glow ? mi355tts_glow_manifest((const mi355tts_glow_hparams*)hp, i, name, (int)sizeof name, &numel)
                     : mi355tts_hifigan_manifest((const mi355tts_hifigan_hparams*)hp, i, name, (int)sizeof name, &numel);
    if (rc == 1) break; /* past the end */
    if (rc != 0) return NULL;
    total += numel;
  }
  float* blob = (float*)malloc((size_t)total * sizeof(float));
  int64_t pos = 0;
  for (i = 0;; ++i) {
    int rc = is_glow ? mi355tts_glow_manifest((const mi355tts_glow_hparams*)hp, i, name, (int)sizeof name, &numel)
                     : mi355tts_hifigan_manifest((const mi355tts_hifigan_hparams*)hp, i, name, (int)sizeof name, &numel);
    int64_t k;
    if (rc != 0) break;
    /* keep activations O(1): LayerNorm gains near 1, the pre-inverted 4x4s near identity, the rest small */
    for (k = 0; k < numel; ++k) {
      float v = 0.08f * lcg_unit();
      if (strstr(name, "gamma")) v += 1.0f;
      if (strstr(name, "_inv") && numel == 16 && (k % 5) == 0) v += 1.0f;
      blob[pos + k] = v;
    }
    pos += numel;
  }
  *total_out = total;
  return blob;
}

int main(int argc, char** argv) {
  const int device = argc > 1 ? atoi(argv[1]) : 0;
  if (mi355tts_abi_version() != MI355TTS_ABI_VERSION) {
    fprintf(stderr, "ABI version mismatch\n");
    return 1;
  }
  mi355tts_glow_hparams g;
  memset(&g, 0, sizeof g);
  g.num_symbols = 20; g.hidden_channels = 32; g.filter_channels = 48; g.filter_channels_dp = 24;
  g.kernel_size = 3; g.n_blocks_dec = 2; g.n_layers_enc = 1; g.n_heads = 2;
  g.dilation_rate = 1; g.kernel_size_dec = 5; g.n_block_layers = 2; g.n_sqz = 2;
  g.prenet = 1; g.window_size = 4; g.n_split = 4; g.mel_channels = 8;
  g.prenet_kernel_size = 5; g.prenet_layers = 3;
  mi355tts_hifigan_hparams v;
  memset(&v, 0, sizeof v);
  v.resblock_type = 1; v.num_upsamples = 2;
  v.upsample_rates[0] = 4; v.upsample_rates[1] = 2;
  v.upsample_kernel_sizes[0] = 8; v.upsample_kernel_sizes[1] = 4;
  v.upsample_initial_channel = 16; v.num_kernels = 2;
  v.resblock_kernel_sizes[0] = 3; v.resblock_kernel_sizes[1] = 7;
  v.num_dilations = 3;
  { int k, d; const int dil[3] = {1, 3, 5}; for (k = 0; k < 2; ++k) for (d = 0; d < 3; ++d) v.resblock_dilations[k][d] = dil[d]; }
  v.num_mels = 8;
  const mi355tts_audio_settings audio = {1, 1, 1, 1, 1, -100.0f, 1.0f, 20.0f, 1.0f};

  mi355tts_ctx* ctx = NULL;
  CHECK(mi355tts_create(device, &ctx));
  int64_t ng = 0, nv = 0;
  float* gb = make_blob(1, &g, &ng);
  float* vb = make_blob(0, &v, &nv);
  if (!gb || !vb) { fprintf(stderr, "manifest failed: %s\n", mi355tts_last_error()); return 1; }
  int glow = -1, voc = -1;
  CHECK(mi355tts_load_glow(ctx, &g, gb, ng, 0, &glow));
  CHECK(mi355tts_load_hifigan(ctx, &v, vb, nv, 0, &voc));
  /* a wrong-sized blob must be refused, not read out of bounds */
  { int bad = -1; if (mi355tts_load_glow(ctx, &g, gb, ng - 1, 0, &bad) == 0) { fprintf(stderr, "short blob accepted\n"); return 1; } }

  enum { B = 2, LD = 9 };
  const int64_t ids[B][LD] = {{3, 5, 9, 3, 12, 7, 3, 2, 0}, {3, 4, 18, 6, 2, 0, 0, 0, 0}};
  const int32_t lens[B] = {8, 5};
  mi355tts_mel* mel = NULL;
  CHECK(mi355tts_glow_infer(ctx, glow, &ids[0][0], lens, B, LD, 0.0f, 1.0f, NULL, 0, 1u, &audio, 0u, &mel));
  int32_t frames[B];
  CHECK(mi355tts_mel_frames(mel, frames));
  const int M = mi355tts_mel_channels(mel), F = mi355tts_mel_max_frames(mel);
  const int hop = mi355tts_hifigan_hop(ctx, voc);
  if (mi355tts_mel_batch(mel) != B || M != 8 || hop != 8 || F < frames[0] || F < frames[1]) { fprintf(stderr, "bad geometry\n"); return 1; }
  float* raw = (float*)malloc(sizeof(float) * B * M * F);
  CHECK(mi355tts_mel_copy(mel, 0, raw, F));
  const int64_t wav_ld = (int64_t)F * hop;
  float* wav = (float*)calloc((size_t)(B * wav_ld), sizeof(float));
  int16_t* pcm = (int16_t*)calloc((size_t)(B * wav_ld), sizeof(int16_t));
  CHECK(mi355tts_hifigan_infer(ctx, voc, mel, 0.0f, wav, pcm, wav_ld, 0u));
  /* too small an output row must be refused */
  if (mi355tts_hifigan_infer(ctx, voc, mel, 0.0f, wav, pcm, wav_ld - 1, 0u) != MI355TTS_ERR_TOO_SMALL) { fprintf(stderr, "short row accepted\n"); return 1; }
  double mel_sum = 0.0, wav_sum = 0.0;
  long pcm_sum = 0;
  int b, i;
  for (b = 0; b < B; ++b) {
    for (i = 0; i < M * F; ++i) mel_sum += raw[(size_t)b * M * F + i];
    for (i = 0; i < wav_ld; ++i) {
      wav_sum += wav[(size_t)b * wav_ld + i] * (double)((i % 7) + 1);
      pcm_sum += pcm[(size_t)b * wav_ld + i] * (long)((i % 5) + 1);
      if (i >= frames[b] * hop && (wav[(size_t)b * wav_ld + i] != 0.0f || pcm[(size_t)b * wav_ld + i] != 0)) { fprintf(stderr, "tail not zero\n"); return 1; }
    }
  }
  /* round-2 entry points: workers reserved up front; the fused one-call form with SSML pause padding must give the
   * two-call result shifted by pad_before; an output row that is too small is refused with the frame counts
   * filled in; the precision switch is accepted by both kinds of model and rejects unknown values */
  CHECK(mi355tts_reserve(ctx, 2, glow, voc, B, LD, 64, 0, 16));
  {
    enum { PB = 5, PA = 3 };
    const int64_t ld2 = wav_ld + PB + PA;
    int16_t* pcm2 = (int16_t*)calloc((size_t)(B * ld2), sizeof(int16_t));
    int32_t fr2[B] = {0, 0};
    CHECK(mi355tts_synthesize(ctx, glow, voc, &ids[0][0], lens, B, LD, 0.0f, 1.0f, NULL, 0, 1u, &audio, 0.0f, PB, PA, fr2, NULL, pcm2, ld2, 0u));
    for (b = 0; b < B; ++b) {
      if (fr2[b] != frames[b]) { fprintf(stderr, "fused call: frame count differs\n"); return 1; }
      for (i = 0; i < ld2; ++i) {
        const int j = i - PB;
        const int16_t want = (j >= 0 && j < wav_ld) ? pcm[(size_t)b * wav_ld + j] : 0;
        if (pcm2[(size_t)b * ld2 + i] != want) { fprintf(stderr, "fused call differs at row %d sample %d\n", b, i); return 1; }
      }
    }
    fr2[0] = fr2[1] = -1;
    if (mi355tts_synthesize(ctx, glow, voc, &ids[0][0], lens, B, LD, 0.0f, 1.0f, NULL, 0, 1u, &audio, 0.0f, PB, PA, fr2, NULL, pcm2, 8, 0u) !=
            MI355TTS_ERR_TOO_SMALL || fr2[0] != frames[0] || fr2[1] != frames[1]) { fprintf(stderr, "short fused row not reported\n"); return 1; }
    free(pcm2);
  }
  CHECK(mi355tts_model_set_precision(ctx, voc, MI355TTS_PRECISION_BF16X3));
  CHECK(mi355tts_model_set_precision(ctx, voc, MI355TTS_PRECISION_F32));
  /* GlowTTS: the split-bf16 request is reported as a no-op; F16 puts the decoder's WaveNets in fp16 where the kernel covers the
   * geometry (0) and is a reported no-op elsewhere; F32 returns 0 */
  if (mi355tts_model_set_precision(ctx, glow, MI355TTS_PRECISION_BF16X3) != MI355TTS_PRECISION_NOOP) { fprintf(stderr, "GlowTTS split-bf16 request not reported as a no-op\n"); return 1; }
  {
    int rcg = mi355tts_model_set_precision(ctx, glow, MI355TTS_PRECISION_F16);
    if (rcg != 0 && rcg != MI355TTS_PRECISION_NOOP) { fprintf(stderr, "GlowTTS fp16 request: unexpected status %d\n", rcg); return 1; }
  }
  CHECK(mi355tts_model_set_precision(ctx, glow, MI355TTS_PRECISION_F32));
  { /* the native fp16 vocoder: accepted where its tiles cover the geometry, refused with a reason where not */
    int rc16 = mi355tts_model_set_precision(ctx, voc, MI355TTS_PRECISION_F16);
    if (rc16 != 0 && rc16 != MI355TTS_ERR_INVALID) { fprintf(stderr, "fp16 request: unexpected status %d\n", rc16); return 1; }
    if (rc16 == 0) {
      CHECK(mi355tts_hifigan_infer(ctx, voc, mel, 0.0f, wav, pcm, wav_ld, 0u));
      for (int b = 0; b < B; ++b)
        for (int i = frames[b] * hop; i < wav_ld; ++i)
          if (wav[(size_t)b * wav_ld + i] != 0.0f || pcm[(size_t)b * wav_ld + i] != 0) { fprintf(stderr, "fp16: tail not zero\n"); return 1; }
      CHECK(mi355tts_model_set_precision(ctx, voc, MI355TTS_PRECISION_F32));
    }
  }
  if (mi355tts_model_set_precision(ctx, voc, 99) == 0) { fprintf(stderr, "unknown precision accepted\n"); return 1; }
  if (mi355tts_broadcast_weights(ctx, NULL, 0, NULL, 0, NULL) == 0) { fprintf(stderr, "null communicator accepted\n"); return 1; }
  printf("frames %d %d mel_sum %.6e wav_sum %.6e pcm_sum %ld\n", (int)frames[0], (int)frames[1], mel_sum, wav_sum, pcm_sum);
  mi355tts_mel_free(mel);
  CHECK(mi355tts_unload(ctx, glow));
  CHECK(mi355tts_unload(ctx, voc));
  mi355tts_destroy(ctx);
  free(gb); free(vb); free(raw); free(wav); free(pcm);
  return 0;
}
