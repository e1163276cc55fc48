/* mi355tts — C ABI of the MI355X-native Larynx hot path
 * (phoneme ids -> GlowTTS -> mel transform -> HiFi-GAN -> waveform).
 *
 * This is what a Larynx maintainer binds (ctypes) behind the
 * reference's own model interface.  Every entry point names the reference
 * interface it replaces (paths relative to rhasspy/larynx v1.1.0):
 *
 *   larynx/constants.py:62-72   TextToSpeechModel.phonemes_to_mels
 *   larynx/constants.py:90-100  VocoderModel.mels_to_audio
 *   larynx/glow_tts.py:109-170  GlowTextToSpeech.phonemes_to_mels  (ORT feed dict :161-168)
 *   larynx/hifi_gan.py:130-169  HiFiGanVocoder.mels_to_audio
 *   larynx/__init__.py:214-285  _sentence_task (mel transforms :242-249)
 *   larynx/audio.py:83-125      AudioSettings.denormalize/db_to_amp/
 *                               dynamic_range_compression, audio_float_to_int16
 *   glow_tts/checkpoint.py:26-68, hifi_gan/checkpoint.py:36-70   load_checkpoint
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative mi355tts_status, never throws or aborts;
 * mi355tts_last_error() gives the calling thread's last message.  All entry
 * points are re-entrant: many host threads may call into one context
 * concurrently (the reference calls its models from a ThreadPoolExecutor,
 * larynx/__init__.py:146); each call runs on its own HIP stream + workspace.
 * Tensors are row-major, channel-major like the reference's `[1, 80, F]` mels:
 * [batch][channel][time], time fastest.
 */
#ifndef MI355TTS_H
#define MI355TTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355TTS_ABI_VERSION 2 /* 2: n_speakers / gin_channels in mi355tts_glow_hparams, the *_speakers entry points */

typedef enum {
  MI355TTS_OK = 0,
  MI355TTS_ERR_INVALID = -1,   /* bad argument / unsupported hyper-parameter  */
  MI355TTS_ERR_HIP = -2,       /* a HIP runtime call failed (message has it)   */
  MI355TTS_ERR_NOMEM = -3,
  MI355TTS_ERR_TOO_SMALL = -4, /* caller-provided buffer too small             */
  MI355TTS_ERR_NO_MODEL = -5
} mi355tts_status;

typedef struct mi355tts_ctx mi355tts_ctx; /* one per process per GPU              */
typedef struct mi355tts_mel mi355tts_mel; /* device-resident mel batch             */

/* glow_tts/config.py:36-62 (ModelConfig) + audio.mel_channels */
typedef struct {
  int32_t num_symbols, hidden_channels, filter_channels, filter_channels_dp;
  int32_t kernel_size, n_blocks_dec, n_layers_enc, n_heads;
  int32_t dilation_rate, kernel_size_dec, n_block_layers, n_sqz;
  int32_t prenet, window_size, n_split, mel_channels;
  int32_t prenet_kernel_size, prenet_layers; /* glow_tts/models.py:96-97 (5, 3) */
  /* multi-speaker voices (glow_tts/config.py:56,60; models.py:304-306): n_speakers > 1 with gin_channels in [1, 1024], or a
   * single-speaker voice: n_speakers <= 1 and gin_channels = 0 */
  int32_t n_speakers, gin_channels;
} mi355tts_glow_hparams;

/* hifi_gan/config.py:29-41 (ModelConfig) */
#define MI355TTS_MAX_STAGES 8
typedef struct {
  int32_t resblock_type; /* 1 = ResBlock1, 2 = ResBlock2                       */
  int32_t num_upsamples;
  int32_t upsample_rates[MI355TTS_MAX_STAGES];
  int32_t upsample_kernel_sizes[MI355TTS_MAX_STAGES];
  int32_t upsample_initial_channel;
  int32_t num_kernels;
  int32_t resblock_kernel_sizes[MI355TTS_MAX_STAGES];
  int32_t num_dilations;
  int32_t resblock_dilations[MI355TTS_MAX_STAGES][MI355TTS_MAX_STAGES];
  int32_t num_mels;
} mi355tts_hifigan_hparams;

/* larynx/audio.py:26-50 (the fields the mel transforms read) */
typedef struct {
  int32_t signal_norm, symmetric_norm, clip_norm;
  int32_t convert_db_to_amp, do_dynamic_range_compression;
  float min_level_db, max_norm, ref_level_db, spec_gain;
} mi355tts_audio_settings;

/* flags for the inference calls */
#define MI355TTS_IN_DEVICE 1u  /* ids / noise / mel input pointers are device memory */
#define MI355TTS_OUT_DEVICE 2u /* waveform output pointers are device memory          */

int mi355tts_abi_version(void);
const char* mi355tts_last_error(void);

int mi355tts_create(int device, mi355tts_ctx** out);
void mi355tts_destroy(mi355tts_ctx* ctx);

/* ---- weights ---------------------------------------------------------------
 * A model is loaded from ONE flat fp32 blob: the reference checkpoint's tensors
 * (weight-norm folded, `remove_weight_norm` / `store_inverse` semantics) laid end
 * to end in the order the manifest enumerates.  `*_manifest` returns the i-th
 * tensor's reference state-dict name and element count (1 = past the end), so
 * the host-side converter never hard-codes the order.  `on_device` != 0 means
 * `blob` is device memory (e.g. the receive buffer of an RCCL broadcast).
 * Replaces load_checkpoint + .eval() in larynx/glow_tts.py:66-95 and
 * larynx/hifi_gan.py:71-100. */
int mi355tts_glow_manifest(const mi355tts_glow_hparams* hp, int index, char* name, int name_cap, int64_t* numel);
int mi355tts_hifigan_manifest(const mi355tts_hifigan_hparams* hp, int index, char* name, int name_cap, int64_t* numel);
int mi355tts_load_glow(mi355tts_ctx* ctx, const mi355tts_glow_hparams* hp, const float* blob, int64_t numel,
                       int on_device, int* model_out);
int mi355tts_load_hifigan(mi355tts_ctx* ctx, const mi355tts_hifigan_hparams* hp, const float* blob, int64_t numel,
                          int on_device, int* model_out);
int mi355tts_unload(mi355tts_ctx* ctx, int model);
/* Utterance-level data parallelism: the ONE collective of the path — rank `root`'s folded weight
 * blob (device memory, `numel` floats: what mi355tts_load_*(on_device = 1) ingests) is broadcast over xGMI to the
 * same-sized device buffer of every rank of the caller's RCCL communicator (`nccl_comm` is an ncclComm_t).  The
 * library does not link RCCL: ncclBroadcast is resolved from the RCCL library already loaded in the caller's
 * process (`rccl_library`: its path or soname, NULL = "librccl.so.1").  The reference is a single process and has
 * no counterpart; a Python host broadcasts with torch.distributed instead (larynx_amd/sharding.py). */
int mi355tts_broadcast_weights(mi355tts_ctx* ctx, void* nccl_comm, int root, float* device_blob, int64_t numel,
                               const char* rccl_library);
/* The reference's `half` switch (TextToSpeechModelConfig.half / VocoderModelConfig.half,
 * larynx/constants.py:58,85; `.half()` at larynx/glow_tts.py:90-91, larynx/hifi_gan.py:96-97).
 * MI355TTS_PRECISION_F32 (default): exact f32 MFMA everywhere — the parity mode.
 * MI355TTS_PRECISION_F16: the native 16-bit mode, the analogue of the reference's `.half()` on the generator (hifi_gan/models.py
 * :186-202 under half weights and activations): fp16 weights, fp16 activation planes in HBM between ALL layers of the vocoder
 * (conv_pre, upsamplers, every ResBlock conv of every stage, conv_post), ONE v_mfma_f32_32x32x16_f16 per product, f32 accumulation;
 * bias, residual and activation are applied to the f32 accumulator before its one rounding to fp16 (csrc/conv_f16.h,
 * csrc/hifigan_f16.h).  Accuracy is that of a half-precision forward (tests compare with the reference's own generator run
 * under .half() / .bfloat16()).  Returns MI355TTS_ERR_INVALID with the reason when the vocoder's geometry is not covered
 * (channel counts not multiples of 8, upsampler kernel != 2 x stride, taps outside 3 / 5 / 7 / 11).
 * MI355TTS_PRECISION_BF16X3: the accurate reduced mode — the HiFi-GAN ResBlock convs and upsamplers run on the bf16 matrix cores
 * with split operands (x = hi + lo, three bf16 MFMAs per product, f32 accumulate, f32 planes): ~1e-5 relative error per layer.
 * MI355TTS_PRECISION_BF16: the BF16X3 kernels with ONE bf16 MFMA per product (kept for A/B runs).
 * GlowTTS models (the reference calls `.half()` on the whole FlowGenerator, larynx/glow_tts.py:90-91): MI355TTS_PRECISION_F16
 * puts the decoder's WaveNets — every gate conv and res_skip of every coupling block (glow_tts/layers.py:138-162): 84 % of the
 * acoustic model's FLOPs and 97 of its ~140 launches — on the fp16 matrix cores, ONE launch per block (csrc/wn_f16.h: fp16 weights,
 * hidden state and gated activations, f32 accumulation, f32 skip sum).  The encoder, the duration predictor (so the FRAME COUNTS
 * are the f32 model's: the reference's whole-model .half() changes them on 3 of 8 golden cases), the start / end convs, the
 * affine coupling, InvConvNear and ActNorm stay f32.  Accuracy: the mel within the error of the reference's own decoder under
 * .half() (tests/golden/glow_half_reference.json, made by oracle/make_golden_glow_half.py).  The kernel covers hidden_channels
 * 192 (the released voices; 32 for tests), kernel_size_dec 5, dilation_rate 1, single-speaker models; on any other GlowTTS
 * geometry, and for the split-bf16 requests on every GlowTTS model, the request changes nothing and returns
 * MI355TTS_PRECISION_NOOP (= 1, a positive status: not an error, not a silent success).  MI355TTS_PRECISION_F32 returns 0. */
#define MI355TTS_PRECISION_F32 0
#define MI355TTS_PRECISION_BF16X3 1
#define MI355TTS_PRECISION_BF16 2
#define MI355TTS_PRECISION_F16 3
#define MI355TTS_PRECISION_NOOP 1 /* return value: accepted, no effect on this model */
int mi355tts_model_set_precision(mi355tts_ctx* ctx, int model, int precision);

/* ---- GlowTTS: replaces GlowTextToSpeech.phonemes_to_mels --------------------
 * ids [B][ids_ld] int64 (row b valid for id_lens[b] entries; the reference's
 * "input" / "input_lengths"), scales as in the reference's "scales" input.
 * noise: optional N(0,1) tensor [B][mel_channels][noise_ld] standing in for
 * torch.randn_like (glow_tts/models.py:348) — parity mode; NULL draws from a
 * counter-based generator keyed by `seed` (row b of a batch: the stream seed + b — a batched call consumes the B
 * consecutive streams seed .. seed + B - 1, so successive batched calls must advance `seed` by at least B, or pass explicit
 * per-row seeds to mi355tts_glow_infer_rows, to draw independent fields).  `audio` (optional) additionally
 * produces the vocoder-input mel (the three numpy transforms of _sentence_task).
 * The frame count is data dependent; the result stays on the device. */
int mi355tts_glow_infer(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                        float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                        const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out);

/* The same with one noise-stream seed PER ROW (host array [B]) for the device generator: row b draws the field a
 * batch-1 call with seed = row_seeds[b] draws, so a work list cut into micro-batches (the utterance shards of
 * larynx/__init__.py:146-157 run as padded batches) gives every utterance the audio it gets on its own.
 * (mi355tts_glow_infer's rows use seed + b.) */
int mi355tts_glow_infer_rows(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                             float noise_scale, float length_scale, const uint64_t* row_seeds,
                             const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out);

/* Multi-speaker voices: the reference's `speaker_id` setting (larynx/glow_tts.py:116-130 -> `g` of
 * FlowGenerator.forward, glow_tts/models.py:318-319: g = F.normalize(emb_g(speaker)), which conditions every WaveNet
 * layer of the decoder, layers.py:141-154, and is concatenated to the duration predictor's input, models.py:128-132).
 * `speaker_ids` (host array [B], one speaker per row, each in [0, n_speakers)) is REQUIRED for a model loaded with
 * n_speakers > 1 and must be NULL for a single-speaker model — the reference fails on both mismatches too (no emb_g /
 * a duration predictor that expects hidden + gin input channels).  mi355tts_glow_infer / _rows / mi355tts_synthesize on
 * a multi-speaker model return MI355TTS_ERR_INVALID.  Everything else as in mi355tts_glow_infer; `row_seeds` (optional,
 * host [B]) as in mi355tts_glow_infer_rows (then `seed` is ignored). */
int mi355tts_glow_infer_speakers(mi355tts_ctx* ctx, int glow, const int64_t* ids, const int32_t* id_lens, int B, int ids_ld,
                                 float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                                 const uint64_t* row_seeds, const int32_t* speaker_ids, const mi355tts_audio_settings* audio,
                                 uint32_t flags, mi355tts_mel** out);

int mi355tts_mel_batch(const mi355tts_mel* mel);
int mi355tts_mel_channels(const mi355tts_mel* mel);
int mi355tts_mel_max_frames(const mi355tts_mel* mel);
int mi355tts_mel_frames(const mi355tts_mel* mel, int32_t* frames /* [B] */);
/* which: 0 = raw GlowTTS output, 1 = vocoder input.  dst is host [B][M][ld], ld >= max_frames */
int mi355tts_mel_copy(const mi355tts_mel* mel, int which, float* dst, int ld);
void mi355tts_mel_free(mi355tts_mel* mel); /* must precede mi355tts_destroy() of the owning context */
/* Wrap a host (or device) mel [B][M][ld]; apply the mel transforms iff `audio` != NULL. */
int mi355tts_mel_from_buffer(mi355tts_ctx* ctx, const float* mel, const int32_t* frames, int B, int M, int ld,
                             const mi355tts_audio_settings* audio, uint32_t flags, mi355tts_mel** out);

/* ---- HiFi-GAN: replaces HiFiGanVocoder.mels_to_audio ------------------------
 * Consumes the vocoder-input mel; writes per row frames[b]*hop samples (tail up
 * to wav_ld zero-filled).  wav_f32 (optional) is the generator output before
 * audio_float_to_int16; wav_i16 (optional) after it.  wav_ld >= max_frames*hop.
 * denoiser_strength > 0 additionally runs the reference's spectral-subtraction
 * denoiser (larynx/hifi_gan.py:152-203, STFT helpers larynx/audio.py:232-306) on
 * the device before the int16 conversion; its bias spectrum is computed once per
 * vocoder from an all-zero mel, as `maybe_init_denoiser` does. */
int mi355tts_hifigan_hop(mi355tts_ctx* ctx, int vocoder);
int mi355tts_hifigan_infer(mi355tts_ctx* ctx, int vocoder, const mi355tts_mel* mel, float denoiser_strength,
                           float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags);

/* The same call with the SSML pause padding `_sentence_task` applies on the host with np.pad
 * (larynx/__init__.py:277-283) done by the int16 kernel: per row `pad_before` zero samples,
 * the audio, then zeros (at least `pad_after`) up to wav_ld.  wav_ld >= pad_before +
 * max_frames*hop + pad_after. */
int mi355tts_hifigan_infer_padded(mi355tts_ctx* ctx, int vocoder, const mi355tts_mel* mel, float denoiser_strength,
                                  float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags, int32_t pad_before,
                                  int32_t pad_after);

/* ---- fused call: replaces the model half of _sentence_task -------------------
 * (larynx/__init__.py:229-283: phonemes_to_mels -> mel transforms -> mels_to_audio -> pause
 * padding) with ONE call on one stream: arguments as in mi355tts_glow_infer +
 * mi355tts_hifigan_infer_padded.  `frames_out[b]` receives each row's mel frame count (row b
 * of the output holds pad_before + frames_out[b]*hop samples of padding + audio, then zeros).
 * The frame count is data dependent: if `wav_ld` turns out too small the call returns
 * MI355TTS_ERR_TOO_SMALL with frames_out filled, and the caller retries with a larger buffer.
 * MI355TTS_IN_DEVICE applies to ids / noise, MI355TTS_OUT_DEVICE to the waveform pointers. */
int mi355tts_synthesize(mi355tts_ctx* ctx, int glow, int vocoder, const int64_t* ids, const int32_t* id_lens, int B,
                        int ids_ld, float noise_scale, float length_scale, const float* noise, int noise_ld, uint64_t seed,
                        const mi355tts_audio_settings* audio, float denoiser_strength, int32_t pad_before,
                        int32_t pad_after, int32_t* frames_out, float* wav_f32, int16_t* wav_i16, int64_t wav_ld,
                        uint32_t flags);

/* mi355tts_synthesize for a multi-speaker voice: `speaker_ids` as in mi355tts_glow_infer_speakers. */
int mi355tts_synthesize_speakers(mi355tts_ctx* ctx, int glow, int vocoder, const int64_t* ids, const int32_t* id_lens, int B,
                                 int ids_ld, float noise_scale, float length_scale, const float* noise, int noise_ld,
                                 uint64_t seed, const int32_t* speaker_ids, const mi355tts_audio_settings* audio,
                                 float denoiser_strength, int32_t pad_before, int32_t pad_after, int32_t* frames_out,
                                 float* wav_f32, int16_t* wav_i16, int64_t wav_ld, uint32_t flags);

/* Serving set-up (the reference warms its model caches the same way, larynx/__init__.py:290,412):
 * pre-create `workers` per-call workers (stream, side streams, pinned staging) and size
 * their workspaces and the result-block pool for calls of up to max_batch rows x max_ids ids
 * x max_frames mel frames (+ max_pad_samples of pause padding; denoiser != 0 also sizes the
 * STFT scratch), so that no steady-state call allocates device memory.  glow / vocoder may
 * be 0 to size for one model only. */
int mi355tts_reserve(mi355tts_ctx* ctx, int workers, int glow, int vocoder, int max_batch, int max_ids, int max_frames,
                     int denoiser, int max_pad_samples);
/* Measurement: the hardware-queue group of every worker of the context, in creation order (groups[i] for worker i; -1 = not probed:
 * a worker created on demand after the last mi355tts_reserve).  The HIP runtime deals its few hardware queues (4) to streams as
 * they are first used, and two streams of one queue run their kernels one after the other; mi355tts_reserve measures which of its
 * worker streams share a queue (one kernel waits for a host flag on stream A while another raises a flag on stream B: if B's does
 * not come up, B sits behind A) and a call then takes the free worker whose queue carries the fewest calls.  Which worker a call
 * gets never changes its result.  Returns the number of workers (only `capacity` entries are written).  MI355TTS_NO_QUEUE_PROBE=1
 * skips the measurement (every worker -1: the most recently released free worker is taken, as before round 6).
 * MI355TTS_QUEUE_POLICY=1: an idle queue first, otherwise the BUSIEST one (as many calls as there are queues get a queue to
 * themselves, the rest share one) instead of the least busy one: measured better only for long runs of the fp16 mode, whose calls
 * are launch-latency chains (1228 against 1166 utterances/s in 200-utterance regions; 1134 against 1149 in the 20-utterance
 * regions; f32: 289 against 294 / 296.6 both) — profiles/r06_queue_map.txt.  No reference
 * counterpart (the reference has one model instance and Python threads: larynx/__init__.py:146-157). */
int mi355tts_worker_queue_groups(mi355tts_ctx* ctx, int32_t* groups, int capacity);

/* ---- single operators (kernel-level parity tests, drop-in conv) ------------- */
/* y[B][Cout][L] = act_out(bias + conv1d(lrelu_slope(x[B][Cin][L]), w[Cout][Cin][K], dilation, "same" padding)) */
int mi355tts_op_conv1d(mi355tts_ctx* ctx, const float* x, int B, int Cin, int L, const int32_t* lens, const float* w,
                       const float* bias, int Cout, int K, int dilation, float in_slope, int out_act, float* y);
/* y[B][Cout][L*stride] = bias + conv_transpose1d(lrelu_slope(x), w[Cin][Cout][K], stride, padding=(K-stride)/2) */
int mi355tts_op_conv_transpose1d(mi355tts_ctx* ctx, const float* x, int B, int Cin, int L, const float* w,
                                 const float* bias, int Cout, int K, int stride, float in_slope, float* y);

/* out[B][N] = denoise(wav[B][N], bias_spec[513], strength): the STFT kernels alone
 * (host buffers), N a multiple of 256 and > 1024 */
int mi355tts_op_denoise(mi355tts_ctx* ctx, const float* wav, int B, int64_t N, const float* bias_spec, float strength,
                        float* out);

/* out[B][C][T] (host) = the device noise generator's N(0,1) draw for (seed, row, channel, frame) —
 * the stand-in for torch.randn_like (glow_tts/models.py:348) that mi355tts_glow_infer uses when
 * `noise` is NULL; exposed so its distribution can be tested. */
int mi355tts_op_gauss_noise(mi355tts_ctx* ctx, uint64_t seed, int B, int C, int T, float* out);

/* Kernel micro-benchmark: `iters` back-to-back launches of the conv kernel on
 * device-resident random data of the given geometry (tile_shape -1 = the
 * launcher's own choice, 0/1/2 = pinned), timed with HIP events on the launch
 * stream; *ms_per_launch receives the average. */
int mi355tts_bench_conv1d(mi355tts_ctx* ctx, int B, int Cin, int Cout, int K, int dilation, int L, int tile_shape,
                          int iters, float* ms_per_launch);

/* ---- measurement -------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the
 * call's own stream and accumulated per kernel class.  `mi355tts_profile_json`
 * writes {"class": {"launches": n, "ms": t, "flop": f}, ...}. */
int mi355tts_set_profiling(mi355tts_ctx* ctx, int enabled);
/* Options, by what they may change in a RESULT.
 *
 * (1) Schedule options that compute THE SAME BITS under every setting (the same tiles run the same arithmetic; only which
 * launch, stream or workgroup order carries them changes):
 *   "serial_branches" (0/1, default 0) — the three MRF ResBlock chains of a HiFi-GAN stage one after another, one conv per
 *     launch (per-kernel timing); "mrf_group" (0/1, default 1) — the same-geometry launches of the three chains as one grouped
 *     launch; "adaptive_schedule" (0/1, default 0) — while other calls are in flight, launch the members of a group one by one;
 *   "rb_conv" (0/1, default 1) — the grouped 128-row ResBlock launches on the continuous-stream tile of rb_conv.h (0 = the
 *     chunked tile of conv_mfma.h);
 *   "group_snake" (0/1, default 1) — the workgroup ORDER of fully resident grouped launches (a snake over the dispatcher's
 *     rounds).  The one option the library may change by itself: mi355tts_dispatch_selfcheck switches it off on a device where
 *     the order does not win;
 *   "gate16_wide" (default 512) — gate convs / 1 x 1 convs of passes with at least this many 16-row tiles (padded batches) take
 *     two / four row tiles per workgroup from one staged input tile (0 = never).
 *   Not an option, the same class: while ANOTHER call holds a worker of the context, a batch-1 grouped ResBlock launch with more
 *     128-column tiles than the chip holds at once runs on them (rb_group_kernel<11, 7, 3, 4>) instead of on 64-column tiles — the
 *     same chain per output element, so which one a call gets depends on the load and its result does not (environment
 *     MI355TTS_RB_NB4_MIN_TILES pins the threshold whatever the load; 0 = never).
 * (2) Tile options: another tile = another f32 SUMMATION ORDER; results equal to f32 round-off, never changed by the library
 * itself (a timing never picks a tile):
 *   "gate16", "glow_fuse", "mrf_small" (0/1, default 1) — the small-launch kernels of gate16.h / coltile.h / mrf_small.h
 *     (0 = the generic tiles);
 *   "rb_pair" (0/1, default 1) — the fused ResBlock steps of the 64- / 32-channel stages on the 4-wave tile without a k-split
 *     (rb_pair.h; 0 = the 8-wave k-split tile of resblock_pair.h).  In the fp16 mode: conv1 + conv2 of a ResBlock1 step as ONE launch
 *     (pair_f16.h; 0 = two launches — there the two forms give the SAME bits);
 *   "group_promote" (0/1, default 1) — at batch 1 the same-geometry ResBlock convs of a step that are too short for the 128-row
 *     tile's own threshold move to it when the round-robin deal of their (all resident) workgroups stays balanced: decided from
 *     the device's CU count and the launch geometry alone (0 = they keep the 64 x 32 k-split tile);
 *   "voc_out" (0/1, default 1) — conv_post + tanh + the rows' peaks as one dedicated launch and the delivery of the rows (pause |
 *     samples | zeros) as one more (voc_out.h; 0 = the generic conv tile, zero_tail, absmax, to_int16 and a copy / fill per
 *     piece of every row: float rows equal to f32 round-off, int16 within 1 LSB).
 * (0) "sync_mode" (0-3, default 3 or MI355TTS_SYNC_MODE; process-wide) — how a caller thread waits for its stream: 0 =
 * hipStreamSynchronize (spins a core), 1 = a blocking event, 2 = hipStreamQuery + 20 us sleeps, 3 = 60 us of polling, then
 * query + 20 us sleeps with the calling thread's timer slack lowered to 1 us FOR THE DURATION OF THE WAIT (prctl
 * PR_SET_TIMERSLACK on the caller's own thread, restored before the call returns; skipped where prctl is denied).
 * (3) "call_coalesce" / "call_coalesce_window_us" (below): with it on, WHICH tiles compute a batch-1 call depends on the calls
 * that happened to share its pass — results equal to f32 round-off across loads, not bit-reproducible. */
int mi355tts_set_option(mi355tts_ctx* ctx, const char* name, int value);
/* Option "call_coalesce" (lanes; 0 = off; the built-in default: mi355tts_call_coalesce_default): concurrent batch-1
 * mi355tts_synthesize calls (the reference's per-sentence thread pool, larynx/__init__.py:146-157, 187-190) become the rows of
 * fused padded calls — acoustic pass AND vocoder — with at most `lanes` such passes in flight: the callers waiting when a lane
 * frees ride one pass, each row with its own seed's noise stream, its own pause padding and its own output buffers.  A lone
 * caller never waits (its pass is its solitary call, same bits); a caller that finds a lane free while other passes are in
 * flight gathers followers for at most "call_coalesce_window_us" (default 300).  Calls with explicit noise, speaker ids,
 * B > 1 or more than 768 ids always run alone.  A row of a padded batch is computed by other tiles than its solitary call:
 * equal to f32 round-off (frames identical, int16 within 1 LSB), not bit-identical.  Counters since the context was created:
 * fused passes run, rows they carried. */
int mi355tts_call_coalesce_default(void);
int mi355tts_coalesce_stats(mi355tts_ctx* ctx, int64_t* passes, int64_t* rows);
/* One-off check of the dispatcher rule the batch-1 ResBlock schedule relies on (grouped launches whose workgroups are all
 * resident go out in a "snake" order that assumes workgroup i lands on CU i mod #CUs; the promotion of a step to the 128-row
 * tile relies on that order): the grouped launch of a 256-channel ResBlock step in the plain and in the snake order, timed on
 * this device.  Runs by itself on the first load of a vocoder with a >= 256-channel stage; where the snake is not at least as
 * fast (2 % margin) the ORDER option "group_snake" is switched off for the context — results are the same bits either way; the
 * promotion rule ("group_promote": a tile choice) is never touched by a timing.  state: 1 = kept, 2 = switched off, 3 = skipped
 * (MI355TTS_NO_SELFCHECK, unsuitable CU count), 4 = running on another thread; times in microseconds per launch.  No
 * reference counterpart. */
int mi355tts_dispatch_selfcheck(mi355tts_ctx* ctx, int* state, float* plain_us, float* snake_us);
int mi355tts_profile_reset(mi355tts_ctx* ctx);
int mi355tts_profile_json(mi355tts_ctx* ctx, char* buf, int cap);
/* The same sums per kernel NAME and launch sub-key (the output rows of a conv launch / the channels of a fused ResBlock step):
 * {"class": {"rb_group_kernel.snake/256": {"launches": n, "ms": t, "flop": f}, ...}, ...}; kernels without a counted name are
 * filed under "-".  No reference counterpart (bench.py's `roofline.by_kernel`). */
int mi355tts_profile_kernels_json(mi355tts_ctx* ctx, char* buf, int cap);
/* Launches per kernel NAME since the last mi355tts_profile_reset, {"rb_group_kernel": n, ...}; counted whether profiling is
 * on or not.  The class counters cannot tell a kernel from the fallback that would silently take its place (same launch count,
 * same bits by design): the device tests assert on these.  No reference counterpart (measurement only). */
int mi355tts_kernel_counts_json(mi355tts_ctx* ctx, char* buf, int cap);
/* Median elapsed time (microseconds) of `pairs` EMPTY event pairs on an idle stream: what the two hipEventRecord calls around a
 * profiled launch add to its event-timed duration (rocprofv3's kernel durations do not contain it). */
int mi355tts_profile_event_overhead(mi355tts_ctx* ctx, int pairs, double* us_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355TTS_H */
