// mi355tts host runtime — profiling scopes and the conv / fused-pair launchers (tile-shape choice)
// (one translation unit: included once by mi355tts.hip, after the kernel headers)
#pragma once

// ------------------------------------------------------------------ launch helpers
struct ProfScope {
  mi355tts_ctx* ctx;
  Worker* w;
  bool on;
  ProfEvent ev;
  hipStream_t st;
  ProfScope(mi355tts_ctx* c, Worker* wk, int cls, double flop, hipStream_t stream = nullptr)
      : ctx(c), w(wk), on(c->profiling.load() && !wk->quiet), st(stream ? stream : wk->stream) {
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
    ev.flop = flop * wk->flop_scale;
    g_last_kn = -1;
    g_last_sub = 0;
    hipEventRecord(ev.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    ev.kn = g_last_kn;
    ev.sub = g_last_sub;
    hipEventRecord(ev.b, st);
    w->events.push_back(ev);
  }
};

template <int K, int CI_C, int MB, int NB, int WN, int KS, int HALO, int EPI>
static void launch_conv_inst(hipStream_t s, dim3 grid, const ConvArgs& a) {
  kn_add(KN_CONV_MFMA);
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
//   W128  : 2 time-waves x 4 k-groups, 128 columns x ONE 32-row m-tile, 2 column blocks per wave — half the
//           weight bytes per MFMA of a 64-row tile and 4 k-groups; measured 7-14 % faster than the
//           4 x 2-wave 64-row tile it replaced, and equal or better than SMALL at the same tile count
//   NB2   : 4 time-waves x 2 k-groups, 256 columns (64x64 outputs per wave)
//   M128  : 4 row groups of waves (256 threads), no k-split: 128 rows x 64 columns from ONE staged input tile (the 32-row
//           shapes stage the same input once per m-tile: 4x at 128 channels) and no k-group reduction; ResBlock convs
//           whose rows are whole 128-row groups and that yield >= 256 128-column tiles.  (Measured against the 8-wave
//           128 x 128 form: +1.4 % on the class — finer granularity, four workgroups per CU.)
enum TileShape { TILE_SMALL = 0, TILE_W128 = 1, TILE_NB2 = 2, TILE_TINY = 3, TILE_LAST = 3, TILE_M128 = 4 };
static thread_local int g_pin_tile = -1;  // set by mi355tts_bench_conv1d only

// option "rb_conv" as the running call saw it (set by run_plan from the worker's snapshot: launch_conv_k has no worker)
static thread_local bool g_rb_conv_on = true;
template <int K, int EPI>
static int launch_conv_k(hipStream_t s, int MB, int shape, dim3 grid, const ConvArgs& a) {
  constexpr int HALO = ConvCfg<K>::HALO;
  constexpr int CI_SMALL = (K == 1) ? 64 : 32;
  constexpr bool PAIRED = (EPI == EPI_GATE || EPI == EPI_COUPLING);
  // the staged tile starts at the 4-aligned column t0 - roundup(pad, 4)
  if ((K - 1) * a.dil + ((4 - a.pad % 4) % 4) > HALO)
    return fail(MI355TTS_ERR_INVALID, "conv K=%d dilation=%d exceeds the staged halo", K, a.dil);
  if (a.x_ld % 4) return fail(MI355TTS_ERR_INVALID, "internal: activation row stride %d is not a multiple of 4", a.x_ld);
  if constexpr (EPI == EPI_LINEAR && K >= 3) {
    if (shape == TILE_M128) {
      kn_add(KN_CONV_M128);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_kernel<K, 16, 1, 2, 1, 1, HALO, EPI, 4>), grid, dim3(256), 0, s, a);
      return 0;
    }
  }
  if constexpr (EPI == EPI_UPSAMPLE) {
    if (shape == TILE_M128) {  // the polyphase upsampler's virtual rows, 128 per workgroup from one staged input tile
      if constexpr (K == 2) {
        // the continuous-stream tile (rb_conv.h; same bits): taps 2 — every upsampler of the shipped vocoders (k_u = 2 u)
        static const bool rb_off = [] { const char* e = std::getenv("MI355TTS_NO_RB_CONV"); return e && std::atoi(e) != 0; }();
        if (!rb_off && g_rb_conv_on && a.Cin % 16 == 0 && !a.res && !a.accum && a.alpha == 1.0f) {
          if (a.x2 && a.x3) hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_conv_kernel<2, 4, EPI_UPSAMPLE, true>), grid, dim3(256), 0, s, a);
          else if (!a.x2) hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_conv_kernel<2, 4, EPI_UPSAMPLE, false>), grid, dim3(256), 0, s, a);
          if ((a.x2 && a.x3) || !a.x2) {
            kn_add(KN_RB_CONV);
            return 0;
          }
        }
      }
      kn_add(KN_CONV_M128);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_mfma_kernel<K, 16, 1, 2, 1, 1, HALO, EPI, 4>), grid, dim3(256), 0, s, a);
      return 0;
    }
  }
  if (shape == TILE_M128) return fail(MI355TTS_ERR_INVALID, "internal: the 128-row tile is a ResBlock conv / upsampler shape");
  if (MB == 1) {
    if (shape == TILE_TINY) launch_conv_inst<K, 64, 1, 1, 1, 8, HALO, EPI>(s, grid, a);
    else if (shape == TILE_SMALL) launch_conv_inst<K, CI_SMALL, 1, 1, 2, 4, HALO, EPI>(s, grid, a);
    else if (shape == TILE_W128) launch_conv_inst<K, 32, 1, 2, 2, 4, HALO, EPI>(s, grid, a);
    else launch_conv_inst<K, 16, 1, 2, 4, 2, HALO, EPI>(s, grid, a);
    return 0;
  }
  if constexpr (!PAIRED) {
    if (shape == TILE_TINY) launch_conv_inst<K, 64, 2, 1, 1, 8, HALO, EPI>(s, grid, a);
    else if (shape == TILE_SMALL) launch_conv_inst<K, CI_SMALL, 2, 1, 2, 4, HALO, EPI>(s, grid, a);
    else if (shape == TILE_W128) return fail(MI355TTS_ERR_INVALID, "internal: the 128-column tile is one m-tile high");
    else launch_conv_inst<K, 16, 2, 2, 4, 2, HALO, EPI>(s, grid, a);
    return 0;
  }
  return fail(MI355TTS_ERR_INVALID, "paired epilogues run on 32-row tiles (MB == 1)");
}

// workgroups a launch must yield before the 128-row tile is used (tests lower it to reach the shape at small sizes)
static long long m128_min_tiles() {
  const char* e = std::getenv("MI355TTS_M128_MIN_TILES");
  return e ? std::atoll(e) : 256;
}

// A conv launch, decided but not yet issued: arguments, tile shape and grid.
struct ConvPlan {
  ConvArgs a;
  int K = 0, MB = 1, shape = TILE_TINY, epi = EPI_LINEAR, cls = 0;
  dim3 grid;
  double flop = 0;
  bool empty = true;
  int bf16 = 0;  // 0 = f32 kernel; 3 = split-bf16 kernel, 1 = plain bf16 kernel (conv_bf16.h); `shape` is then a Bf16Cfg
  int n_max = 0;        // columns of the implicit GEMM (longest row)
  bool pinned = false;  // the tile shape was forced (MI355TTS_FORCE_TILE, pin_tile): run_group leaves it alone
};

// Tile configurations of the split-bf16 kernel (all 4 waves; a wave owns 1 x NB blocks over all input channels)
enum Bf16Cfg {
  BF_A = 0,  // 4 x 1 waves, NB = 4: 128 rows x 128 columns
  BF_B = 1,  // 4 x 1 waves, NB = 1: 128 rows x  32 columns (few-tile launches: stage 0 of 'high' at batch 1; measured 2 % better than 64 columns)
  BF_C = 2,  // 2 x 2 waves, NB = 2:  64 rows x 128 columns (64-channel stages)
  BF_D = 3,  // 1 x 4 waves, NB = 1:  32 rows x 128 columns (32-channel stages; 256 columns would need 85 KB of LDS)
  BF_K = 4,  // 4 x 1 waves x 2 k-groups, NB = 4: 128 rows x 128 columns by 8 waves (few-tile launches with >= 64 channels in)
};

// `a` arrives with every tensor/epilogue field filled; this picks the tile and
// template instance.  n_max = largest GEMM-N extent over the batch rows.
static int plan_conv(const DevConv& c, ConvArgs a, int epi, int B, int n_max, int cls, int min_tiles, int host_len, ConvPlan* out,
                     int precision = 0) {
  out->empty = true;
  out->bf16 = 0;
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
  const bool half_on = (precision == MI355TTS_PRECISION_BF16X3 || precision == MI355TTS_PRECISION_BF16) && c.w16 && (a.x_ld % 4) == 0;
  const bool bf_linear = half_on && epi == EPI_LINEAR && !a.x2 && !a.y2 && a.split >= c.rows && a.out_act == ACT_NONE &&
                         (c.K == 3 || c.K == 5 || c.K == 7 || c.K == 11) &&
                         (c.K - 1) * a.dil + ((4 - a.pad % 4) % 4) <= (c.K == 3 ? 16 : c.K == 5 ? 28 : c.K == 7 ? 76 : 56);
  // the polyphase upsamplers (two taps) in the split-bf16 mode too: 0.24 ms of f32 work per 'high' utterance otherwise
  static const bool no_bf_ups = [] { const char* e = std::getenv("MI355TTS_NO_BF16_UPS"); return e && std::atoi(e) != 0; }();
  const bool bf_ups = half_on && !no_bf_ups && epi == EPI_UPSAMPLE && c.K == 2 && (c.K - 1) * a.dil + ((4 - a.pad % 4) % 4) <= 4;
  if (bf_linear || bf_ups) {
    a.w16 = c.w16;
    a.nslab = c.nslab16;
    a.rows_major = 0;
    if (epi == EPI_UPSAMPLE) {
      // deal ROW tiles to the XCDs when the weights are the bigger operand and the input fits an L2 (stage 0: 8.4 MB of
      // fragments against 1.3 MB of input) — as the f32 path does
      const double w_bytes = (double)c.mtiles16 * c.nslab16 * c.K * 2048.0, x_bytes = (double)c.Cin * (double)n_max * 4.0 * B;
      a.rows_major = (w_bytes > x_bytes && x_bytes < 3.0e6 && c.mtiles16 >= 32) ? 1 : 0;
    }
    int cfg, rows_t, cols_t;
    if (c.mtiles16 % 4 == 0) {
      const long long tiles_a = (long long)((n_max + 127) / 128) * (c.mtiles16 / 4) * B;
      static const bool no_k = [] { const char* e = std::getenv("MI355TTS_NO_BF_K"); return e && std::atoi(e) != 0; }();
      cfg = tiles_a >= 256 ? BF_A : (no_k ? BF_B : BF_K);
      rows_t = 128;
      cols_t = cfg == BF_B ? 32 : 128;
    } else if (c.mtiles16 == 2) {
      cfg = BF_C;
      rows_t = 64;
      cols_t = 128;
    } else {
      cfg = BF_D;
      rows_t = 32;
      cols_t = 128;
    }
    out->a = a;
    out->K = c.K;
    out->MB = 1;
    out->shape = cfg;
    out->epi = epi;
    out->cls = cls;
    out->bf16 = precision == MI355TTS_PRECISION_BF16 ? 1 : 3;
    out->n_max = n_max;
    out->pinned = false;
    out->grid = dim3((n_max + cols_t - 1) / cols_t, (c.mtiles16 * 32) / rows_t, B);
    out->flop = 2.0 * (double)c.Cout * c.Cin * (epi == EPI_UPSAMPLE ? c.K * a.up : c.K) * (double)n_max * B;
    out->empty = false;
    return 0;
  }
  int MB = c.MB;
  int ytiles = c.mtiles / MB;
  // Tile shape: the largest tile that still yields >= min_tiles workgroups, otherwise the
  // smallest tile.  1024 (4 per CU) is the measured sweet spot for a kernel that has the
  // chip to itself (tools/conv_sweep.py).
  const int rows32 = (c.rows + 31) / 32;  // m-tiles when a workgroup is one m-tile high (the 128-column shape)
  auto tiles = [&](int width) { return (long long)((n_max + width - 1) / width) * (width == 128 ? rows32 : ytiles) * B; };
  const long long want = min_tiles;
  int shape = TILE_TINY;
  if (tiles(256) >= want) shape = TILE_NB2;
  else if (tiles(128) >= want) shape = TILE_W128;
  else if (tiles(64) >= want) shape = TILE_SMALL;
  bool pinned = false;
  {  // tuning / test knob: MI355TTS_FORCE_TILE=0|1|2 pins the tile shape
    static const int forced = [] {
      const char* e = std::getenv("MI355TTS_FORCE_TILE");
      return e ? std::atoi(e) : -1;
    }();
    int f = forced;
    if (const char* dyn = std::getenv("MI355TTS_FORCE_TILE_DYNAMIC")) f = std::atoi(dyn);
    if (g_pin_tile >= 0) f = g_pin_tile;
    if (f >= TILE_SMALL && f <= TILE_LAST) {
      shape = f;
      pinned = true;
    }
  }
  // a launch that cannot even give every CU one workgroup: halve the row tile too
  // (32-row m-tiles are independent in the packed weights; paired epilogues need both)
  if (shape == TILE_TINY && MB == 2 && (epi == EPI_LINEAR || epi == EPI_UPSAMPLE) && tiles(32) < 256) {
    MB = 1;
    ytiles = rows32;
  }
  {
    // A 64-row upsampler (the last stage of 'high': 64 -> 32 channels x 2 phases, two taps) is bound by its input planes, not
    // by its matrix work (81 MB against 1.3 GFLOP): the 128-column shape is ONE m-tile high, so two workgroups stage every
    // input tile; the 64-row x 64-column shape stages it once.  MI355TTS_UPS64=0: the shape rule above (A/B runs).
    static const bool ups64 = [] { const char* e = std::getenv("MI355TTS_UPS64"); return !e || std::atoi(e) != 0; }();
    if (ups64 && !pinned && epi == EPI_UPSAMPLE && c.MB == 2 && rows32 == 2 && shape == TILE_W128 && tiles(64) >= 256) shape = TILE_SMALL;
  }
  if (shape == TILE_W128 && MB == 2) {
    MB = 1;
    ytiles = rows32;
  }
  {
    static const bool no_m128 = [] { const char* e = std::getenv("MI355TTS_NO_M128"); return e && std::atoi(e) != 0; }();
    static const bool no_m128u = [] { const char* e = std::getenv("MI355TTS_NO_M128_UPS"); return e && std::atoi(e) != 0; }();
    const bool resblock_ok = epi == EPI_LINEAR && cls == KC_RESBLOCK && c.K >= 3;
    const bool upsample_ok = epi == EPI_UPSAMPLE && !no_m128u && c.K >= 1 && c.K <= 3;  // 32 m-tiles re-stage the same input otherwise
    if (!no_m128 && !pinned && (resblock_ok || upsample_ok) && rows32 % 4 == 0 && c.rows == rows32 * 32 &&
        (long long)((n_max + 127) / 128) * (rows32 / 4) * B >= m128_min_tiles()) {
      shape = TILE_M128;
      MB = 1;
      ytiles = rows32 / 4;
    }
  }
  const int T_T = shape == TILE_TINY ? 32 : (shape == TILE_SMALL || shape == TILE_M128) ? 64 : (shape == TILE_NB2 ? 256 : 128);
  // Which operand the 8 XCD L2s replicate: dealing TIME tiles across the XCDs makes every L2 fetch all the
  // weights (8 W + X bytes from memory, and W must fit 4 MB or it is re-streamed per time tile); dealing ROW
  // tiles makes every L2 fetch the whole input and 1/8 of the weights (W + 8 X).  Rows when the weights are
  // the bigger operand and the input fits an L2 (GlowTTS launches, conv_pre, the stage-0 upsampler).
  {
    const double w_bytes = (double)c.mtiles * c.noct * c.K * 1024.0;
    const double x_bytes = (double)c.Cin * (double)n_max * 4.0 * B;
    a.rows_major = (w_bytes > x_bytes && x_bytes < 3.0e6 && ytiles >= 8) ? 1 : 0;
  }
  out->a = a;
  out->K = c.K;
  out->MB = MB;
  out->shape = shape;
  out->epi = epi;
  out->cls = cls;
  out->n_max = n_max;
  out->pinned = pinned;
  out->grid = dim3((n_max + T_T - 1) / T_T, ytiles, B);
  out->flop = 2.0 * (double)c.Cout * c.Cin * (epi == EPI_UPSAMPLE ? c.K * a.up : c.K) * (double)n_max * B;
  out->empty = false;
  return 0;
}

static int run_plan(mi355tts_ctx* ctx, Worker* w, const ConvPlan& p, hipStream_t stream = nullptr) {
  if (p.empty) return 0;
  hipStream_t s = stream ? stream : w->stream;
  ProfScope ps(ctx, w, p.cls, p.flop, s);
  g_last_sub = p.a.rows;
  const ConvArgs& a = p.a;
  const int MB = p.MB, shape = p.shape;
  const dim3 grid = p.grid;
  int rc = 0;
  g_rb_conv_on = w->o_rb_conv;
  g_kn = w->quiet ? nullptr : ctx->kn;
  if (p.bf16) {
    kn_add(KN_CONV_BF16);
#define BF16_LAUNCH_T(KK, TT)                                                                                                      \
  if (shape == BF_A) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<KK, 1, 4, 4, 1, ConvCfg<KK>::HALO, TT>), grid, dim3(256), 0, s, a);      \
  else if (shape == BF_B) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<KK, 1, 1, 4, 1, ConvCfg<KK>::HALO, TT>), grid, dim3(256), 0, s, a); \
  else if (shape == BF_C) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<KK, 1, 2, 2, 2, ConvCfg<KK>::HALO, TT>), grid, dim3(256), 0, s, a); \
  else if (shape == BF_K) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<KK, 1, 4, 4, 1, ConvCfg<KK>::HALO, TT, 2>), grid, dim3(512), 0, s, a); \
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<KK, 1, 1, 1, 4, ConvCfg<KK>::HALO, TT>), grid, dim3(256), 0, s, a)
#define BF16_LAUNCH(KK)            \
  if (p.bf16 == 3) {               \
    BF16_LAUNCH_T(KK, 3);          \
  } else {                         \
    BF16_LAUNCH_T(KK, 1);          \
  }
#define BF16_UPS_T(TT)                                                                                                                              \
  if (shape == BF_A) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<2, 1, 4, 4, 1, 4, TT, 1, EPI_UPSAMPLE>), grid, dim3(256), 0, s, a);      \
  else if (shape == BF_B) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<2, 1, 1, 4, 1, 4, TT, 1, EPI_UPSAMPLE>), grid, dim3(256), 0, s, a); \
  else if (shape == BF_C) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<2, 1, 2, 2, 2, 4, TT, 1, EPI_UPSAMPLE>), grid, dim3(256), 0, s, a); \
  else if (shape == BF_K) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<2, 1, 4, 4, 1, 4, TT, 2, EPI_UPSAMPLE>), grid, dim3(512), 0, s, a); \
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_kernel<2, 1, 1, 1, 4, 4, TT, 1, EPI_UPSAMPLE>), grid, dim3(256), 0, s, a)
    if (p.epi == EPI_UPSAMPLE) {
      if (p.K != 2) return fail(MI355TTS_ERR_INVALID, "bf16 upsampler needs two taps");
      if (p.bf16 == 3) {
        BF16_UPS_T(3);
      } else {
        BF16_UPS_T(1);
      }
      return 0;
    }
#undef BF16_UPS_T
    switch (p.K) {
      case 3: BF16_LAUNCH(3); break;
      case 5: BF16_LAUNCH(5); break;
      case 7: BF16_LAUNCH(7); break;
      case 11: BF16_LAUNCH(11); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported conv kernel size %d in bf16 mode", p.K);
    }
#undef BF16_LAUNCH_T
#undef BF16_LAUNCH
    return rc;
  }
  if (p.epi == EPI_LINEAR) {
    switch (p.K) {
      case 1: rc = launch_conv_k<1, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 3: rc = launch_conv_k<3, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 5: rc = launch_conv_k<5, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 7: rc = launch_conv_k<7, EPI_LINEAR>(s, MB, shape, grid, a); break;
      case 11: rc = launch_conv_k<11, EPI_LINEAR>(s, MB, shape, grid, a); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported conv kernel size %d", p.K);
    }
  } else if (p.epi == EPI_GATE) {
    switch (p.K) {
      case 3: rc = launch_conv_k<3, EPI_GATE>(s, MB, shape, grid, a); break;
      case 5: rc = launch_conv_k<5, EPI_GATE>(s, MB, shape, grid, a); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported WaveNet kernel size %d", p.K);
    }
  } else if (p.epi == EPI_COUPLING) {
    if (p.K == 1) rc = launch_conv_k<1, EPI_COUPLING>(s, MB, shape, grid, a);
    else rc = fail(MI355TTS_ERR_INVALID, "coupling conv must be 1x1");
  } else {
    switch (p.K) {
      case 1: rc = launch_conv_k<1, EPI_UPSAMPLE>(s, MB, shape, grid, a); break;
      case 2: rc = launch_conv_k<2, EPI_UPSAMPLE>(s, MB, shape, grid, a); break;
      case 3: rc = launch_conv_k<3, EPI_UPSAMPLE>(s, MB, shape, grid, a); break;
      default: rc = fail(MI355TTS_ERR_INVALID, "unsupported upsample taps %d", p.K);
    }
  }
  return rc;
}

static int launch_conv(mi355tts_ctx* ctx, Worker* w, const DevConv& c, ConvArgs a, int epi, int B, int n_max, int cls,
                       hipStream_t stream = nullptr, int min_tiles = 1024, int host_len = -1, int precision = 0) {
  ConvPlan p;
  CHECK(plan_conv(c, a, epi, B, n_max, cls, min_tiles, host_len, &p, precision));
  return run_plan(ctx, w, p, stream);
}

// ---- grouped launch: the same-geometry convs of the MRF chains of a stage in ONE launch
template <int K0, int K1, int K2, int CI_C, int MB, int NB, int WN, int KS>
static void launch_group_inst(hipStream_t s, dim3 grid, const ConvGroupArgs& g) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_group_kernel<K0, K1, K2, CI_C, MB, NB, WN, KS, ConvCfg<K0>::HALO, ConvCfg<K1>::HALO, ConvCfg<K2>::HALO>),
                     grid, dim3(64 * WN * KS), 0, s, g);
}
template <int K0, int K1, int K2>
static int launch_group_k(hipStream_t s, int MB, int shape, dim3 grid, const ConvGroupArgs& g) {
  kn_add(KN_CONV_GROUP);
  // the tile shapes the batch-1 ... batch-8 ResBlock launches of the shipped vocoders use
  if (shape == TILE_TINY && MB == 2) launch_group_inst<K0, K1, K2, 64, 2, 1, 1, 8>(s, grid, g);
  else if (shape == TILE_TINY && MB == 1) launch_group_inst<K0, K1, K2, 64, 1, 1, 1, 8>(s, grid, g);
  else if (shape == TILE_SMALL && MB == 2) launch_group_inst<K0, K1, K2, 32, 2, 1, 2, 4>(s, grid, g);
  else if (shape == TILE_W128 && MB == 1) launch_group_inst<K0, K1, K2, 32, 1, 2, 2, 4>(s, grid, g);
  else if (shape == TILE_NB2 && MB == 2) launch_group_inst<K0, K1, K2, 16, 2, 2, 4, 2>(s, grid, g);
  else if (shape == TILE_M128)  // 16-channel chunks, one time-wave: <= 128 VGPRs, four 4-wave workgroups per CU
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_group_kernel<K0, K1, K2, 16, 1, 2, 1, 1, ConvCfg<K0>::HALO, ConvCfg<K1>::HALO, ConvCfg<K2>::HALO, 4>),
                       grid, dim3(256), 0, s, g);
  else {
    if (g_kn) g_kn[KN_CONV_GROUP].fetch_sub(1, std::memory_order_relaxed);
    return 1;
  }
  return 0;
}
// Returns 0 = launched as one group, 1 = not groupable (caller launches the members one by one), < 0 = error.
// a member of a grouped launch the continuous-stream tile (rb_conv.h) covers: a plain ResBlock conv — bias, optional residual
static bool rb_member_ok(const ConvArgs& a, int K) {
  const int halo = K == 11 ? RbCfg<11>::HALO : K == 7 ? RbCfg<7>::HALO : RbCfg<3>::HALO;
  return !a.x2 && !a.x3 && a.bias && a.alpha == 1.0f && !a.accum && a.out_act == ACT_NONE && a.split >= a.rows && !a.y2 &&
         a.rows % 128 == 0 && (K - 1) * a.dil + ((4 - a.pad % 4) % 4) <= halo;
}

// CUs of the device as the dispatch-order logic sees them (MI355TTS_GROUP_NCU overrides: tests reach the multi-round shapes at
// emulator sizes with it)
static int group_ncu(const mi355tts_ctx* ctx) {
  if (const char* e = std::getenv("MI355TTS_GROUP_NCU")) return std::atoi(e);
  return ctx->ncu;
}

// The same-geometry convs of the three MRF chains of a step (hifigan_forward.h plans them together).  At batch 1, members that
// plan_conv left on the small tiles (a stage too short for its 128-row threshold: the 256-channel stage of 'high') move to the
// 128-row tile when together they give every CU more than one workgroup: as ONE grouped launch of the continuous-stream tile
// with the dispatch laid out as a snake (run_group, group_snake_order) that step runs at 121 us where the 64 x 32 k-split tile
// needs 133 (one stream; 108 against 124 with two streams in flight — profiles/r04_rb_diag_snake_order.txt).  Decided on the
// plans, before the schedule is: the forked / one-by-one schedules then run the same tile arithmetic (same bits).  (f32 only: the
// same move in the split-bf16 mode — 128 x 64 tiles instead of the 8-wave k-split tile — measured 56.8 us against 54.4 us,
// profiles/r04_ab17.txt.)
static void promote_group_plans(mi355tts_ctx* ctx, Worker* w, ConvPlan* const* plans, int n) {
  // (option "rb_conv" = 0 / MI355TTS_NO_RB_CONV then run the chunked 128-row kernel in the plain order: same bits, slower)
  static const bool no_promote = [] { const char* e = std::getenv("MI355TTS_NO_GROUP_PROMOTE"); return e && std::atoi(e) != 0; }();
  if (n != 3 || no_promote || !w->o_group_promote) return;
  int total = 0, taps = 0;
  int tiles[3] = {0, 0, 0};  // by member in tap order 11, 7, 3
  for (int i = 0; i < 3; ++i) {
    const ConvPlan& p = *plans[i];
    if (p.empty || p.bf16 || p.pinned || p.epi != EPI_LINEAR || p.cls != KC_RESBLOCK || p.shape == TILE_M128 || p.grid.z != 1 ||
        p.n_max <= 0 || (p.a.x_ld % 4) || (p.K != 11 && p.K != 7 && p.K != 3) || !rb_member_ok(p.a, p.K))
      return;
    taps |= p.K == 11 ? 1 : p.K == 7 ? 2 : 4;
    tiles[p.K == 11 ? 0 : p.K == 7 ? 1 : 2] = ((p.n_max + 63) / 64) * (p.a.rows / 128);
    total += (((p.n_max + 63) / 64) * (p.a.rows / 128) + 7) & ~7;
  }
  const int ncu = group_ncu(ctx);
  if (taps != 7 || total <= ncu) return;
  if (total <= 4 * ncu) {
    // All resident at once: nothing is dealt dynamically, so the launch lasts as long as its busiest CU.  The big tile runs at
    // ~0.83 of peak against ~0.65-0.70 for the k-split tile it replaces (whose many small workgroups ARE dealt dynamically):
    // it wins while the snake keeps the busiest CU within ~1.3x of the mean (measured by utterance length, profiles/r04_ab18.txt; 1.09 at 624 frames of 'high': 121 us against 133; just above one
    // workgroup per CU — shorter utterances, narrower stages — the few second-round tiles double the busiest CUs' work).
    const char* env = std::getenv("MI355TTS_PROMOTE_MAX_IMBALANCE");  // (read per step, like MI355TTS_GROUP_NCU: tests move it)
    // Threshold 1.35 (1.25 until round 6, chosen on lone launches, where the two tiles are even between 1.2 and 1.3): with other
    // calls in flight — when another call's workgroups fill what the deal leaves idle — the big tile wins through that band too:
    // the headline's utterances of 560-580 and ~680 frames (imbalance 1.21-1.30) promoted: 293.5 -> 297.0 utterances/s A B A B,
    // the lone call's latency unchanged (profiles/r06_promote_ab.txt).  A geometry rule, not a load rule: the tile changes the
    // summation order, so it must not depend on who else is running.
    const double max_imbalance = env ? std::atof(env) : 1.35;
    ConvGroupArgs g;
    g.off[0] = 0;
    for (int m = 0; m < 3; ++m) g.off[m + 1] = g.off[m] + ((tiles[m] + 7) & ~7);
    group_snake_order(g, ncu, 4 * ncu);
    const double cost[3] = {11.0, 7.0, 3.0};
    if (group_order_imbalance(g, ncu, tiles, cost) > max_imbalance) return;
  }
  for (int i = 0; i < 3; ++i) {
    ConvPlan& p = *plans[i];
    p.shape = TILE_M128;
    p.MB = 1;
    p.a.rows_major = 0;
    p.grid = dim3((p.n_max + 63) / 64, p.a.rows / 128, 1);
  }
}

static int run_group(mi355tts_ctx* ctx, Worker* w, const ConvPlan* plans, int n, hipStream_t s) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_GROUP"); return e && std::atoi(e) != 0; }();
  if (off || n != 3) return 1;
  // members ordered by tap count, longest-running first
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (plans[ord[j]].K > plans[ord[i]].K) std::swap(ord[i], ord[j]);
  static const bool rb_off = [] { const char* e = std::getenv("MI355TTS_NO_RB_CONV"); return e && std::atoi(e) != 0; }();
  const int ncu = group_ncu(ctx);
  const ConvPlan& p0 = plans[ord[0]];
  for (int i = 0; i < 3; ++i) {
    const ConvPlan& p = plans[ord[i]];
    if (p.empty || p.epi != EPI_LINEAR || p.shape != p0.shape || p.MB != p0.MB || p.grid.z != p0.grid.z || p.bf16 != p0.bf16) return 1;
    if ((p.K - 1) * p.a.dil + ((4 - p.a.pad % 4) % 4) > (p.K == 3 ? 16 : p.K == 5 ? 28 : p.K == 7 ? 76 : p.K == 11 ? 56 : -1)) return 1;
    if (p.a.x_ld % 4) return 1;
  }
  ConvGroupArgs g;
  double flop = 0;
  int off_wg = 0;
  for (int i = 0; i < 3; ++i) {
    const ConvPlan& p = plans[ord[i]];
    g.c[i] = p.a;
    g.gx[i] = (int)p.grid.x;
    g.gy[i] = (int)p.grid.y;
    g.off[i] = off_wg;
    off_wg += ((int)(p.grid.x * p.grid.y) + 7) & ~7;
    flop += p.flop;
  }
  g.off[3] = off_wg;
  const dim3 grid(off_wg, 1, p0.grid.z);
  const int k0 = plans[ord[0]].K, k1 = plans[ord[1]].K, k2 = plans[ord[2]].K;
  const bool taps_ok = (k0 == 11 && k1 == 7 && k2 == 3) || (k0 == 7 && k1 == 5 && k2 == 3);
  g_kn = w->quiet ? nullptr : ctx->kn;
  if (p0.bf16) {
    if (!taps_ok) return 1;
    ProfScope ps(ctx, w, p0.cls, flop, s);
    g_last_sub = p0.a.rows;
    kn_add(KN_CONV_BF16_GROUP);
#define BF16_GROUP_T(KA, KB, KC, TT)                                                                                                                \
  if (p0.shape == BF_A)                                                                                                                            \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_group_kernel<KA, KB, KC, 1, 4, 4, 1, ConvCfg<KA>::HALO, ConvCfg<KB>::HALO, ConvCfg<KC>::HALO, TT>), \
                       grid, dim3(256), 0, s, g);                                                                                                  \
  else if (p0.shape == BF_B)                                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_group_kernel<KA, KB, KC, 1, 1, 4, 1, ConvCfg<KA>::HALO, ConvCfg<KB>::HALO, ConvCfg<KC>::HALO, TT>), \
                       grid, dim3(256), 0, s, g);                                                                                                  \
  else if (p0.shape == BF_C)                                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_group_kernel<KA, KB, KC, 1, 2, 2, 2, ConvCfg<KA>::HALO, ConvCfg<KB>::HALO, ConvCfg<KC>::HALO, TT>), \
                       grid, dim3(256), 0, s, g);                                                                                                  \
  else if (p0.shape == BF_K)                                                                                                                       \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_group_kernel<KA, KB, KC, 1, 4, 4, 1, ConvCfg<KA>::HALO, ConvCfg<KB>::HALO, ConvCfg<KC>::HALO, TT, 2>), \
                       grid, dim3(512), 0, s, g);                                                                                                  \
  else                                                                                                                                             \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_bf16_group_kernel<KA, KB, KC, 1, 1, 1, 4, ConvCfg<KA>::HALO, ConvCfg<KB>::HALO, ConvCfg<KC>::HALO, TT>), \
                       grid, dim3(256), 0, s, g)
#define BF16_GROUP(KA, KB, KC)      \
  if (p0.bf16 == 3) {               \
    BF16_GROUP_T(KA, KB, KC, 3);    \
  } else {                          \
    BF16_GROUP_T(KA, KB, KC, 1);    \
  }
    if (k0 == 11) { BF16_GROUP(11, 7, 3); }
    else { BF16_GROUP(7, 5, 3); }
#undef BF16_GROUP_T
#undef BF16_GROUP
    return 0;
  }
  const bool shape_ok = (p0.shape == TILE_TINY) || (p0.shape == TILE_SMALL && p0.MB == 2) || (p0.shape == TILE_W128 && p0.MB == 1) ||
                        (p0.shape == TILE_NB2 && p0.MB == 2) || p0.shape == TILE_M128;
  if (!shape_ok || !taps_ok) return 1;
  ProfScope ps(ctx, w, p0.cls, flop, s);
  g_last_sub = p0.a.rows;
  // The 128-row tile with the continuous matrix stream (rb_conv.h; same bits as the chunked tile) where the launch is
  // what it was written for: plain ResBlock convs (bias, optional residual), taps 11 / 7 / 3, dilation within its halos.
  if (p0.shape == TILE_M128 && k0 == 11 && !rb_off && w->o_rb_conv) {
    bool rb_ok = true;
    for (int i = 0; i < 3; ++i) rb_ok = rb_ok && rb_member_ok(g.c[i], i == 0 ? 11 : i == 1 ? 7 : 3);
    if (rb_ok) {
      static const bool no_snake = [] { const char* e = std::getenv("MI355TTS_NO_SNAKE"); return e && std::atoi(e) != 0; }();
      // 128-column tiles (NB = 4: half the weight-fragment bytes and staged halo per MFMA, three workgroups per CU) when the launch
      // still more than fills the chip with them.  Same chains per output element: same bits.
      // Measured A B A B (profiles/r06_nb4_ab.txt): the 128-channel stage's launch alone 240 -> 250 us (927 workgroups deal worse over
      // 768 slots than 1851 over 1024), 8 calls in flight +0.8-1.1 % utterances/s; the 256-channel stage (234 workgroups) -1.6 % — so
      // only launches with more 128-column tiles than the chip holds at once take them (the dispatcher then balances by itself), and
      // only while ANOTHER call holds a worker of this context: a lone call keeps the 64-column tiles (its launch has the chip to
      // itself and is 4 % faster on them).  The choice depends on the load, the result does not (same bits), like the dispatch order.
      // MI355TTS_RB_NB4_MIN_TILES = threshold whatever the load (tests, A/B runs), 0 = never.
      const char* nb4_env = std::getenv("MI355TTS_RB_NB4_MIN_TILES");  // (read per launch, like MI355TTS_GROUP_NCU: tests move it)
      const int nb4_min = nb4_env ? std::atoi(nb4_env) : (ctx->active_calls.load(std::memory_order_relaxed) > 1 ? 3 * ncu : 0);
      if (nb4_min > 0 && grid.z == 1) {
        int tiles4 = 0;
        for (int i = 0; i < 3; ++i) tiles4 += ((plans[ord[i]].n_max + 127) / 128) * g.gy[i];
        if (tiles4 >= nb4_min) {
          int o4 = 0;
          for (int i = 0; i < 3; ++i) {
            g.gx[i] = (plans[ord[i]].n_max + 127) / 128;
            g.off[i] = o4;
            o4 += (g.gx[i] * g.gy[i] + 7) & ~7;
          }
          g.off[3] = o4;
          const dim3 grid4(o4, 1, 1);
          if (!no_snake && w->o_snake) group_snake_order(g, ncu, 3 * ncu);
          kn_add(KN_RB_GROUP_NB4);
          hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_group_kernel<11, 7, 3, 4>), grid4, dim3(256), 0, s, g);
          return 0;
        }
      }
      if (grid.z == 1 && !no_snake && w->o_snake) group_snake_order(g, ncu, 4 * ncu);  // four of these workgroups fit a CU (32 KB, <= 128 VGPRs)
      kn_add(g.nseg ? KN_RB_GROUP_SNAKE : KN_RB_GROUP);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_group_kernel<11, 7, 3>), grid, dim3(256), 0, s, g);
      return 0;
    }
  }
  if (k0 == 11) return launch_group_k<11, 7, 3>(s, p0.MB, p0.shape, grid, g);
  return launch_group_k<7, 5, 3>(s, p0.MB, p0.shape, grid, g);
}

// Tile of the fused split-bf16 pair kernel (resblock_pair_bf16.h): time-waves x column blocks per wave, 256 columns in all
#ifndef P16_WN64
#define P16_WN64 4
#define P16_NB64 2
#endif
#ifndef P16_WN32
#define P16_WN32 4
#define P16_NB32 2
#endif
// Fused ResBlock1 step (conv1 -> lrelu -> conv2 -> + x) for the 32/64-channel stages.
struct PairPlan {
  PairArgs a;
  int K = 0, C = 0, NB = 1;
  dim3 grid;
  double flop = 0;
  bool ok = false;  // geometry covered by the fused kernel
  int bf16 = 0;     // 0 = f32 kernel (resblock_pair.h); 3 / 1 = split / plain bf16 kernel (resblock_pair_bf16.h)
  bool rb = false;  // f32: the 4-wave tile without a k-split (rb_pair.h) — launches with enough tiles (see plan_pair)
};
static void plan_pair(const DevConv& c1, const DevConv& c2, const float* x, float* y, long long bs, int ld, const int* len,
                      int len_mul, int dil, float alpha, int accum, int B, int Lmax, int host_len, PairPlan* out, int precision = 0) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_PAIR_FUSION"); return e && std::atoi(e) != 0; }();
  const int nb64 = 1;  // measured: 128-column tiles beat 256 at C = 64 (163 vs 197 us for the k = 11 pair)
  const int C = c1.Cout, K = c1.K;
  out->ok = false;
  const bool half = precision != MI355TTS_PRECISION_F32 && c1.w16 && c2.w16;
  if (half) {
    static const bool no_fused = [] { const char* e = std::getenv("MI355TTS_NO_BF16_PAIR"); return e && std::atoi(e) != 0; }();
    if (no_fused || c1.nslab16 != C / 16 || c2.nslab16 != C / 16 || c1.mtiles16 != C / 32 || c2.mtiles16 != C / 32) return;  // un-fused bf16 convs
  }
  if (off || (C != 32 && C != 64) || c1.Cin != C || c2.Cin != C || c2.Cout != C || c2.K != K || dil > PAIR_DMAX || dil < 1 ||
      (K != 3 && K != 7 && K != 11) || c1.noct != c2.noct || !c1.has_bias || !c2.has_bias || (ld % 4) || x == y || Lmax <= 0)
    return;
  PairArgs& a = out->a;
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
  out->K = K;
  out->C = C;
  out->NB = (C == 32 || half) ? 2 : nb64;
  out->bf16 = half ? (precision == MI355TTS_PRECISION_BF16 ? 1 : 3) : 0;
  a.w1h = c1.w16;
  a.w2h = c2.w16;
  a.nslab = c1.nslab16;
  const int T2 = 128 * out->NB - (K - 1);
  out->grid = dim3((Lmax + T2 - 1) / T2, 1, B);
  {
    // The 4-wave tile (three workgroups of 4 waves per CU) wins where a launch has many tiles ('high' at batch 1: 1300 per
    // member: -5 ... -9 % per launch); with a few dozen tiles per member (the 64-channel stage of 'medium': 42 at batch 1,
    // ~150 over config 4's ragged batch) the 8-wave k-split tile finishes a tile twice as fast and wins (97 vs 132 us).  The
    // count is the k = 11 member's, so the three members of a grouped launch always agree.
    const char* e = std::getenv("MI355TTS_RB_PAIR_MIN_TILES");  // (read per plan, like MI355TTS_M128_MIN_TILES: tests lower it)
    const long long min_tiles = e ? std::atoll(e) : 512LL;
    const int t2_ref = 128 * out->NB - 10;
    out->rb = !half && (long long)((Lmax + t2_ref - 1) / t2_ref) * B >= min_tiles;
  }
  out->flop = 2.0 * 2.0 * (double)C * C * K * (double)Lmax * B;
  out->ok = true;
}
static int run_pair(mi355tts_ctx* ctx, Worker* w, const PairPlan& p, hipStream_t s) {
  ProfScope ps(ctx, w, KC_RESBLOCK, p.flop, s);
  g_last_sub = p.C;
  const PairArgs& a = p.a;
  const dim3 grid = p.grid;
  g_kn = w->quiet ? nullptr : ctx->kn;
  kn_add(p.bf16 ? KN_PAIR_BF16 : (w->o_rb_pair && p.rb) ? KN_RB_PAIR : KN_PAIR);
  if (p.bf16) {
#define PAIR16_LAUNCH(KK, TT)                                                                                                                  \
  if (p.C == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_bf16_kernel<KK, 1, P16_WN32, P16_NB32, TT>), grid, dim3(64 * P16_WN32), 0, s, a);     \
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_bf16_kernel<KK, 2, P16_WN64, P16_NB64, TT>), grid, dim3(128 * P16_WN64), 0, s, a)
#define PAIR16_K(KK)          \
  if (p.bf16 == 3) {          \
    PAIR16_LAUNCH(KK, 3);     \
  } else {                    \
    PAIR16_LAUNCH(KK, 1);     \
  }
    if (p.K == 3) { PAIR16_K(3); }
    else if (p.K == 7) { PAIR16_K(7); }
    else { PAIR16_K(11); }
#undef PAIR16_K
#undef PAIR16_LAUNCH
    return 0;
  }
  if (w->o_rb_pair && p.rb) {  // the 4-wave tile without a k-split (rb_pair.h): same tiles and arguments
#define RBP_LAUNCH(KK)                                                                                               \
  if (p.C == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_pair_kernel<KK, 1>), grid, dim3(256), 0, s, a);               \
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_pair_kernel<KK, 2>), grid, dim3(256), 0, s, a)
    if (p.K == 3) { RBP_LAUNCH(3); }
    else if (p.K == 7) { RBP_LAUNCH(7); }
    else { RBP_LAUNCH(11); }
#undef RBP_LAUNCH
    return 0;
  }
#define PAIR_LAUNCH(KK, CB, NBB) hipLaunchKernelGGL(HIP_KERNEL_NAME(resblock_pair_kernel<KK, CB, NBB>), grid, dim3(512), 0, s, a)
#define PAIR_K(KK)                                  \
  if (p.C == 32) PAIR_LAUNCH(KK, 1, 2);             \
  else PAIR_LAUNCH(KK, 2, 1)
  if (p.K == 3) { PAIR_K(3); }
  else if (p.K == 7) { PAIR_K(7); }
  else { PAIR_K(11); }
#undef PAIR_K
#undef PAIR_LAUNCH
  return 0;
}
// The three chains' fused steps as ONE launch (k = 11, 7, 3 members).  0 = launched, 1 = not groupable.
static int run_pair_group(mi355tts_ctx* ctx, Worker* w, const PairPlan* plans, int n, hipStream_t s) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_GROUP"); return e && std::atoi(e) != 0; }();
  if (off || n != 3) return 1;
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (plans[ord[j]].K > plans[ord[i]].K) std::swap(ord[i], ord[j]);
  const PairPlan& p0 = plans[ord[0]];
  for (int i = 0; i < 3; ++i) {
    const PairPlan& p = plans[ord[i]];
    if (!p.ok || p.C != p0.C || p.NB != p0.NB || p.grid.z != p0.grid.z || p.bf16 != p0.bf16 || p.rb != p0.rb) return 1;
  }
  if (!(plans[ord[0]].K == 11 && plans[ord[1]].K == 7 && plans[ord[2]].K == 3)) return 1;
  if (!p0.bf16 && !((p0.C == 32 && p0.NB == 2) || (p0.C == 64 && p0.NB == 1))) return 1;
  PairGroupArgs g;
  double flop = 0;
  int off_wg = 0;
  for (int i = 0; i < 3; ++i) {
    const PairPlan& p = plans[ord[i]];
    g.p[i] = p.a;
    g.gx[i] = (int)p.grid.x;
    g.off[i] = off_wg;
    off_wg += ((int)p.grid.x + 7) & ~7;
    flop += p.flop;
  }
  g.off[3] = off_wg;
  const dim3 grid(off_wg, 1, p0.grid.z);
  ProfScope ps(ctx, w, KC_RESBLOCK, flop, s);
  g_last_sub = p0.C;
  g_kn = w->quiet ? nullptr : ctx->kn;
  kn_add(p0.bf16 ? KN_PAIR_BF16_GROUP : (w->o_rb_pair && p0.rb) ? KN_RB_PAIR_GROUP : KN_PAIR_GROUP);
  if (p0.bf16 == 3) {
    if (p0.C == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_bf16_group_kernel<11, 7, 3, 1, P16_WN32, P16_NB32, 3>), grid, dim3(64 * P16_WN32), 0, s, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_bf16_group_kernel<11, 7, 3, 2, P16_WN64, P16_NB64, 3>), grid, dim3(128 * P16_WN64), 0, s, g);
    return 0;
  }
  if (p0.bf16 == 1) {
    if (p0.C == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_bf16_group_kernel<11, 7, 3, 1, P16_WN32, P16_NB32, 1>), grid, dim3(64 * P16_WN32), 0, s, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_bf16_group_kernel<11, 7, 3, 2, P16_WN64, P16_NB64, 1>), grid, dim3(128 * P16_WN64), 0, s, g);
    return 0;
  }
  if (w->o_rb_pair && p0.rb) {
    if (p0.C == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_pair_group_kernel<11, 7, 3, 1>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(rb_pair_group_kernel<11, 7, 3, 2>), grid, dim3(256), 0, s, g);
    return 0;
  }
  if (p0.C == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_group_kernel<11, 7, 3, 1, 2>), grid, dim3(512), 0, s, g);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(pair_group_kernel<11, 7, 3, 2, 1>), grid, dim3(512), 0, s, g);
  return 0;
}

// ---- the one-launch MRF stage of the narrow HiFi-GAN stages (mrf_small.h)
// x -> y = rb_3(x) + rb_7(x), y2 = rb_11(x) ([B][C][ld] planes; the consumer forms (y + y2) / 3 on load);
// len/len_mul as everywhere (row b is len[b]*len_mul long)
static int run_mrf_small(mi355tts_ctx* ctx, Worker* w, const MrfStage& ms, const float* arena, const float* x, float* y, float* y2,
                         long long bs, int ld, const int* len, int len_mul, int B, int Lmax, int host_len, hipStream_t s) {
  if (!ms.ok || (ld % 4) || x == y || x == y2 || y == y2 || Lmax <= 0)
    return fail(MI355TTS_ERR_INVALID, "internal: MRF stage not covered by the fused kernel");
  MrfArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x;
  a.y = y;
  a.y2 = y2;
  a.bs = bs;
  a.ld = ld;
  a.len = (B == 1 && host_len >= 0) ? nullptr : len;
  a.len_mul = len_mul;
  a.len_const = host_len * len_mul;
  a.w = arena + ms.w_off;
  a.bias = arena + ms.b_off;
  a.tab = reinterpret_cast<const int*>(arena + ms.t_off);
  a.nsteps = ms.nsteps;
  a.slope = 0.1f;
  // 256-column tiles: 2 x C x 400 floats of LDS (51 KB at C = 16, 26 KB at C = 8) -> three workgroups per CU.  (512-column
  // tiles measured no better at either width: tools/probe/mrf_bench.hip, profiles/NOTES.md.)
  constexpr int T = 256;
  const dim3 grid(2 * ((Lmax + T - 1) / T), 1, B);  // two workgroups per tile
  ProfScope ps(ctx, w, KC_MRF_NARROW, 2.0 * ms.mac_per_col * (double)Lmax * B, s);
  static const bool mrf8_off = [] { const char* e = std::getenv("MI355TTS_NO_MRF8"); return e && std::atoi(e) != 0; }();
  if (ms.C == 16) {
    kn_hit(ctx, KN_MRF_SMALL);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mrf_small_kernel<16, T, 4, 3, 7, 11>), grid, dim3(256), 0, s, a);
  } else if (!mrf8_off) {
    // 8 channels: the 4x4x1 16-block MFMA (no padding rows), its own fragment packing; two waves per tile
    a.w = arena + ms.w8_off;
    a.tab = reinterpret_cast<const int*>(arena + ms.t8_off);
    kn_hit(ctx, KN_MRF8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mrf8_kernel<T, 3, 7, 11>), grid, dim3(128), 0, s, a);
  } else {
    kn_hit(ctx, KN_MRF_SMALL);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mrf_small_kernel<8, T, 4, 3, 7, 11>), grid, dim3(256), 0, s, a);
  }
  return 0;
}

template <int KK, int JJ>
static void gate16_launch(bool wide, dim3 grid, hipStream_t s, const Gate16Args& g) {
  if constexpr (JJ == 6) {
    if (wide) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(gate16_kernel<KK, JJ, 2>), grid, dim3(512), 0, s, g);
      return;
    }
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(gate16_kernel<KK, JJ, 1>), grid, dim3(512), 0, s, g);
}
// ---- the WaveNet gate conv on 16-row tiles (gate16.h).  Returns 1 when this conv / launch is not one the kernel takes
// (the caller then launches the 32-row tile), 0 when launched, < 0 on error.  `a` is the ConvArgs of the same launch.
static int run_gate16(mi355tts_ctx* ctx, Worker* w, const DevConv& c, const ConvArgs& a, int B, int n_max, int cls, hipStream_t s) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_GATE16"); return e && std::atoi(e) != 0; }();
  // more 32-row tiles than this and the launch fills the chip either way (measured: profiles/NOTES.md)
  static const long long max_tiles = [] { const char* e = std::getenv("MI355TTS_GATE16_MAX_TILES"); return e ? std::atoll(e) : 1LL << 40; }();
  if (off || !w->o_gate16 || !c.g16_J || n_max <= 0) return 1;
  const int PA = (a.pad + 3) & ~3;
  if ((PA - a.pad) + 31 + (c.K - 1) * a.dil >= GATE16_XW || a.x_ld % 4 || a.in_mul != a.out_mul || a.in_len != a.out_len) return 1;
  const int gx = (n_max + 31) / 32, gy = (a.half + 7) / 8;
  if ((long long)gx * ((a.half + 15) / 16) * B > max_tiles) return 1;
  Gate16Args g;
  std::memset(&g, 0, sizeof(g));
  g.x = a.x; g.x_bs = a.x_bs; g.x_ld = a.x_ld;
  g.len = a.in_len; g.len_mul = a.in_mul; g.len_const = a.in_const;
  g.w = c.g16_w; g.bias = c.g16_b; g.Cin = c.Cin; g.half = a.half; g.dil = a.dil; g.pad = a.pad;
  g.y = a.y; g.y_bs = a.y_bs; g.y_ld = a.y_ld;
  g.cond = a.cond; g.cond_bs = a.cond_bs;
  ProfScope ps(ctx, w, cls, 2.0 * (double)c.Cout * c.Cin * c.K * (double)n_max * B, s);
  // wide passes (padded batches, coalesced passes): two row tiles per workgroup from one staged tile — same bits (gate16.h)
  const long long wide_min = w->o_gate16_wide;
  const bool wide = wide_min > 0 && c.g16_J == 6 && (gy % 2) == 0 && (long long)gx * gy * B >= wide_min;  // (the released voices' width)
  const dim3 grid(gx, wide ? gy / 2 : gy, B);
#define GATE16_LAUNCH(KK, JJ) gate16_launch<KK, JJ>(wide, grid, s, g)
#define GATE16_J(KK)                      \
  switch (c.g16_J) {                      \
    case 1: GATE16_LAUNCH(KK, 1); break;  \
    case 2: GATE16_LAUNCH(KK, 2); break;  \
    case 3: GATE16_LAUNCH(KK, 3); break;  \
    case 4: GATE16_LAUNCH(KK, 4); break;  \
    case 6: GATE16_LAUNCH(KK, 6); break;  \
    case 8: GATE16_LAUNCH(KK, 8); break;  \
    default: return 1;                    \
  }
  if (c.K == 3) {
    GATE16_J(3)
  } else if (c.K == 5) {
    GATE16_J(5)
  } else {
    return 1;
  }
#undef GATE16_J
#undef GATE16_LAUNCH
  kn_hit(ctx, wide ? KN_GATE16_WIDE : KN_GATE16);
  return 0;
}

// ---- a plain encoder conv on 16-row tiles with the whole input tile staged once (lin16_kernel, gate16.h).  Returns 1 when
// this conv / launch is not one the kernel takes (the caller then launches the generic tile), 0 when launched.
// `ln` != nullptr: the conv's input is LayerNorm'ed first (lin16_kernel<..., true>): gamma / beta, ReLU behind the norm or not,
// and where the normalised tensor is also stored (nullptr = nowhere)
struct Lin16Ln {
  const float* gamma;
  const float* beta;
  int relu;
  float* out;
};
static int run_lin16(mi355tts_ctx* ctx, Worker* w, const DevConv& c, const ConvArgs& a, const float* arena, int B, int n_max, int cls,
                     int host_len, bool solo_tiles = false, const Lin16Ln* ln = nullptr) {
  static const bool off = [] { const char* e = std::getenv("MI355TTS_NO_LIN16"); return e && std::atoi(e) != 0; }();
  // more tiles than this and the chunked 32-row tile fills the chip (longer rows, bigger batches)
  static const long long max_tiles = [] { const char* e = std::getenv("MI355TTS_LIN16_MAX_TILES"); return e ? std::atoll(e) : 1LL << 40; }();
  static const bool no_k1 = [] { const char* e = std::getenv("MI355TTS_LIN16_NO_K1"); return e && std::atoi(e) != 0; }();
  if (off || !w->o_glow_fuse || !c.l16_J || n_max <= 0 || (c.K == 1 && no_k1)) return 1;
  const int PA = (a.pad + 3) & ~3;
  if ((PA - a.pad) + (c.K - 1) * a.dil > MI355TTS_G16_HALO || a.x_ld % 4 || a.in_mul != a.out_mul || a.in_len != a.out_len) return 1;
  if (a.x2 || a.alpha != 1.0f || a.accum || a.in_slope != 1.0f || (a.out_act != ACT_NONE && a.out_act != ACT_RELU)) return 1;
  const bool two = a.split < c.rows;  // second output for the rows >= split
  if (two && (!a.y2 || a.split < 0 || a.split % 16)) return 1;
  const int nblk = c.l16_J >= 16 ? 1 : 2;
  const int TC = 16 * nblk;
  const int gx = (n_max + TC - 1) / TC, gy = (c.rows + 15) / 16;
  if ((long long)gx * gy * B > max_tiles) return 1;
  // 1 x 1 convs: not in big padded batches — in config 4's batch of 8 the 840 16-row tiles of a res_skip conv
  // measured 3 % faster alone and 1.2 % slower with 8 calls in flight than the 64-row tile (profiles/NOTES.md)
  // (explicit batches only: a batch-1 call always takes this form, and so does a coalesced pass, whose rows must equal
  // their batch-1 results whatever their lengths)
  // Round 5: such passes take FOUR row tiles per workgroup from one staged tile (lin16_kernel<..., RTW = 4>: same bits as the
  // 16-row launch, a quarter of the staging) — option "gate16_wide" (the pass size from which; 0 = round 4's rule)
  const bool wide = c.K == 1 && c.l16_J == 6 && !ln && nblk == 2 && (gy % 4) == 0 && w->o_gate16_wide > 0 &&
                    (long long)gx * gy * B >= w->o_gate16_wide;
  if (!wide && c.K == 1 && B > 1 && !solo_tiles && (long long)gx * gy * B > 512) return 1;
  Lin16Args g;
  std::memset(&g, 0, sizeof(g));
  g.x = a.x; g.x_bs = a.x_bs; g.x_ld = a.x_ld;
  if (B == 1 && host_len >= 0) { g.len = nullptr; g.len_const = host_len * a.in_mul; } else { g.len = a.in_len; g.len_const = a.in_const; }
  g.len_mul = a.in_mul;
  g.w = arena + c.l16_w_off; g.bias = arena + c.l16_b_off; g.Cin = c.Cin; g.rows = c.rows; g.dil = a.dil; g.pad = a.pad;
  g.y = a.y; g.y_bs = a.y_bs; g.y_ld = a.y_ld; g.res = a.res; g.relu = a.out_act == ACT_RELU;
  g.split = two ? a.split : (1 << 30); g.y2 = a.y2; g.y2_bs = a.y2_bs; g.y2_ld = a.y2_ld; g.accum2 = a.accum2;
  if (ln) {
    const bool ln_shape = (c.K == 1 && c.l16_J == 6) || (c.K == 5 && c.l16_J == 6) || (c.K == 3 && c.l16_J == 8);
    if (!ln_shape) return 1;
    g.ln_gamma = ln->gamma; g.ln_beta = ln->beta; g.ln_eps = 1e-4f; g.ln_relu = ln->relu; g.ln_out = ln->out;
  }
  ProfScope ps(ctx, w, cls, 2.0 * (double)c.Cout * c.Cin * c.K * (double)n_max * B);
  const dim3 grid(gx, wide ? gy / 4 : gy, B);
  hipStream_t s = w->stream;
  if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<1, 6, 2, false, 4>), grid, dim3(512), 0, s, g);
  else if (ln && c.K == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<1, 6, 2, true>), grid, dim3(512), 0, s, g);
  else if (ln && c.K == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<5, 6, 2, true>), grid, dim3(512), 0, s, g);
  else if (ln && c.K == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<3, 8, 2, true>), grid, dim3(512), 0, s, g);
  else if (c.K == 3 && c.l16_J == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<3, 6, 2>), grid, dim3(512), 0, s, g);
  else if (c.K == 3 && c.l16_J == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<3, 8, 2>), grid, dim3(512), 0, s, g);
  else if (c.K == 3 && c.l16_J == 24) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<3, 24, 1>), grid, dim3(512), 0, s, g);
  else if (c.K == 5 && c.l16_J == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<5, 6, 2>), grid, dim3(512), 0, s, g);
  else if (c.K == 1 && c.l16_J == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(lin16_kernel<1, 6, 2>), grid, dim3(512), 0, s, g);
  else return 1;
  kn_hit(ctx, wide ? KN_LIN16_WIDE : ln ? KN_LIN16_LN : KN_LIN16);
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
