// Host-side re-layout of logical conv weights into the MFMA A-fragment stream
// conv_mfma_kernel reads (one float4 per lane per (m-tile, 8-channel octet, tap)).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mi355tts {

struct PackedConv {
  std::vector<float> w;     // [mtiles][noct][K][64 lanes][4]
  std::vector<float> bias;  // [mtiles*32] in virtual-row order (empty = no bias)
  int mtiles = 0;
  int noct = 0;
  int K = 0;
  int rows = 0;  // valid virtual rows
  int Cin = 0;
};

// `rowmap(v)` gives the logical output channel of virtual row v, or -1 for a
// padding row.  `wget(co, ci, k)` reads the logical weight.
template <typename RowMap, typename WGet, typename BGet>
inline PackedConv pack_conv(int vrows, int m_align_tiles, int Cin, int K, RowMap rowmap, WGet wget, BGet bget, bool has_bias,
                            int oct_align = 4) {
  PackedConv p;
  int mt = (vrows + 31) / 32;
  mt = ((mt + m_align_tiles - 1) / m_align_tiles) * m_align_tiles;
  p.mtiles = mt;
  p.noct = (Cin + 7) / 8;
  p.noct = ((p.noct + oct_align - 1) / oct_align) * oct_align;  // whole staged chunks only (zero weights)
  p.K = K;
  p.rows = vrows;
  p.Cin = Cin;
  p.w.assign((size_t)mt * p.noct * K * 256, 0.f);
  if (has_bias) p.bias.assign((size_t)mt * 32, 0.f);
  for (int m = 0; m < mt; ++m) {
    for (int lane = 0; lane < 64; ++lane) {
      const int v = m * 32 + (lane & 31);
      const int co = (v < vrows) ? rowmap(v) : -1;
      if (co < 0) continue;
      if (has_bias && lane < 32) p.bias[v] = bget(co);
      for (int oct = 0; oct < p.noct; ++oct) {
        for (int k = 0; k < K; ++k) {
          float* dst = &p.w[((((size_t)m * p.noct + oct) * K + k) * 64 + lane) * 4];
          for (int j = 0; j < 4; ++j) {
            const int ci = oct * 8 + 2 * j + (lane >> 5);
            dst[j] = (ci < Cin) ? wget(co, ci, k) : 0.f;
          }
        }
      }
    }
  }
  return p;
}

// ---- split-bf16 fragments for conv_bf16.h -------------------------------------------------------
inline uint16_t bf16_rne(float v) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_float(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

struct PackedConv16 {
  std::vector<uint16_t> w;  // [mtiles][nslab][K][plane hi/lo][64 lanes][8]
  int mtiles = 0, nslab = 0, K = 0;
};

// lane l of (m-tile, 16-channel slab, tap) holds weights of row 32*mt + (l & 31), channels
// 16*slab + 8*(l >> 5) .. +7 — the A operand of one v_mfma_f32_32x32x16_bf16 — once as bf16(w)
// ("hi") and once as bf16(w - hi) ("lo").  Rows padded to whole `m_align` tiles, channels to whole
// 32-channel chunks (two slabs), with zeros.
template <typename WGet>
inline PackedConv16 pack_conv_bf16(int rows, int m_align, int Cin, int K, WGet wget) {
  PackedConv16 p;
  int mt = (rows + 31) / 32;
  mt = ((mt + m_align - 1) / m_align) * m_align;
  p.mtiles = mt;
  p.nslab = 2 * ((Cin + 31) / 32);
  p.K = K;
  p.w.assign((size_t)mt * p.nslab * K * 2 * 64 * 8, 0);
  for (int m = 0; m < mt; ++m)
    for (int slab = 0; slab < p.nslab; ++slab)
      for (int k = 0; k < K; ++k)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = m * 32 + (lane & 31);
          if (co >= rows) continue;
          uint16_t* hi = &p.w[(((((size_t)m * p.nslab + slab) * K + k) * 2 + 0) * 64 + lane) * 8];
          uint16_t* lo = hi + 64 * 8;
          for (int i = 0; i < 8; ++i) {
            const int ci = slab * 16 + 8 * (lane >> 5) + i;
            const float v = ci < Cin ? wget(co, ci, k) : 0.f;
            hi[i] = bf16_rne(v);
            lo[i] = bf16_rne(v - bf16_to_float(hi[i]));
          }
        }
  return p;
}

// ---- fp16 fragments for conv_f16.h (the native 16-bit mode) ---------------------------------------
inline uint16_t f16_rne(float v) {
  const _Float16 h = (_Float16)v;  // round to nearest even; overflow -> inf (weights of a sane checkpoint are far from 65504)
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
inline float f16_to_float(uint16_t u) {
  _Float16 h;
  std::memcpy(&h, &u, 2);
  return (float)h;
}
struct PackedConvH {
  std::vector<uint16_t> w;  // [mtiles][nslab][K][64 lanes][8]
  std::vector<float> bias;  // [mtiles * 32] f32, virtual-row order
  int mtiles = 0, nslab = 0, K = 0, rows = 0;
};
// lane l of (m-tile, 16-channel slab, tap) holds the weights of virtual row 32 mt + (l & 31), channels 16 slab + 8 (l >> 5) .. + 7:
// the A operand of one v_mfma_f32_32x32x16_f16.  Rows padded to whole `m_align` tiles, channels to whole `ch_align`-channel
// chunks, with zeros.  `wget(v, ci, k)` reads the logical weight of VIRTUAL row v, `bget(v)` its bias.
template <typename WGet, typename BGet>
inline PackedConvH pack_conv_f16(int rows, int m_align, int Cin, int K, int ch_align, WGet wget, BGet bget, bool has_bias) {
  PackedConvH p;
  int mt = (rows + 31) / 32;
  mt = ((mt + m_align - 1) / m_align) * m_align;
  p.mtiles = mt;
  p.nslab = (ch_align / 16) * ((Cin + ch_align - 1) / ch_align);
  p.K = K;
  p.rows = rows;
  p.w.assign((size_t)mt * p.nslab * K * 64 * 8, 0);
  p.bias.assign((size_t)mt * 32, 0.f);
  for (int m = 0; m < mt; ++m)
    for (int lane = 0; lane < 64; ++lane) {
      const int v = m * 32 + (lane & 31);
      if (v >= rows) continue;
      if (has_bias && lane < 32) p.bias[v] = bget(v);
      for (int slab = 0; slab < p.nslab; ++slab)
        for (int k = 0; k < K; ++k) {
          uint16_t* dst = &p.w[((((size_t)m * p.nslab + slab) * K + k) * 64 + lane) * 8];
          for (int i = 0; i < 8; ++i) {
            const int ci = slab * 16 + 8 * (lane >> 5) + i;
            dst[i] = ci < Cin ? f16_rne(wget(v, ci, k)) : 0;
          }
        }
    }
  return p;
}

// ---- A fragments for mrf_small.h (v_mfma_f32_16x16x4_f32) ---------------------------------------
// One conv w[C][C][K] with C <= 16 as [tap][C/4][64 lanes]: lane l of (tap, channel quad q) holds
// W[co = l & 15][ci = 4q + (l >> 4)][tap] — the A operand A[i = l & 15][k = l >> 4] of one MFMA — and
// zero for the padding rows co >= C.  No channel padding: the K-dim is exactly C x taps deep.
template <typename WGet>
inline std::vector<float> pack_mrf_conv(int C, int K, WGet wget) {
  const int CQ = C / 4;
  std::vector<float> p((size_t)K * CQ * 64, 0.f);
  for (int k = 0; k < K; ++k)
    for (int q = 0; q < CQ; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = lane & 15, ci = 4 * q + (lane >> 4);
        if (co < C) p[((size_t)k * CQ + q) * 64 + lane] = wget(co, ci, k);
      }
  return p;
}

// A fragments for mrf8_kernel (v_mfma_f32_4x4x1_16B_f32 with CBSZ = 4): one conv w[8][8][K] as [tap][64 lanes], lane = 8 ci + co —
// ABID = 2 ci (+ 1) then broadcasts W[0..3 (4..7)][ci][tap] to all sixteen 4-column blocks.
template <typename WGet>
inline std::vector<float> pack_mrf8_conv(int K, WGet wget) {
  std::vector<float> p((size_t)K * 64, 0.f);
  for (int k = 0; k < K; ++k)
    for (int ci = 0; ci < 8; ++ci)
      for (int co = 0; co < 8; ++co) p[(size_t)k * 64 + 8 * ci + co] = wget(co, ci, k);
  return p;
}

// A fragments and biases for gate16_kernel (gate16.h): the WaveNet gate conv w[2*half][Cin][K] on 16-row tiles.  Row
// tile p = gate channels 8p .. 8p+7: tile rows 0-7 are their tanh rows (w row c), rows 8-15 their sigmoid rows (w row
// half + c).  k-group g of 8 takes the 4-channel groups g + 8 j (j < J); lane (m = l & 15, kq = l >> 4) of its step
// (j, tap) holds W[row m][4 (g + 8 j) + kq][tap] — the A operand of v_mfma_f32_16x16x4_f32.  Zero past half / Cin.
struct PackedGate16 {
  std::vector<float> w;     // [ptiles][8][J][K][64]
  std::vector<float> bias;  // [ptiles][16]
  int ptiles = 0, J = 0;
};
template <typename WGet, typename BGet>
inline PackedGate16 pack_gate16(int half, int Cin, int K, WGet wget, BGet bget, bool has_bias) {
  PackedGate16 p;
  p.ptiles = (half + 7) / 8;
  p.J = (Cin + 31) / 32;
  p.w.assign((size_t)p.ptiles * 8 * p.J * K * 64, 0.f);
  p.bias.assign((size_t)p.ptiles * 16, 0.f);
  for (int t = 0; t < p.ptiles; ++t) {
    for (int m = 0; m < 16; ++m) {
      const int c = 8 * t + (m & 7);
      if (c >= half) continue;
      const int co = (m >> 3) * half + c;
      if (has_bias) p.bias[(size_t)t * 16 + m] = bget(co);
      for (int g = 0; g < 8; ++g)
        for (int j = 0; j < p.J; ++j)
          for (int kq = 0; kq < 4; ++kq) {
            const int ci = 4 * (g + 8 * j) + kq;
            if (ci >= Cin) continue;
            for (int k = 0; k < K; ++k)
              p.w[((((size_t)t * 8 + g) * p.J + j) * K + k) * 64 + 16 * kq + m] = wget(co, ci, k);
          }
    }
  }
  return p;
}

// A fragments and biases for lin16_kernel (gate16.h): a plain conv w[rows][Cin][K] on 16-row tiles.  k-group g of 8 takes the
// 4-channel groups g + 8 j (j < J); lane (m = l & 15, kq = l >> 4) of its step (j, tap) holds W[16 p + m][4 (g + 8 j) + kq][tap].
template <typename WGet, typename BGet>
inline PackedGate16 pack_lin16(int rows, int Cin, int K, WGet wget, BGet bget, bool has_bias) {
  PackedGate16 p;
  p.ptiles = (rows + 15) / 16;
  p.J = (Cin + 31) / 32;
  p.w.assign((size_t)p.ptiles * 8 * p.J * K * 64, 0.f);
  p.bias.assign((size_t)p.ptiles * 16, 0.f);
  for (int t = 0; t < p.ptiles; ++t)
    for (int m = 0; m < 16; ++m) {
      const int co = 16 * t + m;
      if (co >= rows) continue;
      if (has_bias) p.bias[(size_t)t * 16 + m] = bget(co);
      for (int g = 0; g < 8; ++g)
        for (int j = 0; j < p.J; ++j)
          for (int kq = 0; kq < 4; ++kq) {
            const int ci = 4 * (g + 8 * j) + kq;
            if (ci >= Cin) continue;
            for (int k = 0; k < K; ++k) p.w[((((size_t)t * 8 + g) * p.J + j) * K + k) * 64 + 16 * kq + m] = wget(co, ci, k);
          }
    }
  return p;
}

// A fragments for the column-owner launches (coltile.h): a dense W[rows][K] on 16-row tiles of v_mfma_f32_16x16x4_f32,
// [row tile][group of 4 k-steps][64 lanes][4]: element j of lane (m = l & 15, kq = l >> 4) is W[16 rt + m][16 q + 4 j + kq].
// Rows and K zero-padded to multiples of 16; the bias to whole row tiles.
struct PackedCol16 {
  std::vector<float> w, bias;
  int RT = 0, KQ4 = 0;
};
template <typename WGet, typename BGet>
inline PackedCol16 pack_col16(int rows, int K, WGet wget, BGet bget, bool has_bias) {
  PackedCol16 p;
  p.RT = (rows + 15) / 16;
  p.KQ4 = (K + 15) / 16;
  p.w.assign((size_t)p.RT * p.KQ4 * 256, 0.f);
  p.bias.assign((size_t)p.RT * 16, 0.f);
  for (int rt = 0; rt < p.RT; ++rt)
    for (int lane = 0; lane < 64; ++lane) {
      const int row = rt * 16 + (lane & 15);
      if (row >= rows) continue;
      if (has_bias && lane < 16) p.bias[row] = bget(row);
      for (int q = 0; q < p.KQ4; ++q)
        for (int j = 0; j < 4; ++j) {
          const int k = 16 * q + 4 * j + (lane >> 4);
          if (k < K) p.w[(((size_t)rt * p.KQ4 + q) * 64 + lane) * 4 + j] = wget(row, k);
        }
    }
  return p;
}

}  // namespace mi355tts
