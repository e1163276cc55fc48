// Host-side re-layout of logical conv weights into the MFMA A-fragment stream
// conv_mfma_kernel reads (one float4 per lane per (m-tile, 8-channel octet, tap)).
#pragma once
#include <cstddef>
#include <cstdint>
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

}  // namespace mi355tts
