// LDS-DMA implicit-GEMM kernel (conv_v2_kernel.h), split-plane "f16x3" instantiations: the layers of RAFT the halo-tile kernel does
// not serve (strided / 7x7 / 1x1 convolutions, sources that are not multiples of 64 channels).  See conv_gemm_v3s.hip.
#include "conv_v2_kernel.h"

namespace pp {

int conv_v2s_dispatch(const ConvParams& p, int cfg, hipStream_t stream) {
  if (p.kchunks % 8 != 0 || p.M >= (1ll << 31) || (long long)p.N * p.H * p.W >= (1ll << 31)) return -1000;
  bool uni = p.ktable_uniform != 0 && p.pad_mode == 0 && (long long)p.cout_pad * p.kchunks * 16 < (1ll << 31);
  for (int i = 0; i < p.nsrc; ++i) uni = uni && (long long)p.N * p.H * p.W * p.src[i].cstride * 2 < (1ll << 31);
  if (cfg >= 100) { uni = false; cfg -= 100; }
  if (cfg == 0) {   // the fp16 dispatch's choices (conv_v2_dispatch)
    // tri-product 1x1 layers with >= 256 couts and K >= 256 (RAFT's convc1: 324 correlation taps -> 256): one 256 x 256 tile fetches a
    // pixel row once for all its couts -- 428 vs 480 us per 35-pair launch, bit-identical (profiles/r6_convc1_tile.txt); short-K layers
    // (the encoders' 128 -> 256 projection) stay on the 128 x 128 tile
    if (p.split == 2 && p.cout_g >= 256 && p.cout_g < 512 && p.kchunks >= 64) cfg = 18;
    else if (p.cout_g >= 512) cfg = 13;
    else if (p.cout_g > 64) cfg = 12;
    else if (p.cout_g > 32) cfg = 22;
    else if (p.cout_g > 16) cfg = 32;
    else cfg = 42;
  }
  if (p.split == 2) {      // tri-product K format: 64-wide K steps only
    switch (cfg) {
      case 12: return launch_v2<128, 128, 64, 2, 2, 2, 0, true, true>(p, uni, stream);
      case 13: return launch_v2<256, 128, 64, 4, 2, 3, 0, true, true>(p, uni, stream);
      case 22: return launch_v2<256, 64, 64, 4, 1, 2, 0, true, true>(p, uni, stream);
      case 18: return launch_v2<256, 256, 64, 4, 2, 2, 0, true, true>(p, uni, stream);   // 8 waves of 64 x 128: every pixel row fetched once for 256 couts
      default: return -1000;
    }
  }
  switch (cfg) {
    case 12: return launch_v2<128, 128, 64, 2, 2, 2, 0, true>(p, uni, stream);
    case 13: return launch_v2<256, 128, 64, 4, 2, 3, 0, true>(p, uni, stream);
    case 22: return launch_v2<256, 64, 64, 4, 1, 2, 0, true>(p, uni, stream);
    case 32: return launch_v2<256, 32, 32, 4, 1, 2, 0, true>(p, uni, stream);
    case 42: return launch_v2<256, 16, 32, 4, 1, 2, 0, true>(p, uni, stream);
    default: return -1000;
  }
}

}  // namespace pp
