// Halo-tile convolution kernel (conv_halo.h), split-plane "f16x3" instantiations: RAFT at the reference's precision class on the
// fp16 matrix cores.  The reference keeps RAFT in fp32 even under --fp16 (inference_propainter.py:311); here every RAFT
// activation is a pair of fp16 planes (hi = fp16(v), lo = fp16(v - hi)), every weight a pair (W_hi, W_lo), and every product runs
// as hi*W_hi + lo*W_hi + hi*W_lo with fp32 accumulation.  TRI-PRODUCT K format (propainter_amd/conv.py, tri_ktable; pp_conv_args_t.split
// == 2): a K block is 32 channels of both planes, a tap step reads the four fragment sets (A_hi, A_lo, W_hi, W_lo) once and issues the
// three products -- the LDS-DMA gather, the swizzles and the LDS image are the fp16 kernel's own (a 128-byte patch row = [hi | lo]).
// The epilogue reads its operands as hi + lo and writes two planes (conv_epilogue.h, SPLIT).
#include "conv_halo.h"
#include <stdlib.h>

namespace pp {

template <int KH, int KW>
static int launch_v3s(const ConvParams& p, int bn, bool shared_w, hipStream_t stream) {
  constexpr int TH = KW == 1 ? 16 : 8, TW = KW == 1 ? 8 : 16;
  // tiles of at most 64 couts: optional wave-private weight stages, no block barrier inside a channel block (conv_halo.h, PRIVB;
  // PP_HALO_PRIVATE_WEIGHTS=1) -- measured neutral (conv_gemm_v3.hip), the shared-stage form ships
  if (bn == 16) return shared_w ? launch_v3<TH, TW, KH, KW, 16, false, 0, 64, true>(p, stream) : launch_v3<TH, TW, KH, KW, 16, false, 0, 64, true, true>(p, stream);
  if (bn == 64) return shared_w ? launch_v3<TH, TW, KH, KW, 64, false, 0, 64, true>(p, stream) : launch_v3<TH, TW, KH, KW, 64, false, 0, 64, true, true>(p, stream);
  return launch_v3<TH, TW, KH, KW, 128, false, 0, 64, true>(p, stream);
}

// Same family and tile choice as conv_v3_dispatch (conv_gemm_v3.hip).  cfg: 0 = auto, 71 = 128-cout tiles, 72 = 64-cout tiles.
int conv_v3s_dispatch(const ConvParams& p, int cfg, hipStream_t stream) {
  const int kh = p.tap_h, kw = p.tap_w;
  if (kh <= 0 || kw <= 0) return -1000;
  if (p.groups != 1 || p.sh != 1 || p.sw != 1 || p.pad_mode != 0 || p.OH != p.H || p.OW != p.W) return -1000;
  if (p.ph != (kh - 1) / 2 || p.pw != (kw - 1) / 2 || !(p.ktable_uniform & 8)) return -1000;
  if (p.kchunks % (8 * kh * kw) != 0 || p.src_gstride != 0 || p.out_gstride != 0) return -1000;
  if ((long long)p.cout_pad * p.kchunks * 16 >= (1ll << 31)) return -1000;
  for (int i = 0; i < p.nsrc; ++i)
    if ((long long)p.N * p.H * p.W * p.src[i].cstride * 2 >= (1ll << 31)) return -1000;
  if ((p.cout_g > 16 && p.cout_g < 48) || p.H < 8 || p.W < 8) return -1000;     // (the host does not build tri-product tables for such layers)
  if (!((kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1))) return -1000;
  const long long blk128 = (long long)p.N * ((p.H + 7) / 8) * ((p.W + 15) / 16) * ((p.cout_g + 127) / 128);
  const bool n64 = cfg == 72 || (cfg != 71 && (p.cout_g <= 64 || (p.cout_g <= 192 && p.cout_g % 128 != 0 && p.cout_g % 128 <= 64) ||
                                               (blk128 <= 256 && p.cout_g % 64 == 0)));
  const int bn = p.cout_g <= 16 ? 16 : (n64 ? 64 : 128);
  ConvParams q = p;
  if (bn != 16 && p.residual != nullptr && p.preadd == nullptr && p.fuse == PP_FUSE_NONE && p.act == PP_ACT_NONE && p.out_f16 &&
      p.out_scale == 1.f && p.cout_g % bn == 0 && ((p.res_cstride | p.res_choff | p.res_lo) & 7) == 0 && ((uintptr_t)p.residual % 16) == 0) {
    // a residual added to a LINEAR convolution is a pre-activation addend (see conv_v3_dispatch): through the matrix cores, both planes
    q.preadd = p.residual; q.preadd_cstride = p.res_cstride; q.preadd_choff = p.res_choff; q.preadd_lo = p.res_lo;
    q.residual = nullptr; q.act = p.act2; q.act_param = 0.f; q.act2 = PP_ACT_NONE;
  }
  if (bn != 16 && (cfg == 82 || cfg == 83 || (cfg == 0 && conv_h8_auto(q, bn)))) return conv_h8_dispatch(q, cfg == 83 ? 64 : (cfg == 82 ? 128 : bn), true, stream);
  static const bool shared_w = !(getenv("PP_HALO_PRIVATE_WEIGHTS") != nullptr && getenv("PP_HALO_PRIVATE_WEIGHTS")[0] == '1');
  if (kh == 3) return launch_v3s<3, 3>(q, bn, shared_w, stream);
  if (kh == 1) return launch_v3s<1, 5>(q, bn, shared_w, stream);
  return launch_v3s<5, 1>(q, bn, shared_w, stream);
}

}  // namespace pp
