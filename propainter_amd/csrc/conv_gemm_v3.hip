// fp16 implicit-GEMM convolution, third generation: HALO TILES.  Same GEMM view, K-chunk table and packed weights as
// conv_gemm_v2.hip (channel-block-major K order), for the stride-1 "same" convolutions with a small rectangular tap
// window (3x3, 1x5, 5x1) whose sources are multiples of 64 channels -- the RAFT update block, the propagation / offset
// convolutions, the decoders: most of the path's FLOPs.
//
// Why: measured on MI355X (profiles/r1_conv_dma_vs_mfma.txt) the v2 kernel is bound by its L2->LDS gather, not by MFMA:
// with the MFMAs removed a GRU convolution still takes 78 % of its time, with the gather removed it runs at
// 1.2-1.3 PFLOP/s.  v2 re-fetches every pixel once per tap (9x for 3x3).  Here a block owns a 2-D tile of TH x TW
// output pixels and, per 64-channel block, DMA-loads the (TH+KH-1) x (TW+KW-1) input patch ONCE (zero padding = the
// buffer range check); all KH*KW taps then read their A fragments from that patch at a row offset dy*PW + dx.  The
// activation traffic into LDS drops by 6.4x (3x3) / 4x (1x5, 5x1); only the weight tiles (L1/L2-resident, 16 KiB per K
// step) still stream every step.
//
//   LDS: 2 patch buffers [PROWS][64 ch] (double buffered across channel blocks; the next block's patch arrives in
//        pieces, one LDS-DMA instruction per wave per tap step) + 2 weight stages [BN][64] + epilogue tile (aliased).
//   Swizzle: patch: 16-byte slot ^ (row & 7) on both the DMA source chunk and the fragment read, keyed by the PATCH row.
//        A fragment is 16 consecutive patch rows starting at ANY row (tile row + tap shift), and ds_read_b128 is serviced
//        in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): with the v2 key
//        ((row >> 1) & 7) only starts that are multiples of 4 rows are conflict-free (24 of 32 alignments are 2-way);
//        (row & 7) is conflict-free at every alignment (exhaustive check: tools/lds_swizzle_check.py).  The weight tile
//        keeps ((row >> 1) & 7): its fragments start at multiples of 16 rows.
//   4 waves (2 x 2), wave tile 64 px x BN/2 couts, v_mfma_f32_16x16x32_f16, fp32 accumulate.  Epilogue as in v2.
#include "conv_halo8.h"
#include <stdlib.h>

namespace pp {

// Tiles of at most 64 couts can run with wave-private weight stages (conv_halo.h, PRIVB: no block barrier inside a channel block).
// MEASURED (profiles/r3h_halo_private_weights_ab.txt, four interleaved bench runs on one box): no difference -- 1 551 / 1 556 ms per clip
// with shared stages, 1 560 / 1 557 with private ones; the per-tap barrier is not what the step waits for.  Shipped default: shared
// stages (one more block per CU for the 16-cout tiles); PP_HALO_PRIVATE_WEIGHTS=1 selects the private form.
static bool halo_shared_weights() {
  static const bool v = !(getenv("PP_HALO_PRIVATE_WEIGHTS") != nullptr && getenv("PP_HALO_PRIVATE_WEIGHTS")[0] == '1');
  return v;
}
template <int TH, int TW, int KH, int KW>
static int launch_plain_k(const ConvParams& p, bool n64, hipStream_t stream) {
  if (!n64) return launch_v3<TH, TW, KH, KW, 128>(p, stream);
  return halo_shared_weights() ? launch_v3<TH, TW, KH, KW, 64>(p, stream) : launch_v3<TH, TW, KH, KW, 64, false, 0, 64, false, true>(p, stream);
}
static int launch_plain(const ConvParams& p, int kh, int kw, bool n64, hipStream_t stream) {
  if (kh == 3 && kw == 3) return launch_plain_k<8, 16, 3, 3>(p, n64, stream);
  if (kh == 1 && kw == 5) return launch_plain_k<8, 16, 1, 5>(p, n64, stream);
  if (kh == 5 && kw == 1) return launch_plain_k<16, 8, 5, 1>(p, n64, stream);
  return -1000;
}
template <int TH, int TW, int KH, int KW>
static int launch_tiny(const ConvParams& p, hipStream_t stream) {     // 16-cout tiles (flow head, RGB decoder)
  return halo_shared_weights() ? launch_v3<TH, TW, KH, KW, 16>(p, stream) : launch_v3<TH, TW, KH, KW, 16, false, 0, 64, false, true>(p, stream);
}

// Ping-pong form (conv_halo8.h): 256-pixel tiles, eight waves in two groups in opposite phases, one block per CU.  MEASURED
// (profiles/r6_halo8_pingpong.txt: kbench + tools/bench_split.py, interleaved with the 128-pixel kernel on one box, bit-identical):
// 3-9 % SLOWER on the fp16 128-cout layers, 17 % slower with 64-cout tiles, 2.5-4 % slower on the split-plane GRU / flow-head
// layers -- with the DMA pieces issued between the MFMA clusters; 9-24 % / 4 % slower with the DMA at the head of the READ phase;
// s_setprio on the MFMA phase, on the READ phase or nowhere: no difference.  Not dispatched automatically: impl 82 / 83 select it
// (128- / 64-cout tiles), PP_HALO8=1 lets the automatic choice take it for 128-cout tiles of launches with >= 256 blocks.
static bool halo8_enabled() {
  static const bool v = getenv("PP_HALO8") != nullptr && getenv("PP_HALO8")[0] == '1';
  return v;
}
template <int BN, bool SPLIT>
static int launch_h8_taps(const ConvParams& p, int kh, int kw, hipStream_t stream) {
  if (kh == 3 && kw == 3) return launch_h8<16, 16, 3, 3, BN, SPLIT>(p, stream);
  if (kh == 1 && kw == 5) return launch_h8<16, 16, 1, 5, BN, SPLIT>(p, stream);
  if (kh == 5 && kw == 1) return launch_h8<16, 16, 5, 1, BN, SPLIT>(p, stream);
  return -1000;
}
// enough 256-pixel tiles to give every CU a block, and images that fill 16 x 16 tiles reasonably
static bool halo8_fits(const ConvParams& p, int bn) {
  const long long blk = (long long)p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16) * ((p.cout_g + bn - 1) / bn);
  return p.H >= 16 && p.W >= 16 && blk >= 256;
}
int conv_h8_dispatch(const ConvParams& p, int bn, bool split, hipStream_t stream) {
  if (split) return bn == 128 ? launch_h8_taps<128, true>(p, p.tap_h, p.tap_w, stream) : launch_h8_taps<64, true>(p, p.tap_h, p.tap_w, stream);
  return bn == 128 ? launch_h8_taps<128, false>(p, p.tap_h, p.tap_w, stream) : launch_h8_taps<64, false>(p, p.tap_h, p.tap_w, stream);
}
bool conv_h8_auto(const ConvParams& p, int bn) { return halo8_enabled() && bn == 128 && halo8_fits(p, bn); }

// Returns -1000 when the shape is outside the halo-tile family (caller falls back to v2).
// cfg: 0 = auto, 70 = force (BN by cout), 71 = BN 128, 72 = BN 64.
int conv_v3_dispatch(const ConvParams& p, int cfg, hipStream_t stream) {
  const int kh = p.tap_h, kw = p.tap_w;
  if (kh <= 0 || kw <= 0) return -1000;
  if (p.groups != 1 || p.sh != 1 || p.sw != 1 || p.pad_mode != 0 || p.OH != p.H || p.OW != p.W) return -1000;
  if (p.ph != (kh - 1) / 2 || p.pw != (kw - 1) / 2 || !(p.ktable_uniform & 8)) return -1000;
  if (p.kchunks % (8 * kh * kw) != 0 || p.src_gstride != 0 || p.out_gstride != 0) return -1000;
  if ((long long)p.cout_pad * p.kchunks * 16 >= (1ll << 31)) return -1000;
  for (int i = 0; i < p.nsrc; ++i)
    if ((long long)p.N * p.H * p.W * p.src[i].cstride * 2 >= (1ll << 31)) return -1000;
  if (cfg == 0 && ((p.cout_g > 16 && p.cout_g < 48) || p.H < 8 || p.W < 8)) return -1000;      // 17..47 couts / tiny maps: v2's narrow tiles do better
  // 64-cout tiles when the last 128-cout tile would be at most half full (cout 192 = 3 x 64: +5 % over 128 + 64-of-128, measured)
  // ... and when the 128-cout tiles would not even give every CU one block (flow-completion chains: 2 x 90 x 160 px = 225 tiles):
  // twice the blocks of half the size run 8-12 % faster there, 12 % slower at 450 tiles (tools/kbench 71 vs 72)
  const long long blk128 = (long long)p.N * ((p.H + 7) / 8) * ((p.W + 15) / 16) * ((p.cout_g + 127) / 128);
  const bool n64 = cfg == 72 || (cfg != 71 && (p.cout_g <= 64 || (p.cout_g <= 192 && p.cout_g % 128 != 0 && p.cout_g % 128 <= 64) ||
                                               (blk128 <= 256 && p.cout_g % 64 == 0)));
  if (cfg == 73 || (cfg == 0 && p.cout_g <= 16)) {   // tiny cout (flow head, RGB decoder): A-bandwidth bound, halo tiles cut the gather 6x
    if (kh == 3 && kw == 3) return launch_tiny<8, 16, 3, 3>(p, stream);
    if (kh == 1 && kw == 5) return launch_tiny<8, 16, 1, 5>(p, stream);
    if (kh == 5 && kw == 1) return launch_tiny<16, 8, 5, 1>(p, stream);
    return -1000;
  }
  if (cfg == 109) {   // the pre-activation addend / residual read by the epilogue instead of going through the matrix cores (A/B; tests/test_ops_gpu.py)
    if (kh == 3 && kw == 3) return launch_v3<8, 16, 3, 3, 128, false, 11>(p, stream);
    if (kh == 1 && kw == 5) return launch_v3<8, 16, 1, 5, 128, false, 11>(p, stream);
    if (kh == 5 && kw == 1) return launch_v3<16, 8, 5, 1, 128, false, 11>(p, stream);
    return -1000;
  }
#if defined(PP_DIAG)      // tools/kbench --prof: in-kernel phase stamps (BN 128); every other schedule experiment of rounds 2-3 lives in experiments/
  if (cfg == 75) {
    if (kh == 3 && kw == 3) return launch_v3<8, 16, 3, 3, 128, true>(p, stream);
    if (kh == 1 && kw == 5) return launch_v3<8, 16, 1, 5, 128, true>(p, stream);
    if (kh == 5 && kw == 1) return launch_v3<16, 8, 5, 1, 128, true>(p, stream);
    return -1000;
  }
#endif
  if (cfg != 109 && p.cout_g >= 32 && p.residual != nullptr && p.preadd == nullptr && p.fuse == PP_FUSE_NONE && p.act == PP_ACT_NONE && p.out_f16 &&
      p.out_scale == 1.f && p.cout_g % (n64 ? 64 : 128) == 0 && ((p.res_cstride | p.res_choff) & 7) == 0 && ((uintptr_t)p.residual % 16) == 0) {
    // a residual added to a LINEAR convolution (out = act2(conv + bias + residual)) is a pre-activation addend: same value, and the
    // addend path goes through the matrix cores instead of being read row by row in the epilogue (see PRE_MFMA in the kernel)
    ConvParams q = p;
    q.preadd = p.residual; q.preadd_cstride = p.res_cstride; q.preadd_choff = p.res_choff;
    q.residual = nullptr; q.act = p.act2; q.act_param = 0.f; q.act2 = PP_ACT_NONE;
    if (cfg == 82 || cfg == 83 || (cfg == 0 && conv_h8_auto(q, n64 ? 64 : 128))) return conv_h8_dispatch(q, cfg == 83 || (cfg == 0 && n64) ? 64 : 128, false, stream);
    return launch_plain(q, kh, kw, n64, stream);
  }
  if (cfg == 82 || cfg == 83 || (cfg == 0 && conv_h8_auto(p, n64 ? 64 : 128))) return conv_h8_dispatch(p, cfg == 83 || (cfg == 0 && n64) ? 64 : 128, false, stream);
  return launch_plain(p, kh, kw, n64, stream);
}

}  // namespace pp

// [diagnostic, not part of the public header] reads and clears the phase counters of the PROF build:
// out[0..7] = {vmcnt wait, barrier wait, DMA issue, compute, main loop total, epilogue, waves, K steps} (cycles summed over waves),
// out[12] / out[13] = earliest block start / latest block end stamp since the last call (16 values)
extern "C" int pp_debug_conv_prof(unsigned long long* out) {
  unsigned long long zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, ~0ull, 0, 0, 0};    // ([12] is a minimum)
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::g_v3_prof), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(pp::g_v3_prof), zero, sizeof(zero));
  return (int)e;
}
