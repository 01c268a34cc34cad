// fp16 implicit-GEMM convolution, second generation: LDS tiles are filled by the LDS-DMA path
// (global_load_lds_dwordx4, one 16-byte K-chunk per lane) instead of being staged through VGPRs, so
// the im2col gather costs no ds_write pass and no staging registers.
//
//   D[co][px] = sum_k Wp[co][k] * A[px][k]     (same GEMM view, K-chunk table and packed weights as
//                                               conv_gemm.hip; see include/propainter_hip.h)
//
// Tile: BM pixels x BN couts x BK (32 or 64) per step, NW waves, each wave owns WM x WN of the tile as
// (WM/16) x (WN/16) v_mfma_f32_16x16x32_f16 accumulators.  S LDS stages: the DMA of step k+S-1 is issued right
// after the single (raw) barrier of step k, and a *counted* s_waitcnt vmcnt((S-2)*G) (G = DMA instructions per
// lane per step) retires only the stage about to be consumed, so S-2 steps of loads stay in flight across
// every barrier (L2 / Infinity-Cache latency is ~1-2k cycles, a 16-MFMA step is 256).  The K-chunk table entries of
// a step are wave-uniform and fetched by scalar loads, so the loop contains no VMEM load besides the DMA (an
// ordinary vector load, or a ds_read of a table kept in LDS, makes hipcc drain vmcnt to 0).
// Epilogue: accumulators go through LDS (fp32) so that every lane stores 8 consecutive output channels
// (16 B fp16 / 32 B fp32) and 8 lanes cover one pixel's 64-cout slice: full-line NHWC stores.
//
// LDS image: row-major [row][BK] fp16 (64- or 128-byte rows), A rows (pixels) then B rows (couts).  An
// LDS-DMA instruction writes 64 lanes x 16 B = 1 KiB *linearly*, i.e. 8 (BK=64) or 16 (BK=32) whole rows,
// so 8 / 4 adjacent lanes fetch one contiguous 128- / 64-byte piece of an NHWC pixel (coalesced).  Bank
// conflicts of the fragment reads (ds_read_b128, row = lane&15, chunk = k/8 + lane>>4) are removed by an XOR
// swizzle of the 16-byte slot inside each row; because the DMA destination is lane-linear the swizzle is
// applied to the per-lane *source* chunk and again on the read (same involution on both sides):
//     BK=64:  slot = chunk ^ ((row >> 1) & 7)          BK=32:  slot = chunk ^ G[(row >> 2) & 3], G = {0,3,2,1}
// Zero padding / K padding: out-of-image taps and padding chunks fetch from the all-zero int4 that
// pp_conv_build_ktable appends after the last table entry (ktable[kchunks]).
#include "conv_v2_kernel.h"

namespace pp {

// `cfg`: 0 = auto; otherwise a tile configuration id (exposed through pp_conv_args_t.impl for the tile sweep in
// tools/bench_conv.py; +100 disables the uniform-step fast path).  Returns -1000 when the shape is not supported by
// this family (caller falls back).
int conv_v2_dispatch(const ConvParams& p, int cfg, hipStream_t stream) {
  if (p.kchunks % 8 != 0 || p.M >= (1ll << 31) || (long long)p.N * p.H * p.W >= (1ll << 31)) return -1000;
  // uniform-step path: zeros padding, 32-bit byte offsets (< 2 GiB per source / weight block)
  bool uni = p.ktable_uniform != 0 && p.pad_mode == 0 && (long long)p.cout_pad * p.kchunks * 16 < (1ll << 31);
  for (int i = 0; i < p.nsrc; ++i) uni = uni && (long long)p.N * p.H * p.W * p.src[i].cstride * 2 < (1ll << 31);
  if (cfg >= 100) { uni = false; cfg -= 100; }
  if (cfg == 0) {   // measured on MI355X with tools/bench_conv.py (profiles/r1_conv_tile_sweep.txt)
    // 256 x 256 tiles move half the LDS-DMA bytes per MFMA of 256 x 128: +7..30 % on the transformer GEMMs (profiles/r4_tile256x256.txt),
    // same K order, so the results are bit-identical; they need at least one block per CU to pay
    if (p.cout_g >= 512) cfg = ((p.M + 255) / 256) * ((p.cout_g + 255) / 256) * p.groups >= 256 ? 18 : 13;
    else if (p.cout_g > 64) cfg = 12;
    else if (p.cout_g > 32) cfg = 22;
    else if (p.cout_g > 16) cfg = p.groups > 1 ? 31 : 32;
    else cfg = 42;
  }
  switch (cfg) {
    //                           BM   BN  BK  WM WN  S
    case 10: return launch_v2<128, 128, 32, 2, 2, 4>(p, uni, stream);   // 4 waves 64x64, 4 stages: 64 KiB -> 2 blocks/CU
    case 11: return launch_v2<256, 128, 32, 4, 2, 4>(p, uni, stream);   // 8 waves 64x64, 4 stages: 96 KiB
    case 12: return launch_v2<128, 128, 64, 2, 2, 2>(p, uni, stream);   // 2 stages, 64 KiB -> 2 blocks/CU
    case 13: return launch_v2<256, 128, 64, 4, 2, 3>(p, uni, stream);   // 8 waves, BK=64, 3 stages: 144 KiB
    case 14: return launch_v2<128, 128, 32, 2, 2, 3>(p, uni, stream);   // 48 KiB -> 3 blocks/CU
    case 15: return launch_v2<128, 128, 64, 2, 2, 3>(p, uni, stream);   // 96 KiB -> 1 block/CU
    case 16: return launch_v2<256, 128, 32, 2, 2, 4>(p, uni, stream);   // 4 waves 128x64, 96 KiB
    case 17: return launch_v2<256, 128, 64, 2, 2, 2>(p, uni, stream);   // 4 waves 128x64, 96 KiB
    case 18: return launch_v2<256, 256, 64, 4, 2, 2>(p, uni, stream);   // 8 waves 64x128, 2 stages: 128 KiB -- half the DMA bytes per MFMA of 13
    // 19 (sweep only, NOT measured yet -- compiled at the end of round 4 without GPU time left): the same block tile as 4 waves of 128 x 128,
    // one wave per SIMD, the 256 accumulator registers in AGPRs (256 VGPRs + 256 AGPRs, no scratch inside the K loop, 784 B/lane in the
    // epilogue): half the LDS fragment reads per MFMA of 18 (8 waves of 64 x 128 read 192 B/clk per CU at the MFMA rate, 75 % of the LDS)
    case 19: return launch_v2<256, 256, 64, 2, 2, 2>(p, uni, stream);
    case 20: return launch_v2<128, 64, 32, 2, 2, 4>(p, uni, stream);    // wave tile 64x32, 48 KiB -> 3 blocks/CU
    case 21: return launch_v2<256, 64, 32, 4, 1, 4>(p, uni, stream);    // wave tile 64x64, 80 KiB
    case 22: return launch_v2<256, 64, 64, 4, 1, 2>(p, uni, stream);    // 80 KiB -> 2 blocks/CU
    case 30: return launch_v2<256, 32, 32, 4, 1, 4>(p, uni, stream);    // wave tile 64x32, 72 KiB
    case 31: return launch_v2<256, 32, 32, 4, 1, 3>(p, uni, stream);    // 54 KiB -> 2 blocks/CU
    case 32: return launch_v2<256, 32, 32, 4, 1, 2>(p, uni, stream);    // 36 KiB
    case 40: return launch_v2<256, 16, 32, 4, 1, 4>(p, uni, stream);    // wave tile 64x16
    case 41: return launch_v2<256, 16, 32, 4, 1, 3>(p, uni, stream);
    case 42: return launch_v2<256, 16, 32, 4, 1, 2>(p, uni, stream);
#if defined(PP_DIAG)      // tuning / diagnostic variants (tools/kbench, PP_DIAG=1 builds only; 60..68 are WRONG BY DESIGN: DMA-only / MFMA-only)
    // SCHED 1 variants (all fragment reads of a K step up front)
    case 50: return launch_v2<128, 128, 64, 2, 2, 2, 1>(p, uni, stream);
    case 51: return launch_v2<256, 128, 64, 4, 2, 3, 1>(p, uni, stream);
    case 52: return launch_v2<128, 128, 32, 2, 2, 4, 1>(p, uni, stream);
    case 53: return launch_v2<128, 128, 32, 2, 2, 3, 1>(p, uni, stream);
    case 54: return launch_v2<256, 128, 32, 4, 2, 4, 1>(p, uni, stream);
    case 60: return launch_v2<128, 128, 64, 2, 2, 2, 2>(p, uni, stream);   // diagnostics (wrong results by design)
    case 61: return launch_v2<128, 128, 64, 2, 2, 2, 3>(p, uni, stream);
    case 64: return launch_v2<128, 128, 32, 2, 2, 4, 2>(p, uni, stream);
    case 65: return launch_v2<128, 128, 64, 2, 2, 3, 2>(p, uni, stream);
    case 66: return launch_v2<256, 128, 32, 4, 2, 4, 2>(p, uni, stream);
    case 67: return launch_v2<128, 128, 32, 2, 2, 3, 2>(p, uni, stream);
    case 68: return launch_v2<128, 128, 32, 2, 2, 4, 3>(p, uni, stream);
    case 62: return launch_v2<256, 128, 64, 4, 2, 3, 2>(p, uni, stream);
    case 63: return launch_v2<256, 128, 64, 4, 2, 3, 3>(p, uni, stream);
#endif
    default: return -1000;
  }
}

}  // namespace pp
