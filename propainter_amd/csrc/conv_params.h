// Kernel-side parameter block shared by the two implicit-GEMM convolution families (conv_gemm.hip: register-staged
// fp32 / fp16 / deformable; conv_gemm_v2.hip: LDS-DMA fp16).  Filled by pp_conv2d from pp_conv_args_t.
#pragma once
#include "common.h"

namespace pp {

struct ConvSrc {
  const char* ptr;
  int cstride, choff, cgroup;
  int lo;                         // split-plane source: element offset of its lo plane (pp_conv_src_t.lo_off; tri-product layers)
};

struct ConvParams {
  int N, H, W, OH, OW, sh, sw, ph, pw, pad_mode;
  int cout_g, cout_pad, kchunks, nsrc;
  ConvSrc src[PP_CONV_MAX_SRC];
  const int4* ktable;
  const char* weight;
  long long weight_gstride;
  const float* bias;
  int act;
  float act_param, out_scale;
  const char* residual;
  int res_cstride, res_choff, act2, out_f16;
  char* out;
  int out_cstride, out_choff, out_cgroup;
  long long src_gstride, out_gstride;
  const char* dcn;
  int dcn_cstride, dcn_mask_off;
  long long M;
  int tiles_m, tiles_n, groups;   // v2 only: 1-D grid decomposition
  int ktable_uniform;             // v2 only: bit 4 / bit 8 set when every 4- / 8-chunk K step is one (tap, source) run
  int tap_h, tap_w;               // v3 only: rectangular dilation-1 tap window (0 = unknown)
  // fused recurrent-cell epilogue (pp_conv_args_t: preadd / fuse*)
  const char* preadd;
  int preadd_cstride, preadd_choff;
  int fuse, fuse_split;
  const char* fuse_a;
  int fuse_a_cstride, fuse_a_choff;
  const char* fuse_b;
  int fuse_b_cstride, fuse_b_choff;
  char* out2;
  int out2_cstride, out2_choff;
  // split-plane ("f16x3") operands (pp_conv_args_t.split): element offset of the lo plane of each fp16 operand the epilogue touches
  int split;
  int out_lo, out2_lo, preadd_lo, res_lo, fuse_a_lo, fuse_b_lo;
  // plain fp32 output without epilogue operands: store the accumulators directly (4 consecutive couts = 16 bytes per lane, 64 contiguous
  // bytes per pixel row and MFMA tile) instead of staging them through LDS -- short-K, output-bound launches (the correlation-volume GEMM)
  int epi_direct;
  unsigned long long* dcn_stats;   // conv_dcn.hip: optional fallback counters (pp_conv_args_t.dcn_stats)
};

// activation of the late (post-staging) epilogue path: same fast forms as the register path of conv_epilogue.h
__device__ __forceinline__ float act_late(float v, int act, float slope) {
  if (act < PP_ACT_SIGMOID) return v > 0.f ? v : v * slope;
  if (act == PP_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
  if (act == PP_ACT_TANH) return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
  return apply_act_special(v, act);
}

// conv_gemm_v2.hip; returns -1000 when the shape is outside that family (caller falls back to conv_gemm.hip)
int conv_v2_dispatch(const ConvParams& p, int cfg, hipStream_t stream);
// conv_gemm_v3.hip (halo tiles); returns -1000 when the shape is outside that family (caller falls back to v2)
int conv_v3_dispatch(const ConvParams& p, int cfg, hipStream_t stream);
// conv_gemm_v3.hip, ping-pong form (conv_halo8.h: 256-pixel tiles, 8 waves): explicit launch / the automatic choice
int conv_h8_dispatch(const ConvParams& p, int bn, bool split, hipStream_t stream);
bool conv_h8_auto(const ConvParams& p, int bn);
// conv_gemm_v3s.hip / conv_gemm_v2s.hip: the same kernels with split-plane epilogues (p.split); -1000 outside the family
int conv_v3s_dispatch(const ConvParams& p, int cfg, hipStream_t stream);
int conv_v2s_dispatch(const ConvParams& p, int cfg, hipStream_t stream);
// conv_dcn.hip (patch-staged modulated deformable 3x3 convolution, fp16); returns -1000 when the layer is outside that family
int conv_dcn_dispatch(const ConvParams& p, hipStream_t stream, int dbg = 0);
// conv_head.hip (streaming 3x3 convolution with at most 4 couts, VALU dot products); -1000 outside that family
int conv_head_dispatch(const ConvParams& p, hipStream_t stream, bool force = false);
// conv_gemm_ast.hip (A-stationary short-K GEMM); returns -1000 when the layer is outside that family
int conv_ast_dispatch(const ConvParams& p, int cfg, hipStream_t stream);

}  // namespace pp
