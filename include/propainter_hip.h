/*
 * libpropainter_hip — C-ABI of the MI355X (gfx950) ProPainter inference kernels.
 *
 * The reference (sczhou/ProPainter) is pure Python: its native arithmetic lives in torch /
 * torchvision wheels.  Each entry point below replaces the native op(s) named in its comment
 * (file:line relative to the reference root).  Conventions:
 *   - every symbol is extern "C"; plain C types; tensors are raw device pointers + explicit sizes;
 *   - the caller owns every buffer (including index tables and workspaces); kernels never allocate,
 *     free or retain pointers; weights are read-only;
 *   - all work is enqueued asynchronously on the `stream` argument (a hipStream_t passed as void*);
 *     no implicit synchronisation;
 *   - return 0 on success, a negative PP_ERR_* code for argument errors, a positive value = raw
 *     hipError_t; the message is retrievable with pp_last_error_string() (thread-local);
 *   - activation layout inside the engine is NHWC ("pixel-major"): element (n,y,x,c) of a tensor lives
 *     at ((n*H + y)*W + x)*cstride + choff + c, so concatenations are expressed as channel windows of
 *     a wider buffer instead of copies;
 *   - dtype codes: PP_F32 = 0, PP_F16 = 1; PP_F16S = 2 where an entry point says so: split-plane fp16 (value = hi + lo, the lo
 *     plane `lo_off` elements after the hi plane in the pixel row; see pp_conv_args_t.split).  Accumulation is always fp32.
 */
#ifndef PROPAINTER_HIP_H
#define PROPAINTER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_F32 0
#define PP_F16 1
#define PP_F16S 2   /* split-plane fp16 pair (hi, lo): only where an entry point documents it */

#define PP_ERR_ARG (-1)        /* bad shape / null pointer / unsupported size        */
#define PP_ERR_DTYPE (-2)      /* unsupported dtype code                              */
#define PP_ERR_ALIGN (-3)      /* pointer or channel count violates alignment rule    */
#define PP_ERR_WORKSPACE (-4)  /* caller-provided table / workspace too small         */

/* activation codes for fused epilogues */
#define PP_ACT_NONE 0
#define PP_ACT_RELU 1
#define PP_ACT_LRELU 2   /* slope = act_param */
#define PP_ACT_SIGMOID 3
#define PP_ACT_TANH 4
#define PP_ACT_GELU 5    /* exact erf GELU (torch.nn.GELU default) */

int pp_version(void);
const char* pp_last_error_string(void);
/* sizeof() of the argument structs as compiled, for bindings to verify their layouts */
int pp_sizeof_conv_args(void);
int pp_sizeof_attn_args(void);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear / batched GEMM on MFMA  (replaces every nn.Conv2d, nn.Conv3d,
 * nn.Linear and torch.matmul on the path: RAFT/extractor.py:168-192, RAFT/update.py:89-136,
 * model/recurrent_flow_completion.py:206-256, model/propainter.py:198-216,266-273,
 * model/modules/sparse_transformer.py:17,39,43,68-69,123-130; and, with `dcn_offmask` set, the
 * sampling + GEMM of torchvision.ops.deform_conv2d at model/propainter.py:67-69 and
 * model/recurrent_flow_completion.py:42-44).
 *
 * GEMM view: M = N*OH*OW output pixels, K = (taps x concatenated source channels), per group.
 * K is walked in chunks of 8 channels described by a device table (pp_conv_build_ktable) so that
 * arbitrary tap lists (dilation, phase-decomposed transposed convs) and channel concatenations of
 * up to PP_CONV_MAX_SRC sources are one kernel.  Weights are pre-packed [groups][cout_pad][K].
 * ---------------------------------------------------------------------------------------------- */
#define PP_CONV_MAX_SRC 4

typedef struct {
  const void* ptr;  /* NHWC base of this source                          */
  int32_t cstride;  /* elements per pixel                                */
  int32_t choff;    /* first channel used (for group 0)                  */
  int32_t cgroup;   /* channel advance per group (0 when groups == 1)    */
  int32_t lo_off;   /* split-plane source (pp_conv_args_t.split == 2): element offset of its lo plane, else 0 */
} pp_conv_src_t;

typedef struct {
  int32_t dtype;                 /* PP_F32 / PP_F16: sources, weights, residual, dcn_offmask   */
  int32_t N, H, W;               /* input extent shared by all sources                         */
  int32_t OH, OW;                /* output extent                                              */
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;          /* input coordinate = out*stride - pad + tap offset           */
  int32_t pad_mode;              /* 0 = zeros, 1 = replicate (clamp)                           */
  int32_t groups;                /* grid.z; also used as the batch count of a batched GEMM     */
  int32_t cout_g;                /* real output channels per group                             */
  int32_t cout_pad;              /* rows per group present in `weight` (>= cout_g; extra rows 0) */
  int32_t kchunks;               /* K/8 per group, multiple of 8 (pp_conv_build_ktable pads)   */
  int32_t nsrc;
  pp_conv_src_t src[PP_CONV_MAX_SRC];
  const int32_t* ktable;         /* device, (kchunks + 1) x int4 {dy, dx, src | g<<8 | tap<<16, choff}; src 255 = zero
                                    chunk; entry [kchunks] is all-zero (the kernels' 16-byte zero page)            */
  const void* weight;            /* device, packed [groups][cout_pad][kchunks*8], dtype        */
  int64_t weight_gstride;        /* elements between groups in `weight` (cout_pad*K normally)  */
  const float* bias;             /* device fp32 [groups*cout_g] or NULL                        */
  int32_t act;                   /* PP_ACT_* applied to (acc + bias)*out_scale                 */
  float act_param;
  float out_scale;
  const void* residual;          /* optional NHWC tensor added after `act` (dtype = dtype)     */
  int32_t res_cstride, res_choff;
  int32_t act2;                  /* PP_ACT_NONE or PP_ACT_RELU applied after the residual add  */
  int32_t out_dtype;             /* PP_F32 / PP_F16                                            */
  void* out;                     /* NHWC [N,OH,OW,...]                                          */
  int32_t out_cstride, out_choff;
  int32_t out_cgroup;            /* channel advance per group in `out` (cout_g normally)       */
  int64_t src_gstride;           /* batched GEMM: element advance of src[0].ptr per group      */
  int64_t out_gstride;           /* batched GEMM: element advance of `out` per group           */
  /* deformable sampling (modulated, 16 offset groups, 3x3): NHWC [N,H,W,dcn_cstride] holding
   * 288 offset channels (dy,dx interleaved per (group,tap)) followed by 144 modulation masks. */
  const void* dcn_offmask;
  int32_t dcn_cstride;
  int32_t dcn_mask_off;          /* channel index of the first modulation mask (288)           */
  int32_t impl;                  /* 0 = auto; 1 = force the register-staged kernel; 2 = LDS-DMA kernel without the
                                    uniform-step fast path; 10..69 = a specific LDS-DMA tile configuration (+100: no
                                    fast path); 70..72 = force the halo-tile kernel (fp16 only; tile sweeps, see
                                    csrc/conv_gemm_v2.hip, csrc/conv_gemm_v3.hip)                                  */
  int32_t ktable_uniform;        /* bit mask describing `ktable`: 4 / 8 set when every aligned run of 4 / 8 chunks is one
                                    (tap, source) with consecutive channel offsets (true when every source has a
                                    multiple of 32 / 64 channels); 0 = unknown (always correct)                   */
  int32_t tap_h, tap_w;          /* > 0: the table describes a dense, dilation-1 tap_h x tap_w window in row-major tap order
                                    (enables the halo-tile kernel for stride-1 "same" convolutions); 0 = unknown    */
  /* ---- fused recurrent-cell epilogue (SepConvGRU, RAFT/update.py:45-60); all optional, groups == 1 ------------
   * v = act((acc + bias) * out_scale + preadd[pixel, co]) -- `preadd` is a partial sum computed ahead of time by
   * another pp_conv2d over the iteration-invariant input channels (NHWC, dtype = dtype), added BEFORE the activation;
   * fuse == PP_FUSE_GRU_ZR (cout_g = 2C, fuse_split = C): co <  C: out[pixel, out_choff + co]      = v           (z)
   *                                                       co >= C: out2[pixel, out2_choff + co - C] = v * a[co-C] (r*h)
   * fuse == PP_FUSE_GRU_H  (cout_g = C):  out[pixel, out_choff + co] = (1 - b[co]) * a[co] + b[co] * v    (new h)
   *   with a = fuse_a (h), b = fuse_b (z), NHWC windows of dtype; `out` may alias fuse_a (element-wise in place).
   * fuse == PP_FUSE_DCN_OFFMASK (act == PP_ACT_NONE; the conv_offset head of DeformableAlignment, model/propainter.py:57-65 and
   *   model/recurrent_flow_completion.py:31-40): co <  fuse_split: out = act_param * tanh(v) + flow[pixel, (co & 1) ? x : y]
   *                                              co >= fuse_split: out = sigmoid(v)
   *   act_param = max_residue_magnitude; fuse_a = optional flow window (NHWC, dtype, channels fuse_a_choff = x, +1 = y,
   *   fuse_a_choff even; NULL = no flow term); fuse_split even.  Same result as pp_dcn_offset_mask_act on the plain output
   *   (bit-identical for fp32; for fp16 the intermediate rounding of the plain output is skipped).                      */
  const void* preadd;
  int32_t preadd_cstride, preadd_choff;
  int32_t fuse;                  /* PP_FUSE_NONE / PP_FUSE_GRU_ZR / PP_FUSE_GRU_H / PP_FUSE_DCN_OFFMASK           */
  int32_t fuse_split;
  const void* fuse_a;
  int32_t fuse_a_cstride, fuse_a_choff;
  const void* fuse_b;
  int32_t fuse_b_cstride, fuse_b_choff;
  void* out2;
  int32_t out2_cstride, out2_choff;
  /* ---- split-plane fp16 tensors ("f16x3": reference-class fp32 arithmetic on the fp16 matrix cores; dtype must be PP_F16) ----
   * A split-plane tensor stores the fp32 value v of logical channel c as hi = fp16(v) at channel c and lo = fp16(v - hi) at
   * channel c + <lo offset> of the same pixel row (22 significand bits).  With split = 1
   *   - the SOURCES need no kernel support: the K table (pp_conv_build_ktable + the host-side expansion of
   *     propainter_amd/conv.py) walks every 64-channel block of a source three times -- hi plane x W_hi, lo plane x W_hi, hi plane x
   *     W_lo with W_hi = fp16(W), W_lo = fp16(W - W_hi) packed in that K order -- so every product is hi*hi + lo*hi + hi*lo with
   *     fp32 accumulation (the dropped lo*lo term is < 2^-22 relative);
   *   - every fp16 operand of the EPILOGUE (out when out_dtype == PP_F16, out2, preadd, residual, fuse_a, fuse_b) is split-plane with
   *     the lo offsets below (multiples of 8 elements); out_dtype == PP_F32 writes plain fp32;
   *   - PP_FUSE_DCN_OFFMASK, groups > 1 and the deformable mode are not available.
   * split = 2: TRI-PRODUCT K format for the halo-tile kernel (tap_h x tap_w in {3x3, 1x5, 5x1}, stride 1, "same" padding, sources
   *   multiples of 32 channels): a K block is 32 channels of BOTH planes of one source -- per tap 8 table chunks, 4 of the hi plane then
   *   the same 4 channel chunks of the lo plane (choff + lo offset) -- and the packed weight row holds [32 ch W_hi | 32 ch W_lo] per tap
   *   in that order; the kernel issues W_hi x A_hi + W_hi x A_lo + W_lo x A_hi per tap step (a third fewer LDS reads, weight-tile
   *   fetches and barriers per product than split = 1).  Epilogue operands as for split = 1.                                       */
  int32_t split;
  int32_t out_lo, out2_lo, preadd_lo, res_lo, fuse_a_lo, fuse_b_lo;
  int32_t pad2_;
  /* ---- optional fallback counters of the patch-staged deformable kernel (NULL: off, the production instantiation).  Device pointer
   * to 2 x uint64: [0] += bilinear samples (pixel x offset group x tap) taken, [1] += samples with a corner OUTSIDE the tile's
   * mean-shifted LDS patch (served by per-corner global reads: the kernel's slow path).  One update per wave; deterministic counts. */
  unsigned long long* dcn_stats;
} pp_conv_args_t;

#define PP_FUSE_NONE 0
#define PP_FUSE_GRU_ZR 1
#define PP_FUSE_GRU_H 2
#define PP_FUSE_DCN_OFFMASK 3  /* offset / modulation-mask head of a deformable alignment (see pp_conv_args_t.fuse) */

/* Host helper: fill `out` (kchunks_padded x 4 int32) for `ntaps` taps (dy[i], dx[i] are input
 * offsets added to out*stride - pad) over `nsrc` sources of src_channels[i] channels each (each a
 * multiple of 8).  `dcn_groups` > 0 additionally records the offset-group / tap id of every chunk.
 * Returns the padded chunk count `kchunks` (multiple of 8) or a negative error; call with out == NULL to size.
 * The buffer must hold kchunks + 1 entries: the last one is written as 16 zero bytes. */
int pp_conv_build_ktable(int ntaps, const int32_t* dy, const int32_t* dx, int nsrc,
                         const int32_t* src_channels, int dcn_groups, int32_t* out, int out_capacity);

/* Host helper: pack a reference-layout weight [cout, cin_g, kh, kw] (fp32, host; cin_g = sum of src_channels, the REAL
 * channel counts per group in source order; nn.Linear = kh = kw = 1) into the [groups][cout_pad][kchunks*8] layout
 * pp_conv2d reads, following `ktable` (host copy of the table pp_conv_build_ktable wrote for the same taps / padded
 * source widths; tap ids index the kh x kw window row-major).  Padded channels / rows are zero.  dtype = element type of
 * `out` (host).  Returns cout_pad (rows per group) or a negative error; call with out == NULL to size
 * (groups * cout_pad * kchunks * 8 elements). */
int pp_conv_pack_weight(const float* weight, int cout, int kh, int kw, int nsrc, const int32_t* src_channels, int groups,
                        const int32_t* ktable, int kchunks, int dtype, void* out, int64_t out_elems);

int pp_conv2d(const pp_conv_args_t* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flow-guided sampling (replaces F.grid_sample as used by model/modules/flow_loss_utils.py:6-45,
 * model/propainter.py:19-31,104-190)
 * ---------------------------------------------------------------------------------------------- */
/* out[n,y,x,c] = sample(x[n,:,:,c], (x + flow[n,y,x,0], y + flow[n,y,x,1])), zeros padding,
 * align_corners=True; mode 0 = bilinear, 1 = nearest (round-half-even).  x/out NHWC with channel
 * windows; flow NHWC [N,H,W,2] (fl_cstride >= 2). C must be a multiple of 8 unless C <= 4. */
int pp_flow_warp(const void* x, int x_cstride, int x_choff, const void* flow, int fl_cstride, int fl_choff,
                 void* out, int out_cstride, int out_choff, int N, int H, int W, int C, int mode,
                 int dtype, void* stream);

/* valid[n,y,x] = |fw + warp(bw, fw)|^2 < 0.01 (|fw|^2 + |warp(bw)|^2) + 0.5  (model/propainter.py:22-31);
 * flows NHWC [N,H,W,2]; valid written as dtype into out[n,y,x,out_choff] (1.0 / 0.0). */
int pp_fb_check(const void* flow_fw, int fw_cstride, const void* flow_bw, int bw_cstride, void* out,
                int out_cstride, int out_choff, int N, int H, int W, int dtype, void* stream);

/* One step of the non-learnable image propagation (model/propainter.py:137-170, learnable=False),
 * planar NCHW frames [N,3,H,W] / masks [N,1,H,W] / flows [N,2,H,W]:
 *   valid = fb_check(flow_prop, flow_check); m_w = bin(warp_bilinear(mask_prop));
 *   u = bin(m_cur*valid*(1-m_w)); x_out = u*warp_<mode>(x_prop) + (1-u)*x_cur;
 *   m_out = bin(m_cur*(1-valid*(1-m_w))).  */
int pp_img_prop_step(const void* x_prop, const void* m_prop, const void* x_cur, const void* m_cur,
                     const void* flow_prop, const void* flow_check, void* x_out, void* m_out, int N, int C,
                     int H, int W, int mode, int dtype, void* stream);

/* Mask dilation of the driver's pre-processing (scipy.ndimage.binary_dilation(mask, iterations=k) with the default
 * cross structuring element, inference_propainter.py:96,105): uint8 planar [N,H,W], non-zero = hole; out = 255 where a
 * non-zero pixel lies within L1 distance k, else 0 (k = 0: plain binarisation).  out must not alias mask. */
int pp_binary_dilate(const void* mask, void* out, int N, int H, int W, int iterations, void* stream);

/* Final resize of the composited uint8 frames on the device (reference: cv2.resize(f, out_size) = INTER_LINEAR,
 * inference_propainter.py:469-470; web-demos/hugging_face/inpainter/base_inpainter.py:368-372): src NHWC uint8 [N,H,W,C] (C <= 4) ->
 * dst [N,OH,OW,C] with OpenCV's fixed-point 8-bit bilinear arithmetic (11-bit weights, two-pass integer rounding; an exact 2:1
 * reduction in both axes averages 2x2 blocks like OpenCV's INTER_AREA shortcut).  OpenCV is an un-vendored dependency
 * (requirements.txt: opencv-python) and absent offline: the arithmetic is restated from its published algorithm. */
int pp_resize_bilinear_u8(const void* src, void* dst, int N, int H, int W, int C, int OH, int OW, void* stream);

/* Ordered uint8 composite of one generator window (inference_propainter.py:435-450) in one launch: for local frame i (clip frame
 * frame_ids[i], HOST array of n <= 32 ids)  img = uint8(((pred + 1) / 2) * 255) with every operation rounded in pred's dtype and the
 * final truncation of .astype(np.uint8);  cur = mask ? img : original;  comp = (blend_bits >> i) & 1 ? uint8(0.5f * comp + 0.5f * cur) :
 * cur.  pred planar [n,3,H,W] (dtype); mask uint8, element (frame, pixel) at (frame * H * W + pixel) * mask_stride, non-zero = hole;
 * original / comp uint8 [L,H,W,3].  Order dependent: windows must be composited in increasing position. */
int pp_composite_window(const void* pred, int dtype, const void* mask, int mask_stride, const void* original, void* comp,
                        const int32_t* frame_ids, uint32_t blend_bits, int n, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RAFT correlation (RAFT/corr.py:13-60, RAFT/utils/utils.py:57-71, RAFT/raft.py:73-84)
 * ---------------------------------------------------------------------------------------------- */
/* 2x2/stride-2 average pooling of per-source-pixel correlation maps: in [M, H, W] fp32 -> out [M, H/2, W/2]. */
int pp_corr_avgpool(const float* in, float* out, int64_t M, int H, int W, void* stream);

/* 4-level 9x9 bilinear lookup.  lvl[l] = [B*h*w, Hl, Wl] fp32 map of every source pixel; coords fp32
 * NHWC [B,h,w,2] (x, y); out NHWC [B,h,w,out_cstride] channel l*81 + a*9 + b samples
 * (x/2^l + a - 4, y/2^l + b - 4) (first index moves x; RAFT/corr.py:36-43); channels 324..C_out_pad-1 = 0.
 * out_dtype PP_F16S: split-plane output, lo plane at out_cstride / 2 (>= out_cpad). */
int pp_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3, const float* coords,
                   void* out, int out_cstride, int out_cpad, int B, int h, int w, int out_dtype, void* stream);

/* RAFT correlation WITHOUT the all-pairs volume (fp16 engine; replaces CorrBlock.__init__ + __call__,
 * RAFT/corr.py:13-60, as a pair): avg_pool(f1 . f2) = f1 . avg_pool(f2), so
 *   pp_corr_feature_pyramid pools f2 (fp16 NHWC [P,h,w,256]) once per pair into levels 1..3
 *     ([P, h>>l, w>>l, 256], mean over the 2^l x 2^l block = l nested avg_pool2d(2,2) with floor sizes), and
 *   pp_corr_lookup_otf computes per iteration the 4 x 100 dot products each pixel's 9x9 windows touch (MFMA) and
 *     blends them (bilinear, zeros padding, the reference's coordinate round trip) into out NHWC [P,h,w,out_cstride]
 *     fp16: channel l*81 + a*9 + b samples (x/2^l + a - 4, y/2^l + b - 4), scaled by 1/sqrt(256); channels
 *     [324, out_cpad) are zeroed.  coords fp32 [P,h,w,2] = (x, y).  Deterministic, batch-invariant. */
int pp_corr_feature_pyramid(const void* f2, void* lvl1, void* lvl2, void* lvl3, int P, int h, int w, void* stream);
/* The same pooling for split-plane ("f16x3") features: f2 and the levels are fp16 NHWC [P, h>>l, w>>l, 512] = 256 hi | 256 lo channels
 * of fp32 values; the means are formed in fp32 and stored as hi = fp16(m), lo = fp16(m - hi).  The volume engine of the fp32-class RAFT
 * builds levels 1..3 of the correlation pyramid (RAFT/corr.py:21-27) as GEMMs of f1 with these pooled features -- pooling is linear --
 * instead of re-reading the level-0 volume (829 MB per pair-direction at 720x1280). */
int pp_corr_feature_pyramid_split(const void* f2, void* lvl1, void* lvl2, void* lvl3, int P, int h, int w, void* stream);
int pp_corr_lookup_otf(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2, const void* f2_lvl3,
                       const float* coords, void* out, int out_cstride, int out_cpad, int P, int h, int w, void* stream);
/* The volume-free lookup at fp32-class precision ("f16x3"; replaces CorrBlock.__init__ + __call__, RAFT/corr.py:13-60, for the engine
 * that keeps RAFT at the reference's precision, inference_propainter.py:311): f1 and the four f2 levels (f2 itself and the outputs of
 * pp_corr_feature_pyramid_split) are SPLIT-PLANE fp16 NHWC [P, h>>l, w>>l, 512] = 256 hi | 256 lo; every dot product is
 * hi.hi + lo.hi + hi.lo on the fp16 matrix cores with fp32 accumulation, the blend uses pp_corr_lookup's per-tap coordinate round trip.
 * out: split-plane fp16 NHWC [P,h,w,out_cstride], hi plane at channel 0, lo plane at out_cstride / 2; level l, tap (a, b) at channel
 * l * 88 + a * 9 + b of each plane (81 taps + 7 zero channels per level: the 1x1 convolution behind it reads full 32-channel blocks;
 * permute its weight columns accordingly); out_cstride >= 704, multiple of 16.  Deterministic, batch-invariant. */
int pp_corr_lookup_otf_split(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2, const void* f2_lvl3,
                             const float* coords, void* out, int out_cstride, int P, int h, int w, void* stream);
/* The same lookups with fallback counters (round 5; `stats` = device pointer to 3 x uint64 or NULL): [0] += (tile, pyramid level)
 * units processed, [1] += units whose bounding box of correlation windows outgrew the LDS tile (processed as 16-pixel sub-tiles),
 * [2] += 16-pixel sub-tiles that fell through to single pixels.  One update per block; the lookup result does not depend on it. */
int pp_corr_lookup_otf_stats(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2, const void* f2_lvl3,
                             const float* coords, void* out, int out_cstride, int out_cpad, int P, int h, int w,
                             unsigned long long* stats, void* stream);
int pp_corr_lookup_otf_split_stats(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2, const void* f2_lvl3,
                                   const float* coords, void* out, int out_cstride, int P, int h, int w,
                                   unsigned long long* stats, void* stream);

/* Input of the motion encoder's 7x7 flow convolution (RAFT/update.py:85,92: convf1 = Conv2d(2, 128, 7, padding=3)) laid out so
 * that the convolution needs K = 7 x 16 instead of 49 taps x 8 padded channels: rows[pixel, 2*kx + c] = flow_c(x + kx - 3, y)
 * (zero outside the image row; channels 14, 15 zero), flow = coords1 - coords0 in fp32, rounded to `dtype`.  The layer is then
 * a 7x1 convolution over the 16 channels with weights w'[co, 2*kx + c, ky] = w[co, c, ky, kx].  Optionally the same flow is also
 * written to flow_out[pixel, flow_choff .. +1] (the GRU input window).  rows: NHWC [P,h,w,16] of dtype.  dtype PP_F16S: rows are
 * [P,h,w,16 hi | 16 lo] and flow_out's lo plane sits flow_cstride / 2 after its hi plane.                                        */
int pp_raft_flow_taps(const float* coords1, const float* coords0, void* rows, void* flow_out, int flow_cstride, int flow_choff,
                      int P, int h, int w, int dtype, void* stream);

/* convex 8x upsampling: flow fp32 NHWC [B,h,w,2]; mask NHWC [B,h,w,576] (channel k*64 + i*8 + j, already
 * scaled by 0.25); out fp32 planar [B,2,8h,8w]. */
int pp_convex_upsample(const float* flow, const void* mask, int mask_cstride, int mask_dtype, float* out, int B,
                       int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse window attention (model/modules/sparse_transformer.py:158-281)
 * ---------------------------------------------------------------------------------------------- */
/* Host helper: window index tables for a padded Hp x Wp token grid with (wh, ww) windows:
 * own[nW*wh*ww] and rolled[nW*n_rolled] flat positions (circular torch.roll semantics, :140-153,181-200).
 * Returns n_rolled (148 for 5x9) or negative error; call with NULL outputs to size. */
int pp_window_tables(int Hp, int Wp, int wh, int ww, int32_t* own, int32_t* rolled, int capacity_rolled);

/* wmask[b, w] = sum over local frames of max over the window of mask (:227-229); mask [B,Lt,Hp,Wp] dtype. */
int pp_window_mask(const void* mask, float* wmask, int B, int Lt, int Hp, int Wp, int wh, int ww, int dtype,
                   void* stream);

typedef struct {
  int32_t dtype;               /* q/k/v/pooled/out dtype                                           */
  int32_t B, T, Hp, Wp, C;     /* padded token grid, C = heads*head_dim                            */
  int32_t heads;               /* head_dim = C / heads must be 128                                  */
  int32_t wh, ww;              /* window (5, 9)                                                    */
  int32_t n_rolled;            /* 148                                                              */
  int32_t P;                   /* pooled tokens per frame                                          */
  int32_t n_tind;              /* number of key frames for masked windows                          */
  const void* q; const void* k; const void* v;   /* [B,T,Hp,Wp,C]                                  */
  int32_t qkv_cstride;         /* elements per token in q/k/v (C, or 3C for a fused qkv buffer)     */
  const void* pk; const void* pv;                /* [B,T,P,C] pooled key/value, pkv_cstride per token */
  int32_t pkv_cstride;
  const int32_t* own;          /* device [nW, wh*ww]                                               */
  const int32_t* rolled;       /* device [nW, n_rolled]                                            */
  const int32_t* tind;         /* device [n_tind] key frame indices                                */
  const float* wmask;          /* device [B, nW]; > 0 => masked window                             */
  void* out;                   /* [B,T,Hp,Wp,C] (cstride = C), every position written               */
  int32_t impl;                /* 0 = auto (MFMA for fp16), 1 = force the scalar reference kernel, 6 = grid-mapped MFMA
                                  launch even when `work` is given                                    */
  int32_t work_ints;           /* capacity of `work` in int32 elements (>= 1 + B*nW)                */
  void* work;                  /* optional device scratch: compacted list of masked windows for the persistent,
                                  load-balanced launch of the masked-window kernel; NULL = grid-mapped launch */
  int32_t out_h, out_w;        /* both 0: `out` is the padded grid [B,T,Hp,Wp,C]; both > 0: `out` is the COMPACT token grid
                                  [B,T,out_h,out_w,C] (out_h <= Hp, out_w <= Wp) -- the crop of sparse_transformer.py:276-277 happens in
                                  the store, padding tokens are not written                                              */
} pp_attn_args_t;

int pp_sparse_window_attention(const pp_attn_args_t* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Token <-> feature-map ops (model/modules/sparse_transformer.py:49-101) and small dense helpers
 * ---------------------------------------------------------------------------------------------- */
/* F.fold(kernel 7, stride 3, pad 3) of tokens [BT, fh*fw, 49*C] into NHWC [BT,H,W,C].  The token features are in
 * TAP-MAJOR order (ky*7 + kx)*C + c -- the caller permutes the rows of the producing Linear (fc1 / SoftComp
 * embedding; the reference order is c*49 + ky*7 + kx) so that the fold reads 16-byte channel runs.  C % 8 == 0.
 * If normalize != 0 the sum is divided by the overlap count (FusionFeedForward :82-95) and `act` is applied (GELU
 * for the FFN, since unfold(gelu(x)) == gelu(unfold(x)) under zero padding). */
int pp_fold_tokens(const void* tokens, void* out, int BT, int fh, int fw, int C, int H, int W, int normalize,
                   int act, int dtype, void* stream);

/* LayerNorm over the last dim (C multiple of 64, <= 1024): in/out [rows, C]. */
int pp_layernorm(const void* in, const float* gamma, const float* beta, void* out, int64_t rows, int C, float eps,
                 int dtype, void* stream);
/* The same over a token grid in [N, gh, gw, C], written into the top-left corner of a PADDED grid out [N, Hp, Wp, C] (Hp >= gh, Wp >=
 * gw): the zero padding AFTER LayerNorm of sparse_transformer.py:169-171 without a copy -- the padding tokens of `out` are not
 * written (the caller zero-fills the buffer once). */
int pp_layernorm_grid(const void* in, const float* gamma, const float* beta, void* out, int N, int gh, int gw, int Hp, int Wp, int C,
                      float eps, int dtype, void* stream);

/* depthwise kxk stride-k conv ("pool_layer", sparse_transformer.py:136,209): in NHWC [N,H,W,C] -> [N,H/k,W/k,C];
 * weight fp32 [C,k,k], bias fp32 [C]. */
int pp_depthwise_pool(const void* in, const float* weight, const float* bias, void* out, int N, int H, int W, int C,
                      int k, int dtype, void* stream);

/* InstanceNorm2d (no affine, eps; RAFT/extractor.py:126-133 fnet norm layers) over NHWC [N,H,W,C] + optional ReLU.
 * Deterministic two-stage reduction (no atomics).  stats_ws: caller-owned fp32 workspace of
 * pp_instance_norm_workspace_floats(N,H,W,C) elements, 16-byte aligned.  C % 8 == 0, C <= 256. */
int64_t pp_instance_norm_workspace_floats(int N, int H, int W, int C);
int pp_instance_norm(const void* in, void* out, float* stats_ws, int N, int H, int W, int C, float eps, int relu,
                     int dtype, void* stream);
/* The same for the split-plane ("f16x3") RAFT feature encoder, with the tail of a ResidualBlock fused (RAFT/extractor.py:44-57):
 * in = fp32 NHWC [N,H,W,C] (a pp_conv2d output with out_dtype PP_F32), out = split-plane fp16 NHWC [N,H,W,2C] (PP_F16S, lo plane at C),
 * out = relu2(relu(IN(in)) + residual); residual = split-plane window (lo plane at res_cstride / 2) or NULL. */
int pp_instance_norm_split(const float* in, void* out, float* stats_ws, int N, int H, int W, int C, float eps, int relu,
                           const void* residual, int res_cstride, int res_choff, int relu2, void* stream);

/* bilinear x2 upsampling, align_corners=True, NHWC [N,H,W,C] -> [N,2H,2W,C] (deconv blocks). */
int pp_upsample2x(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);

/* offset/mask head activation of the deformable alignment (model/propainter.py:58-65,
 * model/recurrent_flow_completion.py:32-40): in place on NHWC [N,H,W,432]: channels [0,288) ->
 * mag*tanh(v) (+ flow_y on even / flow_x on odd channels when flow != NULL), [288,432) -> sigmoid. */
int pp_dcn_offset_mask_act(void* offmask, int cstride, const void* flow, int fl_cstride, int fl_choff, float mag,
                           int64_t npix, int dtype, void* stream);

/* SepConvGRU gating (RAFT/update.py:45-60):  mode 0: out = r*h where r = zr[:, C:2C] (already sigmoid-ed);
 * mode 1: out = (1-z)*h + z*q where z = zr[:, 0:C].  All NHWC with explicit cstride/choff. */
int pp_gru_gate(const void* zr, int zr_cstride, const void* h, int h_cstride, int h_choff, const void* q,
                int q_cstride, void* out, int out_cstride, int out_choff, int64_t npix, int C, int mode, int dtype,
                void* stream);

/* layout packers: planar NCHW [N,C,H,W] (src dtype) <-> NHWC channel window (dst dtype).  pp_nchw_to_nhwc also takes
 * out_dtype PP_F16S (fp32 input): split-plane output, lo plane at out_cstride / 2. */
int pp_nchw_to_nhwc(const void* in, int in_dtype, void* out, int out_dtype, int out_cstride, int out_choff, int N,
                    int C, int H, int W, float scale, void* stream);

/* Encoder input packer: up to three planar sources [N,c_i,H,W] of one dtype (c0 + c1 + c2 <= 8; in1 / in2 may be NULL with c = 0) ->
 * NHWC [N,H,W,8], missing channels zero, one launch writing whole 16-byte rows.  Replaces the reference's
 * torch.cat([frames, masks_in, masks_updated], dim=2) in front of the Encoder (model/propainter.py:334-336) and the
 * cat(masked_flows, masks) in front of the flow-completion encoder (model/recurrent_flow_completion.py:279).
 * dtype PP_F16S: fp32 sources -> split-plane fp16 rows [N,H,W,8 hi | 8 lo] (pp_nchw_to_nhwc's split arithmetic, whole 32-byte rows):
 * the frames entering RAFT's encoders (RAFT/raft.py:96-99). */
int pp_pack_nhwc8(const void* in0, int c0, const void* in1, int c1, const void* in2, int c2, void* out, int N, int H, int W,
                  int dtype, void* stream);
int pp_nhwc_to_nchw(const void* in, int in_dtype, int in_cstride, int in_choff, void* out, int out_dtype, int N,
                    int C, int H, int W, int act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Reference-layout entry points (csrc/ref_layout_ops.hip): the operators of the path as the reference calls them -- planar
 * NCHW device tensors of `dtype`, weights exactly as stored in the reference's state dict (device, `dtype`), biases fp32 --
 * for binders that do not carry the engine's NHWC / packed-weight conventions.  Each call packs its operands into the
 * caller's workspace (size: the *_workspace_size twin; 256-byte aligned device memory), runs the engine kernels and
 * unpacks; nothing is allocated or retained.  One small host->device copy (the K table) per call: not graph-capturable.
 * ---------------------------------------------------------------------------------------------- */
/* torchvision.ops.deform_conv2d(x, offset, weight, bias, stride=1, padding=1, dilation=1, mask) for 3x3 weights and 16
 * offset groups (model/propainter.py:67-69, model/recurrent_flow_completion.py:42-44): x [N,Cin,H,W] (Cin % 128 == 0),
 * offset [N,288,H,W] ((dy, dx) interleaved per (group, tap)), mask [N,144,H,W], weight [Cout,Cin,3,3] -> out [N,Cout,H,W]. */
int64_t pp_deform_conv2d_workspace_size(int N, int Cin, int H, int W, int Cout, int dtype);
int pp_deform_conv2d(const void* x, const void* offset, const void* mask, const void* weight, const float* bias, void* out,
                     int N, int Cin, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* CorrBlock.__init__ (RAFT/corr.py:13-27,52-60): fmap1, fmap2 [B,256,h,w] -> fp32 pyramid lvl[l] = [B*h*w, h>>l, w>>l]
 * (all-pairs dot products / sqrt(256), then three 2x2 average poolings). */
int64_t pp_corr_pyramid_workspace_size(int B, int h, int w, int dtype);
int pp_corr_pyramid(const void* fmap1, const void* fmap2, float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int h, int w,
                    int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* SoftSplit.forward (model/modules/sparse_transformer.py:19-31): x [BT,C,H,W], embedding weight [hidden, C*49] (feature index
 * c*49 + ky*7 + kx), bias fp32 [hidden] -> tokens [BT, fh*fw, hidden], f = (n + 6 - 7) / 3 + 1. */
int64_t pp_softsplit_workspace_size(int BT, int C, int H, int W, int hidden, int dtype);
int pp_softsplit(const void* x, const void* weight, const float* bias, void* tokens, int BT, int C, int H, int W, int hidden,
                 int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* SoftComp.forward (:49-61): tokens [BT, fh*fw, hidden], embedding weight [C*49, hidden] + bias fp32 [C*49], bias_conv weight
 * [C,C,3,3] + bias fp32 [C] -> out [BT,C,H,W] (Linear -> F.fold(7, 3, 3) -> 3x3 convolution).  C, hidden multiples of 8. */
int64_t pp_softcomp_workspace_size(int BT, int C, int H, int W, int hidden, int dtype);
int pp_softcomp(const void* tokens, const void* emb_weight, const float* emb_bias, const void* conv_weight, const float* conv_bias,
                void* out, int BT, int C, int H, int W, int hidden, int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* FusionFeedForward's fold -> normalise -> unfold (:82-98) on features [BT, fh*fw, C*49] in the reference order (C = 40):
 * out = unfold(fold(in) / fold(ones)); in != out. */
int pp_ffn_fold_unfold(const void* in, void* out, int BT, int C, int H, int W, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Diagnostics (tuning tools only: tools/kbench.cpp, tools/bench_attn.py).  Read and clear the in-kernel phase counters of
 * the PROF kernel builds (cycles summed over sampled waves); all zero unless a PP_DIAG=1 build ran a PROF variant.
 * ---------------------------------------------------------------------------------------------- */
int pp_debug_conv_prof(unsigned long long* out12);
int pp_debug_attn_prof(unsigned long long* out8);

#ifdef __cplusplus
}
#endif
#endif /* PROPAINTER_HIP_H */
