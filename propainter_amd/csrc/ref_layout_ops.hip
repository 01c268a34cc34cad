// Reference-layout entry points (SURVEY.md section 8b "minimum set"): the operators of the hot path as a foreign binder
// sees them in the reference -- planar NCHW tensors, weights exactly as they sit in the reference's state dict -- built
// on the engine's kernels.  Every function takes a caller-owned device workspace (sized by its *_workspace_size twin),
// allocates nothing, keeps no pointers, and enqueues everything on `stream`:
//     pack layout / weights into the workspace -> pp_conv2d & friends (NHWC, packed weights) -> unpack.
// The Python engine does not go through these (it keeps activations NHWC and weights packed across calls); they exist so
// that deformable convolution, the correlation pyramid, SoftSplit / SoftComp and the FFN fold-unfold can be called
// through the C-ABI without porting propainter_amd/conv.py.  One host->device copy of the (< 64 KB) K table per call.
#include "conv_params.h"

#include <string.h>
#include <vector>

using namespace pp;

namespace pp {

// weight [cout, cin_g, kh, kw] (reference layout, T) -> packed [cout_pad][K] (T) following the device K table
template <typename T>
__global__ void pack_weight_kernel(const T* __restrict__ w, T* __restrict__ out, const int4* __restrict__ kt, int cout, int cout_pad,
                                   int kchunks, int cin, int kh, int kw, int c0, int c1, int c2, int c3) {
  const long long total = (long long)cout_pad * kchunks * 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    const int kc = (int)((i >> 3) % kchunks);
    const int co = (int)(i / ((long long)kchunks * 8));
    const int4 e = kt[kc];
    const int src = e.z & 0xff, tap = (e.z >> 16) & 0xff, ch = e.w + j;
    const int creal = src == 0 ? c0 : src == 1 ? c1 : src == 2 ? c2 : c3;
    const int base = src == 0 ? 0 : src == 1 ? c0 : src == 2 ? c0 + c1 : c0 + c1 + c2;
    T v = (T)0.f;
    if (co < cout && src != 255 && ch < creal) v = w[(((long long)co * cin + base + ch) * kh + tap / kw) * kw + tap % kw];
    out[i] = v;
  }
}

// [N,288,H,W] offsets + [N,144,H,W] masks (planar) -> NHWC [N,H,W,432]
template <typename T>
__global__ void offmask_to_nhwc_kernel(const T* __restrict__ off, const T* __restrict__ msk, T* __restrict__ out, int N, int HW) {
  const long long total = (long long)N * HW * 432;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 432);
    const long long px = i / 432;
    const long long n = px / HW, hw = px % HW;
    out[i] = c < 288 ? off[(n * 288 + c) * HW + hw] : msk[(n * 144 + (c - 288)) * HW + hw];
  }
}

// FusionFeedForward's fold -> normalise -> unfold on reference-order features (index c*49 + ky*7 + kx), as one gather:
// out[bt, token(ty,tx), c*49 + ky*7 + kx] = folded[bt, c, ty*3 - 3 + ky, tx*3 - 3 + kx] / count (0 outside the map).
template <typename T>
__global__ void fold_unfold_kernel(const T* __restrict__ in, T* __restrict__ out, int BT, int fh, int fw, int C, int H, int W) {
  const long long total = (long long)BT * fh * fw * C * 49;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % 49);
    const int c = (int)((i / 49) % C);
    const long long tok = i / (49LL * C);
    const int tx = (int)(tok % fw), ty = (int)((tok / fw) % fh);
    const long long bt = tok / ((long long)fw * fh);
    const int y = ty * 3 - 3 + k / 7, x = tx * 3 - 3 + k % 7;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      float s = 0.f;
      int cnt = 0;
      for (int ky = 0; ky < 7; ++ky) {
        const int yy = y + 3 - ky;
        if (yy < 0 || yy % 3 != 0 || yy / 3 >= fh) continue;
        for (int kx = 0; kx < 7; ++kx) {
          const int xx = x + 3 - kx;
          if (xx < 0 || xx % 3 != 0 || xx / 3 >= fw) continue;
          s += to_f32(in[((bt * fh + yy / 3) * fw + xx / 3) * ((long long)C * 49) + c * 49 + ky * 7 + kx]);
          ++cnt;
        }
      }
      v = cnt > 0 ? s / (float)cnt : 0.f;
    }
    out[i] = from_f32<T>(v);
  }
}

// tokens [BT, n, C*49] in reference order (c*49 + k) -> tap-major (k*C + c), the order pp_fold_tokens reads
template <typename T>
__global__ void tokens_to_tap_major_kernel(const T* __restrict__ in, T* __restrict__ out, long long ntok, int C) {
  const long long total = ntok * C * 49;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int k = (int)((i / C) % 49);
    const long long t = i / (49LL * C);
    out[i] = in[t * C * 49 + c * 49 + k];
  }
}

static inline int grid1d(long long total) {
  long long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}
static inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }
static inline int esize(int dtype) { return dtype == PP_F16 ? 2 : 4; }
static inline int pad8i(int c) { return (c + 7) / 8 * 8; }

// carves the workspace
struct Carver {
  char* base;
  int64_t off = 0, cap;
  Carver(void* b, int64_t c) : base((char*)b), cap(c) {}
  void* take(int64_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  }
  bool ok() const { return off <= cap; }
};

// builds the K table of a dense kh x kw window (or the deformable order) on the host, copies it into the workspace and
// packs `weight` next to it; fills the conv args' table / weight fields.
static int prepare_layer(pp_conv_args_t& a, Carver& ws, const void* weight, int dtype, int cout, int kh, int kw, int dil, int nsrc,
                         const int* creal, int dcn_groups, hipStream_t st, bool dry) {
  std::vector<int32_t> dy, dx;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) { dy.push_back(ky * dil); dx.push_back(kx * dil); }
  int32_t cpad[PP_CONV_MAX_SRC] = {8, 8, 8, 8};
  bool u32 = true, u64 = true;
  int cin = 0;
  for (int s = 0; s < nsrc; ++s) {
    cpad[s] = pad8i(creal[s]);
    cin += creal[s];
    u32 = u32 && cpad[s] % 32 == 0;
    u64 = u64 && cpad[s] % 64 == 0;
  }
  const int kchunks = pp_conv_build_ktable(kh * kw, dy.data(), dx.data(), nsrc, cpad, dcn_groups, nullptr, 0);
  if (kchunks < 0) return kchunks;
  const int cout_pad = (cout + 15) / 16 * 16;
  void* d_kt = ws.take((int64_t)(kchunks + 1) * 16);
  void* d_w = ws.take((int64_t)cout_pad * kchunks * 8 * esize(dtype));
  a.kchunks = kchunks; a.cout_pad = cout_pad; a.cout_g = cout; a.groups = 1; a.nsrc = nsrc;
  a.weight_gstride = (int64_t)cout_pad * kchunks * 8;
  a.ktable_uniform = dcn_groups ? 0 : ((u32 ? 4 : 0) | (u64 ? 8 : 0));
  a.tap_h = (dil == 1 && !dcn_groups) ? kh : 0; a.tap_w = (dil == 1 && !dcn_groups) ? kw : 0;
  a.ktable = (const int32_t*)d_kt; a.weight = d_w;
  if (dry || !ws.ok()) return 0;
  std::vector<int32_t> kt((size_t)(kchunks + 1) * 4);
  pp_conv_build_ktable(kh * kw, dy.data(), dx.data(), nsrc, cpad, dcn_groups, kt.data(), kchunks + 1);
  hipError_t e = hipMemcpyAsync(d_kt, kt.data(), kt.size() * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);          // kt is a stack-lifetime host buffer
  if (e != hipSuccess) { set_error("ref-layout op: K-table upload: %s", hipGetErrorString(e)); return (int)e; }
  const int c0 = creal[0], c1 = nsrc > 1 ? creal[1] : 0, c2 = nsrc > 2 ? creal[2] : 0, c3 = nsrc > 3 ? creal[3] : 0;
  const long long total = (long long)cout_pad * kchunks * 8;
  if (dtype == PP_F16)
    hipLaunchKernelGGL((pack_weight_kernel<_Float16>), dim3(grid1d(total)), dim3(256), 0, st, (const _Float16*)weight, (_Float16*)d_w,
                       (const int4*)d_kt, cout, cout_pad, kchunks, cin, kh, kw, c0, c1, c2, c3);
  else
    hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(grid1d(total)), dim3(256), 0, st, (const float*)weight, (float*)d_w,
                       (const int4*)d_kt, cout, cout_pad, kchunks, cin, kh, kw, c0, c1, c2, c3);
  return launch_status("ref-layout op: weight packing");
}

static void conv_defaults(pp_conv_args_t& a, int dtype, int N, int H, int W, int OH, int OW, int stride, int pad) {
  ::memset((void*)&a, 0, sizeof(a));
  a.dtype = dtype; a.out_dtype = dtype; a.N = N; a.H = H; a.W = W; a.OH = OH; a.OW = OW;
  a.stride_h = a.stride_w = stride; a.pad_h = a.pad_w = pad; a.out_scale = 1.f;
}

#define RL_CHECK(rc) do { const int rc_ = (rc); if (rc_ != 0) return rc_; } while (0)

}  // namespace pp

// ------------------------------------------------------------------------------------------------------------------
// torchvision.ops.deform_conv2d(x, offset, weight, bias, stride 1, padding 1, dilation 1, mask), 3x3, 16 offset groups
// (model/propainter.py:67-69, model/recurrent_flow_completion.py:42-44)
// ------------------------------------------------------------------------------------------------------------------
static int deform_plan(pp_conv_args_t& a, Carver& ws, const void* weight, int N, int Cin, int H, int W, int Cout, int dtype,
                       hipStream_t st, bool dry, void** x_nhwc, void** om, void** out_nhwc) {
  conv_defaults(a, dtype, N, H, W, H, W, 1, 1);
  const int64_t npix = (int64_t)N * H * W;
  *x_nhwc = ws.take(npix * pad8i(Cin) * esize(dtype));
  *om = ws.take(npix * 432 * esize(dtype));
  *out_nhwc = ws.take(npix * pad8i(Cout) * esize(dtype));
  const int creal[1] = {Cin};
  return prepare_layer(a, ws, weight, dtype, Cout, 3, 3, 1, 1, creal, 16, st, dry);
}

extern "C" int64_t pp_deform_conv2d_workspace_size(int N, int Cin, int H, int W, int Cout, int dtype) {
  if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (dtype != PP_F32 && dtype != PP_F16) || Cin % 128 != 0) return PP_ERR_ARG;
  pp_conv_args_t a;
  Carver ws(nullptr, 0);
  void *x, *om, *o;
  const int rc = deform_plan(a, ws, nullptr, N, Cin, H, W, Cout, dtype, nullptr, true, &x, &om, &o);
  return rc < 0 ? rc : ws.off;
}

extern "C" int pp_deform_conv2d(const void* x, const void* offset, const void* mask, const void* weight, const float* bias, void* out,
                                int N, int Cin, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                                void* stream) {
  PP_REQUIRE(x && offset && mask && weight && out && workspace, PP_ERR_ARG, "pp_deform_conv2d: null pointer");
  PP_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0 && Cin % 128 == 0, PP_ERR_ARG,
             "pp_deform_conv2d: Cin %d must be a multiple of 128 (16 offset groups of 8n channels)", Cin);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_deform_conv2d: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  pp_conv_args_t a;
  Carver ws(workspace, workspace_bytes);
  void *xn, *om, *on;
  {
    Carver probe(nullptr, 0);
    pp_conv_args_t t;
    void *p0, *p1, *p2;
    deform_plan(t, probe, nullptr, N, Cin, H, W, Cout, dtype, nullptr, true, &p0, &p1, &p2);
    PP_REQUIRE(probe.off <= workspace_bytes, PP_ERR_WORKSPACE, "pp_deform_conv2d: workspace %lld bytes, need %lld",
               (long long)workspace_bytes, (long long)probe.off);
  }
  RL_CHECK(deform_plan(a, ws, weight, N, Cin, H, W, Cout, dtype, st, false, &xn, &om, &on));
  const int cin_p = pad8i(Cin), cout_p = pad8i(Cout);
  RL_CHECK(pp_nchw_to_nhwc(x, dtype, xn, dtype, cin_p, 0, N, Cin, H, W, 1.f, stream));
  const long long tot = (long long)N * H * W * 432;
  if (dtype == PP_F16)
    hipLaunchKernelGGL((offmask_to_nhwc_kernel<_Float16>), dim3(grid1d(tot)), dim3(256), 0, st, (const _Float16*)offset,
                       (const _Float16*)mask, (_Float16*)om, N, H * W);
  else
    hipLaunchKernelGGL((offmask_to_nhwc_kernel<float>), dim3(grid1d(tot)), dim3(256), 0, st, (const float*)offset, (const float*)mask,
                       (float*)om, N, H * W);
  RL_CHECK(launch_status("pp_deform_conv2d: offset/mask layout"));
  if (cout_p != Cout) (void)hipMemsetAsync(on, 0, (size_t)N * H * W * cout_p * esize(dtype), st);
  a.src[0].ptr = xn; a.src[0].cstride = cin_p;
  a.bias = bias; a.out = on; a.out_cstride = cout_p; a.out_cgroup = Cout;
  a.dcn_offmask = om; a.dcn_cstride = 432; a.dcn_mask_off = 288;
  RL_CHECK(pp_conv2d(&a, stream));
  return pp_nhwc_to_nchw(on, dtype, cout_p, 0, out, dtype, N, Cout, H, W, PP_ACT_NONE, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// CorrBlock.__init__ (RAFT/corr.py:13-27,52-60): all-pairs volume / sqrt(256) and its 3 average-pooled levels, fp32
// ------------------------------------------------------------------------------------------------------------------
extern "C" int64_t pp_corr_pyramid_workspace_size(int B, int h, int w, int dtype) {
  if (B <= 0 || h < 16 || w < 16 || (dtype != PP_F32 && dtype != PP_F16)) return PP_ERR_ARG;
  return 2 * align256((int64_t)B * h * w * 256 * esize(dtype)) + align256(33 * 16) + 256;      // (K table: 32 chunks + the zero page)
}

extern "C" int pp_corr_pyramid(const void* fmap1, const void* fmap2, float* lvl0, float* lvl1, float* lvl2, float* lvl3, int B, int h,
                               int w, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  PP_REQUIRE(fmap1 && fmap2 && lvl0 && lvl1 && lvl2 && lvl3 && workspace, PP_ERR_ARG, "pp_corr_pyramid: null pointer");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_corr_pyramid: dtype %d", dtype);
  PP_REQUIRE(B > 0 && (h >> 3) >= 2 && (w >> 3) >= 2, PP_ERR_ARG, "pp_corr_pyramid: maps of %dx%d are too small (level 3 needs >= 2x2)", h, w);
  PP_REQUIRE(workspace_bytes >= pp_corr_pyramid_workspace_size(B, h, w, dtype), PP_ERR_WORKSPACE, "pp_corr_pyramid: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Carver ws(workspace, workspace_bytes);
  const int64_t n8 = (int64_t)h * w;
  void* a_ = ws.take(B * n8 * 256 * esize(dtype));
  void* b_ = ws.take(B * n8 * 256 * esize(dtype));
  void* d_kt = ws.take(33 * 16);
  RL_CHECK(pp_nchw_to_nhwc(fmap1, dtype, a_, dtype, 256, 0, B, 256, h, w, 1.f, stream));
  RL_CHECK(pp_nchw_to_nhwc(fmap2, dtype, b_, dtype, 256, 0, B, 256, h, w, 1.f, stream));
  int32_t kt[33 * 4], zero = 0, c256 = 256;
  const int kchunks = pp_conv_build_ktable(1, &zero, &zero, 1, &c256, 0, kt, 33);
  PP_REQUIRE(kchunks == 32, PP_ERR_ARG, "pp_corr_pyramid: unexpected K table size %d", kchunks);
  hipError_t e = hipMemcpyAsync(d_kt, kt, 33 * 16, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  PP_REQUIRE(e == hipSuccess, (int)e, "pp_corr_pyramid: K-table upload: %s", hipGetErrorString(e));
  pp_conv_args_t g;                       // batched GEMM out[b, m, n] = a[b, m, :] . b[b, n, :] / 16 (the engine's formulation)
  ::memset((void*)&g, 0, sizeof(g));
  g.dtype = dtype; g.N = 1; g.H = 1; g.W = (int)n8; g.OH = 1; g.OW = (int)n8; g.stride_h = g.stride_w = 1;
  g.groups = B; g.cout_g = (int)n8; g.cout_pad = (int)n8; g.kchunks = 32; g.nsrc = 1;
  g.src[0].ptr = a_; g.src[0].cstride = 256;
  g.ktable = (const int32_t*)d_kt; g.weight = b_; g.weight_gstride = n8 * 256;
  g.out_scale = 1.f / 16.f; g.out_dtype = PP_F32; g.ktable_uniform = 12;
  g.out = lvl0; g.out_cstride = (int)n8; g.src_gstride = n8 * 256; g.out_gstride = n8 * n8;
  RL_CHECK(pp_conv2d(&g, stream));
  RL_CHECK(pp_corr_avgpool(lvl0, lvl1, B * n8, h, w, stream));
  RL_CHECK(pp_corr_avgpool(lvl1, lvl2, B * n8, h / 2, w / 2, stream));
  return pp_corr_avgpool(lvl2, lvl3, B * n8, h / 4, w / 4, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// SoftSplit (sparse_transformer.py:19-31): unfold(7, stride 3, pad 3) + Linear(C*49 -> hidden) == one 7x7 / stride-3
// convolution with the Linear's weight viewed as [hidden, C, 7, 7]
// ------------------------------------------------------------------------------------------------------------------
static inline int token_dim(int n) { return (n + 2 * 3 - 6 - 1) / 3 + 1; }

extern "C" int64_t pp_softsplit_workspace_size(int BT, int C, int H, int W, int hidden, int dtype) {
  if (BT <= 0 || C <= 0 || H < 7 || W < 7 || hidden <= 0 || (dtype != PP_F32 && dtype != PP_F16)) return PP_ERR_ARG;
  const int64_t kchunks = (int64_t)(49 * pad8i(C) / 8 + 7) / 8 * 8;
  return align256((int64_t)BT * H * W * pad8i(C) * esize(dtype)) + align256((kchunks + 1) * 16) +
         align256((int64_t)((hidden + 15) / 16 * 16) * kchunks * 8 * esize(dtype)) + 256;
}

extern "C" int pp_softsplit(const void* x, const void* weight, const float* bias, void* tokens, int BT, int C, int H, int W, int hidden,
                            int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  PP_REQUIRE(x && weight && tokens && workspace, PP_ERR_ARG, "pp_softsplit: null pointer");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_softsplit: dtype %d", dtype);
  PP_REQUIRE(hidden % 8 == 0, PP_ERR_ARG, "pp_softsplit: hidden %d must be a multiple of 8", hidden);
  const int64_t need = pp_softsplit_workspace_size(BT, C, H, W, hidden, dtype);
  PP_REQUIRE(need > 0 && workspace_bytes >= need, PP_ERR_WORKSPACE, "pp_softsplit: workspace %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  Carver ws(workspace, workspace_bytes);
  const int fh = token_dim(H), fw = token_dim(W), cp = pad8i(C);
  void* xn = ws.take((int64_t)BT * H * W * cp * esize(dtype));
  pp_conv_args_t a;
  conv_defaults(a, dtype, BT, H, W, fh, fw, 3, 3);
  const int creal[1] = {C};
  RL_CHECK(prepare_layer(a, ws, weight, dtype, hidden, 7, 7, 1, 1, creal, 0, st, false));
  a.tap_h = a.tap_w = 0;                                    // strided: not a halo-tile layer
  if (cp != C) (void)hipMemsetAsync(xn, 0, (size_t)BT * H * W * cp * esize(dtype), st);
  RL_CHECK(pp_nchw_to_nhwc(x, dtype, xn, dtype, cp, 0, BT, C, H, W, 1.f, stream));
  a.src[0].ptr = xn; a.src[0].cstride = cp;
  a.bias = bias; a.out = tokens; a.out_cstride = hidden; a.out_cgroup = hidden;
  return pp_conv2d(&a, stream);                              // NHWC [BT, fh, fw, hidden] == tokens [BT, fh*fw, hidden]
}

// ------------------------------------------------------------------------------------------------------------------
// FusionFeedForward's fold / normalise / unfold (sparse_transformer.py:82-98) on [BT, n, C*49] features (C = 40)
// ------------------------------------------------------------------------------------------------------------------
extern "C" int pp_ffn_fold_unfold(const void* in, void* out, int BT, int C, int H, int W, int dtype, void* stream) {
  PP_REQUIRE(in && out && in != out && BT > 0 && C > 0 && H >= 7 && W >= 7, PP_ERR_ARG, "pp_ffn_fold_unfold: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_ffn_fold_unfold: dtype %d", dtype);
  const int fh = token_dim(H), fw = token_dim(W);
  const long long total = (long long)BT * fh * fw * C * 49;
  if (dtype == PP_F16)
    hipLaunchKernelGGL((fold_unfold_kernel<_Float16>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in,
                       (_Float16*)out, BT, fh, fw, C, H, W);
  else
    hipLaunchKernelGGL((fold_unfold_kernel<float>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, (const float*)in, (float*)out,
                       BT, fh, fw, C, H, W);
  return launch_status("pp_ffn_fold_unfold");
}

// ------------------------------------------------------------------------------------------------------------------
// SoftComp (sparse_transformer.py:49-61): Linear(hidden -> C*49) -> F.fold(7, 3, 3) -> 3x3 bias_conv
// ------------------------------------------------------------------------------------------------------------------
extern "C" int64_t pp_softcomp_workspace_size(int BT, int C, int H, int W, int hidden, int dtype) {
  if (BT <= 0 || C <= 0 || C % 8 != 0 || H < 7 || W < 7 || hidden <= 0 || hidden % 8 != 0 || (dtype != PP_F32 && dtype != PP_F16)) return PP_ERR_ARG;
  const int64_t n = (int64_t)BT * token_dim(H) * token_dim(W), es = esize(dtype);
  const int64_t k1 = (hidden / 8 + 7) / 8 * 8, k2 = (9 * C / 8 + 7) / 8 * 8;
  return 2 * align256(n * C * 49 * es) + 2 * align256((int64_t)BT * H * W * C * es) + align256((k1 + 1) * 16) +
         align256((int64_t)((C * 49 + 15) / 16 * 16) * k1 * 8 * es) + align256((k2 + 1) * 16) + align256((int64_t)((C + 15) / 16 * 16) * k2 * 8 * es) + 256;
}

extern "C" int pp_softcomp(const void* tokens, const void* emb_weight, const float* emb_bias, const void* conv_weight,
                           const float* conv_bias, void* out, int BT, int C, int H, int W, int hidden, int dtype, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  PP_REQUIRE(tokens && emb_weight && conv_weight && out && workspace, PP_ERR_ARG, "pp_softcomp: null pointer");
  const int64_t need = pp_softcomp_workspace_size(BT, C, H, W, hidden, dtype);
  PP_REQUIRE(need > 0, PP_ERR_ARG, "pp_softcomp: bad extents / dtype (C and hidden multiples of 8)");
  PP_REQUIRE(workspace_bytes >= need, PP_ERR_WORKSPACE, "pp_softcomp: workspace %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  Carver ws(workspace, workspace_bytes);
  const int fh = token_dim(H), fw = token_dim(W);
  const int64_t n = (int64_t)BT * fh * fw, es = esize(dtype);
  void* emb = ws.take(n * C * 49 * es);
  void* emb_t = ws.take(n * C * 49 * es);
  void* folded = ws.take((int64_t)BT * H * W * C * es);
  void* conv_o = ws.take((int64_t)BT * H * W * C * es);
  // 1. Linear: tokens [n, hidden] x emb_weight [C*49, hidden]^T (+ bias), reference feature order c*49 + k
  pp_conv_args_t a;
  conv_defaults(a, dtype, 1, 1, (int)n, 1, (int)n, 1, 0);
  const int creal1[1] = {hidden};
  RL_CHECK(prepare_layer(a, ws, emb_weight, dtype, C * 49, 1, 1, 1, 1, creal1, 0, st, false));
  a.src[0].ptr = tokens; a.src[0].cstride = hidden;
  a.bias = emb_bias; a.out = emb; a.out_cstride = C * 49; a.out_cgroup = C * 49;
  RL_CHECK(pp_conv2d(&a, stream));
  // 2. fold (the fold kernel reads tap-major features)
  const long long tot = n * C * 49;
  if (dtype == PP_F16)
    hipLaunchKernelGGL((tokens_to_tap_major_kernel<_Float16>), dim3(grid1d(tot)), dim3(256), 0, st, (const _Float16*)emb, (_Float16*)emb_t, n, C);
  else
    hipLaunchKernelGGL((tokens_to_tap_major_kernel<float>), dim3(grid1d(tot)), dim3(256), 0, st, (const float*)emb, (float*)emb_t, n, C);
  RL_CHECK(launch_status("pp_softcomp: feature order"));
  RL_CHECK(pp_fold_tokens(emb_t, folded, BT, fh, fw, C, H, W, 0, PP_ACT_NONE, dtype, stream));
  // 3. bias_conv 3x3
  pp_conv_args_t b;
  conv_defaults(b, dtype, BT, H, W, H, W, 1, 1);
  const int creal2[1] = {C};
  RL_CHECK(prepare_layer(b, ws, conv_weight, dtype, C, 3, 3, 1, 1, creal2, 0, st, false));
  b.src[0].ptr = folded; b.src[0].cstride = C;
  b.bias = conv_bias; b.out = conv_o; b.out_cstride = C; b.out_cgroup = C;
  RL_CHECK(pp_conv2d(&b, stream));
  return pp_nhwc_to_nchw(conv_o, dtype, C, 0, out, dtype, BT, C, H, W, PP_ACT_NONE, stream);
}
