// Streaming 3x3 convolution for layers with at most 4 output channels (the RAFT flow head 256 -> 2, the RGB decoder head 64 -> 3, the
// flow-completion head 32 -> 2: RAFT/update.py:7-9, model/propainter.py:271-273, model/recurrent_flow_completion.py:255).
//
// Through the implicit-GEMM kernels these layers run 2-3 useful couts through a 16-cout MFMA tile with 4-6 MFMAs per barrier: the
// split-plane flow head took 330 us per 504 000 pixels where reading its input once takes 115 us.  They are HBM-bound dot products, so
// here they are computed on the VALU: a block owns a 16 x 16 tile of output pixels, stages the 18 x 18 halo patch of one K block
// (128 bytes per pixel: 64 channels, or 32 channels of BOTH planes of a split-plane source) in LDS once, and every thread accumulates
// the taps of its own pixel with v_dot2_f32_f16 against the block's weights, which all lanes read from LDS at the same address
// (broadcast).  Same K tables and packed weights as pp_conv2d's other kernels (plain fp16 layers: 64-channel blocks; split-plane
// layers: the tri-product format, hi x W_hi + lo x W_hi + hi x W_lo in fp32).  fp32 accumulation; outputs fp16 or fp32.
#include "conv_params.h"
#include <stdlib.h>

namespace pp {

constexpr int HD_T = 16, HD_P = HD_T + 2;                       // tile / patch edge
constexpr int HD_PATCH_BYTES = HD_P * HD_P * 128;                // 41 472

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int COUT, bool TRI>
__global__ __launch_bounds__(256) void conv_head_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char patch[HD_PATCH_BYTES];
  __shared__ __attribute__((aligned(16))) char wl[9 * COUT * 128];                 // [tap][cout][64 halves]
  const int tid = threadIdx.x;
  const int tiles_x = (p.W + HD_T - 1) / HD_T, tiles_y = (p.H + HD_T - 1) / HD_T;
  int tile = blockIdx.x;
  const int txi = tile % tiles_x; tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * HD_T, tx0 = txi * HD_T;
  const int ly = tid >> 4, lx = tid & 15;
  const long long img0 = (long long)n * p.H * p.W;
  const char* src = p.src[0].ptr + p.src[0].choff * 2;
  const int rowb = p.src[0].cstride * 2;
  const int lob = TRI ? p.src[0].lo * 2 : 0;
  const int nblocks = p.kchunks / 72;                            // K blocks: 9 taps x 8 chunks each
  const long long wrow = (long long)p.kchunks * 16;              // bytes per packed cout row

  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;

  for (int blk = 0; blk < nblocks; ++blk) {
    const int choff = p.ktable[blk * 72].w;                      // first channel of the block inside the source window
    __syncthreads();                                             // the previous block's patch / weights are consumed
    // ---- stage the patch: chunk c of patch pixel pp at slot c ^ ((pp >> 1) & 7) of its 128-byte row (zeros outside the image).  All
    //      loads of a thread are issued before its LDS stores (a rolled load -> store loop pays one memory latency per iteration)
    {
      constexpr int NCH = HD_P * HD_P * 8, ITER = (NCH + 255) / 256;
      u32x4 v[ITER];
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        const int pp = i >> 3, c = i & 7;
        const int yy = ty0 - 1 + pp / HD_P, xx = tx0 - 1 + pp % HD_P;
        v[k] = u32x4{0, 0, 0, 0};
        if (i < NCH && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
          const int coff = TRI ? choff * 2 + (c & 3) * 16 + ((c & 4) ? lob : 0) : choff * 2 + c * 16;
          v[k] = *reinterpret_cast<const u32x4*>(src + (img0 + (long long)yy * p.W + xx) * rowb + coff);
        }
      }
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        const int pp = i >> 3, c = i & 7;
        if (i < NCH) *reinterpret_cast<u32x4*>(patch + pp * 128 + ((c ^ ((pp >> 1) & 7)) << 4)) = v[k];
      }
    }
    // ---- the block's weights: [tap][cout][64] (the packed row of a cout holds, per tap, the block's 64 K elements contiguously)
    for (int i = tid; i < 9 * COUT * 8; i += 256) {
      const int c8 = i & 7, co = (i >> 3) % COUT, t = i / (8 * COUT);
      *reinterpret_cast<u32x4*>(wl + (t * COUT + co) * 128 + c8 * 16) =
          *reinterpret_cast<const u32x4*>(p.weight + co * wrow + ((long long)(blk * 9 + t) * 8 + c8) * 16);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int pp = (ly + t / 3) * HD_P + lx + t % 3;
      const char* prow = patch + pp * 128;
      const int key = (pp >> 1) & 7;
      h2_t a[32];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(prow + ((c ^ key) << 4));
        const h2_t* h = reinterpret_cast<const h2_t*>(&raw);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[c * 4 + j] = h[j];
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const char* wp = wl + (t * COUT + co) * 128;
        float s = 0.f;
        if constexpr (TRI) {
          // a[0..15] = 32 ch hi, a[16..31] = the same channels lo; w[0..15] = W_hi, w[16..31] = W_lo
          float s_hl = 0.f, s_lh = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const u32x4 rh = *reinterpret_cast<const u32x4*>(wp + c * 16), rl = *reinterpret_cast<const u32x4*>(wp + 64 + c * 16);
            const h2_t* wh = reinterpret_cast<const h2_t*>(&rh);
            const h2_t* wlo = reinterpret_cast<const h2_t*>(&rl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              s = __builtin_amdgcn_fdot2(a[c * 4 + j], wh[j], s, false);                // hi x W_hi
              s_lh = __builtin_amdgcn_fdot2(a[16 + c * 4 + j], wh[j], s_lh, false);      // lo x W_hi
              s_hl = __builtin_amdgcn_fdot2(a[c * 4 + j], wlo[j], s_hl, false);          // hi x W_lo
            }
          }
          s += s_lh + s_hl;
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const u32x4 rw = *reinterpret_cast<const u32x4*>(wp + c * 16);
            const h2_t* w = reinterpret_cast<const h2_t*>(&rw);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_fdot2(a[c * 4 + j], w[j], s, false);
          }
        }
        acc[co] += s;
      }
    }
  }
  const int oy = ty0 + ly, ox = tx0 + lx;
  if (oy >= p.H || ox >= p.W) return;
  const long long m = img0 + (long long)oy * p.W + ox;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    if (co >= p.cout_g) break;
    const float v = apply_act((acc[co] + (p.bias != nullptr ? p.bias[co] : 0.f)) * p.out_scale, p.act, p.act_param);
    const long long o = m * p.out_cstride + p.out_choff + co;
    if (p.out_f16) reinterpret_cast<_Float16*>(p.out)[o] = (_Float16)v;
    else reinterpret_cast<float*>(p.out)[o] = v;
  }
#endif
}

// Returns -1000 when the layer is outside this kernel's family.
int conv_head_dispatch(const ConvParams& p, hipStream_t stream, bool force) {
  // MEASURED (profiles/r3l_head_kernel_ab.txt, four interleaved bench runs on one box): no gain over the 16-cout MFMA tiles -- 1 516 / 1 522 ms
  // per clip without, 1 523 / 1 526 with this kernel.  Not dispatched by default; PP_HEAD_KERNEL=1 (or impl 110 for fp16 layers) selects it.
  static const bool on = getenv("PP_HEAD_KERNEL") != nullptr && getenv("PP_HEAD_KERNEL")[0] == '1';
  if (!on && !force) return -1000;
  if (p.cout_g > 4 || p.tap_h != 3 || p.tap_w != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return -1000;
  if (p.nsrc != 1 || p.groups != 1 || p.pad_mode != 0 || p.residual != nullptr || p.preadd != nullptr || p.fuse != PP_FUSE_NONE ||
      p.act2 != PP_ACT_NONE || p.dcn != nullptr)
    return -1000;
  if (p.kchunks % 72 != 0 || !(p.ktable_uniform & 8)) return -1000;                 // whole K blocks of 9 taps x 8 chunks, (tap, source)-uniform
  if (p.split == 1 || (p.split == 2 && p.out_f16)) return -1000;                    // split-plane layers: tri-product format, plain fp32 output
  const long long nblk = (long long)p.N * ((p.H + HD_T - 1) / HD_T) * ((p.W + HD_T - 1) / HD_T);
  if (nblk >= (1ll << 31)) return -1000;
  const dim3 grid((unsigned)nblk), block(256);
  const bool tri = p.split == 2;
#define PP_HEAD(CO)                                                                                              \
  do {                                                                                                           \
    if (tri) hipLaunchKernelGGL((conv_head_kernel<CO, true>), grid, block, 0, stream, p);                        \
    else hipLaunchKernelGGL((conv_head_kernel<CO, false>), grid, block, 0, stream, p);                           \
  } while (0)
  if (p.cout_g <= 2) PP_HEAD(2);
  else if (p.cout_g == 3) PP_HEAD(3);
  else PP_HEAD(4);
#undef PP_HEAD
  return launch_status("pp_conv2d(head)");
}

}  // namespace pp
