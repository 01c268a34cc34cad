// Flow-guided sampling kernels (HBM-bound gathers): flow warp, forward-backward consistency check and
// the fused step of the non-learnable image propagation.  Coordinates are always computed in fp32
// (the reference builds its grid in x.dtype, model/modules/flow_loss_utils.py:32, which quantises
// coordinates > 1024 in fp16 mode; we compare against the fp32 oracle instead).
#include "common.h"

namespace pp {

// Bilinear tap set with zeros padding at pixel coordinates (px, py) of an HxW image.
struct Taps {
  int idx[4];     // y*W + x, or -1
  float w[4];
};

__device__ __forceinline__ Taps bilinear_taps(float px, float py, int H, int W) {
  Taps t;
  const float fx = floorf(px), fy = floorf(py);
  const int x0 = (int)fx, y0 = (int)fy;
  const float lx = px - fx, ly = py - fy;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
    const bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H;
    t.idx[c] = ok ? yy * W + xx : -1;
    t.w[c] = ((c & 1) ? lx : 1.f - lx) * ((c >> 1) ? ly : 1.f - ly);
  }
  return t;
}

// NHWC warp: one thread per (pixel, 8-channel chunk) (or per pixel when C <= 4).
template <typename T, bool WIDE>
__global__ void flow_warp_kernel(const T* __restrict__ x, int xcs, int xco, const T* __restrict__ flow, int fcs,
                                 int fco, T* __restrict__ out, int ocs, int oco, int N, int H, int W, int C, int mode) {
  const int cchunks = WIDE ? C / 8 : 1;
  const long long total = (long long)N * H * W * cchunks;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    const long long pix = i / cchunks;
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const long long nb = (pix / ((long long)W * H)) * (long long)H * W;
    const float fx = to_f32(flow[pix * fcs + fco]), fy = to_f32(flow[pix * fcs + fco + 1]);
    const float px = grid_roundtrip((float)xw + fx, W), py = grid_roundtrip((float)yh + fy, H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mode == 1) {
      const float rx = nearbyintf(px), ry = nearbyintf(py);
      const int xi = (int)rx, yi = (int)ry;
      if (xi >= 0 && xi < W && yi >= 0 && yi < H) {
        const T* sp = x + (nb + (long long)yi * W + xi) * xcs + xco + cc * 8;
        if constexpr (WIDE) load8<T>(sp, acc);
        else for (int c = 0; c < C; ++c) acc[c] = to_f32(sp[c]);
      }
    } else {
      const Taps t = bilinear_taps(px, py, H, W);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (t.idx[k] < 0) continue;
        const T* sp = x + (nb + t.idx[k]) * xcs + xco + cc * 8;
        if constexpr (WIDE) {
          float v[8];
          load8<T>(sp, v);
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] += t.w[k] * v[c];
        } else {
          for (int c = 0; c < C; ++c) acc[c] += t.w[k] * to_f32(sp[c]);
        }
      }
    }
    T* op = out + pix * ocs + oco + cc * 8;
    if constexpr (WIDE) store8<T>(op, acc);
    else for (int c = 0; c < C; ++c) op[c] = from_f32<T>(acc[c]);
  }
}

template <typename T>
__global__ void fb_check_kernel(const T* __restrict__ fw, int fwcs, const T* __restrict__ bw, int bwcs, T* __restrict__ out,
                                int ocs, int oco, int N, int H, int W) {
  const long long total = (long long)N * H * W;
  for (long long pix = blockIdx.x * (long long)blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
    const int xw = (int)(pix % W);
    const int yh = (int)((pix / W) % H);
    const long long nb = (pix / ((long long)W * H)) * (long long)H * W;
    const float fx = to_f32(fw[pix * fwcs]), fy = to_f32(fw[pix * fwcs + 1]);
    const Taps t = bilinear_taps(grid_roundtrip((float)xw + fx, W), grid_roundtrip((float)yh + fy, H), H, W);
    float bx = 0.f, by = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (t.idx[k] >= 0) {
        bx += t.w[k] * to_f32(bw[(nb + t.idx[k]) * bwcs]);
        by += t.w[k] * to_f32(bw[(nb + t.idx[k]) * bwcs + 1]);
      }
    // the reference rounds the warped flow to the tensor dtype before the test
    bx = to_f32(from_f32<T>(bx)); by = to_f32(from_f32<T>(by));
    const float dx = fx + bx, dy = fy + by;
    const float mag = fx * fx + fy * fy + bx * bx + by * by;
    out[pix * ocs + oco] = from_f32<T>((dx * dx + dy * dy) < (0.01f * mag + 0.5f) ? 1.f : 0.f);
  }
}

// Planar (NCHW) fused image-propagation step; one thread per pixel.
template <typename T>
__global__ void img_prop_step_kernel(const T* __restrict__ x_prop, const T* __restrict__ m_prop, const T* __restrict__ x_cur,
                                     const T* __restrict__ m_cur, const T* __restrict__ f_prop, const T* __restrict__ f_chk,
                                     T* __restrict__ x_out, T* __restrict__ m_out, int N, int C, int H, int W, int mode) {
  const long long hw = (long long)H * W;
  const long long total = (long long)N * hw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xw = (int)(i % W);
    const int yh = (int)((i / W) % H);
    const long long n = i / hw;
    const long long p = i - n * hw;
    const T* fp = f_prop + n * 2 * hw;
    const T* fc = f_chk + n * 2 * hw;
    const float fx = to_f32(fp[p]), fy = to_f32(fp[hw + p]);
    const float px = grid_roundtrip((float)xw + fx, W), py = grid_roundtrip((float)yh + fy, H);
    const Taps t = bilinear_taps(px, py, H, W);
    float bx = 0.f, by = 0.f, mw = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (t.idx[k] >= 0) {
        bx += t.w[k] * to_f32(fc[t.idx[k]]);
        by += t.w[k] * to_f32(fc[hw + t.idx[k]]);
        mw += t.w[k] * to_f32(m_prop[n * hw + t.idx[k]]);
      }
    bx = to_f32(from_f32<T>(bx)); by = to_f32(from_f32<T>(by)); mw = to_f32(from_f32<T>(mw));
    const float dx = fx + bx, dy = fy + by;
    const float valid = (dx * dx + dy * dy) < (0.01f * (fx * fx + fy * fy + bx * bx + by * by) + 0.5f) ? 1.f : 0.f;
    const float mwb = mw > 0.1f ? 1.f : 0.f;
    const float mc = to_f32(m_cur[n * hw + p]);
    const float u = (mc * valid * (1.f - mwb)) > 0.1f ? 1.f : 0.f;
    const float mo = (mc * (1.f - valid * (1.f - mwb))) > 0.1f ? 1.f : 0.f;
    m_out[n * hw + p] = from_f32<T>(mo);
    int sidx = -1;
    if (mode == 1) {
      const int xi = (int)nearbyintf(px), yi = (int)nearbyintf(py);
      if (xi >= 0 && xi < W && yi >= 0 && yi < H) sidx = yi * W + xi;
    }
    for (int c = 0; c < C; ++c) {
      const T* xp = x_prop + (n * C + c) * hw;
      float wv = 0.f;
      if (mode == 1) {
        if (sidx >= 0) wv = to_f32(xp[sidx]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t.idx[k] >= 0) wv += t.w[k] * to_f32(xp[t.idx[k]]);
        wv = to_f32(from_f32<T>(wv));
      }
      const float cur = to_f32(x_cur[(n * C + c) * hw + p]);
      x_out[(n * C + c) * hw + p] = from_f32<T>(u * wv + (1.f - u) * cur);
    }
  }
}

static inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

// Mask dilation of the driver (inference_propainter.py:96,105: scipy.ndimage.binary_dilation(mask, iterations=k) with the
// default 4-connected structuring element and zero border): k iterations of the cross == the L1 ball of radius k, so
// out = 255 iff some non-zero input pixel lies within |dy| + |dx| <= k.  uint8 planar [N,H,W]; k = 0 -> plain binarisation.
__global__ void binary_dilate_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int N, int H, int W,
                                     int k) {
  const long long total = (long long)N * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long base = (i / ((long long)W * H)) * H * W;
    bool hit = false;
    for (int dy = -k; dy <= k && !hit; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      const int r = k - (dy < 0 ? -dy : dy);
      const int x0 = max(0, x - r), x1 = min(W - 1, x + r);
      for (int xx = x0; xx <= x1; ++xx)
        if (in[base + (long long)yy * W + xx] != 0) { hit = true; break; }
    }
    out[i] = hit ? 255 : 0;
  }
}

// Ordered uint8 composite of one generator window (inference_propainter.py:435-450), all its local frames in ONE launch.
//   img  = uint8( ((pred + 1) / 2) * 255 )      every operation rounded in pred's own dtype like the reference's tensor / numpy ops,
//                                                then truncated towards zero (.astype(np.uint8))
//   cur  = mask ? img : original
//   comp = blend ? uint8(0.5f * comp + 0.5f * cur) : cur          (the frame already carries an earlier window's result)
// pred planar [n, 3, H, W]; mask [L, H, W] (non-zero = hole), original / comp [L, H, W, 3] uint8; frame i of the window is clip frame
// ids[i]; bit i of blend_bits says whether that frame was composited before.  One thread per pixel (3 bytes).
struct CompositeIds { int ids[32]; };
template <typename T>
__global__ void composite_window_kernel(const T* __restrict__ pred, const unsigned char* __restrict__ mask, int mask_stride,
                                        const unsigned char* __restrict__ ori, unsigned char* __restrict__ comp, const CompositeIds fr,
                                        unsigned blend_bits, int n, int HW) {
  const long long total = (long long)n * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i / HW), px = (int)(i - (long long)f * HW);
    const long long g = (long long)fr.ids[f] * HW + px;          // pixel of the clip frame
    const bool hole = mask[g * mask_stride] != 0;
    const bool blend = (blend_bits >> f) & 1u;
    unsigned char* cp = comp + g * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      unsigned char cur = ori[g * 3 + c];
      if (hole) {
        const T one = from_f32<T>(1.f), two = from_f32<T>(2.f), s255 = from_f32<T>(255.f);
        T v = pred[((long long)f * 3 + c) * HW + px];
        v = v + one;             // one rounding per operation in T (fp16: v_add_f16 / v_mul_f16; the build uses -ffp-contract=off)
        v = v / two;
        v = v * s255;
        cur = (unsigned char)to_f32(v);
      }
      if (blend) cur = (unsigned char)((float)cp[c] * 0.5f + (float)cur * 0.5f);
      cp[c] = cur;
    }
  }
}


// ---- final resize of the composited frames (inference_propainter.py:469-470: cv2.resize(f, out_size), INTER_LINEAR on uint8) -------------
// OpenCV's 8-bit bilinear resize is FIXED-POINT: per axis the source position of output d is f = (float)((d + 0.5) * scale - 0.5),
// s = floor(f), clamped to the image with the fractional part dropped at the borders; the two weights are rounded to 11 bits
// (cvRound(w * 2048), INTER_RESIZE_COEF_BITS), the horizontal pass keeps S[s] * a0 + S[s + 1] * a1 as an integer and the vertical pass
// returns (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  An exact 2 : 1 reduction in both axes takes the INTER_AREA
// path instead (cv::resize: "INTER_LINEAR ... iscale 2 -> INTER_AREA"): (a + b + c + d + 2) >> 2.  This kernel restates that arithmetic
// (published algorithm of the un-vendored dependency opencv-python, requirements.txt; cv2 is absent offline: parity pinned against
// propainter_amd/video_io.py's numpy restatement, byte for byte, and against float bilinear interpolation within 1 level).
__device__ __forceinline__ void cv_linear_coeff(int d, double scale, int n_src, int& s, int& a0, int& a1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f -= (float)si;
  if (si < 0) { f = 0.f; si = 0; }
  if (si >= n_src - 1) { f = 0.f; si = n_src - 1; }
  s = si;
  a0 = __float2int_rn((1.f - f) * 2048.f);
  a1 = __float2int_rn(f * 2048.f);
}

__global__ void resize_bilinear_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int N, int H, int W, int C,
                                          int OH, int OW) {
  const long long total = (long long)N * OH * OW;
  const double sx = (double)W / OW, sy = (double)H / OH;
  const bool area2 = (W == 2 * OW) && (H == 2 * OH);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % OW);
    const long long r = i / OW;
    const int y = (int)(r % OH);
    const long long n = r / OH;
    const unsigned char* img = src + n * H * W * C;
    unsigned char* o = dst + i * C;
    if (area2) {
      const unsigned char* p0 = img + ((long long)(2 * y) * W + 2 * x) * C;
      const unsigned char* p1 = p0 + (long long)W * C;
      for (int c = 0; c < C; ++c) o[c] = (unsigned char)((p0[c] + p0[C + c] + p1[c] + p1[C + c] + 2) >> 2);
      continue;
    }
    int x0, y0, a0, a1, b0, b1;
    cv_linear_coeff(x, sx, W, x0, a0, a1);
    cv_linear_coeff(y, sy, H, y0, b0, b1);
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const unsigned char* r0 = img + (long long)y0 * W * C;
    const unsigned char* r1 = img + (long long)y1 * W * C;
    for (int c = 0; c < C; ++c) {
      const int h0 = r0[x0 * C + c] * a0 + r0[x1 * C + c] * a1;
      const int h1 = r1[x0 * C + c] * a0 + r1[x1 * C + c] * a1;
      o[c] = (unsigned char)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
  }
}

}  // namespace pp

using namespace pp;

extern "C" int pp_composite_window(const void* pred, int dtype, const void* mask, int mask_stride, const void* original, void* comp,
                                   const int32_t* frame_ids, uint32_t blend_bits, int n, int H, int W, void* stream) {
  PP_REQUIRE(pred && mask && original && comp && frame_ids && n > 0 && n <= 32 && H > 0 && W > 0 && mask_stride >= 1, PP_ERR_ARG,
             "pp_composite_window: bad arguments (n = %d local frames, at most 32)", n);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_composite_window: dtype %d", dtype);
  CompositeIds fr;
  for (int i = 0; i < 32; ++i) fr.ids[i] = i < n ? frame_ids[i] : 0;
  for (int i = 0; i < n; ++i) PP_REQUIRE(fr.ids[i] >= 0, PP_ERR_ARG, "pp_composite_window: negative frame id");
  const int g = grid_for((long long)n * H * W);
  if (dtype == PP_F16)
    hipLaunchKernelGGL((composite_window_kernel<_Float16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const _Float16*)pred,
                       (const unsigned char*)mask, mask_stride, (const unsigned char*)original, (unsigned char*)comp, fr, blend_bits, n, H * W);
  else
    hipLaunchKernelGGL((composite_window_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)pred,
                       (const unsigned char*)mask, mask_stride, (const unsigned char*)original, (unsigned char*)comp, fr, blend_bits, n, H * W);
  return launch_status("pp_composite_window");
}

extern "C" int pp_flow_warp(const void* x, int x_cstride, int x_choff, const void* flow, int fl_cstride, int fl_choff,
                            void* out, int out_cstride, int out_choff, int N, int H, int W, int C, int mode, int dtype,
                            void* stream) {
  PP_REQUIRE(x && flow && out && N > 0 && H > 0 && W > 0 && C > 0, PP_ERR_ARG, "pp_flow_warp: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_flow_warp: dtype %d", dtype);
  PP_REQUIRE(mode == 0 || mode == 1, PP_ERR_ARG, "pp_flow_warp: mode %d", mode);
  const bool wide = C % 8 == 0;
  PP_REQUIRE(wide || C <= 4, PP_ERR_ALIGN, "pp_flow_warp: C=%d must be a multiple of 8 or <= 4", C);
  const int esz = dtype == PP_F16 ? 2 : 4;
  if (wide)
    PP_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (x_cstride * esz) % 16 == 0 && (x_choff * esz) % 16 == 0 &&
                   (out_cstride * esz) % 16 == 0 && (out_choff * esz) % 16 == 0,
               PP_ERR_ALIGN, "pp_flow_warp: 16-byte alignment violated");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)N * H * W * (wide ? C / 8 : 1);
  const int g = grid_for(total);
#define LAUNCH(T, WIDE)                                                                                              \
  hipLaunchKernelGGL((flow_warp_kernel<T, WIDE>), dim3(g), dim3(256), 0, st, (const T*)x, x_cstride, x_choff,        \
                     (const T*)flow, fl_cstride, fl_choff, (T*)out, out_cstride, out_choff, N, H, W, C, mode)
  if (dtype == PP_F16) { if (wide) LAUNCH(_Float16, true); else LAUNCH(_Float16, false); }
  else { if (wide) LAUNCH(float, true); else LAUNCH(float, false); }
#undef LAUNCH
  return launch_status("pp_flow_warp");
}

extern "C" int pp_fb_check(const void* flow_fw, int fw_cstride, const void* flow_bw, int bw_cstride, void* out,
                           int out_cstride, int out_choff, int N, int H, int W, int dtype, void* stream) {
  PP_REQUIRE(flow_fw && flow_bw && out && N > 0 && H > 0 && W > 0, PP_ERR_ARG, "pp_fb_check: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_fb_check: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for((long long)N * H * W);
  if (dtype == PP_F16)
    hipLaunchKernelGGL((fb_check_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const _Float16*)flow_fw, fw_cstride,
                       (const _Float16*)flow_bw, bw_cstride, (_Float16*)out, out_cstride, out_choff, N, H, W);
  else
    hipLaunchKernelGGL((fb_check_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)flow_fw, fw_cstride,
                       (const float*)flow_bw, bw_cstride, (float*)out, out_cstride, out_choff, N, H, W);
  return launch_status("pp_fb_check");
}

extern "C" int pp_img_prop_step(const void* x_prop, const void* m_prop, const void* x_cur, const void* m_cur,
                                const void* flow_prop, const void* flow_check, void* x_out, void* m_out, int N, int C,
                                int H, int W, int mode, int dtype, void* stream) {
  PP_REQUIRE(x_prop && m_prop && x_cur && m_cur && flow_prop && flow_check && x_out && m_out, PP_ERR_ARG,
             "pp_img_prop_step: null pointer");
  PP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && (mode == 0 || mode == 1), PP_ERR_ARG, "pp_img_prop_step: bad extents");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_img_prop_step: dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for((long long)N * H * W);
  if (dtype == PP_F16)
    hipLaunchKernelGGL((img_prop_step_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const _Float16*)x_prop,
                       (const _Float16*)m_prop, (const _Float16*)x_cur, (const _Float16*)m_cur, (const _Float16*)flow_prop,
                       (const _Float16*)flow_check, (_Float16*)x_out, (_Float16*)m_out, N, C, H, W, mode);
  else
    hipLaunchKernelGGL((img_prop_step_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x_prop, (const float*)m_prop,
                       (const float*)x_cur, (const float*)m_cur, (const float*)flow_prop, (const float*)flow_check,
                       (float*)x_out, (float*)m_out, N, C, H, W, mode);
  return launch_status("pp_img_prop_step");
}

extern "C" int pp_binary_dilate(const void* mask, void* out, int N, int H, int W, int iterations, void* stream) {
  PP_REQUIRE(mask && out && mask != out && N > 0 && H > 0 && W > 0 && iterations >= 0 && iterations <= 64, PP_ERR_ARG,
             "pp_binary_dilate: bad arguments (iterations %d)", iterations);
  const int g = grid_for((long long)N * H * W);
  hipLaunchKernelGGL(binary_dilate_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)mask,
                     (unsigned char*)out, N, H, W, iterations);
  return launch_status("pp_binary_dilate");
}

extern "C" int pp_resize_bilinear_u8(const void* src, void* dst, int N, int H, int W, int C, int OH, int OW, void* stream) {
  PP_REQUIRE(src && dst && src != dst && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C <= 4, PP_ERR_ARG,
             "pp_resize_bilinear_u8: bad arguments (%dx%dx%d -> %dx%d)", H, W, C, OH, OW);
  const int g = grid_for((long long)N * OH * OW);
  hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, (unsigned char*)dst,
                     N, H, W, C, OH, OW);
  return launch_status("pp_resize_bilinear_u8");
}
