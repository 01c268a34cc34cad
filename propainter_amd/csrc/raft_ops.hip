// RAFT correlation pyramid pooling, 9x9x4 bilinear lookup and convex 8x upsampling for gfx950.
// All three are HBM/latency-bound gathers; the all-pairs volume itself is a batched GEMM on pp_conv2d.
#include "common.h"

namespace pp {

// out[m, y, x] = mean of the 2x2 block of in[m] (floor sizes, F.avg_pool2d(2, 2)); one thread per output.
__global__ void corr_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, long long M, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long long total = M * OH * OW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const long long m = i / ((long long)OW * OH);
    const float* s = in + (m * H + 2 * oy) * W + 2 * ox;
    out[i] = (s[0] + s[1] + s[W] + s[W + 1]) * 0.25f;
  }
}

// One WAVE per source pixel, pixels walked with a grid stride by long-lived blocks (round 2: one 256-thread block per pixel, one
// wave per level, a block barrier between staging and blending -- 504 000 four-wave blocks per launch of the 720p chunk, each wave
// with two dependent 4-byte loads in flight: 0.87 ms per launch = 1.5 TB/s of algorithmic bytes).  Per pixel the wave issues the eight
// load rounds of all four levels (10 x 10 positions each, lanes along the rows: 40-byte row segments) back to back, parks them in a
// wave-private LDS patch (zeros outside the map), then blends its 324 outputs -- 4 neighbours with the reference's per-tap coordinate
// round trip (bilinear_sampler's 2c / (W - 1) - 1 normalisation, RAFT/utils/utils.py:61-65) -- and stores them as full rows.
// Output channel l*81 + a*9 + b <- sample (x / 2^l + a - 4, y / 2^l + b - 4): a moves x (RAFT/corr.py:36-43).
// SPLIT (TO = _Float16): split-plane output -- hi = fp16(v) at channel i, lo = fp16(v - hi) at channel i + ocs / 2 (PP_F16S).
template <typename TO, bool SPLIT = false>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                          const float* __restrict__ l2, const float* __restrict__ l3,
                                                          const float* __restrict__ coords, TO* __restrict__ out,
                                                          int ocs, int ocpad, int h, int w, long long npix) {
  __shared__ float patch_all[4][4][10][11];                    // [wave][level][row][col]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*patch)[10][11] = patch_all[wave];
  for (long long pix = (long long)blockIdx.x * 4 + wave; pix < npix; pix += (long long)gridDim.x * 4) {
    const float cx0 = coords[pix * 2], cy0 = coords[pix * 2 + 1];
    // ---- stage: all loads of the pixel first (one exposed latency), then the LDS writes
    float v[4][2];
    int x0s[4], y0s[4];
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl) {
      const float* base = lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3;
      const int Hl = h >> lvl, Wl = w >> lvl;
      const float* map = base + pix * (long long)Hl * Wl;
      const float scale = 1.f / (float)(1 << lvl);
      // all 81 taps share the fractional part when computed on the un-shifted centre; to stay faithful to the reference's per-tap
      // round trip it is evaluated per tap below, the patch is staged from the floor of the centre (taps are centre + integer, so
      // floor(tap) = floor(centre) + integer up to 1-ulp effects that only move weight between two staged neighbours)
      const int x0 = (int)floorf(cx0 * scale) - 4, y0 = (int)floorf(cy0 * scale) - 4;
      x0s[lvl] = x0; y0s[lvl] = y0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = lane + 64 * k;
        const int r = i / 10, c = i - r * 10;
        const int yy = y0 + r, xx = x0 + c;
        v[lvl][k] = (i < 100 && yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) ? map[(long long)yy * Wl + xx] : 0.f;
      }
    }
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = lane + 64 * k;
        if (i < 100) patch[lvl][i / 10][i % 10] = v[lvl][k];
      }
    // (the patch is wave-private: no block barrier; the wave barrier is free on wave64 and keeps the compiler from moving or caching the
    //  dynamically indexed LDS accesses across the hand-over between lanes)
    __builtin_amdgcn_wave_barrier();
    TO* op = out + pix * ocs;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i = lane + 64 * k;                 // output channel
      if (i >= 324) break;
      const int lvl = i / 81, j = i - lvl * 81;
      const int a = j / 9, b = j - a * 9;          // a moves x, b moves y
      const int Hl = h >> lvl, Wl = w >> lvl;
      const float scale = 1.f / (float)(1 << lvl);
      const float cx = cx0 * scale, cy = cy0 * scale;
      const float px = grid_roundtrip(cx + (float)(a - 4), Wl);
      const float py = grid_roundtrip(cy + (float)(b - 4), Hl);
      const float fx = floorf(px), fy = floorf(py);
      const float lx = px - fx, ly = py - fy;
      const int x0 = lvl == 0 ? x0s[0] : lvl == 1 ? x0s[1] : lvl == 2 ? x0s[2] : x0s[3];
      const int y0 = lvl == 0 ? y0s[0] : lvl == 1 ? y0s[1] : lvl == 2 ? y0s[2] : y0s[3];
      const int c0 = (int)fx - x0, r0 = (int)fy - y0;  // position inside the staged patch
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = r0 + (q >> 1), cc = c0 + (q & 1);
        const float wgt = ((q & 1) ? lx : 1.f - lx) * ((q >> 1) ? ly : 1.f - ly);
        float s = 0.f;
        if (rr >= 0 && rr < 10 && cc >= 0 && cc < 10) s = patch[lvl][rr][cc];
        else {
          const int yy = y0 + rr, xx = x0 + cc;   // 1-ulp spill outside the staged window: read directly
          if (yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) {
            const float* base = lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3;
            s = base[pix * (long long)Hl * Wl + (long long)yy * Wl + xx];
          }
        }
        acc += wgt * s;
      }
      op[i] = from_f32<TO>(acc);
      if constexpr (SPLIT) op[i + ocs / 2] = from_f32<TO>(acc - to_f32(from_f32<TO>(acc)));
    }
    for (int i = 324 + lane; i < ocpad; i += 64) {
      op[i] = from_f32<TO>(0.f);
      if constexpr (SPLIT) op[ocs / 2 + i] = from_f32<TO>(0.f);
    }
    __builtin_amdgcn_wave_barrier();      // the next pixel re-stages the patch: not before every lane has blended this one
  }
}

// One thread per fine output pixel pair (both flow channels): softmax over the 9 mask logits of its
// (coarse pixel, sub-position) and convex combination of the 3x3 coarse neighbourhood of 8*flow.
// Thread order: the 64 sub-positions of ONE coarse pixel are the 64 lanes of a wave, so every mask read of a wave is 256 contiguous
// bytes (the mask is the large operand: 576 logits per coarse pixel against 128 output values); the output goes out as eight 32-byte
// row pieces per wave.  (Fine-pixel raster order read the mask in 32-byte pieces of eight different rows: 3.5x the algorithmic HBM
// bytes, profiles/r4z_hbm_traffic_720p.json.)
template <typename TM>
__global__ void convex_upsample_kernel(const float* __restrict__ flow, const TM* __restrict__ mask, int mcs,
                                       float* __restrict__ out, int B, int h, int w) {
  const int OW = 8 * w, OH = 8 * h;
  const long long total = (long long)B * OH * OW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int sub = (int)(i & 63);
    const long long cp = i >> 6;                               // coarse pixel (n, y, x)
    const int x = (int)(cp % w), y = (int)((cp / w) % h);
    const long long n = cp / ((long long)w * h);
    const int X = 8 * x + (sub & 7), Y = 8 * y + (sub >> 3);
    const TM* mp = mask + ((n * h + y) * (long long)w + x) * mcs + sub;
    float lg[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { lg[k] = to_f32(mp[k * 64]); mx = fmaxf(mx, lg[k]); }
    float den = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float e = __expf(lg[k] - mx);
      den += e;
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
        const float* fp = flow + ((n * h + yy) * (long long)w + xx) * 2;
        ax += e * 8.f * fp[0];
        ay += e * 8.f * fp[1];
      }
    }
    out[((n * 2 + 0) * OH + Y) * (long long)OW + X] = ax / den;
    out[((n * 2 + 1) * OH + Y) * (long long)OW + X] = ay / den;
  }
}

static inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace pp

using namespace pp;

extern "C" int pp_corr_avgpool(const float* in, float* out, int64_t M, int H, int W, void* stream) {
  PP_REQUIRE(in && out && M > 0 && H >= 2 && W >= 2, PP_ERR_ARG, "pp_corr_avgpool: bad arguments");
  hipLaunchKernelGGL(corr_avgpool_kernel, dim3(grid_for(M * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, in, out,
                     (long long)M, H, W);
  return launch_status("pp_corr_avgpool");
}

extern "C" int pp_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                              const float* coords, void* out, int out_cstride, int out_cpad, int B, int h, int w,
                              int out_dtype, void* stream) {
  PP_REQUIRE(lvl0 && lvl1 && lvl2 && lvl3 && coords && out, PP_ERR_ARG, "pp_corr_lookup: null pointer");
  PP_REQUIRE(B > 0 && (h >> 3) >= 2 && (w >> 3) >= 2, PP_ERR_ARG,
             "pp_corr_lookup: level-3 map would be %dx%d (RAFT needs >= 2x2: inputs of at least 128x128)", h >> 3, w >> 3);
  PP_REQUIRE(out_cpad >= 324 && out_cstride >= out_cpad, PP_ERR_ARG, "pp_corr_lookup: out_cpad %d / cstride %d", out_cpad, out_cstride);
  PP_REQUIRE(out_dtype == PP_F32 || out_dtype == PP_F16 || out_dtype == PP_F16S, PP_ERR_DTYPE, "pp_corr_lookup: dtype %d", out_dtype);
  PP_REQUIRE(out_dtype != PP_F16S || (out_cstride % 2 == 0 && out_cstride / 2 >= out_cpad), PP_ERR_ARG,
             "pp_corr_lookup: split-plane output needs an even cstride with cstride / 2 >= out_cpad");
  const long long npix = (long long)B * h * w;
  hipStream_t st = (hipStream_t)stream;
  // one wave per pixel, 4 waves per block, blocks walk the pixels with a grid stride: enough blocks to fill every CU eight times
  const long long want = (npix + 3) / 4;
  const unsigned lk_grid = (unsigned)(want < 256 * 8 ? want : 256 * 8);
  if (out_dtype == PP_F16S)
    hipLaunchKernelGGL((corr_lookup_kernel<_Float16, true>), dim3(lk_grid), dim3(256), 0, st, lvl0, lvl1, lvl2, lvl3, coords,
                       (_Float16*)out, out_cstride, out_cpad, h, w, npix);
  else if (out_dtype == PP_F16)
    hipLaunchKernelGGL((corr_lookup_kernel<_Float16>), dim3(lk_grid), dim3(256), 0, st, lvl0, lvl1, lvl2, lvl3, coords,
                       (_Float16*)out, out_cstride, out_cpad, h, w, npix);
  else
    hipLaunchKernelGGL((corr_lookup_kernel<float>), dim3(lk_grid), dim3(256), 0, st, lvl0, lvl1, lvl2, lvl3, coords,
                       (float*)out, out_cstride, out_cpad, h, w, npix);
  return launch_status("pp_corr_lookup");
}

extern "C" int pp_convex_upsample(const float* flow, const void* mask, int mask_cstride, int mask_dtype, float* out, int B,
                                  int h, int w, void* stream) {
  PP_REQUIRE(flow && mask && out && B > 0 && h > 0 && w > 0 && mask_cstride >= 576, PP_ERR_ARG, "pp_convex_upsample: bad arguments");
  PP_REQUIRE(mask_dtype == PP_F32 || mask_dtype == PP_F16, PP_ERR_DTYPE, "pp_convex_upsample: dtype %d", mask_dtype);
  const int g = grid_for((long long)B * 64 * h * w);
  hipStream_t st = (hipStream_t)stream;
  if (mask_dtype == PP_F16)
    hipLaunchKernelGGL((convex_upsample_kernel<_Float16>), dim3(g), dim3(256), 0, st, flow, (const _Float16*)mask, mask_cstride,
                       out, B, h, w);
  else
    hipLaunchKernelGGL((convex_upsample_kernel<float>), dim3(g), dim3(256), 0, st, flow, (const float*)mask, mask_cstride, out,
                       B, h, w);
  return launch_status("pp_convex_upsample");
}

namespace pp {
// One thread per pixel: 7 horizontal taps x (x, y) of the fp32 flow coords1 - coords0 -> 16 channels (32 / 64 contiguous bytes)
// SPLIT (T = _Float16, PP_F16S): rows are [16 hi | 16 lo] (32 elements per pixel), flow_out's lo plane sits fcs / 2 further
template <typename T, bool SPLIT = false>
__global__ void raft_flow_taps_kernel(const float* __restrict__ c1, const float* __restrict__ c0, T* __restrict__ rows,
                                      T* __restrict__ fout, int fcs, int fco, long long npix, int w) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    float v[16];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int xx = x + kx - 3;
      float fx = 0.f, fy = 0.f;
      if (xx >= 0 && xx < w) {
        const long long j = i + (kx - 3);
        const float2 a = *reinterpret_cast<const float2*>(c1 + 2 * j), b = *reinterpret_cast<const float2*>(c0 + 2 * j);
        fx = a.x - b.x; fy = a.y - b.y;
      }
      v[2 * kx] = fx; v[2 * kx + 1] = fy;
    }
    v[14] = v[15] = 0.f;
    if constexpr (SPLIT) {
      store8_split(rows + i * 32, rows + i * 32 + 16, v);
      store8_split(rows + i * 32 + 8, rows + i * 32 + 24, v + 8);
      if (fout != nullptr) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const T hi = from_f32<T>(v[6 + c]);
          fout[i * fcs + fco + c] = hi;
          fout[i * fcs + fcs / 2 + fco + c] = from_f32<T>(v[6 + c] - to_f32(hi));
        }
      }
      continue;
    }
    store8<T>(rows + i * 16, v);
    store8<T>(rows + i * 16 + 8, v + 8);
    if (fout != nullptr) {
      fout[i * fcs + fco] = from_f32<T>(v[6]);
      fout[i * fcs + fco + 1] = from_f32<T>(v[7]);
    }
  }
}
}  // namespace pp

extern "C" int pp_raft_flow_taps(const float* coords1, const float* coords0, void* rows, void* flow_out, int flow_cstride,
                                 int flow_choff, int P, int h, int w, int dtype, void* stream) {
  PP_REQUIRE(coords1 && coords0 && rows && P > 0 && h > 0 && w > 0, PP_ERR_ARG, "pp_raft_flow_taps: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16 || dtype == PP_F16S, PP_ERR_DTYPE, "pp_raft_flow_taps: dtype %d", dtype);
  PP_REQUIRE(flow_out == nullptr || (flow_cstride >= flow_choff + 2 && flow_choff >= 0), PP_ERR_ARG, "pp_raft_flow_taps: flow window");
  PP_REQUIRE(dtype != PP_F16S || flow_out == nullptr || (flow_cstride % 2 == 0 && flow_cstride / 2 >= flow_choff + 2), PP_ERR_ARG,
             "pp_raft_flow_taps: split-plane flow window (lo plane at cstride / 2)");
  PP_REQUIRE(((uintptr_t)rows % 16) == 0 && ((uintptr_t)coords1 % 8) == 0 && ((uintptr_t)coords0 % 8) == 0, PP_ERR_ALIGN,
             "pp_raft_flow_taps: rows must be 16-byte aligned, coords 8-byte aligned");
  const long long npix = (long long)P * h * w;
  const int g = grid_for(npix);
  if (dtype == PP_F16S)
    hipLaunchKernelGGL((raft_flow_taps_kernel<_Float16, true>), dim3(g), dim3(256), 0, (hipStream_t)stream, coords1, coords0, (_Float16*)rows,
                       (_Float16*)flow_out, flow_cstride, flow_choff, npix, w);
  else if (dtype == PP_F16)
    hipLaunchKernelGGL((raft_flow_taps_kernel<_Float16>), dim3(g), dim3(256), 0, (hipStream_t)stream, coords1, coords0, (_Float16*)rows,
                       (_Float16*)flow_out, flow_cstride, flow_choff, npix, w);
  else
    hipLaunchKernelGGL((raft_flow_taps_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, coords1, coords0, (float*)rows,
                       (float*)flow_out, flow_cstride, flow_choff, npix, w);
  return launch_status("pp_raft_flow_taps");
}
