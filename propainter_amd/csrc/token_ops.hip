// Token <-> feature-map kernels and small dense helpers (all HBM-bound, vectorised 16-byte accesses):
// fold (SoftComp / FusionFeedForward), LayerNorm, depthwise pooling, InstanceNorm, bilinear x2
// upsampling, the deformable offset/mask head activation, GRU gating and layout packers.
#include "common.h"

namespace pp {

static inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

// F.fold(7, stride 3, pad 3) as a gather: output pixel (y, x) sums the <= 3x3 patches that cover it.
// tokens [BT, fh*fw, 49*C] in TAP-MAJOR feature order (ky*7 + kx)*C + c: the producing GEMM (fc1 / SoftComp
// embedding) has its weight rows permuted on the host from the reference's c*49 + ky*7 + kx, so that the 8 channels
// a lane folds are one 16-byte load and neighbouring pixels of a patch read neighbouring C-element runs (coalesced);
// every token element is read exactly once.  One thread per (pixel, 8-channel chunk), chunk fastest: full-line
// NHWC stores.  C must be a multiple of 8.
template <typename T>
__global__ __launch_bounds__(256) void fold_tokens_kernel(const T* __restrict__ tok, T* __restrict__ out, int BT, int fh, int fw, int C,
                                                          int H, int W, int normalize, int act) {
  const int cch = C / 8;
  const long long total = (long long)BT * H * W * cch;
  const long long tstride = (long long)C * 49;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cch);
    const long long pix = i / cch;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    // patches py with 3*py - 3 <= y <= 3*py + 3
    const int py0 = max(0, (y - 3 + 2) / 3), py1 = min(fh - 1, (y + 3) / 3);
    const int px0 = max(0, (x - 3 + 2) / 3), px1 = min(fw - 1, (x + 3) / 3);
    const T* base = tok + n * fh * fw * tstride + cc * 8;
    for (int py = py0; py <= py1; ++py) {
      const int ky = y + 3 - 3 * py;
      for (int px = px0; px <= px1; ++px) {
        const int kx = x + 3 - 3 * px;
        float v[8];
        load8<T>(base + ((long long)py * fw + px) * tstride + (ky * 7 + kx) * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
        ++cnt;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = apply_act(normalize ? acc[j] / (float)cnt : acc[j], act, 0.f);
    store8<T>(out + pix * C + cc * 8, acc);
  }
}

// LayerNorm: one wave per row, C/64 elements per lane kept in registers (two-pass, fp32 statistics).
template <typename T, int PER_LANE>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ in, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, long long rows,
                                                        int C, float eps, int gh, int gw, int Hp, int Wp) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long long orow = row;                         // gh > 0: token (n, y, x) of a [N, gh, gw] grid goes to (n, y, x) of a padded [N, Hp, Wp] grid
  if (gh > 0) {
    const int x = (int)(row % gw), y = (int)((row / gw) % gh);
    orow = ((row / ((long long)gw * gh)) * Hp + y) * Wp + x;
  }
  const T* ip = in + row * C;
  float v[PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { v[i] = to_f32(ip[lane + 64 * i]); s += v[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
  T* op = out + orow * C;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int c = lane + 64 * i;
    op[c] = from_f32<T>((v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
}

// depthwise k x k stride-k conv, NHWC; one thread per output (pixel, channel).
template <typename T>
__global__ void depthwise_pool_kernel(const T* __restrict__ in, const float* __restrict__ wgt, const float* __restrict__ bias,
                                      T* __restrict__ out, int N, int H, int W, int C, int k) {
  const int OH = H / k, OW = W / k;
  const long long total = (long long)N * OH * OW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH);
    const long long n = pix / ((long long)OW * OH);
    float acc = bias ? bias[c] : 0.f;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx)
        acc += wgt[(c * k + ky) * k + kx] * to_f32(in[((n * H + oy * k + ky) * (long long)W + ox * k + kx) * C + c]);
    out[i] = from_f32<T>(acc);
  }
}

// Vectorised variant (C % 8 == 0, k*k*C floats of weights fit LDS): one thread per (output pixel, 8 channels), 16-byte
// input loads, weights staged once per block in LDS.  Same accumulation order (bias, then taps in row-major order) as the scalar
// kernel: identical results.
// K > 0: the window size at compile time (the generator's pool_layer is 4 x 4): the K * K input loads of an output are independent and
// all in flight together -- with a run-time k the tap loop stays rolled and a thread waits for one 16-byte load at a time (56 us per
// launch at 2.1 TB/s, profiles/r4z_rocprof_kernel_stats_720p.md).  Weight layout [tap][half][C / 8][4]: the 16 lanes of a ds_read_b128 group
// read 16 consecutive 16-byte slots (the [tap][C] layout put a lane's two reads 32 bytes apart: 2-way conflicts, SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.74).
template <typename T, int K>
__global__ __launch_bounds__(256) void depthwise_pool8_kernel(const T* __restrict__ in, const float* __restrict__ wgt,
                                                              const float* __restrict__ bias, T* __restrict__ out, int N, int H,
                                                              int W, int C, int k_rt) {
  extern __shared__ float wl[];                       // [k*k][2][C / 8][4]
  const int k = K > 0 ? K : k_rt;
  const int kk = k * k, c8 = C / 8;
  for (int i = threadIdx.x; i < kk * C; i += blockDim.x) {
    const int c = i / kk, t = i - c * kk;
    wl[((t * 2 + ((c >> 2) & 1)) * c8 + (c >> 3)) * 4 + (c & 3)] = wgt[i];
  }
  __syncthreads();
  const int OH = H / k, OW = W / k;
  const long long total = (long long)N * OH * OW * c8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const long long pix = i / c8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH);
    const long long n = pix / ((long long)OW * OH);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[cc * 8 + j] : 0.f;
    const T* base = in + ((n * H + oy * k) * (long long)W + ox * k) * C + cc * 8;
    if constexpr (K > 0) {
      float v[K * K][8];
#pragma unroll
      for (int t = 0; t < K * K; ++t) load8<T>(base + ((t / K) * (long long)W + (t % K)) * C, v[t]);
#pragma unroll
      for (int t = 0; t < K * K; ++t) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + ((t * 2) * c8 + cc) * 4), w1 = *reinterpret_cast<const f32x4*>(wl + ((t * 2 + 1) * c8 + cc) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] += w0[j] * v[t][j]; acc[4 + j] += w1[j] * v[t][4 + j]; }
      }
    } else {
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          float v[8];
          load8<T>(base + (ky * (long long)W + kx) * C, v);
          const int t = ky * k + kx;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + ((t * 2) * c8 + cc) * 4), w1 = *reinterpret_cast<const f32x4*>(wl + ((t * 2 + 1) * c8 + cc) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += w0[j] * v[j]; acc[4 + j] += w1[j] * v[4 + j]; }
        }
    }
    store8<T>(out + pix * C + cc * 8, acc);
  }
}

// InstanceNorm statistics, deterministic (no atomics: the flows feed discontinuous 'nearest' warps, so run-to-run
// bit differences are not acceptable).  Stage 1: grid (slice, n); a block streams its slice of pixels with 16-byte
// channel vectors (C/8 lanes per pixel, 256/(C/8) pixels per pass), reduces the per-thread partials through LDS in a
// fixed order and writes part[n][slice][c] = (sum, sumsq).  Stage 2: one thread per (n, c) adds the slices in order
// and writes ws[n][c] = (mean, rstd).
template <typename T>
__global__ __launch_bounds__(256) void inorm_stats_kernel(const T* __restrict__ in, float* __restrict__ part, int HW, int C,
                                                          int slices) {
  __shared__ float red[256][17];
  const int lpp = C / 8;                               // lanes per pixel
  const int ppp = 256 / lpp;                           // pixels per pass
  const int tid = threadIdx.x;
  const int n = blockIdx.y, sl = blockIdx.x;
  const int per = (HW + slices - 1) / slices;
  const int p0 = sl * per, p1 = min(HW, p0 + per);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  const int cc = tid % lpp, pl = tid / lpp;
  if (pl < ppp) {
    for (int p = p0 + pl; p < p1; p += ppp) {
      float v[8];
      load8<T>(in + ((long long)n * HW + p) * C + cc * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] += v[j] * v[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid][j] = s[j]; red[tid][8 + j] = q[j]; }
  __syncthreads();
  if (tid < C) {
    const int c8 = tid / 8, j = tid % 8;
    float ts = 0.f, tq = 0.f;
    for (int r = 0; r < ppp; ++r) { ts += red[r * lpp + c8][j]; tq += red[r * lpp + c8][8 + j]; }
    float* o = part + (((long long)n * slices + sl) * C + tid) * 2;
    o[0] = ts; o[1] = tq;
  }
}

__global__ void inorm_finalize_kernel(const float* __restrict__ part, float* __restrict__ ws, int N, int C, int slices, int HW,
                                      float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  float s = 0.f, q = 0.f;
  for (int sl = 0; sl < slices; ++sl) {
    const float* o = part + (((long long)n * slices + sl) * C + c) * 2;
    s += o[0]; q += o[1];
  }
  const float mean = s / (float)HW;
  const float var = fmaxf(q / (float)HW - mean * mean, 0.f);
  ws[(long long)i * 2] = mean;
  ws[(long long)i * 2 + 1] = rsqrtf(var + eps);
}

template <typename T>
__global__ void inorm_apply_kernel(const T* __restrict__ in, const float* __restrict__ ws, T* __restrict__ out, long long N,
                                   int HW, int C, int relu) {
  const long long total = N * HW * (C / 8);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % (C / 8));
    const long long pix = i / (C / 8);
    const long long n = pix / HW;
    float v[8];
    load8<T>(in + pix * C + cc * 8, v);
    const float4* st = reinterpret_cast<const float4*>(ws + (n * C + cc * 8) * 2);   // (mean, rstd) x 8, 16-byte aligned
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 m = st[j];
      const float r0 = (v[2 * j] - m.x) * m.y, r1 = (v[2 * j + 1] - m.z) * m.w;
      v[2 * j] = relu ? fmaxf(r0, 0.f) : r0;
      v[2 * j + 1] = relu ? fmaxf(r1, 0.f) : r1;
    }
    store8<T>(out + pix * C + cc * 8, v);
  }
}

// InstanceNorm apply of the split-plane ("f16x3") RAFT feature encoder: fp32 convolution output in, split-plane fp16 out, with the
// residual tail of a ResidualBlock fused (RAFT/extractor.py:44-57): out = relu2(relu(IN(in)) + res), res split-plane or none.
__global__ void inorm_apply_split_kernel(const float* __restrict__ in, const float* __restrict__ ws, _Float16* __restrict__ out,
                                         const _Float16* __restrict__ res, int res_cs, int res_co, long long N, int HW, int C, int relu,
                                         int relu2) {
  const long long total = N * HW * (C / 8);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % (C / 8));
    const long long pix = i / (C / 8);
    const long long n = pix / HW;
    float v[8];
    load8<float>(in + pix * C + cc * 8, v);
    const float4* st = reinterpret_cast<const float4*>(ws + (n * C + cc * 8) * 2);   // (mean, rstd) x 8, 16-byte aligned
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 m = st[j];
      const float r0 = (v[2 * j] - m.x) * m.y, r1 = (v[2 * j + 1] - m.z) * m.w;
      v[2 * j] = relu ? relu_split(r0) : r0;
      v[2 * j + 1] = relu ? relu_split(r1) : r1;
    }
    if (res != nullptr) {
      float r[8];
      const _Float16* rp = res + pix * res_cs + res_co + cc * 8;
      load8_split(rp, rp + res_cs / 2, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    if (relu2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_split(v[j]);
    }
    _Float16* op = out + pix * (2 * C) + cc * 8;
    store8_split(op, op + C, v);
  }
}

// bilinear x2, align_corners=True: src = dst * (in-1)/(out-1).  One block per PAIR of output rows (n, 2r), (n, 2r + 1) -- their source
// rows and vertical weights are block-uniform, and with a scale just under 1/2 the two rows read the same source row pair (or the
// second starts at the first's lower row), so the four 16-byte vectors fetched for row 2r serve row 2r + 1 as well: half the loads through
// L1 per output.  32-bit index arithmetic inside the row (the grid-stride form spent its time in 64-bit divisions: 2.6 TB/s).  Every
// output is computed with the same expression as before: identical results.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C) {
  const int OH = 2 * H, OW = 2 * W;
  const int cch = C / 8;
  const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int pairs = N * H, per_row = OW * cch;
  for (int pr = blockIdx.x; pr < pairs; pr += gridDim.x) {
    const int n = pr / H, oya = 2 * (pr - n * H), oyb = oya + 1;
    const float fya = sy * (float)oya, fyb = sy * (float)oyb;
    const int y0a = (int)fya, y0b = (int)fyb;
    const int y1a = min(y0a + 1, H - 1), y1b = min(y0b + 1, H - 1);
    const float lya = fya - (float)y0a, lyb = fyb - (float)y0b;
    const T* r0 = in + ((long long)n * H + y0a) * W * C;
    const T* r1 = in + ((long long)n * H + y1a) * W * C;
    const T* r2 = in + ((long long)n * H + y1b) * W * C;
    const bool same = y0b == y0a;                        // block-uniform; otherwise y0b == y1a (the scale is < 1/2)
    T* orow = out + ((long long)n * OH + oya) * OW * C;
    for (int i = threadIdx.x; i < per_row; i += 256) {
      const int ox = i / cch, cc = i - ox * cch;
      const float fx = sx * (float)ox;
      const int x0 = (int)fx;
      const int x1 = min(x0 + 1, W - 1);
      const float lx = fx - (float)x0;
      float a[8], b[8], c[8], d[8], o[8];
      load8<T>(r0 + x0 * C + cc * 8, a);
      load8<T>(r0 + x1 * C + cc * 8, b);
      load8<T>(r1 + x0 * C + cc * 8, c);
      load8<T>(r1 + x1 * C + cc * 8, d);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = (1.f - lya) * ((1.f - lx) * a[j] + lx * b[j]) + lya * ((1.f - lx) * c[j] + lx * d[j]);
      store8<T>(orow + i * 8, o);
      if (same) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = (1.f - lyb) * ((1.f - lx) * a[j] + lx * b[j]) + lyb * ((1.f - lx) * c[j] + lx * d[j]);
      } else {
        load8<T>(r2 + x0 * C + cc * 8, a);
        load8<T>(r2 + x1 * C + cc * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = (1.f - lyb) * ((1.f - lx) * c[j] + lx * d[j]) + lyb * ((1.f - lx) * a[j] + lx * b[j]);
      }
      store8<T>(orow + (long long)OW * C + i * 8, o);
    }
  }
}

template <typename T>
__global__ void dcn_offmask_act_kernel(T* __restrict__ om, int cs, const T* __restrict__ flow, int fcs, int fco, float mag,
                                       long long npix) {
  const long long total = npix * 54;   // 432 / 8
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % 54);
    const long long pix = i / 54;
    float v[8];
    T* p = om + pix * cs + cc * 8;
    load8<T>(p, v);
    if (cc < 36) {
      float fx = 0.f, fy = 0.f;
      if (flow) { fx = to_f32(flow[pix * fcs + fco]); fy = to_f32(flow[pix * fcs + fco + 1]); }
      // offset + flow.flip(1).repeat(...): even channels (dy) get flow_y, odd channels (dx) get flow_x
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mag * tanhf(v[j]) + ((j & 1) ? fx : fy);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 1.f / (1.f + __expf(-v[j]));
    }
    store8<T>(p, v);
  }
}

template <typename T>
__global__ void gru_gate_kernel(const T* __restrict__ zr, int zcs, const T* __restrict__ h, int hcs, int hco,
                                const T* __restrict__ q, int qcs, T* __restrict__ out, int ocs, int oco, long long npix, int C,
                                int mode) {
  const int cch = C / 8;
  const long long total = npix * cch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cch);
    const long long pix = i / cch;
    float a[8], hv[8], o[8];
    load8<T>(h + pix * hcs + hco + cc * 8, hv);
    if (mode == 0) {
      load8<T>(zr + pix * zcs + C + cc * 8, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = a[j] * hv[j];
    } else {
      float qv[8];
      load8<T>(zr + pix * zcs + cc * 8, a);
      load8<T>(q + pix * qcs + cc * 8, qv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (1.f - a[j]) * hv[j] + a[j] * qv[j];
    }
    store8<T>(out + pix * ocs + oco + cc * 8, o);
  }
}

// planar -> pixel-major with dtype conversion; one thread per (pixel, channel), pixel fastest on the read side.
template <typename TI, typename TO>
__global__ void nchw_to_nhwc_kernel(const TI* __restrict__ in, TO* __restrict__ out, int ocs, int oco, int N, int C, int HW,
                                    float scale) {
  const long long total = (long long)N * C * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const long long n = i / ((long long)HW * C);
    out[(n * HW + p) * ocs + oco + c] = from_f32<TO>(to_f32(in[i]) * scale);
  }
}
// fp32 planar -> split-plane NHWC (PP_F16S): hi at channel c, lo at c + ocs / 2
__global__ void nchw_to_nhwc_split_kernel(const float* __restrict__ in, _Float16* __restrict__ out, int ocs, int oco, int N, int C, int HW,
                                          float scale) {
  const long long total = (long long)N * C * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const long long n = i / ((long long)HW * C);
    const float v = in[i] * scale;
    const _Float16 hi = (_Float16)v;
    out[(n * HW + p) * ocs + oco + c] = hi;
    out[(n * HW + p) * ocs + ocs / 2 + oco + c] = (_Float16)(v - (float)hi);
  }
}
// Up to three planar sources [N,c_i,H,W] -> ONE NHWC row of 8 channels per pixel (missing channels zero): the encoder input
// cat(frame, mask, updated mask) of model/propainter.py:334-336.  One thread per pixel: coalesced plane reads, one 16-byte (fp16) /
// two 16-byte (fp32) row stores -- three pp_nchw_to_nhwc calls into the same buffer write 2 of every 16 bytes each (three partial-sector
// passes over the 236 MB buffer of a 16-frame 720p chunk: 300 us per call).
template <typename T>
__global__ __launch_bounds__(256) void pack_nhwc8_kernel(const T* __restrict__ a, int ca, const T* __restrict__ b, int cb, const T* __restrict__ c,
                                                         int cc, T* __restrict__ out, long long N, int HW) {
  const long long total = N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int p = (int)(i - n * HW);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < ca) v[j] = to_f32(a[(n * ca + j) * HW + p]);
      else if (j < ca + cb) v[j] = to_f32(b[(n * cb + (j - ca)) * HW + p]);
      else if (j < ca + cb + cc) v[j] = to_f32(c[(n * cc + (j - ca - cb)) * HW + p]);
    }
    store8<T>(out + i * 8, v);
  }
}

// The same packer for the split-plane engine (PP_F16S): fp32 planar sources -> [8 ch hi | 8 ch lo] fp16 per pixel, hi = fp16(v), lo = fp16(v - hi)
// (pp_nchw_to_nhwc's split arithmetic), one 32-byte row store per pixel instead of two 2-byte stores per (pixel, channel).
__global__ __launch_bounds__(256) void pack_nhwc8_split_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                                               const float* __restrict__ c, int cc, _Float16* __restrict__ out, long long N, int HW) {
  const long long total = N * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / HW;
    const int p = (int)(i - n * HW);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < ca) v[j] = a[(n * ca + j) * HW + p];
      else if (j < ca + cb) v[j] = b[(n * cb + (j - ca)) * HW + p];
      else if (j < ca + cb + cc) v[j] = c[(n * cc + (j - ca - cb)) * HW + p];
    }
    store8_split(out + i * 16, out + i * 16 + 8, v);
  }
}

template <typename TI, typename TO>
__global__ void nhwc_to_nchw_kernel(const TI* __restrict__ in, int ics, int ico, TO* __restrict__ out, int N, int C, int HW,
                                    int act) {
  const long long total = (long long)N * C * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const long long n = i / ((long long)HW * C);
    out[i] = from_f32<TO>(apply_act(to_f32(in[(n * HW + p) * ics + ico + c]), act, 0.f));
  }
}

}  // namespace pp

using namespace pp;

#define PP_DISPATCH_T(dtype, ...)                          \
  if ((dtype) == PP_F16) { using T = _Float16; __VA_ARGS__ } \
  else { using T = float; __VA_ARGS__ }

extern "C" int pp_fold_tokens(const void* tokens, void* out, int BT, int fh, int fw, int C, int H, int W, int normalize,
                              int act, int dtype, void* stream) {
  PP_REQUIRE(tokens && out && BT > 0 && fh > 0 && fw > 0 && C > 0 && H > 0 && W > 0, PP_ERR_ARG, "pp_fold_tokens: bad arguments");
  PP_REQUIRE(C % 8 == 0 && (uintptr_t)tokens % 16 == 0 && (uintptr_t)out % 16 == 0, PP_ERR_ALIGN,
             "pp_fold_tokens: C=%d must be a multiple of 8 and the buffers 16-byte aligned", C);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_fold_tokens: dtype %d", dtype);
  PP_REQUIRE(fh == (H + 6 - 7) / 3 + 1 && fw == (W + 6 - 7) / 3 + 1, PP_ERR_ARG, "pp_fold_tokens: token grid %dx%d does not match %dx%d", fh, fw, H, W);
  const int g = grid_for((long long)BT * H * W * (C / 8));
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((fold_tokens_kernel<T>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)tokens,
                                          (T*)out, BT, fh, fw, C, H, W, normalize, act);)
  return launch_status("pp_fold_tokens");
}

extern "C" int pp_layernorm(const void* in, const float* gamma, const float* beta, void* out, int64_t rows, int C, float eps,
                            int dtype, void* stream) {
  PP_REQUIRE(in && gamma && beta && out && rows > 0, PP_ERR_ARG, "pp_layernorm: bad arguments");
  PP_REQUIRE(C == 512, PP_ERR_ARG, "pp_layernorm: C=%d (only 512 is instantiated)", C);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_layernorm: dtype %d", dtype);
  const unsigned g = (unsigned)((rows + 3) / 4);
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((layernorm_kernel<T, 8>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)in, gamma,
                                          beta, (T*)out, (long long)rows, C, eps, 0, 0, 0, 0);)
  return launch_status("pp_layernorm");
}

extern "C" int pp_layernorm_grid(const void* in, const float* gamma, const float* beta, void* out, int N, int gh, int gw, int Hp, int Wp,
                                 int C, float eps, int dtype, void* stream) {
  PP_REQUIRE(in && gamma && beta && out && N > 0 && gh > 0 && gw > 0 && Hp >= gh && Wp >= gw, PP_ERR_ARG, "pp_layernorm_grid: bad arguments");
  PP_REQUIRE(C == 512, PP_ERR_ARG, "pp_layernorm_grid: C=%d (only 512 is instantiated)", C);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_layernorm_grid: dtype %d", dtype);
  const long long rows = (long long)N * gh * gw;
  const unsigned g = (unsigned)((rows + 3) / 4);
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((layernorm_kernel<T, 8>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)in, gamma,
                                          beta, (T*)out, rows, C, eps, gh, gw, Hp, Wp);)
  return launch_status("pp_layernorm_grid");
}

extern "C" int pp_depthwise_pool(const void* in, const float* weight, const float* bias, void* out, int N, int H, int W, int C,
                                 int k, int dtype, void* stream) {
  PP_REQUIRE(in && weight && out && N > 0 && H >= k && W >= k && C > 0 && k > 0, PP_ERR_ARG, "pp_depthwise_pool: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_depthwise_pool: dtype %d", dtype);
  if (C % 8 == 0 && (size_t)k * k * C * sizeof(float) <= 48 * 1024 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0) {
    int g8 = grid_for((long long)N * (H / k) * (W / k) * (C / 8));
    if (g8 > 768) g8 = 768;                           // every block stages the weights once: keep the blocks long-lived (3 per CU: 256 / 512 / 768 / 1024 blocks = 50 / 41 / 37 / 45 us at 720p)
    const size_t shm = (size_t)k * k * C * sizeof(float);
    if (k == 4) {
      PP_DISPATCH_T(dtype, hipLaunchKernelGGL((depthwise_pool8_kernel<T, 4>), dim3(g8), dim3(256), shm, (hipStream_t)stream, (const T*)in,
                                              weight, bias, (T*)out, N, H, W, C, k);)
    } else {
      PP_DISPATCH_T(dtype, hipLaunchKernelGGL((depthwise_pool8_kernel<T, 0>), dim3(g8), dim3(256), shm, (hipStream_t)stream, (const T*)in,
                                              weight, bias, (T*)out, N, H, W, C, k);)
    }
    return launch_status("pp_depthwise_pool");
  }
  const int g = grid_for((long long)N * (H / k) * (W / k) * C);
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((depthwise_pool_kernel<T>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)in,
                                          weight, bias, (T*)out, N, H, W, C, k);)
  return launch_status("pp_depthwise_pool");
}

static int inorm_slices(int HW) {
  int slices = HW / 2048; if (slices < 1) slices = 1; if (slices > 256) slices = 256;
  return slices;
}

extern "C" int64_t pp_instance_norm_workspace_floats(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return PP_ERR_ARG;
  return (int64_t)N * C * 2 * (1 + inorm_slices(H * W));
}

extern "C" int pp_instance_norm(const void* in, void* out, float* stats_ws, int N, int H, int W, int C, float eps, int relu,
                                int dtype, void* stream) {
  PP_REQUIRE(in && out && stats_ws && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 256, PP_ERR_ARG,
             "pp_instance_norm: bad arguments (C=%d must be a multiple of 8, <= 256)", C);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_instance_norm: dtype %d", dtype);
  PP_REQUIRE((uintptr_t)stats_ws % 16 == 0, PP_ERR_ALIGN, "pp_instance_norm: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int slices = inorm_slices(HW);
  float* part = stats_ws + (size_t)N * C * 2;           // [N][slices][C][2] after the final (mean, rstd) table
  PP_DISPATCH_T(dtype,
                hipLaunchKernelGGL((inorm_stats_kernel<T>), dim3(slices, N), dim3(256), 0, st, (const T*)in, part, HW, C, slices);
                hipLaunchKernelGGL(inorm_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float*)part, stats_ws,
                                   N, C, slices, HW, eps);
                hipLaunchKernelGGL((inorm_apply_kernel<T>), dim3(grid_for((long long)N * HW * (C / 8))), dim3(256), 0, st,
                                   (const T*)in, (const float*)stats_ws, (T*)out, (long long)N, HW, C, relu);)
  return launch_status("pp_instance_norm");
}

extern "C" int pp_instance_norm_split(const float* in, void* out, float* stats_ws, int N, int H, int W, int C, float eps, int relu,
                                      const void* residual, int res_cstride, int res_choff, int relu2, void* stream) {
  PP_REQUIRE(in && out && stats_ws && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 256, PP_ERR_ARG,
             "pp_instance_norm_split: bad arguments (C=%d must be a multiple of 8, <= 256)", C);
  PP_REQUIRE((uintptr_t)stats_ws % 16 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0, PP_ERR_ALIGN,
             "pp_instance_norm_split: buffers must be 16-byte aligned");
  PP_REQUIRE(residual == nullptr || ((uintptr_t)residual % 16 == 0 && res_cstride % 16 == 0 && res_choff % 8 == 0 && res_choff >= 0 &&
                                     res_cstride / 2 >= res_choff + C),
             PP_ERR_ALIGN, "pp_instance_norm_split: split-plane residual window (cstride %d, choff %d)", res_cstride, res_choff);
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int slices = inorm_slices(HW);
  float* part = stats_ws + (size_t)N * C * 2;
  hipLaunchKernelGGL((inorm_stats_kernel<float>), dim3(slices, N), dim3(256), 0, st, in, part, HW, C, slices);
  hipLaunchKernelGGL(inorm_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float*)part, stats_ws, N, C, slices, HW, eps);
  hipLaunchKernelGGL(inorm_apply_split_kernel, dim3(grid_for((long long)N * HW * (C / 8))), dim3(256), 0, st, in, (const float*)stats_ws,
                     (_Float16*)out, (const _Float16*)residual, res_cstride, res_choff, (long long)N, HW, C, relu, relu2);
  return launch_status("pp_instance_norm_split");
}

extern "C" int pp_upsample2x(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream) {
  PP_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, PP_ERR_ARG, "pp_upsample2x: bad arguments (C=%d)", C);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_upsample2x: dtype %d", dtype);
  PP_REQUIRE((long long)N * 2 * H < (1ll << 31) && (long long)2 * W * C < (1ll << 31), PP_ERR_ARG, "pp_upsample2x: %d x %d x %d x %d too large", N, H, W, C);
  const long long rows = (long long)N * H;
  const int g = (int)(rows < 256 * 64 ? rows : 256 * 64);      // one block per pair of output rows, grid-stride beyond 64 blocks per CU
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2x_kernel<T>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out,
                                          N, H, W, C);)
  return launch_status("pp_upsample2x");
}

extern "C" int pp_dcn_offset_mask_act(void* offmask, int cstride, const void* flow, int fl_cstride, int fl_choff, float mag,
                                      int64_t npix, int dtype, void* stream) {
  PP_REQUIRE(offmask && npix > 0 && cstride >= 432 && cstride % 8 == 0, PP_ERR_ARG, "pp_dcn_offset_mask_act: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_dcn_offset_mask_act: dtype %d", dtype);
  const int g = grid_for((long long)npix * 54);
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((dcn_offmask_act_kernel<T>), dim3(g), dim3(256), 0, (hipStream_t)stream, (T*)offmask,
                                          cstride, (const T*)flow, fl_cstride, fl_choff, mag, (long long)npix);)
  return launch_status("pp_dcn_offset_mask_act");
}

extern "C" int pp_gru_gate(const void* zr, int zr_cstride, const void* h, int h_cstride, int h_choff, const void* q,
                           int q_cstride, void* out, int out_cstride, int out_choff, int64_t npix, int C, int mode, int dtype,
                           void* stream) {
  PP_REQUIRE(zr && h && out && npix > 0 && C > 0 && C % 8 == 0 && (mode == 0 || (mode == 1 && q)), PP_ERR_ARG,
             "pp_gru_gate: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_gru_gate: dtype %d", dtype);
  const int g = grid_for((long long)npix * (C / 8));
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((gru_gate_kernel<T>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)zr, zr_cstride,
                                          (const T*)h, h_cstride, h_choff, (const T*)q, q_cstride, (T*)out, out_cstride, out_choff,
                                          (long long)npix, C, mode);)
  return launch_status("pp_gru_gate");
}

#define PP_DISPATCH_2(di, dout, ...)                                                     \
  if ((di) == PP_F16 && (dout) == PP_F16) { using TI = _Float16; using TO = _Float16; __VA_ARGS__ } \
  else if ((di) == PP_F16) { using TI = _Float16; using TO = float; __VA_ARGS__ }            \
  else if ((dout) == PP_F16) { using TI = float; using TO = _Float16; __VA_ARGS__ }          \
  else { using TI = float; using TO = float; __VA_ARGS__ }

extern "C" int pp_nchw_to_nhwc(const void* in, int in_dtype, void* out, int out_dtype, int out_cstride, int out_choff, int N,
                               int C, int H, int W, float scale, void* stream) {
  PP_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && out_cstride >= C, PP_ERR_ARG, "pp_nchw_to_nhwc: bad arguments");
  const int g = grid_for((long long)N * C * H * W);
  if (out_dtype == PP_F16S) {      // split-plane output (lo plane at out_cstride / 2), fp32 input
    PP_REQUIRE(in_dtype == PP_F32 && out_cstride % 2 == 0 && out_cstride / 2 >= out_choff + C, PP_ERR_ARG,
               "pp_nchw_to_nhwc: split-plane output needs fp32 input and an even cstride with cstride / 2 >= choff + C");
    hipLaunchKernelGGL(nchw_to_nhwc_split_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)in, (_Float16*)out, out_cstride,
                       out_choff, N, C, H * W, scale);
    return launch_status("pp_nchw_to_nhwc");
  }
  PP_REQUIRE((in_dtype | out_dtype) <= 1 && in_dtype >= 0 && out_dtype >= 0, PP_ERR_DTYPE, "pp_nchw_to_nhwc: dtype");
  PP_DISPATCH_2(in_dtype, out_dtype,
                hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, TO>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const TI*)in, (TO*)out,
                                   out_cstride, out_choff, N, C, H * W, scale);)
  return launch_status("pp_nchw_to_nhwc");
}

extern "C" int pp_pack_nhwc8(const void* in0, int c0, const void* in1, int c1, const void* in2, int c2, void* out, int N, int H, int W,
                             int dtype, void* stream) {
  PP_REQUIRE(in0 && out && N > 0 && H > 0 && W > 0 && c0 > 0 && c1 >= 0 && c2 >= 0 && c0 + c1 + c2 <= 8, PP_ERR_ARG,
             "pp_pack_nhwc8: bad arguments (%d + %d + %d channels must be 1..8)", c0, c1, c2);
  PP_REQUIRE((c1 == 0 || in1) && (c2 == 0 || in2), PP_ERR_ARG, "pp_pack_nhwc8: a source with channels needs a pointer");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16 || dtype == PP_F16S, PP_ERR_DTYPE, "pp_pack_nhwc8: dtype %d", dtype);
  PP_REQUIRE((uintptr_t)out % 16 == 0, PP_ERR_ALIGN, "pp_pack_nhwc8: out must be 16-byte aligned");
  const int g = grid_for((long long)N * H * W);
  if (dtype == PP_F16S) {      // fp32 sources -> split-plane rows [N,H,W,8 hi | 8 lo]
    hipLaunchKernelGGL(pack_nhwc8_split_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)in0, c0, (const float*)in1, c1,
                       (const float*)in2, c2, (_Float16*)out, (long long)N, H * W);
    return launch_status("pp_pack_nhwc8");
  }
  PP_DISPATCH_T(dtype, hipLaunchKernelGGL((pack_nhwc8_kernel<T>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const T*)in0, c0, (const T*)in1, c1,
                                          (const T*)in2, c2, (T*)out, (long long)N, H * W);)
  return launch_status("pp_pack_nhwc8");
}

extern "C" int pp_nhwc_to_nchw(const void* in, int in_dtype, int in_cstride, int in_choff, void* out, int out_dtype, int N, int C,
                               int H, int W, int act, void* stream) {
  PP_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && in_cstride >= C, PP_ERR_ARG, "pp_nhwc_to_nchw: bad arguments");
  PP_REQUIRE((in_dtype | out_dtype) <= 1 && in_dtype >= 0 && out_dtype >= 0, PP_ERR_DTYPE, "pp_nhwc_to_nchw: dtype");
  const int g = grid_for((long long)N * C * H * W);
  PP_DISPATCH_2(in_dtype, out_dtype,
                hipLaunchKernelGGL((nhwc_to_nchw_kernel<TI, TO>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const TI*)in, in_cstride,
                                   in_choff, (TO*)out, N, C, H * W, act);)
  return launch_status("pp_nhwc_to_nchw");
}
