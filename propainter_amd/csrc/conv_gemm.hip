// Implicit-GEMM convolution on MFMA for gfx950 (wave64).  One kernel family serves every Conv2d /
// Conv3d(1,k,k) / temporal Conv3d(3,1,1) / Linear / batched GEMM / modulated deformable conv on the
// ProPainter path (see include/propainter_hip.h, pp_conv2d).
//
// GEMM view per group:  D[co][px] = sum_k Wp[co][k] * A[px][k],  px = output pixel, k = (tap, source,
// channel).  A is never materialised: each 16-byte K-chunk (8 channels) of an A row is gathered straight
// from the NHWC sources using a tiny device table {dy, dx, src, choff}; zero / replicate padding and the
// bilinear+modulation of the deformable conv happen on the way into LDS.
//
// Tiling: 256 threads = 4 waves; block tile BM pixels x BN couts x 32 k; register-prefetched global
// loads (issue tile t+1, compute tile t from LDS, then write t+1 into the other LDS buffer -> one
// barrier per k-step).  Weights are the MFMA "A" operand and pixels the "B" operand, so each lane ends
// up with 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
//   fp16: v_mfma_f32_16x16x32_f16 (8 k per lane);  fp32: v_mfma_f32_16x16x4_f32 (exact fp32).
// LDS rows are padded by 16 bytes (stride 80 B fp16 / 144 B fp32) to spread the ds_read_b128 lanes.
#include "conv_params.h"
#include <stdlib.h>

namespace pp {

template <typename T> struct Mma;
template <> struct Mma<_Float16> {
  static constexpr int LDK = 40;  // 32 + 8 pad elements
};
template <> struct Mma<float> {
  static constexpr int LDK = 36;  // 32 + 4 pad elements
};

// 8 consecutive elements of T as raw registers.
template <typename T> struct Chunk;
template <> struct Chunk<_Float16> { u32x4 v; };
template <> struct Chunk<float> { u32x4 v[2]; };

template <typename T> __device__ __forceinline__ Chunk<T> zero_chunk() {
  Chunk<T> c;
  if constexpr (sizeof(T) == 2) c.v = u32x4{0, 0, 0, 0};
  else { c.v[0] = u32x4{0, 0, 0, 0}; c.v[1] = u32x4{0, 0, 0, 0}; }
  return c;
}
template <typename T> __device__ __forceinline__ Chunk<T> load_chunk(const T* p) {
  Chunk<T> c;
  if constexpr (sizeof(T) == 2) c.v = *reinterpret_cast<const u32x4*>(p);
  else { c.v[0] = *reinterpret_cast<const u32x4*>(p); c.v[1] = *reinterpret_cast<const u32x4*>(p + 4); }
  return c;
}
template <typename T> __device__ __forceinline__ void store_chunk(T* p, const Chunk<T>& c) {
  if constexpr (sizeof(T) == 2) *reinterpret_cast<u32x4*>(p) = c.v;
  else { *reinterpret_cast<u32x4*>(p) = c.v[0]; *reinterpret_cast<u32x4*>(p + 4) = c.v[1]; }
}
template <typename T> __device__ __forceinline__ void chunk_to_f32(const Chunk<T>& c, float* f) {
  const T* e = reinterpret_cast<const T*>(&c);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = to_f32(e[i]);
}
template <typename T> __device__ __forceinline__ Chunk<T> chunk_from_f32(const float* f) {
  Chunk<T> c;
  T* e = reinterpret_cast<T*>(&c);
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = from_f32<T>(f[i]);
  return c;
}

// hi/lo fp16 split of 8 fp32 values into an LDS row (hi halves at byte chunk*16, lo halves at 64 + chunk*16)
__device__ __forceinline__ void store_split(char* row, int chunk, const Chunk<float>& c) {
  const float* f = reinterpret_cast<const float*>(&c);
  f16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (_Float16)f[i];
    l[i] = (_Float16)(f[i] - (float)h[i]);
  }
  *reinterpret_cast<f16x8*>(row + chunk * 16) = h;
  *reinterpret_cast<f16x8*>(row + 64 + chunk * 16) = l;
}
__device__ __forceinline__ void store_split(char*, int, const Chunk<_Float16>&) {}

// SPLIT (T = float only): the tensors stay fp32 in memory, every value is split on its way into LDS into
// hi = fp16(x) and lo = fp16(x - hi) (together 22 significand bits), and each product runs on the fp16 matrix cores as
// hi*hi + hi*lo + lo*hi with fp32 accumulation (the dropped lo*lo term is < 2^-22 relative): ~2^-21 per product instead
// of fp32's 2^-24, at 3 fp16 MFMAs (48 cycles per 32-deep k step and 16x16 tile) instead of 8 fp32 MFMAs (256 cycles).
// An LDS row holds the 32 hi halves (64 B), then the 32 lo halves (64 B), then 16 B of padding: the fp32 row size.
// Values beyond the fp16 range (|x| > 65504) are not representable by the split; the RAFT activations it serves are O(10).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool DEFORM, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
  static_assert(!SPLIT || (sizeof(T) == 4 && !DEFORM), "split mode: plain fp32 convolutions");
  constexpr int LDK = Mma<T>::LDK;
  constexpr int A_ROWS = BM / 64;                 // A rows gathered per thread (4 chunks per row)
  constexpr int B_ROWS = (BN + 63) / 64;          // weight rows per thread
  constexpr int WPX = BM / WAVES_M;               // pixels per wave
  constexpr int WCO = BN / WAVES_N;               // couts per wave
  constexpr int TM = WPX / 16, TN = WCO / 16;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");

  __shared__ __attribute__((aligned(16))) T lds[2 * (BM + BN) * LDK];
  constexpr int STAGE = (BM + BN) * LDK;   // A tile followed by the weight tile, two stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int g = blockIdx.z;
  // XCD-aware tile order (as the LDS-DMA kernels): consecutive block ids go round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of M tiles -- the deformable gather of a tile then finds its neighbourhood in that XCD's L2
  // (PMC: 1.2 GB of fabric traffic per 79 MB-algorithmic launch without this).
  int bx = blockIdx.x;
  if (gridDim.y == 1 && gridDim.z == 1) {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bx & 7, loc = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const long long m0 = (long long)bx * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread gather state: A rows (tid>>2) + 64*i, chunk (tid&3) of every 32-wide k step
  const int chunk = tid & 3;
  const int arow0 = tid >> 2;
  int a_iy0[A_ROWS], a_ix0[A_ROWS];
  long long a_pix[A_ROWS];   // n*H*W, or -1 when the row is past M
  long long a_m[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    long long m = m0 + arow0 + 64 * i;
    a_m[i] = m;
    if (m < p.M) {
      int ox = (int)(m % p.OW);
      long long r = m / p.OW;
      int oy = (int)(r % p.OH);
      long long n = r / p.OH;
      a_iy0[i] = oy * p.sh - p.ph;
      a_ix0[i] = ox * p.sw - p.pw;
      a_pix[i] = n * (long long)p.H * p.W;
    } else {
      a_iy0[i] = 0; a_ix0[i] = 0; a_pix[i] = -1;
    }
  }
  const T* wbase = reinterpret_cast<const T*>(p.weight) + (long long)g * p.weight_gstride;
  const long long K = (long long)p.kchunks * 8;

  Chunk<T> a_reg[A_ROWS][DEFORM ? 4 : 1];
  float a_w[A_ROWS][DEFORM ? 4 : 1];   // deform: corner weights (already times modulation mask)
  Chunk<T> b_reg[B_ROWS];

  // Per-source base / stride / channel base in registers.  (Selecting `p.src[s]` with the per-lane source id made hipcc
  // fetch the kernel-argument fields with VECTOR loads inside the K loop -- a dependent load and a vmcnt(0) drain in
  // front of every gather; the same pathology cost the attention kernel 5 k cycles per tile.)
  const T* sp_[PP_CONV_MAX_SRC];
  int cs_[PP_CONV_MAX_SRC], cb_[PP_CONV_MAX_SRC];
#pragma unroll
  for (int i = 0; i < PP_CONV_MAX_SRC; ++i) {
    sp_[i] = reinterpret_cast<const T*>(p.src[i].ptr) + (long long)g * p.src_gstride;
    cs_[i] = p.src[i].cstride;
    cb_[i] = p.src[i].choff + g * p.src[i].cgroup;
  }
  const int imgH = p.H, imgW = p.W, pad_mode = p.pad_mode;

  auto issue_loads = [&](int ks) {
    const int4 e = p.ktable[ks * 4 + chunk];
    const int s = e.z & 0xff;
    int cstride = 0, cbase = 0;
    const T* sp = nullptr;
    if (s != 255) {      // value selects over the (<= 4) sources: v_cndmask, no memory access
      sp = s == 0 ? sp_[0] : s == 1 ? sp_[1] : s == 2 ? sp_[2] : sp_[3];
      cstride = s == 0 ? cs_[0] : s == 1 ? cs_[1] : s == 2 ? cs_[2] : cs_[3];
      cbase = s == 0 ? cb_[0] : s == 1 ? cb_[1] : s == 2 ? cb_[2] : cb_[3];
    }
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      if constexpr (!DEFORM) {
        a_reg[i][0] = zero_chunk<T>();
        if (sp != nullptr && a_pix[i] >= 0) {
          int iy = a_iy0[i] + e.x, ix = a_ix0[i] + e.y;
          bool ok = true;
          if (p.pad_mode == 1) {
            iy = min(max(iy, 0), p.H - 1);
            ix = min(max(ix, 0), p.W - 1);
          } else {
            ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
          }
          if (ok) a_reg[i][0] = load_chunk<T>(sp + (a_pix[i] + (long long)iy * p.W + ix) * cstride + cbase + e.w);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) { a_reg[i][c] = zero_chunk<T>(); a_w[i][c] = 0.f; }
        if (sp != nullptr && a_pix[i] >= 0) {
          const int grp = (e.z >> 8) & 0xff, tap = (e.z >> 16) & 0xff;
          // stride-1 deformable conv: the offset/mask pixel is the output pixel itself
          const T* om = reinterpret_cast<const T*>(p.dcn) + a_m[i] * p.dcn_cstride;
          const float dyv = to_f32(om[2 * (grp * 9 + tap)]);
          const float dxv = to_f32(om[2 * (grp * 9 + tap) + 1]);
          const float mk = to_f32(om[p.dcn_mask_off + grp * 9 + tap]);
          const float py = (float)(a_iy0[i] + e.x) + dyv;
          const float px = (float)(a_ix0[i] + e.y) + dxv;
          if (py > -1.f && py < (float)p.H && px > -1.f && px < (float)p.W) {
            const float fy = floorf(py), fx = floorf(px);
            const int y0 = (int)fy, x0 = (int)fx;
            const float ly = py - fy, lx = px - fx;
            const float wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
              if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
                a_reg[i][c] = load_chunk<T>(sp + (a_pix[i] + (long long)yy * p.W + xx) * cstride + cbase + e.w);
                a_w[i][c] = wy[c >> 1] * wx[c & 1] * mk;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int row = arow0 + 64 * i;
      b_reg[i] = zero_chunk<T>();
      if (row < BN && n0 + row < p.cout_pad)
        b_reg[i] = load_chunk<T>(wbase + (long long)(n0 + row) * K + (long long)ks * 32 + chunk * 8);
    }
  };

  auto commit_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      T* dst = lds + buf * STAGE + (arow0 + 64 * i) * LDK + chunk * 8;
      if constexpr (SPLIT) {
        store_split(reinterpret_cast<char*>(lds + buf * STAGE + (arow0 + 64 * i) * LDK), chunk, a_reg[i][0]);
      } else if constexpr (!DEFORM) {
        store_chunk<T>(dst, a_reg[i][0]);
      } else {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          chunk_to_f32<T>(a_reg[i][c], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += a_w[i][c] * f[j];
        }
        store_chunk<T>(dst, chunk_from_f32<T>(acc));
      }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int row = arow0 + 64 * i;
      if (row < BN) {
        if constexpr (SPLIT) store_split(reinterpret_cast<char*>(lds + buf * STAGE + (BM + row) * LDK), chunk, b_reg[i]);
        else store_chunk<T>(lds + buf * STAGE + (BM + row) * LDK + chunk * 8, b_reg[i]);
      }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.kchunks / 4;
  issue_loads(0);
  commit_lds(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) issue_loads(ks + 1);
    const T* Ab = lds + buf * STAGE + (wm * WPX + (lane & 15)) * LDK;
    const T* Bb = lds + buf * STAGE + (BM + wn * WCO + (lane & 15)) * LDK;
    if constexpr (sizeof(T) == 2) {
      f16x8 af[TM], bf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const f16x8*>(Ab + t * 16 * LDK + (lane >> 4) * 8);
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[t] = *reinterpret_cast<const f16x8*>(Bb + t * 16 * LDK + (lane >> 4) * 8);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[a], af[b], acc[a][b], 0, 0, 0);
    } else if constexpr (SPLIT) {
      const char* Ac = reinterpret_cast<const char*>(Ab) + (lane >> 4) * 16;
      const char* Bc = reinterpret_cast<const char*>(Bb) + (lane >> 4) * 16;
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        ah[t] = *reinterpret_cast<const f16x8*>(Ac + t * 16 * LDK * 4);
        al[t] = *reinterpret_cast<const f16x8*>(Ac + t * 16 * LDK * 4 + 64);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        bh[t] = *reinterpret_cast<const f16x8*>(Bc + t * 16 * LDK * 4);
        bl[t] = *reinterpret_cast<const f16x8*>(Bc + t * 16 * LDK * 4 + 64);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) {      // small terms first
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[a], ah[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[a], al[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[a], ah[b], acc[a][b], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float af[TM], bf[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = Ab[t * 16 * LDK + kk * 4 + (lane >> 4)];
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[t] = Bb[t * 16 * LDK + kk * 4 + (lane >> 4)];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[a], af[b], acc[a][b], 0, 0, 0);
      }
    }
    if (ks + 1 < nk) commit_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds couts (lane>>4)*4 + r (r = 0..3) of pixel (lane&15) of every 16x16 tile
  const int out_cbase = p.out_choff + g * p.out_cgroup;
  char* outp = p.out + (long long)g * p.out_gstride * (p.out_f16 ? 2 : 4);
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const long long m = m0 + wm * WPX + b * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const int co = n0 + wn * WCO + a * 16 + (lane >> 4) * 4;
      if (co >= p.cout_g) continue;
      float v[4];
      const bool late = p.preadd != nullptr || p.fuse != PP_FUSE_NONE;     // fused recurrent-cell epilogue (groups == 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[a][b][r];
        if (p.bias != nullptr && co + r < p.cout_g) x += p.bias[g * p.cout_g + co + r];
        x *= p.out_scale;
        if (late && p.preadd != nullptr && co + r < p.cout_g)
          x += to_f32(reinterpret_cast<const T*>(p.preadd)[m * p.preadd_cstride + p.preadd_choff + co + r]);
        v[r] = apply_act(x, p.act, p.act_param);
      }
      if (p.fuse == PP_FUSE_GRU_ZR && co >= p.fuse_split) {          // r half: r * h -> out2
        const int cr = co - p.fuse_split;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.cout_g) {
            const float hv = to_f32(reinterpret_cast<const T*>(p.fuse_a)[m * p.fuse_a_cstride + p.fuse_a_choff + cr + r]);
            reinterpret_cast<T*>(p.out2)[m * p.out2_cstride + p.out2_choff + cr + r] = from_f32<T>(v[r] * hv);
          }
        continue;
      }
      if (p.fuse == PP_FUSE_DCN_OFFMASK) {                          // offset / mask head (same math as dcn_offmask_act_kernel)
        float fx = 0.f, fy = 0.f;
        if (p.fuse_a != nullptr && co < p.fuse_split) {
          const T* fp_ = reinterpret_cast<const T*>(p.fuse_a) + m * p.fuse_a_cstride + p.fuse_a_choff;
          fx = to_f32(fp_[0]); fy = to_f32(fp_[1]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = co + r < p.fuse_split ? p.act_param * tanhf(v[r]) + ((r & 1) ? fx : fy) : 1.f / (1.f + __expf(-v[r]));
      }
      if (p.fuse == PP_FUSE_GRU_H) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.cout_g) {
            const float hv = to_f32(reinterpret_cast<const T*>(p.fuse_a)[m * p.fuse_a_cstride + p.fuse_a_choff + co + r]);
            const float zv = to_f32(reinterpret_cast<const T*>(p.fuse_b)[m * p.fuse_b_cstride + p.fuse_b_choff + co + r]);
            v[r] = (1.f - zv) * hv + zv * v[r];
          }
      }
      if (p.residual != nullptr) {
        const T* rp = reinterpret_cast<const T*>(p.residual) + m * p.res_cstride + p.res_choff + g * p.out_cgroup + co;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.cout_g) v[r] += to_f32(rp[r]);
      }
      if (p.act2 == PP_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      const long long oidx = m * p.out_cstride + out_cbase + co;
      const bool full = (co + 3 < p.cout_g);
      if (p.out_f16) {
        _Float16* op = reinterpret_cast<_Float16*>(outp) + oidx;
        if (full && ((oidx & 3) == 0)) {
          f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          *reinterpret_cast<f16x4*>(op) = h;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < p.cout_g) op[r] = (_Float16)v[r];
        }
      } else {
        float* op = reinterpret_cast<float*>(outp) + oidx;
        if (full && ((oidx & 3) == 0)) {
          *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < p.cout_g) op[r] = v[r];
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_conv(const ConvParams& p, int groups, bool deform, hipStream_t stream, bool split = false) {
  dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.cout_g + BN - 1) / BN), (unsigned)groups);
  if constexpr (sizeof(T) == 4) {
    if (split) {
      hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false, true>), grid, dim3(256), 0, stream, p);
      return launch_status("pp_conv2d(split)");
    }
  }
  if (deform)
    hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, true>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, false>), grid, dim3(256), 0, stream, p);
  return launch_status("pp_conv2d");
}

template <typename T> static int dispatch_conv(const ConvParams& p, int groups, bool deform, hipStream_t stream, bool split = false) {
  if (p.cout_g > 64) return launch_conv<T, 128, 128, 2, 2>(p, groups, deform, stream, split);
  if (p.cout_g > 32) return launch_conv<T, 128, 64, 2, 2>(p, groups, deform, stream, split);
  if (p.cout_g > 16) return launch_conv<T, 128, 32, 2, 2>(p, groups, deform, stream, split);
  return launch_conv<T, 128, 16, 4, 1>(p, groups, deform, stream, split);
}

}  // namespace pp

extern "C" int pp_conv_build_ktable(int ntaps, const int32_t* dy, const int32_t* dx, int nsrc,
                                    const int32_t* src_channels, int dcn_groups, int32_t* out, int out_capacity) {
  PP_REQUIRE(ntaps > 0 && nsrc > 0 && nsrc <= PP_CONV_MAX_SRC && dy && dx && src_channels, PP_ERR_ARG,
             "pp_conv_build_ktable: bad arguments");
  int ctotal = 0;
  for (int s = 0; s < nsrc; ++s) {
    PP_REQUIRE(src_channels[s] > 0 && src_channels[s] % 8 == 0, PP_ERR_ALIGN,
               "pp_conv_build_ktable: source %d has %d channels (must be a positive multiple of 8)", s, src_channels[s]);
    ctotal += src_channels[s];
  }
  PP_REQUIRE(dcn_groups == 0 || (ctotal % dcn_groups == 0 && (ctotal / dcn_groups) % 8 == 0), PP_ERR_ARG,
             "pp_conv_build_ktable: %d channels do not split into %d offset groups of 8n", ctotal, dcn_groups);
  const int chunks = ntaps * (ctotal / 8);
  const int padded = (chunks + 7) / 8 * 8;
  if (out == nullptr) return padded;
  PP_REQUIRE(out_capacity >= padded + 1, PP_ERR_WORKSPACE, "pp_conv_build_ktable: need %d entries, got %d", padded + 1, out_capacity);
  int k = 0;
  if (dcn_groups) {
    // deformable sampling: GROUP-BLOCK-major.  A block of 32 consecutive channels (4 offset groups of 8 channels, or 2 of 16)
    // runs through all taps before the next block starts: k = ((block * ntaps + tap) * 4 + slot) * 8 + c.  The patch-staged
    // kernel (conv_dcn.hip) stages the 32-channel input patch of a tile once per block and samples all taps from LDS; every
    // chunk still carries its own offset group / tap id, so any kernel can walk the table in order.
    const int cg = ctotal / dcn_groups;                    // channels per offset group (multiple of 8)
    int src_base[PP_CONV_MAX_SRC], acc_c = 0;
    for (int s = 0; s < nsrc; ++s) { src_base[s] = acc_c; acc_c += src_channels[s]; }
    for (int cb = 0; cb < ctotal; cb += 32)
      for (int t = 0; t < ntaps; ++t)
        for (int cglobal = cb; cglobal < cb + 32 && cglobal < ctotal; cglobal += 8, ++k) {
          int s = 0;
          while (s + 1 < nsrc && cglobal >= src_base[s + 1]) ++s;
          out[4 * k + 0] = dy[t];
          out[4 * k + 1] = dx[t];
          out[4 * k + 2] = s | ((cglobal / cg) << 8) | (t << 16);
          out[4 * k + 3] = cglobal - src_base[s];
        }
  } else {
    // channel-block-major: (source, 64-channel block, tap, 8-channel chunk).  All taps of one 128-byte channel block
    // are consecutive K steps, so the pixels a tile re-reads for its neighbouring taps are still in the XCD's L2
    // (the live set per K step is tiles x 128 px x 128 B ~ 1 MB per XCD); with the tap-major order every tap pass
    // streamed the full pixel (all sources, all channels) of the tile's halo through L2 (> 4 MB per XCD at the RAFT /
    // generator sizes), and the gather ran at Infinity-Fabric instead of L2 speed.
    for (int s = 0; s < nsrc; ++s)
      for (int cb = 0; cb < src_channels[s]; cb += 64) {
        const int ce = cb + 64 < src_channels[s] ? cb + 64 : src_channels[s];
        for (int t = 0; t < ntaps; ++t)
          for (int c = cb; c < ce; c += 8, ++k) {
            out[4 * k + 0] = dy[t];
            out[4 * k + 1] = dx[t];
            out[4 * k + 2] = s | (t << 16);
            out[4 * k + 3] = c;
          }
      }
  }
  for (; k < padded; ++k) { out[4 * k] = 0; out[4 * k + 1] = 0; out[4 * k + 2] = 255; out[4 * k + 3] = 0; }
  out[4 * padded] = out[4 * padded + 1] = out[4 * padded + 2] = out[4 * padded + 3] = 0;   // 16 zero bytes: the kernels' zero page
  return padded;
}

extern "C" int pp_conv2d(const pp_conv_args_t* a, void* stream) {
  using namespace pp;
  PP_REQUIRE(a != nullptr, PP_ERR_ARG, "pp_conv2d: null args");
  PP_REQUIRE(a->dtype == PP_F32 || a->dtype == PP_F16, PP_ERR_DTYPE, "pp_conv2d: dtype %d", a->dtype);
  PP_REQUIRE(a->out_dtype == PP_F32 || a->out_dtype == PP_F16, PP_ERR_DTYPE, "pp_conv2d: out_dtype %d", a->out_dtype);
  PP_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->OH > 0 && a->OW > 0 && a->groups > 0 && a->cout_g > 0, PP_ERR_ARG,
             "pp_conv2d: bad extents N=%d H=%d W=%d OH=%d OW=%d groups=%d cout_g=%d", a->N, a->H, a->W, a->OH, a->OW,
             a->groups, a->cout_g);
  PP_REQUIRE(a->nsrc >= 1 && a->nsrc <= PP_CONV_MAX_SRC, PP_ERR_ARG, "pp_conv2d: nsrc %d", a->nsrc);
  PP_REQUIRE(a->kchunks > 0 && a->kchunks % 4 == 0, PP_ERR_ARG, "pp_conv2d: kchunks %d must be a positive multiple of 4", a->kchunks);
  PP_REQUIRE(a->cout_pad >= a->cout_g, PP_ERR_ARG, "pp_conv2d: cout_pad %d (cout_g %d)", a->cout_pad, a->cout_g);
  PP_REQUIRE(a->ktable && a->weight && a->out, PP_ERR_ARG, "pp_conv2d: null ktable/weight/out");
  const int esz = a->dtype == PP_F16 ? 2 : 4;
  for (int s = 0; s < a->nsrc; ++s) {
    PP_REQUIRE(a->src[s].ptr != nullptr, PP_ERR_ARG, "pp_conv2d: source %d is null", s);
    PP_REQUIRE(((uintptr_t)a->src[s].ptr % 16) == 0 && (a->src[s].cstride * esz) % 16 == 0 && (a->src[s].choff * esz) % 16 == 0 &&
                   (a->src[s].cgroup * esz) % 16 == 0,
               PP_ERR_ALIGN, "pp_conv2d: source %d violates 16-byte alignment (cstride %d choff %d cgroup %d)", s,
               a->src[s].cstride, a->src[s].choff, a->src[s].cgroup);
  }
  PP_REQUIRE(((uintptr_t)a->weight % 16) == 0, PP_ERR_ALIGN, "pp_conv2d: weight pointer not 16-byte aligned");
  if (a->dcn_offmask) {
    PP_REQUIRE(a->stride_h == 1 && a->stride_w == 1 && a->OH == a->H && a->OW == a->W, PP_ERR_ARG,
               "pp_conv2d: deformable mode needs stride 1 and same-size output");
  }
  ConvParams p;
  p.N = a->N; p.H = a->H; p.W = a->W; p.OH = a->OH; p.OW = a->OW;
  p.sh = a->stride_h; p.sw = a->stride_w; p.ph = a->pad_h; p.pw = a->pad_w; p.pad_mode = a->pad_mode;
  p.cout_g = a->cout_g; p.cout_pad = a->cout_pad; p.kchunks = a->kchunks; p.nsrc = a->nsrc;
  for (int s = 0; s < PP_CONV_MAX_SRC; ++s) {
    const int ss = s < a->nsrc ? s : 0;
    p.src[s].ptr = (const char*)a->src[ss].ptr; p.src[s].cstride = a->src[ss].cstride;
    p.src[s].choff = a->src[ss].choff; p.src[s].cgroup = a->src[ss].cgroup; p.src[s].lo = a->src[ss].lo_off;
  }
  p.ktable = (const int4*)a->ktable; p.weight = (const char*)a->weight; p.weight_gstride = a->weight_gstride;
  p.bias = a->bias; p.act = a->act; p.act_param = a->act_param; p.out_scale = a->out_scale;
  p.residual = (const char*)a->residual; p.res_cstride = a->res_cstride; p.res_choff = a->res_choff; p.act2 = a->act2;
  p.out_f16 = a->out_dtype == PP_F16; p.out = (char*)a->out; p.out_cstride = a->out_cstride; p.out_choff = a->out_choff;
  p.out_cgroup = a->out_cgroup; p.src_gstride = a->src_gstride; p.out_gstride = a->out_gstride;
  p.dcn = (const char*)a->dcn_offmask; p.dcn_cstride = a->dcn_cstride; p.dcn_mask_off = a->dcn_mask_off;
  p.M = (long long)a->N * a->OH * a->OW;
  const bool deform = a->dcn_offmask != nullptr;
  p.groups = a->groups; p.tiles_m = 0; p.tiles_n = 0; p.ktable_uniform = a->ktable_uniform;
  p.tap_h = a->tap_h; p.tap_w = a->tap_w;
  p.preadd = (const char*)a->preadd; p.preadd_cstride = a->preadd_cstride; p.preadd_choff = a->preadd_choff;
  p.fuse = a->fuse; p.fuse_split = a->fuse_split;
  p.fuse_a = (const char*)a->fuse_a; p.fuse_a_cstride = a->fuse_a_cstride; p.fuse_a_choff = a->fuse_a_choff;
  p.fuse_b = (const char*)a->fuse_b; p.fuse_b_cstride = a->fuse_b_cstride; p.fuse_b_choff = a->fuse_b_choff;
  p.out2 = (char*)a->out2; p.out2_cstride = a->out2_cstride; p.out2_choff = a->out2_choff;
  p.split = a->split; p.out_lo = a->out_lo; p.out2_lo = a->out2_lo; p.preadd_lo = a->preadd_lo; p.res_lo = a->res_lo;
  p.fuse_a_lo = a->fuse_a_lo; p.fuse_b_lo = a->fuse_b_lo;
  p.dcn_stats = a->dcn_stats;
  {   // PP_EPI_DIRECT: 0 = never, 1 (default) = the batched GEMMs (short K, output-bound), 2 = every plain fp32 output (read per launch: A/B runs)
    const char* ed = getenv("PP_EPI_DIRECT");
    const int mode = ed != nullptr ? atoi(ed) : 1;
    p.epi_direct = mode == 2 || (mode == 1 && a->groups > 1 && a->out_gstride != 0);
  }
  if (a->preadd != nullptr || a->fuse != PP_FUSE_NONE) {
    PP_REQUIRE(a->groups == 1 && !deform && a->out_dtype == a->dtype && a->cout_g % 8 == 0, PP_ERR_ARG,
               "pp_conv2d: the fused epilogue (preadd / fuse) needs groups == 1, no deformable sampling, out_dtype == dtype, cout_g %% 8 == 0");
    PP_REQUIRE(a->fuse >= PP_FUSE_NONE && a->fuse <= PP_FUSE_DCN_OFFMASK, PP_ERR_ARG, "pp_conv2d: fuse %d", a->fuse);
    PP_REQUIRE(a->preadd == nullptr || (((uintptr_t)a->preadd % 16) == 0 && ((a->preadd_cstride | a->preadd_choff) & 7) == 0),
               PP_ERR_ALIGN, "pp_conv2d: preadd must be 16-byte aligned with cstride / choff multiples of 8");
    if (a->fuse == PP_FUSE_DCN_OFFMASK) {
      PP_REQUIRE(a->act == PP_ACT_NONE && a->preadd == nullptr && a->fuse_split > 0 && a->fuse_split % 8 == 0 && a->fuse_split <= a->cout_g,
                 PP_ERR_ARG, "pp_conv2d: PP_FUSE_DCN_OFFMASK needs act == none, no preadd, 0 < fuse_split <= cout_g, multiple of 8");
      PP_REQUIRE(a->fuse_a == nullptr || (((uintptr_t)a->fuse_a % 4) == 0 && ((a->fuse_a_cstride | a->fuse_a_choff) & 1) == 0),
                 PP_ERR_ALIGN, "pp_conv2d: PP_FUSE_DCN_OFFMASK flow window must be 4-byte aligned with even cstride / choff");
    } else if (a->fuse != PP_FUSE_NONE) {
      PP_REQUIRE(a->fuse_a != nullptr && ((uintptr_t)a->fuse_a % 16) == 0 && ((a->fuse_a_cstride | a->fuse_a_choff) & 7) == 0,
                 PP_ERR_ALIGN, "pp_conv2d: fuse_a must be set, 16-byte aligned, cstride / choff multiples of 8");
    }
    PP_REQUIRE(a->residual == nullptr, PP_ERR_ARG, "pp_conv2d: preadd / fuse and residual are exclusive");
    if (a->fuse == PP_FUSE_GRU_ZR)
      PP_REQUIRE(a->out2 != nullptr && a->fuse_split > 0 && a->fuse_split % 8 == 0 && a->fuse_split < a->cout_g &&
                     ((uintptr_t)a->out2 % 16) == 0 && ((a->out2_cstride | a->out2_choff) & 7) == 0,
                 PP_ERR_ARG, "pp_conv2d: PP_FUSE_GRU_ZR needs out2 (aligned) and 0 < fuse_split < cout_g, multiple of 8");
    if (a->fuse == PP_FUSE_GRU_H)
      PP_REQUIRE(a->fuse_b != nullptr && ((uintptr_t)a->fuse_b % 16) == 0 && ((a->fuse_b_cstride | a->fuse_b_choff) & 7) == 0,
                 PP_ERR_ALIGN, "pp_conv2d: PP_FUSE_GRU_H needs fuse_b (z), 16-byte aligned, cstride / choff multiples of 8");
  }
  hipStream_t st = (hipStream_t)stream;
  if (a->split) {
    // split-plane ("f16x3") layer: the LDS-DMA kernels with the split epilogue only -- there is no register-staged fallback
    PP_REQUIRE((a->split == 1 || a->split == 2) && a->dtype == PP_F16 && !deform && a->fuse != PP_FUSE_DCN_OFFMASK, PP_ERR_ARG,
               "pp_conv2d: split-plane layers (split 1 / 2) need dtype PP_F16, no deformable sampling, no PP_FUSE_DCN_OFFMASK");
    PP_REQUIRE(a->groups == 1 || (a->out_dtype == PP_F32 && a->residual == nullptr && a->preadd == nullptr && a->fuse == PP_FUSE_NONE), PP_ERR_ARG,
               "pp_conv2d: grouped / batched split-plane layers write plain fp32 and take no epilogue operands (the volume GEMM)");
    PP_REQUIRE(((a->out_lo | a->out2_lo | a->preadd_lo | a->res_lo | a->fuse_a_lo | a->fuse_b_lo) & 7) == 0 && a->out_lo >= 0 &&
                   a->out2_lo >= 0 && a->preadd_lo >= 0 && a->res_lo >= 0 && a->fuse_a_lo >= 0 && a->fuse_b_lo >= 0,
               PP_ERR_ALIGN, "pp_conv2d: split-plane lo offsets must be non-negative multiples of 8 elements");
    PP_REQUIRE(a->out_dtype == PP_F32 || (a->out_lo > 0 && ((uintptr_t)a->out % 16) == 0 && ((a->out_cstride | a->out_choff) & 7) == 0),
               PP_ERR_ALIGN, "pp_conv2d: a split-plane fp16 output needs out_lo > 0, a 16-byte aligned base, cstride / choff multiples of 8");
    PP_REQUIRE(a->residual == nullptr || (((uintptr_t)a->residual % 16) == 0 && ((a->res_cstride | a->res_choff) & 7) == 0 && a->res_lo > 0),
               PP_ERR_ALIGN, "pp_conv2d: a split-plane residual needs res_lo > 0, 16-byte alignment, cstride / choff multiples of 8");
    PP_REQUIRE((a->preadd == nullptr || a->preadd_lo > 0) && (a->fuse == PP_FUSE_NONE || a->fuse_a_lo > 0) &&
                   (a->fuse != PP_FUSE_GRU_ZR || a->out2_lo > 0) && (a->fuse != PP_FUSE_GRU_H || a->fuse_b_lo > 0),
               PP_ERR_ARG, "pp_conv2d: split-plane epilogue operands need their lo offsets");
    // split == 2: TRI-PRODUCT K format (per tap 4 hi + 4 lo chunks of 32 channels, weights [W_hi | W_lo]): the halo-tile kernel only;
    // split == 1: every block walked three times by a plain K loop: the LDS-DMA (v2) kernel
    // ... or, outside the halo family (1x1, strided, batched GEMM, sources that are no multiples of 32 channels), the v2 kernel's tri step
    const int v2cfg = (a->impl >= 10 && a->impl < 70) || (a->impl > 110 && (a->impl < 116 || a->impl > 118)) ? a->impl : 0;      // (82 / 83: halo8, below)
    int rc = -1000;
    if (a->split == 2 && (a->impl == 0 || a->impl == 110)) rc = conv_head_dispatch(p, st, a->impl == 110);   // 3x3 heads with <= 4 couts (opt-in)
    PP_REQUIRE(a->impl != 110 || rc != -1000, PP_ERR_ARG, "pp_conv2d: impl 110 (streaming head kernel) not available for this layer");
    if (rc == -1000 && a->split == 2 && v2cfg == 0) rc = conv_v3s_dispatch(p, a->impl == 71 || a->impl == 72 || a->impl == 82 || a->impl == 83 ? a->impl : 0, st);
    if (rc == -1000) rc = conv_v2s_dispatch(p, v2cfg, st);
    PP_REQUIRE(rc != -1000, PP_ERR_ARG, "pp_conv2d: no split-plane kernel for this layer (split %d, kchunks %d, %dx%d taps, impl %d)", a->split,
               a->kchunks, a->tap_h, a->tap_w, a->impl);
    return rc;
  }
  if (a->dtype == PP_F16 && !deform && (a->impl == 80 || a->impl == 81)) {
    // A-stationary kernel for the short-K single-source linears (transformer GEMMs).  Selected explicitly only: with
    // interleaved A/B rounds (tools/kbench, profiles/r2_conv_epilogue_ab.txt) the 256 x 128 LDS-DMA tile beats it by
    // 7-25 % on the four transformer shapes (round 1 had measured +4 % for it on separate runs)
    const int rc = conv_ast_dispatch(p, a->impl, st);
    if (rc != -1000) return rc;
    PP_REQUIRE(a->impl == 0, PP_ERR_ARG, "pp_conv2d: impl 80 (A-stationary GEMM) not available for this shape");
  }
  if (a->dtype == PP_F16 && !deform && (a->impl == 0 || a->impl == 110)) {
    // 3x3 heads with at most 4 couts (flow / RGB heads) as VALU dot products instead of a 16-cout MFMA tile: measured neutral, opt-in
    // (PP_HEAD_KERNEL=1 / impl 110; conv_head.hip)
    const int rc = conv_head_dispatch(p, st, a->impl == 110);
    if (rc != -1000) return rc;
    PP_REQUIRE(a->impl == 0, PP_ERR_ARG, "pp_conv2d: impl 110 (streaming head kernel) not available for this layer");
  }
  if (a->dtype == PP_F16 && !deform && (a->impl == 0 || (a->impl >= 70 && a->impl < 80) || a->impl == 82 || a->impl == 83 || a->impl == 109)) {
    // halo-tile kernel family (stride-1 "same" 3x3 / 1x5 / 5x1 windows over 64-channel-multiple sources)
    const int rc = conv_v3_dispatch(p, a->impl, st);
    if (rc != -1000) return rc;
    PP_REQUIRE(a->impl == 0, PP_ERR_ARG, "pp_conv2d: impl %d (halo tiles) not available for this shape", a->impl);
  }
  if (a->dtype == PP_F16 && !deform && a->impl != 1) {
    // LDS-DMA kernel family (needs kchunks % 8 == 0 and the trailing all-zero table entry)
    const int rc = conv_v2_dispatch(p, a->impl >= 10 ? a->impl : (a->impl == 2 ? 100 : 0), st);
    if (rc != -1000) return rc;
    PP_REQUIRE(a->impl < 10, PP_ERR_ARG, "pp_conv2d: impl %d not available for this shape", a->impl);
  }
#if defined(PP_DIAG)
  const bool dcn_impl = a->impl == 0 || (a->impl >= 90 && a->impl < 138);
#else
  const bool dcn_impl = a->impl == 0 || a->impl == 90 || a->impl == 106 || a->impl == 122;      // 106 / 122: force 128- / 64-pixel tiles (conv_dcn.hip)
#endif
  if (a->dtype == PP_F16 && deform && dcn_impl) {
    // patch-staged deformable kernel (16 offset groups of 8 / 16 channels, stride 1, 3x3)
    const int rc = conv_dcn_dispatch(p, st, a->impl >= 90 ? a->impl - 90 : 0);
    if (rc != -1000) return rc;
    PP_REQUIRE(a->impl == 0, PP_ERR_ARG, "pp_conv2d: impl 90 (patch-staged deformable kernel) not available for this layer");
  }
  if (a->dtype == PP_F16) return dispatch_conv<_Float16>(p, a->groups, deform, st);
  PP_REQUIRE(a->impl != 3 || !deform, PP_ERR_ARG, "pp_conv2d: impl 3 (split-fp16 products) does not apply to the deformable mode");
  return dispatch_conv<float>(p, a->groups, deform, st, a->impl == 3);
}
