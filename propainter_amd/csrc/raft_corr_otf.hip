// RAFT correlation lookup WITHOUT the all-pairs volume (fp16 engine).
//
// The reference builds corr = f1^T f2 / 16 for every pair of positions (n8 x n8 fp32 per pair: 829 MB at 720p, 5.6 GB
// at 1080p), avg-pools it into a 4-level pyramid (RAFT/corr.py:13-27,52-60) and, in each of the 20 iterations, reads a
// 9x9 bilinear window per level around the current estimate (RAFT/corr.py:29-50).  Average pooling is linear, so
//     avg_pool(f1[p] . f2)[q] = f1[p] . avg_pool(f2)[q]
// and the pyramid of the VOLUME equals the volume of the feature PYRAMID: pp_corr_feature_pyramid pools f2 once per
// pair (9.8 MB fp16 at 720p, L2 / Infinity-Cache resident) and pp_corr_lookup_otf computes, per iteration, exactly the
// (10 x 10 integer neighbourhood) x 4 levels of dot products each pixel needs -- on MFMA -- and blends them bilinearly
// into the 324-channel NHWC tile.  No volume GEMM, no pooling of 1.1 GB per pair, no 40-byte row gathers.
//
// Kernel: one 512-thread block per 8x8 tile of source pixels of one pair.  Per level:
//   1. bounding box (in level coordinates, clipped to the map) of the 64 windows -> LDS;
//   2. the 8 waves share the box's positions in N tiles of 16 targets: B fragments (16 targets x 256 channels of the
//      level's f2) come straight from L2 into registers -- every 64-byte sector is used in full --, A fragments (the 64
//      pixels x 256 channels of f1) stay in registers for the whole block, S = B x A^T on v_mfma_f32_16x16x32_f16 with
//      fp32 accumulation, results to LDS as V[pixel][position] (fp32);
//   3. every (pixel, a, b) output blends its four neighbours of V with the reference's per-tap coordinate round trip
//      (bilinear_sampler's 2c/(W-1)-1 normalisation, zeros outside the map) and lands in an LDS staging tile;
//   finally the staging tile goes out as full 656-byte NHWC rows.
// The box of a smooth flow field is ~17x17 positions at level 0 (64 pixels share 289 targets instead of 6400).  A
// box too large for LDS (wildly divergent flow inside one tile) is handled by the same code on pixel subsets: the four
// 4x4 quadrants one after another, and single pixels in the worst case -- slower, never wrong.  The blend order is
// fixed, nothing is accumulated with atomics: results are deterministic and independent of the batch.
#include "common.h"

namespace pp {

struct CorrOtfParams {
  const char* f1;            // fp16 NHWC [P, h, w, 256]
  const char* f2[4];         // fp16 NHWC [P, h >> l, w >> l, 256]
  const float* coords;       // fp32 [P, h, w, 2] (x, y)
  _Float16* out;             // fp16 NHWC [P, h, w, ocs]; channels [0, 324) written, [324, ocpad) zeroed
  int P, h, w, ocs, ocpad, tiles_x, tiles_y;
  float scale;               // 1 / sqrt(256)
};

constexpr int OTF_VTOT = 28928;                    // floats of V storage (113 KB): 64 x 452, 16 x 1808, 1 x 28928
constexpr int OTF_OROW = 328;                      // staging row (fp16 elements)
constexpr int OTF_LDS = OTF_VTOT * 4 + 64 * OTF_OROW * 2 + 64 * 8 + 5 * 16;

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__global__ __launch_bounds__(512) void corr_otf_kernel(const CorrOtfParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[OTF_LDS];
  float* const V = reinterpret_cast<float*>(lds);
  _Float16* const stage = reinterpret_cast<_Float16*>(lds + OTF_VTOT * 4);
  float* const cxy = reinterpret_cast<float*>(lds + OTF_VTOT * 4 + 64 * OTF_OROW * 2);     // [64][2]; x = NaN marks a pixel outside the image
  int* const boxes = reinterpret_cast<int*>(lds + OTF_VTOT * 4 + 64 * OTF_OROW * 2 + 64 * 8);   // [5][4] = {bx0, by0, bw, bh}: levels 0..3 of the whole tile, [4] = scratch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;

  // ---- XCD-aware tile order: every XCD walks a contiguous run of tiles (neighbouring tiles share f2 rows in its L2)
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int txi = bid % p.tiles_x;
  const int tyi = (bid / p.tiles_x) % p.tiles_y;
  const int n = bid / (p.tiles_x * p.tiles_y);
  const int ty0 = tyi * 8, tx0 = txi * 8;
  // pixel q of the tile: M tile (quadrant) q >> 4, inside it row (q >> 2) & 3, column q & 3
  auto tile_xy = [&](int q, int& x, int& y) {
    x = tx0 + ((q >> 4) & 1) * 4 + (q & 3);
    y = ty0 + (q >> 5) * 4 + ((q >> 2) & 3);
  };

  // Bounding box (level coordinates, clipped to the map) of the windows [floor(c) - 4, floor(c) + 5] of the pixels
  // [p0, p0 + np) -> dst[0..3]; executed by ONE whole wave.
  auto wave_box = [&](int lvl, int p0, int np, int* dst) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    const float lscale = 1.f / (float)(1 << lvl);
    int x0 = 1 << 30, x1 = -(1 << 30), y0 = 1 << 30, y1 = -(1 << 30);
    if (lane < np) {
      const float cx = cxy[(p0 + lane) * 2], cy = cxy[(p0 + lane) * 2 + 1];
      if (cx == cx) {
        const int fx = (int)floorf(cx * lscale), fy = (int)floorf(cy * lscale);
        x0 = max(fx - 4, 0); x1 = min(fx + 5, Wl - 1);
        y0 = max(fy - 4, 0); y1 = min(fy + 5, Hl - 1);
        if (x1 < x0 || y1 < y0) { x0 = y0 = 1 << 30; x1 = y1 = -(1 << 30); }     // window entirely outside the map
      }
    }
    x0 = wave_min(x0); y0 = wave_min(y0); x1 = wave_max(x1); y1 = wave_max(y1);
    if (lane == 0) {
      dst[0] = x0; dst[1] = y0;
      dst[2] = x1 >= x0 ? x1 - x0 + 1 : 0;
      dst[3] = y1 >= y0 ? y1 - y0 + 1 : 0;
    }
  };

  if (tid < 64) {
    int x, y;
    tile_xy(tid, x, y);
    float cx = __builtin_nanf(""), cy = 0.f;
    if (x < p.w && y < p.h) {
      const float* c = p.coords + (((long long)n * p.h + y) * p.w + x) * 2;
      cx = c[0];
      cy = c[1];
    }
    cxy[tid * 2] = cx;
    cxy[tid * 2 + 1] = cy;
  }
  // ---- f1 tile (64 pixels x 512 B) once per block through LDS (aliases V; 16-byte chunk j of pixel q at chunk j ^ (q & 31))
  {
    const __amdgpu_buffer_rsrc_t r1 = uniform_buffer_rsrc(p.f1 + (long long)n * p.h * p.w * 512, p.h * p.w * 512);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + 512 * i, q = id >> 5, j = id & 31;
      int x, y;
      tile_xy(q, x, y);
      const int voff = (x < p.w && y < p.h) ? (y * p.w + x) * 512 + j * 16 : (int)0x80000000;     // out of range -> zeros
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
      *reinterpret_cast<u32x4*>(lds + q * 512 + ((j ^ (q & 31)) << 4)) = raw;
    }
  }
  __syncthreads();
  if (wave < 4) wave_box(wave, 0, 64, boxes + wave * 4);
  // ---- A fragments: resident in registers for the whole block (4 M tiles x 8 k steps)
  f16x8 afrag[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int q = mt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      afrag[mt][ks] = *reinterpret_cast<const f16x8*>(lds + q * 512 + (((ks * 4 + l4) ^ (q & 31)) << 4));
  }
  __syncthreads();             // the f1 tile is consumed (V may be written), boxes[] are visible

  // B fragments of N tile nt of a box: 16 positions x 256 channels straight from L2 (64-byte sectors used in full)
  auto load_b = [&](const __amdgpu_buffer_rsrc_t r2, int Wl, int bx0, int by0, int bw, int area, int nt, u32x4 (&b)[8]) {
    const int pos = nt * 16 + l15;
    const int ry = pos / bw, rx = pos - ry * bw;
    const int voff = pos < area ? ((by0 + ry) * Wl + bx0 + rx) * 512 + l4 * 16 : (int)0x80000000;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) b[ks] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, ks * 64, 0);
  };
  auto level_rsrc = [&](int lvl) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    return uniform_buffer_rsrc(p.f2[lvl] + (long long)n * Hl * Wl * 512, Hl * Wl * 512);
  };

  u32x4 bcur[8], bnxt[8];
  bool prefetched = false;       // bcur already holds this wave's first N tile of the level (issued before the previous blend)

  for (int lvl = 0; lvl < 4; ++lvl) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    const float lscale = 1.f / (float)(1 << lvl);
    const __amdgpu_buffer_rsrc_t r2 = level_rsrc(lvl);

    // Processes the pixel set [p0, p0 + np), np in {64, 16, 1}, whose box is bx[0..3].  Returns false (nothing done) when
    // the box does not fit the V capacity of that set size.  Block-uniform control flow throughout.
    auto process = [&](const int p0, const int np, const int* bx) -> bool {
      const int bx0 = bx[0], by0 = bx[1], bw = bx[2], bh = bx[3];
      const int area = bw * bh;
      const int vstride = OTF_VTOT / np;       // floats per pixel row of V: 452 (= 4 mod 32: conflict-free tile stores) / 1808 / 28928
      const int ntiles = (area + 15) >> 4;       // N tiles of 16 positions; whole tiles are stored, so they must fit the row
      if (ntiles * 16 > vstride) return false;
      const int mt0 = p0 >> 4;       // first M tile of the set
      const bool one_tile = np <= 16;
      // -- S = B x A^T over the box, N tiles of 16 positions round-robin over the 8 waves
      if (!prefetched && wave < ntiles) load_b(r2, Wl, bx0, by0, bw, area, wave, bcur);
      prefetched = false;
      for (int nt = wave; nt < ntiles; nt += 8) {
        const bool more = nt + 8 < ntiles;
        if (more) load_b(r2, Wl, bx0, by0, bw, area, nt + 8, bnxt);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (one_tile && mt != mt0) continue;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bcur[ks]), afrag[mt][ks], acc[mt], 0, 0, 0);
        }
        // acc[mt][r] = S[position nt*16 + l4*4 + r][pixel mt*16 + l15]
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (one_tile && mt != mt0) continue;
          const int q = mt * 16 + l15 - p0;                 // pixel index inside the set
          if (q >= 0 && q < np)
            *reinterpret_cast<f32x4*>(V + q * vstride + nt * 16 + l4 * 4) = acc[mt];
        }
        if (more) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) bcur[ks] = bnxt[ks];
        }
      }
      // -- the first N tile of the NEXT level's whole-tile box is requested now: its L2 latency hides behind the blend
      if (np == 64 && lvl < 3) {
        const int* nb = boxes + (lvl + 1) * 4;
        const int narea = nb[2] * nb[3], nnt = (narea + 15) >> 4;
        if (nnt * 16 <= OTF_VTOT / 64) {
          if (wave < nnt) load_b(level_rsrc(lvl + 1), p.w >> (lvl + 1), nb[0], nb[1], nb[2], narea, wave, bcur);
          prefetched = true;
        }
      }
      __syncthreads();
      // -- bilinear blend (RAFT/corr.py:36-43: tap a moves x, tap b moves y; zeros outside the map).  One work item =
      //    (pixel, a): the 10 x 2 neighbourhood columns (c, c + 1) are read once, lerped along x, then along y for the 9
      //    taps b.  All taps of a pixel share the fractional offsets (tap = centre + integer).
      for (int item = tid; item < np * 9; item += 512) {
        const int q = item / 9, a = item - q * 9;
        const float cx0 = cxy[(p0 + q) * 2];
        if (!(cx0 == cx0)) continue;
        const float cx = cx0 * lscale, cy = cxy[(p0 + q) * 2 + 1] * lscale;
        const float fx0 = floorf(cx), fy0 = floorf(cy);
        const float lx = cx - fx0, ly = cy - fy0;
        const int c = (int)fx0 - 4 + a - bx0, r0 = (int)fy0 - 4 - by0;
        const bool okc0 = (unsigned)c < (unsigned)bw, okc1 = (unsigned)(c + 1) < (unsigned)bw;
        const float* vrow = V + q * vstride;
        float hx[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const int rr = r0 + j;
          const bool okr = (unsigned)rr < (unsigned)bh;
          const int i0 = (okr && okc0) ? rr * bw + c : 0, i1 = (okr && okc1) ? rr * bw + c + 1 : 0;   // always-valid addresses
          float v0 = vrow[i0], v1 = vrow[i1];
          v0 = (okr && okc0) ? v0 : 0.f;
          v1 = (okr && okc1) ? v1 : 0.f;
          hx[j] = (1.f - lx) * v0 + lx * v1;
        }
        _Float16* so = stage + (p0 + q) * OTF_OROW + lvl * 81 + a * 9;
#pragma unroll
        for (int b = 0; b < 9; ++b) so[b] = (_Float16)(((1.f - ly) * hx[b] + ly * hx[b + 1]) * p.scale);
      }
      __syncthreads();              // V is reused
      return true;
    };

    if (!process(0, 64, boxes + lvl * 4)) {
      prefetched = false;
      for (int g = 0; g < 4; ++g) {
        if (wave == 0) wave_box(lvl, g * 16, 16, boxes + 16);
        __syncthreads();
        const bool ok = process(g * 16, 16, boxes + 16);
        if (!ok) {
          for (int q = 0; q < 16; ++q) {                       // a single window (<= 100 positions) always fits
            __syncthreads();
            if (wave == 0) wave_box(lvl, g * 16 + q, 1, boxes + 16);
            __syncthreads();
            process(g * 16 + q, 1, boxes + 16);
          }
        }
        __syncthreads();            // boxes[4] is rewritten by the next quadrant
      }
    }
  }

  // ---- staging tile -> NHWC rows (16-byte chunks; channels [324, ocpad) are zero)
  if (tid < 64) {
#pragma unroll
    for (int c = 324; c < OTF_OROW; ++c) stage[tid * OTF_OROW + c] = (_Float16)0.f;
  }
  __syncthreads();
  const int chunks = p.ocpad / 8;
  for (int o = tid; o < 64 * chunks; o += 512) {
    const int q = o / chunks, c = o - q * chunks;
    int x, y;
    tile_xy(q, x, y);
    if (x < p.w && y < p.h)
      *reinterpret_cast<u32x4*>(p.out + (((long long)n * p.h + y) * p.w + x) * p.ocs + c * 8) =
          *reinterpret_cast<const u32x4*>(stage + q * OTF_OROW + c * 8);
  }
#endif
}

// ================================================================================================================================
// The same lookup at fp32-class precision ("f16x3"): SPLIT-PLANE features (rows of 512 fp16 = 256 hi | 256 lo of fp32 values), every dot
// product as B_hi x A_hi + B_lo x A_hi + B_hi x A_lo on the fp16 matrix cores with fp32 accumulation (the tri-product of the split-plane
// convolutions), the blend with the PER-TAP coordinate round trip of pp_corr_lookup (RAFT/utils/utils.py:61-65 evaluates every tap's grid
// coordinate separately; at fp32-class precision the 1-ulp differences between taps are visible), split-plane output.
// Replaces, for the fp32-class engine, the fp32 all-pairs volume (829 MB per pair-direction at 720x1280, 5.6 GB at 1080x1920), its four
// GEMMs and the 40-byte row gathers of pp_corr_lookup (RAFT/corr.py:13-60).
//
// Differences from the fp16 kernel above, all forced by the doubled operand size:
//   * A (the f1 tile) is 64 px x 1 KB: it no longer fits ONE wave's registers next to the B fragments, so the 8 waves form two groups of
//     four -- group g owns M tiles 2g, 2g + 1 (32 pixels: 128 VGPRs of hi + lo fragments) and walks the box's N tiles round-robin over
//     its four waves; both groups stream every B tile (L2 -> registers; the kernel is latency-, not bandwidth-bound);
//   * B arrives in K halves (4 hi + 4 lo 16-byte loads per lane), two half buffers in flight alternately: the loads of half k + 1 are
//     issued before the MFMAs of half k;
//   * output channels are laid out PER LEVEL in 88-channel groups (81 taps + 7 zeros: 176 bytes = 11 16-byte chunks per plane), so a
//     level's results leave LDS right after its blend (22.5 KB staging instead of 90 KB for whole rows: V keeps its 113 KB) and the 1x1
//     convolution behind it (convc1) walks 11 full 32-channel blocks per plane (328 channels: 10 full + 1 ragged).  Its weight columns
//     are permuted accordingly by the engine (flow_comp_raft.py).
//   out: fp16 [P, h, w, ocs], hi plane at channel 0, lo plane at ocs / 2; level l, tap (a, b) at channel l * 88 + a * 9 + b.
constexpr int OTFS_LVC = 88;                        // channels per level group
constexpr int OTFS_STAGE = 64 * 2 * OTFS_LVC * 2;   // bytes: [64 px][hi | lo][88]
constexpr int OTFS_LDS = OTF_VTOT * 4 + OTFS_STAGE + 64 * 8 + 5 * 16;

__global__ __launch_bounds__(512, 2) void corr_otf_split_kernel(const CorrOtfParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[OTFS_LDS];
  float* const V = reinterpret_cast<float*>(lds);
  _Float16* const stage = reinterpret_cast<_Float16*>(lds + OTF_VTOT * 4);
  float* const cxy = reinterpret_cast<float*>(lds + OTF_VTOT * 4 + OTFS_STAGE);
  int* const boxes = reinterpret_cast<int*>(lds + OTF_VTOT * 4 + OTFS_STAGE + 64 * 8);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wig = wave & 3;          // M group (pixels 32 grp .. 32 grp + 31), wave inside the group
  const int l15 = lane & 15, l4 = lane >> 4;

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int txi = bid % p.tiles_x;
  const int tyi = (bid / p.tiles_x) % p.tiles_y;
  const int n = bid / (p.tiles_x * p.tiles_y);
  const int ty0 = tyi * 8, tx0 = txi * 8;
  auto tile_xy = [&](int q, int& x, int& y) {
    x = tx0 + ((q >> 4) & 1) * 4 + (q & 3);
    y = ty0 + (q >> 5) * 4 + ((q >> 2) & 3);
  };
  auto wave_box = [&](int lvl, int p0, int np, int* dst) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    const float lscale = 1.f / (float)(1 << lvl);
    int x0 = 1 << 30, x1 = -(1 << 30), y0 = 1 << 30, y1 = -(1 << 30);
    if (lane < np) {
      const float cx = cxy[(p0 + lane) * 2], cy = cxy[(p0 + lane) * 2 + 1];
      if (cx == cx) {
        const int fx = (int)floorf(cx * lscale), fy = (int)floorf(cy * lscale);
        x0 = max(fx - 4, 0); x1 = min(fx + 5, Wl - 1);
        y0 = max(fy - 4, 0); y1 = min(fy + 5, Hl - 1);
        if (x1 < x0 || y1 < y0) { x0 = y0 = 1 << 30; x1 = y1 = -(1 << 30); }
      }
    }
    x0 = wave_min(x0); y0 = wave_min(y0); x1 = wave_max(x1); y1 = wave_max(y1);
    if (lane == 0) {
      dst[0] = x0; dst[1] = y0;
      dst[2] = x1 >= x0 ? x1 - x0 + 1 : 0;
      dst[3] = y1 >= y0 ? y1 - y0 + 1 : 0;
    }
  };

  if (tid < 64) {
    int x, y;
    tile_xy(tid, x, y);
    float cx = __builtin_nanf(""), cy = 0.f;
    if (x < p.w && y < p.h) {
      const float* c = p.coords + (((long long)n * p.h + y) * p.w + x) * 2;
      cx = c[0];
      cy = c[1];
    }
    cxy[tid * 2] = cx;
    cxy[tid * 2 + 1] = cy;
  }
  // the 7 pad channels of every staging row stay zero for the whole block (the blend writes channels 0..80 only)
  for (int i = tid; i < 128 * 7; i += 512) stage[(i / 7) * OTFS_LVC + 81 + (i % 7)] = (_Float16)0.f;
  // ---- f1 tile (64 pixels x 1 KB: 32 hi chunks | 32 lo chunks) once per block through LDS (aliases V; chunk j of pixel q at j ^ (q & 31))
  {
    const __amdgpu_buffer_rsrc_t r1 = uniform_buffer_rsrc(p.f1 + (long long)n * p.h * p.w * 1024, p.h * p.w * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 512 * i, q = id >> 6, j = id & 63;
      int x, y;
      tile_xy(q, x, y);
      const int voff = (x < p.w && y < p.h) ? (y * p.w + x) * 1024 + j * 16 : (int)0x80000000;
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
      *reinterpret_cast<u32x4*>(lds + q * 1024 + ((j ^ (q & 31)) << 4)) = raw;
    }
  }
  __syncthreads();
  if (wave < 4) wave_box(wave, 0, 64, boxes + wave * 4);
  // ---- A fragments of the group's two M tiles, both planes: resident for the whole block
  f16x8 ahi[2][8], alo[2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int q = (grp * 2 + m) * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      ahi[m][ks] = *reinterpret_cast<const f16x8*>(lds + q * 1024 + (((ks * 4 + l4) ^ (q & 31)) << 4));
      alo[m][ks] = *reinterpret_cast<const f16x8*>(lds + q * 1024 + (((32 + ks * 4 + l4) ^ (q & 31)) << 4));
    }
  }
  __syncthreads();

  // K half `half` (channels 128 half .. +127 of both planes) of N tile nt: 16 positions straight from L2
  auto load_half = [&](const __amdgpu_buffer_rsrc_t r2, int Wl, int bx0, int by0, int bw, int area, int nt, int half, u32x4 (&bh)[4], u32x4 (&bl)[4]) {
    const int pos = nt * 16 + l15;
    const int ry = pos / bw, rx = pos - ry * bw;
    const int voff = pos < area ? ((by0 + ry) * Wl + bx0 + rx) * 1024 + l4 * 16 + half * 256 : (int)0x80000000;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bh[k] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, k * 64, 0);
      bl[k] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 512 + k * 64, 0);
    }
  };
  auto level_rsrc = [&](int lvl) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    return uniform_buffer_rsrc(p.f2[lvl] + (long long)n * Hl * Wl * 1024, Hl * Wl * 1024);
  };
  auto tri = [&](f32x4& acc, const u32x4& bh, const u32x4& bl, const f16x8& ah, const f16x8& al) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bl), ah, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh), al, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh), ah, acc, 0, 0, 0);
  };

  u32x4 b0h[4], b0l[4], b1h[4], b1l[4];
  bool prefetched = false;       // (b0h, b0l) already hold half 0 of this wave's first N tile of the level

  for (int lvl = 0; lvl < 4; ++lvl) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    const float lscale = 1.f / (float)(1 << lvl);
    const __amdgpu_buffer_rsrc_t r2 = level_rsrc(lvl);

    auto process = [&](const int p0, const int np, const int* bx) -> bool {
      const int bx0 = bx[0], by0 = bx[1], bw = bx[2], bh_ = bx[3];
      const int area = bw * bh_;
      const int vstride = OTF_VTOT / np;
      const int ntiles = (area + 15) >> 4;
      if (ntiles * 16 > vstride) return false;
      const int mt0 = p0 >> 4;                         // first M tile of the set
      const bool one_tile = np <= 16;
      const bool mine = !one_tile || (mt0 >> 1) == grp;      // a single M tile belongs to one group: the other one idles
      if (mine) {
        if (!prefetched && wig < ntiles) load_half(r2, Wl, bx0, by0, bw, area, wig, 0, b0h, b0l);
        for (int nt = wig; nt < ntiles; nt += 4) {
          load_half(r2, Wl, bx0, by0, bw, area, nt, 1, b1h, b1l);
          f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (one_tile && (grp * 2 + m) != mt0) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) tri(acc[m], b0h[k], b0l[k], ahi[m][k], alo[m][k]);
          }
          if (nt + 4 < ntiles) load_half(r2, Wl, bx0, by0, bw, area, nt + 4, 0, b0h, b0l);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (one_tile && (grp * 2 + m) != mt0) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) tri(acc[m], b1h[k], b1l[k], ahi[m][4 + k], alo[m][4 + k]);
          }
          // acc[m][r] = S[position nt*16 + l4*4 + r][pixel (2 grp + m)*16 + l15]
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (one_tile && (grp * 2 + m) != mt0) continue;
            const int q = (grp * 2 + m) * 16 + l15 - p0;
            if (q >= 0 && q < np) *reinterpret_cast<f32x4*>(V + q * vstride + nt * 16 + l4 * 4) = acc[m];
          }
        }
      }
      prefetched = false;
      if (np == 64 && lvl < 3) {       // half 0 of the first N tile of the NEXT level's whole-tile box: its latency hides behind the blend
        const int* nb = boxes + (lvl + 1) * 4;
        const int narea = nb[2] * nb[3], nnt = (narea + 15) >> 4;
        if (nnt * 16 <= OTF_VTOT / 64) {
          if (wig < nnt) load_half(level_rsrc(lvl + 1), p.w >> (lvl + 1), nb[0], nb[1], nb[2], narea, wig, 0, b0h, b0l);
          prefetched = true;
        }
      }
      __syncthreads();
      // -- blend: one work item per output (pixel, a, b) with pp_corr_lookup's arithmetic (per-tap grid round trip, the four corners
      //    accumulated in the same order); corners outside the box are outside the map (zeros) -- or, when a tap's round trip crosses
      //    an integer by one ulp, one cell beyond the box, where their weight is <= 1 ulp of the coordinate
      for (int item = tid; item < np * 81; item += 512) {
        const int q = item / 81, j = item - q * 81;
        const float cx0 = cxy[(p0 + q) * 2];
        if (!(cx0 == cx0)) continue;
        const int a = j / 9, b = j - a * 9;
        const float cx = cx0 * lscale, cy = cxy[(p0 + q) * 2 + 1] * lscale;
        const float px = grid_roundtrip(cx + (float)(a - 4), Wl);
        const float py = grid_roundtrip(cy + (float)(b - 4), Hl);
        const float fx = floorf(px), fy = floorf(py);
        const float lx = px - fx, ly = py - fy;
        const int c0 = (int)fx - bx0, r0 = (int)fy - by0;
        const float* vrow = V + q * vstride;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = r0 + (k >> 1), cc = c0 + (k & 1);
          const float wgt = ((k & 1) ? lx : 1.f - lx) * ((k >> 1) ? ly : 1.f - ly);
          const bool ok = (unsigned)rr < (unsigned)bh_ && (unsigned)cc < (unsigned)bw;
          float s = vrow[ok ? rr * bw + cc : 0];
          s = ok ? s * p.scale : 0.f;                  // (the volume holds f1 . f2 / 16: scale before the blend, as RAFT/corr.py:60)
          acc += wgt * s;
        }
        const _Float16 hi = (_Float16)acc;
        _Float16* so = stage + (p0 + q) * 2 * OTFS_LVC + j;
        so[0] = hi;
        so[OTFS_LVC] = (_Float16)(acc - (float)hi);
      }
      __syncthreads();
      return true;
    };

    if (!process(0, 64, boxes + lvl * 4)) {
      prefetched = false;
      for (int g = 0; g < 4; ++g) {
        if (wave == 0) wave_box(lvl, g * 16, 16, boxes + 16);
        __syncthreads();
        const bool ok = process(g * 16, 16, boxes + 16);
        if (!ok) {
          for (int q = 0; q < 16; ++q) {
            __syncthreads();
            if (wave == 0) wave_box(lvl, g * 16 + q, 1, boxes + 16);
            __syncthreads();
            process(g * 16 + q, 1, boxes + 16);
          }
        }
        __syncthreads();
      }
    }
    // ---- this level's 88-channel group of both planes -> NHWC rows (11 16-byte chunks per pixel and plane)
    for (int o = tid; o < 64 * 2 * 11; o += 512) {
      const int q = o / 22, r = o - q * 22, pl = r / 11, c = r - pl * 11;
      int x, y;
      tile_xy(q, x, y);
      if (x < p.w && y < p.h)
        *reinterpret_cast<u32x4*>(p.out + (((long long)n * p.h + y) * p.w + x) * p.ocs + pl * (p.ocs >> 1) + lvl * OTFS_LVC + c * 8) =
            *reinterpret_cast<const u32x4*>(stage + (q * 2 + pl) * OTFS_LVC + c * 8);
    }
    // (the next level's blend rewrites the staging tile only after its own __syncthreads(), which every thread reaches after these reads)
  }
#endif
}

// Level l of the f2 feature pyramid: mean over the 2^l x 2^l block at (y << l, x << l) -- what l nested
// F.avg_pool2d(2, 2) (floor sizes) compute --, accumulated in fp32 from level 0 and rounded to fp16 once.
// SPLIT: split-plane ("f16x3") features -- rows of 512 fp16 = [256 hi | 256 lo]; the fp32 values hi + lo are averaged in fp32 and the
// mean is stored as hi = fp16(m), lo = fp16(m - hi) (22 significand bits: the fp32-class operand of the level's correlation GEMM).
template <bool SPLIT>
__global__ void corr_feature_pool_kernel(const _Float16* __restrict__ f2, _Float16* __restrict__ out, int P, int h, int w, int lvl) {
  constexpr int RS = SPLIT ? 512 : 256;                          // fp16 elements per pixel row
  const int Hl = h >> lvl, Wl = w >> lvl, s = 1 << lvl;
  const long long total = (long long)P * Hl * Wl * 32;           // 8-channel chunks
  const float inv = 1.f / (float)(s * s);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i & 31);
    long long r = i >> 5;
    const int x = (int)(r % Wl); r /= Wl;
    const int y = (int)(r % Hl);
    const long long n = r / Hl;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8];
    for (int dy = 0; dy < s; ++dy)
      for (int dx = 0; dx < s; ++dx) {
        const _Float16* src = f2 + ((n * h + (y * s + dy)) * (long long)w + (x * s + dx)) * RS + c * 8;
        if constexpr (SPLIT) load8_split(src, src + 256, v);
        else load8<_Float16>(src, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    _Float16* dst = out + ((n * Hl + y) * (long long)Wl + x) * RS + c * 8;
    if constexpr (SPLIT) store8_split(dst, dst + 256, acc);
    else store8<_Float16>(dst, acc);
  }
}

}  // namespace pp

using namespace pp;

static int feature_pyramid(const void* f2, void* lvl1, void* lvl2, void* lvl3, int P, int h, int w, bool split, void* stream, const char* who) {
  PP_REQUIRE(f2 && lvl1 && lvl2 && lvl3 && P > 0, PP_ERR_ARG, "%s: bad arguments", who);
  PP_REQUIRE((h >> 3) >= 2 && (w >> 3) >= 2, PP_ERR_ARG,
             "%s: level-3 map would be %dx%d (RAFT needs >= 2x2: inputs of at least 128x128)", who, h >> 3, w >> 3);
  PP_REQUIRE(((uintptr_t)f2 % 16) == 0 && ((uintptr_t)lvl1 % 16) == 0 && ((uintptr_t)lvl2 % 16) == 0 && ((uintptr_t)lvl3 % 16) == 0, PP_ERR_ALIGN,
             "%s: pointers must be 16-byte aligned", who);
  void* outs[3] = {lvl1, lvl2, lvl3};
  for (int l = 1; l <= 3; ++l) {
    const long long total = (long long)P * (h >> l) * (w >> l) * 32;
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (split)
      hipLaunchKernelGGL(corr_feature_pool_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const _Float16*)f2,
                         (_Float16*)outs[l - 1], P, h, w, l);
    else
      hipLaunchKernelGGL(corr_feature_pool_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const _Float16*)f2,
                         (_Float16*)outs[l - 1], P, h, w, l);
  }
  return launch_status(who);
}

extern "C" int pp_corr_feature_pyramid(const void* f2, void* lvl1, void* lvl2, void* lvl3, int P, int h, int w, void* stream) {
  return feature_pyramid(f2, lvl1, lvl2, lvl3, P, h, w, false, stream, "pp_corr_feature_pyramid");
}

extern "C" int pp_corr_feature_pyramid_split(const void* f2, void* lvl1, void* lvl2, void* lvl3, int P, int h, int w, void* stream) {
  return feature_pyramid(f2, lvl1, lvl2, lvl3, P, h, w, true, stream, "pp_corr_feature_pyramid_split");
}

extern "C" int pp_corr_lookup_otf(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                  const void* f2_lvl3, const float* coords, void* out, int out_cstride, int out_cpad, int P,
                                  int h, int w, void* stream) {
  PP_REQUIRE(f1 && f2_lvl0 && f2_lvl1 && f2_lvl2 && f2_lvl3 && coords && out, PP_ERR_ARG, "pp_corr_lookup_otf: null pointer");
  PP_REQUIRE(P > 0 && (h >> 3) >= 2 && (w >> 3) >= 2, PP_ERR_ARG,
             "pp_corr_lookup_otf: level-3 map would be %dx%d (RAFT needs >= 2x2: inputs of at least 128x128)", h >> 3, w >> 3);
  PP_REQUIRE(out_cpad >= 324 && out_cpad <= OTF_OROW && out_cpad % 8 == 0 && out_cstride >= out_cpad && out_cstride % 8 == 0, PP_ERR_ARG,
             "pp_corr_lookup_otf: out_cpad %d (324..328, multiple of 8) / cstride %d", out_cpad, out_cstride);
  PP_REQUIRE((long long)h * w * 512 < (1ll << 31), PP_ERR_ARG, "pp_corr_lookup_otf: feature map of %dx%d exceeds 2 GiB per pair", h, w);
  PP_REQUIRE(((uintptr_t)f1 % 16) == 0 && ((uintptr_t)out % 16) == 0, PP_ERR_ALIGN, "pp_corr_lookup_otf: pointers must be 16-byte aligned");
  CorrOtfParams p;
  p.f1 = (const char*)f1;
  p.f2[0] = (const char*)f2_lvl0; p.f2[1] = (const char*)f2_lvl1; p.f2[2] = (const char*)f2_lvl2; p.f2[3] = (const char*)f2_lvl3;
  p.coords = coords; p.out = (_Float16*)out;
  p.P = P; p.h = h; p.w = w; p.ocs = out_cstride; p.ocpad = out_cpad;
  p.tiles_x = (w + 7) / 8; p.tiles_y = (h + 7) / 8;
  p.scale = 1.f / 16.f;
  const long long nblk = (long long)P * p.tiles_x * p.tiles_y;
  PP_REQUIRE(nblk < (1ll << 31), PP_ERR_ARG, "pp_corr_lookup_otf: too many tiles");
  hipLaunchKernelGGL(corr_otf_kernel, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, p);
  return launch_status("pp_corr_lookup_otf");
}

extern "C" int pp_corr_lookup_otf_split(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                        const void* f2_lvl3, const float* coords, void* out, int out_cstride, int P,
                                        int h, int w, void* stream) {
  PP_REQUIRE(f1 && f2_lvl0 && f2_lvl1 && f2_lvl2 && f2_lvl3 && coords && out, PP_ERR_ARG, "pp_corr_lookup_otf_split: null pointer");
  PP_REQUIRE(P > 0 && (h >> 3) >= 2 && (w >> 3) >= 2, PP_ERR_ARG,
             "pp_corr_lookup_otf_split: level-3 map would be %dx%d (RAFT needs >= 2x2: inputs of at least 128x128)", h >> 3, w >> 3);
  PP_REQUIRE(out_cstride >= 2 * 4 * OTFS_LVC && out_cstride % 16 == 0, PP_ERR_ARG,
             "pp_corr_lookup_otf_split: out_cstride %d (two planes of >= %d channels, multiple of 16)", out_cstride, 4 * OTFS_LVC);
  PP_REQUIRE((long long)h * w * 1024 < (1ll << 31), PP_ERR_ARG, "pp_corr_lookup_otf_split: feature map of %dx%d exceeds 2 GiB per pair", h, w);
  PP_REQUIRE(((uintptr_t)f1 % 16) == 0 && ((uintptr_t)out % 16) == 0, PP_ERR_ALIGN, "pp_corr_lookup_otf_split: pointers must be 16-byte aligned");
  CorrOtfParams p;
  p.f1 = (const char*)f1;
  p.f2[0] = (const char*)f2_lvl0; p.f2[1] = (const char*)f2_lvl1; p.f2[2] = (const char*)f2_lvl2; p.f2[3] = (const char*)f2_lvl3;
  p.coords = coords; p.out = (_Float16*)out;
  p.P = P; p.h = h; p.w = w; p.ocs = out_cstride; p.ocpad = 4 * OTFS_LVC;
  p.tiles_x = (w + 7) / 8; p.tiles_y = (h + 7) / 8;
  p.scale = 1.f / 16.f;
  const long long nblk = (long long)P * p.tiles_x * p.tiles_y;
  PP_REQUIRE(nblk < (1ll << 31), PP_ERR_ARG, "pp_corr_lookup_otf_split: too many tiles");
  hipLaunchKernelGGL(corr_otf_split_kernel, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, p);
  return launch_status("pp_corr_lookup_otf_split");
}
