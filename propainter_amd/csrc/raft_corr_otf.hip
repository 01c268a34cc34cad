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
#include <stdlib.h>

namespace pp {

struct CorrOtfParams {
  const char* f1;            // fp16 NHWC [P, h, w, 256]
  const char* f2[4];         // fp16 NHWC [P, h >> l, w >> l, 256]
  const float* coords;       // fp32 [P, h, w, 2] (x, y)
  _Float16* out;             // fp16 NHWC [P, h, w, ocs]; channels [0, 324) written, [324, ocpad) zeroed
  int P, h, w, ocs, ocpad, tiles_x, tiles_y;
  float scale;               // 1 / sqrt(256)
  int dbg;                   // [PP_OTF_DBG, tuning only: selects the instrumented instantiation] 1 no blend, 2 no MFMA, 4 no B loads, 8 no write-out, 16 stamps
  unsigned long long* stats; // optional (nullptr: off) fallback counters, one update per block: [0] += (tile, level) units, [1] += units whose
                             // bounding box outgrew the V tile (sub-tile pass), [2] += 16-pixel groups that fell through to single pixels
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
  int n_sub = 0, n_single = 0;   // fallback counts of this block (p.stats)

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
      ++n_sub;
      for (int g = 0; g < 4; ++g) {
        if (wave == 0) wave_box(lvl, g * 16, 16, boxes + 16);
        __syncthreads();
        const bool ok = process(g * 16, 16, boxes + 16);
        if (!ok) {
          ++n_single;
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

  if (p.stats != nullptr && tid == 0) {       // (uniform counts: every thread of the block took the same path)
    atomicAdd(p.stats + 0, 4ull);
    if (n_sub) atomicAdd(p.stats + 1, (unsigned long long)n_sub);
    if (n_single) atomicAdd(p.stats + 2, (unsigned long long)n_single);
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
// Shape of the kernel, and why it differs from the fp16 one above (measured on MI355X, tools/bench_otf.py + PP_OTF_DBG ablations,
// profiles/r4_corr_otf_split.txt): a first version kept the 8 x 8 tile / 512-thread block (A = 64 px x 1 KB no longer fits one wave's
// registers, so two groups of four waves each owned 32 pixels and BOTH streamed every B tile) and ran 65 us per block of which the
// phases -- prologue + V stores 19, B loads 21, blend 18, write-out 3 -- simply added up: one block per CU, every phase behind a
// block barrier, nothing to overlap with.  So:
//   * a block is a 4 x 8 pixel tile and 256 threads: all four waves hold the SAME A fragments (2 M tiles x 8 K steps, hi + lo: 128
//     VGPRs) and share the box's N tiles round-robin -- every B tile is loaded once per block -- and TWO blocks live on a CU
//     (54 KB of LDS each, 2 waves per SIMD at <= 256 registers): one block's loads and barriers hide behind the other's blend;
//   * B arrives in K halves (4 hi + 4 lo 16-byte loads per lane), two half buffers in flight alternately: the loads of half k + 1
//     are issued before the MFMAs of half k;
//   * the blend reads per-(pixel, tap column) / per-(pixel, tap row) tables {cell index, fraction} built once per level (18 grid
//     round trips per pixel instead of 162: the round trip of tap (a, b) depends on a for x and on b for y only) -- the same values,
//     so the arithmetic of every output is still pp_corr_lookup's;
//   * output channels are laid out PER LEVEL in 88-channel groups (81 taps + 7 zeros: 176 bytes = 11 16-byte chunks per plane), so a
//     level's results leave LDS right after its blend (11 KB staging instead of whole rows) and the 1x1 convolution behind it
//     (convc1) walks 11 full 32-channel blocks per plane (328 channels: 10 full + 1 ragged).  Its weight columns are permuted
//     accordingly by the engine (flow_comp_raft.py).
//   out: fp16 [P, h, w, ocs], hi plane at channel 0, lo plane at ocs / 2; level l, tap (a, b) at channel l * 88 + a * 9 + b.
// A box too large for the V tile falls back to the two 2 x 8 halves (one M tile each), then to single pixels -- slower, never wrong.
// [tuning only, PP_OTF_DBG bit 16] cycle stamps of wave 0 summed over the blocks: {prologue, tables + loads + MFMA + V stores, barrier after
// them, blend, barrier after it, write-out, whole block, blocks}
__device__ unsigned long long g_otf_prof[8];
constexpr int OTFS_LVC = 88;                        // channels per level group
constexpr int OTFS_NPX = 32;                        // pixels per block: 4 rows x 8 columns
constexpr int OTFS_STRIP = 5;                       // tile columns per strip of the block order (see the kernel)
constexpr int OTFS_VTOT = OTFS_NPX * 324;           // floats of V (41 KB): 32 x 324 (20 N tiles; 324 = 4 mod 32: conflict-free tile stores), 16 x 648, 1 x 10368
constexpr int OTFS_STAGE = OTFS_NPX * 2 * OTFS_LVC * 2;   // bytes: [32 px][hi | lo][88]
constexpr int OTFS_TAB = OTFS_NPX * 18 * 8;         // bytes: [32 px][9 a + 9 b] x {int cell, float fraction}
constexpr int OTFS_LDS = OTFS_VTOT * 4 + OTFS_STAGE + OTFS_TAB + OTFS_NPX * 8 + 5 * 16;

template <bool PROF>
__global__ __launch_bounds__(256, 2) void corr_otf_split_kernel(const CorrOtfParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[OTFS_LDS];
  float* const V = reinterpret_cast<float*>(lds);
  _Float16* const stage = reinterpret_cast<_Float16*>(lds + OTFS_VTOT * 4);
  int2* const tab = reinterpret_cast<int2*>(lds + OTFS_VTOT * 4 + OTFS_STAGE);                  // {cell (level coordinates), fraction bits}
  float* const cxy = reinterpret_cast<float*>(lds + OTFS_VTOT * 4 + OTFS_STAGE + OTFS_TAB);     // [32][2]; x = NaN: pixel outside the image
  int* const boxes = reinterpret_cast<int*>(lds + OTFS_VTOT * 4 + OTFS_STAGE + OTFS_TAB + OTFS_NPX * 8);   // [5][4]: levels 0..3 of the whole tile, [4] scratch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  constexpr bool prof = PROF;
  unsigned long long tprev = prof ? __builtin_amdgcn_s_memtime() : 0ull, tstart = tprev, tacc[6] = {0, 0, 0, 0, 0, 0};
  auto stamp = [&](int slot) {
    if (PROF && (p.dbg & 16)) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tacc[slot] += t - tprev;
      tprev = t;
    }
  };

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // tile order inside a pair: COLUMN STRIPS of OTFS_STRIP tiles (40 pixels) walked top to bottom.  A tile's level-0 box is 14 rows x 18
  // positions x 1 KB; row-major over the whole 160-pixel map keeps 14 full rows (2.2 MB + the other levels) live between two tile
  // rows -- with the f1 stream and the output rows more than an XCD's 4 MB L2 holds, and f2 was fetched 1.8 x (PMC, profiles/r4y_*);
  // a strip re-uses 10 of its 14 box rows out of ~0.7 MB.
  const int per_pair = p.tiles_x * p.tiles_y;
  const int n = bid / per_pair;
  int txi, tyi;
  {
    const int t = bid - n * per_pair;
    const int full = (p.tiles_x / OTFS_STRIP) * OTFS_STRIP * p.tiles_y;      // tiles inside full-width strips
    if (t < full) {
      const int s_ = t / (OTFS_STRIP * p.tiles_y), r_ = t - s_ * (OTFS_STRIP * p.tiles_y);
      tyi = r_ / OTFS_STRIP;
      txi = s_ * OTFS_STRIP + (r_ - tyi * OTFS_STRIP);
    } else {                                                                 // the ragged last strip
      const int wlast = p.tiles_x - (p.tiles_x / OTFS_STRIP) * OTFS_STRIP, r_ = t - full;
      tyi = r_ / wlast;
      txi = (p.tiles_x / OTFS_STRIP) * OTFS_STRIP + (r_ - tyi * wlast);
    }
  }
  const int ty0 = tyi * 4, tx0 = txi * 8;
  // pixel q of the tile: row q >> 3, column q & 7 (M tile q >> 4 = rows 2m, 2m + 1)
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

  // ---- f1 tile (32 pixels x 1 KB: 32 hi chunks | 32 lo chunks) once per block through LDS (aliases V; chunk j of pixel q at j ^ (q & 31)).
  //      Its loads are issued FIRST: they do not depend on the coordinates, whose own load latency would otherwise precede them
  u32x4 f1raw[8];
  {
    const __amdgpu_buffer_rsrc_t r1 = uniform_buffer_rsrc(p.f1 + (long long)n * p.h * p.w * 1024, p.h * p.w * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + 256 * i, q = id >> 6, j = id & 63;
      const int x = tx0 + (q & 7), y = ty0 + (q >> 3);
      const int voff = (x < p.w && y < p.h) ? (y * p.w + x) * 1024 + j * 16 : (int)0x80000000;
      f1raw[i] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
    }
  }
  {
    const int q = tid & 31;
    const int x = tx0 + (q & 7), y = ty0 + (q >> 3);
    float cx = __builtin_nanf(""), cy = 0.f;
    if (x < p.w && y < p.h) {
      const float2 c = *reinterpret_cast<const float2*>(p.coords + (((long long)n * p.h + y) * p.w + x) * 2);
      cx = c.x;
      cy = c.y;
    }
    if (tid < OTFS_NPX) {
      cxy[tid * 2] = cx;
      cxy[tid * 2 + 1] = cy;
    }
  }
  // the 7 pad channels of every staging row stay zero for the whole block (the blend writes channels 0..80 only)
  for (int i = tid; i < OTFS_NPX * 2 * 7; i += 256) stage[(i / 7) * OTFS_LVC + 81 + (i % 7)] = (_Float16)0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = tid + 256 * i, q = id >> 6, j = id & 63;
    *reinterpret_cast<u32x4*>(lds + q * 1024 + ((j ^ (q & 31)) << 4)) = f1raw[i];
  }
  __syncthreads();
  wave_box(wave, 0, OTFS_NPX, boxes + wave * 4);           // wave l computes the whole-tile box of level l
  // ---- A fragments of the two M tiles, both planes: resident for the whole block (the same in all four waves)
  f16x8 ahi[2][8], alo[2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int q = m * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      ahi[m][ks] = *reinterpret_cast<const f16x8*>(lds + q * 1024 + (((ks * 4 + l4) ^ (q & 31)) << 4));
      alo[m][ks] = *reinterpret_cast<const f16x8*>(lds + q * 1024 + (((32 + ks * 4 + l4) ^ (q & 31)) << 4));
    }
  }
  __syncthreads();
  stamp(0);

  // K half `half` (channels 128 half .. +127 of both planes) of N tile nt: 16 positions straight from L2 (64-byte sectors used in full)
  auto load_half = [&](const __amdgpu_buffer_rsrc_t r2, int Wl, int bx0, int by0, int bw, int area, int nt, int half, u32x4 (&bh)[4], u32x4 (&bl)[4]) {
    const int pos = nt * 16 + l15;
    const int ry = pos / bw, rx = pos - ry * bw;
    const int voff = (pos < area && !(PROF && (p.dbg & 4))) ? ((by0 + ry) * Wl + bx0 + rx) * 1024 + l4 * 16 + half * 256 : (int)0x80000000;
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
    if (PROF && (p.dbg & 2)) { acc[0] += __builtin_bit_cast(float, bh[0] ^ bl[1]); return; }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bl), ah, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh), al, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bh), ah, acc, 0, 0, 0);
  };

  u32x4 b0h[4], b0l[4], b1h[4], b1l[4];
  bool prefetched = false;       // (b0h, b0l) already hold half 0 of this wave's first N tile of the level
  int n_sub = 0, n_single = 0;   // fallback counts of this block (p.stats)

  for (int lvl = 0; lvl < 4; ++lvl) {
    const int Hl = p.h >> lvl, Wl = p.w >> lvl;
    const float lscale = 1.f / (float)(1 << lvl);
    const __amdgpu_buffer_rsrc_t r2 = level_rsrc(lvl);

    // -- per-level tap tables (pp_corr_lookup's per-tap arithmetic, evaluated once per (pixel, a) and (pixel, b)):
    //    entry (q, a) = {floor(px), px - floor(px)}, px = roundtrip(cx + a - 4, Wl); entry (q, 9 + b) likewise along y
    for (int i = tid; i < OTFS_NPX * 18; i += 256) {
      const int q = i / 18, t = i - q * 18;
      const bool isy = t >= 9;
      const float c0 = cxy[q * 2 + (isy ? 1 : 0)];
      const float pc = grid_roundtrip(c0 * lscale + (float)((isy ? t - 9 : t) - 4), isy ? Hl : Wl);
      const float fl = floorf(pc);
      // (a NaN / far-out coordinate: the cell is clamped into int range, every corner then fails the box test -> zeros, as outside the map)
      const float flc = fminf(fmaxf(fl, -1.0e6f), 1.0e6f);
      tab[i] = make_int2((pc == pc) ? (int)flc : -(1 << 20), __builtin_bit_cast(int, pc - fl));
    }

    auto process = [&](const int p0, const int np, const int* bx) -> bool {
      const int bx0 = bx[0], by0 = bx[1], bw = bx[2], bh_ = bx[3];
      const int area = bw * bh_;
      const int vstride = OTFS_VTOT / np;
      const int ntiles = (area + 15) >> 4;
      if (ntiles * 16 > vstride) return false;
      const int mt0 = p0 >> 4;                         // first M tile of the set
      const bool one_tile = np <= 16;
      if (!prefetched && wave < ntiles) load_half(r2, Wl, bx0, by0, bw, area, wave, 0, b0h, b0l);
      prefetched = false;
      for (int nt = wave; nt < ntiles; nt += 4) {
        load_half(r2, Wl, bx0, by0, bw, area, nt, 1, b1h, b1l);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (one_tile && m != mt0) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k) tri(acc[m], b0h[k], b0l[k], ahi[m][k], alo[m][k]);
        }
        if (nt + 4 < ntiles) load_half(r2, Wl, bx0, by0, bw, area, nt + 4, 0, b0h, b0l);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (one_tile && m != mt0) continue;
#pragma unroll
          for (int k = 0; k < 4; ++k) tri(acc[m], b1h[k], b1l[k], ahi[m][4 + k], alo[m][4 + k]);
        }
        // acc[m][r] = S[position nt*16 + l4*4 + r][pixel m*16 + l15]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (one_tile && m != mt0) continue;
          const int q = m * 16 + l15 - p0;
          if (q >= 0 && q < np) *reinterpret_cast<f32x4*>(V + q * vstride + nt * 16 + l4 * 4) = acc[m];
        }
      }
      if (np == OTFS_NPX && lvl < 3) {       // half 0 of this wave's first N tile of the NEXT level: its latency hides behind the blend
        const int* nb = boxes + (lvl + 1) * 4;
        const int narea = nb[2] * nb[3], nnt = (narea + 15) >> 4;
        if (nnt * 16 <= OTFS_VTOT / OTFS_NPX) {
          if (wave < nnt) load_half(level_rsrc(lvl + 1), p.w >> (lvl + 1), nb[0], nb[1], nb[2], narea, wave, 0, b0h, b0l);
          prefetched = true;
        }
      }
      stamp(1);
      __syncthreads();
      stamp(2);
      // -- blend: one work item per output (pixel, a, b), the four corners accumulated in pp_corr_lookup's order; corners outside the box
      //    are outside the map (zeros) -- or, when a tap's round trip crosses an integer by one ulp, one cell beyond the box, where
      //    their weight is <= 1 ulp of the coordinate
#pragma unroll 2
      for (int item = tid; item < ((PROF && (p.dbg & 1)) ? 0 : np * 81); item += 256) {
        const int q = item / 81, jj = item - q * 81;
        const int b = jj / 9, a = jj - b * 9;            // a (the tap's x offset) fastest over the lanes: consecutive V columns, no bank conflicts
        const int j = a * 9 + b;                         // output channel inside the level group (first index moves x: RAFT/corr.py:36-43)
        const int2 ex = tab[(p0 + q) * 18 + a], ey = tab[(p0 + q) * 18 + 9 + b];
        const float lx = __builtin_bit_cast(float, ex.y), ly = __builtin_bit_cast(float, ey.y);
        const int c0 = ex.x - bx0, r0 = ey.x - by0;
        const float* vrow = V + q * vstride;
        float acc = 0.f, sv[4], wg[4];
        int okm = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = r0 + (k >> 1), cc = c0 + (k & 1);
          const float wgt = ((k & 1) ? lx : 1.f - lx) * ((k >> 1) ? ly : 1.f - ly);
          const bool ok = (unsigned)rr < (unsigned)bh_ && (unsigned)cc < (unsigned)bw;
          sv[k] = vrow[ok ? rr * bw + cc : 0];
          okm |= ok ? (1 << k) : 0;
          wg[k] = wgt;
        }
        // (the four reads are issued back to back, unconditionally: left to itself hipcc sinks each one into its own `ok` branch with a
        //  full lgkmcnt(0) wait -- four serialised LDS round trips per output, 3.6k cycles per output and wave measured)
        asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float s = ((okm >> k) & 1) ? sv[k] * p.scale : 0.f;      // (the volume holds f1 . f2 / 16: RAFT/corr.py:60)
          acc += wg[k] * s;
        }
        const _Float16 hi = (_Float16)acc;
        _Float16* so = stage + (p0 + q) * 2 * OTFS_LVC + j;
        so[0] = hi;
        so[OTFS_LVC] = (_Float16)(acc - (float)hi);
      }
      stamp(3);
      __syncthreads();
      stamp(4);
      return true;
    };

    if (!process(0, OTFS_NPX, boxes + lvl * 4)) {
      prefetched = false;
      ++n_sub;
      for (int g = 0; g < 2; ++g) {
        if (wave == 0) wave_box(lvl, g * 16, 16, boxes + 16);
        __syncthreads();
        const bool ok = process(g * 16, 16, boxes + 16);
        if (!ok) {
          ++n_single;
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
    for (int o = tid; o < ((PROF && (p.dbg & 8)) ? 0 : OTFS_NPX * 2 * 11); o += 256) {
      const int q = o / 22, r = o - q * 22, pl = r / 11, c = r - pl * 11;
      const int x = tx0 + (q & 7), y = ty0 + (q >> 3);
      if (x < p.w && y < p.h) {
        u32x4* dst = reinterpret_cast<u32x4*>(p.out + (((long long)n * p.h + y) * p.w + x) * p.ocs + pl * (p.ocs >> 1) + lvl * OTFS_LVC + c * 8);
        const u32x4 val = *reinterpret_cast<const u32x4*>(stage + (q * 2 + pl) * OTFS_LVC + c * 8);
        *dst = val;
      }
    }
    stamp(5);
    // (the next level's tables / blend rewrite `tab` / the staging tile only after a __syncthreads() every thread reaches after these reads:
    //  the table loop writes `tab`, which the write-out does not read; the blend comes after process()'s barrier)
  }
  if (p.stats != nullptr && tid == 0) {
    atomicAdd(p.stats + 0, 4ull);
    if (n_sub) atomicAdd(p.stats + 1, (unsigned long long)n_sub);
    if (n_single) atomicAdd(p.stats + 2, (unsigned long long)n_single);
  }
  if (PROF && (p.dbg & 16) && tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&g_otf_prof[i], tacc[i]);
    atomicAdd(&g_otf_prof[6], __builtin_amdgcn_s_memtime() - tstart);
    atomicAdd(&g_otf_prof[7], 1ull);
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

extern "C" int pp_corr_lookup_otf_stats(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                        const void* f2_lvl3, const float* coords, void* out, int out_cstride, int out_cpad, int P,
                                        int h, int w, unsigned long long* stats, void* stream);

extern "C" int pp_corr_lookup_otf(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                  const void* f2_lvl3, const float* coords, void* out, int out_cstride, int out_cpad, int P,
                                  int h, int w, void* stream) {
  return pp_corr_lookup_otf_stats(f1, f2_lvl0, f2_lvl1, f2_lvl2, f2_lvl3, coords, out, out_cstride, out_cpad, P, h, w, nullptr, stream);
}

extern "C" int pp_corr_lookup_otf_stats(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                        const void* f2_lvl3, const float* coords, void* out, int out_cstride, int out_cpad, int P,
                                        int h, int w, unsigned long long* stats, void* stream) {
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
  p.dbg = 0;
  p.stats = stats;
  const long long nblk = (long long)P * p.tiles_x * p.tiles_y;
  PP_REQUIRE(nblk < (1ll << 31), PP_ERR_ARG, "pp_corr_lookup_otf: too many tiles");
  hipLaunchKernelGGL(corr_otf_kernel, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, p);
  return launch_status("pp_corr_lookup_otf");
}

extern "C" int pp_corr_lookup_otf_split_stats(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                              const void* f2_lvl3, const float* coords, void* out, int out_cstride, int P,
                                              int h, int w, unsigned long long* stats, void* stream);

extern "C" int pp_corr_lookup_otf_split(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                        const void* f2_lvl3, const float* coords, void* out, int out_cstride, int P,
                                        int h, int w, void* stream) {
  return pp_corr_lookup_otf_split_stats(f1, f2_lvl0, f2_lvl1, f2_lvl2, f2_lvl3, coords, out, out_cstride, P, h, w, nullptr, stream);
}

extern "C" int pp_corr_lookup_otf_split_stats(const void* f1, const void* f2_lvl0, const void* f2_lvl1, const void* f2_lvl2,
                                              const void* f2_lvl3, const float* coords, void* out, int out_cstride, int P,
                                              int h, int w, unsigned long long* stats, void* stream) {
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
  p.tiles_x = (w + 7) / 8; p.tiles_y = (h + 3) / 4;
  p.scale = 1.f / 16.f;
  static const int dbg = getenv("PP_OTF_DBG") ? atoi(getenv("PP_OTF_DBG")) : 0;
  p.dbg = dbg;
  p.stats = stats;
  const long long nblk = (long long)P * p.tiles_x * p.tiles_y;
  PP_REQUIRE(nblk < (1ll << 31), PP_ERR_ARG, "pp_corr_lookup_otf_split: too many tiles");
  if (dbg) hipLaunchKernelGGL(corr_otf_split_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(corr_otf_split_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  return launch_status("pp_corr_lookup_otf_split");
}

// [diagnostic, not part of the public header] reads and clears the phase counters of pp_corr_lookup_otf_split (PP_OTF_DBG bit 16)
extern "C" int pp_debug_otf_prof(unsigned long long* out) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::g_otf_prof), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(pp::g_otf_prof), zero, sizeof(zero));
  return (int)e;
}
