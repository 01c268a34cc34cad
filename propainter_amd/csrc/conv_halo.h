// Halo-tile implicit-GEMM convolution kernel (see conv_gemm_v3.hip for the design notes); shared by the fp16 translation unit
// (conv_gemm_v3.hip) and the split-plane "f16x3" one (conv_gemm_v3s.hip).
#pragma once
#include "conv_epilogue.h"

namespace pp {

typedef __attribute__((address_space(3))) void* lptr3_t;

#if defined(__HIP_DEVICE_COMPILE__)
typedef int i32x4s __attribute__((ext_vector_type(4)));
// (free functions with by-value arguments, not capturing lambdas: a by-reference closure of buffer resources ends up in
// scratch memory, and scratch loads share vmcnt with the LDS-DMA stream)
static __device__ __forceinline__ void v3_fetch_entry(const int4* ptr, i32x4s& e) {   // scalar load; complete after v3_entry_ready
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(e) : "s"(ptr));
}
static __device__ __forceinline__ void v3_entry_ready(i32x4s& e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e)::"memory"); }
static __device__ __forceinline__ void v3_dma16(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr3_t)dst, 16, voff, soff, 0, 0);
}
// Several pieces into consecutive KBs of LDS from ONE M0 value: the 12-bit instruction offset IMM is added to the LDS address and to the
// global address alike (MUBUF-to-LDS addressing), so the caller's `voff` carries the global offset MINUS IMM.
template <int IMM>
static __device__ __forceinline__ void v3_dma16_imm(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr3_t)dst, 16, voff, soff, IMM, 0);
}
template <int J, int N>
struct V3WeightPieces {      // pieces J .. N-1 of a wave's weight stage: LDS dst + J KB, global voff[J] (compensated) + soff
  static __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, char* dst, const int (&voff)[N], int soff) {
    v3_dma16_imm<J * 1024>(r, dst, voff[J], soff);
    if constexpr (J + 1 < N) V3WeightPieces<J + 1, N>::issue(r, dst, voff, soff);
  }
};
#endif

// PROF: diagnostic build that accumulates s_memtime deltas per phase into pp_debug_conv_prof() (see tools/bench_conv.py)
static __device__ unsigned long long g_v3_prof[16];     // (per translation unit; [12] / [13] = earliest start / latest end stamp)

// WMT: pixel rows of a wave tile.  64 (default): 64 x 64 wave tiles, 4 waves per 128 x 128 block tile, 2 waves per SIMD.
// 32: 32 x 64 wave tiles, 8 waves per block tile, FOUR waves per SIMD at <= 128 registers -- more LDS fragment traffic per
// MFMA (0.75 vs 0.5 KB) for twice the latency hiding (ablations, tools/kbench impl 86..89: the phases of the 2-waves-per-SIMD
// kernel barely overlap -- MFMA 75 + fragment reads 43 + DMA 38 + epilogue/sync 40 us of a 193 us launch).
// SPLIT: split-plane ("f16x3", pp_conv_args_t.split == 2, "tri-product" K format) -- every fp16 operand is a hi plane + a lo plane,
// value = hi + lo.  A K block is 32 channels of BOTH planes: the 128-byte patch row of a pixel holds [32 ch hi | 32 ch lo] (the DMA
// fetches slots 0..3 from the hi plane and slots 4..7 from the lo plane, `lo offset` = table entry 4 minus entry 0 of the block),
// the weight row of a tap holds [32 ch W_hi | 32 ch W_lo], and a tap step multiplies the FOUR fragment sets it has read as
// W_hi x A_hi + W_hi x A_lo + W_lo x A_hi: 48 MFMAs per 16 fragment reads, one weight stage and one barrier (the plain fp16 step:
// 32 MFMAs per 16 reads; walking a block three times through the plain kernel: 48 MFMAs per 24 reads, 1.5 stages, 1.5 barriers).
// The epilogue reads its operands (preadd, residual, h, z) as hi + lo and writes two planes (conv_epilogue.h).
// PRIVB (BN <= 64): WAVE-PRIVATE weight stages.  A wave needs only the WN couts of its own column of the tile: with tiles of at most 64
// couts a private copy of that slice per wave (2 stages x WN x 128 B: 4 KB / 2 KB per stage for BN 64 / 16) fits next to the patches in
// the 80 KB of a block, every wave DMAs its own slice, and the tap steps of a channel block need NO block barrier -- only a counted
// s_waitcnt on the wave's own DMA; the barrier remains once per channel block (the patch is shared).  MFMAs per barrier: x KH*KW.
// (128-cout tiles would need 64 KB of private stages: one block per CU.)
// (two blocks of 128 pixels per CU: the allocator must stay within 256 registers -- the 5x1 128-cout tile took 259 and ran one wave per SIMD)
#if defined(PP_HALO_LB_ROUND2)   // (A/B builds: the bound the fp16 instantiations had before)
#define HALO_MIN_WAVES(th, tw, bn, wmt, split) ((split) && (bn) == 128 ? 2 : 1)
#else
#define HALO_MIN_WAVES(th, tw, bn, wmt, split) ((th) * (tw) == 128 && (wmt) == 64 && (bn) <= 128 ? 2 : 1)
#endif
template <int TH, int TW, int KH, int KW, int BN, bool PROF = false, int STAGGER = 0, int WMT = 64, bool SPLIT = false, bool PRIVB = false>
__global__ __launch_bounds__(TH * TW * 128 / WMT, WMT == 32 ? 4 : HALO_MIN_WAVES(TH, TW, BN, WMT, SPLIT)) void conv_halo_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 T;
  constexpr int BM = TH * TW;                                   // 128 px (2 blocks/CU) or 256 px (8 waves, 1 block/CU)
  constexpr int WAVES_N = BN >= 32 ? 2 : 1;                      // BN 16 (tiny cout): all waves along the pixels
  constexpr int NW = WMT == 128 ? BM / 64 : BM / 32 * (64 / WMT), WAVES_M = NW / WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int PH = TH + KH - 1, PW = TW + KW - 1, P = PH * PW;
  constexpr int NTAPS = KH * KW;
  constexpr int PIECES = (P + 8 * NW - 1) / (8 * NW) * NW;     // LDS-DMA instructions per patch (8 rows each)
  constexpr int PPW = PIECES / NW;                              // ... per wave
  constexpr int PATCH_BYTES = PIECES * 1024;
  static_assert(!PRIVB || (BN <= 64 && STAGGER == 0 && WMT == 64), "private weight stages: tiles of at most 64 couts");
  constexpr int BW_ROWS = PRIVB ? WN : BN;                     // weight rows of one stage buffer (the wave's own couts when private)
  constexpr int BSTAGE = BW_ROWS * 128;
  constexpr int B_INST = BW_ROWS / 8;                          // weight-tile DMA instructions per stage (8 rows each)
  constexpr int B_PER_WAVE = PRIVB ? B_INST : (B_INST + NW - 1) / NW;
  constexpr bool B_RAGGED = !PRIVB && (B_INST % NW) != 0;      // BN 16, shared stages: only waves 0..B_INST-1 fetch weights
  // BCONTIG (shared stages): a wave's B_PER_WAVE weight pieces are CONSECUTIVE 8-row groups of the stage, so one M0 value + the
  // instruction offset addresses all of them (an M0 rewrite between two LDS-DMA instructions serialises them)
  constexpr bool BCONTIG = !PRIVB && B_PER_WAVE <= 4;
  constexpr int PIPE_BYTES = 2 * PATCH_BYTES + (PRIVB ? 2 * NW : 2) * BSTAGE;
  constexpr int EPI_WN = WN > 64 ? 64 : WN;                    // the epilogue stages at most 64 couts of the wave tile at a time
  constexpr int EPI_LD = EPI_WN + 4;
  constexpr int EPI_BYTES = NW * WM * EPI_LD * 4;
  constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  constexpr bool PRE_MFMA = STAGGER == 0 && BM == 128 && (BN == 128 || BN == 64) && WMT == 64 && PATCH_BYTES >= 16 * 1024;   // (STAGGER 11, impl 109: off, for A/B)
  static_assert((BM == 128 || BM == 256) && (TW == 16 || TW == 8) && PPW <= 3 * NTAPS && (BN % 32 == 0 || BN == 16) &&
                    LDS_BYTES <= (BM == 128 ? 80 : 160) * 1024, "tile");

  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  unsigned long long pf_start = 0;
  if constexpr (PROF) pf_start = __builtin_readcyclecounter();
  char* const patch0 = lds;
  char* const bst0 = lds + 2 * PATCH_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- XCD-aware block order (as v2): each XCD gets a contiguous run of tiles, couts fastest
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tn = bid % p.tiles_n;
  int tile = bid / p.tiles_n;
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
  const int txi = tile % tiles_x; tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int n0 = tn * BN;

  // ---- DMA roles.  Patch piece q = j*NW + wave covers patch rows q*8 .. q*8+7; lane -> (row q*8 + lane/8, slot lane%8).
  // (q*8 + rin) >> 1 & 7 == (4*q + (rin >> 1)) & 7 and q has the parity of `wave` (NW is even): one logical chunk per lane.
  const int rin = lane >> 3, slot = lane & 7;
  const int lc = slot ^ ((4 * (wave & 1) + (rin >> 1)) & 7);        // weight rows: slot ^ ((row >> 1) & 7)
  const int lca = slot ^ rin;                                       // patch rows: slot ^ (row & 7); piece rows start at multiples of 8
  int ppix[PPW];                                          // global pixel index of the lane's patch row, -1 = zero fill
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int r = (j * NW + wave) * 8 + rin;
    const int py = r / PW, px = r - py * PW;
    const int iy = ty0 - p.ph + py, ix = tx0 - p.pw + px;
    const bool ok = (r < P) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
    ppix[j] = ok ? (n * p.H + iy) * p.W + ix : -1;
  }
  int wvoff[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    // shared stages: instruction j of the wave covers tile rows (j * NW + wave) * 8 ..; private stages: rows j * 8 .. of the wave's own
    // WN couts.  Weight rows are swizzled by slot ^ ((row >> 1) & 7) with the row counted inside the stage buffer.
    int row = PRIVB ? n0 + wn * WN + j * 8 + rin : (BCONTIG ? n0 + (wave * B_PER_WAVE + j) * 8 + rin : n0 + (j * NW + wave) * 8 + rin);
    if (row >= p.cout_pad) row = p.cout_pad - 1;          // clamped rows feed accumulators that are never stored
    const int lcj = PRIVB ? (slot ^ ((4 * (j & 1) + (rin >> 1)) & 7)) : (BCONTIG ? (slot ^ ((4 * ((wave * B_PER_WAVE + j) & 1) + (rin >> 1)) & 7)) : lc);
    wvoff[j] = row * p.kchunks * 16 + lcj * 16 - (BCONTIG ? j * 1024 : 0);      // (BCONTIG: minus the instruction offset of piece j)
  }
  const int nrec = p.N * p.H * p.W;
  // (individual scalars, not arrays: a dynamically indexed private array would live in scratch, and scratch loads share
  // vmcnt with the LDS-DMA stream)
  const int rb0 = p.src[0].cstride * 2, rb1 = p.src[1].cstride * 2, rb2 = p.src[2].cstride * 2, rb3 = p.src[3].cstride * 2;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[0].ptr + p.src[0].choff * 2), 0, nrec * rb0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[1].ptr + p.src[1].choff * 2), 0, nrec * rb1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[2].ptr + p.src[2].choff * 2), 0, nrec * rb2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[3].ptr + p.src[3].choff * 2), 0, nrec * rb3, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.weight), 0, p.cout_pad * p.kchunks * 16, 0x00020000);

  // (SPLIT: logical chunks 0..3 of a patch row are 32 channels of the hi plane, chunks 4..7 the same channels of the lo plane,
  //  `lob` bytes further -- lob = 2 * (choff of table entry 4 - choff of entry 0) of the block, wave-uniform)
#define V3_ISSUE_PIECE(j, pbuf, e, lob)                                                                         \
  do {                                                                                                          \
    const int s_ = (e)[2] & 0xff;                                                                               \
    const __amdgpu_buffer_rsrc_t r_ = s_ == 1 ? rs1 : s_ == 2 ? rs2 : s_ == 3 ? rs3 : rs0;                      \
    const int rowbytes_ = s_ == 1 ? rb1 : s_ == 2 ? rb2 : s_ == 3 ? rb3 : rb0;                                  \
    const int coff_ = SPLIT ? (lca & 3) * 16 + ((lca & 4) ? (lob) : 0) : lca * 16;                              \
    const int voff_ = ppix[j] >= 0 ? ppix[j] * rowbytes_ + (e)[3] * 2 + coff_ : (int)0x80000000;                \
    v3_dma16(r_, patch0 + (pbuf) * PATCH_BYTES + ((j) * NW + wave) * 1024, voff_, 0);                           \
  } while (0)
#define V3_ISSUE_B(ks_, par_)                                                                                   \
  do {                                                                                                          \
    if constexpr (BCONTIG) {                                                                                    \
      if (!B_RAGGED || wave < B_INST)                                                                           \
        V3WeightPieces<0, B_PER_WAVE>::issue(rw, bst0 + (par_) * BSTAGE + wave * B_PER_WAVE * 1024, wvoff, (ks_) * 128); \
    } else {                                                                                                    \
      _Pragma("unroll") for (int j_ = 0; j_ < B_PER_WAVE; ++j_)                                                 \
        if (!B_RAGGED || j_ * NW + wave < B_INST)                                                               \
          v3_dma16(rw, bst0 + (PRIVB ? wave * 2 + (par_) : (par_)) * BSTAGE + (PRIVB ? j_ : j_ * NW + wave) * 1024, wvoff[j_], (ks_) * 128); \
    }                                                                                                           \
  } while (0)

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment geometry: A row of fragment t = patch row pp0[t] + tap shift; B row = wn*WN + t*16 + (lane & 15)
  const int l15 = lane & 15, l4 = lane >> 4;
  int pp0[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = wm * WM + t * 16 + l15;
    pp0[t] = (m / TW) * PW + (m % TW);
  }
  const int b_off = (PRIVB ? l15 : wn * WN + l15) * 128;
  const int bswz = (l15 >> 1) & 7;
  const int nblocks = p.kchunks / (8 * NTAPS);
  const int nk = nblocks * NTAPS;
  unsigned long long pf_wait = 0, pf_vm = 0, pf_issue = 0, pf_comp = 0, pf_t0 = 0, pf_a = 0, pf_b = 0, pf_c = 0;
  if constexpr (PROF) pf_t0 = __builtin_readcyclecounter();
  // ---- prologue: patch of block 0 (all pieces) + weights of step 0
  {
    i32x4s e, el;
    v3_fetch_entry(p.ktable, e);
    if constexpr (SPLIT) v3_fetch_entry(p.ktable + 4, el);
    v3_entry_ready(e);
    if constexpr (SPLIT) v3_entry_ready(el);
    const int lob0 = SPLIT ? (el[3] - e[3]) * 2 : 0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) V3_ISSUE_PIECE(j, 0, e, lob0);
    V3_ISSUE_B(0, 0);
  }
  int ks = 0, par = 0;
  for (int blk = 0; blk < nblocks; ++blk) {
    const bool have_next = blk + 1 < nblocks;
    i32x4s en, enl;
    if (have_next) {
      v3_fetch_entry(p.ktable + (blk + 1) * (NTAPS * 8), en);
      if constexpr (SPLIT) v3_fetch_entry(p.ktable + (blk + 1) * (NTAPS * 8) + 4, enl);
    }
    int lobn = 0;
    const char* pcur = patch0 + (blk & 1) * PATCH_BYTES;
    const int pnext = (blk + 1) & 1;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      if constexpr (PROF) pf_a = __builtin_readcyclecounter();
      const int sh = (t / KW) * PW + (t % KW);          // compile-time after unrolling
      if constexpr (PRIVB) {
        // the wave's own weight stage of this step was requested one step ago, BEFORE that step's patch pieces: at most PCP younger
        // requests (pieces of the next block's patch) may stay in flight.  Only the first tap of a channel block needs the block: the
        // patch arrived from all four waves.
        const int tp = (t + NTAPS - 1) % NTAPS;                                              // tap of the previous step (folds after unrolling)
        const int pcp = PPW > tp ? (PPW - tp + NTAPS - 1) / NTAPS : 0;                       // pieces a step issues at tap tp
        if (t == 0 || !have_next || pcp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (pcp == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (pcp == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (PROF) { pf_b = __builtin_readcyclecounter(); pf_vm += pf_b - pf_a; }
      if (!PRIVB || t == 0) __builtin_amdgcn_s_barrier();          // weights of step ks (and, at t == 0, the whole patch of this block) are in LDS
      if constexpr (PROF) { pf_c = __builtin_readcyclecounter(); pf_wait += pf_c - pf_b; }
      if (t == 0 && have_next) {
        v3_entry_ready(en);
        if constexpr (SPLIT) { v3_entry_ready(enl); lobn = (enl[3] - en[3]) * 2; }
      }
      const bool more_b = ks + 1 < nk;
      if (more_b) V3_ISSUE_B(ks + 1, par ^ 1);
      if (have_next) {
#pragma unroll
        for (int jj = t; jj < PPW; jj += NTAPS) V3_ISSUE_PIECE(jj, pnext, en, lobn);     // (one piece per step for the shipped tiles: PPW <= NTAPS)
      }
      if constexpr (PROF) { pf_a = __builtin_readcyclecounter(); pf_issue += pf_a - pf_c; }
      const char* sb = bst0 + (PRIVB ? wave * 2 + par : par) * BSTAGE;
      if constexpr (SPLIT) {
        // tri-product step: fragments of both planes (kk 0 = hi / W_hi, kk 1 = lo / W_lo), then the three products with the
        // accumulators interleaved (an accumulator is revisited every 16 MFMAs; small terms first)
        f16x8 af[2][TM], bf[2][TN];
        PP_SPLIT_PRIO_EARLY();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int f = 0; f < TM; ++f) {
            const int row = pp0[f] + sh;
            af[kk][f] = *reinterpret_cast<const f16x8*>(pcur + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int f = 0; f < TN; ++f) {
            bf[kk][f] = *reinterpret_cast<const f16x8*>(sb + b_off + f * 16 * 128 + (((kk * 4 + l4) ^ bswz) << 4));
          }
        }
        PP_SPLIT_PRIO_BEGIN();
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[0][a], af[1][b], acc[a][b], 0, 0, 0);   // W_hi x A_lo
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[1][a], af[0][b], acc[a][b], 0, 0, 0);   // W_lo x A_hi
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[0][a], af[0][b], acc[a][b], 0, 0, 0);   // W_hi x A_hi
        PP_SPLIT_PRIO_END();
      } else if constexpr (STAGGER == 0 && BN == 128 && WMT == 64 && BM == 128) {
        // the fragments of BOTH K halves are requested before the first MFMA (32 more live fragment registers; 231 in all for 3x3): the
        // reads of the second half complete under the 16 MFMAs of the first instead of being waited for in four small groups
        // (+1.5 % on average over seven layer shapes, bit-identical: profiles/r3q_halo_kernel_phases.txt)
        f16x8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int f = 0; f < TN; ++f) bf[kk][f] = *reinterpret_cast<const f16x8*>(sb + b_off + f * 16 * 128 + (((kk * 4 + l4) ^ bswz) << 4));
#pragma unroll
          for (int f = 0; f < TM; ++f) {
            const int row = pp0[f] + sh;
            af[kk][f] = *reinterpret_cast<const f16x8*>(pcur + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_MFMA_PRIO_BEGIN();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[kk][a], af[kk][b], acc[a][b], 0, 0, 0);
        PP_MFMA_PRIO_END();
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int f = 0; f < TM; ++f) {
          const int row = pp0[f] + sh;
          af[f] = *reinterpret_cast<const f16x8*>(pcur + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int f = 0; f < TN; ++f) bf[f] = *reinterpret_cast<const f16x8*>(sb + b_off + f * 16 * 128 + (((kk * 4 + l4) ^ bswz) << 4));
        PP_MFMA_PRIO_BEGIN();
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[a], af[b], acc[a][b], 0, 0, 0);
        PP_MFMA_PRIO_END();
      }
      }
      if constexpr (PROF) { asm volatile("s_nop 0" ::: "memory"); pf_comp += __builtin_readcyclecounter() - pf_a; }
      ++ks;
      par ^= 1;
    }
  }
  if constexpr (PROF) pf_b = __builtin_readcyclecounter();
  // ---- pre-activation addend through the matrix cores (PRE_MFMA): out = act(conv + preadd) as one more K block with IDENTITY
  // weights -- the preadd tile (128 px x 64 couts per wave column, fp16) is LDS-DMA'd into the free patch buffers and multiplied
  // by 1.0 into the fp32 accumulators (exact), so the epilogue keeps its fast register path instead of reading the addend per
  // output row (measured +63 us on a 193 us launch, profiles/r2_conv_epilogue_ab.txt).  Each wave needs only the 64 addend
  // channels of its own cout half: 2 x 16 KB, 16 MFMAs per wave, identity fragments built in registers (no weight traffic).
  bool preadd_in_acc = false;
  if constexpr (PRE_MFMA) {
    if (p.preadd != nullptr && p.out_scale == 1.f && p.cout_g % BN == 0 &&        // (the addend is not scaled; full cout tiles only;
        (long long)nrec * p.preadd_cstride * 2 < (1ll << 31)) {                       //  32-bit buffer offsets)
      const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.preadd + p.preadd_choff * 2), 0,
                                                                             nrec * p.preadd_cstride * 2, 0x00020000);
      const int lcp = slot ^ rin;
      // the wave's couts are channels c0 .. c0 + WN - 1 of addend patch pj; cout fragment f sits in k chunk (c0 + 16 f) / 32
      const int pj = BN == 128 ? wn : 0, c0 = BN == 128 ? 0 : wn * WN;
      const char* pp = patch0 + pj * PATCH_BYTES;
#pragma unroll
      for (int pl = 0; pl < (SPLIT ? 2 : 1); ++pl) {       // split-plane addend: the hi plane, then the lo plane (both exact)
        __syncthreads();                              // every wave's fragment reads of the last K step (/ plane) are complete
        const int plane_off = pl * p.preadd_lo * 2;   // wave-uniform byte offset of the plane
#pragma unroll
        for (int q = 0; q < 16 / NW; ++q) {
          const int piece = q * NW + wave;
          const int m = piece * 8 + rin;                 // tile pixel (row-major in the TH x TW tile)
          const int iy = ty0 + m / TW, ix = tx0 + m % TW;
          const bool ok = (iy < p.H) & (ix < p.W);
          const int base = ok ? ((n * p.H + iy) * p.W + ix) * (p.preadd_cstride * 2) + n0 * 2 + lcp * 16 : (int)0x80000000;
          v3_dma16(rsp, patch0 + piece * 1024, base, plane_off);
          if constexpr (BN == 128) v3_dma16(rsp, patch0 + PATCH_BYTES + piece * 1024, base, plane_off + 128);     // the second 64 couts of the tile
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int f = 0; f < TN; ++f) {
          const int kk = (c0 + f * 16) >> 5, kpos = (c0 + f * 16) & 31;       // (kk is compile-time for BN 128, wave-uniform for BN 64)
          f16x8 idf;
#pragma unroll
          for (int i = 0; i < 8; ++i) idf[i] = (l4 * 8 + i == kpos + l15) ? (_Float16)1 : (_Float16)0;
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            const int row = wm * WM + b * 16 + l15;
            const f16x8 af = *reinterpret_cast<const f16x8*>(pp + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));
            acc[f][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, af, acc[f][b], 0, 0, 0);
          }
        }
      }
      preadd_in_acc = true;
    }
  }
  __syncthreads();                                    // LDS becomes the epilogue tile
  unsigned long long pf_e1 = 0, pf_e2 = 0;
  if constexpr (PROF) pf_e1 = __builtin_readcyclecounter();

  // ---- epilogue (conv_epilogue.h): wave-private staging tile at lds + wave * (EPI_BYTES / NW)
  if constexpr (PROF) pf_e2 = pf_e1;
  struct RowMap {
    int wm_base, ty0, tx0, H, W; long long nbase;
    __device__ __forceinline__ long long operator()(int prow) const {
      const int mt = wm_base + prow;
      const int iy = ty0 + mt / TW, ix = tx0 + mt % TW;
      return (iy < H && ix < W) ? (nbase + iy) * W + ix : -1ll;
    }
  };
  const RowMap rowmap{wm * WM, ty0, tx0, p.H, p.W, (long long)n * p.H};
  if constexpr (WN <= 64) {
    conv_epilogue<WM, WN, WN / 16, 0, (WM <= 64), true, SPLIT>(p, acc, lds + wave * (EPI_BYTES / NW), lane, n0 + wn * WN, 0, p.out, rowmap, nullptr,
                                                               PROF ? &pf_e2 : nullptr, preadd_in_acc);
  } else {
    // 128-cout wave tiles (BN 256, experimental): two passes of 64 couts through the same wave-private staging tile
    // (LDS operations of one wave execute in order, so the second pass cannot overtake the first pass's reads)
    static_assert(WN == 128, "wave tiles wider than 64 couts are drained in two halves");
    conv_epilogue<WM, 64, TN, 0>(p, acc, lds + wave * (EPI_BYTES / NW), lane, n0 + wn * WN, 0, p.out, rowmap);
    conv_epilogue<WM, 64, TN, 4>(p, acc, lds + wave * (EPI_BYTES / NW), lane, n0 + wn * WN + 64, 0, p.out, rowmap);
  }
  if constexpr (PROF) {
    const unsigned long long te = __builtin_readcyclecounter();
    if (lane == 0 && (blockIdx.x & 63) == 5) {        // a 1/64 sample of the blocks: the atomics must not perturb the run
      atomicAdd(&g_v3_prof[0], pf_vm); atomicAdd(&g_v3_prof[1], pf_wait); atomicAdd(&g_v3_prof[2], pf_issue);
      atomicAdd(&g_v3_prof[3], pf_comp); atomicAdd(&g_v3_prof[4], pf_b - pf_t0); atomicAdd(&g_v3_prof[5], te - pf_b);
      atomicAdd(&g_v3_prof[8], pf_e1 - pf_b); atomicAdd(&g_v3_prof[9], pf_e2 - pf_e1); atomicAdd(&g_v3_prof[10], te - pf_e2);
      atomicAdd(&g_v3_prof[11], pf_t0 - pf_start);
      atomicAdd(&g_v3_prof[6], 1ull); atomicAdd(&g_v3_prof[7], (unsigned long long)nk);
      atomicMin(&g_v3_prof[12], pf_start); atomicMax(&g_v3_prof[13], te);
    }
  }
#undef V3_ISSUE_PIECE
#undef V3_ISSUE_B
#endif
}

template <int TH, int TW, int KH, int KW, int BN, bool PROF = false, int STAGGER = 0, int WMT = 64, bool SPLIT = false, bool PRIVB = false>
static int launch_v3(ConvParams p, hipStream_t stream) {
  p.tiles_n = (p.cout_g + BN - 1) / BN;
  const long long tiles = (long long)p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
  const long long nblk = tiles * p.tiles_n;
  if (nblk >= (1ll << 31)) return -1000;
  hipLaunchKernelGGL((conv_halo_kernel<TH, TW, KH, KW, BN, PROF, STAGGER, WMT, SPLIT, PRIVB>), dim3((unsigned)nblk), dim3(TH * TW * 128 / WMT), 0, stream, p);
  return launch_status("pp_conv2d(v3)");
}

}  // namespace pp
