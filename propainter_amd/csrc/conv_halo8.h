// Halo-tile implicit-GEMM convolution, PING-PONG form (round 6): one 512-thread block per CU owns a 256-pixel x BN-cout tile; its
// eight waves are two GROUPS of four (2 x 2 wave tiles of 64 px x BN/2 couts -- the fragment geometry, swizzles, K order and
// epilogue of conv_halo.h), and the two waves that share a SIMD always sit in OPPOSITE phases of a K step:
//
//      slot 2k     | group 0: READ(k)    = issue LDS-DMA for step k+2 / the next patch, 16 ds_read_b128 of step k, counted waits
//                  | group 1: MFMA(k-1)  = 32 (48 split-plane) MFMAs from registers, s_setprio 1
//      ---- s_barrier ----
//      slot 2k + 1 | group 0: MFMA(k)    | group 1: READ(k)
//      ---- s_barrier ----
//
// Why (docs/DESIGN_LOG.md section 3, profiles/r3q_halo_kernel_phases.txt): in conv_halo.h a wave's K step is a SERIAL chain
// [DMA issue 570 -> fragment reads 400 -> MFMAs 512 / 768 cycles] and the co-resident wave of another block hides it only by
// chance (MFMA pipes busy 0.26 / 0.43).  Here the chain of one wave runs under the MFMA phase of its SIMD partner by construction
// (MI355X_MICROARCH.md, "Two waves per SIMD": the complementary pairing of the tuned 8-wave loops), the weight stage of a step is
// shared by 256 pixels instead of 128 (half the L2 -> LDS weight stream per MFMA), and no wait in the loop drains the DMA queue:
// weights are requested TWO steps ahead into a ring of three stages and waited for with `s_waitcnt vmcnt(pieces issued this slot)`.
//
// LDS-DMA ordering (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it"):
//   * a piece issued in a wave's READ(k) is retired by that wave's counted wait at the END of its READ(k+1), followed by the slot
//     barrier; the first reader of stage k+2 is group 0 in slot 2(k+2), i.e. after the barriers that end READ(k+1) of BOTH groups;
//   * the next channel block's patch pieces are issued on taps 0 .. NTAPS-2 only, so the same argument covers the patch;
//   * every READ ends with lgkmcnt(0) BEFORE its barrier: a ring slot / patch buffer is re-targeted by DMA no earlier than one
//     barrier after the last ds_read of it was retired.
// Same MFMA sequence per accumulator as conv_halo.h (K blocks in table order, taps in order, K halves / plane products in order):
// results are BIT-IDENTICAL to the 128-pixel kernel (tests/test_ops_gpu.py::test_conv_kernel_families_are_bit_identical).
#pragma once
#include "conv_halo.h"

// Variant switches (A/B builds, tools/build_h8_variant.sh):
//   PP_H8_DMA_IN_MFMA 1 (default): the LDS-DMA pieces of a step are issued BETWEEN the MFMA clusters of the wave's MFMA phase (a piece
//                      costs ~60 cycles among bare MFMAs and 100-185 inside a phase that carries fragment reads: MI355X_MICROARCH.md),
//                      the READ phase is the 16 fragment reads and two waits only; 0: issued at the head of the READ phase (first form).
//   PP_H8_PRIO         0 none, 1 (default) s_setprio 1 over the MFMA phase, 2 s_setprio 1 over the READ phase (the partner's MFMAs
//                      need one issue slot in four; the READ phase is the serial chain).
#ifndef PP_H8_DMA_IN_MFMA
#define PP_H8_DMA_IN_MFMA 1
#endif
#ifndef PP_H8_PRIO
#define PP_H8_PRIO 1
#endif
#ifndef PP_H8_HOIST_TAPS
#define PP_H8_HOIST_TAPS 5      // tap windows of at most this many taps keep their swizzled fragment addresses in registers
#endif

namespace pp {

template <int TH, int TW, int KH, int KW, int BN, bool SPLIT>
__global__ __launch_bounds__(512, 1) void conv_halo8_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 T;
  constexpr int BM = TH * TW, NW = 8, GW = 4;                   // 256 px, 8 waves, 4 waves per group
  static_assert(BM == 256 && TW % 16 == 0 && (BN == 128 || BN == 64), "tile");
  constexpr int WM = 64, WN = BN / 2, TM = 4, TN = WN / 16;
  constexpr int PH = TH + KH - 1, PW = TW + KW - 1, P = PH * PW;
  constexpr int NTAPS = KH * KW;
  constexpr int PPW = (P + 8 * NW - 1) / (8 * NW);              // patch pieces per wave and channel block (8 rows each)
  constexpr int PIECES = PPW * NW;
  constexpr int PATCH_BYTES = PIECES * 1024;
  constexpr int BSTAGE = BN * 128;
  constexpr int B_PER_WAVE = BN / 8 / NW;                       // weight pieces per wave and stage: 2 (BN 128) / 1 (BN 64)
  constexpr int NSTAGE = 3;
  constexpr int PIPE_BYTES = 2 * PATCH_BYTES + NSTAGE * BSTAGE;
  constexpr int EPI_LD = WN + 4;
  constexpr int EPI_WAVE = WM * EPI_LD * 4;                     // wave-private staging tile of the epilogue
  constexpr int EPI_BYTES = NW * EPI_WAVE;
  constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  constexpr int TSPREAD = NTAPS - 1;                            // taps that carry pieces of the next patch (never the last one)
  static_assert(LDS_BYTES <= 160 * 1024 && NTAPS >= 2 && 32 * 1024 <= 4 * EPI_WAVE, "LDS budget");
  static_assert(B_PER_WAVE + (PPW + TSPREAD - 1) / TSPREAD <= 5, "counted wait ladder (H8_WAIT_VM)");

  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  char* const patch0 = lds;
  char* const bst0 = lds + 2 * PATCH_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wl = wave & 3;
  const int wm = wl >> 1, wn = wl & 1;

  // ---- XCD-aware block order: each XCD gets a contiguous run of tiles, couts fastest
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

  // ---- DMA roles (all eight waves): patch piece q = j * NW + wave covers patch rows q*8 .. q*8+7; weight pieces of a stage:
  // wave w fetches the consecutive 8-row groups w * B_PER_WAVE + j (one M0 value, instruction offsets)
  const int rin = lane >> 3, slot = lane & 7;
  const int lca = slot ^ rin;
  int ppix[PPW];
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
    const int q = wave * B_PER_WAVE + j;
    int row = n0 + q * 8 + rin;
    if (row >= p.cout_pad) row = p.cout_pad - 1;          // clamped rows feed accumulators that are never stored
    const int lcj = slot ^ ((4 * (q & 1) + (rin >> 1)) & 7);
    wvoff[j] = row * p.kchunks * 16 + lcj * 16 - j * 1024;
  }
  const int nrec = p.N * p.H * p.W;
  const int rb0 = p.src[0].cstride * 2, rb1 = p.src[1].cstride * 2, rb2 = p.src[2].cstride * 2, rb3 = p.src[3].cstride * 2;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[0].ptr + p.src[0].choff * 2), 0, nrec * rb0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[1].ptr + p.src[1].choff * 2), 0, nrec * rb1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[2].ptr + p.src[2].choff * 2), 0, nrec * rb2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[3].ptr + p.src[3].choff * 2), 0, nrec * rb3, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.weight), 0, p.cout_pad * p.kchunks * 16, 0x00020000);

#define H8_ISSUE_PIECE(j, pbuf, e, lob)                                                                         \
  do {                                                                                                          \
    const int s_ = (e)[2] & 0xff;                                                                               \
    const __amdgpu_buffer_rsrc_t r_ = s_ == 1 ? rs1 : s_ == 2 ? rs2 : s_ == 3 ? rs3 : rs0;                      \
    const int rowbytes_ = s_ == 1 ? rb1 : s_ == 2 ? rb2 : s_ == 3 ? rb3 : rb0;                                  \
    const int coff_ = SPLIT ? (lca & 3) * 16 + ((lca & 4) ? (lob) : 0) : lca * 16;                              \
    const int voff_ = ppix[j] >= 0 ? ppix[j] * rowbytes_ + (e)[3] * 2 + coff_ : (int)0x80000000;                \
    v3_dma16(r_, patch0 + (pbuf) * PATCH_BYTES + ((j) * NW + wave) * 1024, voff_, 0);                           \
  } while (0)
#define H8_ISSUE_B(ks_, st_) V3WeightPieces<0, B_PER_WAVE>::issue(rw, bst0 + (st_) * BSTAGE + wave * B_PER_WAVE * 1024, wvoff, (ks_) * 128)

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment geometry: A row of fragment t = patch row of tile pixel (grp * 128 + wm * 64 + t * 16 + l15) + tap shift
  const int l15 = lane & 15, l4 = lane >> 4;
  int pp0[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = grp * 128 + wm * WM + t * 16 + l15;
    pp0[t] = (m / TW) * PW + (m % TW);
  }
  const int b_off = (wn * WN + l15) * 128;
  const int bswz = (l15 >> 1) & 7;
  const int nblocks = p.kchunks / (8 * NTAPS);
  const int nk = nblocks * NTAPS;

  // ---- prologue: the patch of block 0 and the weight stages of steps 0 and 1, drained once
  {
    i32x4s e, el;
    v3_fetch_entry(p.ktable, e);
    if constexpr (SPLIT) v3_fetch_entry(p.ktable + 4, el);
    v3_entry_ready(e);
    if constexpr (SPLIT) v3_entry_ready(el);
    const int lob0 = SPLIT ? (el[3] - e[3]) * 2 : 0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) H8_ISSUE_PIECE(j, 0, e, lob0);
    H8_ISSUE_B(0, 0);
    if (nk > 1) H8_ISSUE_B(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  f16x8 af[2][TM], bf[2][TN];          // the fragments of one K step: read in READ(k), consumed in MFMA(k) one slot later
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int f = 0; f < TM; ++f) af[kk][f] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int f = 0; f < TN; ++f) bf[kk][f] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  // DMA of one step: the weight stage of step ks + 2 and, on taps 0 .. NTAPS-2, this wave's share of the next channel block's patch
  // (the last two steps / the last block issue nothing: no request is in flight when the loop ends and LDS becomes the epilogue's)
#define H8_ISSUE_WEIGHTS() do { if (ks + 2 < nk) H8_ISSUE_B(ks + 2, st2); } while (0)
#define H8_ISSUE_PATCH(t)                                                                                       \
  do {                                                                                                          \
    if (have_next) {                                                                                            \
      if ((t) == 0) {                       /* source of the next channel block: selected once per block, not per piece */ \
        v3_entry_ready(en);                                                                                     \
        if constexpr (SPLIT) { v3_entry_ready(enl); lobn = (enl[3] - en[3]) * 2; }                              \
        const int s_ = en[2] & 0xff;                                                                            \
        rsn = s_ == 1 ? rs1 : s_ == 2 ? rs2 : s_ == 3 ? rs3 : rs0;                                              \
        rbn = s_ == 1 ? rb1 : s_ == 2 ? rb2 : s_ == 3 ? rb3 : rb0;                                              \
        cbn = en[3] * 2 + (SPLIT ? (lca & 3) * 16 + ((lca & 4) ? lobn : 0) : lca * 16);                         \
      }                                                                                                         \
      if ((t) < TSPREAD) {                                                                                      \
        _Pragma("unroll") for (int jj = (t); jj < PPW; jj += TSPREAD) {                                         \
          const int voff_ = ppix[jj] >= 0 ? ppix[jj] * rbn + cbn : (int)0x80000000;                             \
          v3_dma16(rsn, patch0 + pnext * PATCH_BYTES + (jj * NW + wave) * 1024, voff_, 0);                      \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

#define H8_MFMA16(WB, AB)                                                                                       \
  _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                                \
    _Pragma("unroll") for (int b = 0; b < TM; ++b)                                                              \
      acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[WB][a], af[AB][b], acc[a][b], 0, 0, 0)

  // MFMA(ks): the products of one step from the fragment registers (same order per accumulator as conv_halo.h); with
  // PP_H8_DMA_IN_MFMA the step's DMA pieces go between the clusters
#define H8_MFMA(t)                                                                                              \
  do {                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    if (PP_H8_PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                         \
    if constexpr (SPLIT) {                                                                                      \
      H8_MFMA16(0, 1);                                                     /* W_hi x A_lo */                    \
      if (PP_H8_DMA_IN_MFMA) { __builtin_amdgcn_sched_barrier(0); H8_ISSUE_WEIGHTS(); __builtin_amdgcn_sched_barrier(0); } \
      H8_MFMA16(1, 0);                                                     /* W_lo x A_hi */                    \
      if (PP_H8_DMA_IN_MFMA) { __builtin_amdgcn_sched_barrier(0); H8_ISSUE_PATCH(t); __builtin_amdgcn_sched_barrier(0); }  \
      H8_MFMA16(0, 0);                                                     /* W_hi x A_hi */                    \
    } else {                                                                                                    \
      H8_MFMA16(0, 0);                                                                                          \
      if (PP_H8_DMA_IN_MFMA) { __builtin_amdgcn_sched_barrier(0); H8_ISSUE_WEIGHTS(); H8_ISSUE_PATCH(t); __builtin_amdgcn_sched_barrier(0); } \
      H8_MFMA16(1, 1);                                                                                          \
    }                                                                                                           \
    if (PP_H8_PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  } while (0)

  // READ(ks) at tap t: the 16 fragment reads of the step, retired before the slot barrier; then the DMA pieces this wave issued one
  // slot ago (PP_H8_DMA_IN_MFMA: in its MFMA phase, nothing since) must have landed: a plain vmcnt(0), never a drain of younger requests
#define H8_READ(t)                                                                                              \
  do {                                                                                                          \
    const int sh_ = ((t) / KW) * PW + ((t) % KW);                 /* compile-time after unrolling */             \
    if (PP_H8_PRIO == 2) __builtin_amdgcn_s_setprio(1);                                                         \
    if (!PP_H8_DMA_IN_MFMA) { H8_ISSUE_WEIGHTS(); H8_ISSUE_PATCH(t); }                                          \
    const char* sb_ = bst0 + st0 * BSTAGE;                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                          \
      _Pragma("unroll") for (int f = 0; f < TN; ++f)                                                            \
        bf[kk][f] = *reinterpret_cast<const f16x8*>(sb_ + b_off + f * 16 * 128 + (((kk * 4 + l4) ^ bswz) << 4)); \
      _Pragma("unroll") for (int f = 0; f < TM; ++f) {                                                          \
        int r0_ = pp0[f];                                                                                       \
        if (NTAPS > PP_H8_HOIST_TAPS) asm volatile("" : "+v"(r0_));   /* opaque per use: 9 taps x 8 hoisted addresses spill */ \
        const int row = r0_ + sh_;                                                                              \
        af[kk][f] = *reinterpret_cast<const f16x8*>(pcur + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));     \
      }                                                                                                         \
    }                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
    if (PP_H8_DMA_IN_MFMA) {                                                                                    \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    } else {                                                                                                    \
      const int npp_ = (t) < TSPREAD ? (PPW > (t) ? (PPW - (t) + TSPREAD - 1) / TSPREAD : 0) : 0;               \
      const bool more_b_ = ks + 2 < nk;                                                                         \
      if (more_b_ && have_next) H8_WAIT_VM(B_PER_WAVE + npp_);                                                  \
      else if (more_b_) H8_WAIT_VM(B_PER_WAVE);                                                                 \
      else if (have_next) H8_WAIT_VM(npp_);                                                                     \
      else H8_WAIT_VM(0);                                                                                       \
    }                                                                                                           \
    if (PP_H8_PRIO == 2) __builtin_amdgcn_s_setprio(0);                                                         \
  } while (0)

  // counted wait (first form): everything older than the `cnt` pieces issued in this slot has landed
#define H8_WAIT_VM(cnt)                                                                                         \
  do {                                                                                                          \
    if ((cnt) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
    else if ((cnt) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                       \
    else if ((cnt) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                       \
    else if ((cnt) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                       \
    else if ((cnt) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                       \
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                                       \
  } while (0)

  // ONE instruction stream for both groups; group 1 passes one extra barrier first and so runs one slot behind group 0 for the
  // whole loop (its READ(k) beside group 0's MFMA(k), its MFMA(k) beside group 0's READ(k + 1)); group 0 passes the matching
  // barrier after the loop.  s_barrier counts arrivals: the n-th barrier of every wave of the block pairs up, whatever code it sits in.
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int ks = 0, st0 = 0, st2 = 2;        // ring slot of step ks / of step ks + 2
  for (int blk = 0; blk < nblocks; ++blk) {
    const bool have_next = blk + 1 < nblocks;
    i32x4s en, enl;
    if (have_next) {
      v3_fetch_entry(p.ktable + (blk + 1) * (NTAPS * 8), en);
      if constexpr (SPLIT) v3_fetch_entry(p.ktable + (blk + 1) * (NTAPS * 8) + 4, enl);
    }
    int lobn = 0, rbn = 0, cbn = 0;
    __amdgpu_buffer_rsrc_t rsn = rs0;
    const char* pcur = patch0 + (blk & 1) * PATCH_BYTES;
    const int pnext = (blk + 1) & 1;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      H8_READ(t);
      __builtin_amdgcn_s_barrier();
      H8_MFMA(t);
      __builtin_amdgcn_s_barrier();
      ++ks;
      st0 = st0 == NSTAGE - 1 ? 0 : st0 + 1;
      st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();      // (pairs with the stagger barrier of group 1: every wave passes 2 nk + 1 barriers)

  // ---- pre-activation addend through the matrix cores (as conv_halo.h, PRE_MFMA): each GROUP stages the 128 px x BN fp16 addend
  // tile of its own pixels inside its own waves' epilogue staging area (the other group may still be one slot behind)
  bool preadd_in_acc = false;
  if (p.preadd != nullptr && p.out_scale == 1.f && p.cout_g % BN == 0 && (long long)nrec * p.preadd_cstride * 2 < (1ll << 31)) {
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.preadd + p.preadd_choff * 2), 0,
                                                                           nrec * p.preadd_cstride * 2, 0x00020000);
    char* const greg = lds + grp * (GW * EPI_WAVE);            // 2 x 16 KB used (BN 128) / 16 KB (BN 64)
    const int lcp = slot ^ rin;
    const int pj = BN == 128 ? wn : 0, c0 = BN == 128 ? 0 : wn * WN;
    const char* pp = greg + pj * 16384;
#pragma unroll
    for (int pl = 0; pl < (SPLIT ? 2 : 1); ++pl) {
      __syncthreads();
      const int plane_off = pl * p.preadd_lo * 2;
#pragma unroll
      for (int q = 0; q < 16 / GW; ++q) {
        const int piece = q * GW + wl;
        const int m = grp * 128 + piece * 8 + rin;             // tile pixel
        const int iy = ty0 + m / TW, ix = tx0 + m % TW;
        const bool ok = (iy < p.H) & (ix < p.W);
        const int base = ok ? ((n * p.H + iy) * p.W + ix) * (p.preadd_cstride * 2) + n0 * 2 + lcp * 16 : (int)0x80000000;
        v3_dma16(rsp, greg + piece * 1024, base, plane_off);
        if constexpr (BN == 128) v3_dma16(rsp, greg + 16384 + piece * 1024, base, plane_off + 128);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int f = 0; f < TN; ++f) {
        const int kk = (c0 + f * 16) >> 5, kpos = (c0 + f * 16) & 31;
        f16x8 idf;
#pragma unroll
        for (int i = 0; i < 8; ++i) idf[i] = (l4 * 8 + i == kpos + l15) ? (_Float16)1 : (_Float16)0;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          const int row = wm * WM + b * 16 + l15;              // row of the group's 128-pixel addend tile
          const f16x8 a8 = *reinterpret_cast<const f16x8*>(pp + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));
          acc[f][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, a8, acc[f][b], 0, 0, 0);
        }
      }
    }
    preadd_in_acc = true;
  }
  __syncthreads();                                    // LDS becomes the epilogue tiles

  struct RowMap {
    int m_base, ty0, tx0, H, W; long long nbase;
    __device__ __forceinline__ long long operator()(int prow) const {
      const int mt = m_base + prow;
      const int iy = ty0 + mt / TW, ix = tx0 + mt % TW;
      return (iy < H && ix < W) ? (nbase + iy) * W + ix : -1ll;
    }
  };
  const RowMap rowmap{grp * 128 + wm * WM, ty0, tx0, p.H, p.W, (long long)n * p.H};
  conv_epilogue<WM, WN, WN / 16, 0, true, true, SPLIT>(p, acc, lds + wave * EPI_WAVE, lane, n0 + wn * WN, 0, p.out, rowmap, nullptr, nullptr, preadd_in_acc);
#undef H8_ISSUE_PIECE
#undef H8_ISSUE_B
#undef H8_MFMA
#undef H8_MFMA16
#undef H8_ISSUE_WEIGHTS
#undef H8_ISSUE_PATCH
#undef H8_WAIT_VM
#undef H8_READ
#endif
}

template <int TH, int TW, int KH, int KW, int BN, bool SPLIT>
static int launch_h8(ConvParams p, hipStream_t stream) {
  p.tiles_n = (p.cout_g + BN - 1) / BN;
  const long long tiles = (long long)p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
  const long long nblk = tiles * p.tiles_n;
  if (nblk >= (1ll << 31)) return -1000;
  hipLaunchKernelGGL((conv_halo8_kernel<TH, TW, KH, KW, BN, SPLIT>), dim3((unsigned)nblk), dim3(512), 0, stream, p);
  return launch_status("pp_conv2d(halo8)");
}

}  // namespace pp
