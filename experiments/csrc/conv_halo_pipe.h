// Halo-tile implicit-GEMM convolution, SOFTWARE-PIPELINED form of conv_halo.h's kernel: the fragment reads of tap step k+1 are issued
// BEFORE the MFMAs of step k (two fragment register sets), so their LDS latency -- ~400 of the ~1 730 cycles a wave spends per step in
// the plain kernel (profiles/r3q_halo_kernel_phases.txt: 16 ds_read_b128 waited for ahead of 32 MFMAs = 512 pipe cycles) -- runs under
// the matrix work instead of in front of it.
//
// Same LDS image, swizzles, DMA roles, K table, packed weights, PRE_MFMA addend path and epilogue as conv_halo_kernel (shared weight
// stages, 128-pixel tiles, 4 waves = 2 x 2, two blocks per CU); results are bit-identical (same products in the same order).  What moves:
//
//   step k (tap t of channel block b), fragments of step k already requested into register set k & 1:
//     s_waitcnt vmcnt(0) lgkmcnt(0)   this wave's DMA (weights k+1, patch pieces so far) landed; its fragment reads of step k returned
//     s_barrier                       ... for every wave: weights k+1 complete, weight stage k & 1 and (at the last tap) the old patch free
//     DMA   weights k+2 -> stage k & 1; patch piece(s) of block b+1 (all issued by tap NTAPS-2: the last step reads the NEXT patch)
//     READ  fragments of step k+1 (stage (k+1) & 1; patch of block b, or of block b+1 after the last tap) -> set (k+1) & 1
//     MFMA  step k from set k & 1
//
// Two weight stages suffice: the reads of a stage are complete (lgkmcnt) before the barrier after which that stage is refilled.  The
// register sets are selected at compile time: the tap loop is unrolled and channel blocks are walked in pairs (NTAPS is odd for every
// shipped window, so the set parity flips from block to block).
#pragma once
#include "conv_halo.h"

namespace pp {

template <int TH, int TW, int KH, int KW, int BN, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_halo_pipe_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 T;
  constexpr int BM = TH * TW;
  constexpr int WAVES_N = 2, NW = 4, WAVES_M = 2;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int PH = TH + KH - 1, PW = TW + KW - 1, P = PH * PW;
  constexpr int NTAPS = KH * KW;
  constexpr int PIECES = (P + 8 * NW - 1) / (8 * NW) * NW;
  constexpr int PPW = PIECES / NW;
  constexpr int PATCH_BYTES = PIECES * 1024;
  constexpr int BSTAGE = BN * 128;
  constexpr int B_INST = BN / 8;
  constexpr int B_PER_WAVE = B_INST / NW;
  constexpr int PIPE_BYTES = 2 * PATCH_BYTES + 2 * BSTAGE;
  constexpr int EPI_LD = WN + 4;
  constexpr int EPI_BYTES = NW * WM * EPI_LD * 4;
  constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  constexpr bool PRE_MFMA = PATCH_BYTES >= 16 * 1024;
  static_assert(BM == 128 && (TW == 16 || TW == 8) && (BN == 128 || BN == 64) && B_INST % NW == 0 && B_PER_WAVE <= 4 && NTAPS >= 2 &&
                    PPW <= 2 * (NTAPS - 1) && LDS_BYTES <= 80 * 1024, "tile");

  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  char* const patch0 = lds;
  char* const bst0 = lds + 2 * PATCH_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- XCD-aware block order (as conv_halo_kernel): each XCD gets a contiguous run of tiles, couts fastest
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

  // ---- DMA roles (conv_halo_kernel): patch piece q = j * NW + wave covers patch rows q * 8 .. q * 8 + 7; lane -> (row q * 8 + lane / 8,
  // slot lane % 8), logical chunk slot ^ (row & 7); a wave's weight pieces are consecutive 8-row groups of the stage (one M0 value)
  const int rin = lane >> 3, slot = lane & 7;
  const int lca = slot ^ rin;
  // (the global pixel of a lane's patch row is recomputed per piece from an opaque copy of `rin` -- ~15 VALU under the MFMAs -- instead
  //  of living in PPW registers for the whole K loop: two fragment sets leave no room for them in the 3x3 128-cout tile)
  int wvoff[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    int row = n0 + (wave * B_PER_WAVE + j) * 8 + rin;
    if (row >= p.cout_pad) row = p.cout_pad - 1;          // clamped rows feed accumulators that are never stored
    const int lcj = slot ^ ((4 * ((wave * B_PER_WAVE + j) & 1) + (rin >> 1)) & 7);
    wvoff[j] = row * p.kchunks * 16 + lcj * 16 - j * 1024;      // (minus the instruction offset of piece j)
  }
  const int nrec = p.N * p.H * p.W;
  const int rb0 = p.src[0].cstride * 2, rb1 = p.src[1].cstride * 2, rb2 = p.src[2].cstride * 2, rb3 = p.src[3].cstride * 2;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[0].ptr + p.src[0].choff * 2), 0, nrec * rb0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[1].ptr + p.src[1].choff * 2), 0, nrec * rb1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[2].ptr + p.src[2].choff * 2), 0, nrec * rb2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[3].ptr + p.src[3].choff * 2), 0, nrec * rb3, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.weight), 0, p.cout_pad * p.kchunks * 16, 0x00020000);

#define V3P_ISSUE_PIECE(j, pbuf, e, lob)                                                                        \
  do {                                                                                                          \
    int rin_ = rin;                                                                                             \
    asm volatile("" : "+v"(rin_));                                                                              \
    const int r__ = ((j) * NW + wave) * 8 + rin_;                                                               \
    const int py_ = r__ / PW, px_ = r__ - py_ * PW;                                                             \
    const int iy_ = ty0 - p.ph + py_, ix_ = tx0 - p.pw + px_;                                                   \
    const bool ok_ = (r__ < P) & ((unsigned)iy_ < (unsigned)p.H) & ((unsigned)ix_ < (unsigned)p.W);             \
    const int ppix_ = ok_ ? (n * p.H + iy_) * p.W + ix_ : -1;                                                   \
    const int s_ = (e)[2] & 0xff;                                                                               \
    const __amdgpu_buffer_rsrc_t r_ = s_ == 1 ? rs1 : s_ == 2 ? rs2 : s_ == 3 ? rs3 : rs0;                      \
    const int rowbytes_ = s_ == 1 ? rb1 : s_ == 2 ? rb2 : s_ == 3 ? rb3 : rb0;                                  \
    const int coff_ = SPLIT ? (lca & 3) * 16 + ((lca & 4) ? (lob) : 0) : lca * 16;                              \
    const int voff_ = ppix_ >= 0 ? ppix_ * rowbytes_ + (e)[3] * 2 + coff_ : (int)0x80000000;                    \
    v3_dma16(r_, patch0 + (pbuf) * PATCH_BYTES + ((j) * NW + wave) * 1024, voff_, 0);                           \
  } while (0)
#define V3P_ISSUE_B(ks_, st_) V3WeightPieces<0, B_PER_WAVE>::issue(rw, bst0 + (st_) * BSTAGE + wave * B_PER_WAVE * 1024, wvoff, (ks_) * 128)

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, l4 = lane >> 4;
  int pp0[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = wm * WM + t * 16 + l15;
    pp0[t] = (m / TW) * PW + (m % TW);
  }
  const int b_off = (wn * WN + l15) * 128;
  const int bswz = (l15 >> 1) & 7;
  // the two fragment register sets: [set][K half (SPLIT: plane)][fragment]
  f16x8 af[2][2][TM], bf[2][2][TN];

// fragments of one tap step: weight stage st_, patch buffer pb_, tap shift sh_ (patch rows) -> set q_
#define V3P_READ(q_, st_, pb_, sh_)                                                                             \
  do {                                                                                                          \
    const char* sb_ = bst0 + (st_) * BSTAGE;                                                                    \
    const char* pc_ = patch0 + (pb_) * PATCH_BYTES;                                                             \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_) {                                                       \
      _Pragma("unroll") for (int f_ = 0; f_ < TN; ++f_)                                                         \
        bf[q_][kk_][f_] = *reinterpret_cast<const f16x8*>(sb_ + b_off + f_ * 16 * 128 + (((kk_ * 4 + l4) ^ bswz) << 4)); \
      _Pragma("unroll") for (int f_ = 0; f_ < TM; ++f_) {                                                       \
        /* (an opaque copy of the fragment's base row: the addresses are recomputed per step -- ~6 VALU per fragment under the MFMAs --  */ \
        /*  instead of being hoisted out of the channel loop: 8 x taps registers, which two fragment sets leave no room for)        */ \
        int pr_ = pp0[f_];                                                                                      \
        asm volatile("" : "+v"(pr_));                                                                           \
        const int row_ = pr_ + (sh_);                                                                           \
        af[q_][kk_][f_] = *reinterpret_cast<const f16x8*>(pc_ + row_ * 128 + (((kk_ * 4 + l4) ^ (row_ & 7)) << 4)); \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)
// the MFMAs of one tap step from set q_ (SPLIT: W_hi x A_lo, W_lo x A_hi, W_hi x A_hi -- small terms first, as conv_halo_kernel)
#define V3P_MFMA(q_)                                                                                            \
  do {                                                                                                          \
    if constexpr (SPLIT) {                                                                                      \
      _Pragma("unroll") for (int a_ = 0; a_ < TN; ++a_)                                                         \
        _Pragma("unroll") for (int b_ = 0; b_ < TM; ++b_)                                                       \
          acc[a_][b_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[q_][0][a_], af[q_][1][b_], acc[a_][b_], 0, 0, 0); \
      _Pragma("unroll") for (int a_ = 0; a_ < TN; ++a_)                                                         \
        _Pragma("unroll") for (int b_ = 0; b_ < TM; ++b_)                                                       \
          acc[a_][b_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[q_][1][a_], af[q_][0][b_], acc[a_][b_], 0, 0, 0); \
      _Pragma("unroll") for (int a_ = 0; a_ < TN; ++a_)                                                         \
        _Pragma("unroll") for (int b_ = 0; b_ < TM; ++b_)                                                       \
          acc[a_][b_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[q_][0][a_], af[q_][0][b_], acc[a_][b_], 0, 0, 0); \
    } else {                                                                                                    \
      _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                                       \
        _Pragma("unroll") for (int a_ = 0; a_ < TN; ++a_)                                                       \
          _Pragma("unroll") for (int b_ = 0; b_ < TM; ++b_)                                                     \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[q_][kk_][a_], af[q_][kk_][b_], acc[a_][b_], 0, 0, 0); \
    }                                                                                                           \
  } while (0)

  const int nblocks = p.kchunks / (8 * NTAPS);
  const int nk = nblocks * NTAPS;
  // ---- prologue: patch of block 0 + weights of step 0; then weights of step 1 and the fragments of step 0 (set 0)
  {
    i32x4s e, el;
    v3_fetch_entry(p.ktable, e);
    if constexpr (SPLIT) v3_fetch_entry(p.ktable + 4, el);
    v3_entry_ready(e);
    if constexpr (SPLIT) v3_entry_ready(el);
    const int lob0 = SPLIT ? (el[3] - e[3]) * 2 : 0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) V3P_ISSUE_PIECE(j, 0, e, lob0);
    V3P_ISSUE_B(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) V3P_ISSUE_B(1, 1);
    V3P_READ(0, 0, 0, 0);
  }

// one channel block: Q0_ = register set of its first tap step (compile time); the weight stage of step k is k & 1 = set parity
#define V3P_BLOCK(blk_, Q0_)                                                                                    \
  do {                                                                                                          \
    const bool have_next_ = (blk_) + 1 < nblocks;                                                               \
    i32x4s en_, enl_;                                                                                           \
    if (have_next_) {                                                                                           \
      v3_fetch_entry(p.ktable + ((blk_) + 1) * (NTAPS * 8), en_);                                               \
      if constexpr (SPLIT) v3_fetch_entry(p.ktable + ((blk_) + 1) * (NTAPS * 8) + 4, enl_);                     \
    }                                                                                                           \
    int lobn_ = 0;                                                                                              \
    const int pcb_ = (blk_) & 1, pnb_ = pcb_ ^ 1;                                                               \
    const int ks0_ = (blk_) * NTAPS;                                                                            \
    _Pragma("unroll") for (int t_ = 0; t_ < NTAPS; ++t_) {                                                      \
      constexpr int qbase_ = (Q0_);                                                                             \
      const int q_ = (qbase_ + t_) & 1;               /* compile time after unrolling */                        \
      /* (the builtin, not inline asm: the compiler's own waitcnt bookkeeping must see that the fragment reads of this step are */ \
      /*  complete, or it waits for them again -- and for the first of the NEXT step's reads -- in front of the MFMAs)           */ \
      __builtin_amdgcn_s_waitcnt(0x0070);             /* vmcnt(0) lgkmcnt(0) */                                  \
      asm volatile("" ::: "memory");                                                                            \
      __builtin_amdgcn_s_barrier();                                                                             \
      if (t_ == 0 && have_next_) {                                                                              \
        v3_entry_ready(en_);                                                                                    \
        if constexpr (SPLIT) { v3_entry_ready(enl_); lobn_ = (enl_[3] - en_[3]) * 2; }                          \
      }                                                                                                         \
      if (ks0_ + t_ + 2 < nk) V3P_ISSUE_B(ks0_ + t_ + 2, q_);                                                   \
      if (have_next_ && t_ < NTAPS - 1) {                                                                       \
        _Pragma("unroll") for (int jj_ = t_; jj_ < PPW; jj_ += NTAPS - 1) V3P_ISSUE_PIECE(jj_, pnb_, en_, lobn_); \
      }                                                                                                         \
      if (t_ + 1 < NTAPS) {                                                                                     \
        V3P_READ(q_ ^ 1, q_ ^ 1, pcb_, ((t_ + 1) / KW) * PW + ((t_ + 1) % KW));                                 \
      } else if (have_next_) {                                                                                  \
        V3P_READ(q_ ^ 1, q_ ^ 1, pnb_, 0);                                                                      \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      V3P_MFMA(q_);                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
  } while (0)

  {
    int blk = 0;
    for (; blk + 1 < nblocks; blk += 2) {
      V3P_BLOCK(blk, 0);
      V3P_BLOCK(blk + 1, NTAPS & 1);
    }
    if (blk < nblocks) V3P_BLOCK(blk, 0);
  }

  // ---- pre-activation addend through the matrix cores (conv_halo_kernel, PRE_MFMA): the addend tile is LDS-DMA'd into the free patch
  // buffers and multiplied by an identity fragment into the accumulators (exact)
  bool preadd_in_acc = false;
  if constexpr (PRE_MFMA) {
    if (p.preadd != nullptr && p.out_scale == 1.f && p.cout_g % BN == 0 && (long long)nrec * p.preadd_cstride * 2 < (1ll << 31)) {
      const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.preadd + p.preadd_choff * 2), 0,
                                                                             nrec * p.preadd_cstride * 2, 0x00020000);
      const int lcp = slot ^ rin;
      const int pj = BN == 128 ? wn : 0, c0 = BN == 128 ? 0 : wn * WN;
      const char* pp = patch0 + pj * PATCH_BYTES;
#pragma unroll
      for (int pl = 0; pl < (SPLIT ? 2 : 1); ++pl) {
        __syncthreads();
        const int plane_off = pl * p.preadd_lo * 2;
#pragma unroll
        for (int q = 0; q < 16 / NW; ++q) {
          const int piece = q * NW + wave;
          const int m = piece * 8 + rin;
          const int iy = ty0 + m / TW, ix = tx0 + m % TW;
          const bool ok = (iy < p.H) & (ix < p.W);
          const int base = ok ? ((n * p.H + iy) * p.W + ix) * (p.preadd_cstride * 2) + n0 * 2 + lcp * 16 : (int)0x80000000;
          v3_dma16(rsp, patch0 + piece * 1024, base, plane_off);
          if constexpr (BN == 128) v3_dma16(rsp, patch0 + PATCH_BYTES + piece * 1024, base, plane_off + 128);
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
            const int row = wm * WM + b * 16 + l15;
            const f16x8 afr = *reinterpret_cast<const f16x8*>(pp + row * 128 + (((kk * 4 + l4) ^ (row & 7)) << 4));
            acc[f][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(idf, afr, acc[f][b], 0, 0, 0);
          }
        }
      }
      preadd_in_acc = true;
    }
  }
  __syncthreads();                                    // LDS becomes the epilogue tile

  struct RowMap {
    int wm_base, ty0, tx0, H, W; long long nbase;
    __device__ __forceinline__ long long operator()(int prow) const {
      const int mt = wm_base + prow;
      const int iy = ty0 + mt / TW, ix = tx0 + mt % TW;
      return (iy < H && ix < W) ? (nbase + iy) * W + ix : -1ll;
    }
  };
  const RowMap rowmap{wm * WM, ty0, tx0, p.H, p.W, (long long)n * p.H};
  conv_epilogue<WM, WN, WN / 16, 0, true, true, SPLIT>(p, acc, lds + wave * (EPI_BYTES / NW), lane, n0 + wn * WN, 0, p.out, rowmap, nullptr,
                                                       nullptr, preadd_in_acc);
#undef V3P_ISSUE_PIECE
#undef V3P_ISSUE_B
#undef V3P_READ
#undef V3P_MFMA
#undef V3P_BLOCK
#endif
}

template <int TH, int TW, int KH, int KW, int BN, bool SPLIT>
static int launch_v3p(ConvParams p, hipStream_t stream) {
  p.tiles_n = (p.cout_g + BN - 1) / BN;
  const long long tiles = (long long)p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
  const long long nblk = tiles * p.tiles_n;
  if (nblk >= (1ll << 31)) return -1000;
  hipLaunchKernelGGL((conv_halo_pipe_kernel<TH, TW, KH, KW, BN, SPLIT>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  return launch_status("pp_conv2d(v3p)");
}

}  // namespace pp
