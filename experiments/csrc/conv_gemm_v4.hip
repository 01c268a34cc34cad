// fp16 implicit-GEMM convolution, WIDE halo tiles: 256 output pixels x 128 couts per block, FOUR waves of 128 px x 64 couts,
// 32-channel K steps.  Same GEMM view, K-chunk table, packed weights and epilogue as conv_gemm_v3.hip; same K terms.
//
// Why (profiles/r2_conv_epilogue_ab.txt): the 128 x 128 / 64 x 64-wave-tile halo kernel is bound by the chip's power budget
// (it runs at ~1.67 GHz; the same instruction stream on zero operands is 35 % faster) and its time splits into MFMA 39 %,
// fragment ds_reads 22 %, LDS-DMA 20 %, epilogue / sync 20 % -- whatever is removed, the time drops by its share.  So the
// lever left is fewer bytes moved per MFMA:
//   wave tile 128 x 64: 12 fragment reads per 32 MFMAs (0.375 KB / MFMA) instead of 8 per 16 (0.5 KB);
//   block tile 256 x 128: the weight tile is fetched once per 256 pixels (half the weight DMA per MFMA), and a 16 x 16 pixel
//   tile has less halo (324 patch rows per 256 px vs 180 per 128 px for 3x3).
// A 256 x 128 tile with 64-channel K steps needs 120 KB of LDS (one block per CU, ONE wave per SIMD: measured 13-30 % slower
// than the 128 x 128 kernel, tools/kbench impl 106).  With 32-channel steps the double-buffered patch is 2 x 24 KB and a weight
// stage 8 KB: 72 KB per block, two blocks (8 waves) per CU as before.
//
//   K order: for each 64-channel block of the table, for each 32-channel half, for each tap.  (The v2 / v3 kernels walk
//   [block][tap][64 channels]: same terms, a different fp32 summation order -> results agree to rounding, not bit for bit.)
//   LDS: 2 patch buffers [PROWS][32 ch] (64-byte rows) + 3 weight stages [128 couts][32 ch] + epilogue tile (aliased).
//   Swizzle (both images): 16-byte slot ^ ((row >> 1) & 2) -- conflict-free for 16 consecutive rows at ANY start under
//   ds_read_b128's lane groups (tools/lds_swizzle_check.py).
//   Weights are requested TWO steps ahead (3 stages), patch pieces one channel-half ahead; s_waitcnt vmcnt is counted.
//
// MEASURED (MI355X, tools/kbench impl 107 vs 70, interleaved rounds, profiles/r2_conv_epilogue_ab.txt): bit-level agreement to
// one fp16 ulp, and NO gain -- 1x5 K1280 cout256 737 vs 776 TFLOP/s, 5x1 cout128 801 vs 800, 3x3 K1152 cout128 758 vs 782,
// 3x3 K2304 cout256 854 vs 901.  A quarter fewer fragment bytes and half the weight DMA per MFMA buy nothing: consistent with
// the power-budget picture (saved cycles come back as a lower clock) rather than with an LDS- or L2-bound kernel.  The kernel
// is therefore compiled in diagnostic builds only (PP_DIAG=1, impl 107) and never dispatched automatically.
#include "conv_epilogue.h"

namespace pp {

#if !defined(PP_DIAG)
int conv_v4_dispatch(const ConvParams&, int, hipStream_t) { return -1000; }
}  // namespace pp
#else

typedef __attribute__((address_space(3))) void* lptr3w_t;

#if defined(__HIP_DEVICE_COMPILE__)
typedef int i32x4w __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void v4_fetch_entry(const int4* ptr, i32x4w& e) { asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(e) : "s"(ptr)); }
static __device__ __forceinline__ void v4_entry_ready(i32x4w& e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e)::"memory"); }
static __device__ __forceinline__ void v4_dma16(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr3w_t)dst, 16, voff, soff, 0, 0);
}
// at most `n` LDS-DMA / vector-memory operations of this wave still in flight (n is wave-uniform, 0..3)
static __device__ __forceinline__ void v4_wait_outstanding(int n) {
  if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

template <int TH, int TW, int KH, int KW>
__global__ __launch_bounds__(256, 2) void conv_halo_wide_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 T;
  constexpr int BM = TH * TW, BN = 128, NW = 4;
  constexpr int WM = 128, WN = 64, TMH = 4, TN = WN / 16;        // wave tile = two 64-row halves of TMH fragments
  constexpr int PH = TH + KH - 1, PW = TW + KW - 1, P = PH * PW;
  constexpr int NTAPS = KH * KW;
  constexpr int PIECES = ((P + 15) / 16 + NW - 1) / NW * NW;     // LDS-DMA instructions per patch (16 rows of 64 B each)
  constexpr int PPW = PIECES / NW;
  constexpr int PATCH_BYTES = PIECES * 1024;
  constexpr int BSTAGE = BN * 64, NST = 3;
  constexpr int PIPE_BYTES = 2 * PATCH_BYTES + NST * BSTAGE;
  constexpr int EPI_BYTES = NW * 64 * (64 + 4) * 4;               // one 64 x 64 fp32 staging tile per wave
  constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  static_assert(BM == 256 && TW == 16 && PPW <= NTAPS && LDS_BYTES <= 80 * 1024, "tile");

  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  char* const patch0 = lds;
  char* const bst0 = lds + 2 * PATCH_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware block order (as v2 / v3): each XCD gets a contiguous run of tiles, couts fastest
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

  // ---- DMA roles.  One instruction = 16 rows x 64 B: lane -> (row lane / 4, physical slot lane % 4); the logical 16-byte chunk
  // it fetches is slot ^ ((row >> 1) & 2), and pieces / weight groups start at multiples of 16 rows
  const int rin = lane >> 2;
  const int lc = (lane & 3) ^ ((lane >> 3) & 2);
  int ppix[PPW];                                          // global pixel index of the lane's patch row, -1 = zero fill
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int r = (j * NW + wave) * 16 + rin;
    const int py = r / PW, px = r - py * PW;
    const int iy = ty0 - p.ph + py, ix = tx0 - p.pw + px;
    const bool ok = (r < P) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
    ppix[j] = ok ? (n * p.H + iy) * p.W + ix : -1;
  }
  int wvoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = n0 + (j * NW + wave) * 16 + rin;
    if (row >= p.cout_pad) row = p.cout_pad - 1;          // clamped rows feed accumulators that are never stored
    wvoff[j] = row * p.kchunks * 16 + lc * 16;
  }
  const int nrec = p.N * p.H * p.W;
  const int rb0 = p.src[0].cstride * 2, rb1 = p.src[1].cstride * 2, rb2 = p.src[2].cstride * 2, rb3 = p.src[3].cstride * 2;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[0].ptr + p.src[0].choff * 2), 0, nrec * rb0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[1].ptr + p.src[1].choff * 2), 0, nrec * rb1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[2].ptr + p.src[2].choff * 2), 0, nrec * rb2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src[3].ptr + p.src[3].choff * 2), 0, nrec * rb3, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.weight), 0, p.cout_pad * p.kchunks * 16, 0x00020000);

  // patch piece j of 32-channel half `half_` of the table block described by entry e, into patch buffer pbuf
#define V4_ISSUE_PIECE(j, pbuf, e, half_)                                                                       \
  do {                                                                                                          \
    const int s_ = (e)[2] & 0xff;                                                                               \
    const __amdgpu_buffer_rsrc_t r_ = s_ == 1 ? rs1 : s_ == 2 ? rs2 : s_ == 3 ? rs3 : rs0;                      \
    const int rowbytes_ = s_ == 1 ? rb1 : s_ == 2 ? rb2 : s_ == 3 ? rb3 : rb0;                                  \
    const int voff_ = ppix[j] >= 0 ? ppix[j] * rowbytes_ + (e)[3] * 2 + (half_) * 64 + lc * 16 : (int)0x80000000; \
    v4_dma16(r_, patch0 + (pbuf) * PATCH_BYTES + ((j) * NW + wave) * 1024, voff_, 0);                           \
  } while (0)
  // weight tile of the step with weight-row byte offset koff_, into stage st_
#define V4_ISSUE_B(koff_, st_)                                                                                  \
  do {                                                                                                          \
    v4_dma16(rw, bst0 + (st_) * BSTAGE + wave * 1024, wvoff[0], (koff_));                                       \
    v4_dma16(rw, bst0 + (st_) * BSTAGE + (NW + wave) * 1024, wvoff[1], (koff_));                                \
  } while (0)

  f32x4 acc[2][TN][TMH];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TMH; ++b) acc[h][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment geometry: A row of fragment f (tile row wm*8 + f) = patch row pp_base + f * PW + tap shift; B row = wn*64 + t*16 + (lane & 15)
  const int l15 = lane & 15, l4 = lane >> 4;
  const int pp_base = wm * 8 * PW + l15;
  const int b_off = (wn * WN + l15) * 64 + ((l4 ^ ((l15 >> 1) & 2)) << 4);

  const int nblocks = p.kchunks / (8 * NTAPS);            // 64-channel table blocks
  const int nu = nblocks * 2;                             // patches (32-channel halves)
  const int nk = nu * NTAPS;                              // K steps of 32
  // weight-row byte offset of step (u, t): u = 2 * block + half
  auto koff = [&](int u, int t) { return ((u >> 1) * NTAPS + t) * 128 + (u & 1) * 64; };

  // ---- prologue: patch 0 (all pieces), weights of steps 0 and 1
  i32x4w ecur;
  v4_fetch_entry(p.ktable, ecur);
  v4_entry_ready(ecur);
#pragma unroll
  for (int j = 0; j < PPW; ++j) V4_ISSUE_PIECE(j, 0, ecur, 0);
  V4_ISSUE_B(koff(0, 0), 0);
  if (nk > 1) V4_ISSUE_B(NTAPS > 1 ? koff(0, 1) : koff(1, 0), 1);
  int ks = 0, st = 0;                                     // step index, its weight stage
  for (int u = 0; u < nu; ++u) {
    const bool have_next = u + 1 < nu;
    // table entry of the NEXT patch: the other half of this block, or the first half of the next block
    i32x4w en = ecur;
    if ((u & 1) && have_next) v4_fetch_entry(p.ktable + ((u + 1) >> 1) * (NTAPS * 8), en);
    const char* pcur = patch0 + (u & 1) * PATCH_BYTES;
    const int pnext = (u + 1) & 1;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const bool more1 = ks + 1 < nk, more2 = ks + 2 < nk;
      // operations this wave issued after the weights of THIS step: [piece of step ks-1] + weights of step ks+1
      // (at t == 0 everything but those weights has to be there: the patch of this half may have got its last piece in step ks-1)
      v4_wait_outstanding((more1 ? 2 : 0) + ((t > 0 && t - 1 < PPW && have_next) ? 1 : 0));
      __builtin_amdgcn_s_barrier();          // weights of step ks (and, at t == 0, the whole patch) are in LDS; stage (ks+2)%3 is free
      if (t == 0 && (u & 1) && have_next) v4_entry_ready(en);
      if (t < PPW && have_next) V4_ISSUE_PIECE(t, pnext, en, (u + 1) & 1);
      if (more2) {
        const int t2 = t + 2 < NTAPS ? t + 2 : t + 2 - NTAPS;         // (NTAPS >= 3 for every instantiation)
        const int u2 = t + 2 < NTAPS ? u : u + 1;
        const int st2 = st + 2 >= NST ? st + 2 - NST : st + 2;
        V4_ISSUE_B(koff(u2, t2), st2);
      }
      const int sh = (t / KW) * PW + (t % KW);          // compile-time after unrolling
      // (the fragment addresses are recomputed every step from an opaque copy of the base: they are invariant across the
      // channel loop, and hoisted out of it -- 8 x NTAPS address registers -- they push the 128 accumulators into scratch)
      int ppb = pp_base;
      asm volatile("" : "+v"(ppb));
      const char* sb = bst0 + st * BSTAGE;
      // B fragments once per step, A fragments one 64-row half at a time (register budget: 128 accumulators + 16 + 16 fragment
      // registers of the 256 a wave may hold at two waves per SIMD)
      f16x8 bf[TN];
#pragma unroll
      for (int f = 0; f < TN; ++f) bf[f] = *reinterpret_cast<const f16x8*>(sb + b_off + f * 16 * 64);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f16x8 af[TMH];
#pragma unroll
        for (int f = 0; f < TMH; ++f) {
          const int row = ppb + (h * TMH + f) * PW + sh;
          af[f] = *reinterpret_cast<const f16x8*>(pcur + row * 64 + ((l4 ^ ((row >> 1) & 2)) << 4));
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TMH; ++b)
            acc[h][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[a], af[b], acc[h][a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);      // keeps the second half's reads behind the first half's MFMAs (live-range cap)
      }
      ++ks;
      st = st + 1 == NST ? 0 : st + 1;
    }
    ecur = en;
  }
  __syncthreads();                                    // LDS becomes the epilogue tile

  // ---- epilogue (conv_epilogue.h), one 64-row half at a time through the wave-private staging tile (LDS operations of one
  // wave execute in order, so the second half cannot overtake the first half's reads)
  struct RowMap {
    int wm_base, ty0, tx0, H, W; long long nbase;
    __device__ __forceinline__ long long operator()(int prow) const {
      const int mt = wm_base + prow;
      const int iy = ty0 + mt / TW, ix = tx0 + mt % TW;
      return (iy < H && ix < W) ? (nbase + iy) * W + ix : -1ll;
    }
  };
  // (two explicit calls: inside a loop the compiler does not unroll the large inlined body, acc[h] becomes a dynamic index and the
  // whole accumulator array moves to scratch memory -- 13x slower, measured)
  {
    const RowMap rowmap{wm * WM, ty0, tx0, p.H, p.W, (long long)n * p.H};
    conv_epilogue<64, WN>(p, acc[0], lds + wave * (EPI_BYTES / NW), lane, n0 + wn * WN, 0, p.out, rowmap);
  }
  {
    const RowMap rowmap{wm * WM + 64, ty0, tx0, p.H, p.W, (long long)n * p.H};
    conv_epilogue<64, WN>(p, acc[1], lds + wave * (EPI_BYTES / NW), lane, n0 + wn * WN, 0, p.out, rowmap);
  }
#undef V4_ISSUE_PIECE
#undef V4_ISSUE_B
#endif
}

constexpr bool kAutoWide = false;     // the auto dispatch takes this kernel only once it has been measured against v3 layer by layer

template <int TH, int TW, int KH, int KW>
static int launch_v4(ConvParams p, hipStream_t stream) {
  p.tiles_n = (p.cout_g + 127) / 128;
  const long long tiles = (long long)p.N * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
  const long long nblk = tiles * p.tiles_n;
  if (nblk >= (1ll << 31)) return -1000;
  hipLaunchKernelGGL((conv_halo_wide_kernel<TH, TW, KH, KW>), dim3((unsigned)nblk), dim3(256), 0, stream, p);
  return launch_status("pp_conv2d(v4)");
}

// Returns -1000 when the shape is outside the wide halo-tile family (caller falls back to the v3 halo kernel).
// cfg: 0 = auto (large maps with >= 128 couts), 107 = force.
int conv_v4_dispatch(const ConvParams& p, int cfg, hipStream_t stream) {
  const int kh = p.tap_h, kw = p.tap_w;
  if (kh <= 0 || kw <= 0) return -1000;
  if (p.groups != 1 || p.sh != 1 || p.sw != 1 || p.pad_mode != 0 || p.OH != p.H || p.OW != p.W) return -1000;
  if (p.ph != (kh - 1) / 2 || p.pw != (kw - 1) / 2 || !(p.ktable_uniform & 8)) return -1000;
  if (p.kchunks % (8 * kh * kw) != 0 || p.src_gstride != 0 || p.out_gstride != 0) return -1000;
  if ((long long)p.cout_pad * p.kchunks * 16 >= (1ll << 31)) return -1000;
  for (int i = 0; i < p.nsrc; ++i)
    if ((long long)p.N * p.H * p.W * p.src[i].cstride * 2 >= (1ll << 31)) return -1000;
  if (cfg == 0) {
    if (!kAutoWide) return -1000;
    // auto: only where the 256-pixel tiles still fill the chip several times over and the cout tiles are full enough
    const long long tiles = (long long)p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16) * ((p.cout_g + 127) / 128);
    if (p.cout_g < 128 || (p.cout_g % 128 != 0 && p.cout_g % 128 <= 64) || tiles < 2048 || p.H < 16 || p.W < 16) return -1000;
  }
  if (kh == 3 && kw == 3) return launch_v4<16, 16, 3, 3>(p, stream);
  if (kh == 1 && kw == 5) return launch_v4<16, 16, 1, 5>(p, stream);
  if (kh == 5 && kw == 1) return launch_v4<16, 16, 5, 1>(p, stream);
  return -1000;
}

}  // namespace pp
#endif  // PP_DIAG
