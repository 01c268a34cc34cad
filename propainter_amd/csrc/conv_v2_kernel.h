// LDS-DMA implicit-GEMM convolution kernel (design notes: conv_gemm_v2.hip); shared by the fp16 translation unit
// (conv_gemm_v2.hip) and the split-plane "f16x3" one (conv_gemm_v2s.hip).
#pragma once
#include "conv_epilogue.h"

namespace pp {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BK> __device__ __forceinline__ int swz_of_row(int row) {
  if constexpr (BK == 64) return (row >> 1) & 7;
  else {
    const int q = (row >> 2) & 3;        // G = {0,3,2,1}
    return (4 - q) & 3;
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// SPLIT: split-plane ("f16x3") epilogue operands (pp_conv_args_t.split; see conv_epilogue.h).
// TRI (with SPLIT, BK 64): TRI-PRODUCT K step (pp_conv_args_t.split == 2, conv.tri_ktable): the 8 chunks of a step are 4 chunks of the hi
// plane and the same 4 channel chunks of the lo plane (the LDS row of a pixel = [32 ch hi | 32 ch lo], of a cout = [W_hi | W_lo]) and
// the step issues W_hi x A_hi + W_hi x A_lo + W_lo x A_hi: 48 MFMAs per 16 fragment reads and per stage instead of 32 (see conv_halo.h).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int S, bool UNI, int SCHED = 0, bool SPLIT = false, bool TRI = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N <= 4 && BM * BN <= 128 * 128) ? 2 : 1) void conv_gemm_v2_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses device-only types (__amdgpu_buffer_rsrc_t): the host pass only needs the stub
  typedef _Float16 T;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int NT = 64 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int CH = BK / 8;             // 16-byte chunks per row
  constexpr int RPI = 64 / CH;           // rows written by one LDS-DMA instruction
  constexpr int ROWB = BK * 2;           // bytes per LDS row
  constexpr int A_INST = BM / RPI;       // DMA instructions per stage (whole block)
  constexpr int B_INST = BN / RPI;
  constexpr int A_PER_WAVE = A_INST / NW;
  constexpr int B_PER_WAVE = (B_INST + NW - 1) / NW;
  constexpr bool B_RAGGED = (B_INST % NW) != 0;      // some waves have no weight rows to fetch: they DMA into a dummy pad
  constexpr int G = A_PER_WAVE + B_PER_WAVE;         // DMA instructions per lane per step (identical for every wave)
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int EPI_WN = WN > 64 ? 64 : WN;          // wave tiles wider than 64 couts are drained in 64-cout halves (as conv_halo.h)
  constexpr int EPI_LD = EPI_WN + 4;                 // fp32 row stride of the epilogue staging tile
  constexpr int EPI_BYTES = NW * WM * EPI_LD * 4;
  constexpr int PIPE_BYTES = S * STAGE + (B_RAGGED ? NW * 1024 : 0);
  constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  static_assert(BM % RPI == 0 && BN % RPI == 0 && TM >= 1 && TN >= 1 && (NW % 2) == 0, "tile");
  static_assert(A_INST % NW == 0, "A rows must split evenly over the waves");
  static_assert(S >= 2 && (S - 2) * G < 64, "pipeline depth");
  static_assert(WN % 8 == 0 || WN == 16, "epilogue");
  static_assert(WN <= 64 || WN == 128, "wave tiles wider than 64 couts: two halves of 64");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");

  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- XCD-aware block order: give each XCD (= linear block id mod 8) a contiguous run of tiles, couts
  // fastest, so tiles that share pixels (all N-tiles of an M-tile, vertically adjacent M-tiles) hit one L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tn = bid % p.tiles_n;
  const int tmg = bid / p.tiles_n;
  const int tmi = tmg % p.tiles_m;
  const int g = tmg / p.tiles_m;
  const long long m0 = (long long)tmi * BM;   // < 2^31
  const int n0 = tn * BN;

  // ---- DMA role of this lane: slot (lane % CH) of row (lane / CH) of each of its wave's instructions;
  // the swizzle term is the same for all of them, so the lane always fetches one logical chunk `lc`.
  const int slot = lane % CH;
  const int rin = lane / CH;
  const int lc = slot ^ swz_of_row<BK>(wave * RPI + rin);     // (q*RPI + rin) has the same swizzle for q = j*NW + wave
  int a_iy0[A_PER_WAVE], a_ix0[A_PER_WAVE];
  int a_pix[A_PER_WAVE];                                        // n*H*W (pixels; N*H*W < 2^31 is checked by pp_conv2d)
  const unsigned Mu = (unsigned)p.M, OWu = (unsigned)p.OW, OHu = (unsigned)p.OH;
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    unsigned m = (unsigned)m0 + (j * NW + wave) * RPI + rin;
    if (m >= Mu) m = Mu - 1;                                    // rows past M are computed but never stored
    const unsigned ox = m % OWu, r = m / OWu;
    const unsigned oy = r % OHu, n = r / OHu;
    a_iy0[j] = (int)oy * p.sh - p.ph;
    a_ix0[j] = (int)ox * p.sw - p.pw;
    a_pix[j] = (int)(n * (unsigned)(p.H * p.W));
  }
  const char* wrow[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    int row = n0 + (j * NW + wave) * RPI + rin;
    if (row >= p.cout_pad) row = p.cout_pad - 1;                // clamped rows feed accumulators that are never stored
    wrow[j] = p.weight + ((long long)g * p.weight_gstride + (long long)row * p.kchunks * 8) * 2 + lc * 16;
  }
  const char* zero16 = reinterpret_cast<const char*>(p.ktable + p.kchunks);
  // per-source base pointers (wave-uniform) with the group / batch offsets folded in
  const char* sbase[PP_CONV_MAX_SRC];
  int srowb[PP_CONV_MAX_SRC];
#pragma unroll
  for (int i = 0; i < PP_CONV_MAX_SRC; ++i) {
    sbase[i] = p.src[i].ptr + ((long long)g * p.src_gstride + p.src[i].choff + g * p.src[i].cgroup) * 2;
    srowb[i] = p.src[i].cstride * 2;
  }
  const bool replicate = p.pad_mode == 1;

  // The CH table entries of a step are wave-uniform: they are fetched with *scalar* loads (SMEM, tracked by lgkmcnt)
  // issued by hand -- hipcc turns a uniform `ktable[...]` read inside the loop into a vector global_load, whose use
  // drains vmcnt to 0 and with it the DMA pipeline -- and every lane then selects the entry of its logical chunk.
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  struct Entries { i32x4 v[CH]; };
  auto fetch_entries = [&](int ks, Entries& E) {       // asynchronous: complete only after entries_ready()
    const int4* ptr = p.ktable + ks * CH;
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(E.v[0]) : "s"(ptr));
    asm volatile("s_load_dwordx4 %0, %1, 0x10" : "=s"(E.v[1]) : "s"(ptr));
    asm volatile("s_load_dwordx4 %0, %1, 0x20" : "=s"(E.v[2]) : "s"(ptr));
    asm volatile("s_load_dwordx4 %0, %1, 0x30" : "=s"(E.v[3]) : "s"(ptr));
    if constexpr (CH == 8) {
      asm volatile("s_load_dwordx4 %0, %1, 0x40" : "=s"(E.v[4]) : "s"(ptr));
      asm volatile("s_load_dwordx4 %0, %1, 0x50" : "=s"(E.v[5]) : "s"(ptr));
      asm volatile("s_load_dwordx4 %0, %1, 0x60" : "=s"(E.v[6]) : "s"(ptr));
      asm volatile("s_load_dwordx4 %0, %1, 0x70" : "=s"(E.v[7]) : "s"(ptr));
    }
  };
  auto entries_ready = [&](Entries& E) {
    if constexpr (CH == 8)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(E.v[0]), "+s"(E.v[1]), "+s"(E.v[2]), "+s"(E.v[3]), "+s"(E.v[4]), "+s"(E.v[5]),
                   "+s"(E.v[6]), "+s"(E.v[7])::"memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(E.v[0]), "+s"(E.v[1]), "+s"(E.v[2]), "+s"(E.v[3])::"memory");
  };
  auto issue = [&](int ks, int buf, const Entries& E) {
    // ---- A: gathered pixels (branch-free: invalid taps / padding chunks read the zero page)
    int4 e = make_int4(E.v[0][0], E.v[0][1], E.v[0][2], E.v[0][3]);
#pragma unroll
    for (int i = 1; i < CH; ++i) {
      const bool m = lc == i;
      e.x = m ? E.v[i][0] : e.x; e.y = m ? E.v[i][1] : e.y; e.z = m ? E.v[i][2] : e.z; e.w = m ? E.v[i][3] : e.w;
    }
    const int s = e.z & 0xff;
    const char* sp = s == 0 ? sbase[0] : s == 1 ? sbase[1] : s == 2 ? sbase[2] : sbase[3];
    const int rowbytes = s == 0 ? srowb[0] : s == 1 ? srowb[1] : s == 2 ? srowb[2] : srowb[3];
    sp += e.w * 2;
    const bool live = s != 255;
    char* abase = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      int iy = a_iy0[j] + e.x, ix = a_ix0[j] + e.y;
      const bool inside = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      iy = min(max(iy, 0), p.H - 1);
      ix = min(max(ix, 0), p.W - 1);
      const bool ok = live & (inside | replicate);
      const char* cand = sp + (long long)(a_pix[j] + iy * p.W + ix) * rowbytes;
      const char* src = ok ? cand : zero16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abase + (j * NW + wave) * 1024), 16, 0, 0);
    }
    // ---- B: packed weights (dense rows)
    char* bbase = lds + buf * STAGE + BM * ROWB;
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      if (!B_RAGGED || (j * NW + wave) < B_INST) {
        __builtin_amdgcn_global_load_lds((gptr_t)(wrow[j] + (long long)ks * BK * 2), (lptr_t)(bbase + (j * NW + wave) * 1024), 16, 0, 0);
      } else {   // keep the per-wave DMA count uniform (the counted vmcnt relies on it): fetch zeros into a private pad
        __builtin_amdgcn_global_load_lds((gptr_t)zero16, (lptr_t)(lds + S * STAGE + wave * 1024), 16, 0, 0);
      }
    }
  };

  // ---- UNI (uniform-step) fast path: the host guarantees that the CH chunks of every K step belong to one
  // (tap, source) with consecutive channel offsets (all sources are multiples of BK channels), so ONE scalar table
  // entry describes the step, the source select is scalar, and the gather uses buffer_load ... lds with a 32-bit
  // per-lane offset: out-of-image taps get an offset past num_records and the hardware range check returns zeros
  // (no zero page, no 64-bit address math, no divergent control flow).
  int u_rowpix[A_PER_WAVE];
  int u_wvoff[B_PER_WAVE];
  __amdgpu_buffer_rsrc_t u_rs[PP_CONV_MAX_SRC];
  __amdgpu_buffer_rsrc_t u_rw;
  if constexpr (UNI) {
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) u_rowpix[j] = a_pix[j] + a_iy0[j] * p.W + a_ix0[j];
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      int row = n0 + (j * NW + wave) * RPI + rin;
      if (row >= p.cout_pad) row = p.cout_pad - 1;
      u_wvoff[j] = row * p.kchunks * 16 + lc * 16;
    }
    const int nrec = (int)min((long long)p.N * p.H * p.W, (long long)0x7fffffff);   // pixels; bytes = nrec * rowbytes < 2^31 (checked on the host)
#pragma unroll
    for (int i = 0; i < PP_CONV_MAX_SRC; ++i)
      u_rs[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sbase[i]), 0, nrec * srowb[i], 0x00020000);
    u_rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.weight + (long long)g * p.weight_gstride * 2), 0,
                                             p.cout_pad * p.kchunks * 16, 0x00020000);
  }
  typedef int i32x4u __attribute__((ext_vector_type(4)));
  auto fetch_entry = [&](int ks, i32x4u& e) {            // asynchronous scalar load; complete after entry_ready()
    const int4* ptr = p.ktable + ks * CH;
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(e) : "s"(ptr));
  };
  auto entry_ready = [&](i32x4u& e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e)::"memory"); };
  auto issue_uni = [&](int ks, int buf, const i32x4u e) {
    const int dy = e[0], dx = e[1], s = e[2] & 0xff;
    const __amdgpu_buffer_rsrc_t rs = s == 1 ? u_rs[1] : s == 2 ? u_rs[2] : s == 3 ? u_rs[3] : u_rs[0];
    const int rowbytes = s == 1 ? srowb[1] : s == 2 ? srowb[2] : s == 3 ? srowb[3] : srowb[0];
    const bool live = s != 255;
    const int tapoff = dy * p.W + dx;
    // (TRI: logical chunks 0..3 = 32 channels of the hi plane, 4..7 = the same channels of the source's lo plane)
    const int lob = TRI ? (s == 1 ? p.src[1].lo : s == 2 ? p.src[2].lo : s == 3 ? p.src[3].lo : p.src[0].lo) * 2 : 0;
    const int coff = TRI ? e[3] * 2 + (lc & 3) * 16 + ((lc & 4) ? lob : 0) : e[3] * 2 + lc * 16;
    char* abase = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W) & live;
      const int voff = ok ? (u_rowpix[j] + tapoff) * rowbytes + coff : (int)0x80000000;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(abase + (j * NW + wave) * 1024), 16, voff, 0, 0, 0);
    }
    char* bbase = lds + buf * STAGE + BM * ROWB;
    const int wso = ks * BK * 2;
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      if (!B_RAGGED || (j * NW + wave) < B_INST)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rw, (lptr_t)(bbase + (j * NW + wave) * 1024), 16, u_wvoff[j], wso, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rw, (lptr_t)(lds + S * STAGE + wave * 1024), 16, (int)0x80000000, 0, 0, 0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = base + t*16 + (lane&15); slot = (kk*4 + (lane>>4)) ^ swz(row)
  const int frow = lane & 15;
  const int fswz = swz_of_row<BK>(frow);
  const int a_off = (wm * WM + frow) * ROWB;
  const int b_off = (BM + wn * WN + frow) * ROWB;

  const int nk = p.kchunks / CH;
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) {
      if constexpr (UNI) {
        i32x4u e;
        fetch_entry(s, e);
        entry_ready(e);
        issue_uni(s, s, e);
      } else {
        Entries E;
        fetch_entries(s, E);
        entries_ready(E);
        issue(s, s, E);
      }
    }
  int buf = 0, nbuf = S - 1;                          // stage consumed at step ks / stage filled for step ks+S-1
  for (int ks = 0; ks < nk; ++ks) {
    // retire the DMA of stage `buf` only: the up-to S-2 younger steps stay in flight across the barrier
    const int ahead = min(S - 2, nk - 1 - ks);
    const bool more = ks + S - 1 < nk;
    Entries E;
    i32x4u e1;
    if (more) {                                        // SMEM latency overlaps the DMA wait + barrier below
      if constexpr (UNI) fetch_entry(ks + S - 1, e1);
      else fetch_entries(ks + S - 1, E);
    }
    if (S >= 4 && ahead == 2) wait_vmcnt<(S >= 4 ? 2 : 0) * G>();
    else if (S >= 3 && ahead == 1) wait_vmcnt<(S >= 3 ? 1 : 0) * G>();
    else if (S >= 5 && ahead == 3) wait_vmcnt<(S >= 5 ? 3 : 0) * G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                     // stage `buf` complete for all waves; stage `nbuf` (read at step ks-1) is free
    const char* sb = lds + buf * STAGE;
    if constexpr (SCHED == 2) {                     // [diagnostic] DMA + barriers only
      if (more) {
        if constexpr (UNI) { entry_ready(e1); issue_uni(ks + S - 1, nbuf, e1); }
        else { entries_ready(E); issue(ks + S - 1, nbuf, E); }
      }
    } else if constexpr (SCHED == 3) {              // [diagnostic] LDS reads + MFMA only (stale tiles)
      if (more) { if constexpr (UNI) entry_ready(e1); else entries_ready(E); }
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        const int so = ((kk * 4 + (lane >> 4)) ^ fswz) * 16;
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const f16x8*>(sb + a_off + t * 16 * ROWB + so);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[t] = *reinterpret_cast<const f16x8*>(sb + b_off + t * 16 * ROWB + so);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[a], af[b], acc[a][b], 0, 0, 0);
      }
    } else if constexpr (SCHED == 0 && TRI) {
      static_assert(!TRI || (SPLIT && BK == 64), "tri-product step: split-plane layers, 64-wide K steps");
      if (more) {
        if constexpr (UNI) {
          entry_ready(e1);
          issue_uni(ks + S - 1, nbuf, e1);
        } else {
          entries_ready(E);
          issue(ks + S - 1, nbuf, E);
        }
      }
      f16x8 af[2][TM], bf[2][TN];          // kk 0 = hi plane / W_hi, kk 1 = lo plane / W_lo
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int so = ((kk * 4 + (lane >> 4)) ^ fswz) * 16;
#pragma unroll
        for (int t = 0; t < TM; ++t) af[kk][t] = *reinterpret_cast<const f16x8*>(sb + a_off + t * 16 * ROWB + so);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[kk][t] = *reinterpret_cast<const f16x8*>(sb + b_off + t * 16 * ROWB + so);
      }
      PP_MFMA_PRIO_BEGIN();
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
      PP_MFMA_PRIO_END();
    } else if constexpr (SCHED == 0) {
      if (more) {
        if constexpr (UNI) {
          entry_ready(e1);
          issue_uni(ks + S - 1, nbuf, e1);
        } else {
          entries_ready(E);
          issue(ks + S - 1, nbuf, E);
        }
      }
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        const int so = ((kk * 4 + (lane >> 4)) ^ fswz) * 16;
        f16x8 af[TM], bf[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const f16x8*>(sb + a_off + t * 16 * ROWB + so);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[t] = *reinterpret_cast<const f16x8*>(sb + b_off + t * 16 * ROWB + so);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[a], af[b], acc[a][b], 0, 0, 0);
      }
    } else {
      // SCHED 1: every fragment of the K step is requested from LDS first (one exposed LDS latency per step instead of
      // one per 4-MFMA group), the next stage's DMA is issued while those reads are in flight, then the MFMAs run
      // back to back.  sched_barrier fences keep hipcc from sinking the reads back next to their uses.
      constexpr int KK = BK / 32;
      f16x8 af[KK][TM], bf[KK][TN];
      if (more) {                                      // the table entry (SMEM, requested before the barrier) must be
        if constexpr (UNI) entry_ready(e1);            // retired BEFORE the LDS reads are queued: lgkmcnt counts both
        else entries_ready(E);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int so = ((kk * 4 + (lane >> 4)) ^ fswz) * 16;
#pragma unroll
        for (int t = 0; t < TM; ++t) af[kk][t] = *reinterpret_cast<const f16x8*>(sb + a_off + t * 16 * ROWB + so);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[kk][t] = *reinterpret_cast<const f16x8*>(sb + b_off + t * 16 * ROWB + so);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        if constexpr (UNI) issue_uni(ks + S - 1, nbuf, e1);
        else issue(ks + S - 1, nbuf, E);
      }
      __builtin_amdgcn_sched_barrier(0);
      PP_MFMA_PRIO_BEGIN();
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[kk][a], af[kk][b], acc[a][b], 0, 0, 0);
      PP_MFMA_PRIO_END();
    }
    buf = buf + 1 == S ? 0 : buf + 1;
    nbuf = nbuf + 1 == S ? 0 : nbuf + 1;
  }
  __syncthreads();                                    // every wave is done with the stages: LDS becomes the epilogue tile

  // ---- pre-activation addend / residual of a LINEAR layer through the matrix cores (as in conv_gemm_v3.hip: an operand the
  // epilogue reads per output row costs like an un-overlapped stream): the BM x 128 fp16 tile is LDS-DMA'd into the free stages and
  // multiplied by an identity fragment into the fp32 accumulators (exact); the epilogue then sees a plain layer.  64 x 64 wave tiles.
  bool addend_in_acc = false;
  int act_after = p.act;
  if constexpr (WM == 64 && WN == 64 && BN == 128 && BK == 64 && 2 * BM * 128 <= PIPE_BYTES) {
    const bool lin_res = p.residual != nullptr && p.preadd == nullptr && p.act == PP_ACT_NONE;      // out = act2(conv + bias + residual)
    const char* ad = p.preadd != nullptr ? p.preadd : p.residual;
    const int ad_cs = p.preadd != nullptr ? p.preadd_cstride : p.res_cstride;
    const int ad_co = p.preadd != nullptr ? p.preadd_choff : p.res_choff;
    if ((p.preadd != nullptr || lin_res) && p.groups == 1 && p.fuse == PP_FUSE_NONE && p.out_f16 && p.out_scale == 1.f &&
        p.cout_g % 128 == 0 && (ad_cs & 7) == 0 && (ad_co & 7) == 0 && ((unsigned long long)ad & 15) == 0 &&
        p.M * (long long)ad_cs * 2 < (1ll << 31)) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ad + ad_co * 2), 0, (int)(p.M * ad_cs * 2), 0x00020000);
      const int ad_lo = p.preadd != nullptr ? p.preadd_lo : p.res_lo;
      const int s8 = lane & 7, r8 = lane >> 3;         // one instruction = 8 rows x 128 B; logical chunk = slot ^ (row & 7)
      const int lcp = s8 ^ r8;
      const int l15 = lane & 15, l4 = lane >> 4;
      const char* pp = lds + wn * (BM * 128);          // the wave's 64 couts are the 64 channels of half wn
#pragma unroll
      for (int pl = 0; pl < (SPLIT ? 2 : 1); ++pl) {     // split-plane addend: hi plane, then lo plane (both exact)
        if (pl) __syncthreads();                         // the first plane's fragment reads are complete
        const int plane_off = pl * ad_lo * 2;
#pragma unroll
        for (int q = 0; q < BM / 8 / NW; ++q) {
          const int piece = q * NW + wave;
          const long long m = m0 + piece * 8 + r8;
          const int off = m < p.M ? (int)(m * ad_cs * 2) + n0 * 2 + lcp * 16 : (int)0x80000000;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + piece * 1024), 16, off, plane_off, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + BM * 128 + piece * 1024), 16, off, plane_off + 128, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int f = 0; f < TN; ++f) {
          const int kk = f >> 1, kpos = (f & 1) * 16;
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
      addend_in_acc = true;
      if (lin_res) act_after = p.act2;
      __syncthreads();                                // the operand tile has been consumed: LDS becomes the epilogue tile
    }
  }

  // ---- epilogue (conv_epilogue.h): wave-private staging tile; 8 consecutive couts (16 B fp16 / 32 B fp32) per lane
  struct RowMap {
    long long m_base, M;
    __device__ __forceinline__ long long operator()(int prow) const {
      const long long m = m_base + prow;
      return m < M ? m : -1ll;
    }
  };
  const RowMap rowmap{m0 + wm * WM, p.M};
  if constexpr (WN == 128) {
    // 128-cout wave tiles (256 x 256 block tile: half the LDS-DMA bytes per MFMA of the 256 x 128 tile): two passes of 64 couts through the
    // same wave-private staging tile (LDS operations of one wave execute in order: the second pass cannot overtake the first pass's reads)
    char* ob = p.out + (long long)g * p.out_gstride * (p.out_f16 ? 2 : 4);
    conv_epilogue<WM, 64, TN, 0, true, true, SPLIT>(p, acc, lds + wave * (WM * EPI_LD * 4), lane, n0 + wn * WN, g, ob, rowmap);
    conv_epilogue<WM, 64, TN, 4, true, true, SPLIT>(p, acc, lds + wave * (WM * EPI_LD * 4), lane, n0 + wn * WN + 64, g, ob, rowmap);
  } else if constexpr (WN >= 16 && WN % 8 == 0) {
    if (addend_in_acc) {
      ConvParams pe = p;
      pe.preadd = nullptr; pe.residual = nullptr; pe.act = act_after;
      if (act_after != p.act) { pe.act_param = 0.f; pe.act2 = PP_ACT_NONE; }      // (linear-residual conversion: act2 became the activation; a genuine preadd keeps its act2)
      conv_epilogue<WM, WN, WN / 16, 0, true, true, SPLIT>(pe, acc, lds + wave * (WM * EPI_LD * 4), lane, n0 + wn * WN, g,
                            p.out + (long long)g * p.out_gstride * (p.out_f16 ? 2 : 4), rowmap);
    } else {
      conv_epilogue<WM, WN, WN / 16, 0, true, true, SPLIT>(p, acc, lds + wave * (WM * EPI_LD * 4), lane, n0 + wn * WN, g,
                            p.out + (long long)g * p.out_gstride * (p.out_f16 ? 2 : 4), rowmap);
    }
  }
#endif
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int S, int SCHED = 0, bool SPLIT = false, bool TRI = false>
static int launch_v2(ConvParams p, bool uni, hipStream_t stream) {
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (p.cout_g + BN - 1) / BN;
  const long long nblk = (long long)p.tiles_m * p.tiles_n * p.groups;
  const dim3 grid((unsigned)nblk), block(64 * WAVES_M * WAVES_N);
  // the uniform-step fast path needs uniform steps at this BK (flag bit = chunks per step) -- see conv_v2_dispatch
  if (uni && (p.ktable_uniform & (BK / 8)))
    hipLaunchKernelGGL((conv_gemm_v2_kernel<BM, BN, BK, WAVES_M, WAVES_N, S, true, SCHED, SPLIT, TRI>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((conv_gemm_v2_kernel<BM, BN, BK, WAVES_M, WAVES_N, S, false, SCHED, SPLIT, TRI>), grid, block, 0, stream, p);
  return launch_status("pp_conv2d(v2)");
}

}  // namespace pp
