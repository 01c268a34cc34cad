// Sparse window attention for gfx950 (model/modules/sparse_transformer.py:158-281).
//
// Key set of a *masked* window (any local frame has a hole inside it), per key frame f in T_ind:
//   45 own-window tokens | 148 rolled-neighbour tokens (circular, index table) | P pooled tokens
// -> one softmax over n_tind * (193 + P) keys for all T*45 queries of the window.  *Unmasked* windows
// do plain 45x45 self attention per frame.  The reference materialises the rolled / pooled K,V per
// window (1.6 GB each at 720p); here K/V rows are gathered on the fly from the q/k/v token grids through
// the index tables, flash-style (online softmax), and the masked/unmasked decision is read from a device
// flag so there is no host synchronisation (the reference's nonzero()).
//
// attn_mfma_kernel (fp16): 256 threads = 4 waves x 16 queries, 64-key tiles.
//   S^T = K Q^T   (v_mfma_f32_16x16x32_f16: A = K rows from LDS, B = Q fragments held in registers)
//   -> every lane owns 16 scores of ONE query, so the row max/sum need only two cross-lane shuffles;
//   O^T += V^T P^T  with P^T taken directly from the score accumulators (keys of a 32-key step are
//   relabelled so that no data movement is needed); V is staged ROW-major and its fragments come out of
//   ds_read_b64_tr_b16 (gfx950's transposing LDS read).
// attn_ref_kernel (any dtype): one wave per query, fp32 math; the parity path for fp32 and the
//   on-device cross-check of the MFMA kernel.
#include "common.h"

namespace pp {

struct AttnParams {
  int B, T, Hp, Wp, C, heads, wsz, n_rolled, P, n_tind, nW;
  const char* q; const char* k; const char* v;
  int qkv_cs;
  const char* pk; const char* pv;
  int pkv_cs;
  const int* own; const int* rolled; const int* tind;
  const float* wmask;
  char* out;
  const int* work;     // [1 + B*nW]: count of masked windows, then their flat (b*nW + w) ids (attn_compact_kernel)
  int out_h, out_w;    // > 0: `out` is the COMPACT [B, T, out_h, out_w, C] grid (padding tokens are not written); 0: padded [B, T, Hp, Wp, C]
};

// element offset (in tokens) of query token `idx` (= y * Wp + x in the padded grid) of frame (b, f) in `out`, or -1 for a padding
// token of a compact output
__device__ __forceinline__ long long attn_out_token(const AttnParams& p, int b, int f, int idx) {
  if (p.out_h <= 0) return ((long long)b * p.T + f) * p.Hp * p.Wp + idx;
  const int y = idx / p.Wp, x = idx - y * p.Wp;
  return (y < p.out_h && x < p.out_w) ? (((long long)b * p.T + f) * p.out_h + y) * p.out_w + x : -1ll;
}

constexpr int HD = 128;  // head dim

// pointer to the 128-wide head slice of key `r` (0..192 grid tokens, >= 193 pooled) of frame f
template <typename T>
__device__ __forceinline__ const T* key_row(const AttnParams& p, const T* grid, const T* pooled, int b, int f, int r,
                                            const int* idx_lds, int head) {
  const int ngrid = p.wsz + p.n_rolled;
  if (r < ngrid)
    return grid + (((long long)b * p.T + f) * p.Hp * p.Wp + idx_lds[r]) * p.qkv_cs + head * HD;
  return pooled + (((long long)b * p.T + f) * p.P + (r - ngrid)) * p.pkv_cs + head * HD;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_ref_kernel(const AttnParams p) {
  __shared__ int idx_lds[256];
  const int head = blockIdx.x % p.heads;
  const int w = (blockIdx.x / p.heads) % p.nW;
  const int b = blockIdx.x / (p.heads * p.nW);
  const int ngrid = p.wsz + p.n_rolled;
  for (int i = threadIdx.x; i < ngrid; i += 256) idx_lds[i] = i < p.wsz ? p.own[w * p.wsz + i] : p.rolled[w * p.n_rolled + i - p.wsz];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (qi >= p.T * p.wsz) return;
  const bool masked = p.wmask[b * p.nW + w] > 0.f;
  const int fq = qi / p.wsz, tok = qi % p.wsz;
  const T* qg = reinterpret_cast<const T*>(p.q);
  const T* kg = reinterpret_cast<const T*>(p.k);
  const T* vg = reinterpret_cast<const T*>(p.v);
  const T* pkg = reinterpret_cast<const T*>(p.pk);
  const T* pvg = reinterpret_cast<const T*>(p.pv);
  const long long qoff = (((long long)b * p.T + fq) * p.Hp * p.Wp + idx_lds[tok]);
  const T* qp = qg + qoff * p.qkv_cs + head * HD;
  const float q0 = to_f32(qp[lane]), q1 = to_f32(qp[lane + 64]);
  const float scale = rsqrtf((float)HD);
  const int kpf = masked ? ngrid + p.P : p.wsz;
  const int nkeys = masked ? p.n_tind * kpf : p.wsz;
  float m = -1e30f, l = 0.f, a0 = 0.f, a1 = 0.f;
  for (int kk = 0; kk < nkeys; ++kk) {
    const int f = masked ? p.tind[kk / kpf] : fq;
    const int r = kk % kpf;
    const T* kp = key_row<T>(p, kg, pkg, b, f, r, idx_lds, head);
    const T* vp = key_row<T>(p, vg, pvg, b, f, r, idx_lds, head);
    float s = q0 * to_f32(kp[lane]) + q1 * to_f32(kp[lane + 64]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s *= scale;
    const float mn = fmaxf(m, s);
    const float alpha = __expf(m - mn), pe = __expf(s - mn);
    l = l * alpha + pe;
    a0 = a0 * alpha + pe * to_f32(vp[lane]);
    a1 = a1 * alpha + pe * to_f32(vp[lane + 64]);
    m = mn;
  }
  const long long otok = attn_out_token(p, b, fq, idx_lds[tok]);
  if (otok < 0) return;
  T* op = reinterpret_cast<T*>(p.out) + otok * p.C + head * HD;
  op[lane] = from_f32<T>(a0 / l);
  op[lane + 64] = from_f32<T>(a1 / l);
}

constexpr int KT = 64;          // keys per tile
constexpr int ATTN_MAX_KEY_FRAMES = 256;     // key frames of a masked window per launch (the LDS table one block of 256 threads fills in one step;
                                             // round 5 had 64 -- a 432x240 clip of more than ~1 200 frames at ref_stride 10 needs more)
constexpr int KS_LD = HD;       // K tile row stride (elements): 256-byte rows, 16-byte slots XOR-swizzled by the key row
constexpr int VS_LD = HD + 4;   // V tile row stride (elements): 264-byte rows (8-byte skew per key row for the transposing reads)

#if defined(__HIP_DEVICE_COMPILE__)
typedef __fp16 fp16v4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16v4 lds_fp16v4;
// ds_read_b64_tr_b16: within each group of 16 lanes, lane i supplies the address of 4 consecutive fp16 (row i / 4, column
// chunk i % 4 of a [4][16] block) and receives column i of that block (4 values, one per row).
static __device__ __forceinline__ f16x4 lds_read_tr16(const _Float16* p) {
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16v4*)p));
}
#endif

// Flash-style MFMA kernel.  256 threads = 4 waves; every wave owns QT tiles of 16 queries (QT = 2 -> 128 queries per
// block for masked windows, whose long key lists dominate the work; QT = 1 -> 64 >= 45 queries for the per-frame
// self attention of unmasked windows).  The masked / unmasked decision is a device flag, so BOTH instantiations are
// launched over all windows and each block exits at once if the window is of the other kind (no host sync).
//
// Per 64-key tile both K and V rows are staged row-major with 16-/8-byte stores (round 1 staged V transposed with 32
// ds_write_b16 per lane and tile: rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.44 for this kernel, the
// LDS array busier than the matrix pipe).
//   K tile [64 keys][256 B], 16-byte slot ^ (key & 15): the QK fragments (16 key rows starting at a multiple of 16) are
//     conflict-free under ds_read_b128's lane groups (tools/lds_swizzle_check.py), and so are the 8-lane store groups.
//   V tile [64 keys][264 B]: the PV operand (8 keys of ONE channel per lane) is read with ds_read_b64_tr_b16: the 16 lanes
//     of a group hand in the 8-byte pieces of a [4 keys][16 channels] block (lane i: key i / 4, channels 4 * (i % 4) ..)
//     and lane i receives column i.  The 16 columns of tile dt are the channels m * 32 + dt * 4 + e (m = 0..3, e = 0..3),
//     so accumulator row i of tile dt is channel (i / 4) * 32 + dt * 4 + i % 4 and every lane ends up with 32 CONSECUTIVE
//     channels of its query -> 16-byte output stores.  The 8-byte row skew (264 = 256 + 8) spreads the 8 key rows x 4
//     pieces of a 32-lane half over all 64 banks.
// Global loads of tile k+1 are issued before the MFMAs of tile k (register prefetch), hiding the gather latency behind
// the matrix work.
__device__ unsigned long long g_attn_prof[8];

#if defined(__HIP_DEVICE_COMPILE__)
// Reductions over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48 hold values of the same query) on the VALU:
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even rows between two registers -- with both operands the
// same value, the two results are the value and its partner's.  __shfl_xor is a ds_bpermute: an LDS round trip (~100+ cycles of
// latency each, four in a row per tile of the online softmax).
typedef unsigned int attn_u32x2 __attribute__((ext_vector_type(2)));
// (elements through .x / .y and __uint_as_float: with __builtin_bit_cast(float, r[i]) hipcc 7.2 reads element 0 twice -- max(r0, r1)
//  compiled to r0 and the reduction silently covered one row pair only; tools: grep the ISA for the v_max after the swap)
static __device__ __forceinline__ float rows_max(float v) {
  attn_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const unsigned a0 = r.x, a1 = r.y;
  v = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const unsigned b0 = r.x, b1 = r.y;
  return fmaxf(__uint_as_float(b0), __uint_as_float(b1));
}
static __device__ __forceinline__ float rows_sum(float v) {
  attn_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const unsigned a0 = r.x, a1 = r.y;
  v = __uint_as_float(a0) + __uint_as_float(a1);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const unsigned b0 = r.x, b1 = r.y;
  return __uint_as_float(b0) + __uint_as_float(b1);
}
#endif

template <int QT, bool MASKED, bool PROF>
__device__ __forceinline__ void attn_block(const AttnParams& p, const int b, const int w, const int head, const int yb,
                                           _Float16* Ks, _Float16* Vs, int* idx_lds, int* tind_lds, int* koff_lds) {
#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-descriptor / LDS-transpose builtins exist in the device pass only)
  typedef _Float16 T;
  constexpr int QB = 64 * QT;                                  // queries per block
  const int nq_total = p.T * p.wsz;
  if (MASKED && yb * QB >= nq_total) return;                   // masked windows need ceil(T*45/QB) blocks only
  if (!MASKED && yb >= p.T) return;
  const int ngrid = p.wsz + p.n_rolled;
  for (int i = threadIdx.x; i < ngrid; i += 256) {
    const int tok = i < p.wsz ? p.own[w * p.wsz + i] : p.rolled[w * p.n_rolled + i - p.wsz];
    idx_lds[i] = tok;
    koff_lds[i] = tok * p.qkv_cs * 2;                          // byte offset of the token's K / V row inside its frame
  }
  // key frames in LDS: a global tind[] read inside the tile loop put a dependent load (and, vmcnt being in-order, a
  // drain of the whole K/V prefetch) in front of every gather -- 6.5 k of the 10.7 k cycles per tile (tools/bench_attn.py)
  if (MASKED && threadIdx.x < p.n_tind) tind_lds[threadIdx.x] = p.tind[threadIdx.x];
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* qg = reinterpret_cast<const T*>(p.q);
  const T* kg = reinterpret_cast<const T*>(p.k);
  const T* vg = reinterpret_cast<const T*>(p.v);
  const T* pkg = reinterpret_cast<const T*>(p.pk);
  const T* pvg = reinterpret_cast<const T*>(p.pv);

  // ---- this lane's queries (column lane&15 of each of the wave's QT 16-query tiles), held as MFMA B fragments
  long long qoff[QT], otok[QT];
  bool qvalid[QT];
  f16x8 qf[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int ql = (wave * QT + qt) * 16 + (lane & 15);
    int fq, tok;
    if (MASKED) {
      const int qi = yb * QB + ql;
      qvalid[qt] = qi < nq_total;
      fq = qvalid[qt] ? qi / p.wsz : 0;
      tok = qvalid[qt] ? qi % p.wsz : 0;
    } else {
      qvalid[qt] = ql < p.wsz;
      fq = yb;
      tok = qvalid[qt] ? ql : 0;
    }
    qoff[qt] = ((long long)b * p.T + fq) * p.Hp * p.Wp + idx_lds[tok];
    otok[qt] = attn_out_token(p, b, fq, idx_lds[tok]);
    const T* qp = qg + qoff[qt] * p.qkv_cs + head * HD + (lane >> 4) * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (qvalid[qt]) qf[qt][s] = *reinterpret_cast<const f16x8*>(qp + s * 32);
      else qf[qt][s] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  const int n_gridkeys = MASKED ? ngrid : p.wsz;               // keys [0, n_gridkeys) come from the token grid
  const int pool0 = (n_gridkeys + 3) & ~3;                     // first pooled key (see load_part)
  const int kpf = MASKED ? pool0 + p.P : p.wsz;                // keys per key frame (incl. the filler keys)
  const int tpf = (kpf + KT - 1) / KT;                         // 64-key tiles per frame (the last one is partial: masked out)
  const int ntiles = (MASKED ? p.n_tind : 1) * tpf;
  const float sc2 = rsqrtf((float)HD) * 1.4426950408889634f;   // softmax scale folded with log2(e)
  const int frame_blk = yb;   // key frame of an unmasked window's block

  f32x4 oacc[QT][8];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -1e30f;
    l_run[qt] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) oacc[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- tile staging: 64 keys x 16 chunks of 8 channels; 16 consecutive lanes read one 256-byte key row
  // (coalesced); rows past the key list are zero.
  u32x4 kr[4], vr[4];
  // K/V rows are gathered with raw buffer loads: one descriptor per tensor over this batch item (built once), the key frame
  // in the scalar offset, and per lane only a 32-bit byte offset inside the frame -- looked up in LDS for the window / rolled
  // tokens (koff_lds, pre-multiplied by the row pitch), one 32-bit multiply for the pooled tokens.  Rows past the key list get the
  // out-of-range offset and come back as zeros.  (Round 1 built 64-bit flat addresses per lane: ~70 VALU instructions per
  // 16-key quarter, 1.2 k of the 5.5 k cycles a wave spends per tile -- in-kernel stamps, tools/bench_attn.py.)
  const int hoff = head * HD;
  const long long item_tok = (long long)b * p.T * p.Hp * p.Wp, item_pool = (long long)b * p.T * p.P;
  const int frame_grid_bytes = p.Hp * p.Wp * p.qkv_cs * 2, frame_pool_bytes = p.P * p.pkv_cs * 2;
  const long long sg64 = (long long)p.T * frame_grid_bytes, sp64 = (long long)p.T * frame_pool_bytes;    // (the host entry refuses items >= 2 GiB)
  const int span_grid = (int)(sg64 < 0x7fffffffll ? sg64 : 0x7fffffffll), span_pool = (int)(sp64 < 0x7fffffffll ? sp64 : 0x7fffffffll);
  const __amdgpu_buffer_rsrc_t rs_k = uniform_buffer_rsrc(kg + item_tok * p.qkv_cs + hoff, span_grid);
  const __amdgpu_buffer_rsrc_t rs_v = uniform_buffer_rsrc(vg + item_tok * p.qkv_cs + hoff, span_grid);
  const __amdgpu_buffer_rsrc_t rs_pk = uniform_buffer_rsrc(pkg != nullptr ? pkg + item_pool * p.pkv_cs + hoff : kg, pkg != nullptr ? span_pool : 0);
  const __amdgpu_buffer_rsrc_t rs_pv = uniform_buffer_rsrc(pvg != nullptr ? pvg + item_pool * p.pkv_cs + hoff : vg, pvg != nullptr ? span_pool : 0);
  const int pool_pitch = p.pkv_cs * 2;
  const int dch16 = (tid & 15) * 16;
  // Key list of one key frame: [0, n_gridkeys) token-grid rows, then the pooled rows.  The pooled part starts at the next
  // multiple of 4 (pool0): every wave fetches 4 consecutive keys per quarter, so a quarter is all-grid or all-pool and needs
  // one wave-uniform branch and one pair of loads; the (at most 3) filler keys in between load zeros and are masked to -inf.
  // one quarter (16 keys) of a tile: 2 x 16-byte loads per lane, issued between the QK MFMA groups of the previous tile
  auto load_part = [&](int fi, int r0, int i) {                 // keys r0 .. r0+63 of key frame #fi
    const int f = __builtin_amdgcn_readfirstlane(MASKED ? tind_lds[fi] : frame_blk);
    const int rq = __builtin_amdgcn_readfirstlane(r0 + 16 * i + wave * 4);   // the wave's 4 keys of this quarter: rq .. rq + 3
    const int r = rq + (lane >> 4);
    if (rq < n_gridkeys) {
      const int vo = r < n_gridkeys ? koff_lds[r] + dch16 : (int)0x80000000;
      const int so = f * frame_grid_bytes;
      kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo, so, 0);
      vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, so, 0);
    } else if (rq < kpf) {
      const int vo = r < kpf ? (r - pool0) * pool_pitch + dch16 : (int)0x80000000;
      const int so = f * frame_pool_bytes;
      kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_pk, vo, so, 0);
      vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_pv, vo, so, 0);
    } else {
      kr[i] = u32x4{0, 0, 0, 0};
      vr[i] = u32x4{0, 0, 0, 0};
    }
  };
  auto load_tile = [&](int fi, int r0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) load_part(fi, r0, i);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int key = c >> 4, dch = c & 15;
      *reinterpret_cast<u32x4*>(&Ks[key * KS_LD + ((dch ^ (key & 15)) << 3)]) = kr[i];
      // two 8-byte stores (the skewed rows are only 8-byte aligned); the halves are written in opposite order by the lanes
      // of chunks 0-7 and 8-15 so that the 16 lanes of a ds_write_b64 group cover 16 different 8-byte bank pairs
      const int sel = dch >> 3;
      const u32x2 lo = {vr[i][0], vr[i][1]}, hi = {vr[i][2], vr[i][3]};
      T* vrow = &Vs[key * VS_LD + dch * 8];
      *reinterpret_cast<u32x2*>(vrow + sel * 4) = sel ? hi : lo;
      *reinterpret_cast<u32x2*>(vrow + (sel ^ 1) * 4) = sel ? lo : hi;
    }
  };

  unsigned long long pf[6] = {0, 0, 0, 0, 0, 0}, ta = 0, tb = 0;
  load_tile(0, 0);
  int fi_n = 0, r0_n = 0;                                       // position of the tile being prefetched
  for (int ti = 0; ti < ntiles; ++ti) {
    const int r0 = r0_n;                                        // first key (within its frame) of the tile consumed now
    r0_n += KT;
    if (r0_n >= kpf) { r0_n = 0; ++fi_n; }
    if constexpr (PROF) ta = __builtin_readcyclecounter();
    store_tile();
    if constexpr (PROF) { tb = __builtin_readcyclecounter(); pf[0] += tb - ta; }
    __syncthreads();
    if constexpr (PROF) { ta = __builtin_readcyclecounter(); pf[1] += ta - tb; }
    const bool more = ti + 1 < ntiles;                  // next tile: its four quarters are requested inside the kt loop below
    if constexpr (PROF) { tb = __builtin_readcyclecounter(); pf[2] += tb - ta; }

    // the next tile's eight loads per lane are requested in ONE batch ahead of the S^T MFMAs: their address lookups (key frame, row
    // offsets: LDS) then cost one wait instead of one per quarter in between the fragment reads (S phase 2 308 -> 1 829 cycles per
    // wave and tile, profiles/r4_attention.txt)
    if (more) load_tile(fi_n, r0_n);
    // ---- S^T tiles: sacc[qt][kt][r] = score(key = k0 + kt*16 + (lane>>4)*4 + r, query = lane&15 of tile qt)
    f32x4 sacc[QT][4];
    PP_ATTN_PRIO_BEGIN();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) sacc[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const T* kp = &Ks[(kt * 16 + (lane & 15)) * KS_LD];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(kp + (((s * 4 + (lane >> 4)) ^ (lane & 15)) << 3));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][s], sacc[qt][kt], 0, 0, 0);
      }
    }
    PP_ATTN_PRIO_END();
    if constexpr (PROF) { ta = __builtin_readcyclecounter(); pf[3] += ta - tb; }
    // ---- online softmax per query tile, in the exp2 domain: p = 2^(s*c - m), c = scale*log2(e), m = running max of s*c.
    // (one fma + one v_exp per score; the key-validity mask only on the partial last tile of a frame; the accumulator
    // rescale only when some lane's running max actually moved)
    f16x8 pfr[QT][2];
    const bool partial = r0 + KT > kpf || (MASKED && pool0 != n_gridkeys && r0 <= n_gridkeys && n_gridkeys < r0 + KT);   // wave-uniform
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (partial) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk_ = r0 + kt * 16 + (lane >> 4) * 4 + r;       // past the list, or a filler key between grid and pool
            if (kk_ >= kpf || (kk_ >= n_gridkeys && kk_ < pool0)) sacc[qt][kt][r] = -1e30f;
          }
      }
      float tmax = sacc[qt][0][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sacc[qt][kt][r]);
      tmax = rows_max(tmax);
      const float m_new = fmaxf(m_run[qt], tmax * sc2);
      const bool moved = m_new > m_run[qt];
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(sacc[qt][kt][r], sc2, -m_new));
          psum += e;
          pfr[qt][kt >> 1][(kt & 1) * 4 + r] = (_Float16)e;
        }
      if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {       // wave-uniform: rare after the first tiles
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[qt][dt][r] *= alpha;
        }
      }
      m_run[qt] = m_new;
      l_run[qt] += psum;                        // per-lane partial; the 4 lane groups are summed at the end
    }
    if constexpr (PROF) { tb = __builtin_readcyclecounter(); pf[4] += tb - ta; }
    // ---- O^T += V^T P^T : k-slot (lane>>4)*8 + i of step j <-> key 32j + (i>>2)*16 + (lane>>4)*4 + (i&3);
    // accumulator row i of tile dt is channel (i/4)*32 + dt*4 + i%4 (column order of the transposing reads)
    const T* vbase = &Vs[((lane >> 4) * 4 + ((lane & 15) >> 2)) * VS_LD + (lane & 3) * 32];
    PP_ATTN_PRIO_BEGIN();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const f16x4 v0 = lds_read_tr16(vbase + (32 * j) * VS_LD + dt * 4);
        const f16x4 v1 = lds_read_tr16(vbase + (32 * j + 16) * VS_LD + dt * 4);
        const f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfr[qt][j], oacc[qt][dt], 0, 0, 0);
      }
    }
    PP_ATTN_PRIO_END();
    if constexpr (PROF) { ta = __builtin_readcyclecounter(); pf[5] += ta - tb; }
    __syncthreads();
  }
  if constexpr (PROF) {
    if (lane == 0 && (blockIdx.x & 15) == 3) {
      for (int i = 0; i < 6; ++i) atomicAdd(&g_attn_prof[i], pf[i]);
      atomicAdd(&g_attn_prof[6], 1ull);
      atomicAdd(&g_attn_prof[7], (unsigned long long)ntiles);
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = rows_sum(l_run[qt]);
    if (qvalid[qt] && otok[qt] >= 0) {
      const float inv = 1.f / l;
      // lane holds rows (lane>>4)*4 + r of every tile dt -> channels (lane>>4)*32 + dt*4 + r: 32 consecutive channels
      T* op = reinterpret_cast<T*>(p.out) + otok[qt] * p.C + head * HD + (lane >> 4) * 32;
#pragma unroll
      for (int d2 = 0; d2 < 4; ++d2) {
        f16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (_Float16)(oacc[qt][2 * d2][r] * inv);
          o[4 + r] = (_Float16)(oacc[qt][2 * d2 + 1][r] * inv);
        }
        *reinterpret_cast<f16x8*>(op + d2 * 8) = o;
      }
    }
  }
#endif
}

// Grid-mapped launch: blockIdx.x = (b, window, head), blockIdx.y = query block / frame.  The masked / unmasked decision is
// a device flag, so blocks of the other kind exit at once.
template <int QT, bool MASKED, bool PROF = false>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) _Float16 Ks[KT * KS_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Vs[KT * VS_LD];
  __shared__ int idx_lds[256];
  __shared__ int tind_lds[ATTN_MAX_KEY_FRAMES];
  __shared__ int koff_lds[256];
  const int head = blockIdx.x % p.heads;
  const int w = (blockIdx.x / p.heads) % p.nW;
  const int b = blockIdx.x / (p.heads * p.nW);
  const bool masked = p.wmask[b * p.nW + w] > 0.f;
  if (masked != MASKED) return;                                // the other instantiation owns this window
  attn_block<QT, MASKED, PROF>(p, b, w, head, (int)blockIdx.y, Ks, Vs, idx_lds, tind_lds, koff_lds);
}

// Persistent launch for the masked windows (the long key lists: ~all of the attention time).  With the grid-mapped
// launch the working blocks (25 % of the grid at the benchmark mask) land unevenly on the CUs and the slowest CU runs 3
// rounds where 2 would do.  Here a compacted list of masked windows (attn_compact_kernel) is walked by 2 blocks per CU.
// Items are ordered (window, head, query block): the ~7 query blocks of one (window, head) read the SAME K / V rows, neighbouring
// windows share their 148 rolled rows.  Consecutive block ids sit on different XCDs (id mod 8), each with its own L2, so every XCD
// gets one CONTIGUOUS eighth of the item list and its blocks walk it with a stride of blocks-per-XCD: at any moment an XCD works on
// ~64 consecutive items = ~9 (window, head) groups, whose rows are fetched into that L2 once (round 2's flat stride spread the
// query blocks of a group over 7 XCDs: PMC fabric traffic 2.05x the algorithmic bytes, profiles/r2v_hbm_traffic_720p.json).
template <int QT, bool PROF = false>
__global__ __launch_bounds__(256, 2) void attn_mfma_persistent_kernel(const AttnParams p, const int gy) {
  __shared__ __attribute__((aligned(16))) _Float16 Ks[KT * KS_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Vs[KT * VS_LD];
  __shared__ int idx_lds[256];
  __shared__ int tind_lds[ATTN_MAX_KEY_FRAMES];
  __shared__ int koff_lds[256];
  const int total = p.work[0] * p.heads * gy;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;     // (gridDim.x is a multiple of 8)
  const int chunk = (total + 7) >> 3;
  for (int it = loc; it < chunk; it += nxb) {
    const int item = xcd * chunk + it;
    if (item >= total) break;
    const int yb = item % gy;
    const int wh = item / gy;
    const int head = wh % p.heads;
    const int bw = p.work[1 + wh / p.heads];
    attn_block<QT, true, PROF>(p, bw / p.nW, bw % p.nW, head, yb, Ks, Vs, idx_lds, tind_lds, koff_lds);
    __syncthreads();                                           // the LDS tables are rewritten by the next item
  }
}

// work[0] = number of masked windows, work[1..] = their flat ids, ascending (one block; B*nW is a few hundred)
__global__ void attn_compact_kernel(const float* __restrict__ wmask, int n, int* __restrict__ work) {
  __shared__ int cnt[256];
  const int tid = threadIdx.x;
  const int per = (n + 255) / 256;
  int c = 0;
  for (int i = tid * per; i < min(n, (tid + 1) * per); ++i) c += wmask[i] > 0.f;
  cnt[tid] = c;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < tid; ++i) base += cnt[i];
  for (int i = tid * per; i < min(n, (tid + 1) * per); ++i)
    if (wmask[i] > 0.f) work[1 + base++] = i;
  if (tid == 255) work[0] = base;
}

// wmask[b, w] = sum_lt max_{window} mask.  One wave per (b, w): lane = position inside the window (wh * ww <= 64), the frames' loads are
// independent (unrolled by 4), the maximum of a frame by a wave reduction, the sum over the frames in frame order.  (One THREAD per window
// walked Lt x wh x ww dependent loads one after the other: 58 us for 144 windows of 11 frames, on the critical path of every generator window.)
template <typename T>
__global__ __launch_bounds__(64) void window_mask_kernel(const T* __restrict__ mask, float* __restrict__ wmask, int B, int Lt, int Hp, int Wp, int wh,
                                                         int ww) {
  const int nww = Wp / ww, nW = (Hp / wh) * nww;
  const int i = blockIdx.x;
  const int b = i / nW, w = i % nW;
  const int y0 = (w / nww) * wh, x0 = (w % nww) * ww;
  const int lane = threadIdx.x, wsz = wh * ww;
  float s = 0.f;
  if (wsz <= 64) {                       // the model's 5 x 9 window: one position per lane
    const bool on = lane < wsz;
    const int y = on ? lane / ww : 0, x = on ? lane % ww : 0;
    const T* mp = mask + ((long long)b * Lt * Hp + y0 + y) * Wp + x0 + x;
#pragma unroll 4
    for (int t = 0; t < Lt; ++t) {
      float mx = on ? to_f32(mp[(long long)t * Hp * Wp]) : -INFINITY;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      s += mx;
    }
  } else {                               // any window size: the lanes loop over the positions
    for (int t = 0; t < Lt; ++t) {
      float mx = -INFINITY;
      for (int q = lane; q < wsz; q += 64)
        mx = fmaxf(mx, to_f32(mask[(((long long)b * Lt + t) * Hp + y0 + q / ww) * Wp + x0 + q % ww]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      s += mx;
    }
  }
  if (lane == 0) wmask[i] = s;
}

}  // namespace pp

using namespace pp;

extern "C" int pp_window_tables(int Hp, int Wp, int wh, int ww, int32_t* own, int32_t* rolled, int capacity_rolled) {
  PP_REQUIRE(Hp > 0 && Wp > 0 && wh > 0 && ww > 0 && Hp % wh == 0 && Wp % ww == 0, PP_ERR_ARG,
             "pp_window_tables: grid %dx%d is not a multiple of the %dx%d window", Hp, Wp, wh, ww);
  const int eh = (wh + 1) / 2, ew = (ww + 1) / 2;
  // kept positions of the four rolled copies (sparse_transformer.py:144-153)
  int n_rolled = 0;
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < wh; ++i)
      for (int j = 0; j < ww; ++j) {
        const bool zr = (k < 2) ? (i < wh - eh) : (i >= eh);
        const bool zc = (k % 2 == 0) ? (j < ww - ew) : (j >= ew);
        if (!(zr && zc)) ++n_rolled;
      }
  if (own == nullptr || rolled == nullptr) return n_rolled;
  const int nwh = Hp / wh, nww = Wp / ww;
  PP_REQUIRE(capacity_rolled >= nwh * nww * n_rolled, PP_ERR_WORKSPACE, "pp_window_tables: rolled capacity %d < %d", capacity_rolled,
             nwh * nww * n_rolled);
  for (int wy = 0; wy < nwh; ++wy)
    for (int wx = 0; wx < nww; ++wx) {
      const int w = wy * nww + wx;
      int n = 0;
      for (int i = 0; i < wh; ++i)
        for (int j = 0; j < ww; ++j) own[w * wh * ww + i * ww + j] = (wy * wh + i) * Wp + wx * ww + j;
      for (int k = 0; k < 4; ++k) {
        // torch.roll(x, shift)[i] = x[(i - shift) mod n]; shifts: tl (-eh,-ew) tr (-eh,+ew) bl (+eh,-ew) br (+eh,+ew)
        const int sh = (k < 2) ? -eh : eh, sw = (k % 2 == 0) ? -ew : ew;
        for (int i = 0; i < wh; ++i)
          for (int j = 0; j < ww; ++j) {
            const bool zr = (k < 2) ? (i < wh - eh) : (i >= eh);
            const bool zc = (k % 2 == 0) ? (j < ww - ew) : (j >= ew);
            if (zr && zc) continue;
            const int y = ((wy * wh + i - sh) % Hp + Hp) % Hp, x = ((wx * ww + j - sw) % Wp + Wp) % Wp;
            rolled[w * n_rolled + n++] = y * Wp + x;
          }
      }
    }
  return n_rolled;
}

extern "C" int pp_window_mask(const void* mask, float* wmask, int B, int Lt, int Hp, int Wp, int wh, int ww, int dtype,
                              void* stream) {
  PP_REQUIRE(mask && wmask && B > 0 && Lt > 0 && wh > 0 && ww > 0 && Hp % wh == 0 && Wp % ww == 0, PP_ERR_ARG,
             "pp_window_mask: bad arguments (window %d x %d: any window size)", wh, ww);
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_window_mask: dtype %d", dtype);
  const int n = B * (Hp / wh) * (Wp / ww);
  if (dtype == PP_F16)
    hipLaunchKernelGGL((window_mask_kernel<_Float16>), dim3(n), dim3(64), 0, (hipStream_t)stream,
                       (const _Float16*)mask, wmask, B, Lt, Hp, Wp, wh, ww);
  else
    hipLaunchKernelGGL((window_mask_kernel<float>), dim3(n), dim3(64), 0, (hipStream_t)stream, (const float*)mask,
                       wmask, B, Lt, Hp, Wp, wh, ww);
  return launch_status("pp_window_mask");
}

extern "C" int pp_sparse_window_attention(const pp_attn_args_t* a, void* stream) {
  PP_REQUIRE(a != nullptr, PP_ERR_ARG, "pp_sparse_window_attention: null args");
  PP_REQUIRE(a->dtype == PP_F32 || a->dtype == PP_F16, PP_ERR_DTYPE, "pp_sparse_window_attention: dtype %d", a->dtype);
  PP_REQUIRE(a->B > 0 && a->T > 0 && a->heads > 0 && a->C == a->heads * HD, PP_ERR_ARG,
             "pp_sparse_window_attention: C=%d heads=%d (head dim must be %d)", a->C, a->heads, HD);
  PP_REQUIRE(a->Hp % a->wh == 0 && a->Wp % a->ww == 0, PP_ERR_ARG, "pp_sparse_window_attention: grid not padded to the window");
  PP_REQUIRE(a->wh * a->ww + a->n_rolled <= 256 && a->wh * a->ww <= 64, PP_ERR_ARG, "pp_sparse_window_attention: window too large");
  PP_REQUIRE(a->q && a->k && a->v && a->own && a->rolled && a->tind && a->wmask && a->out && (a->P == 0 || (a->pk && a->pv)),
             PP_ERR_ARG, "pp_sparse_window_attention: null pointer");
  PP_REQUIRE(a->n_tind > 0 && a->n_tind <= a->T && a->n_tind <= ATTN_MAX_KEY_FRAMES, PP_ERR_ARG, "pp_sparse_window_attention: n_tind %d (1..min(T, %d))", a->n_tind,
             ATTN_MAX_KEY_FRAMES);
  const int esz = a->dtype == PP_F16 ? 2 : 4;
  PP_REQUIRE((a->qkv_cstride * esz) % 16 == 0 && (a->pkv_cstride * esz) % 16 == 0 && (uintptr_t)a->q % 16 == 0 &&
                 (uintptr_t)a->k % 16 == 0 && (uintptr_t)a->v % 16 == 0 && (uintptr_t)a->out % 16 == 0,
             PP_ERR_ALIGN, "pp_sparse_window_attention: 16-byte alignment violated");
  AttnParams p;
  p.B = a->B; p.T = a->T; p.Hp = a->Hp; p.Wp = a->Wp; p.C = a->C; p.heads = a->heads; p.wsz = a->wh * a->ww;
  p.n_rolled = a->n_rolled; p.P = a->P; p.n_tind = a->n_tind; p.nW = (a->Hp / a->wh) * (a->Wp / a->ww);
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.qkv_cs = a->qkv_cstride;
  p.pk = (const char*)a->pk; p.pv = (const char*)a->pv; p.pkv_cs = a->pkv_cstride;
  p.own = a->own; p.rolled = a->rolled; p.tind = a->tind; p.wmask = a->wmask; p.out = (char*)a->out; p.work = nullptr;
  PP_REQUIRE((a->out_h == 0 && a->out_w == 0) || (a->out_h > 0 && a->out_h <= a->Hp && a->out_w > 0 && a->out_w <= a->Wp), PP_ERR_ARG,
             "pp_sparse_window_attention: compact output grid %dx%d outside the padded grid %dx%d", a->out_h, a->out_w, a->Hp, a->Wp);
  p.out_h = a->out_h; p.out_w = a->out_w;
  hipStream_t st = (hipStream_t)stream;
  const unsigned gx = (unsigned)(p.B * p.nW * p.heads);
  if (a->dtype == PP_F16 && a->impl != 1) {
    // the MFMA kernels address K / V rows with 32-bit byte offsets inside one batch item (raw buffer loads)
    PP_REQUIRE((long long)a->T * a->Hp * a->Wp * a->qkv_cstride * 2 < (1ll << 31) && (long long)a->T * a->P * a->pkv_cstride * 2 < (1ll << 31),
               PP_ERR_ARG, "pp_sparse_window_attention: the K/V tensors of one batch item must stay below 2 GiB (T=%d, %dx%d tokens)", a->T, a->Hp, a->Wp);
    // masked windows: 128-query blocks over the window's T*45 queries; unmasked: one 64-query block per frame.  Both
    // grids cover every window; blocks of the wrong kind return immediately (device-side flag, no host sync).
    const unsigned gy_m = (unsigned)((p.T * p.wsz + 127) / 128);
    if (a->work != nullptr && a->impl != 6) {
      PP_REQUIRE(a->work_ints >= 1 + p.B * p.nW, PP_ERR_WORKSPACE, "pp_sparse_window_attention: work needs %d ints, got %d", 1 + p.B * p.nW,
                 a->work_ints);
      p.work = (const int*)a->work;
      hipLaunchKernelGGL(attn_compact_kernel, dim3(1), dim3(256), 0, st, p.wmask, p.B * p.nW, (int*)a->work);
      static int n_cu = 0;
      if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
      }
      if (a->impl == 5) hipLaunchKernelGGL((attn_mfma_persistent_kernel<2, true>), dim3((2 * n_cu + 7) / 8 * 8), dim3(256), 0, st, p, (int)gy_m);
      else hipLaunchKernelGGL((attn_mfma_persistent_kernel<2>), dim3((2 * n_cu + 7) / 8 * 8), dim3(256), 0, st, p, (int)gy_m);
    } else {
      hipLaunchKernelGGL((attn_mfma_kernel<2, true>), dim3(gx, gy_m), dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL((attn_mfma_kernel<1, false>), dim3(gx, (unsigned)p.T), dim3(256), 0, st, p);
  } else {
    const unsigned gy = (unsigned)((p.T * p.wsz + 3) / 4);
    if (a->dtype == PP_F16) hipLaunchKernelGGL((attn_ref_kernel<_Float16>), dim3(gx, gy), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_ref_kernel<float>), dim3(gx, gy), dim3(256), 0, st, p);
  }
  return launch_status("pp_sparse_window_attention");
}

// [diagnostic] phase counters of attn_mfma_kernel<2, true, PROF>: {store_tile, barrier, load issue, S mfma, softmax, PV mfma,
// waves, tiles} (cycles summed over the sampled waves); read-and-clear.
extern "C" int pp_debug_attn_prof(unsigned long long* out) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::g_attn_prof), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(pp::g_attn_prof), zero, sizeof(zero));
  return (int)e;
}
