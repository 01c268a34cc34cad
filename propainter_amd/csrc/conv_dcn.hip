// Modulated deformable 3x3 convolution (torchvision.ops.deform_conv2d as called at model/propainter.py:67-69 and
// model/recurrent_flow_completion.py:42-44: stride 1, pad 1, dilation 1, 16 offset groups), fp16, PATCH-STAGED.
//
// The first kernel (conv_gemm.hip, DEFORM = true) gathers 4 corners x 16 bytes per (pixel, group, tap) straight from
// HBM/L2: 64-byte sectors for 16 useful bytes, 36 reads of every input pixel per tile -- measured 672 MB of fabric traffic
// per 69 MB-algorithmic launch and 86 TFLOP/s.  Here a block owns an 8 x 16 tile of output pixels and walks the K range in
// blocks of 32 input channels (4 offset groups of 8 channels / 2 of 16; the table order of pp_conv_build_ktable):
//   * the offsets (dy, dx) and modulation masks of the block's groups for all 9 taps are staged in LDS with wide loads;
//   * the 32-channel input patch of the tile -- the tile grown by 7 + 6 pixels and SHIFTED by the tile's mean offset (the
//     generator adds the optical flow to every offset pair, propainter.py:61-62, so the samples of a tile move together) --
//     is staged ONCE (64 contiguous bytes per pixel = one full sector) and serves all 9 taps x 4 corners from LDS;
//   * per tap (one 32-deep K step) each lane computes exactly its own MFMA A fragment: the bilinear, mask-modulated
//     sample of 8 channels (corner weights in fp32, the 4-corner blend in packed fp16: the sample is an fp16 MFMA
//     operand anyway) for pixel (lane & 15) and group slot (lane >> 4) -- no A tile in LDS at all;
//   * the weights of a tap (128 couts x 32 k = 8 KB, the same for every block of the launch) go through ONE LDS stage per block:
//     two LDS-DMA pieces per wave, issued right after every wave has read the previous tap's fragments, land while the MFMAs of that
//     tap and the sampling of the next one run.  (Rounds 2-3 had every wave fetch all 128 couts per lane straight from L2: 32 KB per
//     block and tap through the texture path, the largest item of the ablations in profiles/r2_dcn_patch_kernel.txt; the stage fits
//     since the modulation masks are staged per channel block -- 80-byte rows instead of 144-byte rows holding two blocks.)
//   * a sample whose corners fall outside the staged patch (|offset - tile mean| >= 5) reads those corners from global
//     memory -- all corner reads of the tap in flight together, chosen per wave (round 6, see the sampling section of the kernel):
//     slower, never wrong.  Outside the image every corner contributes zero (torchvision's bilinear_interpolate).
// 4 waves, wave tile 32 pixels x 128 couts, v_mfma_f32_16x16x32_f16, fp32 accumulation; 2 blocks per CU.
#include "conv_params.h"

namespace pp {

constexpr int DCN_TW = 16, DCN_R0 = 6, DCN_PW = DCN_TW + 13;   // tile width, patch margin, patch columns
constexpr int DCN_OSTR = 72;                                 // (dy, dx) pairs: fp16 per pixel and stage row: 144 bytes = 9 x 16-byte units
constexpr int DCN_MSTR = 40;                                 // modulation masks: 36 fp16 (nine 8-byte units) per pixel in 80-byte rows
constexpr int DCN_WST_BYTES = 128 * 64;                      // weight stage of one tap: 128 couts x 32 k, 64-byte rows, 16-byte slot ^ ((row >> 1) & 3)

typedef __attribute__((address_space(3))) void* dcn_lptr3_t;
#if defined(__HIP_DEVICE_COMPILE__)
// (a free function with by-value arguments: see conv_halo.h)
static __device__ __forceinline__ void dcn_dma_weights(__amdgpu_buffer_rsrc_t r, char* dst, int voff0, int voff1, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (dcn_lptr3_t)dst, 16, voff0, soff, 0, 0);
  // second piece from the same M0 value: the instruction offset is added to the LDS address AND to the global one (voff1 carries -1024)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (dcn_lptr3_t)dst, 16, voff1, soff, 1024, 0);
}
#endif

// CG: channels per offset group (8 or 16).  GB = 32 / CG groups per 32-channel block.
// DCN_TH: tile rows.  8 (128 pixels, two 16-pixel M tiles per wave) is the throughput form; 4 (64 pixels, one M tile per wave) halves a
// block's work for launches that do not fill the chip anyway: the recurrent steps of flow completion run over 2 x 90 x 160 pixels = 225
// tiles of 128 pixels for 512 block slots -- their time is ONE block's latency.
template <int CG, int DCN_TH, bool STATS = false>
__global__ __launch_bounds__(256, 2) void conv_dcn_patch_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NPX = DCN_TH * DCN_TW, MT = NPX / 64;       // pixels per tile, 16-pixel M tiles per wave
  constexpr int DCN_PH = DCN_TH + 13;                        // patch rows
  constexpr int DCN_PATCH_BYTES = DCN_PH * DCN_PW * 64;
  constexpr int DCN_OFFS_BYTES = NPX * DCN_OSTR * 2;
  constexpr int DCN_MSKS_BYTES = NPX * DCN_MSTR * 2;
  constexpr int DCN_LDS = DCN_PATCH_BYTES + DCN_OFFS_BYTES + DCN_MSKS_BYTES + DCN_WST_BYTES + 64;
  static_assert(DCN_LDS <= 80 * 1024 && (MT == 1 || MT == 2), "two blocks per CU");
  constexpr int GB = 32 / CG;                 // offset groups per channel block
  constexpr int NOFF = GB * 9 * 2;            // fp16 offsets per pixel and block (dy, dx interleaved)
  constexpr int NMSK = GB * 9;                // fp16 masks per pixel and block
  __shared__ __attribute__((aligned(16))) char lds[DCN_LDS];
  char* const patch = lds;
  _Float16* const offs = reinterpret_cast<_Float16*>(lds + DCN_PATCH_BYTES);          // [128][72]: 36 (dy, dx) pairs
  _Float16* const msks = reinterpret_cast<_Float16*>(lds + DCN_PATCH_BYTES + DCN_OFFS_BYTES);      // [128][40]: 36 masks
  char* const wst = lds + DCN_PATCH_BYTES + DCN_OFFS_BYTES + DCN_MSKS_BYTES;          // [128 couts][64 B]
  constexpr int OG = DCN_OSTR / NOFF;         // channel blocks served by one staged offset row (gen 1, flow completion 2)
  constexpr int MG = 36 / NMSK;               // ... by one staged mask row (1 / 2)
  float* const red = reinterpret_cast<float*>(lds + DCN_PATCH_BYTES + DCN_OFFS_BYTES + DCN_MSKS_BYTES + DCN_WST_BYTES);     // [4 waves][2] + shift[2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tn = bid % p.tiles_n;
  int tile = bid / p.tiles_n;
  const int tiles_x = (p.W + DCN_TW - 1) / DCN_TW, tiles_y = (p.H + DCN_TH - 1) / DCN_TH;
  const int txi = tile % tiles_x; tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * DCN_TH, tx0 = txi * DCN_TW;
  const int n0 = tn * 128;
  const long long img0 = (long long)n * p.H * p.W;

  // the lane's pixels (M tiles of the wave): tile-local index q = (wave * MT + mt) * 16 + l15
  int oy[MT], ox[MT];
  bool pin[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int q = (wave * MT + mt) * 16 + l15;
    oy[mt] = ty0 + q / DCN_TW;
    ox[mt] = tx0 + q % DCN_TW;
    pin[mt] = oy[mt] < p.H && ox[mt] < p.W;
  }
  // group slot of the lane inside a 32-channel block: CG 8 -> group l4, channels l4*8..; CG 16 -> group l4 >> 1, half l4 & 1
  const int gi = CG == 8 ? l4 : (l4 >> 1);

  const __amdgpu_buffer_rsrc_t rw = uniform_buffer_rsrc(p.weight, p.cout_pad * p.kchunks * 16);
  // weight stage of a K step: pieces 2 * wave, 2 * wave + 1 (16 couts x 64 B each) by this wave; lane -> (row lane >> 2, physical
  // slot lane & 3), which holds the 16 bytes at k = step * 32 + 8 * (slot ^ ((row >> 1) & 3))
  int wdma_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = n0 + (wave * 2 + j) * 16 + (lane >> 2);
    if (row >= p.cout_pad) row = p.cout_pad - 1;            // clamped rows feed accumulators that are never stored
    wdma_voff[j] = row * p.kchunks * 16 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4) - j * 1024;     // (piece 1: see dcn_dma_weights)
  }
  char* const wst_wave = wst + wave * 2048;
  // fragment nt of the lane: cout row nt * 16 + l15, k slot l4
  const int wfrag = l15 * 64 + ((l4 ^ ((l15 >> 1) & 3)) << 4);

  f32x4 acc[8][MT];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] unsigned n_samples = 0, n_outside = 0;      // STATS instantiation only (pp_conv_args_t.dcn_stats)

  const int nblocks = p.kchunks / 36;         // 32-channel blocks (9 taps x 4 chunks each)
  u32x4 wcur[8];
  dcn_dma_weights(rw, wst_wave, wdma_voff[0], wdma_voff[1], 0);      // K step 0: lands during the first block's staging
  int shift_y = 0, shift_x = 0;

  for (int cb = 0; cb < nblocks; ++cb) {
    const int4 e0 = p.ktable[cb * 36];                       // first chunk of the block: source id, first group, channel offset
    const int s = e0.z & 0xff, g0 = (e0.z >> 8) & 0xff;
    const char* sptr = s == 1 ? p.src[1].ptr : s == 2 ? p.src[2].ptr : s == 3 ? p.src[3].ptr : p.src[0].ptr;
    const int scs = s == 1 ? p.src[1].cstride : s == 2 ? p.src[2].cstride : s == 3 ? p.src[3].cstride : p.src[0].cstride;
    const int sco = (s == 1 ? p.src[1].choff : s == 2 ? p.src[2].choff : s == 3 ? p.src[3].choff : p.src[0].choff) + e0.w;
    __syncthreads();                                         // previous block's patch / offsets fully consumed
#if defined(PP_DIAG)
    const int dbg = p.tap_w;                                 // [diagnostic] tools/bench_dcn.py ablations (impl 91..105)
#else
    constexpr int dbg = 0;
#endif
    // ---- stage (dy, dx) pairs / masks: 144 contiguous bytes per pixel (nine 16-byte loads) hold the offsets of OG channel
    //      blocks and the masks of MG channel blocks; loads of a thread are issued together (a rolled loop pays one memory
    //      latency per unit: measured 42 of the kernel's 121 us with 4- and 8-byte units)
    auto stage144 = [&](const int first_channel, _Float16* dst) {
      constexpr int NU = NPX * 9, ITER = (NU + 255) / 256;
      u32x4 v[ITER];
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        const int q = i / 9, u = i - q * 9;
        const int py = ty0 + q / DCN_TW, px = tx0 + q % DCN_TW;
        v[k] = u32x4{0, 0, 0, 0};
        if (i < NU && py < p.H && px < p.W)
          v[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const _Float16*>(p.dcn) +
                                                 (img0 + (long long)py * p.W + px) * p.dcn_cstride + first_channel + u * 8);
      }
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        if (i < NU) *reinterpret_cast<u32x4*>(dst + (i / 9) * DCN_OSTR + (i % 9) * 8) = v[k];
      }
    };
    // ---- masks: 72 contiguous bytes per pixel (nine 8-byte loads: a block's masks start at a multiple of 8 bytes only) hold the
    //      masks of MG channel blocks
    auto stage72 = [&](const int first_channel, _Float16* dst) {
      constexpr int NU = NPX * 9, ITER = (NU + 255) / 256;
      u32x2 v[ITER];
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        const int q = i / 9, u = i - q * 9;
        const int py = ty0 + q / DCN_TW, px = tx0 + q % DCN_TW;
        v[k] = u32x2{0, 0};
        if (i < NU && py < p.H && px < p.W)
          v[k] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const _Float16*>(p.dcn) +
                                                 (img0 + (long long)py * p.W + px) * p.dcn_cstride + first_channel + u * 4);
      }
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        if (i < NU) *reinterpret_cast<u32x2*>(dst + (i / 9) * DCN_MSTR + (i % 9) * 4) = v[k];
      }
    };
    if (!(dbg & 1)) {
      if (cb % OG == 0) stage144(2 * 9 * g0, offs);
      if (cb % MG == 0) stage72(p.dcn_mask_off + 9 * g0, msks);
    }
    __syncthreads();
    if (cb == 0) {
      // ---- mean offset of the tile (first block's groups) -> integer patch shift, once per block
      float sy = 0.f, sx = 0.f;
      for (int i = tid; i < NPX * 36; i += 256) {            // all 36 pairs of the staged row
        const int q = i / 36, j = i - q * 36;
        sy += (float)offs[q * DCN_OSTR + 2 * j];
        sx += (float)offs[q * DCN_OSTR + 2 * j + 1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { sy += __shfl_xor(sy, o); sx += __shfl_xor(sx, o); }
      if (lane == 0) { red[wave * 2] = sy; red[wave * 2 + 1] = sx; }
      __syncthreads();
      if (tid == 0) {
        const float inv = 1.f / (float)(NPX * 36);
        float my = (red[0] + red[2] + red[4] + red[6]) * inv, mx = (red[1] + red[3] + red[5] + red[7]) * inv;
        my = fminf(fmaxf(my, -4096.f), 4096.f);
        mx = fminf(fmaxf(mx, -4096.f), 4096.f);
        red[8] = rintf(my);
        red[9] = rintf(mx);
      }
      __syncthreads();
      shift_y = (int)red[8];
      shift_x = (int)red[9];
    }
    // ---- stage the 32-channel patch: rows [py0, py0 + PH), columns [px0, px0 + PW); 16-byte slot c of patch pixel i is
    //      stored at slot (c + 2 * (i >> 2)) & 3.  A corner read is a ds_read_b128 by lane (pixel l15, channel slot l4); the hardware
    //      serves it in the lane groups {0-3, 12-15, 20-27}, ... = pixels {0-3, 12-15} of slot l4 = 0 with pixels {4-11} of slot 1: for a
    //      smooth offset field the 16 lanes hit 16 consecutive patch pixels, whose (pixel & 3, physical slot) pairs must all differ.
    //      With the rotation (c + (i >> 2)) the pixels 12-15 of slot 0 and 8-11 of slot 1 met in the same banks (2-way conflict on
    //      half of the lanes: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50, profiles/r2p_lds_conflicts.txt); 2 * (i >> 2) separates
    //      them for every alignment of the 16 pixels (tools/lds_swizzle_check.py, tests/test_host_logic_cpu.py).
    const int py0 = ty0 - 1 - DCN_R0 + 1 + shift_y - 0, px0 = tx0 - 1 - DCN_R0 + 1 + shift_x - 0;   // = tile origin - 6 + shift
    if (!(dbg & 2)) {
      constexpr int NC = DCN_PH * DCN_PW * 4, ITER = (NC + 255) / 256, BATCH = ITER % 5 == 0 ? 5 : (ITER % 4 == 0 ? 4 : (ITER % 3 == 0 ? 3 : 1));
      static_assert(ITER % BATCH == 0, "patch chunks per thread");
#pragma unroll 1
      for (int k0 = 0; k0 < ITER; k0 += BATCH) {
        u32x4 vp[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int i = tid + (k0 + k) * 256;
          const int pi = i >> 2, c = i & 3;
          const int yy = py0 + pi / DCN_PW, xx = px0 + pi % DCN_PW;
          vp[k] = u32x4{0, 0, 0, 0};
          if (i < NC && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
            vp[k] = *reinterpret_cast<const u32x4*>(sptr + ((img0 + (long long)yy * p.W + xx) * scs + sco + c * 8) * 2);
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int i = tid + (k0 + k) * 256;
          const int pi = i >> 2, c = i & 3;
          if (i < NC) *reinterpret_cast<u32x4*>(patch + pi * 64 + (((c + 2 * (pi >> 2)) & 3) << 4)) = vp[k];
        }
      }
    }
    __syncthreads();

    // ---- 9 taps = 9 K steps of 32
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      const int step = cb * 9 + t;
      const bool more = step + 1 < nblocks * 9;
      f16x8 af[MT];
      const int trow = t / 3, tcol = t - trow * 3;
      // Sampling.  Per M tile: corner weights and patch position of the lane's sample.  Then one of two forms, chosen PER WAVE:
      //   * no lane of the wave leaves the staged patch (the common case on smooth flows): four LDS reads and the packed fp16 blend per
      //     M tile, as in rounds 2-5;
      //   * some lane does: EVERY corner read of the tap is put in flight before the first one is waited for -- the far lanes' corners
      //     as raw buffer loads straight from the image (an offset beyond the image's bytes returns zeros: corners outside the image
      //     and lanes that stay inside the patch cost no traffic), the patch reads of the other lanes, then one blend in torchvision's
      //     corner order over whichever copy the lane uses.  The first form of this path read the corners of an out-of-patch sample one
      //     by one, each inside its own validity branch: four serialised L2 round trips per sample (61 ms per clip on the stress clip's
      //     flows against 30 on smooth ones; profiles/r6_dcn_far_samples.txt).  The arithmetic of both forms is the same.
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      _Float16 hw[MT][4];
      int ry0[MT], rx0[MT];
      bool use[MT], inpatch[MT];
      bool far_lane = false;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        use[mt] = pin[mt] && !(dbg & 4);
        const int q = (wave * MT + mt) * 16 + l15;
        const h2 dd = *reinterpret_cast<const h2*>(offs + q * DCN_OSTR + (cb % OG) * NOFF + 2 * (gi * 9 + t));
        const float mk = (float)msks[q * DCN_MSTR + (cb % MG) * NMSK + gi * 9 + t];
        const float py = (float)(oy[mt] - 1 + trow) + (float)dd[0];
        const float px = (float)(ox[mt] - 1 + tcol) + (float)dd[1];
        const float fy = floorf(py), fx = floorf(px);
        const float ly = py - fy, lx = px - fx;
        // corner weights (x modulation mask) in fp16 pairs; the patch is zero outside the image, so corners outside the
        // image contribute zero by themselves (= torchvision's per-corner test and its whole-sample test)
        const float w11 = ly * lx * mk, w10 = (ly - ly * lx) * mk, w01 = (lx - ly * lx) * mk;
        const float w00 = mk - w11 - w10 - w01;
        hw[mt][0] = (_Float16)w00; hw[mt][1] = (_Float16)w01; hw[mt][2] = (_Float16)w10; hw[mt][3] = (_Float16)w11;
        // clamp far-out samples before the int conversion (they are outside the patch and outside the image anyway)
        ry0[mt] = (int)fminf(fmaxf(fy, -1.0e6f), 1.0e6f) - py0;
        rx0[mt] = (int)fminf(fmaxf(fx, -1.0e6f), 1.0e6f) - px0;
        inpatch[mt] = (unsigned)ry0[mt] < (unsigned)(DCN_PH - 1) && (unsigned)rx0[mt] < (unsigned)(DCN_PW - 1);
        if constexpr (STATS) {            // one sample per (pixel, offset group, tap): lanes with the same (pixel, group) count once
          if (use[mt] && (CG == 8 || (l4 & 1) == 0)) {
            ++n_samples;
            n_outside += inpatch[mt] ? 0u : 1u;
          }
        }
        far_lane = far_lane || (use[mt] && !inpatch[mt]);
      }
      if (__builtin_amdgcn_ballot_w64(far_lane) == 0ull) {
        // ---- every sample of the wave inside the staged patch: 4 LDS reads, packed fp16 blend
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          h2 r2[4] = {h2{0, 0}, h2{0, 0}, h2{0, 0}, h2{0, 0}};
          if (use[mt]) {
            const int pi0 = ry0[mt] * DCN_PW + rx0[mt];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int pi = pi0 + (c >> 1) * DCN_PW + (c & 1);
              const u32x4 raw = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                  (const __attribute__((address_space(3))) char*)patch + pi * 64 + (((l4 + 2 * (pi >> 2)) & 3) << 4));
              const h2* hv = reinterpret_cast<const h2*>(&raw);
              const h2 wv = h2{hw[mt][c], hw[mt][c]};
#pragma unroll
              for (int j = 0; j < 4; ++j) r2[j] = __builtin_elementwise_fma(wv, hv[j], r2[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { af[mt][2 * j] = r2[j][0]; af[mt][2 * j + 1] = r2[j][1]; }
        }
      } else {
        // ---- some sample of the wave leaves the patch: all reads of the tap first, one wait, one blend
        const __amdgpu_buffer_rsrc_t rimg = uniform_buffer_rsrc(sptr + (img0 * scs + sco) * 2, p.H * p.W * scs * 2 - sco * 2);
        u32x4 graw[MT][4], lraw[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bool far = use[mt] && !inpatch[mt];
          const int y0 = ry0[mt] + py0, x0 = rx0[mt] + px0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yy = y0 + (c >> 1), xx = x0 + (c & 1);
            const bool ok = far && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            graw[mt][c] = __builtin_amdgcn_raw_buffer_load_b128(rimg, ok ? ((yy * p.W + xx) * scs + l4 * 8) * 2 : (int)0x80000000, 0, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int pi0 = inpatch[mt] ? ry0[mt] * DCN_PW + rx0[mt] : 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int pi = pi0 + (c >> 1) * DCN_PW + (c & 1);
            lraw[mt][c] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                (const __attribute__((address_space(3))) char*)patch + pi * 64 + (((l4 + 2 * (pi >> 2)) & 3) << 4));
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          h2 r2[4] = {h2{0, 0}, h2{0, 0}, h2{0, 0}, h2{0, 0}};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            u32x4 raw = inpatch[mt] ? lraw[mt][c] : graw[mt][c];
            if (!use[mt]) raw = u32x4{0, 0, 0, 0};
            const h2* hv = reinterpret_cast<const h2*>(&raw);
            const h2 wv = h2{hw[mt][c], hw[mt][c]};
#pragma unroll
            for (int j = 0; j < 4; ++j) r2[j] = __builtin_elementwise_fma(wv, hv[j], r2[j]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { af[mt][2 * j] = r2[j][0]; af[mt][2 * j + 1] = r2[j][1]; }
        }
      }
      // ---- this step's weights: every wave's two pieces have landed (own DMA: vmcnt, the others': barrier), the fragments go to
      //      registers, and once every wave holds its fragments the stage is refilled with the next step's weights
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        wcur[nt] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((const __attribute__((address_space(3))) char*)wst + nt * 1024 + wfrag);
      __syncthreads();
      if (more && !(dbg & 8)) dcn_dma_weights(rw, wst_wave, wdma_voff[0], wdma_voff[1], (step + 1) * 64);
      if (!(dbg & 8))
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wcur[nt]), af[mt], acc[nt][mt], 0, 0, 0);
    }
  }

  if constexpr (STATS) {
    if (tn == 0) {                          // every cout tile samples the same positions: count them once
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { n_samples += __shfl_xor(n_samples, o); n_outside += __shfl_xor(n_outside, o); }
      if (lane == 0) {
        atomicAdd(p.dcn_stats + 0, (unsigned long long)n_samples);
        atomicAdd(p.dcn_stats + 1, (unsigned long long)n_outside);
      }
    }
  }
  // ---- epilogue: lane holds couts nt*16 + l4*4 + r (r = 0..3) of pixel mt*16 + l15
  float bv[8][4];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = n0 + nt * 16 + l4 * 4 + r;
      bv[nt][r] = (p.bias != nullptr && co < p.cout_g) ? p.bias[co] : 0.f;          // 32 independent loads: one latency
    }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!pin[mt]) continue;
    const long long m = img0 + (long long)oy[mt] * p.W + ox[mt];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int co = n0 + nt * 16 + l4 * 4;
      if (co >= p.cout_g) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = apply_act((acc[nt][mt][r] + bv[nt][r]) * p.out_scale, p.act, p.act_param);
      }
      _Float16* op = reinterpret_cast<_Float16*>(p.out) + m * p.out_cstride + p.out_choff + co;
      if (co + 3 < p.cout_g && (((m * p.out_cstride + p.out_choff + co) & 3) == 0)) {
        *reinterpret_cast<f16x4*>(op) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < p.cout_g) op[r] = (_Float16)v[r];
      }
    }
  }
#endif
}

// Returns -1000 when the layer is outside this kernel's family (caller falls back to the register-staged gather).
int conv_dcn_dispatch(const ConvParams& pin_, hipStream_t stream, int dbg) {
  ConvParams p = pin_;
  p.tap_w = dbg;
  if (p.dcn == nullptr || p.groups != 1 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1 || p.OH != p.H || p.OW != p.W) return -1000;
  if (p.out_f16 == 0 || p.residual != nullptr || p.preadd != nullptr || p.fuse != 0 || p.act2 != 0 || p.act >= PP_ACT_SIGMOID) return -1000;
  if (p.kchunks % 36 != 0 || p.dcn_mask_off != 288) return -1000;
  const int cin = p.kchunks / 9 * 8;                      // total input channels (all sources)
  if (cin != 128 && cin != 256) return -1000;             // 16 offset groups of 8 / 16 channels
  if ((p.dcn_cstride & 7) != 0 || ((uintptr_t)p.dcn & 15) != 0 || (long long)p.cout_pad * p.kchunks * 16 >= (1ll << 31)) return -1000;
  for (int i = 0; i < p.nsrc; ++i)
    if ((p.src[i].cstride & 7) != 0 || (p.src[i].choff & 7) != 0) return -1000;
  p.tiles_n = (p.cout_g + 127) / 128;
  const long long tiles_x = (p.W + DCN_TW - 1) / DCN_TW;
  const long long nblk8 = (long long)p.N * ((p.H + 7) / 8) * tiles_x * p.tiles_n, nblk4 = (long long)p.N * ((p.H + 3) / 4) * tiles_x * p.tiles_n;
  if (nblk4 >= (1ll << 31)) return -1000;
  // launches of at most one block per CU: 64-pixel tiles (twice the blocks, half the latency of a block); dbg 16 / 32 force 8 / 4 rows
  const bool small = (dbg & 32) || (!(dbg & 16) && nblk8 <= 256);
  p.tap_w = dbg & 15;
  if (p.dcn_stats != nullptr) {          // counting instantiations (bench.py's fallback report; same arithmetic, same results)
    if (small) {
      if (cin == 128) hipLaunchKernelGGL((conv_dcn_patch_kernel<8, 4, true>), dim3((unsigned)nblk4), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((conv_dcn_patch_kernel<16, 4, true>), dim3((unsigned)nblk4), dim3(256), 0, stream, p);
    } else {
      if (cin == 128) hipLaunchKernelGGL((conv_dcn_patch_kernel<8, 8, true>), dim3((unsigned)nblk8), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((conv_dcn_patch_kernel<16, 8, true>), dim3((unsigned)nblk8), dim3(256), 0, stream, p);
    }
  } else if (small) {
    if (cin == 128) hipLaunchKernelGGL((conv_dcn_patch_kernel<8, 4>), dim3((unsigned)nblk4), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_dcn_patch_kernel<16, 4>), dim3((unsigned)nblk4), dim3(256), 0, stream, p);
  } else {
    if (cin == 128) hipLaunchKernelGGL((conv_dcn_patch_kernel<8, 8>), dim3((unsigned)nblk8), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_dcn_patch_kernel<16, 8>), dim3((unsigned)nblk8), dim3(256), 0, stream, p);
  }
  return launch_status("pp_conv2d(dcn)");
}

}  // namespace pp
