// Lean shared epilogue of the LDS-DMA convolution kernels (conv_gemm_v2.hip / conv_gemm_v3.hip).
//
// Measured with in-kernel cycle stamps (tools/bench_conv.py impl 75): the first epilogue (fp32 staging tile, then
// bias / activation / residual per element inside the store loop, every uniform option re-tested per element) cost
// 12-13 k cycles per 64x64 wave tile -- instruction-bound, as much as 8 K-steps of MFMA work.  Here:
//   phase 1 (registers): bias + scale + activation on the accumulators, with the uniform options hoisted into ONE switch;
//   staging: fp16 pairs straight into LDS when the output is fp16 and there is no residual (half the LDS traffic),
//            otherwise fp32 (the residual is added in fp32 before the single final rounding, as before);
//   phase 2: one ds_read_b128 (+ residual add) + one 16-byte global store per 8 couts, nothing else.
// Every lane handles 64 outputs either way; the minimum is ~6 VALU per output.
#pragma once
#include "conv_params.h"

namespace pp {

// RowMap: prow (0..WM-1, pixel row of the wave tile) -> flat output pixel index, or -1 when the row is outside the image.
// TNT / A0: the accumulator array may be wider than the WN couts handled by this call (128-cout wave tiles are drained in
// two calls of 64): tiles A0 .. A0 + WN/16 - 1 of acc[TNT][TM] are used.
// PREFETCH: issue the global reads of phase 2 for all passes at once (needs 3 x 4 x NP free VGPRs after the staging; the
// A-stationary kernel, which sits at the register cap, turns it off and loads inside the passes).
// SPLIT: split-plane ("f16x3", pp_conv_args_t.split) operands: every fp16 operand read or written here is a hi plane plus a lo
// plane `*_lo` elements further along the pixel row; value = (float)hi + (float)lo, stores write hi = fp16(v), lo = fp16(v - hi)
// (22 significand bits).  fp32 outputs (out_f16 == 0) stay plain.  The phase-2 reads are prefetched in two halves (the
// register budget of the six planes of h / z / addend is the plain path's).
template <int WM, int WN, int TNT = WN / 16, int A0 = 0, bool PREFETCH = true, bool ALLOW_LATE = true, bool SPLIT = false, typename RowMap>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc_)[TNT][WM / 16], char* wave_lds, int lane,
                                              int co_wave /* first cout of the wave tile within the group */, int g,
                                              char* outp, const RowMap rowmap, const f32x4* bias_pre = nullptr,
                                              unsigned long long* stamp = nullptr /* diagnostic: cycle counter after phase 1 */,
                                              const bool preadd_done = false /* the caller already folded preadd into the accumulators (measured slower in the halo kernel: 8-byte accumulator-layout reads; kept for callers that have the addend in registers) */) {
  constexpr int TM = WM / 16, TN = WN / 16;
  typedef _Float16 T;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool has_res = p.residual != nullptr;
  // "late" path (partial-sum pre-add and / or fused GRU gating): phase 1 only adds the bias, the activation and the gate
  // arithmetic run in phase 2 on 8 consecutive couts per lane, where preadd / h / z are read with 16-byte loads
  const bool pre_late = ALLOW_LATE && p.preadd != nullptr && !preadd_done;     // the addend still has to be read (before the activation)
  const bool late = ALLOW_LATE && (pre_late || p.fuse != PP_FUSE_NONE);       // (ALLOW_LATE false: the caller's dispatch excludes such layers)
  const bool stage16 = !SPLIT && p.out_f16 && !has_res && !late;
  constexpr int LPR = WN / 8;                       // lanes per pixel row (8 couts each)
  constexpr int RPP = 64 / LPR;                     // pixel rows per pass
  constexpr int NP = WM / RPP;                      // passes of phase 2
  const int cl = (lane % LPR) * 8;
  const int co = co_wave + cl;
  const int nval = min(8, p.cout_g - co);
  const int out_cbase = p.out_choff + g * p.out_cgroup;
  const int res_cbase = p.res_choff + g * p.out_cgroup;
  // ---- phase 1: bias, scale, activation in registers (each lane: 4 consecutive couts of 16 pixel rows per tile)
  {
    const float scale = p.out_scale;
    const int act = pre_late ? PP_ACT_NONE : p.act;       // with the addend outstanding the activation moves to phase 2
    const float slope = act == PP_ACT_NONE ? 1.f : (act == PP_ACT_LRELU ? p.act_param : 0.f);
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      const int c0 = co_wave + a * 16 + l4 * 4;
      if (bias_pre != nullptr) {                      // fetched by the caller ahead of time (zeros where there is no bias)
        b4 = bias_pre[a];
      } else if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = c0 + r < p.cout_g ? p.bias[g * p.cout_g + c0 + r] : 0.f;
      }
      if (act < PP_ACT_SIGMOID) {
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = (acc_[A0 + a][b][r] + b4[r]) * scale;
            acc_[A0 + a][b][r] = v > 0.f ? v : v * slope;
          }
      } else if (act == PP_ACT_SIGMOID) {             // 1 / (1 + e^-v): v_exp + v_rcp (1 ulp; the result is rounded to fp16)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc_[A0 + a][b][r] = __builtin_amdgcn_rcpf(1.f + __expf(-(acc_[A0 + a][b][r] + b4[r]) * scale));
      } else if (act == PP_ACT_TANH) {                // 1 - 2 / (e^2v + 1); saturates correctly at +-inf
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc_[A0 + a][b][r] = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * (acc_[A0 + a][b][r] + b4[r]) * scale) + 1.f);
      } else {
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_[A0 + a][b][r] = apply_act_special((acc_[A0 + a][b][r] + b4[r]) * scale, act);
      }
    }
  }
  if (stamp != nullptr) { asm volatile("s_nop 0" ::: "memory"); *stamp = __builtin_readcyclecounter(); }
  if (!has_res && !late && p.act2 == PP_ACT_RELU) {          // act2 is defined as "after the residual add"; without a residual it still applies
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc_[A0 + a][b][r] = SPLIT ? relu_split(acc_[A0 + a][b][r]) : fmaxf(acc_[A0 + a][b][r], 0.f);
  }
  if (stage16) {
    // ---- fp16 staging: row stride WN*2 + 16 bytes
    constexpr int LD = WN * 2 + 16;
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        f16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (_Float16)acc_[A0 + a][b][r];
        *reinterpret_cast<f16x4*>(wave_lds + (b * 16 + l15) * LD + (a * 16 + l4 * 4) * 2) = h;
      }
    const bool vec_ok = ((p.out_cstride | out_cbase) & 7) == 0;
    _Float16* ob = reinterpret_cast<_Float16*>(outp) + out_cbase + co;
#pragma unroll 4
    for (int pass = 0; pass < WM / RPP; ++pass) {
      const int prow = pass * RPP + lane / LPR;
      const long long m = rowmap(prow);
      if (m < 0 || nval <= 0) continue;
      const u32x4 raw = *reinterpret_cast<const u32x4*>(wave_lds + prow * LD + cl * 2);
      _Float16* op = ob + m * p.out_cstride;
      if (nval == 8 && vec_ok) *reinterpret_cast<u32x4*>(op) = raw;
      else {
        const _Float16* h = reinterpret_cast<const _Float16*>(&raw);
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r < nval) op[r] = h[r];
      }
    }
    return;
  }
  if (p.epi_direct && !p.out_f16 && !has_res && !late) {
    // ---- plain fp32 output: accumulator layout straight to memory.  acc[a][b] of a lane = couts co_wave + a*16 + l4*4 .. +3 of pixel row
    // b*16 + l15: one 16-byte store per tile, the four l4 groups of a pixel row cover 64 contiguous bytes, the TN tiles the row's WN couts
    float* ob = reinterpret_cast<float*>(outp) + out_cbase;
    const bool vec_ok = ((p.out_cstride | out_cbase) & 3) == 0 && ((unsigned long long)outp & 15) == 0;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const long long m = rowmap(b * 16 + l15);
      if (m < 0) continue;
      float* orow = ob + m * p.out_cstride;
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int c0 = co_wave + a * 16 + l4 * 4;
        const int nv = p.cout_g - c0;
        if (nv >= 4 && vec_ok) *reinterpret_cast<f32x4*>(orow + c0) = acc_[A0 + a][b];
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nv) orow[c0 + r] = acc_[A0 + a][b][r];
        }
      }
    }
    return;
  }
  // ---- fp32 staging (residual and / or fp32 output): row stride WN + 4 floats
  constexpr int LDF = WN + 4;
  float* et = reinterpret_cast<float*>(wave_lds);
#pragma unroll
  for (int b = 0; b < TM; ++b)
#pragma unroll
    for (int a = 0; a < TN; ++a) *reinterpret_cast<f32x4*>(et + (b * 16 + l15) * LDF + a * 16 + l4 * 4) = acc_[A0 + a][b];
  // ---- every global READ of phase 2 (pre-activation addend or residual, GRU state h, gate z) is issued here for ALL passes,
  // back to back, right after the accumulators have been staged (their registers are free now): one exposed memory
  // latency per wave tile instead of one per pass (measured: the per-pass dependent loads made the fused GRU
  // convolutions ~2x slower than their K loop alone)
  const bool pre_vec = pre_late && nval == 8 && ((p.preadd_cstride | p.preadd_choff | (SPLIT ? p.preadd_lo : 0)) & 7) == 0;
  const bool res_vec = has_res && !late && nval == 8 && ((p.res_cstride | res_cbase | (SPLIT ? p.res_lo : 0)) & 7) == 0;
  const bool zr_r = p.fuse == PP_FUSE_GRU_ZR && co >= p.fuse_split;
  const bool gh = p.fuse == PP_FUSE_GRU_H;
  const bool om_flow = p.fuse == PP_FUSE_DCN_OFFMASK && p.fuse_a != nullptr && co < p.fuse_split;   // offsets: + flow (x, y) of the pixel
  constexpr int NPL = SPLIT ? 2 : 1;                // planes per fp16 operand
  constexpr int NQ = PREFETCH ? (SPLIT && NP >= 2 ? NP / 2 : NP) : 1;     // passes whose global reads are in flight together
  static_assert(NP % NQ == 0 && NQ >= 1, "prefetch rounds");
  u32x4 q0[NQ][NPL], q1[NQ][NPL], q2[NQ][NPL];     // [preadd | residual], h, z
  const int pre_lo = SPLIT ? p.preadd_lo : 0, res_lo = SPLIT ? p.res_lo : 0, fa_lo = SPLIT ? p.fuse_a_lo : 0, fb_lo = SPLIT ? p.fuse_b_lo : 0;
  const bool relu2 = p.act2 == PP_ACT_RELU;
  auto unpack8 = [](const u32x4 (&raw)[NPL], float* f) {
    const _Float16* hh = reinterpret_cast<const _Float16*>(&raw[0]);
#pragma unroll
    for (int r = 0; r < 8; ++r) f[r] = (float)hh[r];
    if constexpr (SPLIT) {
      const _Float16* ll = reinterpret_cast<const _Float16*>(&raw[NPL - 1]);
#pragma unroll
      for (int r = 0; r < 8; ++r) f[r] += (float)ll[r];
    }
  };
  auto fetch = [&](int qi, long long m) {            // every global READ of phase 2 for the pixel row m (16-byte loads)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if (pre_vec) q0[qi][pl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.preadd) + m * p.preadd_cstride + p.preadd_choff + co + pl * pre_lo);
      if (res_vec) q0[qi][pl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.residual) + m * p.res_cstride + res_cbase + co + pl * res_lo);
      if (zr_r) q1[qi][pl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.fuse_a) + m * p.fuse_a_cstride + p.fuse_a_choff + co - p.fuse_split + pl * fa_lo);
      if (gh) {
        q1[qi][pl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.fuse_a) + m * p.fuse_a_cstride + p.fuse_a_choff + co + pl * fa_lo);
        q2[qi][pl] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.fuse_b) + m * p.fuse_b_cstride + p.fuse_b_choff + co + pl * fb_lo);
      }
    }
    if (om_flow) q1[qi][0][0] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const T*>(p.fuse_a) + m * p.fuse_a_cstride + p.fuse_a_choff);
  };
#pragma unroll
  for (int rnd = 0; rnd < NP / NQ; ++rnd) {
    if (PREFETCH && (late || has_res)) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        const long long m = rowmap((rnd * NQ + qi) * RPP + lane / LPR);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) q0[qi][pl] = q1[qi][pl] = q2[qi][pl] = u32x4{0, 0, 0, 0};
        if (m < 0 || nval <= 0) continue;
        fetch(qi, m);
      }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const int pass = rnd * NQ + qi;
      const int prow = pass * RPP + lane / LPR;
      const long long m = rowmap(prow);
      if (m < 0 || nval <= 0) continue;
      if constexpr (!PREFETCH) fetch(0, m);
      float v[8];
      const f32x4 lo = *reinterpret_cast<const f32x4*>(et + prow * LDF + cl);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(et + prow * LDF + cl + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
      if (late) {
        if (pre_late) {
          float pv[8];
          const T* pp_ = reinterpret_cast<const T*>(p.preadd) + m * p.preadd_cstride + p.preadd_choff + co;
          if (pre_vec) unpack8(q0[qi], pv);
          else {
#pragma unroll
            for (int r = 0; r < 8; ++r) pv[r] = r < nval ? to_f32(pp_[r]) + (SPLIT ? to_f32(pp_[r + pre_lo]) : 0.f) : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += pv[r];
          const float slope2 = p.act == PP_ACT_NONE ? 1.f : (p.act == PP_ACT_LRELU ? p.act_param : 0.f);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = act_late(v[r], p.act, slope2);
        }
        if (p.fuse == PP_FUSE_GRU_ZR) {
          if (co >= p.fuse_split) {                       // r half: r * h -> out2 (8-cout chunks never straddle the split)
            const int cr = co - p.fuse_split;
            float hv[8];
            unpack8(q1[qi], hv);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= hv[r];
            T* o2 = reinterpret_cast<T*>(p.out2) + m * p.out2_cstride + p.out2_choff + cr;
            if constexpr (SPLIT) store8_split(o2, o2 + p.out2_lo, v);
            else store8<T>(o2, v);
            continue;
          }
        } else if (p.fuse == PP_FUSE_GRU_H) {
          float hv[8], zv[8];
          unpack8(q1[qi], hv);
          unpack8(q2[qi], zv);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = (1.f - zv[r]) * hv[r] + zv[r] * v[r];
        } else if (p.fuse == PP_FUSE_DCN_OFFMASK) {       // dcn_offmask_act_kernel's formulas (token_ops.hip) on the unrounded sums
          if (co < p.fuse_split) {
            float fx = 0.f, fy = 0.f;
            if (om_flow) {
              const _Float16* fl = reinterpret_cast<const _Float16*>(&q1[qi][0]);
              fx = (float)fl[0]; fy = (float)fl[1];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)       // tanh as 1 - 2 / (e^2v + 1) (v_exp + v_rcp, as PP_ACT_TANH above; the result is rounded to fp16)
              v[r] = p.act_param * (1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v[r]) + 1.f)) + ((r & 1) ? fx : fy);
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = __builtin_amdgcn_rcpf(1.f + __expf(-v[r]));
          }
        }
        if (!has_res && relu2) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = SPLIT ? relu_split(v[r]) : fmaxf(v[r], 0.f);
        }
      }
      if (has_res) {
        const T* rp = reinterpret_cast<const T*>(p.residual) + m * p.res_cstride + res_cbase + co;
        if (res_vec) {
          float rv[8];
          unpack8(q0[qi], rv);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += rv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (r < nval) v[r] += to_f32(rp[r]) + (SPLIT ? to_f32(rp[r + res_lo]) : 0.f);
        }
        if (relu2) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = SPLIT ? relu_split(v[r]) : fmaxf(v[r], 0.f);
        }
      }
      const long long oidx = m * p.out_cstride + out_cbase + co;
      if (p.out_f16) {
        _Float16* op = reinterpret_cast<_Float16*>(outp) + oidx;
        if constexpr (SPLIT) {
          if (nval == 8 && ((p.out_cstride | out_cbase | p.out_lo) & 7) == 0) store8_split(op, op + p.out_lo, v);
          else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
              if (r < nval) {
                const _Float16 h = (_Float16)v[r];
                op[r] = h;
                op[r + p.out_lo] = (_Float16)(v[r] - (float)h);
              }
          }
        } else if (nval == 8 && ((p.out_cstride | out_cbase) & 7) == 0) store8<_Float16>(op, v);
        else {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (r < nval) op[r] = (_Float16)v[r];
        }
      } else {
        float* op = reinterpret_cast<float*>(outp) + oidx;
        if (nval == 8 && ((p.out_cstride | out_cbase) & 3) == 0) store8<float>(op, v);
        else {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (r < nval) op[r] = v[r];
        }
      }
    }
  }
}

}  // namespace pp
