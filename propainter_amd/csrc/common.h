// Shared helpers for the gfx950 kernels of libpropainter_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/propainter_hip.h"

// s_setprio around an MFMA cluster (cdna_hip_programming.md T5: the CU's issue arbiter prefers the wave that is entering its matrix
// instructions over a co-resident wave that is issuing loads / scalar work).  Build-time switches so that tools/build_variant.sh can A/B
// them; the shipped values follow the measurement (profiles/r5_setprio_ab.txt, interleaved rounds on one box, bit-identical outputs):
//   PP_SETPRIO_SPLIT (tri-product cluster of the split-plane halo kernel, 48 MFMAs):  convc2 -3.5 %, flow head -1.5 %, convm -1.3 %, GRU gate
//                    convolutions -0.8 ... -1.0 %                                                                        -> ON
//   PP_SETPRIO       (fp16 halo clusters of 32 MFMAs, tiled v2 kernel incl. its tri step): -0.5 ... +4 % (worse on the fp16 GRU / convc2
//                    shapes, +1.5 % on the split 1x1)                                                                    -> off
//   PP_SETPRIO_ATTN  (S and PV clusters of the attention kernel): 0.692 vs 0.691 ms alone, but 71.1 -> 68.7 ms per clip inside the pass
//                    (two window lanes overlap)                                                                           -> ON
#ifndef PP_SETPRIO
#define PP_SETPRIO 0
#endif
#ifndef PP_SETPRIO_SPLIT
#define PP_SETPRIO_SPLIT 1
#endif
#if PP_SETPRIO_SPLIT == 2      // (variant: raised from the fragment reads on)
#define PP_SPLIT_PRIO_EARLY() __builtin_amdgcn_s_setprio(1)
#define PP_SPLIT_PRIO_BEGIN() ((void)0)
#define PP_SPLIT_PRIO_END() __builtin_amdgcn_s_setprio(0)
#elif PP_SETPRIO_SPLIT
#define PP_SPLIT_PRIO_EARLY() ((void)0)
#define PP_SPLIT_PRIO_BEGIN() __builtin_amdgcn_s_setprio(1)
#define PP_SPLIT_PRIO_END() __builtin_amdgcn_s_setprio(0)
#else
#define PP_SPLIT_PRIO_EARLY() ((void)0)
#define PP_SPLIT_PRIO_BEGIN() ((void)0)
#define PP_SPLIT_PRIO_END() ((void)0)
#endif
#ifndef PP_SETPRIO_ATTN
#define PP_SETPRIO_ATTN 1      // pass level (profiles/r5_setprio_ab.txt): attention class 71.1 -> 68.7 ms per clip, bit-identical
#endif
#if PP_SETPRIO_ATTN
#define PP_ATTN_PRIO_BEGIN() __builtin_amdgcn_s_setprio(1)
#define PP_ATTN_PRIO_END() __builtin_amdgcn_s_setprio(0)
#else
#define PP_ATTN_PRIO_BEGIN() ((void)0)
#define PP_ATTN_PRIO_END() ((void)0)
#endif
#if PP_SETPRIO
#define PP_MFMA_PRIO_BEGIN() __builtin_amdgcn_s_setprio(1)
#define PP_MFMA_PRIO_END() __builtin_amdgcn_s_setprio(0)
#else
#define PP_MFMA_PRIO_BEGIN() ((void)0)
#define PP_MFMA_PRIO_END() ((void)0)
#endif

namespace pp {

void set_error(const char* fmt, ...);

// Checks the launch and turns a hipError_t into the C-ABI return convention.
inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define PP_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) {                  \
      pp::set_error(__VA_ARGS__);   \
      return (code);                \
    }                               \
  } while (0)

// 16-byte vector of raw bits used for all wide global/LDS moves.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int code = PP_F32; };
template <> struct dtype_of<_Float16> { static constexpr int code = PP_F16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }

// Transcendental activations live in one out-of-line function so that the (fully unrolled) GEMM epilogues stay small.
static __device__ __noinline__ float apply_act_special(float v, int act) {
  switch (act) {
    case PP_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case PP_ACT_TANH: return tanhf(v);
    case PP_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    default: return v;
  }
}

__device__ __forceinline__ float apply_act(float v, int act, float param) {
  // none / relu / leaky-relu share one branch-free form: v > 0 ? v : v * slope
  const float slope = act == PP_ACT_NONE ? 1.f : (act == PP_ACT_LRELU ? param : 0.f);
  if (act >= PP_ACT_SIGMOID) return apply_act_special(v, act);
  return v > 0.f ? v : (act == PP_ACT_RELU ? 0.f : v * slope);
}

// Loads `n` (<= 8) consecutive elements as fp32.
template <typename T> __device__ __forceinline__ void load8(const T* p, float* v) {
  if constexpr (sizeof(T) == 2) {
    u32x4 raw = *reinterpret_cast<const u32x4*>(p);
    const _Float16* h = reinterpret_cast<const _Float16*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
  } else {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
  }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v) {
  if constexpr (sizeof(T) == 2) {
    u32x4 raw;
    _Float16* h = reinterpret_cast<_Float16*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
    *reinterpret_cast<u32x4*>(p) = raw;
  } else {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
}

// ReLU of the split-plane ("f16x3") producers.  fmaxf(NaN, 0) = 0 would ERASE the trace of a value that left fp16's range (hi plane inf,
// lo plane NaN: the pair reads back as NaN): this form propagates NaN (and turns -inf, which only an overflow produces, into NaN), so an
// overflow anywhere in the split-plane engine reaches the flows as NaN and RAFT_bi's finite-flow guard reports it.
__device__ __forceinline__ float relu_split(float v) { return v > 0.f ? v : v * 0.f; }

// Split-plane ("f16x3") store: hi = fp16(v), lo = fp16(v - hi); hi + lo carries 22 significand bits of v (|v| < 65504).
__device__ __forceinline__ void store8_split(_Float16* hi_p, _Float16* lo_p, const float* v) {
  u32x4 rh, rl;
  _Float16* h = reinterpret_cast<_Float16*>(&rh);
  _Float16* l = reinterpret_cast<_Float16*>(&rl);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = (_Float16)v[i];
    l[i] = (_Float16)(v[i] - (float)h[i]);
  }
  *reinterpret_cast<u32x4*>(hi_p) = rh;
  *reinterpret_cast<u32x4*>(lo_p) = rl;
}
// ... and the matching 8-element load: hi + lo as fp32
__device__ __forceinline__ void load8_split(const _Float16* hi_p, const _Float16* lo_p, float* v) {
  const u32x4 rh = *reinterpret_cast<const u32x4*>(hi_p), rl = *reinterpret_cast<const u32x4*>(lo_p);
  const _Float16* h = reinterpret_cast<const _Float16*>(&rh);
  const _Float16* l = reinterpret_cast<const _Float16*>(&rl);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i];
}

// grid_sample(align_corners=True) coordinate round trip exactly as torch computes it in fp32:
// g = 2*c/(size-1) - 1 ; c' = ((g + 1) / 2) * (size - 1).   Keeps 'nearest' ties identical.
__device__ __forceinline__ float grid_roundtrip(float c, int size) {
  float d = (float)(size > 1 ? size - 1 : 1);
  float g = 2.0f * c / d - 1.0f;
  return ((g + 1.0f) * 0.5f) * (float)(size - 1);
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#if defined(__HIP_DEVICE_COMPILE__)
// Buffer descriptor over [base, base + bytes) whose inputs are PROVABLY wave-uniform to the compiler (readfirstlane on
// the pointer halves and the size): without this hipcc wraps every buffer op that uses a descriptor derived from
// block-id arithmetic in a "waterfall" loop (4 x v_readfirstlane + compare + saveexec per memory op), which serialises
// consecutive loads.  The caller guarantees that base / bytes really are the same in every lane of the wave.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_buffer_rsrc(const void* base, int bytes) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int nb = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
#endif

}  // namespace pp
