// Library-level entry points: version and thread-local error string.
#include "common.h"
#include <stdarg.h>

namespace pp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pp

extern "C" int pp_version(void) { return 100; }  // 0.1.0
extern "C" const char* pp_last_error_string(void) { return pp::g_err; }

// ABI self-description so that foreign-language bindings can verify their struct layouts.
extern "C" int pp_sizeof_conv_args(void) { return (int)sizeof(pp_conv_args_t); }
extern "C" int pp_sizeof_attn_args(void) { return (int)sizeof(pp_attn_args_t); }

// ------------------------------------------------------------------------------------------------------------------
// Host-side weight packer: reference layout [cout, cin_g, kh, kw] (fp32) -> the [groups][cout_pad][K] layout pp_conv2d
// reads, K ordered by the SAME table pp_conv_build_ktable produced (the table is the single source of truth: chunk kc
// = 8 consecutive channels starting at `choff` of source `src`, seen through tap `tap`).  What propainter_amd/conv.py
// (pack_weight) does in numpy; here for binders that do not run Python.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int pp_conv_pack_weight(const float* weight, int cout, int kh, int kw, int nsrc, const int32_t* src_channels,
                                   int groups, const int32_t* ktable, int kchunks, int dtype, void* out, int64_t out_elems) {
  using namespace pp;
  PP_REQUIRE(weight && src_channels && ktable && cout > 0 && kh > 0 && kw > 0 && nsrc > 0 && nsrc <= PP_CONV_MAX_SRC && groups > 0 &&
                 kchunks > 0 && cout % groups == 0,
             PP_ERR_ARG, "pp_conv_pack_weight: bad arguments");
  PP_REQUIRE(dtype == PP_F32 || dtype == PP_F16, PP_ERR_DTYPE, "pp_conv_pack_weight: dtype %d", dtype);
  int cin_g = 0, base[PP_CONV_MAX_SRC];
  for (int s = 0; s < nsrc; ++s) {
    PP_REQUIRE(src_channels[s] > 0, PP_ERR_ARG, "pp_conv_pack_weight: source %d has %d channels", s, src_channels[s]);
    base[s] = cin_g;
    cin_g += src_channels[s];
  }
  const int cout_g = cout / groups;
  const int cout_pad = (cout_g + 15) / 16 * 16;
  const int64_t K = (int64_t)kchunks * 8;
  const int64_t need = (int64_t)groups * cout_pad * K;
  if (out == nullptr) return cout_pad;
  PP_REQUIRE(out_elems >= need, PP_ERR_WORKSPACE, "pp_conv_pack_weight: need %lld elements, got %lld", (long long)need, (long long)out_elems);
  for (int g = 0; g < groups; ++g)
    for (int co = 0; co < cout_pad; ++co) {
      const float* wrow = co < cout_g ? weight + (int64_t)(g * cout_g + co) * cin_g * kh * kw : nullptr;
      for (int kc = 0; kc < kchunks; ++kc) {
        const int32_t* e = ktable + 4 * kc;
        const int src = e[2] & 0xff, tap = (e[2] >> 16) & 0xff, choff = e[3];
        for (int j = 0; j < 8; ++j) {
          float v = 0.f;
          if (wrow != nullptr && src != 255 && src < nsrc && tap < kh * kw && choff + j < src_channels[src])
            v = wrow[((int64_t)(base[src] + choff + j) * kh + tap / kw) * kw + tap % kw];
          const int64_t o = ((int64_t)g * cout_pad + co) * K + kc * 8 + j;
          if (dtype == PP_F16) reinterpret_cast<_Float16*>(out)[o] = (_Float16)v;
          else reinterpret_cast<float*>(out)[o] = v;
        }
      }
    }
  return cout_pad;
}

// ------------------------------------------------------------------------------------------------------------------
// [diagnosis, round 6] Explicit cache maintenance as a kernel of its own: every XCD of the chip has its own L2, which is not
// coherent with the others' for ordinary device memory -- visibility between kernels is the command processor's job (release =
// write-back at the end of the producer, acquire = invalidate before the consumer).  This kernel does both by hand, one wave
// per block, enough blocks that every XCD runs some (blocks are dealt to the XCDs round-robin):
//   mode & 1: buffer_wbl2 sc0 sc1 -- write the XCD's dirty L2 lines back (what a producer's release does);
//   mode & 2: buffer_inv  sc0 sc1 -- drop the XCD's non-coherent lines (what a consumer's acquire does).
// sharding.StreamingClipGraph places it on the cross-branch edges of the stage-pipelined single graph under PP_SG_FENCE=1 to test
// whether the run-to-run deviation of that form is a visibility defect of cross-queue edges (profiles/r6_graph_queues.txt).
// Not part of include/propainter_hip.h: no product path calls it.
// ------------------------------------------------------------------------------------------------------------------
namespace pp {
__global__ __launch_bounds__(64) void cache_fence_kernel(int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (mode & 1) asm volatile("buffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
  if (mode & 2) asm volatile("buffer_inv sc0 sc1" ::: "memory");
#endif
}
}  // namespace pp

extern "C" int pp_debug_cache_fence(int mode, void* stream) {
  hipLaunchKernelGGL(pp::cache_fence_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, mode);
  return pp::launch_status("pp_debug_cache_fence");
}
