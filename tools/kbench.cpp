// Stand-alone micro-benchmark of pp_conv2d kernel variants through the C-ABI (no Python / torch: starts in milliseconds
// on a fresh GPU box).  Build: tools/build_kbench.sh -> build/kbench.  TEST / TUNING TOOL, not part of the product.
//
//   kbench conv N H W KH KW COUT SRC[,SRC..] [--impls 0,70,12,1] [--reps 20] [--act A] [--late zr|h|pre] [--res] [--prof]
//
// Random fp16 sources / weights (uniform [-1, 1) scaled by 1/sqrt(K)), every impl is timed with HIP events over `reps`
// launches after 3 warm-up launches and compared with the first impl's output (the kernel families walk K in the same
// order and must agree bit for bit).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>

#include "../include/propainter_hip.h"

extern "C" int pp_debug_conv_prof(unsigned long long* out);

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static bool zero_fill = false;   // --zero: all-zero operands (DVFS probe: same instruction stream, less switching power)
static float urand() {   // [-1, 1)
  if (zero_fill) return 0.f;
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng_state >> 40) & 0xffffff) / 8388608.0f - 1.0f;
}

static void* dev_f16(size_t n, float scale) {
  std::vector<_Float16> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(urand() * scale);
  void* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

static std::vector<int> parse_list(const char* s) {
  std::vector<int> v;
  const char* p = s;
  while (*p) {
    v.push_back(atoi(p));
    const char* c = strchr(p, ',');
    if (!c) break;
    p = c + 1;
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 9 || strcmp(argv[1], "conv") != 0) {
    fprintf(stderr, "usage: kbench conv N H W KH KW COUT SRC[,SRC..] [--impls a,b] [--reps n] [--act a] [--late zr|h|pre] [--res] [--prof]\n");
    return 1;
  }
  const int N = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), KH = atoi(argv[5]), KW = atoi(argv[6]), COUT = atoi(argv[7]);
  std::vector<int> srcs = parse_list(argv[8]);
  std::vector<int> impls = {0};
  int reps = 20, act = PP_ACT_NONE;
  std::string late;
  bool res = false, prof = false;
  int rounds = 1;
  for (int i = 9; i < argc; ++i) {
    if (!strcmp(argv[i], "--impls")) impls = parse_list(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--act")) act = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--late")) late = argv[++i];
    else if (!strcmp(argv[i], "--res")) res = true;
    else if (!strcmp(argv[i], "--prof")) prof = true;
    else if (!strcmp(argv[i], "--rounds")) rounds = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--zero")) zero_fill = true;
  }
  const int nsrc = (int)srcs.size();
  std::vector<int32_t> dy, dx, cpad(nsrc), creal(nsrc);
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) { dy.push_back(ky); dx.push_back(kx); }
  int cin = 0;
  for (int s = 0; s < nsrc; ++s) { creal[s] = srcs[s]; cpad[s] = (srcs[s] + 7) / 8 * 8; cin += srcs[s]; }
  const int kchunks = pp_conv_build_ktable(KH * KW, dy.data(), dx.data(), nsrc, cpad.data(), 0, nullptr, 0);
  if (kchunks < 0) { fprintf(stderr, "ktable: %s\n", pp_last_error_string()); return 2; }
  std::vector<int32_t> kt((size_t)(kchunks + 1) * 4);
  pp_conv_build_ktable(KH * KW, dy.data(), dx.data(), nsrc, cpad.data(), 0, kt.data(), kchunks + 1);
  const int K = kchunks * 8;
  std::vector<float> w((size_t)COUT * cin * KH * KW);
  const float wscale = 1.0f / sqrtf((float)(cin * KH * KW));
  for (auto& v : w) v = urand() * wscale * 1.7f;
  const int cout_pad = pp_conv_pack_weight(w.data(), COUT, KH, KW, nsrc, creal.data(), 1, kt.data(), kchunks, PP_F16, nullptr, 0);
  std::vector<_Float16> wp((size_t)cout_pad * K);
  pp_conv_pack_weight(w.data(), COUT, KH, KW, nsrc, creal.data(), 1, kt.data(), kchunks, PP_F16, wp.data(), (int64_t)wp.size());
  void *d_w, *d_kt, *d_bias;
  CK(hipMalloc(&d_w, wp.size() * 2));
  CK(hipMemcpy(d_w, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_kt, kt.size() * 4));
  CK(hipMemcpy(d_kt, kt.data(), kt.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> bias(COUT);
  for (auto& v : bias) v = urand() * 0.1f;
  CK(hipMalloc(&d_bias, COUT * 4));
  CK(hipMemcpy(d_bias, bias.data(), COUT * 4, hipMemcpyHostToDevice));
  const size_t npix = (size_t)N * H * W;
  const int ocs = (COUT + 7) / 8 * 8;
  std::vector<void*> d_src(nsrc);
  for (int s = 0; s < nsrc; ++s) d_src[s] = dev_f16(npix * cpad[s], 1.0f);
  void* d_res = res ? dev_f16(npix * ocs, 1.0f) : nullptr;
  void* d_pre = !late.empty() ? dev_f16(npix * ocs, 0.5f) : nullptr;
  const int C2 = COUT / 2;
  void* d_h = (late == "zr" || late == "h") ? dev_f16(npix * ocs, 1.0f) : nullptr;
  void* d_z = late == "h" ? dev_f16(npix * ocs, 0.5f) : nullptr;
  void* d_out2 = nullptr;
  if (late == "zr") CK(hipMalloc(&d_out2, npix * ocs * 2));
  std::vector<void*> d_out(impls.size());
  for (auto& o : d_out) { CK(hipMalloc(&o, npix * ocs * 2)); CK(hipMemset(o, 0, npix * ocs * 2)); }

  pp_conv_args_t a;
  memset(&a, 0, sizeof(a));
  a.dtype = PP_F16; a.N = N; a.H = H; a.W = W; a.OH = H; a.OW = W; a.stride_h = a.stride_w = 1;
  a.pad_h = (KH - 1) / 2; a.pad_w = (KW - 1) / 2; a.groups = 1; a.cout_g = COUT; a.cout_pad = cout_pad; a.kchunks = kchunks;
  a.nsrc = nsrc;
  bool u32 = true, u64 = true;
  for (int s = 0; s < nsrc; ++s) {
    a.src[s].ptr = d_src[s]; a.src[s].cstride = cpad[s];
    u32 = u32 && cpad[s] % 32 == 0; u64 = u64 && cpad[s] % 64 == 0;
  }
  a.ktable = (const int32_t*)d_kt; a.weight = d_w; a.weight_gstride = (int64_t)cout_pad * K; a.bias = (const float*)d_bias;
  a.act = act; a.act_param = 0.1f; a.out_scale = 1.f; a.out_dtype = PP_F16; a.out_cstride = ocs; a.out_cgroup = COUT;
  a.ktable_uniform = (u32 ? 4 : 0) | (u64 ? 8 : 0); a.tap_h = KH; a.tap_w = KW;
  if (res) { a.residual = d_res; a.res_cstride = ocs; }
  if (!late.empty()) { a.preadd = d_pre; a.preadd_cstride = ocs; a.bias = nullptr; }
  if (late == "zr") { a.fuse = PP_FUSE_GRU_ZR; a.fuse_split = C2; a.fuse_a = d_h; a.fuse_a_cstride = ocs; a.out2 = d_out2; a.out2_cstride = ocs; }
  if (late == "h") { a.fuse = PP_FUSE_GRU_H; a.fuse_a = d_h; a.fuse_a_cstride = ocs; a.fuse_b = d_z; a.fuse_b_cstride = ocs; }

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flops = 2.0 * (double)npix * COUT * (double)cin * KH * KW;
  printf("conv N%d %dx%d k%dx%d cin", N, H, W, KH, KW);
  for (int s : srcs) printf(" %d", s);
  printf(" cout %d  K %d  %.2f GFLOP  late=%s res=%d act=%d\n", COUT, K, flops / 1e9, late.empty() ? "-" : late.c_str(), (int)res, act);
  std::vector<_Float16> ref(npix * ocs), got(npix * ocs);
  for (int round = 0; round < rounds; ++round)
  for (size_t ii = 0; ii < impls.size(); ++ii) {
    a.impl = impls[ii];
    a.out = d_out[ii];
    int rc = 0;
    for (int i = 0; i < 3 && rc == 0; ++i) rc = pp_conv2d(&a, st);
    if (rc != 0) { printf("  impl %3d: refused (%d) %s\n", impls[ii], rc, pp_last_error_string()); continue; }
    CK(hipStreamSynchronize(st));
    unsigned long long pf[16];
    if (prof) pp_debug_conv_prof(pf);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) pp_conv2d(&a, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    CK(hipMemcpy(ii == 0 ? ref.data() : got.data(), d_out[ii], npix * ocs * 2, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    size_t nbad = 0;
    if (ii > 0)
      for (size_t i = 0; i < ref.size(); ++i) {
        const double d = fabs((double)got[i] - (double)ref[i]);
        if (d > maxd) maxd = d;
        if (d != 0) ++nbad;
      }
    for (size_t i = 0; i < ref.size(); ++i) maxv = fmax(maxv, fabs((double)ref[i]));
    printf("  impl %3d: %9.1f us  %7.1f TFLOP/s", impls[ii], us, flops / us / 1e6);
    if (ii > 0) printf("   vs impl %d: max|d| %.3g  differing %zu / %zu (ref max %.3g)", impls[0], maxd, nbad, ref.size(), maxv);
    printf("\n");
    if (prof) {
      pp_debug_conv_prof(pf);
      // blocks of the halo launch (128-pixel tiles; 128-cout tiles unless the layer has at most 64 couts)
      const double nblk_print = (double)N * (KW == 1 ? ((H + 15) / 16) * ((W + 7) / 8) : ((H + 7) / 8) * ((W + 15) / 16)) * (COUT <= 64 ? 1 : (COUT + 127) / 128);
      if (pf[6]) {
        const double wv = (double)pf[6];
        printf("      prof/wave: vmwait %.0f barrier %.0f issue %.0f compute %.0f | loop %.0f epilogue %.0f (sync %.0f) prologue %.0f | steps %.1f | epi: phase1 %.0f rest %.0f\n",
               pf[0] / wv, pf[1] / wv, pf[2] / wv, pf[3] / wv, pf[4] / wv, pf[5] / wv, pf[8] / wv, pf[11] / wv, pf[7] / wv, pf[9] / wv, pf[10] / wv);
        // stamp clock and average residency: (latest end - earliest start) spans the `reps` launches the events timed
        const double span = (double)(pf[13] - pf[12]), tick_ghz = span / (us * reps) / 1e3;
        const double life = (pf[4] + pf[5] + pf[11]) / wv;       // prologue + loop + epilogue per wave, in ticks
        printf("      stamp clock %.3f GHz | block lifetime %.2f us | blocks %lld -> average resident blocks %.1f\n", tick_ghz, life / tick_ghz / 1e3,
               (long long)nblk_print, nblk_print * (life / tick_ghz / 1e3) / us);
      }
    }
  }
  return 0;
}
