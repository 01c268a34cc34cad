// MFMA-only power/clock probe: sustained TFLOP/s of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 on zero vs random
// register operands (no memory traffic at all).  Tuning tool: hipcc --offload-arch=gfx950 -O3 tools/mfma_power.cpp -o build/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const f16x8* in, float* out, int iters, unsigned long long* stamps) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) % 65536]; b[i] = in[(tid * 8 + 4 + i) % 65536]; }
  if constexpr (SHAPE == 16) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[tid] = s;
    const unsigned long long t1 = __builtin_readcyclecounter();      // (after the accumulators were consumed: the MFMAs have retired)
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stamps[0], t1 - t0); atomicAdd(&stamps[1], 1ull); atomicMin(&stamps[2], t0); atomicMax(&stamps[3], t1); }
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i * 2 + kk], b[j * 2 + kk], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[tid] = s;
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stamps[0], t1 - t0); atomicAdd(&stamps[1], 1ull); atomicMin(&stamps[2], t0); atomicMax(&stamps[3], t1); }
  }
}

int main() {
  f16x8* in; float* out; unsigned long long* stamps;
  hipMalloc(&in, 65536 * 16); hipMalloc(&out, 1 << 24); hipMalloc(&stamps, 32);
  _Float16* h = (_Float16*)malloc(65536 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int fill = 0; fill < 3; ++fill) {
    for (int i = 0; i < 65536 * 8; ++i) h[i] = fill == 0 ? (_Float16)0.f : fill == 1 ? (_Float16)((rand() % 2001 - 1000) * 1e-3f) : (_Float16)((rand() % 1000) * 1e-3f);
    hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    for (int shape = 16; shape <= 32; shape += 16) {
      const int iters = 20000;
      for (int wps = 1; wps <= 2; ++wps) {           // waves per SIMD: 1 or 2 blocks of 4 waves per CU
      const int blocks = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        const unsigned long long init[4] = {0, 0, ~0ull, 0};
        hipMemcpy(stamps, init, 32, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters, stamps);
        else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters, stamps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[4];
        hipMemcpy(st, stamps, 32, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * (shape == 16 ? 16 * 16384.0 : 8 * 32768.0);
        const double per_wave = (double)st[0] / (double)st[1], n_mfma = (double)iters * (shape == 16 ? 16 : 8);
        // s_memtime cycles the SIMD spends per MFMA (a wave's loop time / its MFMAs / the waves sharing the SIMD), and the stamp clock
        if (rep) printf("fill %s  mfma %dx%d  %d wave(s)/SIMD: %.1f ms  %.0f TFLOP/s | %.2f stamp cycles per MFMA and SIMD (nominal %d) | stamp clock %.3f GHz\n",
                        fill == 0 ? "zero  " : fill == 1 ? "random" : "pos   ", shape, shape, wps, ms, flops / ms / 1e9, per_wave / n_mfma / wps,
                        shape == 16 ? 16 : 32, (double)(st[3] - st[2]) / (ms * 1e6));
      }
      }
    }
  }
  return 0;
}
