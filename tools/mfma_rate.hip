// Scratch experiment (GPU box): sustained issue rate of v_mfma_f32_16x16x4_f32 per SIMD -- independent
// accumulators (pure issue rate) and a dependent chain (latency), for 1..8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void k_mfma(float* out, int iters, float seed) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = {seed, seed, seed, seed};
  const float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CHAINS], 0, 0, 0);
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 1234.5f) out[0] = r;
}

// does VALU work overlap with the matrix pipe?  per trip: 8 independent MFMAs + NV independent v_fma / v_exp
template <int NV, bool EXP>
__global__ void k_mix(float* out, int iters, float seed) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = {seed, seed, seed, seed};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  const float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f, c = seed * 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV / 8; ++k) v[(u + k) & 7] = EXP ? __builtin_amdgcn_exp2f(v[(u + k) & 7]) : fmaf(v[(u + k) & 7], b, c);
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  if (r == 1234.5f) out[0] = r;
}

int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int wps : {1, 2, 4, 8})
    for (int chains : {8, 4, 1}) {
      dim3 grid(256 * wps), block(256);
      auto launch = [&] {
        if (chains == 8) hipLaunchKernelGGL(k_mfma<8>, grid, block, 0, 0, out, iters, 1.0001f);
        else if (chains == 4) hipLaunchKernelGGL(k_mfma<4>, grid, block, 0, 0, out, iters, 1.0001f);
        else hipLaunchKernelGGL(k_mfma<1>, grid, block, 0, 0, out, iters, 1.0001f);
      };
      launch(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)iters * 8 * wps;  // MFMAs per SIMD
      printf("16x16x4 f32, %d independent chain(s), %d waves/SIMD: %6.2f cycles per MFMA per SIMD @2.4GHz, %6.1f TFLOP/s\n", chains, wps,
             ms * 1e6 / n * 2.4, n * 1024 * 2048 / (ms * 1e-3) / 1e12);
    }
  for (int wps : {1, 2, 4})
    for (int mode = 0; mode < 4; ++mode) {
      dim3 grid(256 * wps), block(256);
      auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL((k_mix<32, false>), grid, block, 0, 0, out, iters, 1.0001f);
        else if (mode == 1) hipLaunchKernelGGL((k_mix<64, false>), grid, block, 0, 0, out, iters, 1.0001f);
        else if (mode == 2) hipLaunchKernelGGL((k_mix<16, true>), grid, block, 0, 0, out, iters, 1.0001f);
        else hipLaunchKernelGGL((k_mix<32, true>), grid, block, 0, 0, out, iters, 1.0001f);
      };
      launch(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const char* nm[] = {"8 MFMA + 32 v_fma", "8 MFMA + 64 v_fma", "8 MFMA + 16 v_exp", "8 MFMA + 32 v_exp"};
      const double valu[] = {32 * 2.3, 64 * 2.3, 16 * 8.3, 32 * 8.3};
      printf("%s per trip, %d waves/SIMD: %6.1f cycles per trip per wave-slot (MFMA alone 256, VALU alone %.0f)\n", nm[mode], wps,
             ms * 1e6 / ((double)iters * wps) * 2.4, valu[mode]);
    }
  return 0;
}
