// Scratch experiment (GPU box): sustained issue rate of v_fma_f32 / v_pk_fma_f32 / v_exp_f32 per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k_rate(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float b = seed * 0.5f, c = seed * 0.25f;
  const v2f bb = {b, b}, cc = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
        a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        p0 = __builtin_elementwise_fma(p0, bb, cc); p1 = __builtin_elementwise_fma(p1, bb, cc);
        p2 = __builtin_elementwise_fma(p2, bb, cc); p3 = __builtin_elementwise_fma(p3, bb, cc);
        p4 = __builtin_elementwise_fma(p4, bb, cc); p5 = __builtin_elementwise_fma(p5, bb, cc);
        p6 = __builtin_elementwise_fma(p6, bb, cc); p7 = __builtin_elementwise_fma(p7, bb, cc);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
        a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
      }
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
  if (r == 1234.5f) out[0] = r;
}

int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {
    for (int mode = 0; mode < 3; ++mode) {
      dim3 grid(256 * wps), block(256);  // wps waves per SIMD on 256 CUs
      auto launch = [&] {
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, grid, block, 0, 0, out, iters, 1.0001f);
        else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, grid, block, 0, 0, out, iters, 1.0001f);
        else hipLaunchKernelGGL(k_rate<2>, grid, block, 0, 0, out, iters, 1.0001f);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * 64 * wps;  // wave-instructions per SIMD
      const char* nm[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32"};
      printf("%-13s waves/SIMD %d: %7.3f ms -> %.2f ns per wave-instr per SIMD = %.2f cycles @2.4GHz; %.1f TFLOP/s\n", nm[mode], wps, ms,
             ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4,
             mode == 2 ? 0.0 : (double)iters * 64 * wps * 1024 * 64 * (mode == 1 ? 4 : 2) / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
