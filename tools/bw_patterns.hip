// Scratch experiment (GPU box): achievable HBM read bandwidth of the projection's access pattern
// -- a block reads PIECE contiguous bytes from each of C channel slabs -- versus piece size and
// loads in flight.  Build: hipcc -O3 --offload-arch=gfx950 tools/bw_patterns.hip -o build/bw_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// block = THREADS lanes; each lane reads float4 at [c][p4] for c = 0..C-1 (UNROLL loads in flight)
template <int UNROLL>
__global__ void k_strided(const float4* __restrict__ x, float* __restrict__ out, int C, int P4, int tiles) {
  const int g = blockIdx.y;
  const int p4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (p4 >= P4) return;
  const float4* src = x + (size_t)g * C * P4 + p4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int c = 0; c < C; c += UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[(size_t)(c + u) * P4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w;
}

// same bytes, but each block reads one contiguous span
template <int UNROLL>
__global__ void k_contig(const float4* __restrict__ x, float* __restrict__ out, size_t n4, int per_block4) {
  const size_t base = (size_t)blockIdx.x * per_block4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < per_block4; i += blockDim.x * UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t idx = base + i + (size_t)u * blockDim.x;
      v[u] = (i + u * (int)blockDim.x < per_block4 && idx < n4) ? x[idx] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w;
}

// occupancy-limited variant: dynamic LDS caps the blocks per CU (256-thread blocks)
template <int UNROLL>
__global__ void k_strided_occ(const float4* __restrict__ x, float* __restrict__ out, int C, int P4) {
  extern __shared__ float pad[];
  const int g = blockIdx.y;
  const int p4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) pad[0] = 0.f;
  if (p4 >= P4) return;
  const float4* src = x + (size_t)g * C * P4 + p4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int c = 0; c < C; c += UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[(size_t)(c + u) * P4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w + pad[0];
}

// chanpair-like gather: a wave reads 16-position (64 B) or 32-position (128 B) pieces of ROWS channel rows
template <int LANES_PER_ROW, int NK>
__global__ void k_rowpieces(const float4* __restrict__ x, float* __restrict__ out, int ROWS, int P4, int span4) {
  extern __shared__ float pad[];
  const int g = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) pad[0] = 0.f;
  const int pbeg = (blockIdx.x * 4 + wave) * span4;
  const int pend = min(P4, pbeg + span4);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int p = pbeg; p < pend; p += LANES_PER_ROW) {
    float4 v[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int L = lane + 64 * k;
      const int row = min(L / LANES_PER_ROW, ROWS - 1);
      const int q = L % LANES_PER_ROW;
      v[k] = x[((size_t)g * ROWS + row) * P4 + min(p + q, P4 - 1)];
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
  }
  if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w + pad[0];
}

int main() {
  const int G = 96, C = 72, N = 883, T = 12, P4 = N * T / 4;
  const size_t n4 = (size_t)G * C * P4;
  float4* x; float* out;
  CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&out, 4));
  CK(hipMemset(x, 0, n4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 100, n4 * 16 / (ms / 10 * 1e-3) / 1e9);
  };
  for (int threads : {64, 128, 256, 512, 1024}) {
    dim3 grid((P4 + threads - 1) / threads, G);
    char name[128];
    snprintf(name, 128, "strided piece=%5dB unroll 4", threads * 16);
    time([&] { hipLaunchKernelGGL(k_strided<4>, grid, dim3(threads), 0, 0, x, out, C, P4, 0); }, name);
    snprintf(name, 128, "strided piece=%5dB unroll 8", threads * 16);
    time([&] { hipLaunchKernelGGL(k_strided<8>, grid, dim3(threads), 0, 0, x, out, C, P4, 0); }, name);
    snprintf(name, 128, "strided piece=%5dB unroll 24", threads * 16);
    time([&] { hipLaunchKernelGGL(k_strided<24>, grid, dim3(threads), 0, 0, x, out, C, P4, 0); }, name);
  }
  for (int lds_kb : {8, 20, 40, 80}) {
    dim3 grid((P4 + 255) / 256, G);
    char name[128];
    hipFuncSetAttribute((const void*)k_strided_occ<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_strided_occ<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    snprintf(name, 128, "strided 4KB pieces, %d KB LDS/blk (occ cap), unroll 8", lds_kb);
    time([&] { hipLaunchKernelGGL(k_strided_occ<8>, grid, dim3(256), lds_kb * 1024, 0, x, out, C, P4); }, name);
    snprintf(name, 128, "strided 4KB pieces, %d KB LDS/blk (occ cap), unroll 4", lds_kb);
    time([&] { hipLaunchKernelGGL(k_strided_occ<4>, grid, dim3(256), lds_kb * 1024, 0, x, out, C, P4); }, name);
  }
  {
    // rows = 72 channels of a group; span 256 positions per wave
    const int span4 = 64;
    dim3 grid((P4 + 4 * span4 - 1) / (4 * span4), G);
    for (int lds_kb : {8, 40, 80}) {
      char name[128];
      hipFuncSetAttribute((const void*)k_rowpieces<4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute((const void*)k_rowpieces<8, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      snprintf(name, 128, "row pieces 64B x72 rows, %d KB LDS/blk", lds_kb);
      time([&] { hipLaunchKernelGGL((k_rowpieces<4, 5>), grid, dim3(256), lds_kb * 1024, 0, x, out, C, P4, span4); }, name);
      snprintf(name, 128, "row pieces 128B x72 rows, %d KB LDS/blk", lds_kb);
      time([&] { hipLaunchKernelGGL((k_rowpieces<8, 9>), grid, dim3(256), lds_kb * 1024, 0, x, out, C, P4, span4); }, name);
      hipFuncSetAttribute((const void*)k_rowpieces<16, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute((const void*)k_rowpieces<32, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute((const void*)k_rowpieces<64, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      snprintf(name, 128, "row pieces 256B x36 rows/pass, %d KB LDS/blk", lds_kb);
      time([&] { hipLaunchKernelGGL((k_rowpieces<16, 18>), grid, dim3(256), lds_kb * 1024, 0, x, out, C, P4, span4); }, name);
      snprintf(name, 128, "row pieces 512B, %d KB LDS/blk", lds_kb);
      time([&] { hipLaunchKernelGGL((k_rowpieces<32, 36>), grid, dim3(256), lds_kb * 1024, 0, x, out, C, P4, span4); }, name);
    }
  }
  for (int per_block4 : {2649}) {
    const int blocks = (int)((n4 + per_block4 - 1) / per_block4);
    char name[128];
    snprintf(name, 128, "contiguous %6d B per block, 256 thr, unroll 4", per_block4 * 16);
    time([&] { hipLaunchKernelGGL(k_contig<4>, dim3(blocks), dim3(256), 0, 0, x, out, n4, per_block4); }, name);
    snprintf(name, 128, "contiguous %6d B per block, 1024 thr, unroll 2", per_block4 * 16);
    time([&] { hipLaunchKernelGGL(k_contig<2>, dim3(blocks), dim3(1024), 0, 0, x, out, n4, per_block4); }, name);
  }
  return 0;
}
