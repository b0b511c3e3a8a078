// Scratch experiment (GPU box): what HBM write rate does the projection's store pattern get, versus how many channel
// rows a block writes and how long the contiguous piece per row is?  out is [G][C][P] floats (P = 10596: PEMSD7's N*T),
// every element written exactly once per launch; buffers rotate through 4 copies (1.2 GB) to defeat the 256 MB cache.
// Build: hipcc -O3 --offload-arch=gfx950 tools/write_patterns.hip -o build/write_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// block = 256 lanes; covers ROWS channel rows x PIECE4 float4 of positions; wave-instruction = 64 lanes x 16 B laid out
// as (64 / LPR) rows x LPR float4 (LPR = lanes per row piece: 16 = the projection's 4 rows x 256 B, 64 = 1 row x 1 KiB)
template <int LPR>
__global__ void k_rows(float4* __restrict__ out, int C, int P4, int rows_per_block, int piece4) {
  const int g = blockIdx.z;
  const int c0 = blockIdx.y * rows_per_block;
  const int p0 = blockIdx.x * piece4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rsub = lane / LPR, lcol = lane % LPR;
  constexpr int RPI = 64 / LPR;  // rows per wave-instruction
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
  // the block's (rows x piece) tile as a list of (row group of RPI rows, column group of LPR float4); waves take them round-robin
  const int ncg = (piece4 + LPR - 1) / LPR, nrg = (rows_per_block + RPI - 1) / RPI;
  for (int i = wave; i < ncg * nrg; i += 4) {
    const int rg = i / ncg, cg = i - rg * ncg;   // column groups of one row group are consecutive: adjacent pieces by adjacent waves
    const int c = c0 + rg * RPI + rsub, p4 = p0 + cg * LPR + lcol;
    if (c < C && c < c0 + rows_per_block && p4 < P4 && p4 < p0 + piece4) out[((size_t)g * C + c) * P4 + p4] = v;
  }
}

__global__ void k_fill(float4* __restrict__ out, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}

int main() {
  const int G = 96, C = 72, P4 = 2649;
  const size_t n4 = (size_t)G * C * P4;
  std::vector<float4*> bufs(4);
  for (auto& b : bufs) CK(hipMalloc(&b, n4 * sizeof(float4)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 4; ++i) launch(bufs[i % 4]);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch(bufs[i % 4]);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, n4 * 16.0 / (ms / reps * 1e-3) / 1e12);
  };
  timeit("contiguous fill, 4096 blocks", [&](float4* o) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, o, n4); });
  struct Cfg { int rows, piece4, lpr; };
  const Cfg cfgs[] = {{72, 64, 16}, {72, 64, 64}, {72, 256, 64}, {72, 2649, 64}, {16, 64, 16}, {16, 256, 64}, {16, 2649, 64},
                      {8, 2649, 64}, {4, 2649, 64}, {1, 2649, 64}, {24, 64, 16}, {24, 256, 64}};
  for (const Cfg& c : cfgs) {
    char name[128];
    snprintf(name, sizeof name, "%2d rows x %4d float4 per block, %s per wave-instruction", c.rows, c.piece4,
             c.lpr == 16 ? "4 rows x 256 B" : "1 row x 1 KiB");
    dim3 grid((P4 + c.piece4 - 1) / c.piece4, (C + c.rows - 1) / c.rows, G);
    if (c.lpr == 16) timeit(name, [&](float4* o) { hipLaunchKernelGGL(k_rows<16>, grid, dim3(256), 0, 0, o, C, P4, c.rows, c.piece4); });
    else timeit(name, [&](float4* o) { hipLaunchKernelGGL(k_rows<64>, grid, dim3(256), 0, 0, o, C, P4, c.rows, c.piece4); });
  }
  return 0;
}
