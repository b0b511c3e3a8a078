// Scratch experiment (GPU box): how fast can the aggregate kernel's skeleton move data?  Each variant copies
// 2304 slabs of [883,12] fp32 (98 MB in, 98 MB out) through LDS the way k_agg_lds stages them, without the gather.
// Build: hipcc -O3 --offload-arch=gfx950 tools/slab_copy.hip -o build/slab_copy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one block = one slab, staged through LDS, barrier, streamed out
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_slab(const float4* __restrict__ in, float4* __restrict__ out, int n4) {
  extern __shared__ float4 slab[];
  const size_t base = (size_t)blockIdx.x * n4;
  for (int i = threadIdx.x; i < n4; i += THREADS) slab[i] = in[base + i];
  __syncthreads();
  for (int i = threadIdx.x; i < n4; i += THREADS) out[base + i] = slab[(i * 7 + 3) % n4 == -1 ? 0 : i];
}

// one block = one 4-timestep column of a slab (16-B pieces, 48-B stride); XCD-contiguous block map
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_col(const float4* __restrict__ in, float4* __restrict__ out, int N, int T4,
                                                 int nwork, int xcd_map) {
  extern __shared__ float4 slab[];
  int work = blockIdx.x;
  if (xcd_map) {
    const int per = (nwork + 7) >> 3;
    work = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (work >= nwork) return;
  }
  const int sl = work / T4, j = work - sl * T4;
  const size_t base = (size_t)sl * N * T4 + j;
  for (int n = threadIdx.x; n < N; n += THREADS) slab[n] = in[base + (size_t)n * T4];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += THREADS) out[base + (size_t)n * T4] = slab[n];
}

// the aggregate's phases added one at a time: LEVEL 0 = staged copy, 1 = + row extents, 2 = + 4-edge windows
// (index + weight), 3 = + the LDS gathers and FMAs (the full kernel, lane = (node, 4 timesteps))
struct __attribute__((packed, aligned(4))) i4u { int v[4]; };
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
template <int LEVEL>
__global__ __launch_bounds__(1024) void k_phase(const float4* __restrict__ in, float4* __restrict__ out, int N, int T4,
                                                const int* __restrict__ ptr, const int* __restrict__ idx,
                                                const float* __restrict__ E, int nnz, int slabs_per_g) {
  extern __shared__ float4 slab[];
  const int n4 = N * T4;
  const size_t base = (size_t)blockIdx.x * n4;
  const float* Eg = E + (size_t)(blockIdx.x / slabs_per_g) * nnz;
  for (int i = threadIdx.x; i < n4; i += 1024) slab[i] = in[base + i];
  __syncthreads();
  for (int s = threadIdx.x; s < n4; s += 1024) {
    const int n = s / T4, j = s - n * T4;
    float4 acc = slab[s];
    if (LEVEL >= 1) {
      const int e0 = ptr[n], e1 = ptr[n + 1];
      if (LEVEL == 1) acc.x += (float)(e1 - e0);
      if (LEVEL >= 2) {
        const int b = max(min(e0, nnz - 4), 0);
        const i4u m = *reinterpret_cast<const i4u*>(idx + b);
        const f4u w = *reinterpret_cast<const f4u*>(Eg + b);
        if (LEVEL == 2) acc.x += w.v[0] + w.v[1] + w.v[2] + w.v[3] + (float)(m.v[0] + m.v[1] + m.v[2] + m.v[3]);
        if (LEVEL >= 3) {
          acc = make_float4(0, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float wk = (b + i >= e0 && b + i < e1) ? w.v[i] : 0.f;
            const float4 r = slab[m.v[i] * T4 + j];
            acc.x += wk * r.x; acc.y += wk * r.y; acc.z += wk * r.z; acc.w += wk * r.w;
          }
        }
      }
    }
    out[base + s] = acc;
  }
}

// plain grid-stride copy
__global__ __launch_bounds__(256) void k_plain(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

template <class F>
static float time_us(F f, int reps = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const int N = 883, T4 = 3, n4 = N * T4, slabs = 96 * 24;
  const size_t tot4 = (size_t)slabs * n4;
  float4 *in, *out;
  CK(hipMalloc(&in, tot4 * 16)); CK(hipMalloc(&out, tot4 * 16));
  CK(hipMemset(in, 1, tot4 * 16));
  if (getenv("RANDOM_DATA")) {   // is the rate data dependent?  (constant bytes vs random floats)
    std::vector<float> h(tot4 * 4);
    unsigned st = 777;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) / 16777216.f - 0.5f; }
    CK(hipMemcpy(in, h.data(), tot4 * 16, hipMemcpyHostToDevice));
    printf("random input data\n");
  }
  const double mb = 2.0 * tot4 * 16 / 1e6;
  auto report = [&](const char* name, float us) { printf("%-44s %8.1f us  %7.1f GB/s\n", name, us, mb / us * 1e3 / 1e3); };
  CK(hipFuncSetAttribute((const void*)k_slab<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  for (int pad : {0, 22 * 1024, 38 * 1024}) {   // extra LDS caps blocks per CU: 3, 2, 1 ... for 1024 lanes: 2, 2, 1
    char nm[96];
    snprintf(nm, sizeof nm, "slab via LDS, 1024 lanes, lds %d KB", (n4 * 16 + pad) / 1024);
    report(nm, time_us([&] { hipLaunchKernelGGL(k_slab<1024>, dim3(slabs), dim3(1024), n4 * 16 + pad, 0, in, out, n4); }));
    snprintf(nm, sizeof nm, "slab via LDS,  512 lanes, lds %d KB", (n4 * 16 + pad) / 1024);
    report(nm, time_us([&] { hipLaunchKernelGGL(k_slab<512>, dim3(slabs), dim3(512), n4 * 16 + pad, 0, in, out, n4); }));
    snprintf(nm, sizeof nm, "slab via LDS,  256 lanes, lds %d KB", (n4 * 16 + pad) / 1024);
    report(nm, time_us([&] { hipLaunchKernelGGL(k_slab<256>, dim3(slabs), dim3(256), n4 * 16 + pad, 0, in, out, n4); }));
  }
  {  // random sorted CSR: self loop + ~2 random neighbours per node
    std::vector<int> hp(N + 1), hi;
    unsigned st = 12345;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (int n = 0; n < N; ++n) {
      hp[n] = (int)hi.size();
      std::vector<int> nb = {n};
      const int d = 1 + rnd() % 3;
      for (int k = 0; k < d; ++k) nb.push_back(rnd() % N);
      std::sort(nb.begin(), nb.end());
      nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
      for (int v : nb) hi.push_back(v);
    }
    hp[N] = (int)hi.size();
    const int nnz = hp[N];
    int *dp, *di; float* dE;
    CK(hipMalloc(&dp, (N + 1) * 4)); CK(hipMalloc(&di, nnz * 4)); CK(hipMalloc(&dE, (size_t)96 * nnz * 4));
    CK(hipMemcpy(dp, hp.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, hi.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dE, 0, (size_t)96 * nnz * 4));
    printf("graph: N %d nnz %d\n", N, nnz);
    report("phase 0: staged copy", time_us([&] { hipLaunchKernelGGL(k_phase<0>, dim3(slabs), dim3(1024), n4 * 16, 0, in, out, N, T4, dp, di, dE, nnz, 24); }));
    report("phase 1: + row extents", time_us([&] { hipLaunchKernelGGL(k_phase<1>, dim3(slabs), dim3(1024), n4 * 16, 0, in, out, N, T4, dp, di, dE, nnz, 24); }));
    report("phase 2: + edge windows", time_us([&] { hipLaunchKernelGGL(k_phase<2>, dim3(slabs), dim3(1024), n4 * 16, 0, in, out, N, T4, dp, di, dE, nnz, 24); }));
    report("phase 3: + LDS gathers", time_us([&] { hipLaunchKernelGGL(k_phase<3>, dim3(slabs), dim3(1024), n4 * 16, 0, in, out, N, T4, dp, di, dE, nnz, 24); }));
  }
  const int nwork = slabs * T4;
  for (int xm : {0, 1}) {
    char nm[96];
    snprintf(nm, sizeof nm, "column via LDS, 256 lanes, xcd map %d", xm);
    report(nm, time_us([&] { hipLaunchKernelGGL(k_col<256>, dim3(8 * ((nwork + 7) / 8)), dim3(256), N * 16, 0, in, out, N, T4, nwork, xm); }));
    snprintf(nm, sizeof nm, "column via LDS, 512 lanes, xcd map %d", xm);
    report(nm, time_us([&] { hipLaunchKernelGGL(k_col<512>, dim3(8 * ((nwork + 7) / 8)), dim3(512), N * 16, 0, in, out, N, T4, nwork, xm); }));
  }
  for (int blocks : {1024, 2048, 4096, 8192})  {
    char nm[96];
    snprintf(nm, sizeof nm, "plain grid-stride copy, %d blocks", blocks);
    report(nm, time_us([&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, 0, in, out, tot4); }));
  }
  return 0;
}
