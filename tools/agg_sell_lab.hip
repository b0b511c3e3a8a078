// DIAGNOSTIC ONLY -- never part of libmsgat_hip.so.  k_agg_sell (ms_gat_amd/csrc/aggregate.hip) with ONE phase removed,
// to see what each phase costs at the stress graph (profiles/r02/stress_agg_sell_lab.txt):
//   lab = 0 complete kernel, 1 LDS gathers made conflict-free, 2 no edge loads (synthetic indices),
//         3 no column staging, 4 no result store.
// Built by `python -m ms_gat_amd.build --lab` into build/lab/libmsgat_lab.so (the product sources plus this unit); the
// extra entry point below is not in include/msgat_hip.h.  `python tools/stress_kernels.py --lab N` times it.
#include "sell.hpp"

namespace msgat {

// the same trip without global loads
__device__ __forceinline__ void sell_fake(int t, SellTrip& x) {
  const unsigned l = threadIdx.x & 63;
  x.id = make_uint2(((l * 37 + t * 101) & 8191) | (((l * 53 + t * 211) & 8191) << 16),
                    ((l * 71 + t * 307) & 8191) | (((l * 89 + t * 401) & 8191) << 16));
  x.e = make_float4(1.f, 0.5f, 0.25f, 0.125f);
}

template <int LAB>
__device__ __forceinline__ void lab_gather(const SellTrip& x, const float4* slab, float4& acc) {
  if (LAB == 1) {  // conflict-free LDS reads that still depend on the loaded indices
    const int4 id = sell_unpack(x.id);
    const int l = threadIdx.x & 63;
    f4fma(x.e.x, slab[(id.x & 0) + l], acc);
    f4fma(x.e.y, slab[(id.y & 0) + l + 64], acc);
    f4fma(x.e.z, slab[(id.z & 0) + l + 128], acc);
    f4fma(x.e.w, slab[(id.w & 0) + l + 192], acc);
    return;
  }
  sell_gather(x, slab, acc);
}

template <int T4, int LAB>
__global__ __launch_bounds__(kAggBlock) void k_agg_sell_lab(
    const int* __restrict__ slice_off, const int* __restrict__ lane_row, const uint16_t* __restrict__ sidx,
    const float4* __restrict__ u4, const float* __restrict__ Es, float4* __restrict__ v4, int G, int Cu, int N, int n_pos,
    int n_slices) {
  extern __shared__ float4 slab[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int j = slot % T4;
  const int c = (slot / T4) % Cu;
  const int g = (slot / (T4 * Cu)) * 8 + xcd;
  if (g >= G) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t base = ((size_t)g * Cu + c) * N * T4;
  const float* Eg = Es + (size_t)g * n_pos;
  if (LAB != 3) stage_column<T4>(slab, u4 + base, j, N);
  __syncthreads();
  float4 accs[kSellMaxSlices];
#pragma unroll
  for (int i = 0; i < kSellMaxSlices; ++i) {
    accs[i] = f4zero();
    const int s = wave + i * (kAggBlock / 64);
    if (s < n_slices) {
      const int off = slice_off[s];
      const int ntrip = (slice_off[s + 1] - off) >> 8;
      const uint2* pi4 = reinterpret_cast<const uint2*>(sidx + off) + lane;
      const float4* pe4 = reinterpret_cast<const float4*>(Eg + off) + lane;
      float4 acc = f4zero();
      if (ntrip > 0) {
        SellTrip a, b, c4, d;
        if (LAB == 2) sell_fake(0, a); else sell_issue(pi4, pe4, 0, ntrip, a);
        if (LAB == 2) sell_fake(1, b); else sell_issue(pi4, pe4, 1, ntrip, b);
        if (LAB == 2) sell_fake(2, c4); else sell_issue(pi4, pe4, 2, ntrip, c4);
        if (LAB == 2) sell_fake(3, d); else sell_issue(pi4, pe4, 3, ntrip, d);
        for (int t = 0; t < ntrip; t += kSD) {
          lab_gather<LAB>(a, slab, acc);
          if (LAB == 2) sell_fake(t + 4, a); else sell_issue(pi4, pe4, t + 4, ntrip, a);
          if (t + 1 < ntrip) lab_gather<LAB>(b, slab, acc);
          if (LAB == 2) sell_fake(t + 5, b); else sell_issue(pi4, pe4, t + 5, ntrip, b);
          if (t + 2 < ntrip) lab_gather<LAB>(c4, slab, acc);
          if (LAB == 2) sell_fake(t + 6, c4); else sell_issue(pi4, pe4, t + 6, ntrip, c4);
          if (t + 3 < ntrip) lab_gather<LAB>(d, slab, acc);
          if (LAB == 2) sell_fake(t + 7, d); else sell_issue(pi4, pe4, t + 7, ntrip, d);
        }
      }
      accs[i] = acc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kSellMaxSlices; ++i) {
    const int s = wave + i * (kAggBlock / 64);
    if (s < n_slices) {
      const int row = lane_row[64 * s + lane];
      if (row >= 0) slab[row] = accs[i];
    }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += kAggBlock) {
    const float4 acc = slab[n];
    if (LAB != 4 || acc.x == 12345.678f) v4[base + (size_t)n * T4 + j] = acc;
  }
}

}  // namespace msgat

// T = 12 only; Es = the edge coefficients already in the SELL position order of graph->sell_rows
extern "C" int msgat_lab_aggregate_sell(const msgat_shape_t* sh, const msgat_graph_t* gr, int32_t Cu, const float* u,
                                        const float* Es, float* v, int32_t lab, void* stream) {
  using namespace msgat;
  if (!sh || !gr || !u || !Es || !v || sh->T != 12 || gr->sell_rows.n_slices == 0) return MSGAT_ERR_UNSUPPORTED;
  const msgat_sell_t& sl = gr->sell_rows;
  const int G = sh->R * sh->Bg, N = sh->N;
  const size_t lds = (size_t)N * sizeof(float4);
  const dim3 grid((unsigned)cdiv(G, 8) * 8 * Cu * 3), block(kAggBlock);
#define MSGAT_LAB_RUN(L)                                                                                             \
  do {                                                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_sell_lab<3, L>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    hipLaunchKernelGGL((k_agg_sell_lab<3, L>), grid, block, lds, (hipStream_t)stream, sl.slice_off, sl.lane_row,     \
                       sl.idx, (const float4*)u, Es, (float4*)v, G, Cu, N, sl.n_pos, sl.n_slices);                   \
  } while (0)
  switch (lab) {
    case 0: MSGAT_LAB_RUN(0); break;
    case 1: MSGAT_LAB_RUN(1); break;
    case 2: MSGAT_LAB_RUN(2); break;
    case 3: MSGAT_LAB_RUN(3); break;
    case 4: MSGAT_LAB_RUN(4); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_LAB_RUN
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}
