// Device helpers of the kernels that walk the SELL-64 edge layout (msgat_sell_t): shared by aggregate.hip and by the
// diagnostic build of tools/agg_sell_lab.hip.
#pragma once
#include "common.hpp"

namespace msgat {

constexpr int kAggBlock = 1024;
constexpr int kSD = 4;  // trips (of 4 edges per row) in flight per wave
constexpr int kSellMaxSlices = 10;  // slices per wave: ceil(ceil(10176 / 64) / 16), 10176 = the most nodes whose float4 column fits LDS

struct SellTrip {
  uint2 id;  // 4 neighbour indices, 16 bits each
  float4 e;
};

// Requests trip min(t, ntrip-1) of a slice: unconditional, so hipcc keeps counted vmcnt waits; a trip index past
// the slice re-reads its last trip (the lines are in L2) and is simply not consumed.  pi2 / pe4 point at this
// lane's entry of trip 0.
__device__ __forceinline__ void sell_issue(const uint2* __restrict__ pi2, const float4* pe4, int t, int ntrip,
                                           SellTrip& x) {
  const int tc = min(t, ntrip - 1);
  x.id = pi2[64 * tc];
  x.e = pe4[64 * tc];
}
__device__ __forceinline__ int4 sell_unpack(uint2 v) {
  return make_int4((int)(v.x & 0xffffu), (int)(v.x >> 16), (int)(v.y & 0xffffu), (int)(v.y >> 16));
}

__device__ __forceinline__ void sell_gather(const SellTrip& x, const float4* slab, float4& acc) {
  const int4 id = sell_unpack(x.id);
  f4fma(x.e.x, slab[id.x], acc);
  f4fma(x.e.y, slab[id.y], acc);
  f4fma(x.e.z, slab[id.z], acc);
  f4fma(x.e.w, slab[id.w], acc);
}

// stage column j of the [N][T4] slab at `src`: NB loads in flight per lane; lanes past N re-write entry N-1 with
// the value they re-read from it (no branch, so the loads stay in flight together)
template <int T4, int NB = 8>
__device__ __forceinline__ void stage_column(float4* slab, const float4* __restrict__ src, int j, int N) {
  for (int n0 = 0; n0 < N; n0 += NB * kAggBlock) {
    float4 t[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) t[i] = src[(size_t)min(n0 + i * kAggBlock + (int)threadIdx.x, N - 1) * T4 + j];
#pragma unroll
    for (int i = 0; i < NB; ++i) slab[min(n0 + i * kAggBlock + (int)threadIdx.x, N - 1)] = t[i];
  }
}


}  // namespace msgat
