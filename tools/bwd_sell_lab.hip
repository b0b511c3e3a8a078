// DIAGNOSTIC ONLY -- never part of libmsgat_hip.so.  Timing model of the N = 8192 backward as ONE staged column for du AND
// dE (round-4 review, item 4): block = (group on XCD g % 8, part q of Q), NW waves, every wave one PAIR of slices of the CSC
// SELL layout (graph->sell_cols) with its neighbour indices and its per-edge dE sums in registers across all Cu x T/4
// passes (k_sddmm_sellreg's plan).  A pass: stage column j of dz[g,c] (every block of the group stages it), then per trip
// one coalesced 16-B coefficient load (L2), four LDS gathers, du += Ec dz[row_k] and dE_k += dz[row_k] . u[m].
//   lab 0  du stored straight from the gather (64 lanes x 16 B to 64 scattered rows: the unpartitioned layout's cost)
//   lab 1  du not stored (what everything else costs)
//   lab 2  du through the column's LDS space, then N / Q CONSECUTIVE rows stored per block: the store pattern of a
//          row-partitioned SELL build (values of rows the block does not own are stale: TIMING ONLY)
// dE comes out in sell_cols position order.  Built by `python -m ms_gat_amd.build --lab`; `tools/stress_kernels.py --fused NW`.
#include "sell.hpp"

namespace msgat {

constexpr int kLabTrips = 12;

template <int T4, int NW, int LAB>
__global__ __launch_bounds__(64 * NW) void k_bwd_sell_fused_lab(
    const int* __restrict__ slice_off, const int* __restrict__ lane_row, const uint16_t* __restrict__ sidx,
    const float4* __restrict__ dz4, const float4* __restrict__ u4, const float* __restrict__ Es,
    float4* __restrict__ du4, float* __restrict__ dE, int G, int Cu, int N, int n_pos, int n_slices, int Q) {
  extern __shared__ float4 slab[];  // [N]
  constexpr int kThreads = 64 * NW;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = slot % Q;
  const int g = (slot / Q) * 8 + xcd;
  if (g >= G) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int W = q * NW + wave;
  const int sA = W, sB = n_slices - 1 - W;
  const bool hasA = sA < n_slices && sA <= sB, hasB = hasA && sB > sA;
  const int sAc = hasA ? sA : 0, sBc = hasB ? sB : 0;
  const int offA = slice_off[sAc], offB = slice_off[sBc];
  const int ntA = hasA ? (slice_off[sAc + 1] - offA) >> 8 : 0;
  const int ntB = hasB ? (slice_off[sBc + 1] - offB) >> 8 : 0;
  const int nt = ntA + ntB;
  const int rowA = max(lane_row[64 * sAc + lane], 0), rowB = max(lane_row[64 * sBc + lane], 0);
  const uint2* piA = reinterpret_cast<const uint2*>(sidx + offA) + lane;
  const uint2* piB = reinterpret_cast<const uint2*>(sidx + offB) + lane;
  const float4* peA = reinterpret_cast<const float4*>(Es + (size_t)g * n_pos + offA) + lane;
  const float4* peB = reinterpret_cast<const float4*>(Es + (size_t)g * n_pos + offB) + lane;
  uint2 ids[kLabTrips];
  float4 acc[kLabTrips];
#pragma unroll
  for (int t = 0; t < kLabTrips; ++t) {
    const int tt = min(t, max(nt - 1, 0));
    const uint2* p = (tt < ntA) ? piA + 64 * tt : piB + 64 * (tt - ntA);
    ids[t] = (nt > 0) ? *p : make_uint2(0u, 0u);
    acc[t] = f4zero();
  }
  const int rp = (N + Q - 1) / Q;   // lab 2: rows a block stores per pass
  for (int pass = 0; pass < Cu * T4; ++pass) {
    const int c = pass / T4, jr = pass - c * T4;
    const int j = (jr + q) % T4;
    const size_t base = ((size_t)g * Cu + c) * N * T4;
    const float4 uA = u4[base + (size_t)rowA * T4 + j];
    const float4 uB = u4[base + (size_t)rowB * T4 + j];
    for (int n0 = 0; n0 < N; n0 += 4 * kThreads) {   // stage column j of dz[g,c]
      float4 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = dz4[base + (size_t)min(n0 + i * kThreads + (int)threadIdx.x, N - 1) * T4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[min(n0 + i * kThreads + (int)threadIdx.x, N - 1)] = t[i];
    }
    __syncthreads();
    float4 duA = f4zero(), duB = f4zero();
    float4 e_cur = (nt > 0) ? ((0 < ntA) ? peA[0] : peB[0]) : f4zero();
#pragma unroll
    for (int t = 0; t < kLabTrips; ++t) {
      if (t < nt) {  // wave-uniform
        const int tn = min(t + 1, nt - 1);
        const float4 e_nxt = (tn < ntA) ? peA[64 * tn] : peB[64 * (tn - ntA)];   // next trip's coefficients in flight
        const bool inA = t < ntA;
        const float4 uo = make_float4(inA ? uA.x : uB.x, inA ? uA.y : uB.y, inA ? uA.z : uB.z, inA ? uA.w : uB.w);
        const int4 id = sell_unpack(ids[t]);
        const float4 r0 = slab[id.x], r1 = slab[id.y], r2 = slab[id.z], r3 = slab[id.w];
        float4 d = f4zero();
        f4fma(e_cur.x, r0, d); f4fma(e_cur.y, r1, d); f4fma(e_cur.z, r2, d); f4fma(e_cur.w, r3, d);
        if (inA) { duA.x += d.x; duA.y += d.y; duA.z += d.z; duA.w += d.w; }
        else { duB.x += d.x; duB.y += d.y; duB.z += d.z; duB.w += d.w; }
        acc[t].x = f4dot(uo, r0, acc[t].x);
        acc[t].y = f4dot(uo, r1, acc[t].y);
        acc[t].z = f4dot(uo, r2, acc[t].z);
        acc[t].w = f4dot(uo, r3, acc[t].w);
        e_cur = e_nxt;
      }
    }
    __syncthreads();
    if (LAB == 0) {
      if (hasA && lane_row[64 * sAc + lane] >= 0) du4[base + (size_t)rowA * T4 + j] = duA;
      if (hasB && lane_row[64 * sBc + lane] >= 0) du4[base + (size_t)rowB * T4 + j] = duB;
    } else if (LAB == 1) {
      if (duA.x + duB.x == 12345.678f) du4[base + j] = duA;
    } else {
      if (hasA) slab[rowA] = duA;
      if (hasB) slab[rowB] = duB;
      __syncthreads();
      for (int n = q * rp + (int)threadIdx.x; n < min((q + 1) * rp, N); n += kThreads) du4[base + (size_t)n * T4 + j] = slab[n];
      __syncthreads();
    }
  }
  float* out = dE + (size_t)g * n_pos;
  float4* poA = reinterpret_cast<float4*>(out + offA) + lane;
  float4* poB = reinterpret_cast<float4*>(out + offB) + lane;
#pragma unroll
  for (int t = 0; t < kLabTrips; ++t)
    if (t < nt) *((t < ntA) ? poA + 64 * t : poB + 64 * (t - ntA)) = acc[t];
}

}  // namespace msgat

// T = 12 only.  Es: coefficients in the position order of graph->sell_cols; dE: [G, n_pos] in that order.
extern "C" int msgat_lab_bwd_sell_fused(const msgat_shape_t* sh, const msgat_graph_t* gr, int32_t Cu, const float* dz,
                                        const float* u, const float* Es, float* du, float* dE, int32_t nw, int32_t lab,
                                        void* stream) {
  using namespace msgat;
  if (!sh || !gr || !dz || !u || !Es || !du || !dE || sh->T != 12 || gr->sell_cols.n_slices == 0) return MSGAT_ERR_UNSUPPORTED;
  const msgat_sell_t& sl = gr->sell_cols;
  if (sl.pair_trips > kLabTrips) return MSGAT_ERR_UNSUPPORTED;
  const int G = sh->R * sh->Bg, N = sh->N;
  const size_t lds = (size_t)N * sizeof(float4);
#define MSGAT_LABF(NW, L)                                                                                              \
  do {                                                                                                                 \
    const int Q = cdiv(sl.n_slices, 2 * NW);                                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bwd_sell_fused_lab<3, NW, L>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((k_bwd_sell_fused_lab<3, NW, L>), dim3((unsigned)cdiv(G, 8) * 8 * Q), dim3(64 * NW), lds,       \
                       (hipStream_t)stream, sl.slice_off, sl.lane_row, sl.idx, (const float4*)dz, (const float4*)u, Es, \
                       (float4*)du, dE, G, Cu, N, sl.n_pos, sl.n_slices, Q);                                           \
  } while (0)
#define MSGAT_LABF2(NW)                                                              \
  switch (lab) {                                                                     \
    case 0: MSGAT_LABF(NW, 0); break;                                                \
    case 1: MSGAT_LABF(NW, 1); break;                                                \
    case 2: MSGAT_LABF(NW, 2); break;                                                \
    default: return MSGAT_ERR_UNSUPPORTED;                                           \
  }
  if (nw == 16) { MSGAT_LABF2(16) } else if (nw == 12) { MSGAT_LABF2(12) } else if (nw == 8) { MSGAT_LABF2(8) } else return MSGAT_ERR_UNSUPPORTED;
#undef MSGAT_LABF2
#undef MSGAT_LABF
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}
