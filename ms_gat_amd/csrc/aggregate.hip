// The sparse side of the attention: neighbour aggregation over the CSR/CSC edges.
//
//   forward    v[g,c,n,:] = sum_{e in row n} E[g,e] u[g,c,col_e,:]          attention.py:36
//   backward   du[g,c,m,:] = sum_{e into m}  E[g,e] dv[g,c,row_e,:]          (same kernel on the CSC)
//              dE[g,e]     = sum_{c,t} dv[g,c,row_e,t] u[g,c,col_e,t]        (SDDMM)
//
// Layout.  The reference's [B,C,N,T] layout makes one (group, channel) pair a contiguous
// [N,T] fp32 slab of N*48 bytes (42 KB at PEMSD7's N = 883).  A whole slab fits in the
// CU's 160 KB LDS, so a block streams its slab(s) in with perfectly coalesced 16-B loads
// (HBM sees every byte of u exactly once), gathers neighbour rows from LDS, and streams
// the result out coalesced.  Slabs that do not fit (N > ~3400) fall back to a
// gather-from-L2 kernel.  Either way the kernel is HBM-bound: algorithmic bytes =
// read u once + write v once.
#include "common.hpp"

namespace msgat {

// ---- LDS-slab aggregate ---------------------------------------------------------------------
// 1024 lanes per block: the gather phase is a chain of dependent loads (row extent -> edge
// index/weight -> LDS row), so it is latency-, not bandwidth-limited; 16 waves per block and two
// blocks per CU keep the CU's 32 wave slots full while one block streams its slab in or out.
constexpr int kAggBlock = 1024;

// 4 edges per trip, fetched as ONE 16-byte load of indices and one of weights.  Edge ranges start
// at arbitrary dword offsets; gfx950 global loads of 128 bits need only dword alignment, which the
// packed types tell the compiler.  A trip may read past the row's last edge (into the next row's,
// always inside the array: the window is clamped to end at nnz) -- those slots get weight 0.
struct __attribute__((packed, aligned(4))) int4u { int v[4]; };
struct __attribute__((packed, aligned(4))) float4u { float v[4]; };

template <int T4, typename RowPtr>
__device__ __forceinline__ float4 gather_row(const int* __restrict__ idx, const float* __restrict__ Eg,
                                             int e0, int e1, int nnz, RowPtr rows, int j) {
  float4 acc = f4zero();
  if (nnz < 8) {  // tiny graphs: scalar walk
    for (int e = e0; e < e1; ++e) f4fma(Eg[e], rows[idx[e] * T4 + j], acc);
    return acc;
  }
  // Two windows (8 edges) are fetched unconditionally and together.  A per-window loop costs every wave
  // one dependent global round trip per extra window as soon as ONE of its rows is longer than 4 edges
  // -- and with PEMS-like degrees (1 + ~Poisson(2)) that is practically every wave: the loop form ran
  // at 45 us against 33 us for the same kernel on rows of at most 4 edges (tools/slab_copy.hip).
  const int b0 = min(e0, nnz - 8);
  const int4u m0 = *reinterpret_cast<const int4u*>(idx + b0);
  const int4u m1 = *reinterpret_cast<const int4u*>(idx + b0 + 4);
  const float4u w0 = *reinterpret_cast<const float4u*>(Eg + b0);
  const float4u w1 = *reinterpret_cast<const float4u*>(Eg + b0 + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float wa = (b0 + k >= e0 && b0 + k < e1) ? w0.v[k] : 0.f;
    const float wb = (b0 + 4 + k >= e0 && b0 + 4 + k < e1) ? w1.v[k] : 0.f;
    f4fma(wa, rows[m0.v[k] * T4 + j], acc);
    f4fma(wb, rows[m1.v[k] * T4 + j], acc);
  }
  for (int e = b0 + 8; e < e1; e += 4) {  // rows with more than 8 edges
    const int b = min(e, nnz - 4);
    const int4u m = *reinterpret_cast<const int4u*>(idx + b);
    const float4u w = *reinterpret_cast<const float4u*>(Eg + b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float wk = (b + k >= e && b + k < e1) ? w.v[k] : 0.f;
      f4fma(wk, rows[m.v[k] * T4 + j], acc);
    }
  }
  return acc;
}

template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_agg_lds(
    const int* __restrict__ ptr, const int* __restrict__ idx, const float4* __restrict__ u4,
    const float* __restrict__ E, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, float4* __restrict__ v4, int Bg, int Cu, int N, int nnz,
    int CH) {
  extern __shared__ float4 slab[];  // [ch][N][T4]
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int c0 = blockIdx.x * CH;
  const int ch = min(CH, Cu - c0);
  const int NT4 = N * T4;
  const size_t base = ((size_t)g * Cu + c0) * NT4;
  const int total = ch * NT4;

  for (int i = threadIdx.x; i < total; i += kAggBlock) slab[i] = u4[base + i];
  __syncthreads();

  const float* Eg = E + (size_t)g * nnz;
  for (int s = threadIdx.x; s < NT4; s += kAggBlock) {
    const int n = s / T4;
    const int j = s - n * T4;
    const int e0 = ptr[n], e1 = ptr[n + 1];
    float4 ex = f4zero();
    if (addvec != nullptr) ex = extra4[(size_t)g * NT4 + s];
    for (int c = 0; c < ch; ++c) {
      float4 acc = gather_row<T4>(idx, Eg, e0, e1, nnz, slab + c * NT4, j);
      if (addvec != nullptr) f4fma(addvec[r * Cu + c0 + c], ex, acc);
      v4[base + (size_t)c * NT4 + s] = acc;
    }
  }
}

// ---- LDS aggregate for slabs larger than LDS: one 4-timestep column of the slab at a time --------------
// N*T*4 bytes exceed the CU's LDS from N ~ 3400 (T = 12), but one float4 column of the slab (N*16 B)
// fits up to N ~ 10 000 (the N = 8192 stress graph: 128 KB).  The block walks the T/4 columns: stage
// column j (16-B pieces, 4T-byte stride -- every line of the slab is fetched T/4 times, but from
// L2 / infinity cache, within microseconds), gather neighbour rows from LDS, store column j of v.
// Against the gather-from-L2 kernel below this trades ~deg x 128-B line fetches per output row for
// T/4 passes over the slab: 8.3 ms -> 4.4 ms per launch at N = 8192, degree 17, 256 groups x 24 channels
// (requesting several rows' windows at once did not help: the pass is bound by LDS bank conflicts of
// the random 16-B gathers plus the strided edge-window loads, not by latency).
template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_agg_cols(
    const int* __restrict__ ptr, const int* __restrict__ idx, const float4* __restrict__ u4,
    const float* __restrict__ E, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, float4* __restrict__ v4, int Bg, int Cu, int N, int nnz) {
  extern __shared__ float4 slab[];  // [N]: column j of the [N][T4] slab
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int c = blockIdx.x;
  const size_t base = ((size_t)g * Cu + c) * N * T4;
  const float* Eg = E + (size_t)g * nnz;
  const float av = (addvec != nullptr) ? addvec[r * Cu + c] : 0.f;
  for (int j = 0; j < T4; ++j) {
    for (int n = threadIdx.x; n < N; n += kAggBlock) slab[n] = u4[base + (size_t)n * T4 + j];
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += kAggBlock) {
      float4 acc = gather_row<1>(idx, Eg, ptr[n], ptr[n + 1], nnz, slab, 0);
      if (addvec != nullptr) f4fma(av, extra4[((size_t)g * N + n) * T4 + j], acc);
      v4[base + (size_t)n * T4 + j] = acc;
    }
    __syncthreads();
  }
}

// ---- column aggregate on the sliced jagged-diagonal edge layout (msgat_jds_t) --------------------------------
// Same LDS plan as k_agg_cols (one 4-timestep column of the slab per pass) but the edges are read from the
// JDS form: wave w owns slices w, w+16, ...; for jagged column k of its slice lane l reads neighbour index and
// coefficient at colstart[k] + l -- one coalesced 4-B-per-lane load each, no dependent address, kJU columns
// requested per trip and two trips in flight (register double buffer).  Measured at the stress graph
// (N = 8192, degree 17, G = 256, Cu = 24): see profiles/r02/.  The CSR form it replaces spent its time in
// ~4 dependent round trips per row (row extent -> window -> window ...) of 16-B pieces at a ~68-B stride.
constexpr int kJU = MSGAT_JDS_PAD;  // jagged columns per trip; colstart is padded by this many entries

struct JdsTrip {
  int id[kJU];
  float e[kJU];
  int cnt[kJU];  // wave-uniform (SGPRs)
};

// requests one trip: all loads unconditional at clamped addresses (a load inside a branch costs hipcc's
// counted vmcnt waits, see mfma.hip); inactive lanes re-read the column's last entry and are masked at use
__device__ __forceinline__ void jds_issue(const int* __restrict__ colstart, const int* __restrict__ jidx,
                                          const float* Eg, int k, int k1, int lane, int last,
                                          JdsTrip& t) {
  int cs[kJU + 1];
#pragma unroll
  for (int u = 0; u <= kJU; ++u) cs[u] = colstart[k + u];  // padded: always in bounds
#pragma unroll
  for (int u = 0; u < kJU; ++u) {
    t.cnt[u] = (k + u < k1) ? cs[u + 1] - cs[u] : 0;
    const int p = min(cs[u] + min(lane, max(t.cnt[u] - 1, 0)), last);
    t.id[u] = jidx[p];
    t.e[u] = Eg[p];
  }
}

__device__ __forceinline__ float jds_mask(float e, int lane, int cnt) {
  // bitwise, not a select or a multiply: a select lets hipcc sink the load into a branch, a multiply would
  // turn another row's Inf/NaN coefficient into a NaN here
  return __int_as_float(__float_as_int(e) & -(int)(lane < cnt));
}

template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_agg_jds(
    const int* __restrict__ slice, const int* __restrict__ colstart, const int* __restrict__ lane_row,
    const int* __restrict__ jidx, const float4* __restrict__ u4, const float* __restrict__ Ej,
    const float* __restrict__ addvec, const float4* __restrict__ extra4, float4* __restrict__ v4, int Bg,
    int Cu, int N, int nnz, int n_slices) {
  extern __shared__ float4 slab[];  // [N]: column j of the [N][T4] slab
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int c = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t base = ((size_t)g * Cu + c) * N * T4;
  const float* Eg = Ej + (size_t)g * nnz;
  const float av = (addvec != nullptr) ? addvec[r * Cu + c] : 0.f;
  const int last = nnz - 1;
  for (int j = 0; j < T4; ++j) {
    // stage column j: 8 loads in flight per lane, clamped addresses, masked LDS writes
    for (int n0 = 0; n0 < N; n0 += 8 * kAggBlock) {
      float4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = u4[base + (size_t)min(n0 + i * kAggBlock + (int)threadIdx.x, N - 1) * T4 + j];
#pragma unroll
      for (int i = 0; i < 8; ++i)  // lanes past N re-write entry N-1 with the value they re-read from it: no branch,
        slab[min(n0 + i * kAggBlock + (int)threadIdx.x, N - 1)] = t[i];  // so the 8 loads stay in flight together
    }
    __syncthreads();
    for (int s = wave; s < n_slices; s += kAggBlock / 64) {
      const int k0 = slice[s], k1 = slice[s + 1];
      const int row = lane_row[64 * s + lane];
      float4 ex = f4zero();
      if (addvec != nullptr && row >= 0) ex = extra4[((size_t)g * N + row) * T4 + j];
      float4 acc = f4zero();
      JdsTrip a, b;
      jds_issue(colstart, jidx, Eg, k0, k1, lane, last, a);
      for (int k = k0; k < k1; k += 2 * kJU) {
        jds_issue(colstart, jidx, Eg, min(k + kJU, k1), k1, lane, last, b);
#pragma unroll
        for (int u = 0; u < kJU; ++u) f4fma(jds_mask(a.e[u], lane, a.cnt[u]), slab[a.id[u]], acc);
        jds_issue(colstart, jidx, Eg, min(k + 2 * kJU, k1), k1, lane, last, a);
#pragma unroll
        for (int u = 0; u < kJU; ++u) f4fma(jds_mask(b.e[u], lane, b.cnt[u]), slab[b.id[u]], acc);
      }
      if (row >= 0) {
        if (addvec != nullptr) f4fma(av, ex, acc);
        v4[base + (size_t)row * T4 + j] = acc;
      }
    }
    __syncthreads();
  }
}

// ---- gather-from-L2 aggregate (not even one column fits LDS) ------------------------------------------
template <int T4>
__global__ __launch_bounds__(kBlock) void k_agg_glb(
    const int* __restrict__ ptr, const int* __restrict__ idx, const float4* __restrict__ u4,
    const float* __restrict__ E, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, float4* __restrict__ v4, int Bg, int Cu, int N, int nnz) {
  const int g = blockIdx.z;
  const int r = g / Bg;
  const int c = blockIdx.y;
  const int NT4 = N * T4;
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= NT4) return;
  const int n = s / T4;
  const int j = s - n * T4;
  const float4* sl = u4 + ((size_t)g * Cu + c) * NT4;
  float4 acc = gather_row<T4>(idx, E + (size_t)g * nnz, ptr[n], ptr[n + 1], nnz, sl, j);
  if (addvec != nullptr) f4fma(addvec[r * Cu + c], extra4[(size_t)g * NT4 + s], acc);
  v4[((size_t)g * Cu + c) * NT4 + s] = acc;
}

template <int T4>
static int launch_aggregate_t(const int* ptr, const int* idx, int nnz, const msgat_jds_t* jds, const float* u,
                              const float* E, const float* addvec, const float* extra, float* v, int G, int Bg,
                              int Cu, int N, hipStream_t s) {
  const int T = 4 * T4;
  const int CH = slab_channels(N, T, Cu, kLdsBudget);
  if (jds != nullptr) {  // E is in the JDS order of this structure (the caller permuted it by jds->src)
    const size_t lds = (size_t)N * sizeof(float4);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_jds<T4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e;
    }
    hipLaunchKernelGGL(k_agg_jds<T4>, dim3(Cu, G), dim3(kAggBlock), lds, s, jds->slice, jds->colstart,
                       jds->lane_row, jds->idx, (const float4*)u, E, addvec, (const float4*)extra, (float4*)v, Bg,
                       Cu, N, nnz, jds->n_slices);
  } else if (CH >= 1) {
    const size_t lds = (size_t)CH * N * T * sizeof(float);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_lds<T4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e;
    }
    dim3 grid(cdiv(Cu, CH), G);
    hipLaunchKernelGGL(k_agg_lds<T4>, grid, dim3(kAggBlock), lds, s, ptr, idx, (const float4*)u, E, addvec,
                       (const float4*)extra, (float4*)v, Bg, Cu, N, nnz, CH);
  } else if ((size_t)N * sizeof(float4) <= (size_t)kLdsMax - 1024 && nnz >= 8) {
    const size_t lds = (size_t)N * sizeof(float4);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_cols<T4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e;
    }
    hipLaunchKernelGGL(k_agg_cols<T4>, dim3(Cu, G), dim3(kAggBlock), lds, s, ptr, idx, (const float4*)u, E,
                       addvec, (const float4*)extra, (float4*)v, Bg, Cu, N, nnz);
  } else {
    dim3 grid(cdiv(N * T4, kBlock), Cu, G);
    hipLaunchKernelGGL(k_agg_glb<T4>, grid, dim3(kBlock), 0, s, ptr, idx, (const float4*)u, E, addvec,
                       (const float4*)extra, (float4*)v, Bg, Cu, N, nnz);
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_aggregate(const int* ptr, const int* idx, int nnz, const msgat_jds_t* jds, const float* u,
                     const float* E, const float* addvec, const float* extra, float* v, int G, int Bg, int Cu,
                     int N, int T, hipStream_t s) {
  switch (T) {
    case 4: return launch_aggregate_t<1>(ptr, idx, nnz, jds, u, E, addvec, extra, v, G, Bg, Cu, N, s);
    case 8: return launch_aggregate_t<2>(ptr, idx, nnz, jds, u, E, addvec, extra, v, G, Bg, Cu, N, s);
    case 12: return launch_aggregate_t<3>(ptr, idx, nnz, jds, u, E, addvec, extra, v, G, Bg, Cu, N, s);
    case 16: return launch_aggregate_t<4>(ptr, idx, nnz, jds, u, E, addvec, extra, v, G, Bg, Cu, N, s);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// E in CSC order for the transposed aggregate of the backward pass: Ec[g,k] = E[g, cperm[k]]
__global__ __launch_bounds__(kBlock) void k_permute_edges(const float* __restrict__ E,
                                                          const int* __restrict__ cperm,
                                                          float* __restrict__ Ec, int nnz) {
  const int g = blockIdx.y;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < nnz) Ec[(size_t)g * nnz + k] = E[(size_t)g * nnz + cperm[k]];
}

int launch_permute_edges(const float* E, const int* cperm, float* Ec, int G, int nnz, hipStream_t s) {
  if (nnz == 0) return MSGAT_OK;
  dim3 grid(cdiv(nnz, kBlock), G);
  hipLaunchKernelGGL(k_permute_edges, grid, dim3(kBlock), 0, s, E, cperm, Ec, nnz);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- aggregate, then project (C <= Co: the reference's own order, msgat.py:26-28) -------------------
// One lane owns one (node, 4 timesteps) slot: for each input channel it gathers the
// neighbour rows (x is small here -- C is 1 or 3 in the first MEAM -- so it lives in L2),
// optionally stores y for the backward pass, and folds it into OT output channels.
template <int T4, int OT>
__global__ __launch_bounds__(kBlock) void k_agg_proj(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float4* __restrict__ x4,
    const float* __restrict__ E, const float* __restrict__ W, float4* __restrict__ y4,
    float4* __restrict__ z4, int Bg, int C, int Co, int N, int nnz) {
  extern __shared__ float Wl[];  // [C][OT]
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int o0 = blockIdx.z * OT;
  for (int i = threadIdx.x; i < C * OT; i += kBlock) {
    const int c = i / OT, oo = i - c * OT, o = o0 + oo;
    Wl[i] = (o < Co) ? W[((size_t)r * Co + o) * C + c] : 0.f;
  }
  __syncthreads();
  const int NT4 = N * T4;
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= NT4) return;
  const int n = s / T4;
  const int j = s - n * T4;
  const float* Eg = E + (size_t)g * nnz;
  const int e0 = rowptr[n], e1 = rowptr[n + 1];

  float4 acc[OT];
#pragma unroll
  for (int oo = 0; oo < OT; ++oo) acc[oo] = f4zero();
  for (int c = 0; c < C; ++c) {
    const float4* sl = x4 + ((size_t)g * C + c) * NT4;
    float4 y = gather_row<T4>(col, Eg, e0, e1, nnz, sl, j);
    if (y4 != nullptr && blockIdx.z == 0) y4[((size_t)g * C + c) * NT4 + s] = y;
    const float4* wrow = reinterpret_cast<const float4*>(Wl + c * OT);
#pragma unroll
    for (int o4 = 0; o4 < OT / 4; ++o4) {
      const float4 w = wrow[o4];
      f4fma(w.x, y, acc[4 * o4 + 0]);
      f4fma(w.y, y, acc[4 * o4 + 1]);
      f4fma(w.z, y, acc[4 * o4 + 2]);
      f4fma(w.w, y, acc[4 * o4 + 3]);
    }
  }
#pragma unroll
  for (int oo = 0; oo < OT; ++oo) {
    const int o = o0 + oo;
    if (o < Co) z4[((size_t)g * Co + o) * NT4 + s] = acc[oo];
  }
}

template <int T4, int OT>
static int launch_agg_proj_t(const msgat_graph_t& gr, const float* x, const float* E, const float* W,
                             float* y, float* z, int G, int Bg, int C, int Co, int N,
                             hipStream_t s) {
  dim3 grid(cdiv(N * T4, kBlock), G, cdiv(Co, OT));
  const size_t lds = (size_t)C * OT * sizeof(float);
  hipLaunchKernelGGL((k_agg_proj<T4, OT>), grid, dim3(kBlock), lds, s, gr.rowptr, gr.col,
                     (const float4*)x, E, W, (float4*)y, (float4*)z, Bg, C, Co, N, gr.nnz);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

template <int T4>
static int launch_agg_proj_o(const msgat_graph_t& gr, const float* x, const float* E, const float* W,
                             float* y, float* z, int G, int Bg, int C, int Co, int N,
                             hipStream_t s) {
  if (Co % 24 == 0) return launch_agg_proj_t<T4, 24>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
  if (Co % 32 == 0) return launch_agg_proj_t<T4, 32>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
  if (Co % 16 == 0) return launch_agg_proj_t<T4, 16>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
  if (Co <= 4) return launch_agg_proj_t<T4, 4>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
  return launch_agg_proj_t<T4, 8>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
}

int launch_aggregate_project(const msgat_graph_t& gr, const float* x, const float* E,
                             const float* W, float* y, float* z, int G, int Bg, int C, int Co,
                             int N, int T, hipStream_t s) {
  switch (T) {
    case 4: return launch_agg_proj_o<1>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
    case 8: return launch_agg_proj_o<2>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
    case 12: return launch_agg_proj_o<3>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
    case 16: return launch_agg_proj_o<4>(gr, x, E, W, y, z, G, Bg, C, Co, N, s);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// ---- SDDMM: dE partials per channel chunk ---------------------------------------------------------
// One lane per CSR edge (n -> m): <dv[c,n,:], u[c,m,:]> summed over the chunk's channels.
// u rows are the random side (gathered from the LDS slab); dv rows follow the edge order,
// which is row-sorted, so neighbouring lanes read the same or adjacent rows.
template <int T4, bool USE_LDS>
__global__ __launch_bounds__(kBlock) void k_sddmm(
    const int* __restrict__ erow, const int* __restrict__ col, const float4* __restrict__ u4,
    const float4* __restrict__ dv4, float* __restrict__ dEp, int Cu, int N, int nnz, int CH,
    int nchunks) {
  extern __shared__ float4 slab[];
  const int g = blockIdx.z;
  const int k = blockIdx.y;
  const int c0 = k * CH;
  const int ch = min(CH, Cu - c0);
  const int NT4 = N * T4;
  const size_t base = ((size_t)g * Cu + c0) * NT4;
  if (USE_LDS) {
    const int total = ch * NT4;
    for (int i = threadIdx.x; i < total; i += kBlock) slab[i] = u4[base + i];
    __syncthreads();
  }
  float* out = dEp + ((size_t)g * nchunks + k) * nnz;
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < nnz; e += gridDim.x * kBlock) {
    const int n = erow[e], m = col[e];
    float acc = 0.f;
    for (int c = 0; c < ch; ++c) {
      const float4* a = dv4 + base + (size_t)c * NT4 + (size_t)n * T4;
      const float4* b = USE_LDS ? (slab + c * NT4 + m * T4) : (u4 + base + (size_t)c * NT4 + (size_t)m * T4);
#pragma unroll
      for (int j = 0; j < T4; ++j) acc = f4dot(a[j], b[j], acc);
    }
    out[e] = acc;
  }
}

// ---- SDDMM on the JDS layout: one 4-timestep column of u in LDS per pass -------------------------------
// Block (chunk k, group g) walks its channels x T/4 columns; per pass it stages column j of u[c] and every
// lane (= row, through lane_row) dots its dv[c,row,4j..4j+3] against the staged neighbour entries of its
// row's edges.  The per-edge partial lives in dEp (JDS order: lane l of jagged column k owns position
// colstart[k] + l, so the read-modify-write is one coalesced 4-B-per-lane load + store, private to the
// lane: no atomics, fixed (c, j) summation order).
template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_sddmm_jds(
    const int* __restrict__ slice, const int* __restrict__ colstart, const int* __restrict__ lane_row,
    const int* __restrict__ jidx, const float4* __restrict__ u4, const float4* __restrict__ dv4,
    float* __restrict__ dEp, int Cu, int N, int nnz, int n_slices, int CH, int nchunks) {
  extern __shared__ float4 slab[];  // [N]
  const int g = blockIdx.y;
  const int kc = blockIdx.x;
  const int c0 = kc * CH;
  const int ch = min(CH, Cu - c0);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* out = dEp + ((size_t)g * nchunks + kc) * nnz;
  const int last = nnz - 1;
  for (int pass = 0; pass < ch * T4; ++pass) {
    const int c = pass / T4, j = pass - c * T4;
    const size_t base = ((size_t)g * Cu + c0 + c) * N * T4;
    for (int n0 = 0; n0 < N; n0 += 8 * kAggBlock) {
      float4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = u4[base + (size_t)min(n0 + i * kAggBlock + (int)threadIdx.x, N - 1) * T4 + j];
#pragma unroll
      for (int i = 0; i < 8; ++i)  // lanes past N re-write entry N-1 with the value they re-read from it: no branch,
        slab[min(n0 + i * kAggBlock + (int)threadIdx.x, N - 1)] = t[i];  // so the 8 loads stay in flight together
    }
    __syncthreads();
    for (int s = wave; s < n_slices; s += kAggBlock / 64) {
      const int k0 = slice[s], k1 = slice[s + 1];
      const int row = lane_row[64 * s + lane];
      const float4 a = dv4[base + (size_t)max(row, 0) * T4 + j];  // lanes past N hold no edge: never stored
      JdsTrip ta, tb;  // .e carries the running partial of the edge (previous passes)
      jds_issue(colstart, jidx, out, k0, k1, lane, last, ta);
      for (int k = k0; k < k1; k += 2 * kJU) {
        jds_issue(colstart, jidx, out, min(k + kJU, k1), k1, lane, last, tb);
#pragma unroll
        for (int u = 0; u < kJU; ++u)
          if (lane < ta.cnt[u]) {  // a store in a branch costs nothing; the loads above stay unconditional
            const float prev = (pass == 0) ? 0.f : ta.e[u];
            out[colstart[min(k + u, k1)] + lane] = f4dot(a, slab[ta.id[u]], prev);
          }
        jds_issue(colstart, jidx, out, min(k + 2 * kJU, k1), k1, lane, last, ta);
#pragma unroll
        for (int u = 0; u < kJU; ++u)
          if (lane < tb.cnt[u]) {
            const float prev = (pass == 0) ? 0.f : tb.e[u];
            out[colstart[min(k + kJU + u, k1)] + lane] = f4dot(a, slab[tb.id[u]], prev);
          }
      }
    }
    __syncthreads();
  }
}

// channel chunks of the SDDMM (= partial buffers of [G,nnz]): LDS slab form: as many channels as fit the LDS
// budget per chunk; JDS form: enough chunks to give every CU a block, at most one per channel
int sddmm_chunks(int G, int Cu, int N, int T, bool jds) {
  if (jds) {
    const int want = max(1, min(Cu, cdiv(256, max(G, 1))));
    return cdiv(Cu, cdiv(Cu, want));  // every chunk owns at least one channel
  }
  const int CH = slab_channels(N, T, Cu, kLdsBudget);
  return CH >= 1 ? cdiv(Cu, CH) : 1;
}

template <int T4>
static int launch_sddmm_t(const msgat_graph_t& gr, const float* u, const float* dv, float* dEp,
                          int G, int Cu, int N, hipStream_t s) {
  const int T = 4 * T4;
  const int CH = slab_channels(N, T, Cu, kLdsBudget);
  if (jds_usable(gr.jds_rows, gr.nnz, N, T)) {  // partials come out in JDS order (k_edge_grad reads them through pos)
    const msgat_jds_t& jd = gr.jds_rows;
    const int nch = sddmm_chunks(G, Cu, N, T, true);
    const int chj = cdiv(Cu, nch);
    const size_t lds = (size_t)N * sizeof(float4);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sddmm_jds<T4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e;
    }
    hipLaunchKernelGGL(k_sddmm_jds<T4>, dim3(nch, G), dim3(kAggBlock), lds, s, jd.slice, jd.colstart, jd.lane_row,
                       jd.idx, (const float4*)u, (const float4*)dv, dEp, Cu, N, gr.nnz, jd.n_slices, chj, nch);
  } else if (CH >= 1) {
    const size_t lds = (size_t)CH * N * T * sizeof(float);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sddmm<T4, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e;
    }
    const int nchunks = cdiv(Cu, CH);
    dim3 grid(1, nchunks, G);
    hipLaunchKernelGGL((k_sddmm<T4, true>), grid, dim3(kBlock), lds, s, gr.erow, gr.col,
                       (const float4*)u, (const float4*)dv, dEp, Cu, N, gr.nnz, CH, nchunks);
  } else {
    dim3 grid(min(cdiv(gr.nnz, kBlock), 1024), 1, G);
    hipLaunchKernelGGL((k_sddmm<T4, false>), grid, dim3(kBlock), 0, s, gr.erow, gr.col,
                       (const float4*)u, (const float4*)dv, dEp, Cu, N, gr.nnz, Cu, 1);
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_sddmm(const msgat_graph_t& gr, const float* u, const float* dv, float* dEp, int G,
                 int Cu, int N, int T, hipStream_t s) {
  if (gr.nnz == 0) return MSGAT_OK;
  switch (T) {
    case 4: return launch_sddmm_t<1>(gr, u, dv, dEp, G, Cu, N, s);
    case 8: return launch_sddmm_t<2>(gr, u, dv, dEp, G, Cu, N, s);
    case 12: return launch_sddmm_t<3>(gr, u, dv, dEp, G, Cu, N, s);
    case 16: return launch_sddmm_t<4>(gr, u, dv, dEp, G, Cu, N, s);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

}  // namespace msgat
