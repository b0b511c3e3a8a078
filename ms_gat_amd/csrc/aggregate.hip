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
#include "sell.hpp"

namespace msgat {

// ---- LDS-slab aggregate ---------------------------------------------------------------------
// 1024 lanes per block: the gather phase is a chain of dependent loads (row extent -> edge
// index/weight -> LDS row), so it is latency-, not bandwidth-limited; 16 waves per block and two
// blocks per CU keep the CU's 32 wave slots full while one block streams its slab in or out.
// kAggBlock (1024 lanes per block): sell.hpp

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

// DOT: the block also leaves dap[g,c] = sum_s <extra[g,s], xdot[g,c,s]> for its channels (ch <= kAggDotMaxC) --
// dalpha = dq . x of the AGG_FIRST / PLAIN backward, whose dq is this kernel's `extra` anyway: one more coalesced
// float4 per lane instead of a contraction launch over dq and x.  Fixed order: lane trips, shuffles, waves.
template <int T4, bool DOT>
__global__ __launch_bounds__(kAggBlock) void k_agg_lds(
    const int* __restrict__ ptr, const int* __restrict__ idx, const float4* __restrict__ u4,
    const float* __restrict__ E, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, float4* __restrict__ v4, int Bg, int Cu, int N, int nnz,
    int CH, const float4* __restrict__ xdot4, float* __restrict__ dap) {
  extern __shared__ float4 slab[];  // [ch][N][T4]
  __shared__ float dred[kAggBlock / 64][kAggDotMaxC];
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int c0 = blockIdx.x * CH;
  const int ch = min(CH, Cu - c0);
  const int NT4 = N * T4;
  const size_t base = ((size_t)g * Cu + c0) * NT4;
  const int total = ch * NT4;

  for (int i = threadIdx.x; i < total; i += kAggBlock) slab[i] = u4[base + i];
  __syncthreads();

  // gridDim.z blocks share a slab's output rows (each stages the whole slab: the re-reads come from L2) -- with one block
  // per (group, chunk) the first MEAM's backward aggregate (C = 1 or 3: G blocks) ran on a third of the CUs
  const int seg = cdiv(NT4, (int)gridDim.z);
  const int s_end = min((int)(blockIdx.z + 1) * seg, NT4);
  float da[kAggDotMaxC] = {0.f, 0.f, 0.f, 0.f};
  const float* Eg = E + (size_t)g * nnz;
  for (int s = blockIdx.z * seg + threadIdx.x; s < s_end; s += kAggBlock) {
    const int n = s / T4;
    const int j = s - n * T4;
    const int e0 = ptr[n], e1 = ptr[n + 1];
    float4 ex = f4zero();
    if (addvec != nullptr) ex = extra4[(size_t)g * NT4 + s];
    if (DOT) {
#pragma unroll
      for (int c = 0; c < kAggDotMaxC; ++c)
        if (c < ch) da[c] = f4dot(ex, xdot4[base + (size_t)c * NT4 + s], da[c]);
    }
    for (int c = 0; c < ch; ++c) {
      float4 acc = gather_row<T4>(idx, Eg, e0, e1, nnz, slab + c * NT4, j);
      if (addvec != nullptr) f4fma(addvec[r * Cu + c0 + c], ex, acc);
      v4[base + (size_t)c * NT4 + s] = acc;
    }
  }
  if (DOT) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < kAggDotMaxC; ++c) {
      float w = da[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
      if (lane == 0) dred[wave][c] = w;
    }
    __syncthreads();
    if (threadIdx.x < ch) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kAggBlock / 64; ++w) t += dred[w][threadIdx.x];
      dap[((size_t)g * gridDim.z + blockIdx.z) * Cu + c0 + threadIdx.x] = t;   // one partial per (group, row split)
    }
  }
}

// ---- LDS aggregate as a persistent ring (small graphs: three [N,T] slabs fit LDS) ---------------------------------
// k_agg_lds gives every slab its own block: stage, barrier, row extents -> edge windows -> gather, store -- a chain of
// dependent round trips per slab that two resident blocks per CU only half hide (3.8 TB/s on cold operands, 72 % of
// what a plain copy gets).  Here ONE block per CU walks a run of consecutive slabs:
//   - slabs land in a ring of three LDS buffers by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, the
//     slab's flat order = the LDS order): while slab t is gathered, slabs t+1 and t+2 are in flight, with no
//     registers and no ds_write pass; counted s_waitcnt vmcnt + one raw barrier per slab;
//   - a run stays within one or two groups, and the edge windows of a lane's slots (8 indices, 8 coefficients, masked
//     once) live in registers for all channels of a group: the per-slab work is LDS gathers and one store per slot;
//   - stores count on vmcnt next to the LDS-DMA loads, in issue order, so the counted waits need every lane to store
//     every time: a slot past the slab's end recomputes the slab's LAST slot (same inputs, same bits) and stores that.
#ifndef MSGAT_RING_AUX
#define MSGAT_RING_AUX 0
#endif
template <int T4, int KI>
__global__ __launch_bounds__(kAggBlock) void k_agg_ring(
    const int* __restrict__ ptr, const int* __restrict__ idx, const float4* __restrict__ u4,
    const float* __restrict__ E, float4* __restrict__ v4, int Cu, int N, int nnz, int nslab, int per) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  extern __shared__ float4 ring[];             // 3 x kBufF4
  constexpr int kWaves = kAggBlock / 64;
  constexpr int kBufF4 = KI * kAggBlock;        // >= N * T4 (host-checked)
  const int NT4 = N * T4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int s0 = blockIdx.x * per, ns = min(per, nslab - s0);
  if (ns <= 0) return;

  auto issue = [&](int t) {   // slab s0 + t (clamped to the run) into buffer t % 3
    const float4* src = u4 + (size_t)(s0 + min(t, ns - 1)) * NT4;
    float4* buf = ring + (t % 3) * kBufF4;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int i0 = (wave + kWaves * k) * 64;   // wave-uniform
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + min(i0 + lane, NT4 - 1)), (lds_ptr_t)(buf + i0), 16, 0, MSGAT_RING_AUX);
    }
  };

  // the first two slabs are requested before anything else: the edge windows below are three dependent round trips
  // that then run under the slabs' flight (hipcc drains vmcnt before the windows' first use -- slab 0 is needed by then)
  issue(0);
  issue(1);

  // this lane's slots: float4 number tid + 1024 k of a slab = (node, 4 timesteps)
  int slot[KI], jof[KI], e0s[KI], e1s[KI], b0s[KI];
  int m[KI][8];
  float w[KI][8];
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    slot[k] = min((int)threadIdx.x + kAggBlock * k, NT4 - 1);
    const int n = slot[k] / T4;
    jof[k] = slot[k] - n * T4;
    e0s[k] = ptr[n];
    e1s[k] = ptr[n + 1];
    b0s[k] = min(e0s[k], nnz - 8);
    const int4u ia = *reinterpret_cast<const int4u*>(idx + b0s[k]), ib = *reinterpret_cast<const int4u*>(idx + b0s[k] + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // rows of edges outside [e0, e1) read row 0 with coefficient 0
      m[k][q] = (b0s[k] + q >= e0s[k] && b0s[k] + q < e1s[k]) ? ia.v[q] * T4 + jof[k] : jof[k];
      m[k][4 + q] = (b0s[k] + 4 + q >= e0s[k] && b0s[k] + 4 + q < e1s[k]) ? ib.v[q] * T4 + jof[k] : jof[k];
    }
  }
  auto load_weights = [&](int g) {
    const float* Eg = E + (size_t)g * nnz;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const float4u wa = *reinterpret_cast<const float4u*>(Eg + b0s[k]), wb = *reinterpret_cast<const float4u*>(Eg + b0s[k] + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w[k][q] = (b0s[k] + q >= e0s[k] && b0s[k] + q < e1s[k]) ? wa.v[q] : 0.f;
        w[k][4 + q] = (b0s[k] + 4 + q >= e0s[k] && b0s[k] + 4 + q < e1s[k]) ? wb.v[q] : 0.f;
      }
    }
  };
  int gcur = s0 / Cu;
  load_weights(gcur);

  for (int t = 0; t < ns; ++t) {
    // slab t has landed: behind its loads the wave issued the stores of slab t-2, the loads of t+1 and the stores of t-1
    if (t == 0) wait_vmcnt<KI>();
    else if (t == 1) wait_vmcnt<2 * KI>();
    else wait_vmcnt<3 * KI>();
    lds_barrier();    // ... everyone's pieces too; and nobody gathers from slab t-1's buffer any more
    issue(t + 2);
    const int g = (s0 + t) / Cu;
    if (g != gcur) {  // block-uniform; the coefficients of the next group (these loads make hipcc drain vmcnt: once per group)
      gcur = g;
      load_weights(g);
    }
    const float4* buf = ring + (t % 3) * kBufF4;
    float4* out = v4 + (size_t)(s0 + t) * NT4;
    const float* Eg = E + (size_t)g * nnz;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      float4 acc = f4zero();
#pragma unroll
      for (int q = 0; q < 8; ++q) f4fma(w[k][q], buf[m[k][q]], acc);
      for (int e = b0s[k] + 8; e < e1s[k]; ++e) f4fma(Eg[e], buf[idx[e] * T4 + jof[k]], acc);   // rows with more than 8 edges
      out[slot[k]] = acc;
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

// ---- column aggregate on the SELL-64 edge layout (msgat_sell_t) ----------------------------------------------
// Same LDS plan as k_agg_cols (one 4-timestep column of the slab in LDS), but the edges come from the degree-sorted
// sliced-ELLPACK form: wave w owns slices w, w+16, ...; edges 4t .. 4t+3 of a slice's 64 rows ("trip" t) sit at
// slice_off + 256 t + 4 lane, so a trip is ONE coalesced 16-B-per-lane load per array (neighbour indices,
// coefficients) with no address arithmetic and no mask (padding carries coefficient 0), several trips in flight.
// Per edge that leaves 4 FMAs, one shift and one LDS read.  The CSR form spent ~16 vector + ~9 scalar
// instructions and ~4 dependent global round trips per row on the same work (profiles/r02/stress_*).
template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_agg_sell(
    const int* __restrict__ slice_off, const int* __restrict__ lane_row, const uint16_t* __restrict__ sidx,
    const float4* __restrict__ u4, const float* __restrict__ Es, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, float4* __restrict__ v4, int G, int Bg, int Cu, int N, int n_pos,
    int n_slices) {
  extern __shared__ float4 slab[];  // [N]: column j of the [N][T4] slab
  // XCD-aware block -> (group, channel, column) map.  Workgroups are dealt round-robin over the 8 XCDs (block b
  // runs on XCD b % 8), each with its own 4 MB L2.  Two things must meet in ONE L2 at the same time:
  //   - the Cu x T/4 blocks of a group all read the same E[g] (0.6 MB at the stress graph): group g is pinned to
  //     XCD g % 8, so an L2 serves one or two groups' coefficients instead of eleven;
  //   - the T/4 column blocks of one (group, channel) slab touch the same 128-B lines (a column is 16 B of every
  //     48-B row) for reading u and for writing v: they take consecutive slots of the XCD, run side by side, and
  //     the fabric sees every line once instead of T/4 times, the partial-line stores merging in L2.
  // Without either the kernel moved 3-5x its algorithmic bytes over the fabric and ran at the same 4.2 ms
  // whatever its instruction count (profiles/r02/stress_*).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int j = slot % T4;
  const int c = (slot / T4) % Cu;
  const int g = (slot / (T4 * Cu)) * 8 + xcd;
  if (g >= G) return;  // grid is padded to a multiple of 8 groups
  const int r = g / Bg;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t base = ((size_t)g * Cu + c) * N * T4;
  const float* Eg = Es + (size_t)g * n_pos;
  const float av = (addvec != nullptr) ? addvec[r * Cu + c] : 0.f;
  stage_column<T4>(slab, u4 + base, j, N);
  __syncthreads();
  // The rows of a slice are scattered over N (the layout is sorted by degree), so results are not stored from
  // here: 64 lanes x 16 B to 64 different 128-B lines per instruction cost 1.07 of the kernel's 2.95 ms (lab
  // variants, profiles/r02/stress_agg_sell_lab.txt).  They wait in registers until every wave is done with the
  // staged column, go back into its LDS space at their row, and leave in row order: consecutive lanes write
  // consecutive rows (and read `extra` the same way).
  float4 accs[kSellMaxSlices];
#pragma unroll
  for (int i = 0; i < kSellMaxSlices; ++i) {
    accs[i] = f4zero();
    const int s = wave + i * (kAggBlock / 64);
    if (s < n_slices) {  // wave-uniform
      const int off = slice_off[s];
      const int ntrip = (slice_off[s + 1] - off) >> 8;  // 256 entries per trip
      const uint2* pi4 = reinterpret_cast<const uint2*>(sidx + off) + lane;
      const float4* pe4 = reinterpret_cast<const float4*>(Eg + off) + lane;
      float4 acc = f4zero();
      if (ntrip > 0) {  // wave-uniform (a slice of rows without edges has no trips)
        // kSD trips in flight, rotating through four register sets; consumption is guarded by wave-uniform
        // branches that contain no load
        SellTrip a, b, c4, d;
        sell_issue(pi4, pe4, 0, ntrip, a);
        sell_issue(pi4, pe4, 1, ntrip, b);
        sell_issue(pi4, pe4, 2, ntrip, c4);
        sell_issue(pi4, pe4, 3, ntrip, d);
        for (int t = 0; t < ntrip; t += kSD) {
          sell_gather(a, slab, acc);
          sell_issue(pi4, pe4, t + 4, ntrip, a);
          if (t + 1 < ntrip) sell_gather(b, slab, acc);
          sell_issue(pi4, pe4, t + 5, ntrip, b);
          if (t + 2 < ntrip) sell_gather(c4, slab, acc);
          sell_issue(pi4, pe4, t + 6, ntrip, c4);
          if (t + 3 < ntrip) sell_gather(d, slab, acc);
          sell_issue(pi4, pe4, t + 7, ntrip, d);
        }
      }
      accs[i] = acc;
    }
  }
  __syncthreads();  // every wave is done reading the staged column
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
    float4 acc = slab[n];
    if (addvec != nullptr) f4fma(av, extra4[((size_t)g * N + n) * T4 + j], acc);
    v4[base + (size_t)n * T4 + j] = acc;
  }
}

// (A half-column form -- 2-timestep columns of 64 KB, two resident blocks per CU, so that one block's staging and stores run
// under the other's gathers -- was measured in round 4 and lost, 2.77 -> 4.00 ms: profiles/r04/stress_agg_sell_lab.txt has
// the numbers and the reading; the kernel is in the history at commit 2d1af2e.)

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
static int launch_aggregate_t(const int* ptr, const int* idx, int nnz, const msgat_sell_t* sell, const float* u,
                              const float* E, const float* addvec, const float* extra, float* v, int G, int Bg,
                              int Cu, int N, hipStream_t s, const float* xdot, float* dap, int* dot_done) {
  const int T = 4 * T4;
  const int CH = slab_channels(N, T, Cu, kLdsBudget);
  if (sell != nullptr) {  // E is in the position order of this layout (the caller permuted it by sell->src)
    const size_t lds = (size_t)N * sizeof(float4);
    static LdsGrant granted;
    if (int st = grant_dynamic_lds(&k_agg_sell<T4>, lds, granted)) return st;
    hipLaunchKernelGGL((k_agg_sell<T4>), dim3((unsigned)cdiv(G, 8) * 8 * Cu * T4), dim3(kAggBlock), lds, s,
                       sell->slice_off, sell->lane_row, sell->idx, (const float4*)u, E, addvec, (const float4*)extra,
                       (float4*)v, G, Bg, Cu, N, sell->n_pos, sell->n_slices);
  } else if (addvec == nullptr && xdot == nullptr && nnz >= 8 && N * T4 <= 3 * kAggBlock && G * Cu >= 512) {
    // plain aggregate on a small graph: the persistent LDS-DMA ring, one block per CU
    const int ncu = device_cu_count();
    const int nslab = G * Cu, per = cdiv(nslab, ncu), nb = cdiv(nslab, per);
    const int KI = cdiv(N * T4, kAggBlock);
    const size_t lds = (size_t)3 * KI * kAggBlock * sizeof(float4);
#define MSGAT_RING(ki)                                                                                            \
    {                                                                                                               \
      static LdsGrant granted;                                                                                      \
      if (int st = grant_dynamic_lds(&k_agg_ring<T4, ki>, lds, granted)) return st;                                 \
      hipLaunchKernelGGL((k_agg_ring<T4, ki>), dim3(nb), dim3(kAggBlock), lds, s, ptr, idx, (const float4*)u, E,    \
                         (float4*)v, Cu, N, nnz, nslab, per);                                                       \
    }
    if (KI == 1) MSGAT_RING(1) else if (KI == 2) MSGAT_RING(2) else MSGAT_RING(3)
#undef MSGAT_RING
  } else if (CH >= 1) {
    const size_t lds = (size_t)CH * N * T * sizeof(float);
    static LdsGrant granted_plain, granted_dot;
    if (int st = grant_dynamic_lds(&k_agg_lds<T4, false>, lds, granted_plain)) return st;
    if (int st = grant_dynamic_lds(&k_agg_lds<T4, true>, lds, granted_dot)) return st;
    // few blocks (the first MEAM's backward: G of them): up to kAggMaxSplit blocks share a slab's rows
    const int nblocks = cdiv(Cu, CH) * G;
    int split = device_cu_count() / max(nblocks, 1);
    split = max(1, min(min(split, kAggMaxSplit), (N * T4) / 256));
    dim3 grid(cdiv(Cu, CH), G, split);
    if (xdot != nullptr && dap != nullptr && addvec != nullptr && Cu <= kAggDotMaxC) {
      hipLaunchKernelGGL((k_agg_lds<T4, true>), grid, dim3(kAggBlock), lds, s, ptr, idx, (const float4*)u, E, addvec,
                         (const float4*)extra, (float4*)v, Bg, Cu, N, nnz, CH, (const float4*)xdot, dap);
      if (dot_done != nullptr) *dot_done = split;   // partials per group
    } else {
      hipLaunchKernelGGL((k_agg_lds<T4, false>), grid, dim3(kAggBlock), lds, s, ptr, idx, (const float4*)u, E, addvec,
                         (const float4*)extra, (float4*)v, Bg, Cu, N, nnz, CH, nullptr, nullptr);
    }
  } else if ((size_t)N * sizeof(float4) <= (size_t)kLdsMax - 1024 && nnz >= 8) {
    const size_t lds = (size_t)N * sizeof(float4);
    static LdsGrant granted;
    if (int st = grant_dynamic_lds(&k_agg_cols<T4>, lds, granted)) return st;
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

int launch_aggregate(const int* ptr, const int* idx, int nnz, const msgat_sell_t* sell, const float* u,
                     const float* E, const float* addvec, const float* extra, float* v, int G, int Bg, int Cu,
                     int N, int T, hipStream_t s, const float* xdot, float* dap, int* dot_done) {
  if (dot_done != nullptr) *dot_done = 0;
  switch (T) {
    case 4: return launch_aggregate_t<1>(ptr, idx, nnz, sell, u, E, addvec, extra, v, G, Bg, Cu, N, s, xdot, dap, dot_done);
    case 8: return launch_aggregate_t<2>(ptr, idx, nnz, sell, u, E, addvec, extra, v, G, Bg, Cu, N, s, xdot, dap, dot_done);
    case 12: return launch_aggregate_t<3>(ptr, idx, nnz, sell, u, E, addvec, extra, v, G, Bg, Cu, N, s, xdot, dap, dot_done);
    case 16: return launch_aggregate_t<4>(ptr, idx, nnz, sell, u, E, addvec, extra, v, G, Bg, Cu, N, s, xdot, dap, dot_done);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// E re-ordered for a gather: Eo[g,k] = E[g, perm[k]] (CSC order: perm = cperm; SELL position order: perm =
// sell.src, whose padding entries are -1 and get coefficient 0).  The output row stride may exceed n (SELL).
__global__ __launch_bounds__(kBlock) void k_permute_edges(const float* __restrict__ E,
                                                          const int* __restrict__ perm,
                                                          float* __restrict__ Eo, int nnz, int n) {
  const int g = blockIdx.y;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const int e = perm[k];
  Eo[(size_t)g * n + k] = (e >= 0) ? E[(size_t)g * nnz + e] : 0.f;
}

int launch_permute_edges(const float* E, const int* perm, float* Eo, int G, int nnz, int n, hipStream_t s) {
  if (n == 0) return MSGAT_OK;
  dim3 grid(cdiv(n, kBlock), G);
  hipLaunchKernelGGL(k_permute_edges, grid, dim3(kBlock), 0, s, E, perm, Eo, nnz, n);
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

// ---- backward aggregate + SDDMM in one pass over dv (PROJ_FIRST) ------------------------------------------------
//   du[g,c,m,:] = sum_{k into m} Ec[g,k] dv[g,c,crow[k],:]      dEp[g,chunk,k] = sum_{c,t} dv[g,c,crow[k],t] u[g,c,m,t]
// Both walk the same pairs (column m, CSC edge k) and both want the dv slab in LDS for the random side, so one block
// stages the slab once, and lane (m, j) -- which already holds dv[crow[k]][4j..4j+3] for the gather -- dots it with
// its own u[m][4j..4j+3] (one coalesced float4, the mirror image of the du store).  Against k_agg_lds + k_sddmm
// this reads dv once instead of twice: 3 slab units of HBM traffic instead of 4.  The T/4 lanes of a column leave
// their shares of an edge's dot in LDS (lane-private entries, no atomics); after a barrier lane k adds them in a
// fixed order and stores the chunk's partial in CSC order (k_edge_grad_csc maps it back through cperm).
template <int T4, bool FIRST>
__device__ __forceinline__ float4 gather_row_dot(const int* __restrict__ idx, const float* __restrict__ Eg, int e0,
                                                 int e1, int nnz, const float4* rows, int j, const float4& own,
                                                 float* __restrict__ pd, int dummy) {
  float4 acc = f4zero();
  // branch-free: an edge slot outside the column's range multiplies by 0 and parks its dot in the lane's dummy
  // entry -- a branch per edge would put every LDS row read in its own basic block, one LDS round trip after another
  auto edge = [&](int k, bool valid, float w, const float4& r) {
    f4fma(valid ? w : 0.f, r, acc);
    const float d = f4dot(r, own, 0.f);
    const int a = valid ? k * T4 + j : dummy;
    pd[a] = FIRST ? d : pd[a] + d;
  };
  const int b0 = min(e0, nnz - 8);  // nnz >= 8 (launch condition)
  const int4u m0 = *reinterpret_cast<const int4u*>(idx + b0);
  const int4u m1 = *reinterpret_cast<const int4u*>(idx + b0 + 4);
  const float4u w0 = *reinterpret_cast<const float4u*>(Eg + b0);
  const float4u w1 = *reinterpret_cast<const float4u*>(Eg + b0 + 4);
  {  // four LDS rows in flight at a time: eight cost the second resident block its registers
    float4 ra[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ra[k] = rows[m0.v[k] * T4 + j];
#pragma unroll
    for (int k = 0; k < 4; ++k) edge(b0 + k, b0 + k >= e0 && b0 + k < e1, w0.v[k], ra[k]);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    float4 rb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rb[k] = rows[m1.v[k] * T4 + j];
#pragma unroll
    for (int k = 0; k < 4; ++k) edge(b0 + 4 + k, b0 + 4 + k >= e0 && b0 + 4 + k < e1, w1.v[k], rb[k]);
  }
  for (int e = b0 + 8; e < e1; e += 4) {  // columns with more than 8 edges
    const int b = min(e, nnz - 4);
    const int4u m = *reinterpret_cast<const int4u*>(idx + b);
    const float4u w = *reinterpret_cast<const float4u*>(Eg + b);
#pragma unroll
    for (int k = 0; k < 4; ++k) edge(b + k, b + k >= e && b + k < e1, w.v[k], rows[m.v[k] * T4 + j]);
  }
  return acc;
}

// 768 lanes: the branch-free gather needs 72 registers, and 12 waves per block keep TWO blocks resident per CU
// (one streams its slab while the other gathers): 67.8 us against 77.4 us with 1024 lanes (one resident block)
constexpr int kAgsBlock = 768;

template <int T4>
__global__ __launch_bounds__(kAgsBlock) void k_agg_sddmm(
    const int* __restrict__ ptr, const int* __restrict__ idx, const float4* __restrict__ dv4,
    const float* __restrict__ Ec, const float4* __restrict__ u4, float4* __restrict__ du4,
    float* __restrict__ dEp, int Cu, int N, int nnz, int CH, int nchunks, int dvgs) {
  extern __shared__ float4 slab[];  // [CH][N][T4] of dv, then nnz * T4 floats of per-edge shares + a dummy per lane
  const int g = blockIdx.y;
  const int kc = blockIdx.x;
  const int c0 = kc * CH;
  const int ch = min(CH, Cu - c0);
  const int NT4 = N * T4;
  const size_t base = ((size_t)g * Cu + c0) * NT4;
  const int total = ch * NT4;
  float* pd = reinterpret_cast<float*>(slab + CH * NT4);
  const int dummy = nnz * T4 + threadIdx.x;

  const size_t vbase = ((size_t)g * dvgs + c0) * NT4;  // dv may be a channel slice of a wider tensor (dvgs channels per group)
  for (int i = threadIdx.x; i < total; i += kAgsBlock) slab[i] = dv4[vbase + i];
  __syncthreads();

  const float* Eg = Ec + (size_t)g * nnz;
  for (int s = threadIdx.x; s < NT4; s += kAgsBlock) {
    const int n = s / T4;
    const int j = s - n * T4;
    const int e0 = ptr[n], e1 = ptr[n + 1];
    du4[base + s] = gather_row_dot<T4, true>(idx, Eg, e0, e1, nnz, slab, j, u4[base + s], pd, dummy);
    for (int c = 1; c < ch; ++c) {
      const float4 own = u4[base + (size_t)c * NT4 + s];
      du4[base + (size_t)c * NT4 + s] = gather_row_dot<T4, false>(idx, Eg, e0, e1, nnz, slab + c * NT4, j, own, pd, dummy);
    }
  }
  __syncthreads();
  float* out = dEp + ((size_t)g * nchunks + kc) * nnz;
  for (int k = threadIdx.x; k < nnz; k += kAgsBlock) {
    float a = pd[k * T4];
#pragma unroll
    for (int j = 1; j < T4; ++j) a += pd[k * T4 + j];
    out[k] = a;
  }
}

static size_t agg_sddmm_lds(int N, int T, int Cu, int nnz) {
  const int CH = slab_channels(N, T, Cu, kLdsBudget);
  return (size_t)CH * N * T * sizeof(float) + ((size_t)nnz * (T / 4) + kAgsBlock) * sizeof(float);
}

// the fused pass needs a CSC graph whose slab + per-edge shares leave room for two blocks per CU
bool agg_sddmm_fusable(const msgat_graph_t& gr, int N, int T, int Cu) {
  if (gr.nnz < 8 || sell_usable(gr.sell_rows, gr.nnz, N, T) || sell_usable(gr.sell_cols, gr.nnz, N, T)) return false;
  if (slab_channels(N, T, Cu, kLdsBudget) < 1) return false;
  return agg_sddmm_lds(N, T, Cu, gr.nnz) <= (size_t)(kLdsMax / 2);
}

template <int T4>
static int launch_agg_sddmm_t(const msgat_graph_t& gr, const float* dv, const float* Ec, const float* u, float* du,
                              float* dEp, int G, int Cu, int N, hipStream_t s, int dvgs) {
  const int T = 4 * T4;
  const int CH = slab_channels(N, T, Cu, kLdsBudget);
  const size_t lds = agg_sddmm_lds(N, T, Cu, gr.nnz);
  {
    static LdsGrant granted;
    if (int st = grant_dynamic_lds(&k_agg_sddmm<T4>, lds, granted)) return st;
  }
  const int nchunks = cdiv(Cu, CH);
  hipLaunchKernelGGL(k_agg_sddmm<T4>, dim3(nchunks, G), dim3(kAgsBlock), lds, s, gr.colptr, gr.crow, (const float4*)dv,
                     Ec, (const float4*)u, (float4*)du, dEp, Cu, N, gr.nnz, CH, nchunks, dvgs > 0 ? dvgs : Cu);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_agg_sddmm(const msgat_graph_t& gr, const float* dv, const float* Ec, const float* u, float* du, float* dEp,
                     int G, int Cu, int N, int T, hipStream_t s, int dvgs) {
  switch (T) {
    case 4: return launch_agg_sddmm_t<1>(gr, dv, Ec, u, du, dEp, G, Cu, N, s, dvgs);
    case 8: return launch_agg_sddmm_t<2>(gr, dv, Ec, u, du, dEp, G, Cu, N, s, dvgs);
    case 12: return launch_agg_sddmm_t<3>(gr, dv, Ec, u, du, dEp, G, Cu, N, s, dvgs);
    case 16: return launch_agg_sddmm_t<4>(gr, dv, Ec, u, du, dEp, G, Cu, N, s, dvgs);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// ---- SDDMM on the SELL layout: one 4-timestep column of u in LDS per pass -------------------------------------
// Block (chunk kc, group g) walks its channels x T/4 columns; per pass it stages column j of u[c] and every
// lane (= row, through lane_row) dots its dv[c,row,4j..4j+3] against the staged entries of its row's
// neighbours.  The per-edge partial lives in dEp in SELL position order, so the read-modify-write is one
// coalesced 4-B-per-lane load + store, private to the lane: no atomics, fixed (c, j) summation order.
// Padding positions accumulate garbage nobody reads (k_edge_grad goes through sell.pos).
template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_sddmm_sell(
    const int* __restrict__ slice_off, const int* __restrict__ lane_row, const uint16_t* __restrict__ sidx,
    const float4* __restrict__ u4, const float4* __restrict__ dv4, float* dEp, int Cu, int N, int n_pos,
    int n_slices, int CH, int nchunks) {
  extern __shared__ float4 slab[];  // [N]
  const int g = blockIdx.y;
  const int kc = blockIdx.x;
  const int c0 = kc * CH;
  const int ch = min(CH, Cu - c0);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* out = dEp + ((size_t)g * nchunks + kc) * n_pos;
  for (int pass = 0; pass < ch * T4; ++pass) {
    const int c = pass / T4, j = pass - c * T4;
    const size_t base = ((size_t)g * Cu + c0 + c) * N * T4;
    stage_column<T4>(slab, u4 + base, j, N);
    __syncthreads();
    for (int s = wave; s < n_slices; s += kAggBlock / 64) {
      const int off = slice_off[s];
      const int ntrip = (slice_off[s + 1] - off) >> 8;
      if (ntrip == 0) continue;  // wave-uniform
      const int row = lane_row[64 * s + lane];
      const uint2* pi4 = reinterpret_cast<const uint2*>(sidx + off) + lane;
      float4* po4 = reinterpret_cast<float4*>(out + off) + lane;
      const float4 a = dv4[base + (size_t)max(row, 0) * T4 + j];  // lanes past N own padding only
      auto finish = [&](const SellTrip& x, int t) {  // .e carries the running partials of the trip's 4 edges
        const int4 id = sell_unpack(x.id);
        float4 o;
        o.x = f4dot(a, slab[id.x], pass == 0 ? 0.f : x.e.x);
        o.y = f4dot(a, slab[id.y], pass == 0 ? 0.f : x.e.y);
        o.z = f4dot(a, slab[id.z], pass == 0 ? 0.f : x.e.z);
        o.w = f4dot(a, slab[id.w], pass == 0 ? 0.f : x.e.w);
        po4[64 * t] = o;
      };
      SellTrip ta, tb;
      sell_issue(pi4, po4, 0, ntrip, ta);
      for (int t = 0; t < ntrip; t += 2) {
        sell_issue(pi4, po4, t + 1, ntrip, tb);
        finish(ta, t);
        sell_issue(pi4, po4, t + 2, ntrip, ta);  // clamped past the end: re-reads the last trip, not consumed
        if (t + 1 < ntrip) finish(tb, t + 1);    // wave-uniform; only stores inside
      }
    }
    __syncthreads();
  }
}

// ---- SDDMM on the SELL layout with the per-edge sums in registers ------------------------------------------------
// The read-modify-write of k_sddmm_sell moves every partial through L2 once per (channel, column) pass -- 72 times
// at the stress graph, 1.2 MB per pass and block, far more than an XCD's L2 holds for its 32 blocks: it ran at
// 6.7 ms, no faster than the CSR kernel it replaced.  Here the sums never leave the registers: the slices of a
// group are dealt to Q = ceil(n_slices / 32) blocks, each wave owning ONE PAIR of slices -- slice i from the
// wide end of the degree-sorted order and slice n_slices-1-i from the narrow end, so every pair needs about the
// same ~2 x (mean degree) / 4 trips -- whose neighbour indices (16-bit) are loaded once and whose sums live in
// kRegTrips float4 accumulators across all Cu x T/4 passes.  A pass is: stage column j of u[c] (every block of the
// group stages the same column; they share an XCD, so L2 serves the repeats), one dv value per slice, then LDS
// gathers and FMAs only.  Usable when sell.pair_trips <= kRegTrips (host-checked); k_sddmm_sell otherwise.
constexpr int kRegTrips = 12;

template <int T4>
__global__ __launch_bounds__(kAggBlock) void k_sddmm_sellreg(
    const int* __restrict__ slice_off, const int* __restrict__ lane_row, const uint16_t* __restrict__ sidx,
    const float4* __restrict__ u4, const float4* __restrict__ dv4, float* __restrict__ dE, int G, int Cu, int N,
    int n_pos, int n_slices, int Q) {
  extern __shared__ float4 slab[];  // [N]
  // XCD-aware map (see k_agg_sell): the Q blocks of a group sit in consecutive slots of XCD g % 8
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = slot % Q;
  const int g = (slot / Q) * 8 + xcd;
  if (g >= G) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int W = q * (kAggBlock / 64) + wave;  // pair index
  const int sA = W, sB = n_slices - 1 - W;
  const bool hasA = sA < n_slices && sA <= sB, hasB = hasA && sB > sA;  // wave-uniform
  const int sAc = hasA ? sA : 0, sBc = hasB ? sB : 0;
  const int offA = slice_off[sAc], offB = slice_off[sBc];
  const int ntA = hasA ? (slice_off[sAc + 1] - offA) >> 8 : 0;
  const int ntB = hasB ? (slice_off[sBc + 1] - offB) >> 8 : 0;
  const int nt = ntA + ntB;
  const int rowA = max(lane_row[64 * sAc + lane], 0), rowB = max(lane_row[64 * sBc + lane], 0);  // lanes past N: padding only

  // this wave's neighbour indices, for the whole kernel: trips 0 .. ntA-1 of slice A, then those of slice B
  const uint2* piA = reinterpret_cast<const uint2*>(sidx + offA) + lane;
  const uint2* piB = reinterpret_cast<const uint2*>(sidx + offB) + lane;
  uint2 ids[kRegTrips];
  float4 acc[kRegTrips];
#pragma unroll
  for (int t = 0; t < kRegTrips; ++t) {
    const int tt = min(t, max(nt - 1, 0));
    const uint2* p = (tt < ntA) ? piA + 64 * tt : piB + 64 * (tt - ntA);  // wave-uniform choice, load unconditional
    ids[t] = (nt > 0) ? *p : make_uint2(0u, 0u);
    acc[t] = f4zero();
  }

  // Pass order.  Every 128-B line of a [N,T] slab holds the T/4 columns of 2.67 nodes, so the T/4 passes over one
  // slab touch the SAME lines T/4 times, microseconds apart, with 32 blocks streaming through the XCD's 4 MB L2 in
  // between.  The Q blocks of a group (side by side on one XCD, in step with each other) therefore walk the columns
  // of a slab in ROTATED order, block q starting at column q % T4: at any moment they cover all T4 columns of the
  // slab together and share its lines (2.96 -> 2.73 ms).  What was tried beyond that, and what it showed
  // (profiles/r03/stress_sddmm_lab.txt): one block per (group, slice quarter, COLUMN) with per-column partials cuts the
  // fabric traffic from 2.8x to 1.4x the algorithmic bytes -- an XCD then holds 2.7 groups' lines instead of 8 -- and
  // runs no faster (2.99 ms + 0.17 ms more in k_edge_grad): the kernel is not fabric-bound but serialised inside the
  // CU (staging round trip, then ~770 conflicted ds_read_b128 per pass, one block per CU); reading a wave's dv rows
  // whole, once per channel, needs 16 more registers than 4 waves per SIMD leave (scratch: 5.3 ms).
  for (int pass = 0; pass < Cu * T4; ++pass) {
    const int c = pass / T4, jr = pass - c * T4;
    const int j = (jr + q) % T4;
    const size_t base = ((size_t)g * Cu + c) * N * T4;
    const float4 aA = dv4[base + (size_t)rowA * T4 + j];
    const float4 aB = dv4[base + (size_t)rowB * T4 + j];
    stage_column<T4, 4>(slab, u4 + base, j, N);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kRegTrips; ++t) {
      if (t < nt) {  // wave-uniform; LDS reads and FMAs only
        const bool inA = t < ntA;
        const float4 a = make_float4(inA ? aA.x : aB.x, inA ? aA.y : aB.y, inA ? aA.z : aB.z, inA ? aA.w : aB.w);
        const int4 id = sell_unpack(ids[t]);
        acc[t].x = f4dot(a, slab[id.x], acc[t].x);
        acc[t].y = f4dot(a, slab[id.y], acc[t].y);
        acc[t].z = f4dot(a, slab[id.z], acc[t].z);
        acc[t].w = f4dot(a, slab[id.w], acc[t].w);
      }
    }
    __syncthreads();
  }

  float* out = dE + (size_t)g * n_pos;
  float4* poA = reinterpret_cast<float4*>(out + offA) + lane;
  float4* poB = reinterpret_cast<float4*>(out + offB) + lane;
#pragma unroll
  for (int t = 0; t < kRegTrips; ++t)
    if (t < nt) *((t < ntA) ? poA + 64 * t : poB + 64 * (t - ntA)) = acc[t];
}

static bool sddmm_in_registers(const msgat_sell_t& sl) { return sl.pair_trips > 0 && sl.pair_trips <= kRegTrips; }

// channel chunks of the SDDMM (= partial buffers of [G,nnz]): LDS slab form: as many channels as fit the LDS
// budget per chunk; SELL form: enough chunks to give every CU a block, at most one per channel
int sddmm_chunks(int G, int Cu, int N, int T, const msgat_sell_t* sell) {
  if (sell != nullptr) {
    if (sddmm_in_registers(*sell)) return 1;  // the sums leave the registers once, complete
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
  if (sell_usable(gr.sell_rows, gr.nnz, N, T)) {  // partials come out in SELL position order (k_edge_grad: pos)
    const msgat_sell_t& sl = gr.sell_rows;
    const int nch = sddmm_chunks(G, Cu, N, T, &sl);
    const int chj = cdiv(Cu, nch);
    const size_t lds = (size_t)N * sizeof(float4);
    if (sddmm_in_registers(sl)) {
      {
        static LdsGrant granted;
        if (int st = grant_dynamic_lds(&k_sddmm_sellreg<T4>, lds, granted)) return st;
      }
      const int Q = cdiv(sl.n_slices, 2 * (kAggBlock / 64));
      hipLaunchKernelGGL(k_sddmm_sellreg<T4>, dim3((unsigned)cdiv(G, 8) * 8 * Q), dim3(kAggBlock), lds, s, sl.slice_off,
                         sl.lane_row, sl.idx, (const float4*)u, (const float4*)dv, dEp, G, Cu, N, sl.n_pos, sl.n_slices, Q);
      MSGAT_CHECK_LAUNCH();
      return MSGAT_OK;
    }
    {
      static LdsGrant granted;
      if (int st = grant_dynamic_lds(&k_sddmm_sell<T4>, lds, granted)) return st;
    }
    hipLaunchKernelGGL(k_sddmm_sell<T4>, dim3(nch, G), dim3(kAggBlock), lds, s, sl.slice_off, sl.lane_row, sl.idx,
                       (const float4*)u, (const float4*)dv, dEp, Cu, N, sl.n_pos, sl.n_slices, chj, nch);
  } else if (CH >= 1) {
    const size_t lds = (size_t)CH * N * T * sizeof(float);
    {
      static LdsGrant granted;
      if (int st = grant_dynamic_lds(&k_sddmm<T4, true>, lds, granted)) return st;
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
