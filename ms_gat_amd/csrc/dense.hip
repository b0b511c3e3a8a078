// The dense side of the attention: everything that needs ALL N columns of a row.
//
// attention.py:34 takes softmax over the full [N] row of S = (k Wg) q^T and only then
// (attention.py:36) multiplies by the adjacency, so the normaliser of every edge
// coefficient is a sum over all N columns.  Nothing N x N is ever written here:
//
//   forward   kW = q Wg;  lse[n] = log2 sum_m 2^(kW2[n].q[m])     (online max/sum, flash style)
//             pq[n] = sum_m softmax(S)[n,m] q[m]                   (only when training)
//             E[e]  = 2^(kW2[row_e].q[col_e] - lse[row_e]) * adj_e (edges only)
//   backward  dq[m] -= sum_n softmax(S)[n,m] delta[n] kW[n]        (the dense column term)
//
// These kernels read [G,N,T] and are bound by instruction issue, not HBM.  Both matrix products of a 16x16 tile run on the
// matrix cores:
//   1. scores   S = own . streamed^T      K = T     -> T/4 exact-fp32 v_mfma_f32_16x16x4_f32 (3 at T = 12): a k-ordered fmaf
//      chain, bit-compatible with the VALU chain of the edge pass and with the backward's re-creation of the tile
//   2. payload  acc += P . payload        K = 16    -> since round 6 two v_mfma_f32_16x16x32_f16 on two-term fp16 operands
//      (halfsplit.hpp; rounds 1-5: four fp32 MFMAs, 128 of a tile's ~320 clocks).  The accumulator of (1), after the exp, is in
//      the B layout of (2) already; it is split per tile on the VALU (8 instructions), the payload matrix once per staged chunk
// and the VALU is left with 4 exp + 8 split + ~15 bookkeeping instructions per tile.  An all-VALU version of these loops issued
// ~32 instructions per (row, column) pair and ran at ~75 us; a version with (1) on MFMA and (2) on the VALU was paced by LDS
// broadcast traffic (in-kernel stamps).  From 1536 nodes both products run on split bf16 / fp16 operands (dense_bf16.hip).
//
// 16x16x4 layouts: A[i][k]: lane (i = lane & 15, k = lane >> 4); B[k][j]: lane (j = lane & 15,
// k = lane >> 4); D[4*quad + r][j] in register r of lane (j = lane & 15, quad = lane >> 4).
// The MFMA is a k-ordered fmaf chain starting from C, the same order as the VALU chain that
// computes the edge scores, so edge pass and backward agree on every score bit for bit.
#include "common.hpp"
#include "halfsplit.hpp"

#include <cstdlib>

namespace msgat {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// NOT inline asm: hipcc's hazard recognizer does not look inside an asm statement, so an asm v_max3 that
// reads an MFMA result straight out of VGPRs gets no wait states after the MFMA and sees stale registers
// (found with 8-wave blocks, where the accumulators live in VGPRs: the deferred re-base then missed new
// row maxima and saturated softmax rows overflowed, differently from run to run).  The compiler fuses
// the nested fmaxf into v_max3_f32 by itself.
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Where a node's q row comes from: the q array (XC = 0), or -- the first MEAM of a component on a one-feature dataset --
// q = sum_c alpha[c] x[c] computed on the fly from the XC channel slabs of x[g] (attention.py:33).  The
// fma order is k_qonly's (c ascending from zero), so the bits are the ones that kernel wrote; the block stores the rows it
// owns for the backward.  That removes the k_qonly launch (4.9 us at PEMSD7 size) from the forward.
template <int T, int XC>
struct QRows {
  const float* base;      // XC == 0: q[g] ([N,T]);  XC > 0: x[g] ([XC,N,T])
  float a[XC > 0 ? XC : 1];
  size_t cs;              // channel stride of x in floats (N * T)
  __device__ __forceinline__ float4 row4(size_t node, int t4) const {
    if (XC == 0) return reinterpret_cast<const float4*>(base + node * T)[t4];
    float4 v[XC > 0 ? XC : 1];
#pragma unroll
    for (int c = 0; c < XC; ++c) v[c] = reinterpret_cast<const float4*>(base + c * cs + node * T)[t4];
    float4 acc = f4zero();
#pragma unroll
    for (int c = 0; c < XC; ++c) f4fma(a[c], v[c], acc);
    return acc;
  }
};
template <int T, int XC>
__device__ __forceinline__ QRows<T, XC> make_qrows(const float* qsrc, const float* alpha, int g, int r, int N) {
  QRows<T, XC> qr;
  qr.cs = (size_t)N * T;
  qr.base = qsrc + (size_t)g * (XC > 0 ? XC : 1) * N * T;
#pragma unroll
  for (int c = 0; c < (XC > 0 ? XC : 1); ++c) qr.a[c] = XC > 0 ? alpha[r * XC + c] : 0.f;
  return qr;
}

// AGG_FIRST forward with 1 or 3 input channels (the first MEAM of every component): the block has just written the edge
// coefficients of its rows, so it can finish the layer itself -- y[c,n,:] = sum_e E_e x[c,col_e,:] (attention.py:36) and
// z[o,n,:] = sum_c W[o,c] y[c,n,:] (msgat.py:27) for its rows -- instead of a separate k_agg_proj launch that re-reads E and the
// row extents (9.3 us of a 73-us forward at PEMSD4 size).  Lane = (row, 4 timesteps); four edges per trip with clamped loads;
// the output channels in chunks of eight so that the tail's registers stay below the score loop's.
// The row extents, and -- when the block's rows have at most kTailEdges edges (CACHED) -- the column indices and the
// coefficients the edge pass has just computed, wait in LDS: the tail's only global round trip is the gather of x rows.
constexpr int kTailEdges = 1024;
typedef const int __attribute__((address_space(3))) * lds_ci;
typedef const float __attribute__((address_space(3))) * lds_cf;

template <int T, int XC, bool CACHED>
__device__ __forceinline__ void agg_proj_tail(const QRows<T, XC>& xr, lds_ci rp, lds_ci lcol, lds_cf lE,
                                              const int* __restrict__ col, const float* __restrict__ Eg,
                                              const float* __restrict__ Wr, int Co, float* __restrict__ yg,
                                              float* __restrict__ zg, int n0, int rows, int N) {
  constexpr int T4 = T / 4;
  constexpr int C = XC > 0 ? XC : 1;
  const size_t NT = (size_t)N * T;
  const int eb = rp[0];
  for (int s = threadIdx.x; s < rows * T4; s += blockDim.x) {
    const int rl = s / T4, j = s - rl * T4;
    const int n = n0 + rl;
    const int e0 = rp[rl], e1 = rp[rl + 1];
    float4 y[C];
#pragma unroll
    for (int c = 0; c < C; ++c) y[c] = f4zero();
    for (int e = e0; e < e1; e += 4) {
      int cc[4];
      float w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ee = min(e + i, e1 - 1);
        cc[i] = CACHED ? lcol[ee - eb] : col[ee];
        const float raw = CACHED ? lE[ee - eb] : Eg[ee];
        w[i] = (e + i < e1) ? raw : 0.f;
      }
      float4 xv[4][C];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) xv[i][c] = reinterpret_cast<const float4*>(xr.base + c * xr.cs + (size_t)cc[i] * T)[j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) f4fma(w[i], xv[i][c], y[c]);
    }
    if (yg != nullptr) {
#pragma unroll
      for (int c = 0; c < C; ++c) reinterpret_cast<float4*>(yg + c * NT + (size_t)n * T)[j] = y[c];
    }
    for (int o0 = 0; o0 < Co; o0 += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int o = min(o0 + k, Co - 1);
        float4 acc = f4zero();
#pragma unroll
        for (int c = 0; c < C; ++c) f4fma(Wr[o * C + c], y[c], acc);
        if (o0 + k < Co) reinterpret_cast<float4*>(zg + o * NT + (size_t)n * T)[j] = acc;
      }
    }
  }
}

// 8 waves = 2 per SIMD.  Tried (round 2): 7 waves per block, which makes the PEMSD7 grid (N = 883, G = 96) exactly
// 3.0 blocks per CU instead of 2.6 (3 rounds of 112 rows instead of 3 of 128): k_scores 60.7 -> 63.7 us,
// k_bwd_dense_col 60.2 -> 61.0 us -- 7 waves spread 2,2,2,1 over the SIMDs, and the matrix pipe is per SIMD.
#ifndef MSGAT_DWAVES
#define MSGAT_DWAVES 8
#endif
constexpr int kDWaves = MSGAT_DWAVES;  // the whole grid must be resident at once: leftover blocks run as a second round
constexpr int kDBlock = 64 * kDWaves;
constexpr int kDRows = 16 * kDWaves;   // own rows (forward) / columns (backward) per block
constexpr int kDMC = 128;              // streamed columns (forward) / rows (backward) staged per step
constexpr int kPS = 20;                // floats per staged payload row: T <= 16 values, zero padded;
                                       // 20 makes the payload fragment read bank-conflict free
constexpr float kDefer = 8.f;          // re-base the running max only when a score exceeds it by 2^8

// ---- the payload product on the fp16 matrix core (round 6; halfsplit.hpp) ---------------------------------------------------
// The score product stays on v_mfma_f32_16x16x4_f32 (exact fp32: edge pass, forward tile and backward tile agree bit for
// bit), but its four payload MFMAs per tile -- 128 of a tile's ~320 clocks, at the vector rate and on the VALU's issue port --
// become two v_mfma_f32_16x16x32_f16 on P split into two fp16 terms (8 VALU instructions per tile).  The payload matrix of a
// staged chunk (q rows forward, delta kW rows backward; fp32 in LDS for the score product anyway) is converted ONCE per chunk
// by the block into fp16 "planes" [tile of 16 items][quad][row s] x 16 B = the terms (h | m) of items 4 quad .. 4 quad + 3 at
// payload row s: a lane's A fragment is one conflict-free ds_read_b128.  fp16 has no range to spare, so the planes carry a
// power-of-two scale that follows the largest entry seen so far (chunk maxima ride along with the staging; when a chunk
// raises the maximum the accumulators are re-scaled, like the running maximum of the softmax), and P is carried times 2^7
// (forward) / 2^14 (backward).  What fp16 drops is below 2^-23 of a value or 2^-38 of the largest payload entry.
constexpr int kPlaneU4 = (kDMC / 16) * 4 * 16;   // uint4 per chunk of kDMC staged items

// rows [items][stride] fp32 (T payload values each, zero past `items`) -> planes; ONES_ROW: row s = T carries 1 for live items
// (the payload product then leaves the row sum of P at D2[s = T]).  Rows s > T of the planes are zeroed once per kernel.
template <int T, bool ONES_ROW>
__device__ __forceinline__ void build_payload_planes(const float* rows, int stride, int items, float scale, float one,
                                                     uint4* planes, int tid, int nthreads) {
  constexpr int RW = T + (ONES_ROW ? 1 : 0);
  const int ntq = ((items + 15) >> 4) * 4;
  for (int u = tid; u < ntq * RW; u += nthreads) {
    const int tq = u / RW, sr = u - tq * RW;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * tq + r;
      v[r] = (sr < T) ? rows[c * stride + sr] * scale : ((c < items) ? 1.f : 0.f);
    }
    uint4 f;
    split2_f16(v, one, f.x, f.y, f.z, f.w);
    planes[tq * 16 + sr] = f;
  }
}
// The whole grid of a dense pass is resident, 2-3 blocks to a CU, and a SIMD arbitrates its waves by priority, then AGE: at equal
// priority the block dispatched first takes most of the issue slots, finishes at ~60 % of the kernel, and the last block of a
// CU runs the final quarter alone at two waves per SIMD (in-kernel stamps at PEMSD7 size: blocks take 65 K .. 113 K clocks,
// 768 / 512 / 256 of them alive over time).  A wave therefore LOWERS its priority as it advances through its chunks: whoever
// is behind gets the slots, and the blocks of a CU finish together.
__device__ __forceinline__ void prio_by_progress(int done, int total) {
  const int lvl = 3 - min(3, (4 * done) / max(total, 1));
  if (lvl == 3) __builtin_amdgcn_s_setprio(3);
  else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
  else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
// largest |entry| of the block's next chunk: every lane's own maximum -> wave maximum -> wmax[wave] (read by all after the
// staging barrier)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float max_abs4(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// Every block of these grids does the same work and the whole grid is resident at once, but the
// dispatcher packs workgroups onto a CU until a resource runs out (in-kernel stamps: some CUs held
// 8 blocks, others 3, and the kernel ran 1.7x longer than its median block).  Asking for unused
// dynamic LDS caps the blocks per CU at ceil(blocks / CUs), which forces an even spread.
static size_t balance_pad_bytes(int nblocks, size_t static_lds) {
  const int ncu = device_cu_count();
#ifdef MSGAT_DCAP
  const int per_cu = MSGAT_DCAP;        // lab: resident blocks per CU capped, the rest dispatched as slots free up
#else
  const int per_cu = cdiv(nblocks, ncu);
#endif
#ifndef MSGAT_LDS_GRAIN
#define MSGAT_LDS_GRAIN 4096
#endif
  // LDS a block may occupy so that exactly per_cu fit: rounded DOWN to a coarse grain -- the allocator rounds a block's
  // request up, and with 3 x 54 016 B of the 163 840 only TWO blocks were resident (in-kernel stamps: the third block of
  // 160 CUs started when the first had finished; k_scores 61.7 us)
  const size_t share = ((size_t)kLdsMax / per_cu) / MSGAT_LDS_GRAIN * MSGAT_LDS_GRAIN;
  if (share <= static_lds + 1024) return 0;        // already limited by its own LDS
  if ((size_t)(per_cu + 1) * share <= (size_t)kLdsMax) return 0;   // (cannot happen for grains below 160 KB / per_cu^2)
  size_t pad = share - static_lds;
  if (pad > 64 * 1024 - 256) pad = 64 * 1024 - 256;  // stay under the default dynamic-LDS limit
  return pad & ~(size_t)255;
}

// ---- forward -----------------------------------------------------------------------------------------
// Wave w owns rows n0 + 16w .. +15 (B operand of the score product: their kW2) and streams all
// columns.  Score tile D[i = column][j = row]: lane (row j, quad) holds its row against columns
// m0 + 4*quad + r.  Payload tile D2[i = timestep s][j = row] += q[m][s] P[row][m].
// F16P: the payload product on the fp16 matrix core (halfsplit.hpp); false: on four exact-fp32 MFMAs per tile as in rounds 1-5 --
// kept for grids that leave CUs empty (PEMSD4's single relation: 192 blocks, two waves per SIMD), where the matrix pipe is idle
// most of the time anyway and the conversion of the payload planes with its third barrier per chunk is pure latency
// (0.2168 -> 0.2205 ms per hot-path step with the fp16 form there, same box).
template <int T, bool WITH_PQ, int XC = 0, bool F16P = true>
__global__ __launch_bounds__(kDBlock) void k_scores(
    const float* __restrict__ q, const float* __restrict__ Wg, const int* __restrict__ rowptr,
    const int* __restrict__ col, const float* __restrict__ val, const int* __restrict__ erow,
    float* __restrict__ kW, float* __restrict__ lse, float* __restrict__ pq, float* __restrict__ E,
    const int* __restrict__ cpos, float* __restrict__ Ec, int Bg, int N, int nnz,
    const float* __restrict__ alpha, float* __restrict__ qout,   // XC > 0: `q` is x[G,XC,N,T], the q rows go to qout
    const float* __restrict__ apW, int apCo, float* __restrict__ apY, float* __restrict__ apZ,   // XC > 0 and apZ: the aggregate + projection tail
    float one) {   // 1.0f, opaque to the compiler (split2_f16)
  constexpr int T4 = T / 4;
  __shared__ float4 qs4[kDMC * kPS / 4];  // staged columns: [column][q(T) | zeros]
  __shared__ uint4 pl4[WITH_PQ && F16P ? kPlaneU4 : 1];   // their payload planes (fp16 terms, see build_payload_planes)
  __shared__ float wmax[kDWaves];
  __shared__ float kw2s[kDRows][T];       // the block's rows, log2-scaled, for the edge pass
  __shared__ float lse2s[kDRows];
  const float* qsw = reinterpret_cast<const float*>(qs4);

  const int g = blockIdx.y;
  const int r = g / Bg;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, quad = lane >> 4;
  const int n0 = blockIdx.x * kDRows;
  const int n = n0 + 16 * wave + j;  // this lane's row (shared by its 4 quads)
  const bool valid = n < N;
  const QRows<T, XC> qrows = make_qrows<T, XC>(q, alpha, g, r, N);
  const float* wg = Wg + (size_t)r * T * T;

  MSGAT_STAMP(0);
  // B fragment of the row: kW2[n][4*kk + quad]; kW itself is stored unscaled
  float bfrag[T4];
  {
    float qr[T];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      float4 v = f4zero();
      if (valid) v = qrows.row4((size_t)n, t4);
      qr[4 * t4 + 0] = v.x; qr[4 * t4 + 1] = v.y; qr[4 * t4 + 2] = v.z; qr[4 * t4 + 3] = v.w;
      if (XC > 0 && valid && quad == 0) reinterpret_cast<float4*>(qout + ((size_t)g * N + n) * T)[t4] = v;   // kept for backward
    }
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) {
      const int s = 4 * kk + quad;
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) a = fmaf(qr[t], wg[t * T + s], a);
      if (valid) kW[((size_t)g * N + n) * T + s] = a;
      bfrag[kk] = a * kLog2e;  // scores in log2 units: exp(x) = 2^(x log2 e)
      kw2s[16 * wave + j][s] = bfrag[kk];
    }
  }

  MSGAT_STAMP(1);
  // WITH_PQ and a spare row in the payload tile (T < 16): the row sum rides in the payload product -- staged
  // column m carries a 1 behind its T values of q, so D2[s = T][row] accumulates sum_m p[row][m] on the matrix core
  // and three v_add per tile leave the VALU stream (which the fp32 MFMAs share an issue port with)
  constexpr bool ONES = WITH_PQ && T < 16;
  constexpr bool HP = WITH_PQ && F16P;              // the fp16 payload path
  constexpr float kPOff = HP ? kPOffF : 0.f;        // P = 2^(S - m) is carried times 2^kPOff through the fp16 payload product
  float m = -3.0e38f;  // running max of the row, identical in its 4 quads; finite floor, not -inf
  float mo = m;        // m - kPOff: what the exponent subtracts
  float lsum = 0.f;    // this quad's share of sum_m 2^(S - mo) (unused with ONES)
  f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da, dc = da, dd = da;  // payload accumulators (fp16 path: tile a / tile b of a trip)
  int sexp = 100;      // the payload planes carry q * 2^sexp (block-uniform; follows the largest |q| staged so far)
  if (HP)
    for (int i = threadIdx.x; i < kPlaneU4; i += kDBlock) pl4[i] = make_uint4(0u, 0u, 0u, 0u);   // rows past the ones row stay zero

  // Staging is double-buffered through registers: the loads of chunk c+1 are issued before chunk c
  // is multiplied and land in LDS after it.  All blocks of the grid are resident at once, so a
  // block's own critical path -- not throughput -- set the kernel time while every chunk began
  // with an exposed global-load round trip.
  constexpr int kF4 = kPS / 4;                                   // float4s per staged column
  constexpr int kSt = (kDMC * kF4 + kDBlock - 1) / kDBlock;       // staging float4s per lane per chunk
  float4 pre[kSt];
  float premax = 0.f;   // largest |q| among this lane's share of the prefetched chunk
  auto prefetch = [&](int c0) {
    const int cols = min(kDMC, N - c0);
    premax = 0.f;
#pragma unroll
    for (int k = 0; k < kSt; ++k) {
      const int i = threadIdx.x + k * kDBlock;
      const int c = i / kF4, f = i - c * kF4;
      const bool live = (c < cols) && (f < T4);
      const float4 v = qrows.row4((size_t)(c0 + (live ? c : 0)), live ? f : 0);
      const float keep = live ? 1.f : 0.f;  // multiply, not select: keeps the load out of a branch
      const float onec = (!F16P && ONES && c < cols && f == T4) ? 1.f : 0.f;  // fp32 payload: the ones column, at index T of the staged row
      pre[k] = make_float4(fmaf(v.x, keep, onec), v.y * keep, v.z * keep, v.w * keep);
      if (HP) premax = fmaxf(premax, max_abs4(pre[k]));
    }
  };
  prefetch(0);
  for (int c0 = 0; c0 < N; c0 += kDMC) {
    const int cols = min(kDMC, N - c0);
    const int cols16 = (cols + 15) & ~15;
    prio_by_progress(c0, N);
    __syncthreads();  // every wave is done with the previous chunk
#pragma unroll
    for (int k = 0; k < kSt; ++k) {
      const int i = threadIdx.x + k * kDBlock;
      if (i < kDMC * kF4) qs4[i] = pre[k];
    }
    if (HP) {
      const float wm = wave_max(premax);
      if (lane == 0) wmax[wave] = wm;
    }
    __syncthreads();
    if (HP) {   // the chunk's payload planes, at the running scale (block-uniform arithmetic)
      float cmax = wmax[0];
#pragma unroll
      for (int w = 1; w < kDWaves; ++w) cmax = fmaxf(cmax, wmax[w]);
      const int ec = payload_scale_exp(cmax);
      if (ec < sexp) {   // a larger entry than any before: the sums so far move to the new scale (the ones row does not)
        const float f = pow2i(ec - sexp);
        if (quad < T4) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) { da[rr] *= f; db[rr] *= f; }
        }
        sexp = ec;
      }
      build_payload_planes<T, ONES>(qsw, kPS, cols, pow2i(sexp), one, pl4, threadIdx.x, kDBlock);
      __syncthreads();
    }
    prefetch(min(c0 + kDMC, max(N - 1, 0) / kDMC * kDMC));  // next chunk (the last trip re-reads its own)
    // two score tiles per trip: one max / vote / re-base decision for 32 columns, and the second tile's score chain
    // is independent of the first tile's exponentials.  An odd last tile is a tile of padding columns (masked).
    for (int m0 = 0; m0 < cols16; m0 += 32) {
      f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) S0 = mfma16(qsw[(m0 + j) * kPS + 4 * kk + quad], bfrag[kk], S0);
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) S1 = mfma16(qsw[(m0 + 16 + j) * kPS + 4 * kk + quad], bfrag[kk], S1);
      const int mq = m0 + 4 * quad;  // this lane's columns: mq .. mq+3 and mq+16 .. mq+19
      float sv[8];
      if (m0 + 32 > cols) {  // wave-uniform: only the chunk's last trip can hold padding columns
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          sv[rr] = (mq + rr < cols) ? S0[rr] : -3.0e38f;
          sv[4 + rr] = (mq + 16 + rr < cols) ? S1[rr] : -3.0e38f;
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { sv[rr] = S0[rr]; sv[4 + rr] = S1[rr]; }
      }
      const float cm = max3(max3(sv[0], sv[1], sv[2]), max3(sv[3], sv[4], sv[5]), fmaxf(sv[6], sv[7]));
      if (__any(cm > m + kDefer)) {  // rare (deferred re-base); the row's 4 quads must agree on m
        float cx = fmaxf(cm, __shfl_xor(cm, 16));
        cx = fmaxf(cx, __shfl_xor(cx, 32));
        const float mn = fmaxf(m, cx);
        const float sc = fast_exp2(m - mn);  // m at its floor on the first tile -> 0
        m = mn;
        mo = mn - kPOff;
        lsum *= sc;
        if (WITH_PQ) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) { da[rr] *= sc; db[rr] *= sc; dc[rr] *= sc; dd[rr] *= sc; }
        }
      }
      float p[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) p[rr] = fast_exp2(sv[rr] - mo);
      if (!ONES) lsum += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      if (HP) {
        // A2[i = s][k] = the planes' (h | m) of columns m0 + 4 quad .. +3 at row s = j; B2[k][j = row] = (Ph | Pm), then (Pm | Ph)
        const uint4 Fa = split_p(p, one), A2a = pl4[((m0 >> 4) * 4 + quad) * 16 + j];
        da = mfma_h(A2a, Fa, da); da = mfma_h(A2a, swap_halves(Fa), da);
        const uint4 Fb = split_p(p + 4, one), A2b = pl4[((m0 >> 4) * 4 + 4 + quad) * 16 + j];
        db = mfma_h(A2b, Fb, db); db = mfma_h(A2b, swap_halves(Fb), db);
      } else if (WITH_PQ) {
        // A2[i = s][k = quad] = q[column mq + rr][s = j]; B2[k = quad][j = row] = p[rr]
        da = mfma16(qsw[(mq + 0) * kPS + j], p[0], da);
        db = mfma16(qsw[(mq + 1) * kPS + j], p[1], db);
        dc = mfma16(qsw[(mq + 2) * kPS + j], p[2], dc);
        dd = mfma16(qsw[(mq + 3) * kPS + j], p[3], dd);
        da = mfma16(qsw[(mq + 16) * kPS + j], p[4], da);
        db = mfma16(qsw[(mq + 17) * kPS + j], p[5], db);
        dc = mfma16(qsw[(mq + 18) * kPS + j], p[6], dc);
        dd = mfma16(qsw[(mq + 19) * kPS + j], p[7], dd);
      }
    }
    if (c0 == 0) MSGAT_STAMP(2);
  }

  MSGAT_STAMP(3);
  if (ONES) {  // D2[s = T][row] sits in register T % 4 of the lanes with quad == T / 4
    const float mine = (da[T % 4] + db[T % 4]) + (dc[T % 4] + dd[T % 4]);
    lsum = __shfl(mine, j + 16 * (T / 4));
  } else {     // the row's 4 quads share m: their partial sums simply add
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
  }
  const float lse2 = mo + fast_log2(lsum);   // lsum carries 2^kPOff, like the payload sums
  if (quad == 0) {
    lse2s[16 * wave + j] = lse2;
    if (valid) lse[(size_t)g * N + n] = lse2;  // log2 units: backward re-creates the exponent bit for bit
  }
  if (WITH_PQ && valid && quad < T4) {  // D2[s = 4*quad + rr][row]: this lane owns pq[n][4*quad .. +3]
    const float inv = (HP ? pow2i(-sexp) : 1.0f) / lsum;   // the planes' scale leaves here; P's 2^kPOff cancels against lsum
    reinterpret_cast<float4*>(pq + ((size_t)g * N + n) * T)[quad] =
        make_float4(((da[0] + db[0]) + (dc[0] + dd[0])) * inv, ((da[1] + db[1]) + (dc[1] + dd[1])) * inv,
                    ((da[2] + db[2]) + (dc[2] + dd[2])) * inv, ((da[3] + db[3]) + (dc[3] + dd[3])) * inv);
  }
  __syncthreads();
  MSGAT_STAMP(4);

  // edge coefficients of this block's rows: one lane per CSR edge, coalesced over e.  The VALU
  // chain below is the MFMA's k order starting from 0, exactly what k_bwd_dense_col re-creates.
  const int e0 = rowptr[n0];
  const int e1 = rowptr[min(n0 + kDRows, N)];
  __shared__ int tl_rp[XC > 0 ? kDRows + 1 : 1];
  __shared__ int tl_col[XC > 0 ? kTailEdges : 1];
  __shared__ float tl_E[XC > 0 ? kTailEdges : 1];
  const bool with_tail = XC > 0 && apZ != nullptr;               // kernel-uniform
  const bool tail_cached = with_tail && e1 - e0 <= kTailEdges;    // block-uniform
  if (with_tail && (int)threadIdx.x <= min(kDRows, N - n0)) tl_rp[threadIdx.x] = rowptr[n0 + threadIdx.x];
  for (int e = e0 + threadIdx.x; e < e1; e += kDBlock) {
    const int nl = erow[e] - n0;
    const size_t ce = (size_t)col[e];
    float a = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = qrows.row4(ce, t4);
      a = fmaf(v.x, kw2s[nl][4 * t4 + 0], a);
      a = fmaf(v.y, kw2s[nl][4 * t4 + 1], a);
      a = fmaf(v.z, kw2s[nl][4 * t4 + 2], a);
      a = fmaf(v.w, kw2s[nl][4 * t4 + 3], a);
    }
    const float ev = fast_exp2(a - lse2s[nl]) * val[e];
    E[(size_t)g * nnz + e] = ev;
    // the same coefficient at its CSC position: the transposed passes of backward (du = E^T dv on the CSC) then
    // start without a re-ordering launch
    if (Ec != nullptr) Ec[(size_t)g * nnz + cpos[e]] = ev;
    if (tail_cached) { tl_col[e - e0] = (int)ce; tl_E[e - e0] = ev; }
  }
  MSGAT_STAMP(5);
  if (with_tail) {
    __syncthreads();                // the block's coefficients are in LDS (or, uncached, in memory)
    const size_t NT = (size_t)N * T;
    const float* Wr = apW + (size_t)r * apCo * XC;
    float* yg = apY != nullptr ? apY + (size_t)g * XC * NT : nullptr;
    float* zg = apZ + (size_t)g * apCo * NT;
    if (tail_cached)
      agg_proj_tail<T, XC, true>(qrows, (lds_ci)tl_rp, (lds_ci)tl_col, (lds_cf)tl_E, col, E + (size_t)g * nnz, Wr, apCo, yg, zg, n0,
                                 min(kDRows, N - n0), N);
    else
      agg_proj_tail<T, XC, false>(qrows, (lds_ci)tl_rp, (lds_ci)tl_col, (lds_cf)tl_E, col, E + (size_t)g * nnz, Wr, apCo, yg, zg,
                                  n0, min(kDRows, N - n0), N);
  }
}

// ---- forward, 7 owner waves + 1 helper wave ---------------------------------------------------------------------
// At N = 883 k_scores has 7 x 96 = 672 blocks of 8 waves for 256 CUs: 5.25 waves per SIMD, i.e. some SIMDs carry 6
// and the kernel takes what those take (87.5 % balance, dense_lab.txt).  Here a block owns 112 rows (7 waves x 16) and
// its eighth wave takes the LAST columns of all seven row groups: 8 x 96 = 768 blocks are exactly 3 per CU, every wave
// does the same number of score tiles (owner: Ta column tiles of its row group; helper: 7 x Th with Ta + Th = N / 16,
// Ta ~ 7 Th), and every SIMD carries 6 waves of 7/8 of the old work.  The helper's columns (at most 128) are staged once;
// it keeps step with the owners' chunk barriers, leaves its partial (running max, payload sums) per row group in LDS, and
// each owner folds that into its own before the logsumexp.  Same products in the same k order per tile as k_scores.
constexpr int kHOwners = 7;
constexpr int kHRows = 16 * kHOwners;

template <int T, bool WITH_PQ, int XC = 0>
__global__ __launch_bounds__(kDBlock) void k_scores7(
    const float* __restrict__ q, const float* __restrict__ Wg, const int* __restrict__ rowptr,
    const int* __restrict__ col, const float* __restrict__ val, const int* __restrict__ erow,
    float* __restrict__ kW, float* __restrict__ lse, float* __restrict__ pq, float* __restrict__ E,
    const int* __restrict__ cpos, float* __restrict__ Ec, int Bg, int N, int nnz, int Ca,
    const float* __restrict__ alpha, float* __restrict__ qout,
    const float* __restrict__ apW, int apCo, float* __restrict__ apY, float* __restrict__ apZ, float one) {
  constexpr int T4 = T / 4;
  constexpr bool ONES = WITH_PQ && T < 16;
  constexpr float kPOff = WITH_PQ ? kPOffF : 0.f;               // see k_scores
  constexpr int kOThreads = 64 * kHOwners;                      // lanes that stage the owners' chunks
  __shared__ float4 qs4[kDMC * kPS / 4];  // owners' chunk of columns [0, Ca)
  __shared__ float4 qh4[kDMC * kPS / 4];  // the helper's columns [Ca, N), staged once
  __shared__ uint4 pl4[WITH_PQ ? kPlaneU4 : 1];   // payload planes of the owners' chunk (see build_payload_planes)
  __shared__ uint4 ph4[WITH_PQ ? kPlaneU4 : 1];   //                of the helper's columns, built once
  __shared__ float wmax[kDWaves];
  __shared__ float kw2s[kHRows][T];
  __shared__ float lse2s[kHRows];
  __shared__ float hmax[kHRows];          // helper partial: running max of the row over its columns
  __shared__ float hsum[kHRows];          //                 sum of 2^(S - max) (when the payload tile has no ones row)
  __shared__ float hpay[kHRows][17];      //                 payload sums D2[s][row], [row][s] padded
  const float* qsw = reinterpret_cast<const float*>(qs4);
  const float* qhw = reinterpret_cast<const float*>(qh4);

  const int g = blockIdx.y;
  const int r = g / Bg;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, quad = lane >> 4;
  const bool owner = wave < kHOwners;   // wave-uniform
  const int n0 = blockIdx.x * kHRows;
  const int n = n0 + 16 * wave + j;
  const bool valid = owner && n < N;
  const QRows<T, XC> qrows = make_qrows<T, XC>(q, alpha, g, r, N);
  const float* wg = Wg + (size_t)r * T * T;
  const int colsh = N - Ca;             // 1 .. kDMC (host-checked)
  const int nchunk = cdiv(Ca, kDMC);

  // columns [0, Ca) go through LDS in chunks, register-prefetched by the owners' lanes (see k_scores).  The first chunk is
  // requested before anything else: its round trip then overlaps the helper's columns, their planes and the rows' kW -- all
  // 768 blocks start together, and as the FIRST thing after two barriers it was 9 K of a block's 90 K clocks (in-kernel stamps)
  constexpr int kSt = (kDMC * (kPS / 4) + kOThreads - 1) / kOThreads;
  float4 pre[kSt];
  float premax = 0.f;
  auto prefetch = [&](int c0) {
    constexpr int kF4p = kPS / 4;
    const int cols = min(kDMC, Ca - c0);
    premax = 0.f;
#pragma unroll
    for (int k = 0; k < kSt; ++k) {
      const int i = threadIdx.x + k * kOThreads;
      const int c = i / kF4p, f = i - c * kF4p;
      const bool live = (c < cols) && (f < T4);
      const float4 v = qrows.row4((size_t)(c0 + (live ? c : 0)), live ? f : 0);
      const float keep = live ? 1.f : 0.f;
      pre[k] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
      premax = fmaxf(premax, max_abs4(pre[k]));
    }
  };
  if (owner) prefetch(0);

  // the helper's columns, by every lane of the block
  constexpr int kF4 = kPS / 4;
  MSGAT_STAMP(0);
  float hmaxabs = 0.f;
  for (int i = threadIdx.x; i < kDMC * kF4; i += kDBlock) {
    const int c = i / kF4, f = i - c * kF4;
    const bool live = (c < colsh) && (f < T4);
    const float4 v = qrows.row4((size_t)(Ca + (live ? c : 0)), live ? f : 0);
    const float keep = live ? 1.f : 0.f;
    const float4 w = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
    qh4[i] = w;
    hmaxabs = fmaxf(hmaxabs, max_abs4(w));
  }
  int sexp = 100, sexph = 100;   // scale exponents of the owners' planes (running) and of the helper's (fixed)
  if (WITH_PQ) {
    for (int i = threadIdx.x; i < kPlaneU4; i += kDBlock) { pl4[i] = make_uint4(0u, 0u, 0u, 0u); ph4[i] = make_uint4(0u, 0u, 0u, 0u); }
    const float wm = wave_max(hmaxabs);
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();
    float cmax = wmax[0];
#pragma unroll
    for (int w = 1; w < kDWaves; ++w) cmax = fmaxf(cmax, wmax[w]);
    sexph = payload_scale_exp(cmax);
    build_payload_planes<T, ONES>(qhw, kPS, colsh, pow2i(sexph), one, ph4, threadIdx.x, kDBlock);
    __syncthreads();   // (wmax is re-used by the owners' chunks)
  }

  float bfrag[T4];

  float m = -3.0e38f, mo = m;   // running max; mo = m - kPOff is what the exponent subtracts
  float lsum = 0.f;
  f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da;

  // one trip = two score tiles of the staged columns m0 .. m0 + 31 of `buf` (k_scores' inner trip)
  auto trip = [&](const float* buf, const uint4* planes, int m0, int cols) {
    f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) S0 = mfma16(buf[(m0 + j) * kPS + 4 * kk + quad], bfrag[kk], S0);
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) S1 = mfma16(buf[(m0 + 16 + j) * kPS + 4 * kk + quad], bfrag[kk], S1);
    const int mq = m0 + 4 * quad;
    float sv[8];
    if (m0 + 32 > cols) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        sv[rr] = (mq + rr < cols) ? S0[rr] : -3.0e38f;
        sv[4 + rr] = (mq + 16 + rr < cols) ? S1[rr] : -3.0e38f;
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { sv[rr] = S0[rr]; sv[4 + rr] = S1[rr]; }
    }
    const float cm = max3(max3(sv[0], sv[1], sv[2]), max3(sv[3], sv[4], sv[5]), fmaxf(sv[6], sv[7]));
    if (__any(cm > m + kDefer)) {
      float cx = fmaxf(cm, __shfl_xor(cm, 16));
      cx = fmaxf(cx, __shfl_xor(cx, 32));
      const float mn = fmaxf(m, cx);
      const float sc = fast_exp2(m - mn);
      m = mn;
      mo = mn - kPOff;
      lsum *= sc;
      if (WITH_PQ) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { da[rr] *= sc; db[rr] *= sc; }
      }
    }
    float p[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) p[rr] = fast_exp2(sv[rr] - mo);
    if (!ONES) lsum += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    if (WITH_PQ) {   // see k_scores
      const uint4 Fa = split_p(p, one), A2a = planes[((m0 >> 4) * 4 + quad) * 16 + j];
      da = mfma_h(A2a, Fa, da); da = mfma_h(A2a, swap_halves(Fa), da);
      const uint4 Fb = split_p(p + 4, one), A2b = planes[((m0 >> 4) * 4 + 4 + quad) * 16 + j];
      db = mfma_h(A2b, Fb, db); db = mfma_h(A2b, swap_halves(Fb), db);
    }
  };

  if (owner) {
    // (the first chunk was requested at the top of the kernel; it lands while the rows' kW is computed)
    {
      float qr[T];
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) {
        float4 v = f4zero();
        if (valid) v = qrows.row4((size_t)n, t4);
        qr[4 * t4 + 0] = v.x; qr[4 * t4 + 1] = v.y; qr[4 * t4 + 2] = v.z; qr[4 * t4 + 3] = v.w;
        if (XC > 0 && valid && quad == 0) reinterpret_cast<float4*>(qout + ((size_t)g * N + n) * T)[t4] = v;   // kept for backward
      }
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) {
        const int s = 4 * kk + quad;
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) a = fmaf(qr[t], wg[t * T + s], a);
        if (valid) kW[((size_t)g * N + n) * T + s] = a;
        bfrag[kk] = a * kLog2e;
        kw2s[16 * wave + j][s] = bfrag[kk];
      }
    }
    MSGAT_STAMP(1);
    for (int c0 = 0; c0 < Ca; c0 += kDMC) {
      const int cols = min(kDMC, Ca - c0);
      const int cols16 = (cols + 15) & ~15;
      prio_by_progress(c0, Ca);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kSt; ++k) {
        const int i = threadIdx.x + k * kOThreads;
        if (i < kDMC * kF4) qs4[i] = pre[k];
      }
      if (WITH_PQ) {
        const float wm = wave_max(premax);
        if (lane == 0) wmax[wave] = wm;
      }
      __syncthreads();
      if (WITH_PQ) {   // the chunk's payload planes at the running scale, by the owners' lanes (see k_scores)
        float cmax = wmax[0];
#pragma unroll
        for (int w = 1; w < kHOwners; ++w) cmax = fmaxf(cmax, wmax[w]);
        const int ec = payload_scale_exp(cmax);
        if (ec < sexp) {
          const float f = pow2i(ec - sexp);
          if (quad < T4) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { da[rr] *= f; db[rr] *= f; }
          }
          sexp = ec;
        }
        build_payload_planes<T, ONES>(qsw, kPS, cols, pow2i(sexp), one, pl4, threadIdx.x, kOThreads);
        __syncthreads();
      }
      prefetch(min(c0 + kDMC, max(Ca - 1, 0) / kDMC * kDMC));
      for (int m0 = 0; m0 < cols16; m0 += 32) trip(qsw, pl4, m0, cols);
      if (c0 == 0) MSGAT_STAMP(2);
    }
    MSGAT_STAMP(3);
  } else {
    // the helper: row group rg's last columns during the owners' chunk rg (two barriers per chunk, like them)
    const int colsh16 = (colsh + 15) & ~15;
    constexpr int kBar = WITH_PQ ? 3 : 2;   // barriers of the owners per chunk
    const float hf = (WITH_PQ && quad < T4) ? pow2i(-sexph) : 1.f;   // payload rows leave in true units; the ones row is unscaled
    int nbar = 0;
    for (int rg = 0; rg < kHOwners; ++rg) {
      if (nbar < kBar * nchunk) {
#pragma unroll
        for (int bb = 0; bb < kBar; ++bb) __syncthreads();
        nbar += kBar;
      }
      prio_by_progress(rg, kHOwners);
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) bfrag[kk] = kw2s[16 * rg + j][4 * kk + quad];
      m = mo = -3.0e38f;
      lsum = 0.f;
      da = db = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int m0 = 0; m0 < colsh16; m0 += 32) trip(qhw, ph4, m0, colsh);
      if (!ONES) {
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
      }
      if (quad == 0) { hmax[16 * rg + j] = m; hsum[16 * rg + j] = lsum; }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) hpay[16 * rg + j][4 * quad + rr] = (da[rr] + db[rr]) * hf;
    }
    for (; nbar < kBar * nchunk; ++nbar) __syncthreads();
  }
  __syncthreads();   // the helper's partials are in LDS

  if (owner) {
    // fold the helper's share of this row group in: both parts re-based to the common maximum
    const int row = 16 * wave + j;
    const float mh = hmax[row];
    const float mt = fmaxf(m, mh);
    const float so = fast_exp2(m - mt), sh = fast_exp2(mh - mt);
    const float fo = (WITH_PQ && quad < T4) ? pow2i(-sexp) * so : so;   // the own payload sums to true units (not the ones row)
    float tot[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) tot[rr] = (da[rr] + db[rr]) * fo + hpay[row][4 * quad + rr] * sh;
    if (ONES) {
      lsum = __shfl(tot[T % 4], j + 16 * (T / 4));
    } else {
      lsum += __shfl_xor(lsum, 16);
      lsum += __shfl_xor(lsum, 32);
      lsum = lsum * so + hsum[row] * sh;
    }
    const float lse2 = (mt - kPOff) + fast_log2(lsum);   // both parts carry P times 2^kPOff
    if (quad == 0) {
      lse2s[row] = lse2;
      if (valid) lse[(size_t)g * N + n] = lse2;
    }
    if (WITH_PQ && valid && quad < T4) {
      const float inv = 1.0f / lsum;
      reinterpret_cast<float4*>(pq + ((size_t)g * N + n) * T)[quad] =
          make_float4(tot[0] * inv, tot[1] * inv, tot[2] * inv, tot[3] * inv);
    }
  }
  __syncthreads();
  MSGAT_STAMP(4);

  // edge coefficients of this block's rows (see k_scores)
  const int e0 = rowptr[min(n0, N)];
  const int e1 = rowptr[min(n0 + kHRows, N)];
  // the tail's row extents / edge list / coefficients take over the staging buffers (dead behind the barrier above): with
  // LDS of their own the block would not fit three to a CU beside the payload planes
  static_assert((kHRows + 4 + 2 * kTailEdges) * 4 <= (int)sizeof(float4) * 2 * (kDMC * kPS / 4), "tail arrays exceed the staging buffers");
  int* tl_rp = reinterpret_cast<int*>(qs4);
  int* tl_col = tl_rp + kHRows + 4;
  float* tl_E = reinterpret_cast<float*>(tl_col + kTailEdges);
  const bool with_tail = XC > 0 && apZ != nullptr;               // kernel-uniform
  const bool tail_cached = with_tail && e1 - e0 <= kTailEdges;    // block-uniform
  if (with_tail && (int)threadIdx.x <= min(kHRows, N - n0)) tl_rp[threadIdx.x] = rowptr[n0 + threadIdx.x];
  for (int e = e0 + threadIdx.x; e < e1; e += kDBlock) {
    const int nl = erow[e] - n0;
    const size_t ce = (size_t)col[e];
    float a = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = qrows.row4(ce, t4);
      a = fmaf(v.x, kw2s[nl][4 * t4 + 0], a);
      a = fmaf(v.y, kw2s[nl][4 * t4 + 1], a);
      a = fmaf(v.z, kw2s[nl][4 * t4 + 2], a);
      a = fmaf(v.w, kw2s[nl][4 * t4 + 3], a);
    }
    const float ev = fast_exp2(a - lse2s[nl]) * val[e];
    E[(size_t)g * nnz + e] = ev;
    if (Ec != nullptr) Ec[(size_t)g * nnz + cpos[e]] = ev;
    if (tail_cached) { tl_col[e - e0] = (int)ce; tl_E[e - e0] = ev; }
  }
  MSGAT_STAMP(5);
  if (with_tail) {   // see k_scores
    __syncthreads();
    const size_t NT = (size_t)N * T;
    const float* Wr = apW + (size_t)r * apCo * XC;
    float* yg = apY != nullptr ? apY + (size_t)g * XC * NT : nullptr;
    float* zg = apZ + (size_t)g * apCo * NT;
    if (tail_cached)
      agg_proj_tail<T, XC, true>(qrows, (lds_ci)tl_rp, (lds_ci)tl_col, (lds_cf)tl_E, col, E + (size_t)g * nnz, Wr, apCo, yg, zg, n0,
                                 min(kHRows, N - n0), N);
    else
      agg_proj_tail<T, XC, false>(qrows, (lds_ci)tl_rp, (lds_ci)tl_col, (lds_cf)tl_E, col, E + (size_t)g * nnz, Wr, apCo, yg, zg,
                                  n0, min(kHRows, N - n0), N);
  }
}

// Column split of the 7 + 1 form: the helper takes Th = 2 round(Tn / 18) of the Tn = ceil(N / 16) column tiles (an even
// count: trips are two tiles; a helper tile costs about what an owner tile costs, and 7 Th ~ Tn - Th).  Returns the owners'
// column count Ca (a multiple of 16; an odd multiple when Tn is odd -- the kernels mask the empty half of the owners' last
// two-tile trip, and the balance estimate below charges the owners for it), or 0 when the form does not apply: helper
// columns must fit one staged chunk, and it only pays where the 128-row blocks leave the CUs unevenly loaded while the
// 112-row blocks do not.
static int scores7_owner_columns(int N, int G) {
  const int ncu = device_cu_count();
  const int Tn = cdiv(N, 16);
  const int Th = 2 * ((Tn + 9) / 18);
  if (Th < 2 || Th * 16 > kDMC || Tn - Th < 2) return 0;
  const double b8 = (double)cdiv(N, kDRows) * G / ncu, b7 = (double)cdiv(N, kHRows) * G / ncu;
  const double eff8 = b8 / ceil(b8), eff7 = b7 / ceil(b7) * (7.0 * Tn / (8.0 * (Tn - Th + ((Tn - Th) & 1))));   // useful share of the busiest CU's time
  if (b7 > 6.0 || eff7 <= eff8 + 0.03) return 0;
  return (Tn - Th) * 16;
}

// XC > 0: `q` is read-only x[G,XC,N,T] and the q rows are WRITTEN to qout (see QRows); XC = 0: `q` is the q array
template <int T, int XC>
static int launch_scores_x(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse, float* pq,
                           float* E, float* Ec, int G, int Bg, int N, hipStream_t s, const float* alpha, float* qout,
                           const float* apW = nullptr, int apCo = 0, float* apY = nullptr, float* apZ = nullptr) {
#ifndef MSGAT_NO_SCORES7
  if (const int Ca = scores7_owner_columns(N, G)) {
    dim3 grid7(cdiv(N, kHRows), G);
    const size_t lds7 = sizeof(float) * (2 * kDMC * kPS + kHRows * T + 3 * kHRows + kHRows * 17 + kDWaves) +
                        (pq != nullptr ? 2 * sizeof(uint4) * kPlaneU4 : 2 * sizeof(uint4));   // + the payload planes (the tail re-uses the staging buffers)
    const size_t pad7 = balance_pad_bytes((int)(grid7.x * grid7.y), lds7);
    if (pq != nullptr)
      hipLaunchKernelGGL((k_scores7<T, true, XC>), grid7, dim3(kDBlock), pad7, s, q, Wg, gr.rowptr, gr.col, gr.val,
                         gr.erow, kW, lse, pq, E, gr.cpos, Ec, Bg, N, gr.nnz, Ca, alpha, qout, apW, apCo, apY, apZ, 1.0f);
    else
      hipLaunchKernelGGL((k_scores7<T, false, XC>), grid7, dim3(kDBlock), pad7, s, q, Wg, gr.rowptr, gr.col, gr.val,
                         gr.erow, kW, lse, pq, E, gr.cpos, Ec, Bg, N, gr.nnz, Ca, alpha, qout, apW, apCo, apY, apZ, 1.0f);
    MSGAT_CHECK_LAUNCH();
    return MSGAT_OK;
  }
#endif
  dim3 grid(cdiv(N, kDRows), G);
  const size_t static_lds = sizeof(float) * (kDMC * kPS + kDRows * T + kDRows + kDWaves) +
                            (pq != nullptr ? sizeof(uint4) * kPlaneU4 : sizeof(uint4)) +
                            (XC > 0 ? sizeof(int) * (kDRows + 1 + 2 * kTailEdges) : 0);
  const size_t pad = balance_pad_bytes((int)(grid.x * grid.y), static_lds);
  if (pq != nullptr && (int)(grid.x * grid.y) <= device_cu_count())   // a grid that leaves CUs empty: the fp32 payload product
    hipLaunchKernelGGL((k_scores<T, true, XC, false>), grid, dim3(kDBlock), pad, s, q, Wg, gr.rowptr, gr.col, gr.val,
                       gr.erow, kW, lse, pq, E, gr.cpos, Ec, Bg, N, gr.nnz, alpha, qout, apW, apCo, apY, apZ, 1.0f);
  else if (pq != nullptr)
    hipLaunchKernelGGL((k_scores<T, true, XC>), grid, dim3(kDBlock), pad, s, q, Wg, gr.rowptr, gr.col, gr.val,
                       gr.erow, kW, lse, pq, E, gr.cpos, Ec, Bg, N, gr.nnz, alpha, qout, apW, apCo, apY, apZ, 1.0f);
  else
    hipLaunchKernelGGL((k_scores<T, false, XC>), grid, dim3(kDBlock), pad, s, q, Wg, gr.rowptr, gr.col, gr.val,
                       gr.erow, kW, lse, pq, E, gr.cpos, Ec, Bg, N, gr.nnz, alpha, qout, apW, apCo, apY, apZ, 1.0f);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

template <int T>
static int launch_scores_t(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse, float* pq,
                           float* E, float* Ec, int G, int Bg, int N, hipStream_t s, const float* x, const float* alpha,
                           int C, float* qout, const float* apW, int apCo, float* apY, float* apZ) {
  if (x != nullptr && C == 1) return launch_scores_x<T, 1>(gr, x, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, alpha, qout, apW, apCo, apY, apZ);
  if (x != nullptr && C == 3) return launch_scores_x<T, 3>(gr, x, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, alpha, qout, apW, apCo, apY, apZ);
  return launch_scores_x<T, 0>(gr, q, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, nullptr, nullptr);
}

// q inside the score kernel: with ONE input channel always (PEMSD7 / PEMSD8 / PEMSD3: -4.9 us per hot-path step); with three
// (PEMSD4 / PEMSD8) only together with the aggregate + projection tail -- on their own the three loads per staged float4 cost
// the kernel what the k_qonly launch costs (17.1 us against 12.1 + 4.7 at N = 307), with the tail a launch of 9.3 us goes too.
bool scores_take_x(int C, bool with_tail) { return C == 1 || (C == 3 && with_tail); }

// dense_bf16.hip: the same two passes on the bf16 / fp16 matrix core (split operands, fp32 accumulate)
size_t dense_split_scratch_bytes(int G, int N, int T);
int launch_scores_b(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse, float* pq, float* E,
                    float* Ec, int G, int Bg, int N, hipStream_t s, void* scratch);
int launch_bwd_dense_col_b(const msgat_graph_t& gr, const float* q, const float* kW, const float* lse, const float* delta,
                           const float* gE, float* dq, int G, int N, hipStream_t s, void* scratch);

// Which arithmetic the dense passes of a shape run in -- a function of the shape alone, so that forward and backward of one
// layer always agree (backward re-creates the forward's score bits).  The split form pays two small launches per pass for
// its operand images (~14 us at PEMSD7 size, where the pass itself goes 57 -> 44 us: a tie, profiles/r06/dense_split_lab.txt)
// and wins from there on: N = 8192, 9.0 -> 5.8 ms.  MSGAT_DENSE_SPLIT=0 / 1 forces the choice (A/B runs, small-N tests).
bool dense_split_selected(int N, int T) {
  static const int force = [] { const char* e = getenv("MSGAT_DENSE_SPLIT"); return e == nullptr ? -1 : (e[0] == '1' ? 1 : 0); }();
  if (T != 12 || force == 0) return false;
  return force == 1 || N >= 1536;
}
size_t dense_scratch_bytes(int G, int N, int T) { return dense_split_selected(N, T) ? dense_split_scratch_bytes(G, N, T) : 0; }

int launch_scores(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse,
                  float* pq, float* E, float* Ec, int G, int Bg, int N, int T, hipStream_t s, const float* x,
                  const float* alpha, int C, float* qout, const float* apW, int apCo, float* apY, float* apZ, void* scratch) {
  if (dense_split_selected(N, T)) {
    if (x != nullptr) return MSGAT_ERR_UNSUPPORTED;   // the caller computes q first (api.hip)
    if (scratch == nullptr) return MSGAT_ERR_WORKSPACE;
    return launch_scores_b(gr, q, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, scratch);
  }
  if (x != nullptr && (!scores_take_x(C, apZ != nullptr) || alpha == nullptr || qout == nullptr)) return MSGAT_ERR_UNSUPPORTED;
  if (apZ != nullptr && (x == nullptr || apW == nullptr || apCo <= 0)) return MSGAT_ERR_UNSUPPORTED;
  switch (T) {
    case 4: return launch_scores_t<4>(gr, q, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, x, alpha, C, qout, apW, apCo, apY, apZ);
    case 8: return launch_scores_t<8>(gr, q, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, x, alpha, C, qout, apW, apCo, apY, apZ);
    case 12: return launch_scores_t<12>(gr, q, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, x, alpha, C, qout, apW, apCo, apY, apZ);
    case 16: return launch_scores_t<16>(gr, q, Wg, kW, lse, pq, E, Ec, G, Bg, N, s, x, alpha, C, qout, apW, apCo, apY, apZ);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// The sparse part of the column pass: sum over the in-edges of column m of g_e kW[row_e], this lane's four timesteps.
// Four edges per trip with clamped, unconditional loads (the surplus masked by a zero coefficient): a trip is two
// dependent round trips (edge ids, then coefficient and kW row) behind the column extent, whatever the degree.  As one
// edge per iteration the loop was 2 x degree dependent round trips at the END of a kernel whose blocks are all resident
// -- pure tail.  Same order of the sum, same bits.
template <int T>
__device__ __forceinline__ float4 in_edge_term(const int* __restrict__ colptr, const int* __restrict__ crow,
                                               const int* __restrict__ cperm, const float* __restrict__ gEg,
                                               const float* __restrict__ kWg, int mcol, int quad) {
  const int c0 = colptr[mcol], c1 = colptr[mcol + 1];
  float4 sp = f4zero();
  for (int k = c0; k < c1; k += 4) {
    int ep[4], er[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = min(k + i, c1 - 1);
      ep[i] = cperm[kk];
      er[i] = crow[kk];
    }
    float ge[4];
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float raw = gEg[ep[i]];
      ge[i] = (k + i < c1) ? raw : 0.f;
      v[i] = reinterpret_cast<const float4*>(kWg + (size_t)er[i] * T)[quad];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sp.x = fmaf(ge[i], v[i].x, sp.x);
      sp.y = fmaf(ge[i], v[i].y, sp.y);
      sp.z = fmaf(ge[i], v[i].z, sp.z);
      sp.w = fmaf(ge[i], v[i].w, sp.w);
    }
  }
  return sp;
}

// ---- backward: dense column pass ------------------------------------------------------------------------
// Wave w owns columns m0 + 16w .. +15 (B operand of the score product: their q) and streams all rows
// through LDS: kW2 rows (A operand), delta*kW rows (payload) and lse2.  Score tile D[i = row][j = column]:
// lane (column j, quad) holds rows rb + 4*quad + r.  Payload tile D2[i = s][j = column] +=
// (delta kW)[row][s] P[row][column].
//   dq[m] += sum_{e into m} g_e kW[row_e]  -  sum_n 2^(kW2[n].q[m] - lse2[n]) delta[n] kW[n]
template <int T, bool F16P = true>   // F16P: see k_scores
__global__ __launch_bounds__(kDBlock) void k_bwd_dense_col(
    const float* __restrict__ q, const float* __restrict__ kW, const float* __restrict__ lse,
    const float* __restrict__ delta, const float* __restrict__ gE, const int* __restrict__ colptr,
    const int* __restrict__ crow, const int* __restrict__ cperm, float* __restrict__ dq, int N,
    int nnz, float one) {
  constexpr int T4 = T / 4;
  __shared__ float4 kwr4[kDMC * T4];        // [row][kW2(T)]
  __shared__ float4 dkr4[kDMC * kPS / 4];   // [row][delta*kW(T) | zeros]: fp32, the source of the payload planes
  __shared__ float4 lse4[kDMC / 4];         // [row] lse2 - 14 (+inf past the end): P is carried times 2^14
  __shared__ uint4 pl4[F16P ? kPlaneU4 : 1];   // payload planes of the staged rows (fp16 terms of delta kW, see build_payload_planes)
  __shared__ float wmax[kDWaves];
  const float* kwr = reinterpret_cast<const float*>(kwr4);
  const float* dkr = reinterpret_cast<const float*>(dkr4);
  float* lsew = reinterpret_cast<float*>(lse4);

  const int g = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, quad = lane >> 4;
  const int mcol = blockIdx.x * kDRows + 16 * wave + j;  // this lane's column (shared by its 4 quads)
  const bool valid = mcol < N;
  const float* qg = q + (size_t)g * N * T;
  const float* kWg = kW + (size_t)g * N * T;

  float bfrag[T4];  // B fragment: q[mcol][4*kk + quad]
#pragma unroll
  for (int kk = 0; kk < T4; ++kk) bfrag[kk] = valid ? qg[(size_t)mcol * T + 4 * kk + quad] : 0.f;

  f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da, dc = da, dd = da;  // payload accumulators (fp16 path: row tile a / b of a trip)
  int sexp = 100;                              // the planes carry delta kW * 2^sexp (running, see k_scores)
  constexpr float kPOff = F16P ? kPOffB : 0.f;
  if (F16P)
    for (int i = threadIdx.x; i < kPlaneU4; i += kDBlock) pl4[i] = make_uint4(0u, 0u, 0u, 0u);   // rows s >= T stay zero

  // register-prefetched staging (see k_scores): one row per lane per chunk
  static_assert(kDMC <= kDBlock, "at most one staged row per lane");
  float4 prek[T4];
  float pred = 0.f, prel = 0.f, premax = 0.f;
  auto prefetch = [&](int r0) {
    const int rows = min(kDMC, N - r0);
    const bool live = (int)threadIdx.x < rows;  // lanes >= kDMC never stage
    const int nr = r0 + (live ? (int)threadIdx.x : 0);
    const float keep = live ? 1.f : 0.f;  // multiply, not select: keeps the loads out of a branch
    const float4* kr = reinterpret_cast<const float4*>(kWg + (size_t)nr * T);
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = kr[t4];
      prek[t4] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
    }
    pred = delta[(size_t)g * N + nr] * keep;
    const float lv = lse[(size_t)g * N + nr];
    prel = live ? lv - kPOff : INFINITY;  // exp2(s - inf) = 0 for rows past the end
    premax = 0.f;
    if (F16P) {
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) premax = fmaxf(premax, max_abs4(prek[t4]));
      premax *= fabsf(pred);
    }
  };
  prefetch(0);
  for (int r0 = 0; r0 < N; r0 += kDMC) {
    const int rows = min(kDMC, N - r0);
    const int rows16 = (rows + 15) & ~15;
    prio_by_progress(r0, N);
    __syncthreads();  // every wave is done with the previous chunk
    if (threadIdx.x < kDMC) {
      const int i = threadIdx.x;
#pragma unroll
      for (int t4 = 0; t4 < kPS / 4; ++t4) {
        const float4 v = (t4 < T4) ? prek[t4 < T4 ? t4 : 0] : f4zero();
        if (t4 < T4) kwr4[i * T4 + t4] = make_float4(v.x * kLog2e, v.y * kLog2e, v.z * kLog2e, v.w * kLog2e);
        dkr4[i * (kPS / 4) + t4] = make_float4(v.x * pred, v.y * pred, v.z * pred, v.w * pred);
      }
      lsew[i] = prel;
    }
    if (F16P) {
      const float wm = wave_max(premax);
      if (lane == 0) wmax[wave] = wm;
    }
    __syncthreads();
    if (F16P) {   // the chunk's payload planes at the running scale (see k_scores)
      float cmax = wmax[0];
#pragma unroll
      for (int w = 1; w < kDWaves; ++w) cmax = fmaxf(cmax, wmax[w]);
      const int ec = payload_scale_exp(cmax);
      if (ec < sexp) {
        const float f = pow2i(ec - sexp);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { da[rr] *= f; db[rr] *= f; }
        sexp = ec;
      }
      build_payload_planes<T, false>(dkr, kPS, rows, pow2i(sexp), one, pl4, threadIdx.x, kDBlock);
      __syncthreads();
    }
    prefetch(min(r0 + kDMC, max(N - 1, 0) / kDMC * kDMC));  // next chunk (the last trip re-reads its own)
    // same products in the same k order as the forward's edge pass, accumulator starting at 0: the
    // score is re-created bit for bit, so 2^(s - lse2) equals the forward's softmax value (rows that
    // are one-hot on an edge cancel against the sparse term; a 1e-4 slip in the exponent would not).
    // two row tiles per trip (rows past the end carry lse = +inf and zero payload, so an odd last tile is harmless)
    for (int rb = 0; rb < rows16; rb += 32) {
      f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) S0 = mfma16(kwr[(rb + j) * T + 4 * kk + quad], bfrag[kk], S0);
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) S1 = mfma16(kwr[(rb + 16 + j) * T + 4 * kk + quad], bfrag[kk], S1);
      const int rq = rb + 4 * quad;            // this lane's rows: rq .. rq+3 and rq+16 .. rq+19
      const float4 l4 = lse4[rq >> 2], l5 = lse4[(rq + 16) >> 2];         // quad-uniform
      const float p0 = fast_exp2(S0[0] - l4.x), p1 = fast_exp2(S0[1] - l4.y);
      const float p2 = fast_exp2(S0[2] - l4.z), p3 = fast_exp2(S0[3] - l4.w);
      const float p4 = fast_exp2(S1[0] - l5.x), p5 = fast_exp2(S1[1] - l5.y);
      const float p6 = fast_exp2(S1[2] - l5.z), p7 = fast_exp2(S1[3] - l5.w);
      // A2[i = s][k] = the planes' (h | m) of rows rb + 4 quad .. +3 at s = j; B2[k][j = column] = (Ph | Pm), then (Pm | Ph)
      if (F16P) {
        const float pa[4] = {p0, p1, p2, p3}, pb[4] = {p4, p5, p6, p7};
        const uint4 Fa = split_p(pa, one), A2a = pl4[((rb >> 4) * 4 + quad) * 16 + j];
        da = mfma_h(A2a, Fa, da); da = mfma_h(A2a, swap_halves(Fa), da);
        const uint4 Fb = split_p(pb, one), A2b = pl4[((rb >> 4) * 4 + 4 + quad) * 16 + j];
        db = mfma_h(A2b, Fb, db); db = mfma_h(A2b, swap_halves(Fb), db);
      } else {   // A2[i = s][k = quad] = (delta kW)[row rq + rr][s = j]; B2[k = quad][j = column] = p[rr]
        da = mfma16(dkr[(rq + 0) * kPS + j], p0, da);
        db = mfma16(dkr[(rq + 1) * kPS + j], p1, db);
        dc = mfma16(dkr[(rq + 2) * kPS + j], p2, dc);
        dd = mfma16(dkr[(rq + 3) * kPS + j], p3, dd);
        da = mfma16(dkr[(rq + 16) * kPS + j], p4, da);
        db = mfma16(dkr[(rq + 17) * kPS + j], p5, db);
        dc = mfma16(dkr[(rq + 18) * kPS + j], p6, dc);
        dd = mfma16(dkr[(rq + 19) * kPS + j], p7, dd);
      }
    }
  }
  if (!valid || quad >= T4) return;
  const float unscale = F16P ? pow2i(-sexp - (int)kPOffB) : 1.f;   // the planes' scale and P's 2^14

  // D2[s = 4*quad + rr][column]: this lane owns dq[mcol][4*quad .. +3]; add the sparse in-edge term
  const float4 sp = in_edge_term<T>(colptr, crow, cperm, gE + (size_t)g * nnz, kWg, mcol, quad);
  float4* dst = reinterpret_cast<float4*>(dq + ((size_t)g * N + mcol) * T) + quad;
  float4 v = *dst;
  v.x += sp.x - ((da[0] + db[0]) + (dc[0] + dd[0])) * unscale;
  v.y += sp.y - ((da[1] + db[1]) + (dc[1] + dd[1])) * unscale;
  v.z += sp.z - ((da[2] + db[2]) + (dc[2] + dd[2])) * unscale;
  v.w += sp.w - ((da[3] + db[3]) + (dc[3] + dd[3])) * unscale;
  *dst = v;
}

// ---- backward column pass, 7 owner waves + 1 helper wave (see k_scores7) -------------------------------------------
// A block owns 112 columns; owner w streams rows [0, Ra) for its 16 columns, the helper the LAST rows [Ra, N) (at most
// 128, staged once) for all seven column groups.  The partial column sums simply add: owner total = own + helper's.
template <int T>
__global__ __launch_bounds__(kDBlock) void k_bwd_dense_col7(
    const float* __restrict__ q, const float* __restrict__ kW, const float* __restrict__ lse,
    const float* __restrict__ delta, const float* __restrict__ gE, const int* __restrict__ colptr,
    const int* __restrict__ crow, const int* __restrict__ cperm, float* __restrict__ dq, int N,
    int nnz, int Ra, float one) {
  constexpr int T4 = T / 4;
  constexpr int kOThreads = 64 * kHOwners;
  __shared__ float4 kwr4[kDMC * T4];        // owners' chunk: [row][kW2(T)]
  __shared__ float4 dkr4[kDMC * kPS / 4];   //                [row][delta*kW(T) | zeros]
  __shared__ float4 lse4[kDMC / 4];         //                [row] lse2 (+inf past the end)
  __shared__ float4 kwh4[kDMC * T4];        // the helper's rows [Ra, N), staged once
  __shared__ float4 lsh4[kDMC / 4];
  __shared__ uint4 pl4[kPlaneU4];           // payload planes of the owners' chunk / of the helper's rows (see k_bwd_dense_col)
  __shared__ uint4 ph4[kPlaneU4];
  __shared__ float wmax[kDWaves];
  __shared__ float qfs[kHRows][T];          // q rows of the block's columns (the helper's B fragments)
  __shared__ float hpay[kHRows][17];        // helper partial: D2[s][column]
  const float* kwr = reinterpret_cast<const float*>(kwr4);
  const float* dkr = reinterpret_cast<const float*>(dkr4);
  float* lsew = reinterpret_cast<float*>(lse4);
  const float* kwh = reinterpret_cast<const float*>(kwh4);
  const float* dkh = dkr;   // the helper's fp32 delta kW rows pass through the owners' staging buffer: they are only the source of ph4
  float* lshw = reinterpret_cast<float*>(lsh4);

  const int g = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, quad = lane >> 4;
  const bool owner = wave < kHOwners;
  const int m0c = blockIdx.x * kHRows;
  const int mcol = m0c + 16 * wave + j;
  const bool valid = owner && mcol < N;
  const float* qg = q + (size_t)g * N * T;
  const float* kWg = kW + (size_t)g * N * T;
  const int rowsh = N - Ra;   // 1 .. kDMC
  const int nchunk = cdiv(Ra, kDMC);

  // the owners' first chunk is requested before anything else (see k_scores7)
  // rows [0, Ra) in chunks, one staged row per lane of the first two owner waves (see k_bwd_dense_col)
  static_assert(kDMC <= kOThreads, "at most one staged row per owner lane");
  float4 prek[T4];
  float pred = 0.f, prel = 0.f, premax = 0.f;
  auto prefetch = [&](int r0) {
    const int rows = min(kDMC, Ra - r0);
    const bool live = (int)threadIdx.x < rows;
    const int nr = r0 + (live ? (int)threadIdx.x : 0);
    const float keep = live ? 1.f : 0.f;
    const float4* kr = reinterpret_cast<const float4*>(kWg + (size_t)nr * T);
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = kr[t4];
      prek[t4] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
    }
    pred = delta[(size_t)g * N + nr] * keep;
    const float lv = lse[(size_t)g * N + nr];
    prel = live ? lv - kPOffB : INFINITY;
    premax = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) premax = fmaxf(premax, max_abs4(prek[t4]));
    premax *= fabsf(pred);
  };
  if (owner) prefetch(0);

  // the helper's rows and the block's q rows, by every lane
  float hmaxabs = 0.f;
  if (threadIdx.x < kDMC) {
    const int i = threadIdx.x;
    const bool live = i < rowsh;
    const int nr = Ra + (live ? i : 0);
    const float keep = live ? 1.f : 0.f;
    const float pd = delta[(size_t)g * N + nr] * keep;
    const float lv = lse[(size_t)g * N + nr];
    const float4* kr = reinterpret_cast<const float4*>(kWg + (size_t)nr * T);
#pragma unroll
    for (int t4 = 0; t4 < kPS / 4; ++t4) {
      float4 v = f4zero();
      if (t4 < T4) {
        v = kr[t4 < T4 ? t4 : 0];
        v = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
        kwh4[i * T4 + t4] = make_float4(v.x * kLog2e, v.y * kLog2e, v.z * kLog2e, v.w * kLog2e);
      }
      const float4 w = make_float4(v.x * pd, v.y * pd, v.z * pd, v.w * pd);
      dkr4[i * (kPS / 4) + t4] = w;
      hmaxabs = fmaxf(hmaxabs, max_abs4(w));
    }
    lshw[i] = live ? lv - kPOffB : INFINITY;
  }
  for (int i = threadIdx.x; i < kPlaneU4; i += kDBlock) { pl4[i] = make_uint4(0u, 0u, 0u, 0u); ph4[i] = make_uint4(0u, 0u, 0u, 0u); }
  {
    const float wm = wave_max(hmaxabs);
    if (lane == 0) wmax[wave] = wm;
  }
  for (int i = threadIdx.x; i < kHRows * T; i += kDBlock) {
    const int cl = i / T, t = i - cl * T;
    qfs[cl][t] = (m0c + cl < N) ? qg[(size_t)(m0c + cl) * T + t] : 0.f;
  }

  __syncthreads();
  int sexp = 100, sexph;   // scale exponents of the owners' planes (running) and of the helper's (fixed)
  {
    float cmax = wmax[0];
#pragma unroll
    for (int w = 1; w < kDWaves; ++w) cmax = fmaxf(cmax, wmax[w]);
    sexph = payload_scale_exp(cmax);
    build_payload_planes<T, false>(dkh, kPS, rowsh, pow2i(sexph), one, ph4, threadIdx.x, kDBlock);
    __syncthreads();   // (wmax is re-used by the owners' chunks)
  }

  float bfrag[T4];
  if (owner) {
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) bfrag[kk] = valid ? qg[(size_t)mcol * T + 4 * kk + quad] : 0.f;
  }
  f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da;

  // one trip = two row tiles rb .. rb + 31 of the staged rows (k_bwd_dense_col's inner trip)
  auto trip = [&](const float* kw, const uint4* planes, const float4* l4s, int rb) {
    f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) S0 = mfma16(kw[(rb + j) * T + 4 * kk + quad], bfrag[kk], S0);
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) S1 = mfma16(kw[(rb + 16 + j) * T + 4 * kk + quad], bfrag[kk], S1);
    const int rq = rb + 4 * quad;
    const float4 l4 = l4s[rq >> 2], l5 = l4s[(rq + 16) >> 2];
    const float p0 = fast_exp2(S0[0] - l4.x), p1 = fast_exp2(S0[1] - l4.y);
    const float p2 = fast_exp2(S0[2] - l4.z), p3 = fast_exp2(S0[3] - l4.w);
    const float p4 = fast_exp2(S1[0] - l5.x), p5 = fast_exp2(S1[1] - l5.y);
    const float p6 = fast_exp2(S1[2] - l5.z), p7 = fast_exp2(S1[3] - l5.w);
    const float pa[4] = {p0, p1, p2, p3}, pb[4] = {p4, p5, p6, p7};   // see k_bwd_dense_col
    const uint4 Fa = split_p(pa, one), A2a = planes[((rb >> 4) * 4 + quad) * 16 + j];
    da = mfma_h(A2a, Fa, da); da = mfma_h(A2a, swap_halves(Fa), da);
    const uint4 Fb = split_p(pb, one), A2b = planes[((rb >> 4) * 4 + 4 + quad) * 16 + j];
    db = mfma_h(A2b, Fb, db); db = mfma_h(A2b, swap_halves(Fb), db);
  };

  if (owner) {
    // (the first chunk was requested at the top of the kernel)
    for (int r0 = 0; r0 < Ra; r0 += kDMC) {
      const int rows = min(kDMC, Ra - r0);
      const int rows16 = (rows + 15) & ~15;
      prio_by_progress(r0, Ra);
      __syncthreads();
      if (threadIdx.x < kDMC) {
        const int i = threadIdx.x;
#pragma unroll
        for (int t4 = 0; t4 < kPS / 4; ++t4) {
          const float4 v = (t4 < T4) ? prek[t4 < T4 ? t4 : 0] : f4zero();
          if (t4 < T4) kwr4[i * T4 + t4] = make_float4(v.x * kLog2e, v.y * kLog2e, v.z * kLog2e, v.w * kLog2e);
          dkr4[i * (kPS / 4) + t4] = make_float4(v.x * pred, v.y * pred, v.z * pred, v.w * pred);
        }
        lsew[i] = prel;
      }
      {
        const float wm = wave_max(premax);
        if (lane == 0) wmax[wave] = wm;
      }
      __syncthreads();
      {   // the chunk's payload planes at the running scale, by the owners' lanes
        float cmax = wmax[0];
#pragma unroll
        for (int w = 1; w < kHOwners; ++w) cmax = fmaxf(cmax, wmax[w]);
        const int ec = payload_scale_exp(cmax);
        if (ec < sexp) {
          const float f = pow2i(ec - sexp);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) { da[rr] *= f; db[rr] *= f; }
          sexp = ec;
        }
        build_payload_planes<T, false>(dkr, kPS, rows, pow2i(sexp), one, pl4, threadIdx.x, kOThreads);
        __syncthreads();
      }
      prefetch(min(r0 + kDMC, max(Ra - 1, 0) / kDMC * kDMC));
      for (int rb = 0; rb < rows16; rb += 32) trip(kwr, pl4, lse4, rb);
    }
  } else {
    const int rowsh16 = (rowsh + 15) & ~15;
    const float hf = pow2i(-sexph - (int)kPOffB);   // to true units
    int nbar = 0;
    for (int cg = 0; cg < kHOwners; ++cg) {
      if (nbar < 3 * nchunk) { __syncthreads(); __syncthreads(); __syncthreads(); nbar += 3; }   // the owners' three barriers per chunk
      prio_by_progress(cg, kHOwners);
#pragma unroll
      for (int kk = 0; kk < T4; ++kk) bfrag[kk] = qfs[16 * cg + j][4 * kk + quad];
      da = db = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int rb = 0; rb < rowsh16; rb += 32) trip(kwh, ph4, lsh4, rb);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) hpay[16 * cg + j][4 * quad + rr] = (da[rr] + db[rr]) * hf;
    }
    for (; nbar < 3 * nchunk; ++nbar) __syncthreads();
  }
  __syncthreads();   // the helper's partials are in LDS
  if (!valid || quad >= T4) return;

  const int cl = 16 * wave + j;
  const float4 sp = in_edge_term<T>(colptr, crow, cperm, gE + (size_t)g * nnz, kWg, mcol, quad);
  float4* dst = reinterpret_cast<float4*>(dq + ((size_t)g * N + mcol) * T) + quad;
  float4 v = *dst;
  const float unscale = pow2i(-sexp - (int)kPOffB);   // the own sums to true units (the helper's are already)
  v.x += sp.x - ((da[0] + db[0]) * unscale + hpay[cl][4 * quad + 0]);
  v.y += sp.y - ((da[1] + db[1]) * unscale + hpay[cl][4 * quad + 1]);
  v.z += sp.z - ((da[2] + db[2]) * unscale + hpay[cl][4 * quad + 2]);
  v.w += sp.w - ((da[3] + db[3]) * unscale + hpay[cl][4 * quad + 3]);
  *dst = v;
}

int launch_bwd_dense_col(const msgat_graph_t& gr, const float* q, const float* kW,
                         const float* lse, const float* delta, const float* gE, float* dq, int G,
                         int N, int T, hipStream_t s, void* scratch) {
  if (dense_split_selected(N, T)) {
    if (scratch == nullptr) return MSGAT_ERR_WORKSPACE;
    return launch_bwd_dense_col_b(gr, q, kW, lse, delta, gE, dq, G, N, s, scratch);
  }
#ifndef MSGAT_NO_SCORES7
  if (const int Ra = scores7_owner_columns(N, G)) {   // the same split, over rows
    dim3 grid7(cdiv(N, kHRows), G);
    const size_t lds7 = sizeof(float) * (2 * (kDMC * T + kDMC) + kDMC * kPS + kHRows * T + kHRows * 17 + kDWaves) +
                        2 * sizeof(uint4) * kPlaneU4;
    const size_t pad7 = balance_pad_bytes((int)(grid7.x * grid7.y), lds7);
#define MSGAT_DCOL7(TT)                                                                                    \
  hipLaunchKernelGGL(k_bwd_dense_col7<TT>, grid7, dim3(kDBlock), pad7, s, q, kW, lse, delta, gE, gr.colptr, \
                     gr.crow, gr.cperm, dq, N, gr.nnz, Ra, 1.0f)
    switch (T) {
      case 4: MSGAT_DCOL7(4); break;
      case 8: MSGAT_DCOL7(8); break;
      case 12: MSGAT_DCOL7(12); break;
      case 16: MSGAT_DCOL7(16); break;
      default: return MSGAT_ERR_UNSUPPORTED;
    }
#undef MSGAT_DCOL7
    MSGAT_CHECK_LAUNCH();
    return MSGAT_OK;
  }
#endif
  dim3 grid(cdiv(N, kDRows), G);
  const size_t static_lds = sizeof(float) * (kDMC * T + kDMC * kPS + kDMC + kDWaves) + sizeof(uint4) * kPlaneU4;
  const size_t pad = balance_pad_bytes((int)(grid.x * grid.y), static_lds);
  const bool f16p = (int)(grid.x * grid.y) > device_cu_count();   // see k_scores: F16P
#define MSGAT_DCOL(TT)                                                                                                       \
  if (f16p)                                                                                                                  \
    hipLaunchKernelGGL((k_bwd_dense_col<TT, true>), grid, dim3(kDBlock), pad, s, q, kW, lse, delta, gE, gr.colptr, gr.crow, \
                       gr.cperm, dq, N, gr.nnz, 1.0f);                                                                       \
  else                                                                                                                       \
    hipLaunchKernelGGL((k_bwd_dense_col<TT, false>), grid, dim3(kDBlock), pad, s, q, kW, lse, delta, gE, gr.colptr, gr.crow, \
                       gr.cperm, dq, N, gr.nnz, 1.0f)
  switch (T) {
    case 4: MSGAT_DCOL(4); break;
    case 8: MSGAT_DCOL(8); break;
    case 12: MSGAT_DCOL(12); break;
    case 16: MSGAT_DCOL(16); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_DCOL
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

}  // namespace msgat
