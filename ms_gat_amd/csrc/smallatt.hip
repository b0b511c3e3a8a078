// The two tiny attention matrices of a MEAM block, one launch each way.
//
//   channel attention (attention.py:88-94 + the 1x1 convolution of CACN, msgat.py:93-94)
//       p = node-pooled signals [C,T];  att = softmax_rows((p Wc) p^T) [C,C];  Mc = conv att [cb,C]
//       -- the per-sample channel matrix the mixing pass applies to the activation
//   temporal attention (attention.py:58-66) as the taps of TACN's first causal convolution (msgat.py:66-74)
//       q = channel-pooled signals [N,T];  left = q^T Wt1^T, right = q^T Wt2^T [T,K];  att = softmax_rows(left right^T)
//       taps[1] = att,  taps[0] = att shifted down by the dilation (rows < d zero) -- what k_tmix applies
//
// In PyTorch each is a chain of batched [12 x 12] / [72 x 72] matmuls, a softmax, transposes, pads and stacks:
// ~20 launches forward and ~30 backward per block, every one shorter than its launch overhead
// (profiles/r02/full_step_before_smallatt.txt: 205 of a step's 313 launches were such ops).  One workgroup per
// (relation, sample) group keeps the whole problem in LDS / registers; parameter gradients leave as one partial per
// group and are summed over the relation's groups in a fixed order (launch_reduce_groups): no atomics.
#include "common.hpp"

namespace msgat {

constexpr int kSaBlock = 256;

__device__ __forceinline__ float block_row_softmax_prep(const float* row, int n, float& inv) {
  float m = -3.0e38f;
  for (int i = 0; i < n; ++i) m = fmaxf(m, row[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += __expf(row[i] - m);
  inv = 1.0f / s;
  return m;
}

// ---- channel attention -----------------------------------------------------------------------------------------
// One block per group is 96 blocks for 256 CUs at PEMSD7 size, so the block is the machine: 1024 lanes (four waves per
// SIMD) over the [C x C] loops instead of 256 (one wave per SIMD, nothing to hide a latency behind): both blocks of a training step together forward 28.5 -> 23.7 us,
// backward 43.8 -> 31.2 us.
constexpr int kCaBlock = 1024;
template <int T>
__global__ __launch_bounds__(kCaBlock) void k_chanatt_fwd(const float* __restrict__ pooled,
                                                          const float* __restrict__ Wc,
                                                          const float* __restrict__ conv, float* __restrict__ att,
                                                          float* __restrict__ Mc, int Bg, int C, int cb) {
  extern __shared__ float sm[];
  float* p = sm;                 // [C][T]
  float* wc = p + C * T;         // [T][T]
  float* t1 = wc + T * T;        // [C][T] = p Wc
  float* S = t1 + C * T;         // [C][C+1]
  const int g = blockIdx.x, r = g / Bg, Cs = C + 1;
  float* cwl = S + C * Cs;       // [cb][C]: the convolution's weights (read C times per output below: from LDS, not through L1)
  for (int i = threadIdx.x; i < C * T; i += kCaBlock) p[i] = pooled[(size_t)g * C * T + i];
  for (int i = threadIdx.x; i < T * T; i += kCaBlock) wc[i] = Wc[(size_t)r * T * T + i];
  for (int i = threadIdx.x; i < cb * C; i += kCaBlock) cwl[i] = conv[(size_t)r * cb * C + i];
  __syncthreads();
  for (int i = threadIdx.x; i < C * T; i += kCaBlock) {
    const int c = i / T, s = i - c * T;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) a = fmaf(p[c * T + t], wc[t * T + s], a);
    t1[i] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += kCaBlock) {
    const int c = i / C, c2 = i - c * C;
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < T; ++s) a = fmaf(t1[c * T + s], p[c2 * T + s], a);
    S[c * Cs + c2] = a;
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {  // softmax over the row (C <= 256: one lane per row)
    float* row = S + threadIdx.x * Cs;
    float inv;
    const float m = block_row_softmax_prep(row, C, inv);
    for (int i = 0; i < C; ++i) row[i] = __expf(row[i] - m) * inv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += kCaBlock) att[(size_t)g * C * C + i] = S[(i / C) * Cs + (i % C)];
  for (int i = threadIdx.x; i < cb * C; i += kCaBlock) {
    const int o = i / C, c2 = i - o * C;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(cwl[o * C + c], S[c * Cs + c2], a);
    Mc[(size_t)g * cb * C + i] = a;
  }
}

template <int T>
__global__ __launch_bounds__(kCaBlock) void k_chanatt_bwd(const float* __restrict__ dMc,
                                                          const float* __restrict__ att,
                                                          const float* __restrict__ pooled,
                                                          const float* __restrict__ Wc,
                                                          const float* __restrict__ conv, float* __restrict__ dpooled,
                                                          float* __restrict__ dWc_part, float* __restrict__ dconv_part,
                                                          int Bg, int C, int cb) {
  extern __shared__ float sm[];
  const int Cs = C + 1;
  float* S = sm;                  // [C][C+1] att
  float* D = S + C * Cs;          // [C][C+1] d att, then dS
  float* p = D + C * Cs;          // [C][T]
  float* wc = p + C * T;          // [T][T]
  float* t1 = wc + T * T;         // [C][T]
  float* dt1 = t1 + C * T;        // [C][T]
  float* dm = dt1 + C * T;        // [cb][C]
  float* cwl = dm + cb * C;       // [cb][C]: the convolution's weights
  const int g = blockIdx.x, r = g / Bg;
  for (int i = threadIdx.x; i < C * C; i += kCaBlock) S[(i / C) * Cs + (i % C)] = att[(size_t)g * C * C + i];
  for (int i = threadIdx.x; i < C * T; i += kCaBlock) p[i] = pooled[(size_t)g * C * T + i];
  for (int i = threadIdx.x; i < T * T; i += kCaBlock) wc[i] = Wc[(size_t)r * T * T + i];
  for (int i = threadIdx.x; i < cb * C; i += kCaBlock) dm[i] = dMc[(size_t)g * cb * C + i];
  for (int i = threadIdx.x; i < cb * C; i += kCaBlock) cwl[i] = conv[(size_t)r * cb * C + i];
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += kCaBlock) {  // d att = conv^T dMc
    const int c = i / C, c2 = i - c * C;
    float a = 0.f;
    for (int o = 0; o < cb; ++o) a = fmaf(cwl[o * C + c], dm[o * C + c2], a);
    D[c * Cs + c2] = a;
  }
  for (int i = threadIdx.x; i < cb * C; i += kCaBlock) {  // d conv (this group's share) = dMc att^T
    const int o = i / C, c = i - o * C;
    float a = 0.f;
    for (int c2 = 0; c2 < C; ++c2) a = fmaf(dm[o * C + c2], S[c * Cs + c2], a);
    dconv_part[(size_t)g * cb * C + i] = a;
  }
  for (int i = threadIdx.x; i < C * T; i += kCaBlock) {
    const int c = i / T, s = i - c * T;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) a = fmaf(p[c * T + t], wc[t * T + s], a);
    t1[i] = a;
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {  // softmax backward, row by row: dS = att (d att - <d att, att>)
    float* d = D + threadIdx.x * Cs;
    const float* a = S + threadIdx.x * Cs;
    float dot = 0.f;
    for (int i = 0; i < C; ++i) dot = fmaf(d[i], a[i], dot);
    for (int i = 0; i < C; ++i) d[i] = a[i] * (d[i] - dot);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * T; i += kCaBlock) {  // d t1 = dS p
    const int c = i / T, s = i - c * T;
    float a = 0.f;
    for (int c2 = 0; c2 < C; ++c2) a = fmaf(D[c * Cs + c2], p[c2 * T + s], a);
    dt1[i] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * T; i += kCaBlock) {  // d p = dS^T t1 + d t1 Wc^T
    const int c = i / T, t = i - c * T;
    float a = 0.f;
    for (int c1 = 0; c1 < C; ++c1) a = fmaf(D[c1 * Cs + c], t1[c1 * T + t], a);
#pragma unroll
    for (int s = 0; s < T; ++s) a = fmaf(dt1[c * T + s], wc[t * T + s], a);
    dpooled[(size_t)g * C * T + i] = a;
  }
  for (int i = threadIdx.x; i < T * T; i += kCaBlock) {  // d Wc (this group's share) = p^T d t1
    const int t = i / T, s = i - t * T;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(p[c * T + t], dt1[c * T + s], a);
    dWc_part[(size_t)g * T * T + i] = a;
  }
}

static size_t chanatt_fwd_lds(int C, int cb, int T) {
  return sizeof(float) * (size_t)(2 * C * T + T * T + C * (C + 1) + cb * C);
}
static size_t chanatt_bwd_lds(int C, int cb, int T) {
  return sizeof(float) * (size_t)(2 * C * (C + 1) + 3 * C * T + T * T + 2 * cb * C);
}

size_t chanatt_partial_floats(int G, int C, int cb, int T) { return (size_t)G * ((size_t)T * T + (size_t)cb * C); }

// one grant record per kernel instantiation (the template argument makes the static distinct)
template <typename K, K kernel>
static int raise_lds(size_t lds) {
  static LdsGrant granted;
  return grant_dynamic_lds(kernel, lds, granted);
}

int launch_chanatt_fwd(const float* pooled, const float* Wc, const float* conv, float* att, float* Mc, int G, int R,
                       int C, int cb, int T, hipStream_t s) {
  const size_t lds = chanatt_fwd_lds(C, cb, T);
  if (lds > (size_t)kLdsMax - 1024) return MSGAT_ERR_UNSUPPORTED;
#define MSGAT_CA_FWD(TT)                                                                                      \
  {                                                                                                           \
    int st = raise_lds<decltype(&k_chanatt_fwd<TT>), &k_chanatt_fwd<TT>>(lds);                                                              \
    if (st) return st;                                                                                        \
    hipLaunchKernelGGL(k_chanatt_fwd<TT>, dim3(G), dim3(kCaBlock), lds, s, pooled, Wc, conv, att, Mc, G / R, C, cb); \
  }
  switch (T) {
    case 4: MSGAT_CA_FWD(4) break;
    case 8: MSGAT_CA_FWD(8) break;
    case 12: MSGAT_CA_FWD(12) break;
    case 16: MSGAT_CA_FWD(16) break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_CA_FWD
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_chanatt_bwd(const float* dMc, const float* att, const float* pooled, const float* Wc, const float* conv,
                       float* dpooled, float* dWc, float* dconv, float* part, int G, int R, int C, int cb, int T,
                       hipStream_t s) {
  const size_t lds = chanatt_bwd_lds(C, cb, T);
  if (lds > (size_t)kLdsMax - 1024) return MSGAT_ERR_UNSUPPORTED;
  float* pWc = part;                         // [G,T,T]
  float* pconv = part + (size_t)G * T * T;   // [G,cb,C]
#define MSGAT_CA_BWD(TT)                                                                                       \
  {                                                                                                            \
    int st = raise_lds<decltype(&k_chanatt_bwd<TT>), &k_chanatt_bwd<TT>>(lds);                                                               \
    if (st) return st;                                                                                         \
    hipLaunchKernelGGL(k_chanatt_bwd<TT>, dim3(G), dim3(kCaBlock), lds, s, dMc, att, pooled, Wc, conv, dpooled, pWc, \
                       pconv, G / R, C, cb);                                                                   \
  }
  switch (T) {
    case 4: MSGAT_CA_BWD(4) break;
    case 8: MSGAT_CA_BWD(8) break;
    case 12: MSGAT_CA_BWD(12) break;
    case 16: MSGAT_CA_BWD(16) break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_CA_BWD
  MSGAT_CHECK_LAUNCH();
  ReduceJobs jobs{};   // both sums in one launch
  if (int st = launch_reduce_groups_defer(pWc, R, G / R, T * T, dWc, s, &jobs)) return st;
  if (int st = launch_reduce_groups_defer(pconv, R, G / R, cb * C, dconv, s, &jobs)) return st;
  return launch_reduce_jobs(jobs, s);
}

// ---- temporal attention -> taps of the first causal convolution ---------------------------------------------------
constexpr int kTaRankMax = 16;
constexpr int kTaTile = 128;  // nodes per LDS tile of the projections (64: 14 trips = 26.2 us at N = 883; 128: 7 trips = 18.2 us)

template <int T>
__global__ __launch_bounds__(kSaBlock) void k_tempatt_fwd(const float* __restrict__ pooled,
                                                          const float* __restrict__ Wt1,
                                                          const float* __restrict__ Wt2, float* __restrict__ lr,
                                                          float* __restrict__ att, float* __restrict__ taps, int Bg,
                                                          int N, int K, int dil) {
  __shared__ float lrs[2 * T * kTaRankMax];  // left [T][K], right [T][K]
  __shared__ float S[T * (T + 1)];
  __shared__ float qt[2][kTaTile * T];                        // tiles of q: [node][T], double-buffered
  __shared__ float wt[2][2 * kTaRankMax * (kTaTile + 1)];     // tiles of both projections: [which*K + k][node], padded rows
  const int g = blockIdx.x, r = g / Bg;
  const float* q = pooled + (size_t)g * N * T;
  // one lane per (which, t, k): q[:,t] . Wt[k,:], the node axis walked in LDS tiles (both operands are re-used by
  // 2K resp. T lanes; read straight from global the loop is one dependent L1 round trip per node: 74-94 us per launch).
  // Round 5: the next tile's global loads are in flight while the current one is multiplied (two LDS buffers, one
  // barrier per tile) -- with load, barrier, multiply, barrier in sequence the 14 tiles of N = 883 were 14 exposed round
  // trips on one block per group (96 blocks for 256 CUs): 54 us per launch.
  const bool owner = (int)threadIdx.x < 2 * T * K;
  const int ow = owner ? threadIdx.x / (T * K) : 0, orem = owner ? threadIdx.x - ow * T * K : 0;
  const int ot = orem / K, ok = orem - ot * K;
  constexpr int kQn = (kTaTile * T + kSaBlock - 1) / kSaBlock;              // q elements a thread stages per tile
  constexpr int kWn = (2 * kTaRankMax * kTaTile + kSaBlock - 1) / kSaBlock;  // projection elements (at most)
  float qreg[kQn], wreg[kWn];
  auto fetch = [&](int n0) {   // every load unconditional (clamped), zero past the end of the node axis
    const int nn = min(kTaTile, N - n0);
#pragma unroll
    for (int u = 0; u < kQn; ++u) {
      const int i = threadIdx.x + u * kSaBlock;
      const float v = q[(size_t)n0 * T + min(i, max(nn * T - 1, 0))];
      qreg[u] = (i < nn * T) ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kWn; ++u) {
      const int i = min((int)threadIdx.x + u * kSaBlock, 2 * K * kTaTile - 1);
      const int row = i / kTaTile, n = i - row * kTaTile;
      const float* W = (row >= K ? Wt2 : Wt1) + ((size_t)r * K + (row >= K ? row - K : row)) * N;
      const float v = W[n0 + min(n, max(nn - 1, 0))];
      wreg[u] = (n < nn) ? v : 0.f;
    }
  };
  auto stash = [&](int b) {
#pragma unroll
    for (int u = 0; u < kQn; ++u) {
      const int i = threadIdx.x + u * kSaBlock;
      if (i < kTaTile * T) qt[b][i] = qreg[u];
    }
#pragma unroll
    for (int u = 0; u < kWn; ++u) {
      const int i = threadIdx.x + u * kSaBlock;
      if (i < 2 * K * kTaTile) wt[b][(i / kTaTile) * (kTaTile + 1) + (i % kTaTile)] = wreg[u];
    }
  };
  float acc = 0.f;
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int n0 = 0; n0 < N; n0 += kTaTile) {
    const int nnext = min(n0 + kTaTile, ((N - 1) / kTaTile) * kTaTile);   // clamped: the last trip re-reads its own tile
    fetch(nnext);
    if (owner) {
      const float* wr = wt[buf] + (ow * K + ok) * (kTaTile + 1);
      const float* qb = qt[buf];
      float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
      for (int n = 0; n < kTaTile; n += 2) {
        a0 = fmaf(qb[n * T + ot], wr[n], a0);
        a1 = fmaf(qb[(n + 1) * T + ot], wr[n + 1], a1);
      }
      acc += a0 + a1;
    }
    stash(buf ^ 1);      // the other buffer: nobody reads it during this trip
    __syncthreads();
    buf ^= 1;
  }
  if (owner) {
    lrs[threadIdx.x] = acc;
    lr[(size_t)g * 2 * T * K + threadIdx.x] = acc;
  }
  __syncthreads();
  if ((int)threadIdx.x < T * T) {
    const int t = threadIdx.x / T, t2 = threadIdx.x - t * T;
    float a = 0.f;
    for (int k = 0; k < K; ++k) a = fmaf(lrs[t * K + k], lrs[T * K + t2 * K + k], a);
    S[t * (T + 1) + t2] = a;
  }
  __syncthreads();
  if ((int)threadIdx.x < T) {
    float* row = S + threadIdx.x * (T + 1);
    float inv;
    const float m = block_row_softmax_prep(row, T, inv);
    for (int i = 0; i < T; ++i) row[i] = __expf(row[i] - m) * inv;
  }
  __syncthreads();
  if ((int)threadIdx.x < T * T) {
    const int t = threadIdx.x / T, i = threadIdx.x - t * T;
    const float a = S[t * (T + 1) + i];
    att[(size_t)g * T * T + threadIdx.x] = a;
    float* tp = taps + (size_t)g * 2 * T * T;
    tp[T * T + threadIdx.x] = a;                                              // tap 1: att
    tp[threadIdx.x] = (t >= dil) ? S[(t - dil) * (T + 1) + i] : 0.f;          // tap 0: att shifted down by the dilation
  }
}

// 1024 lanes: the node loop below is one lane per node with a dependent burst of loads per trip, and one block per group
// leaves 160 of 256 CUs idle anyway -- four waves per SIMD take N = 883 in one trip instead of four (27.7 -> 13.7 us).
constexpr int kTbBlock = 1024;

template <int T>
__global__ __launch_bounds__(kTbBlock) void k_tempatt_bwd(const float* __restrict__ dtaps,
                                                          const float* __restrict__ att,
                                                          const float* __restrict__ lr,
                                                          const float* __restrict__ pooled,
                                                          const float* __restrict__ Wt1,
                                                          const float* __restrict__ Wt2, float* __restrict__ dpooled,
                                                          float* __restrict__ dW1p, float* __restrict__ dW2p, int Bg,
                                                          int N, int K, int dil) {
  __shared__ float lrs[2 * T * kTaRankMax];
  __shared__ float D[T * (T + 1)];
  __shared__ float dl[2 * T * kTaRankMax];  // d left [T][K], d right [T][K]
  const int g = blockIdx.x, r = g / Bg;
  if ((int)threadIdx.x < 2 * T * K) lrs[threadIdx.x] = lr[(size_t)g * 2 * T * K + threadIdx.x];
  if ((int)threadIdx.x < T * T) {
    const int t = threadIdx.x / T, i = threadIdx.x - t * T;
    const float* tp = dtaps + (size_t)g * 2 * T * T;
    D[t * (T + 1) + i] = tp[T * T + threadIdx.x] + ((t + dil < T) ? tp[(t + dil) * T + i] : 0.f);
  }
  __syncthreads();
  if ((int)threadIdx.x < T) {
    float* d = D + threadIdx.x * (T + 1);
    const float* a = att + (size_t)g * T * T + threadIdx.x * T;
    float av[T];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < T; ++i) { av[i] = a[i]; dot = fmaf(d[i], av[i], dot); }
#pragma unroll
    for (int i = 0; i < T; ++i) d[i] = av[i] * (d[i] - dot);
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * T * K) {
    const int w = threadIdx.x / (T * K), rem = threadIdx.x - w * T * K;
    const int t = rem / K, k = rem - t * K;
    float a = 0.f;
    if (w == 0) {  // d left[t][k] = sum_t2 dS[t][t2] right[t2][k]
      for (int t2 = 0; t2 < T; ++t2) a = fmaf(D[t * (T + 1) + t2], lrs[T * K + t2 * K + k], a);
    } else {       // d right[t][k] = sum_t1 dS[t1][t] left[t1][k]
      for (int t1 = 0; t1 < T; ++t1) a = fmaf(D[t1 * (T + 1) + t], lrs[t1 * K + k], a);
    }
    dl[threadIdx.x] = a;
  }
  __syncthreads();
  const float* q = pooled + (size_t)g * N * T;
  const float* W1 = Wt1 + (size_t)r * K * N;
  const float* W2 = Wt2 + (size_t)r * K * N;
  for (int n = threadIdx.x; n < N; n += kTbBlock) {  // lane = node: its q row, its column of both projections
    float qv[T], dq[T];
#pragma unroll
    for (int t4 = 0; t4 < T / 4; ++t4) {
      const float4 v = reinterpret_cast<const float4*>(q + (size_t)n * T)[t4];
      qv[4 * t4] = v.x; qv[4 * t4 + 1] = v.y; qv[4 * t4 + 2] = v.z; qv[4 * t4 + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) dq[t] = 0.f;
    // the node's column of both projections, requested together (a load and its use per trip of the k loop was 2 K
    // dependent round trips per node)
    float w1v[kTaRankMax], w2v[kTaRankMax];
#pragma unroll
    for (int k = 0; k < kTaRankMax; ++k) {
      const int kc = min(k, K - 1);
      w1v[k] = W1[(size_t)kc * N + n];
      w2v[k] = W2[(size_t)kc * N + n];
    }
#pragma unroll
    for (int k = 0; k < kTaRankMax; ++k) {
      if (k >= K) break;                       // kernel-uniform
      const float w1 = w1v[k], w2 = w2v[k];
      float g1 = 0.f, g2 = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        dq[t] = fmaf(dl[t * K + k], w1, dq[t]);
        dq[t] = fmaf(dl[T * K + t * K + k], w2, dq[t]);
        g1 = fmaf(dl[t * K + k], qv[t], g1);
        g2 = fmaf(dl[T * K + t * K + k], qv[t], g2);
      }
      dW1p[((size_t)g * K + k) * N + n] = g1;
      dW2p[((size_t)g * K + k) * N + n] = g2;
    }
#pragma unroll
    for (int t4 = 0; t4 < T / 4; ++t4)
      reinterpret_cast<float4*>(dpooled + ((size_t)g * N + n) * T)[t4] =
          make_float4(dq[4 * t4], dq[4 * t4 + 1], dq[4 * t4 + 2], dq[4 * t4 + 3]);
  }
}

size_t tempatt_partial_floats(int G, int K, int N) { return (size_t)2 * G * K * N; }

int launch_tempatt_fwd(const float* pooled, const float* Wt1, const float* Wt2, float* lr, float* att, float* taps,
                       int G, int R, int N, int K, int T, int dil, hipStream_t s) {
  if (K < 1 || K > kTaRankMax || 2 * T * K > kSaBlock) return MSGAT_ERR_UNSUPPORTED;
  switch (T) {
    case 4: hipLaunchKernelGGL(k_tempatt_fwd<4>, dim3(G), dim3(kSaBlock), 0, s, pooled, Wt1, Wt2, lr, att, taps, G / R, N, K, dil); break;
    case 8: hipLaunchKernelGGL(k_tempatt_fwd<8>, dim3(G), dim3(kSaBlock), 0, s, pooled, Wt1, Wt2, lr, att, taps, G / R, N, K, dil); break;
    case 12: hipLaunchKernelGGL(k_tempatt_fwd<12>, dim3(G), dim3(kSaBlock), 0, s, pooled, Wt1, Wt2, lr, att, taps, G / R, N, K, dil); break;
    case 16: hipLaunchKernelGGL(k_tempatt_fwd<16>, dim3(G), dim3(kSaBlock), 0, s, pooled, Wt1, Wt2, lr, att, taps, G / R, N, K, dil); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_tempatt_bwd(const float* dtaps, const float* att, const float* lr, const float* pooled, const float* Wt1,
                       const float* Wt2, float* dpooled, float* dWt1, float* dWt2, float* part, int G, int R, int N,
                       int K, int T, int dil, hipStream_t s) {
  if (K < 1 || K > kTaRankMax || 2 * T * K > kSaBlock) return MSGAT_ERR_UNSUPPORTED;
  float* p1 = part;
  float* p2 = part + (size_t)G * K * N;
  switch (T) {
    case 4: hipLaunchKernelGGL(k_tempatt_bwd<4>, dim3(G), dim3(kTbBlock), 0, s, dtaps, att, lr, pooled, Wt1, Wt2, dpooled, p1, p2, G / R, N, K, dil); break;
    case 8: hipLaunchKernelGGL(k_tempatt_bwd<8>, dim3(G), dim3(kTbBlock), 0, s, dtaps, att, lr, pooled, Wt1, Wt2, dpooled, p1, p2, G / R, N, K, dil); break;
    case 12: hipLaunchKernelGGL(k_tempatt_bwd<12>, dim3(G), dim3(kTbBlock), 0, s, dtaps, att, lr, pooled, Wt1, Wt2, dpooled, p1, p2, G / R, N, K, dil); break;
    case 16: hipLaunchKernelGGL(k_tempatt_bwd<16>, dim3(G), dim3(kTbBlock), 0, s, dtaps, att, lr, pooled, Wt1, Wt2, dpooled, p1, p2, G / R, N, K, dil); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  ReduceJobs jobs{};   // both sums in one launch
  if (int st = launch_reduce_groups_defer(p1, R, G / R, K * N, dWt1, s, &jobs)) return st;
  if (int st = launch_reduce_groups_defer(p2, R, G / R, K * N, dWt2, s, &jobs)) return st;
  return launch_reduce_jobs(jobs, s);
}

}  // namespace msgat
