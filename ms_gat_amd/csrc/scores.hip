// The dense side of the attention: everything that needs ALL N columns of a row.
//
// attention.py:34 takes softmax over the full [N] row of S = (k Wg) q^T and only then
// (attention.py:36) multiplies by the adjacency, so the normaliser of every edge
// coefficient is a sum over all N columns.  Nothing N x N is ever written here:
//
//   forward   kW = q Wg;  lse[n] = log sum_m exp(kW[n].q[m])   (online max/sum, flash style)
//             pq[n] = sum_m softmax(S)[n,m] q[m]               (only when training)
//             E[e]  = exp(kW[row_e].q[col_e] - lse[row_e]) * adj_e      (edges only)
//   backward  dq[m] -= sum_n softmax(S)[n,m] delta[n] kW[n]    (the dense column term)
//
// K = T = 12 dot products feed an exp, so everything stays fp32 (bf16 MFMA would break
// the 1e-4 bar); the kernels are bound by fp32 FMA + v_exp_f32 issue, not by HBM.
#include "common.hpp"

namespace msgat {

constexpr int kRL = 2;          // rows (or columns) per lane: packed fp32 math + half the LDS broadcasts
constexpr int kRT = 64 * kRL;   // rows (or columns) per block
constexpr int kNS = 8;          // the 8 waves of a block split the reduction axis (VALU-bound: needs waves)
constexpr int kSBlock = 64 * kNS;  // lanes per block of the dense kernels
constexpr int kMC = 256;        // columns of q staged in LDS per step
constexpr int kSub = 8;         // columns per online-softmax update

// Two independent rows per lane as a pair of scalars.  Deliberately NOT a packed vector type:
// v_pk_fma_f32 issues at half rate on gfx950 (no throughput gain) and made hipcc route the LDS
// broadcasts through v_readfirstlane + SGPR operands, 12 extra instructions per column.
struct v2f {
  float x, y;
  __device__ __forceinline__ v2f& operator*=(const v2f& o) { x *= o.x; y *= o.y; return *this; }
  __device__ __forceinline__ v2f& operator+=(const v2f& o) { x += o.x; y += o.y; return *this; }
};
__device__ __forceinline__ v2f operator-(const v2f& a, const v2f& b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }
__device__ __forceinline__ v2f fma2(const v2f& a, const v2f& b, const v2f& c) { return v2f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ v2f max2(const v2f& a, const v2f& b) { return v2f{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ v2f exp2v(const v2f& a) { return v2f{fast_exp2(a.x), fast_exp2(a.y)}; }

// Lane l of every wave owns rows n0 + l and n0 + 64 + l (the two halves of a v2f); wave w
// takes columns j = 8w .. 8w+7 (mod 32) of each staged tile.  Per column: 3 LDS broadcasts of
// q[m], then 2 x 12 v_fma_f32 for the two scores and (training) 2 x 12 more for pq.
template <int T, bool WITH_PQ>
__global__ __launch_bounds__(kSBlock) void k_scores(
    const float* __restrict__ q, const float* __restrict__ Wg, const int* __restrict__ rowptr,
    const int* __restrict__ col, const float* __restrict__ val, const int* __restrict__ erow,
    float* __restrict__ kW, float* __restrict__ lse, float* __restrict__ pq, float* __restrict__ E,
    int Bg, int N, int nnz) {
  constexpr int T4 = T / 4;
  constexpr int kMG = 4;                                   // splits merged per phase
  constexpr int RED = kMG * kRT * (T + 2);                 // floats for the split merge
  constexpr int STAGE = kMC * T;                           // floats for a column tile
  __shared__ float4 pool4[(RED > STAGE ? RED : STAGE) / 4];  // column tile, then merge scratch
  __shared__ float kw2s[kRT][T];
  __shared__ float lse2s[kRT];
  float4* qs4 = pool4;
  float* red = reinterpret_cast<float*>(pool4);            // [kMG][kRT][T+2]

  const int g = blockIdx.y;
  const int r = g / Bg;
  const int lane = threadIdx.x & (kWave - 1);
  const int split = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * kRT;
  const float* qg = q + (size_t)g * N * T;
  const float* wg = Wg + (size_t)r * T * T;

  // the lane's two rows: q[n] and kW[n] = q[n] Wg (stored unscaled, used scaled by log2 e)
  v2f kw2[T];
#pragma unroll
  for (int h = 0; h < kRL; ++h) {
    const int n = n0 + h * 64 + lane;
    const bool valid = n < N;
    float qr[T];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      float4 v = f4zero();
      if (valid) v = reinterpret_cast<const float4*>(qg + (size_t)n * T)[t4];
      qr[4 * t4 + 0] = v.x; qr[4 * t4 + 1] = v.y; qr[4 * t4 + 2] = v.z; qr[4 * t4 + 3] = v.w;
    }
    float kw[T];
#pragma unroll
    for (int s = 0; s < T; ++s) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) a = fmaf(qr[t], wg[t * T + s], a);
      kw[s] = a;
    }
    if (split == 0 && valid) {
      float4* dst = reinterpret_cast<float4*>(kW + ((size_t)g * N + n) * T);
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) dst[t4] = make_float4(kw[4 * t4], kw[4 * t4 + 1], kw[4 * t4 + 2], kw[4 * t4 + 3]);
    }
#pragma unroll
    for (int s = 0; s < T; ++s) {  // scores in log2 units: exp(x) = 2^(x log2 e)
      if (h == 0) kw2[s].x = kw[s] * kLog2e; else kw2[s].y = kw[s] * kLog2e;
    }
  }

  v2f m = splat(-INFINITY), l = splat(0.f);
  v2f racc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) racc[t] = splat(0.f);

  for (int c0 = 0; c0 < N; c0 += kMC) {
    const int cols = min(kMC, N - c0);
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(qg + (size_t)c0 * T);
      for (int i = threadIdx.x; i < cols * T4; i += kSBlock) qs4[i] = src[i];
    }
    __syncthreads();
    for (int j = split * kSub; j < cols; j += kNS * kSub) {
      v2f s[kSub];
      v2f cm = splat(-INFINITY);
#pragma unroll
      for (int jj = 0; jj < kSub; ++jj) {
        v2f a = splat(-INFINITY);
        if (j + jj < cols) {  // wave-uniform
          const float4* qc = &qs4[(j + jj) * T4];
          a = splat(0.f);
          // the fma order (t ascending, acc last) is what k_bwd_dense_col reproduces bit for bit
#pragma unroll
          for (int t4 = 0; t4 < T4; ++t4) {
            const float4 v = qc[t4];
            a = fma2(kw2[4 * t4 + 0], splat(v.x), a);
            a = fma2(kw2[4 * t4 + 1], splat(v.y), a);
            a = fma2(kw2[4 * t4 + 2], splat(v.z), a);
            a = fma2(kw2[4 * t4 + 3], splat(v.w), a);
          }
        }
        s[jj] = a;
        cm = max2(cm, a);
      }
      const v2f mn = max2(m, cm);
      const v2f sc = exp2v(m - mn);  // m == -inf on the first update -> 0
      m = mn;
      l *= sc;
      if (WITH_PQ) {
#pragma unroll
        for (int t = 0; t < T; ++t) racc[t] *= sc;
      }
#pragma unroll
      for (int jj = 0; jj < kSub; ++jj) {
        if (j + jj < cols) {
          const v2f p = exp2v(s[jj] - mn);
          l += p;
          if (WITH_PQ) {
            const float4* qc = &qs4[(j + jj) * T4];
#pragma unroll
            for (int t4 = 0; t4 < T4; ++t4) {
              const float4 v = qc[t4];
              racc[4 * t4 + 0] = fma2(p, splat(v.x), racc[4 * t4 + 0]);
              racc[4 * t4 + 1] = fma2(p, splat(v.y), racc[4 * t4 + 1]);
              racc[4 * t4 + 2] = fma2(p, splat(v.z), racc[4 * t4 + 2]);
              racc[4 * t4 + 3] = fma2(p, splat(v.w), racc[4 * t4 + 3]);
            }
          }
        }
      }
    }
  }

  // merge the column splits of each row, kMG splits per phase through the (now dead) column tile
  float M = -INFINITY, L = 0.f;
  float R[T];
#pragma unroll
  for (int t = 0; t < T; ++t) R[t] = 0.f;
  for (int ph = 0; ph < kNS / kMG; ++ph) {
    __syncthreads();
    if (split / kMG == ph) {
#pragma unroll
      for (int h = 0; h < kRL; ++h) {
        float* dst = red + ((size_t)(split % kMG) * kRT + h * 64 + lane) * (T + 2);
#pragma unroll
        for (int t = 0; t < T; ++t) dst[t] = h == 0 ? racc[t].x : racc[t].y;
        dst[T] = h == 0 ? m.x : m.y;
        dst[T + 1] = h == 0 ? l.x : l.y;
      }
    }
    __syncthreads();
    if (threadIdx.x < kRT) {
      const float* r0 = red + (size_t)threadIdx.x * (T + 2);
      constexpr int SS = kRT * (T + 2);  // stride between splits
      float Mn = M;
#pragma unroll
      for (int i = 0; i < kMG; ++i) Mn = fmaxf(Mn, r0[i * SS + T]);
      const float w0 = fast_exp2(M - Mn);  // first phase: M = -inf -> 0 (a row always has >= 1 column)
      L *= w0;
      if (WITH_PQ) {
#pragma unroll
        for (int t = 0; t < T; ++t) R[t] *= w0;
      }
      M = Mn;
#pragma unroll
      for (int i = 0; i < kMG; ++i) {
        const float w = fast_exp2(r0[i * SS + T] - M);  // a split that saw no column has m = -inf -> 0
        L = fmaf(w, r0[i * SS + T + 1], L);
        if (WITH_PQ) {
#pragma unroll
          for (int t = 0; t < T; ++t) R[t] = fmaf(w, r0[i * SS + t], R[t]);
        }
      }
    }
  }
  if (threadIdx.x < kRT) {
    const int row = threadIdx.x;
    const int n = n0 + row;
    const float lse2 = M + fast_log2(L);
    lse2s[row] = lse2;
    if (n < N) {
      lse[(size_t)g * N + n] = lse2;  // kept in log2 units so backward re-creates the exponent bit for bit
      if (WITH_PQ) {
        const float inv = 1.0f / L;
        float4* dst = reinterpret_cast<float4*>(pq + ((size_t)g * N + n) * T);
#pragma unroll
        for (int t4 = 0; t4 < T4; ++t4)
          dst[t4] = make_float4(R[4 * t4] * inv, R[4 * t4 + 1] * inv, R[4 * t4 + 2] * inv, R[4 * t4 + 3] * inv);
      }
    }
  }
  if (split == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      kw2s[lane][t] = kw2[t].x;
      kw2s[64 + lane][t] = kw2[t].y;
    }
  }
  __syncthreads();

  // edge coefficients of this block's rows: one lane per CSR edge, coalesced over e
  const int e0 = rowptr[n0];
  const int e1 = rowptr[min(n0 + kRT, N)];
  for (int e = e0 + threadIdx.x; e < e1; e += kSBlock) {
    const int nl = erow[e] - n0;
    const float4* qm = reinterpret_cast<const float4*>(qg + (size_t)col[e] * T);
    float a = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = qm[t4];
      a = fmaf(kw2s[nl][4 * t4 + 0], v.x, a);
      a = fmaf(kw2s[nl][4 * t4 + 1], v.y, a);
      a = fmaf(kw2s[nl][4 * t4 + 2], v.z, a);
      a = fmaf(kw2s[nl][4 * t4 + 3], v.w, a);
    }
    E[(size_t)g * nnz + e] = fast_exp2(a - lse2s[nl]) * val[e];
  }
}

template <int T>
static int launch_scores_t(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW,
                           float* lse, float* pq, float* E, int G, int Bg, int N, hipStream_t s) {
  dim3 grid(cdiv(N, kRT), G);
  if (pq != nullptr)
    hipLaunchKernelGGL((k_scores<T, true>), grid, dim3(kSBlock), 0, s, q, Wg, gr.rowptr, gr.col, gr.val,
                       gr.erow, kW, lse, pq, E, Bg, N, gr.nnz);
  else
    hipLaunchKernelGGL((k_scores<T, false>), grid, dim3(kSBlock), 0, s, q, Wg, gr.rowptr, gr.col, gr.val,
                       gr.erow, kW, lse, pq, E, Bg, N, gr.nnz);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_scores(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse,
                  float* pq, float* E, int G, int Bg, int N, int T, hipStream_t s) {
  switch (T) {
    case 4: return launch_scores_t<4>(gr, q, Wg, kW, lse, pq, E, G, Bg, N, s);
    case 8: return launch_scores_t<8>(gr, q, Wg, kW, lse, pq, E, G, Bg, N, s);
    case 12: return launch_scores_t<12>(gr, q, Wg, kW, lse, pq, E, G, Bg, N, s);
    case 16: return launch_scores_t<16>(gr, q, Wg, kW, lse, pq, E, G, Bg, N, s);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// ---- backward: edge and row passes ---------------------------------------------------------
// k_edge_grad (one lane per edge, coalesced over e):  dE_e = sum of the per-chunk SDDMM partials,
//   g_e = E_e dE_e.
// k_bwd_row (one lane per row): delta_n = sum_e g_e;  dkW[n] = sum_e g_e (q[col_e] - pq[n]);
//   dq[n] = dkW[n] Wg^T  (the row-local part of dq).
__global__ __launch_bounds__(kBlock) void k_edge_grad(const float* __restrict__ dEp, int nchunks,
                                                      const float* __restrict__ E,
                                                      float* __restrict__ gE, int nnz) {
  const int g = blockIdx.y;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  const float* p = dEp + (size_t)g * nchunks * nnz + e;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int k = 0;
  for (; k + 4 <= nchunks; k += 4) {  // 4 independent loads in flight; fixed summation order
    a0 += p[(size_t)(k + 0) * nnz];
    a1 += p[(size_t)(k + 1) * nnz];
    a2 += p[(size_t)(k + 2) * nnz];
    a3 += p[(size_t)(k + 3) * nnz];
  }
  for (; k < nchunks; ++k) a0 += p[(size_t)k * nnz];
  gE[(size_t)g * nnz + e] = E[(size_t)g * nnz + e] * ((a0 + a1) + (a2 + a3));
}

template <int T>
__global__ __launch_bounds__(kBlock) void k_bwd_row(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ gE,
    const float* __restrict__ q, const float* __restrict__ pq, const float* __restrict__ Wg,
    float* __restrict__ delta, float* __restrict__ dkW, float* __restrict__ dq, int Bg, int N,
    int nnz) {
  constexpr int T4 = T / 4;
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= N) return;
  float d = 0.f;
  float dk[T], pr[T];
#pragma unroll
  for (int t4 = 0; t4 < T4; ++t4) {
    const float4 v = reinterpret_cast<const float4*>(pq + ((size_t)g * N + n) * T)[t4];
    pr[4 * t4 + 0] = v.x; pr[4 * t4 + 1] = v.y; pr[4 * t4 + 2] = v.z; pr[4 * t4 + 3] = v.w;
    dk[4 * t4 + 0] = 0.f; dk[4 * t4 + 1] = 0.f; dk[4 * t4 + 2] = 0.f; dk[4 * t4 + 3] = 0.f;
  }
  // dkW[n] = sum_e g_e q[col_e] - delta_n pq[n] = sum_e g_e (q[col_e] - pq[n]): subtracting
  // first keeps a saturated (one-hot) row exact -- pq[n] then equals q[col_e] bit for bit.
  // Two edges per trip: their index / weight loads are independent and issue together.
  const float* gEg = gE + (size_t)g * nnz;
  const float* qg = q + (size_t)g * N * T;
  const int e1 = rowptr[n + 1];
  for (int e = rowptr[n]; e < e1; e += 2) {
    const int eb = min(e + 1, e1 - 1);
    const float ga = gEg[e];
    const float gb = (e + 1 < e1) ? gEg[eb] : 0.f;
    const float4* qa = reinterpret_cast<const float4*>(qg + (size_t)col[e] * T);
    const float4* qb = reinterpret_cast<const float4*>(qg + (size_t)col[eb] * T);
    d += ga;
    d += gb;
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 va = qa[t4], vb = qb[t4];
      dk[4 * t4 + 0] = fmaf(ga, va.x - pr[4 * t4 + 0], dk[4 * t4 + 0]);
      dk[4 * t4 + 1] = fmaf(ga, va.y - pr[4 * t4 + 1], dk[4 * t4 + 1]);
      dk[4 * t4 + 2] = fmaf(ga, va.z - pr[4 * t4 + 2], dk[4 * t4 + 2]);
      dk[4 * t4 + 3] = fmaf(ga, va.w - pr[4 * t4 + 3], dk[4 * t4 + 3]);
      dk[4 * t4 + 0] = fmaf(gb, vb.x - pr[4 * t4 + 0], dk[4 * t4 + 0]);
      dk[4 * t4 + 1] = fmaf(gb, vb.y - pr[4 * t4 + 1], dk[4 * t4 + 1]);
      dk[4 * t4 + 2] = fmaf(gb, vb.z - pr[4 * t4 + 2], dk[4 * t4 + 2]);
      dk[4 * t4 + 3] = fmaf(gb, vb.w - pr[4 * t4 + 3], dk[4 * t4 + 3]);
    }
  }
  float4* dkdst = reinterpret_cast<float4*>(dkW + ((size_t)g * N + n) * T);
#pragma unroll
  for (int t4 = 0; t4 < T4; ++t4) dkdst[t4] = make_float4(dk[4 * t4], dk[4 * t4 + 1], dk[4 * t4 + 2], dk[4 * t4 + 3]);
  delta[(size_t)g * N + n] = d;
  const float* wg = Wg + (size_t)r * T * T;
  float out[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < T; ++s) a = fmaf(dk[s], wg[t * T + s], a);
    out[t] = a;
  }
  float4* dqdst = reinterpret_cast<float4*>(dq + ((size_t)g * N + n) * T);
#pragma unroll
  for (int t4 = 0; t4 < T4; ++t4) dqdst[t4] = make_float4(out[4 * t4], out[4 * t4 + 1], out[4 * t4 + 2], out[4 * t4 + 3]);
}

int launch_bwd_edge(const msgat_graph_t& gr, const float* dEp, int nchunks, const float* E,
                    const float* q, const float* pq, const float* Wg, float* gE, float* delta,
                    float* dkW, float* dq, int G, int Bg, int N, int T, hipStream_t s) {
  if (gr.nnz > 0) {
    dim3 ge(cdiv(gr.nnz, kBlock), G);
    hipLaunchKernelGGL(k_edge_grad, ge, dim3(kBlock), 0, s, dEp, nchunks, E, gE, gr.nnz);
    MSGAT_CHECK_LAUNCH();
  }
  dim3 grid(cdiv(N, kBlock), G);
#define MSGAT_ROW(TT)                                                                               \
  hipLaunchKernelGGL(k_bwd_row<TT>, grid, dim3(kBlock), 0, s, gr.rowptr, gr.col, gE, q, pq, Wg, delta, \
                     dkW, dq, Bg, N, gr.nnz)
  switch (T) {
    case 4: MSGAT_ROW(4); break;
    case 8: MSGAT_ROW(8); break;
    case 12: MSGAT_ROW(12); break;
    case 16: MSGAT_ROW(16); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_ROW
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- backward: dense column pass -----------------------------------------------------------
// One lane per column m; rows stream through LDS as records [kW2(T) | delta*kW(T) | lse2].
//   dq[m] += sum_{e into m} g_e kW[row_e]  -  sum_n exp(kW[n].q[m] - lse[n]) delta[n] kW[n]
constexpr int kRC = 128;  // rows staged per step

template <int T>
__global__ __launch_bounds__(kSBlock) void k_bwd_dense_col(
    const float* __restrict__ q, const float* __restrict__ kW, const float* __restrict__ lse,
    const float* __restrict__ delta, const float* __restrict__ gE, const int* __restrict__ colptr,
    const int* __restrict__ crow, const int* __restrict__ cperm, float* __restrict__ dq, int N,
    int nnz) {
  constexpr int T4 = T / 4;
  constexpr int REC4 = 2 * T4 + 1;  // float4s per row record
  constexpr int RED = kNS * kRT * T;
  constexpr int STAGE = kRC * REC4 * 4;
  __shared__ float4 pool4[(RED > STAGE ? RED : STAGE) / 4];  // row records, then merge scratch
  float4* rec4 = pool4;
  float* red = reinterpret_cast<float*>(pool4);              // [kNS][kRT][T]

  const int g = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int split = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * kRT;
  const float* qg = q + (size_t)g * N * T;
  const float* kWg = kW + (size_t)g * N * T;

  // the lane's two columns m0 + lane and m0 + 64 + lane
  v2f qc[T];
#pragma unroll
  for (int h = 0; h < kRL; ++h) {
    const int mcol = m0 + h * 64 + lane;
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      float4 v = f4zero();
      if (mcol < N) v = reinterpret_cast<const float4*>(qg + (size_t)mcol * T)[t4];
      if (h == 0) { qc[4 * t4].x = v.x; qc[4 * t4 + 1].x = v.y; qc[4 * t4 + 2].x = v.z; qc[4 * t4 + 3].x = v.w; }
      else        { qc[4 * t4].y = v.x; qc[4 * t4 + 1].y = v.y; qc[4 * t4 + 2].y = v.z; qc[4 * t4 + 3].y = v.w; }
    }
  }
  v2f acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = splat(0.f);

  for (int r0 = 0; r0 < N; r0 += kRC) {
    const int rows = min(kRC, N - r0);
    __syncthreads();
    for (int i = threadIdx.x; i < rows; i += kSBlock) {
      const int nr = r0 + i;
      const float4* kr = reinterpret_cast<const float4*>(kWg + (size_t)nr * T);
      const float d = delta[(size_t)g * N + nr];
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) {
        const float4 v = kr[t4];
        rec4[i * REC4 + t4] = make_float4(v.x * kLog2e, v.y * kLog2e, v.z * kLog2e, v.w * kLog2e);
        rec4[i * REC4 + T4 + t4] = make_float4(v.x * d, v.y * d, v.z * d, v.w * d);
      }
      rec4[i * REC4 + 2 * T4] = make_float4(lse[(size_t)g * N + nr], 0.f, 0.f, 0.f);
    }
    __syncthreads();
    for (int row = split; row < rows; row += kNS) {
      const float4* rec = &rec4[row * REC4];
      // same operands in the same fma order as k_scores: the score is re-created bit for
      // bit, so exp2(s - lse2) equals the forward's softmax value (rows that are one-hot on
      // an edge cancel against the sparse term; a 1e-4 slip in the exponent would not)
      v2f s = splat(0.f);
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) {
        const float4 v = rec[t4];
        s = fma2(splat(v.x), qc[4 * t4 + 0], s);
        s = fma2(splat(v.y), qc[4 * t4 + 1], s);
        s = fma2(splat(v.z), qc[4 * t4 + 2], s);
        s = fma2(splat(v.w), qc[4 * t4 + 3], s);
      }
      const v2f p = exp2v(s - splat(rec[2 * T4].x));
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) {
        const float4 v = rec[T4 + t4];
        acc[4 * t4 + 0] = fma2(p, splat(v.x), acc[4 * t4 + 0]);
        acc[4 * t4 + 1] = fma2(p, splat(v.y), acc[4 * t4 + 1]);
        acc[4 * t4 + 2] = fma2(p, splat(v.z), acc[4 * t4 + 2]);
        acc[4 * t4 + 3] = fma2(p, splat(v.w), acc[4 * t4 + 3]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < kRL; ++h) {
    float* dst = red + ((size_t)split * kRT + h * 64 + lane) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) dst[t] = h == 0 ? acc[t].x : acc[t].y;
  }
  __syncthreads();
  if (threadIdx.x >= kRT) return;
  const int mcol = m0 + threadIdx.x;
  if (mcol >= N) return;

  const float* r0p = red + (size_t)threadIdx.x * T;
  constexpr int SS = kRT * T;
  float tot[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < kNS; ++i) a += r0p[i * SS + t];  // fixed order
    tot[t] = a;
  }
  float sp[T];
#pragma unroll
  for (int t = 0; t < T; ++t) sp[t] = 0.f;
  for (int k = colptr[mcol]; k < colptr[mcol + 1]; ++k) {
    const float ge = gE[(size_t)g * nnz + cperm[k]];
    const float4* kr = reinterpret_cast<const float4*>(kWg + (size_t)crow[k] * T);
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = kr[t4];
      sp[4 * t4 + 0] = fmaf(ge, v.x, sp[4 * t4 + 0]);
      sp[4 * t4 + 1] = fmaf(ge, v.y, sp[4 * t4 + 1]);
      sp[4 * t4 + 2] = fmaf(ge, v.z, sp[4 * t4 + 2]);
      sp[4 * t4 + 3] = fmaf(ge, v.w, sp[4 * t4 + 3]);
    }
  }
  float4* dst = reinterpret_cast<float4*>(dq + ((size_t)g * N + mcol) * T);
#pragma unroll
  for (int t4 = 0; t4 < T4; ++t4) {
    float4 v = dst[t4];
    v.x += sp[4 * t4 + 0] - tot[4 * t4 + 0];
    v.y += sp[4 * t4 + 1] - tot[4 * t4 + 1];
    v.z += sp[4 * t4 + 2] - tot[4 * t4 + 2];
    v.w += sp[4 * t4 + 3] - tot[4 * t4 + 3];
    dst[t4] = v;
  }
}

int launch_bwd_dense_col(const msgat_graph_t& gr, const float* q, const float* kW,
                         const float* lse, const float* delta, const float* gE, float* dq, int G,
                         int N, int T, hipStream_t s) {
  dim3 grid(cdiv(N, kRT), G);
#define MSGAT_DCOL(TT)                                                                              \
  hipLaunchKernelGGL(k_bwd_dense_col<TT>, grid, dim3(kSBlock), 0, s, q, kW, lse, delta, gE, gr.colptr, \
                     gr.crow, gr.cperm, dq, N, gr.nnz)
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
