// Backward, sparse bookkeeping between the SDDMM and the dense column pass: edge gradients and the
// row-local part of dq.  (The dense passes live in dense.hip.)
#include "common.hpp"

namespace msgat {

// ---- backward: edge and row passes ---------------------------------------------------------
// k_edge_grad (one lane per edge, coalesced over e):  dE_e = sum of the per-chunk SDDMM partials,
//   g_e = E_e dE_e.
// k_bwd_row (one lane per row): delta_n = sum_e g_e;  dkW[n] = sum_e g_e (q[col_e] - pq[n]);
//   dq[n] = dkW[n] Wg^T  (the row-local part of dq).
__global__ __launch_bounds__(kBlock) void k_edge_grad(const float* __restrict__ dEp, int nchunks,
                                                      const float* __restrict__ E, const int* __restrict__ epos,
                                                      int stride, float* __restrict__ gE, int nnz) {
  const int g = blockIdx.y;
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= nnz) return;
  // epos: the SDDMM ran on the SELL layout and left its partials in position order (chunk stride = n_pos)
  const float* p = dEp + (size_t)g * nchunks * stride + (epos != nullptr ? epos[e] : e);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int k = 0;
  for (; k + 4 <= nchunks; k += 4) {  // 4 independent loads in flight; fixed summation order
    a0 += p[(size_t)(k + 0) * stride];
    a1 += p[(size_t)(k + 1) * stride];
    a2 += p[(size_t)(k + 2) * stride];
    a3 += p[(size_t)(k + 3) * stride];
  }
  for (; k < nchunks; ++k) a0 += p[(size_t)k * stride];
  gE[(size_t)g * nnz + e] = E[(size_t)g * nnz + e] * ((a0 + a1) + (a2 + a3));
}

// the same for partials in CSC order (k_agg_sddmm): lane k reads Ec[k] and its partials coalesced and scatters
// g to the CSR edge cperm[k]
__global__ __launch_bounds__(kBlock) void k_edge_grad_csc(const float* __restrict__ dEp, int nchunks,
                                                          const float* __restrict__ Ec, const int* __restrict__ cperm,
                                                          float* __restrict__ gE, int nnz) {
  const int g = blockIdx.y;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  const float* p = dEp + (size_t)g * nchunks * nnz + k;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
    a0 += p[(size_t)(c + 0) * nnz];
    a1 += p[(size_t)(c + 1) * nnz];
    a2 += p[(size_t)(c + 2) * nnz];
    a3 += p[(size_t)(c + 3) * nnz];
  }
  for (; c < nchunks; ++c) a0 += p[(size_t)c * nnz];
  gE[(size_t)g * nnz + cperm[k]] = Ec[(size_t)g * nnz + k] * ((a0 + a1) + (a2 + a3));
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
                    float* dkW, float* dq, int G, int Bg, int N, int T, hipStream_t s, const float* Ecsc) {
  if (gr.nnz > 0 && Ecsc != nullptr) {
    hipLaunchKernelGGL(k_edge_grad_csc, dim3(cdiv(gr.nnz, kBlock), G), dim3(kBlock), 0, s, dEp, nchunks, Ecsc, gr.cperm, gE,
                       gr.nnz);
    MSGAT_CHECK_LAUNCH();
  } else if (gr.nnz > 0) {
    dim3 ge(cdiv(gr.nnz, kBlock), G);
    const bool sell = sell_usable(gr.sell_rows, gr.nnz, N, T);
    hipLaunchKernelGGL(k_edge_grad, ge, dim3(kBlock), 0, s, dEp, nchunks, E, sell ? gr.sell_rows.pos : nullptr,
                       sell ? gr.sell_rows.n_pos : gr.nnz, gE, gr.nnz);
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

}  // namespace msgat
