// Backward, sparse bookkeeping between the SDDMM and the dense column pass: edge gradients and the
// row-local part of dq.  (The dense passes live in dense.hip.)
#include "common.hpp"

namespace msgat {

// ---- backward: the edge and row work of a group in ONE launch -----------------------------------------------------
// Per row n (one lane per row, a block of kRowBlock rows):
//   g_e      = E_e dE_e for the row's edges:
//                DC = 0: read from gE, where the edge-parallel pass below left it (sum of the SDDMM's chunk partials);
//                DC > 0: computed here -- dE_e = sum_{c < DC, t} dv[c,n,t] u[c,col_e,t] -- when the attention acted on
//                        DC <= 4 channels (the first MEAM of every component): no SDDMM launch, no partial buffer,
//                        no edge pass;
//   delta_n  = sum_e g_e;   dkW[n] = sum_e g_e (q[col_e] - pq[n]);   dq[n] = dkW[n] Wg^T  (the row-local part of dq);
//   dWg partial of the block = sum_rows q[n]^T dkW[n]  (summed over blocks in a fixed order by the caller's reduction).
// Round 2 ran this as k_edge_grad + k_bwd_row + k_dwg, plus k_sddmm for the DC > 0 case: 3-4 launches of 5-11 us each
// per GACN depth, none of which filled the chip.
constexpr int kRowBlock = 256;

// g_e = E_e (sum of the SDDMM's per-chunk partials), one lane per edge, coalesced over the partials' own order:
// CSR / SELL positions (k_edge_grad) or CSC (k_edge_grad_csc, which scatters g to the CSR edge cperm[k]).  Folding
// this into the row pass (a lane walking its row's edges through cpos / sell.pos, 8 chunk partials each) turned 9 us
// of coalesced reads into 40 us of dependent scattered ones.
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
  // eight independent loads in flight per trip (the fused pass leaves one partial per channel: 24 at msgat72's widths, so
  // four per trip were six dependent round trips); fixed summation order
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float ec = Ec[(size_t)g * nnz + k];
  const int dst = cperm[k];
  int c = 0;
  for (; c + 8 <= nchunks; c += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] += p[(size_t)(c + i) * nnz];
  }
  for (; c < nchunks; ++c) a[0] += p[(size_t)c * nnz];
  gE[(size_t)g * nnz + dst] = ec * (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
}


template <int T, int DC>
__global__ __launch_bounds__(kRowBlock) void k_bwd_rows(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ E,
    const float* __restrict__ u, const float* __restrict__ dv,
    const float* __restrict__ q, const float* __restrict__ pq, const float* __restrict__ Wg,
    float* gE, float* __restrict__ delta, float* __restrict__ dkW, float* __restrict__ dq,
    float* __restrict__ dwg_part, int Bg, int N, int nnz, int nblk) {
  constexpr int T4 = T / 4;
  constexpr int DCn = DC > 0 ? DC : 1;
  __shared__ float qs[kRowBlock * T];
  __shared__ float ds[kRowBlock * T];
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int n = blockIdx.x * kRowBlock + threadIdx.x;
  const bool valid = n < N;
  const int nc = valid ? n : 0;
  const float* qg = q + (size_t)g * N * T;
  float d = 0.f;
  float dk[T], pr[T];
#pragma unroll
  for (int t4 = 0; t4 < T4; ++t4) {
    const float4 v = reinterpret_cast<const float4*>(pq + ((size_t)g * N + nc) * T)[t4];
    pr[4 * t4 + 0] = v.x; pr[4 * t4 + 1] = v.y; pr[4 * t4 + 2] = v.z; pr[4 * t4 + 3] = v.w;
    dk[4 * t4 + 0] = 0.f; dk[4 * t4 + 1] = 0.f; dk[4 * t4 + 2] = 0.f; dk[4 * t4 + 3] = 0.f;
  }
  float4 dvr[DCn][T4];  // DC > 0: this row of dv, all DC channels
  if (DC > 0) {
#pragma unroll
    for (int c = 0; c < DCn; ++c)
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4)
        dvr[c][t4] = reinterpret_cast<const float4*>(dv + (((size_t)g * DC + c) * N + nc) * T)[t4];
  }
  // the gradient of one edge coefficient
  auto edge_grad = [&](int e, int ce) -> float {
    float dE;
    if (DC > 0) {
      dE = 0.f;
#pragma unroll
      for (int c = 0; c < DCn; ++c) {
        const float4* um = reinterpret_cast<const float4*>(u + (((size_t)g * DC + c) * N + ce) * T);
#pragma unroll
        for (int t4 = 0; t4 < T4; ++t4) dE = f4dot(dvr[c][t4], um[t4], dE);
      }
    } else {
      return gE[(size_t)g * nnz + e];  // left by k_edge_grad / k_edge_grad_csc
    }
    return E[(size_t)g * nnz + e] * dE;
  };
  // dkW[n] = sum_e g_e q[col_e] - delta_n pq[n] = sum_e g_e (q[col_e] - pq[n]): subtracting
  // first keeps a saturated (one-hot) row exact -- pq[n] then equals q[col_e] bit for bit.
  // Four edges per trip, every load unconditional (indices clamped to the row's last edge, the surplus masked by a
  // zero coefficient): a trip is two dependent round trips -- column indices, then coefficients and q / u rows -- and a
  // wave takes ceil(max degree / 4) of them.  With one or two edges per trip the PEMS rows (1 + ~Poisson(2) edges, some
  // row of every wave has 6-8) cost 3-4 trips; the sums run in edge order either way, so the bits do not change.
  const int e0 = valid ? rowptr[n] : 0;
  const int e1 = valid ? rowptr[n + 1] : 0;
  for (int e = e0; e < e1; e += 4) {
    int ee[4], cc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ee[i] = min(e + i, e1 - 1);
      cc[i] = col[ee[i]];
    }
    float gg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float raw = edge_grad(ee[i], cc[i]);
      gg[i] = (e + i < e1) ? raw : 0.f;
    }
    if (DC > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (e + i < e1) gE[(size_t)g * nnz + ee[i]] = gg[i];
    }
    float4 qv[4][T4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) qv[i][t4] = reinterpret_cast<const float4*>(qg + (size_t)cc[i] * T)[t4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d += gg[i];
#pragma unroll
      for (int t4 = 0; t4 < T4; ++t4) {
        dk[4 * t4 + 0] = fmaf(gg[i], qv[i][t4].x - pr[4 * t4 + 0], dk[4 * t4 + 0]);
        dk[4 * t4 + 1] = fmaf(gg[i], qv[i][t4].y - pr[4 * t4 + 1], dk[4 * t4 + 1]);
        dk[4 * t4 + 2] = fmaf(gg[i], qv[i][t4].z - pr[4 * t4 + 2], dk[4 * t4 + 2]);
        dk[4 * t4 + 3] = fmaf(gg[i], qv[i][t4].w - pr[4 * t4 + 3], dk[4 * t4 + 3]);
      }
    }
  }
  if (valid) {
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
  // dWg partial of this block: rows past N contribute zeros (their dk is zero)
#pragma unroll
  for (int t4 = 0; t4 < T4; ++t4) {
    const float4 v = reinterpret_cast<const float4*>(qg + (size_t)nc * T)[t4];
    reinterpret_cast<float4*>(qs + threadIdx.x * T)[t4] = v;
    reinterpret_cast<float4*>(ds + threadIdx.x * T)[t4] = make_float4(dk[4 * t4], dk[4 * t4 + 1], dk[4 * t4 + 2], dk[4 * t4 + 3]);
  }
  __syncthreads();
  // D[t][s] = sum_rows q[row][t] dkW[row][s] on the matrix core: each of the four waves contracts its 64 rows (16 k-steps of
  // four rows: lane (i, kq) supplies q[row 4k + kq][i] and dkW[row 4k + kq][i]), the four tiles are summed in a fixed order.
  // As 144 lanes walking all 256 rows (two LDS reads per fma, one dependent chain per lane) this tail cost 2.6-3 us of a
  // 13-18 us launch (lab build without it, round 4).
  {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int ic = i < T ? i : 0;
    const float keep = i < T ? 1.f : 0.f;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = 64 * wave + 4 * k + kq;
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[row * T + ic] * keep, ds[row * T + ic] * keep, d, 0, 0, 0);
    }
    __syncthreads();          // everyone is done reading qs: it now carries the four tiles
    static_assert(kRowBlock == 256 && kRowBlock * 4 >= 4 * 256, "four waves, one 16x16 tile each");
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) qs[wave * 256 + reg * 64 + lane] = d[reg];
    __syncthreads();
    const int e = threadIdx.x;                       // element (reg = e >> 6, lane = e & 63) of the tile
    const int t = 4 * ((e & 63) >> 4) + (e >> 6);    // D row 4 quad + reg
    const int sc = e & 15;                           // D column
    const float v = (qs[e] + qs[256 + e]) + (qs[512 + e] + qs[768 + e]);
    if (t < T && sc < T) dwg_part[((size_t)g * nblk + blockIdx.x) * (T * T) + t * T + sc] = v;
  }
}

size_t dwg_partial_floats(int G, int N, int T) { return (size_t)G * cdiv(N, kRowBlock) * T * T; }
int bwd_rows_direct_max_channels() { return 4; }

// Ecsc != nullptr: the SDDMM partials are in CSC order (k_agg_sddmm) and Ecsc holds E in that order; otherwise they
// are in CSR order, or in the SELL position order when the graph carries a usable sell_rows.  direct_c > 0: no
// partials at all, dE from u / dv with direct_c channels.
int launch_bwd_rows(const msgat_graph_t& gr, const float* dEp, int nchunks, const float* Ecsc, int direct_c,
                    const float* u, const float* dv, const float* E, const float* q, const float* pq, const float* Wg,
                    float* gE, float* delta, float* dkW, float* dq, float* dwg_part, float* dWg, int G, int Bg, int N,
                    int T, hipStream_t s, ReduceJobs* defer) {
  const int nblk = cdiv(N, kRowBlock);
  dim3 grid(nblk, G);
  const bool partials_in_csc = Ecsc != nullptr;
  if (direct_c == 0 && gr.nnz > 0) {
    dim3 ge(cdiv(gr.nnz, kBlock), G);
    if (partials_in_csc) {   // k_agg_sddmm: partials in CSC order; E itself is read at cperm[k] (== Ec[k])
      hipLaunchKernelGGL(k_edge_grad_csc, ge, dim3(kBlock), 0, s, dEp, nchunks, Ecsc, gr.cperm, gE, gr.nnz);
    } else {
      const bool sell = sell_usable(gr.sell_rows, gr.nnz, N, T);
      hipLaunchKernelGGL(k_edge_grad, ge, dim3(kBlock), 0, s, dEp, nchunks, E, sell ? gr.sell_rows.pos : nullptr,
                         sell ? gr.sell_rows.n_pos : gr.nnz, gE, gr.nnz);
    }
    MSGAT_CHECK_LAUNCH();
  }
#define MSGAT_ROWS(TT, DC)                                                                                          \
  hipLaunchKernelGGL((k_bwd_rows<TT, DC>), grid, dim3(kRowBlock), 0, s, gr.rowptr, gr.col, E, u, dv, q, pq, Wg, gE,  \
                     delta, dkW, dq, dwg_part, Bg, N, gr.nnz, nblk)
#define MSGAT_ROWS_T(TT)                        \
  switch (direct_c) {                           \
    case 0: MSGAT_ROWS(TT, 0); break;           \
    case 1: MSGAT_ROWS(TT, 1); break;           \
    case 2: MSGAT_ROWS(TT, 2); break;           \
    case 3: MSGAT_ROWS(TT, 3); break;           \
    case 4: MSGAT_ROWS(TT, 4); break;           \
    default: return MSGAT_ERR_UNSUPPORTED;      \
  }
  switch (T) {
    case 4: MSGAT_ROWS_T(4); break;
    case 8: MSGAT_ROWS_T(8); break;
    case 12: MSGAT_ROWS_T(12); break;
    case 16: MSGAT_ROWS_T(16); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_ROWS_T
#undef MSGAT_ROWS
  MSGAT_CHECK_LAUNCH();
  return launch_reduce_groups_defer(dwg_part, G / Bg, Bg * nblk, T * T, dWg, s, defer);
}

}  // namespace msgat
