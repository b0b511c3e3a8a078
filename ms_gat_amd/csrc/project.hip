// Per-node channel mixing: the dense, streaming side of the hot path.
//
//   q[g,p]     = sum_c alpha[r,c] x[g,c,p]                 attention.py:33 (q == k)
//   u[g,o,p]   = sum_c W[r,o,c]  x[g,c,p]                  msgat.py:27, applied BEFORE the
//                                                          aggregation when C > Co
//   dx[g,c,p]  = sum_o W[r,o,c] du[g,o,p] + alpha[r,c] dq[g,p]   (backward of both)
//
// p runs over the flat [N*T] axis of one (group, channel) slab, which is contiguous in
// the reference's [B,C,N,T] layout, so a lane owns 4 consecutive positions (one 16-B
// load per channel) and the channel loop streams the slabs: HBM-bound, x is read once.
#include "common.hpp"

namespace msgat {

// ---- q only (AGG_FIRST / PLAIN) ------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_qonly(const float4* __restrict__ x4,
                                                  const float* __restrict__ alpha,
                                                  float4* __restrict__ q4, int Bg, int C, int P4) {
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int p4 = blockIdx.x * kBlock + threadIdx.x;
  if (p4 >= P4) return;
  const float4* src = x4 + (size_t)g * C * P4 + p4;
  float4 acc = f4zero();
  for (int c = 0; c < C; ++c) f4fma(alpha[r * C + c], src[(size_t)c * P4], acc);
  q4[(size_t)g * P4 + p4] = acc;
}

int launch_qonly(const float* x, const float* alpha, float* q, int G, int Bg, int C, int P,
                 hipStream_t s) {
  const int P4 = P / 4;
  dim3 grid(cdiv(P4, kBlock), G);
  hipLaunchKernelGGL(k_qonly, grid, dim3(kBlock), 0, s, (const float4*)x, alpha, (float4*)q, Bg, C, P4);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- AGG_FIRST backward with few input channels: dy = W^T dz and dW = dz y^T in one pass over dz ---------------
// z = W y with y = E x the aggregated input (C channels, msgat.py:27 after the aggregation when C <= Co).  The first
// MEAM of every component has C = 1 (PEMSD7/8) or 3 (PEMSD3/4), so dz -- Co channels -- is the only large operand:
//   dy[g,c,p] = sum_o W[r,o,c] dz[g,o,p]            dW[r,o,c] = sum_{g in r, p} dz[g,o,p] y[g,c,p]
// As two launches (k_project 24 -> 1 and the channel-pair contraction) dz was read twice: 25 + 25 us at PEMSD7
// size against one pass here.  Lane = 4 positions; the Co loop runs in chunks of 8 channels whose 8*C products are
// wave-reduced by shuffles right away and parked in LDS, so the register footprint does not grow with Co; one
// partial per block, fixed order everywhere.
constexpr int kAfChunk = 8;

// Sum over the 64 lanes of a wave, result in lane 63: six v_add_f32 with DPP operands (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- no LDS crossbar.
// The kernel reduces Co * C sums per wave this way; as __shfl_xor butterflies (ds_bpermute_b32 + address arithmetic per
// step) the reduction was most of its instructions: 24.5 us at PEMSD4 (C = 3) for 28 MB.  Fixed order.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return v;
}

// ONES: a virtual channel of ones behind y's C real ones -- column C of the partial is sum_p dz[o,p], the bias gradient of a
// 1x1 convolution with C <= 3 input channels (round 5: the first block's residual / tap / CACN convolutions take this
// kernel for their whole backward through msgat_contract_mix_segments; as contraction + projection they read the
// Co-channel gradient twice: 78 + 35 us for the 72-channel residual gradient at PEMSD7 size).
template <int C, bool ONES>
__global__ __launch_bounds__(kBlock) void k_aggfirst_bwd(const float4* __restrict__ dz4, const float* __restrict__ W,
                                                         const float4* __restrict__ y4, float4* __restrict__ dy4,
                                                         float* __restrict__ part, int Bg, int Co, int P4, int dzgs) {
  constexpr int CT = C + (ONES ? 1 : 0);    // columns of the weight-gradient partial
  extern __shared__ float lds[];
  float* Wl = lds;                        // [Co][C]
  float* red = lds + Co * C;              // [4 waves][Co*CT]
  const int g = blockIdx.y;
  const int r = g / Bg;
  for (int i = threadIdx.x; i < Co * C; i += kBlock) Wl[i] = W[(size_t)r * Co * C + i];
  const int p4 = blockIdx.x * kBlock + threadIdx.x;
  const float keep = p4 < P4 ? 1.f : 0.f;
  const int p4c = min(p4, P4 - 1);  // clamped, unconditional loads; lanes past the end contribute keep = 0
  float4 yv[CT], acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float4 v = y4[((size_t)g * C + c) * P4 + p4c];
    yv[c] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
    acc[c] = f4zero();
  }
  if (ONES) yv[CT - 1] = make_float4(keep, keep, keep, keep);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4* src = dz4 + (size_t)g * dzgs * P4 + p4c;  // dz may be a channel slice of a wider tensor
  for (int o0 = 0; o0 < Co; o0 += kAfChunk) {
    float4 d[kAfChunk];
#pragma unroll
    for (int k = 0; k < kAfChunk; ++k) d[k] = src[(size_t)min(o0 + k, Co - 1) * P4];
#pragma unroll
    for (int k = 0; k < kAfChunk; ++k) {
      const int o = min(o0 + k, Co - 1);  // a chunk past Co repeats the last channel; its results are not stored
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c < C && o0 + k < Co) f4fma(Wl[o * C + min(c, C - 1)], d[k], acc[min(c, C - 1)]);  // wave-uniform, no load inside
        const float w = wave_sum_to_lane63(f4dot(d[k], yv[c], 0.f));
        if (lane == 63 && o0 + k < Co) red[wave * Co * CT + o * CT + c] = w;
      }
    }
  }
  if (p4 < P4) {
#pragma unroll
    for (int c = 0; c < C; ++c) dy4[((size_t)g * C + c) * P4 + p4] = acc[c];
  }
  __syncthreads();
  float* out = part + ((size_t)g * gridDim.x + blockIdx.x) * (Co * CT);
  for (int i = threadIdx.x; i < Co * CT; i += kBlock)
    out[i] = (red[i] + red[Co * CT + i]) + (red[2 * Co * CT + i] + red[3 * Co * CT + i]);
}

int aggfirst_blocks(int P) { return cdiv(P / 4, kBlock); }

int launch_aggfirst_bwd(const float* dz, const float* W, const float* y, float* dy, float* part, float* dW, int G,
                        int Bg, int C, int Co, int P, hipStream_t s, ReduceJobs* defer, int dzgs, int ones) {
  const int P4 = P / 4, nb = aggfirst_blocks(P);
  const int CT = C + (ones ? 1 : 0);
  if (CT > kAggFirstMaxC) return MSGAT_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(Co * C + 4 * Co * CT) * sizeof(float);
  dim3 grid(nb, G);
#define MSGAT_AF(CC, ON)                                                                                          \
  hipLaunchKernelGGL((k_aggfirst_bwd<CC, ON>), grid, dim3(kBlock), lds, s, (const float4*)dz, W, (const float4*)y, \
                     (float4*)dy, part, Bg, Co, P4, dzgs > 0 ? dzgs : Co)
  switch (ones ? 10 + C : C) {
    case 1: MSGAT_AF(1, false); break;
    case 2: MSGAT_AF(2, false); break;
    case 3: MSGAT_AF(3, false); break;
    case 4: MSGAT_AF(4, false); break;
    case 11: MSGAT_AF(1, true); break;
    case 12: MSGAT_AF(2, true); break;
    case 13: MSGAT_AF(3, true); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_AF
  MSGAT_CHECK_LAUNCH();
  if (ones == 2) return launch_reduce_lastcol(part, G / Bg, Bg * nb, Co, CT, dW, s, defer);
  return launch_reduce_groups_defer(part, G / Bg, Bg * nb, Co * CT, dW, s, defer);
}

// ---- general channel projection --------------------------------------------------------
// One block = 256 lanes x 4 positions of one group, OT output channels (blockIdx.z picks
// the tile).  The [Ci x OT] slice of the matrix sits in LDS and is read as wave-uniform
// broadcasts (one ds_read_b128 feeds 4 output channels x 4 positions = 16 FMAs).
template <int OT, bool SEGS>
__global__ __launch_bounds__(kBlock) void k_project(
    SegList in, const float* __restrict__ M, int m_in_major,
    const float* __restrict__ qvec, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, SegList out, float4* __restrict__ q4, int Bg,
    int P4, MixEpilogue epi) {
  extern __shared__ float lds[];
  const int Ci = in.total(), Co = out.total();
  float* Ml = lds;            // [Ci][OT]
  float* ql = lds + Ci * OT;  // [Ci]
  const int g = blockIdx.y;
  const int r = g / Bg;
  const int o0 = blockIdx.z * OT;
  const bool do_q = (qvec != nullptr) && (blockIdx.z == 0);

  for (int i = threadIdx.x; i < Ci * OT; i += kBlock) {
    const int ci = i / OT, oo = i - ci * OT, o = o0 + oo;
    float w = 0.f;
    if (o < Co) w = m_in_major ? M[((size_t)r * Ci + ci) * Co + o] : M[((size_t)r * Co + o) * Ci + ci];
    Ml[i] = w;
  }
  for (int i = threadIdx.x; i < Ci; i += kBlock) ql[i] = do_q ? qvec[r * Ci + i] : 0.f;
  __syncthreads();

  const int p4 = blockIdx.x * kBlock + threadIdx.x;
  if (p4 >= P4) return;

  float4 acc[OT];
#pragma unroll
  for (int oo = 0; oo < OT; ++oo) acc[oo] = f4zero();
  float4 qa = f4zero();
#pragma unroll 4
  for (int ci = 0; ci < Ci; ++ci) {
    const float4 xv = reinterpret_cast<const float4*>(in.template row<SEGS>(g, ci, 4 * P4))[p4];
    const float4* wrow = reinterpret_cast<const float4*>(Ml + ci * OT);
#pragma unroll
    for (int o4 = 0; o4 < OT / 4; ++o4) {
      const float4 w = wrow[o4];
      f4fma(w.x, xv, acc[4 * o4 + 0]);
      f4fma(w.y, xv, acc[4 * o4 + 1]);
      f4fma(w.z, xv, acc[4 * o4 + 2]);
      f4fma(w.w, xv, acc[4 * o4 + 3]);
    }
    f4fma(ql[ci], xv, qa);
  }

  float4 ex = f4zero();
  if (addvec != nullptr) ex = extra4[(size_t)g * P4 + p4];
#pragma unroll
  for (int oo = 0; oo < OT; ++oo) {
    const int o = o0 + oo;
    if (o < Co) {
      float4 v = acc[oo];
      if (addvec != nullptr) f4fma(addvec[r * Co + o], ex, v);
      reinterpret_cast<float4*>(const_cast<float*>(out.template row<SEGS>(g, o, 4 * P4)))[p4] =
          epi.template apply<SEGS>(v, r, g, o, p4, P4);
    }
  }
  if (do_q) q4[(size_t)g * P4 + p4] = qa;
}

template <int OT>
static int launch_project_t(const SegList& in, const float* M, int m_in_major, const float* qvec,
                            const float* addvec, const float* extra, const SegList& out, float* q, int G,
                            int Bg, int P4, const MixEpilogue& epi, hipStream_t s) {
  const int Ci = in.total(), Co = out.total();
  dim3 grid(cdiv(P4, kBlock), G, cdiv(Co, OT));
  const size_t lds = (size_t)(Ci * OT + Ci) * sizeof(float);
  if (in.n > 1 || out.n > 1 || epi.add.n > 1)
    hipLaunchKernelGGL((k_project<OT, true>), grid, dim3(kBlock), lds, s, in, M, m_in_major, qvec, addvec,
                       (const float4*)extra, out, (float4*)q, Bg, P4, epi);
  else
    hipLaunchKernelGGL((k_project<OT, false>), grid, dim3(kBlock), lds, s, in, M, m_in_major, qvec, addvec,
                       (const float4*)extra, out, (float4*)q, Bg, P4, epi);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_project(const float* in, const float* M, int m_in_major, const float* qvec,
                   const float* addvec, const float* extra, float* out, float* q, int G, int Bg,
                   int Ci, int Co, int P, hipStream_t s) {
  return launch_project_epi(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, Ci, Co, P, MixEpilogue{}, s);
}

int launch_project_epi(const float* in, const float* M, int m_in_major, const float* qvec,
                       const float* addvec, const float* extra, float* out, float* q, int G, int Bg,
                       int Ci, int Co, int P, const MixEpilogue& epi, hipStream_t s) {
  return launch_project_seg(seg_single(in, Ci), M, m_in_major, qvec, addvec, extra, seg_single(out, Co), q, G, Bg, P,
                            epi, s);
}

int launch_project_seg(const SegList& in, const float* M, int m_in_major, const float* qvec, const float* addvec,
                       const float* extra, const SegList& out, float* q, int G, int Bg, int P,
                       const MixEpilogue& epi, hipStream_t s) {
  const int Ci = in.total(), Co = out.total();
  // matrix cores whenever there are enough output channels to fill a tile and the matrix fits LDS
  if (Co >= 8 && project_mfma_lds_bytes(Ci, Co, addvec != nullptr) <= (size_t)kProjLdsMax)
    return launch_project_mfma(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P, epi, s);
  const int P4 = P / 4;
  // VALU fallback (few outputs or a very large matrix): widest tile that divides the work evenly; 24 and 32 cover the reference's widths
  // (Co = 16/24/32 forward, C = 48/72/96 backward)
  if (Co % 24 == 0) return launch_project_t<24>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
  if (Co % 32 == 0) return launch_project_t<32>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
  if (Co % 16 == 0) return launch_project_t<16>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
  if (Co <= 4) return launch_project_t<4>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
  return launch_project_t<8>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
}

}  // namespace msgat
