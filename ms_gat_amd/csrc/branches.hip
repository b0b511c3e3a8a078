// The temporal and channel branches of MEAM: the streaming kernels that
// the hot path's channel-mixing kernels (project.hip / mfma.hip) do not already cover.
//
// Reference: TemporalAttention /root/reference/src/models/attention.py:58-66, ChannelAttention :88-94,
// TACN src/models/msgat.py:57-80 (attention, then causal dilated [1,2] convolutions), CACN :83-100.
//
// Both branches factor into (a) tiny per-sample attention matrices ([T,T], [C,C]) built from pooled
// signals, and (b) applying them to the whole [B,C,N,T] activation.  (b) is what costs: in eager
// PyTorch it is a batched matmul with K = 12 or M = 72 (hipBLASLt at 5-10% of HBM speed) between
// layout copies.  Here:
//   channel side  out = (Wconv att_b) x_b + bias      one channel-mixing pass with a per-sample matrix
//                                                     (k_project_mfma, R = B "relations")
//   temporal side conv_d(TA(x)) = sum_k (W_k x) x_t A_k[b]^T   with A_1 = att_b, A_0 = att_b shifted down
//                 by the dilation: channel mixing FIRST (C -> 2 Co, shared matrix), then ONE pass that
//                 applies the two [T,T] matrices along time and adds the taps (k_tmix below).  Later
//                 convolutions of the stack use the same kernels with constant shift matrices.
//   pooling       pooled[b,c,:] = sum_n w_n x[b,c,n,:]  (k_node_pool);  sum_c alpha_c x[b,c] is k_qonly.
// Every kernel reads and writes the reference's [B,C,N,T] layout: a lane owns one row of T contiguous
// floats (T/4 16-byte loads), neighbouring lanes neighbouring rows.  All are HBM-bound.
#include "common.hpp"
#include "rowtile.hpp"

namespace msgat {

// ---- time mixing --------------------------------------------------------------------------------------
// forward  (BWD = false): dst[g,o,n,t]        = bias[o] + sum_k sum_i A[ga,k,t,i] src[g,k*Co+o,n,i]
// backward (BWD = true):  dst[g,k*Co+o,n,i]   =           sum_t  A[ga,k,t,i] src[g,o,n,t]
// ga = g when the matrices are per sample (a_gstride = K*T*T), 0 when shared (a_gstride = 0).
// The matrices sit in LDS, transposed for the backward form so that both read rows as float4 broadcasts.
template <int T, int K, bool BWD>
__global__ __launch_bounds__(kBlock) void k_tmix(const float* __restrict__ src, const float* __restrict__ A,
                                                 int a_gstride, const float* __restrict__ bias, int Bg,
                                                 float* __restrict__ dst, int Co, int N, int src_gs) {
  // src_gs: channels per group of the tensor src is a channel slice of (backward form: the incoming gradient is a
  // slice [:, a:b] of the block's concatenated gradient, read in place)
  __shared__ float Al[K][T][T];
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const int g = blockIdx.y;
  for (int i = threadIdx.x; i < K * T * T; i += kBlock) {
    const int k = i / (T * T), t = (i / T) % T, c = i % T;
    const float a = A[(size_t)g * a_gstride + i];
    if (BWD) Al[k][c][t] = a; else Al[k][t][c] = a;
  }
  __syncthreads();
  // a wave owns 64 consecutive rows (o, n) of this group, lane = row, moved in flat order (rowtile.hpp)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  const int rows = Co * N;
  const int r0 = (blockIdx.x * (kBlock / kWave) + wave) * kWave;
  if (r0 >= rows) return;                                   // wave-uniform
  const int nf = min(kWave, rows - r0) * (T / 4);
  const int rr = min(r0 + lane, rows - 1);                  // lanes past the last row compute it again; nothing is stored
  const RowTile<T> rt(tiles[wave], lane);
  const int o = rr / N;
  if (!BWD) {
    float4 in[K][T / 4];
#pragma unroll
    for (int k = 0; k < K; ++k) rt.fetch(src + ((size_t)g * K * rows + (size_t)k * rows + r0) * T, nf, in[k]);
    float acc[T];
    const float b = bias ? bias[(g / Bg) * Co + o] : 0.f;  // bias [R,Co], Bg groups per relation
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = b;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float v[T];
      rt.to_row(in[k], v);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < T; ++i) acc[t] = fmaf(Al[k][t][i], v[i], acc[t]);
    }
    rt.store(dst + ((size_t)g * rows + r0) * T, nf, acc);
  } else {
    float4 in[T / 4];
    rt.fetch(src + ((size_t)g * src_gs * N + r0) * T, nf, in);
    float v[T];
    rt.to_row(in, v);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float acc[T];
#pragma unroll
      for (int i = 0; i < T; ++i) {
        acc[i] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[i] = fmaf(Al[k][i][t], v[t], acc[i]);
      }
      rt.store(dst + ((size_t)g * K * rows + (size_t)k * rows + r0) * T, nf, acc);
    }
  }
}

template <int T, int K>
static int launch_tmix_tk(const float* src, const float* A, int per_group, const float* bias, float* dst, int G,
                          int Co, int N, int backward, int Bg, int src_gs, hipStream_t s) {
  dim3 grid(cdiv(Co * N, kBlock), G);
  const int gs = per_group ? K * T * T : 0;
  if (backward) hipLaunchKernelGGL((k_tmix<T, K, true>), grid, dim3(kBlock), 0, s, src, A, gs, bias, Bg, dst, Co, N, src_gs);
  else hipLaunchKernelGGL((k_tmix<T, K, false>), grid, dim3(kBlock), 0, s, src, A, gs, bias, Bg, dst, Co, N, src_gs);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

template <int T>
static int launch_tmix_t(const float* src, const float* A, int per_group, const float* bias, float* dst, int G,
                         int Co, int K, int N, int backward, int Bg, int src_gs, hipStream_t s) {
  if (K == 1) return launch_tmix_tk<T, 1>(src, A, per_group, bias, dst, G, Co, N, backward, Bg, src_gs, s);
  if (K == 2) return launch_tmix_tk<T, 2>(src, A, per_group, bias, dst, G, Co, N, backward, Bg, src_gs, s);
  return MSGAT_ERR_UNSUPPORTED;
}

int launch_tmix(const float* src, const float* A, int per_group, const float* bias, float* dst, int G, int Co,
                int K, int N, int T, int backward, int R, hipStream_t s, int src_gs) {
  const int Bg = G / R;
  if (src_gs <= 0 || !backward) src_gs = Co;
  switch (T) {
    case 4: return launch_tmix_t<4>(src, A, per_group, bias, dst, G, Co, K, N, backward, Bg, src_gs, s);
    case 8: return launch_tmix_t<8>(src, A, per_group, bias, dst, G, Co, K, N, backward, Bg, src_gs, s);
    case 12: return launch_tmix_t<12>(src, A, per_group, bias, dst, G, Co, K, N, backward, Bg, src_gs, s);
    case 16: return launch_tmix_t<16>(src, A, per_group, bias, dst, G, Co, K, N, backward, Bg, src_gs, s);
  }
  return MSGAT_ERR_UNSUPPORTED;
}

// ---- gradient of the time-mixing matrices -------------------------------------------------------------
// dA[g,k,t,i] = sum_{o,n} dout[g,o,n,t] y[g,k*Co+o,n,i]: a [T x T] outer-product sum over the Co*N rows of
// a group.  A lane keeps the T*T accumulators in registers over its rows; waves reduce by shuffles, the
// block through LDS, and kTmixChunks blocks per (g, k) leave partials for the fixed-order final sum.
constexpr int kTmixChunks = 4;
constexpr int kTdaUnroll = 8;  // k-steps (of 4 rows) whose loads are in flight together

// The outer-product sum on the matrix cores: D[t][i] += sum_rows A[t][row] B[row][i] with A = dout^T, B = y, 4 rows per
// v_mfma_f32_16x16x4_f32 (exact fp32).  Lane (m = lane & 15, kq = lane >> 4) supplies dout[row 4s + kq][m] and
// y[row 4s + kq][m]: the 16 lanes of a row read its T contiguous floats, the four rows of a k-step are adjacent, so a
// wave-instruction reads 4 T contiguous floats.  The VALU form kept T*T accumulators per lane (144 registers at
// T = 12, two waves per SIMD) and ran at 108 us for 294 MB; this one needs 4 + the loads in flight.
// Columns m >= T and rows past the chunk's end are clamped addresses times a 0/1 mask (no load in a branch).
// (Round 5 lab: sixteen rows per step as v_mfma_f32_4x4x1_16B_f32 blocks -- lane (row, quarter) loads one dwordx3 of dout
// and of y, nine 8-cycle instructions per 16 rows instead of four 32-cycle ones and two load instructions instead of
// eight -- ran at 83 us against 78: this pass is at its bytes, 390 MB at 5 TB/s, not at its instruction count.)
template <int T>
__global__ __launch_bounds__(kBlock) void k_tmix_dA(const float* __restrict__ dout, const float* __restrict__ y,
                                                    float* __restrict__ part, int Co, int K, int N, int dout_gs) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[kBlock / kWave][256];
  const int g = blockIdx.z, k = blockIdx.y;
  const int rows = Co * N;
  const int per = cdiv(rows, (int)gridDim.x);
  const int r0 = blockIdx.x * per, r1 = min(r0 + per, rows);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int mc = min(m, T - 1);
  const float colmask = m < T ? 1.f : 0.f;
  const float* dbase = dout + (size_t)g * dout_gs * N * T + mc;  // dout may be a channel slice of a wider tensor
  const float* ybase = y + ((size_t)g * K + k) * rows * T + mc;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // wave w takes k-steps w, w + 4, ... of the chunk, kTdaUnroll at a time
  constexpr int kWaves = kBlock / kWave;
  for (int s0 = r0 + 4 * wave * kTdaUnroll; s0 < r1; s0 += 4 * kWaves * kTdaUnroll) {
    float a[kTdaUnroll], b[kTdaUnroll];
#pragma unroll
    for (int u = 0; u < kTdaUnroll; ++u) {
      const int rr = s0 + 4 * u + kq;
      const int rc = min(rr, r1 - 1);
      const float keep = rr < r1 ? colmask : 0.f;
      a[u] = dbase[(size_t)rc * T] * keep;
      b[u] = ybase[(size_t)rc * T] * keep;
    }
#pragma unroll
    for (int u = 0; u < kTdaUnroll; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
  }
  // D[t = 4 * (lane >> 4) + reg][i = lane & 15]; the four waves' tiles are added in a fixed order
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) red[wave][(4 * kq + reg) * 16 + m] = acc[reg];
  __syncthreads();
  if (threadIdx.x < T * T) {
    const int t = threadIdx.x / T, i = threadIdx.x - t * T;
    const int e = t * 16 + i;
    part[(((size_t)g * K + k) * gridDim.x + blockIdx.x) * (T * T) + threadIdx.x] =
        (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
}

size_t tmix_partial_floats(int G, int K, int T) { return (size_t)G * K * kTmixChunks * T * T; }

int launch_tmix_dA(const float* dout, const float* y, float* dA, float* part, int G, int Co, int K, int N, int T,
                   hipStream_t s, int dout_gs) {
  if (dout_gs <= 0) dout_gs = Co;
  dim3 grid(kTmixChunks, K, G);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_tmix_dA<4>, grid, dim3(kBlock), 0, s, dout, y, part, Co, K, N, dout_gs); break;
    case 8: hipLaunchKernelGGL(k_tmix_dA<8>, grid, dim3(kBlock), 0, s, dout, y, part, Co, K, N, dout_gs); break;
    case 12: hipLaunchKernelGGL(k_tmix_dA<12>, grid, dim3(kBlock), 0, s, dout, y, part, Co, K, N, dout_gs); break;
    case 16: hipLaunchKernelGGL(k_tmix_dA<16>, grid, dim3(kBlock), 0, s, dout, y, part, Co, K, N, dout_gs); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  return launch_reduce_groups(part, G * K, kTmixChunks, T * T, dA, s);
}

// ---- node pooling -------------------------------------------------------------------------------------
// pooled[s,t] = sum_n w[n] x[s,n,t] for every (sample, channel) slab s (attention.py:89).
template <int T>
__global__ __launch_bounds__(kBlock) void k_node_pool(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ pooled, int N, int spr, int C, int gs) {
  __shared__ float red[kBlock / kWave][T];
  const size_t sl = blockIdx.x;
  w += (size_t)(blockIdx.x / spr) * N;  // weights [R,N]: spr slabs per relation
  // C > 0: slab (g, c) of a channel slice of a [G, gs, N, T] tensor, read in place
  const size_t xs = (C > 0) ? (size_t)(blockIdx.x / C) * gs + (blockIdx.x % C) : sl;
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const RowTile<T> rt(tiles[wave], lane);
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.f;
  // a wave takes 64 consecutive nodes = rows per trip, read in flat order (rowtile.hpp)
  for (int n0 = wave * kWave; n0 < N; n0 += kBlock) {
    const int nf = min(kWave, N - n0) * (T / 4);
    float4 in[T / 4];
    rt.fetch(x + (xs * N + n0) * T, nf, in);
    const float wl = w[min(n0 + lane, N - 1)];
    const float wn = n0 + lane < N ? wl : 0.f;
    float v[T];
    rt.to_row(in, v);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = fmaf(wn, v[t], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float a = acc[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) red[wave][t] = a;
  }
  __syncthreads();
  if (threadIdx.x < T)
    pooled[sl * T + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dx[s,n,t] = w[n] dpooled[s,t] (+ add[s,n,t]: the gradient that reached x along its other path)
template <int T>
__global__ __launch_bounds__(kBlock) void k_node_pool_dx(const float* __restrict__ w, const float* __restrict__ dp,
                                                         const float* __restrict__ add, float* __restrict__ dx, int N,
                                                         int spr) {
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const size_t sl = blockIdx.x;
  w += (size_t)(blockIdx.x / spr) * N;
  // a wave owns 64 consecutive nodes = rows of this slab, moved in flat order (rowtile.hpp)
  const int lane = threadIdx.x & (kWave - 1);
  const int n0 = blockIdx.y * kBlock + (int)(threadIdx.x - lane);
  if (n0 >= N) return;                                      // wave-uniform
  const int nf = min(kWave, N - n0) * (T / 4);
  const RowTile<T> rt(tiles[threadIdx.x >> 6], lane);
  float v[T];
  const float wn = w[min(n0 + lane, N - 1)];
  if (add != nullptr) {
    float4 in[T / 4];
    rt.fetch(add + (sl * N + n0) * T, nf, in);
    rt.to_row(in, v);
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = fmaf(wn, dp[sl * T + t], v[t]);
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = wn * dp[sl * T + t];
  }
  rt.store(dx + (sl * N + n0) * T, nf, v);
}

// dw partial[g*nck + ck, n] = sum_{c in chunk ck} sum_t x[g,c,n,t] dpooled[g,c,t]
constexpr int kPoolCC = 16;  // channels per block

template <int T>
__global__ __launch_bounds__(kBlock) void k_node_pool_dw(const float* __restrict__ x, const float* __restrict__ dp,
                                                         float* __restrict__ part, int C, int N) {
  __shared__ float dl[kPoolCC][T];
  const int g = blockIdx.z, ck = blockIdx.y;
  const int c0 = ck * kPoolCC, cn = min(kPoolCC, C - c0);
  for (int i = threadIdx.x; i < cn * T; i += kBlock) dl[i / T][i % T] = dp[((size_t)g * C + c0) * T + i];
  __syncthreads();
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const int lane = threadIdx.x & (kWave - 1);
  const int n0 = blockIdx.x * kBlock + (int)(threadIdx.x - lane);   // the wave's 64 nodes: rows read in flat order
  if (n0 >= N) return;                                              // wave-uniform, after the barrier
  const int nf = min(kWave, N - n0) * (T / 4);
  const RowTile<T> rt(tiles[threadIdx.x >> 6], lane);
  float acc = 0.f;
  for (int c = 0; c < cn; ++c) {
    float4 in[T / 4];
    rt.fetch(x + (((size_t)g * C + c0 + c) * N + n0) * T, nf, in);
    float v[T];
    rt.to_row(in, v);
#pragma unroll
    for (int t = 0; t < T; ++t) acc = fmaf(v[t], dl[c][t], acc);
  }
  if (n0 + lane < N) part[((size_t)g * gridDim.y + ck) * N + n0 + lane] = acc;
}

int launch_node_pool(const float* x, const float* w, float* pooled, long long slabs, int N, int T, int R,
                     hipStream_t s, int C, int gs) {
  dim3 grid((unsigned)slabs);
  const int spr = (int)(slabs / R);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_node_pool<4>, grid, dim3(kBlock), 0, s, x, w, pooled, N, spr, C, gs); break;
    case 8: hipLaunchKernelGGL(k_node_pool<8>, grid, dim3(kBlock), 0, s, x, w, pooled, N, spr, C, gs); break;
    case 12: hipLaunchKernelGGL(k_node_pool<12>, grid, dim3(kBlock), 0, s, x, w, pooled, N, spr, C, gs); break;
    case 16: hipLaunchKernelGGL(k_node_pool<16>, grid, dim3(kBlock), 0, s, x, w, pooled, N, spr, C, gs); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_node_pool_dx(const float* w, const float* dp, const float* add, float* dx, long long slabs, int N, int T,
                        int R, hipStream_t s) {
  dim3 grid((unsigned)slabs, cdiv(N, kBlock));
  const int spr = (int)(slabs / R);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_node_pool_dx<4>, grid, dim3(kBlock), 0, s, w, dp, add, dx, N, spr); break;
    case 8: hipLaunchKernelGGL(k_node_pool_dx<8>, grid, dim3(kBlock), 0, s, w, dp, add, dx, N, spr); break;
    case 12: hipLaunchKernelGGL(k_node_pool_dx<12>, grid, dim3(kBlock), 0, s, w, dp, add, dx, N, spr); break;
    case 16: hipLaunchKernelGGL(k_node_pool_dx<16>, grid, dim3(kBlock), 0, s, w, dp, add, dx, N, spr); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

size_t node_pool_partial_floats(int G, int C, int N) { return (size_t)G * cdiv(C, kPoolCC) * N; }

int launch_node_pool_dw(const float* x, const float* dp, float* dw, float* part, int G, int C, int N, int T, int R,
                        hipStream_t s) {
  const int nck = cdiv(C, kPoolCC);
  dim3 grid(cdiv(N, kBlock), nck, G);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_node_pool_dw<4>, grid, dim3(kBlock), 0, s, x, dp, part, C, N); break;
    case 8: hipLaunchKernelGGL(k_node_pool_dw<8>, grid, dim3(kBlock), 0, s, x, dp, part, C, N); break;
    case 12: hipLaunchKernelGGL(k_node_pool_dw<12>, grid, dim3(kBlock), 0, s, x, dp, part, C, N); break;
    case 16: hipLaunchKernelGGL(k_node_pool_dw<16>, grid, dim3(kBlock), 0, s, x, dp, part, C, N); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  return launch_reduce_groups(part, R, (G / R) * nck, N, dw, s);  // dw [R,N]: the groups of a relation are contiguous
}

// ---- prediction head -----------------------------------------------------------------------------------
// TPC's last step (msgat.py:153, applied :158-159): Conv2d(T_in -> T_out, kernel [1, C]) over the transposed
// activation, i.e. out[b,n,o] = bias[o] + sum_c sum_t W[o,t,c] x[b,c,n,t].  MIOpen runs it as one im2col +
// GEMM per sample between layout transposes; here a lane owns one (b, n) and walks a chunk of channels
// (a row of T floats per channel), the chunk's [kHeadCC][T_out][T] weights broadcast from LDS.  Channel
// chunks leave partials that are summed in a fixed order.
constexpr int kHeadCC = 8;
constexpr int kHeadTo = 16;  // T_out <= 16

// out[b,n,o] = bias[o] + sum_{c,t} x[b,c,n,t] W[o,t,c] on the matrix cores: a wave owns 16 nodes of one sample, the K axis
// runs over (channel, timestep).  Lane (i, kq) loads ONE float4 per channel -- x[b,c,n0+i,4kq..4kq+3], the row's kq-th
// quarter (kq < T/4; a wave-instruction reads 16 rows x T contiguous floats) -- and its component sigma is the lane's K
// element of the channel's sigma-th MFMA (K slot kq <-> timestep 4 kq + sigma; slots kq >= T/4 carry zero weights: 4
// MFMAs per channel at T = 12 where 3 dword loads per lane and MFMA would do, but those read 16 B of every 48-B row
// per instruction: 149 us).  Weights come from LDS laid out [c][t][o].  The whole channel sum stays in one accumulator
// tile: no channel-chunk partials (36 MB written and re-read by a reduction launch in the VALU form: 122 + 7 us).
constexpr int kHeadFwdCC = 32;   // channels of W staged in LDS at a time (24 KB at T = 12: six blocks per CU)
constexpr int kHeadFwdUn = 8;    // channels whose loads are in flight together

// Wp[r][c][t][o < 16] = W[r][o][t][0][c] (0 for o >= To): the convolution's weights in the order the kernels stage them.
// Staged straight from the convolution layout, every block gathered its 3 x 6144 weights as 4-byte loads at a
// stride of T*C floats; from Wp a chunk is one contiguous 24 KB copy.
__global__ __launch_bounds__(kBlock) void k_head_wperm(const float* __restrict__ W, float* __restrict__ Wp, int C, int T,
                                                       int To, int total) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int o = i % kHeadTo, t = (i / kHeadTo) % T, c = (i / (kHeadTo * T)) % C, r = i / (kHeadTo * T * C);
  Wp[i] = (o < To) ? W[(((size_t)r * To + o) * T + t) * C + c] : 0.f;
}

// LN: x is the INPUT of the LayerNorm in front of the head (msgat.py:158): a row's T values lie in the T/4 lanes (m, kq) of
// one node, so its mean and variance are two sums over the kq lanes (xor 16, xor 32) per channel, and the normalised
// float4 goes straight into the MFMAs -- the [B,C,N,T] LayerNorm output is never written (round 5: a 586 MB pass, 98 us
// at PEMSD7 size, for an activation only the head reads).
template <int T, bool LN>
__global__ __launch_bounds__(kBlock) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int C, int N, int To, int Bg, const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb, float eps, float* __restrict__ xn) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int T4 = T / 4;
  __shared__ float Wl[kHeadFwdCC * T * kHeadTo];
  const int b = blockIdx.y;
  W += (size_t)(b / Bg) * C * T * kHeadTo;   // pre-laid-out weights Wp [R,C,T,16] (k_head_wperm), bias [R,To]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int n0 = (blockIdx.x * (kBlock / kWave) + wave) * 16;
  const int nrow = min(n0 + m, N - 1);  // clamped: rows past N are computed from row N-1 and never stored
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < C; c0 += kHeadFwdCC) {
    const int cn = min(kHeadFwdCC, C - c0);
    __syncthreads();  // the previous chunk's weights are no longer read
    for (int i = threadIdx.x; i < kHeadFwdCC * T * kHeadTo; i += kBlock)   // one contiguous copy (Wp is zero past To)
      Wl[i] = (i < cn * T * kHeadTo) ? W[(size_t)c0 * T * kHeadTo + i] : 0.f;
    __syncthreads();
    const int kqc = min(kq, T4 - 1);
    const float kmask = kq < T4 ? 1.f : 0.f;
    const float4* src = reinterpret_cast<const float4*>(x + (((size_t)b * C + c0) * N + nrow) * T) + kqc;
    float4 lw4 = make_float4(1.f, 1.f, 1.f, 1.f), lb4 = f4zero();
    if (LN) {
      const int rel = b / Bg;
      if (lnw) lw4 = make_float4(lnw[rel * T + 4 * kqc], lnw[rel * T + 4 * kqc + 1], lnw[rel * T + 4 * kqc + 2], lnw[rel * T + 4 * kqc + 3]);
      if (lnb) lb4 = make_float4(lnb[rel * T + 4 * kqc], lnb[rel * T + 4 * kqc + 1], lnb[rel * T + 4 * kqc + 2], lnb[rel * T + 4 * kqc + 3]);
    }
    for (int cc = 0; cc < cn; cc += kHeadFwdUn) {
      float4 av[kHeadFwdUn];
#pragma unroll
      for (int u = 0; u < kHeadFwdUn; ++u) av[u] = src[(size_t)min(cc + u, cn - 1) * N * T4];
      if (LN) {   // normalise the row: its T values are this lane's float4 and those of the lanes 16 and 32 away
#pragma unroll
        for (int u = 0; u < kHeadFwdUn; ++u) {
          float4 v = av[u];
          float s1 = ((v.x + v.y) + (v.z + v.w)) * kmask;
          s1 += __shfl_xor(s1, 16);
          s1 += __shfl_xor(s1, 32);
          const float mean = s1 * (1.0f / T);
          v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
          float s2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w))) * kmask;
          s2 += __shfl_xor(s2, 16);
          s2 += __shfl_xor(s2, 32);
          const float rstd = rsqrtf(s2 * (1.0f / T) + eps);
          av[u] = make_float4(fmaf(v.x * rstd, lw4.x, lb4.x), fmaf(v.y * rstd, lw4.y, lb4.y), fmaf(v.z * rstd, lw4.z, lb4.z),
                              fmaf(v.w * rstd, lw4.w, lb4.w));
        }
        if (xn != nullptr && kq < T4 && n0 + m < N) {   // training: the weight gradient reads the normalised rows
          float4* dst = reinterpret_cast<float4*>(xn + (((size_t)b * C + c0) * N + n0 + m) * T) + kq;
#pragma unroll
          for (int u = 0; u < kHeadFwdUn; ++u)
            if (cc + u < cn) dst[(size_t)(cc + u) * N * T4] = av[u];
        }
      }
#pragma unroll
      for (int u = 0; u < kHeadFwdUn; ++u)
        if (cc + u < cn) {  // wave-uniform; no loads inside
          const float* wl = Wl + ((cc + u) * T + 4 * kqc) * kHeadTo + m;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, wl[0 * kHeadTo] * kmask, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, wl[1 * kHeadTo] * kmask, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, wl[2 * kHeadTo] * kmask, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, wl[3 * kHeadTo] * kmask, acc, 0, 0, 0);
        }
    }
  }
  // D[node = 4 * (lane >> 4) + reg][o = lane & 15]
  const float bo = (bias != nullptr && m < To) ? bias[(size_t)(b / Bg) * To + m] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int n = n0 + 4 * kq + reg;
    if (n < N && m < To) out[((size_t)b * N + n) * To + m] = acc[reg] + bo;
  }
}

// dx[b,c,n,t] = sum_o W[o,t,c] dout[b,n,o]
template <int T>
__global__ __launch_bounds__(kBlock) void k_head_dx(const float* __restrict__ dout, const float* __restrict__ W,
                                                    float* __restrict__ dx, int C, int N, int To, int Bg) {
  __shared__ float Wl[kHeadCC][T][kHeadTo];
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const int b = blockIdx.z, ck = blockIdx.y;
  W += (size_t)(b / Bg) * To * T * C;
  const int c0 = ck * kHeadCC, cn = min(kHeadCC, C - c0);
  // weights of this channel chunk and the lane's dout row: every load unconditional (clamped index, masked value) and
  // issued together -- as loops with a load and its use per trip hipcc emitted load, s_waitcnt vmcnt(0), use: 6 + 12
  // dependent round trips in front of every block's first store (110 us for a 293 MB write)
  constexpr int kWn = (kHeadCC * kHeadTo * T + kBlock - 1) / kBlock;
  float wv[kWn];
#pragma unroll
  for (int k = 0; k < kWn; ++k) {
    const int i = min((int)threadIdx.x + k * kBlock, kHeadCC * kHeadTo * T - 1);
    const int c = i / (kHeadTo * T), t = (i / kHeadTo) % T, o = i % kHeadTo;
    const float w = W[((size_t)min(o, To - 1) * T + t) * C + c0 + min(c, cn - 1)];
    wv[k] = (c < cn && o < To) ? w : 0.f;
  }
  const int n = blockIdx.x * kBlock + threadIdx.x;
  const float* src = dout + ((size_t)b * N + min(n, N - 1)) * To;
  float d[kHeadTo];
#pragma unroll
  for (int o = 0; o < kHeadTo; ++o) {
    const float v = src[min(o, To - 1)];
    d[o] = o < To ? v : 0.f;
  }
#pragma unroll
  for (int k = 0; k < kWn; ++k) {
    const int i = threadIdx.x + k * kBlock;
    if (i < kHeadCC * kHeadTo * T) (&Wl[0][0][0])[i] = wv[k];
  }
  __syncthreads();
  // a wave's 64 nodes are 64 consecutive rows of every channel: stored in flat order (rowtile.hpp)
  const int n0 = n - (int)(threadIdx.x & (kWave - 1));
  if (n0 >= N) return;                                      // wave-uniform
  const int nf = min(kWave, N - n0) * (T / 4);
  const RowTile<T> rt(tiles[threadIdx.x >> 6], threadIdx.x & (kWave - 1));
  for (int c = 0; c < cn; ++c) {
    float v[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      v[t] = 0.f;
#pragma unroll
      for (int o = 0; o < kHeadTo; ++o) v[t] = fmaf(Wl[c][t][o], d[o], v[t]);
    }
    rt.store(dx + (((size_t)b * C + c0 + c) * N + n0) * T, nf, v);
  }
}

// The head's input gradient AND the backward of the LayerNorm in front of it (msgat.py:158-159: fc(ln(x).transpose(1,3)))
// in ONE pass: dy[b,c,n,:] = sum_o W[o,:,c] dout[b,n,o] is built in registers exactly as k_head_dx builds it and consumed
// on the spot by the LayerNorm backward of row (b,c,n) (layernorm.hip: k_ln_bwd, same formulas, same order), whose x row
// the lane reads itself.  The [B,C,N,T] gradient between the two never exists: k_head_dx's 293 MB write and k_ln_bwd's
// read of it (and one launch) go.  part[block][2T]: the block's share of (dweight | dbias) of the LayerNorm.
template <int T>
__device__ __forceinline__ float lnh_centre_row(float (&x)[T], float eps) {   // = layernorm.hip's centre_row
  const float x0 = x[0];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) { x[t] -= x0; s += x[t]; }
  const float md = s * (1.0f / T);
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) { x[t] -= md; v = fmaf(x[t], x[t], v); }
  return rsqrtf(v * (1.0f / T) + eps);
}

template <int T>
__global__ __launch_bounds__(kBlock) void k_lnhead_bwd(const float* __restrict__ dout, const float* __restrict__ W,
                                                       const float* __restrict__ x, const float* __restrict__ lnw,
                                                       float* __restrict__ dx, float* __restrict__ part, int C, int N,
                                                       int To, int Bg, float eps, int relu_mask) {
  __shared__ __attribute__((aligned(16))) float Wl[kHeadCC][T][kHeadTo];
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  __shared__ float red[kBlock / kWave][2 * T];
  const int b = blockIdx.z, ck = blockIdx.y, rel = b / Bg;
  W += (size_t)rel * To * T * C;
  const int c0 = ck * kHeadCC, cn = min(kHeadCC, C - c0);
  constexpr int kWn = (kHeadCC * kHeadTo * T + kBlock - 1) / kBlock;
  float wv[kWn];
#pragma unroll
  for (int k = 0; k < kWn; ++k) {   // unconditional, clamped: see k_head_dx
    const int i = min((int)threadIdx.x + k * kBlock, kHeadCC * kHeadTo * T - 1);
    const int c = i / (kHeadTo * T), t = (i / kHeadTo) % T, o = i % kHeadTo;
    const float w = W[((size_t)min(o, To - 1) * T + t) * C + c0 + min(c, cn - 1)];
    wv[k] = (c < cn && o < To) ? w : 0.f;
  }
  const int n = blockIdx.x * kBlock + threadIdx.x;
  const int nc = min(n, N - 1);
  const bool live = n < N;
  const float* src = dout + ((size_t)b * N + nc) * To;
  float d[kHeadTo];
#pragma unroll
  for (int o = 0; o < kHeadTo; ++o) {
    const float v = src[min(o, To - 1)];
    d[o] = (o < To && live) ? v : 0.f;     // lanes past N: a zero gradient row
  }
#pragma unroll
  for (int k = 0; k < kWn; ++k) {
    const int i = threadIdx.x + k * kBlock;
    if (i < kHeadCC * kHeadTo * T) (&Wl[0][0][0])[i] = wv[k];
  }
  float gw[T], dwa[T], dba[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { gw[t] = lnw ? lnw[rel * T + t] : 1.f; dwa[t] = 0.f; dba[t] = 0.f; }
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int n0 = n - lane;
  const int nf = max(min(kWave, N - n0), 0) * (T / 4);      // 0: the whole wave lies past N (stores nothing)
  const RowTile<T> rt(tiles[wave], lane);
  // the x row of channel c + 1 is requested before channel c is worked on (clamped channel: the last trip re-reads its own
  // row): a load and its use per trip left one exposed round trip per channel, 8 in a row per block
  float4 xnext[T / 4];
  auto fetch_x = [&](int c) {
    const float4* xr = reinterpret_cast<const float4*>(x + (((size_t)b * C + c0 + min(c, cn - 1)) * N + nc) * T);
#pragma unroll
    for (int t4 = 0; t4 < T / 4; ++t4) xnext[t4] = xr[t4];
  };
  fetch_x(0);
  for (int c = 0; c < cn; ++c) {
    float gv[T], xv[T];
#pragma unroll
    for (int t4 = 0; t4 < T / 4; ++t4) {
      const float4 a = xnext[t4];
      xv[4 * t4 + 0] = a.x; xv[4 * t4 + 1] = a.y; xv[4 * t4 + 2] = a.z; xv[4 * t4 + 3] = a.w;
    }
    fetch_x(c + 1);
#pragma unroll
    for (int t = 0; t < T; ++t) {   // the weights four at a time (one ds_read_b128 per four FMAs; same order of the sum)
      gv[t] = 0.f;
#pragma unroll
      for (int o4 = 0; o4 < kHeadTo / 4; ++o4) {
        const float4 w4 = *reinterpret_cast<const float4*>(&Wl[c][t][4 * o4]);
        gv[t] = fmaf(w4.x, d[4 * o4 + 0], gv[t]);
        gv[t] = fmaf(w4.y, d[4 * o4 + 1], gv[t]);
        gv[t] = fmaf(w4.z, d[4 * o4 + 2], gv[t]);
        gv[t] = fmaf(w4.w, d[4 * o4 + 3], gv[t]);
      }
    }
    if (!live) {
#pragma unroll
      for (int t = 0; t < T; ++t) xv[t] = 0.f;
    }
    unsigned positive = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) positive |= (xv[t] > 0.f ? 1u : 0u) << t;
    const float rstd = lnh_centre_row<T>(xv, eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      xv[t] *= rstd;                  // xhat
      dba[t] += gv[t];
      dwa[t] = fmaf(gv[t], xv[t], dwa[t]);
      gv[t] *= gw[t];
      s1 += gv[t];
      s2 = fmaf(gv[t], xv[t], s2);
    }
    s1 *= (1.0f / T);
    s2 *= (1.0f / T);
#pragma unroll
    for (int t = 0; t < T; ++t) gv[t] = rstd * (gv[t] - s1 - xv[t] * s2);
    if (relu_mask) {
#pragma unroll
      for (int t = 0; t < T; ++t) gv[t] = ((positive >> t) & 1u) ? gv[t] : 0.f;
    }
    if (nf > 0) rt.store(dx + (((size_t)b * C + c0 + c) * N + n0) * T, nf, gv);   // wave-uniform
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float a = dwa[t], cc = dba[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); cc += __shfl_xor(cc, o); }
    if (lane == 0) { red[wave][t] = a; red[wave][T + t] = cc; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * T)
    part[(((size_t)b * gridDim.y + ck) * gridDim.x + blockIdx.x) * 2 * T + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dW partial[c, j, o, t] = sum over the j-th share of samples, all nodes: dout[b,n,o] x[b,c,n,t] -- like k_tmix_dA a
// [To x T] outer-product sum over rows (b, n), on the matrix cores: lane (m, kq) supplies dout[b][n = 4s + kq][o = m]
// and x[b][c][n = 4s + kq][t = m] (the VALU form held To*T = 192 accumulators per lane: 193 us for 293 MB).
// x is LayerNorm(x) as k_head_fwd<T, true> wrote it.  Normalising here instead lost twice in round 5: two 16-lane DPP
// sums per loaded element made this latency-bound kernel (864 blocks) 427 us from 114, and a [B,C,N,2] (mean, 1/std)
// buffer left by k_lnhead_bwd cost that kernel 126 -> 265 us and this one a third load stream (259 us).
constexpr int kHeadChunks = 8;    // sample chunks per (relation, channel): 95.7 / 90.0 / 88.7 us at 4 / 8 / 16 (PEMSD7 size)

template <int T>
__global__ __launch_bounds__(kBlock) void k_head_dW(const float* __restrict__ dout, const float* __restrict__ x,
                                                    float* __restrict__ part, int B, int C, int N, int To) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[kBlock / kWave][256];
  const int c = blockIdx.y, j = blockIdx.x, rel = blockIdx.z;   // B = samples per relation
  const int b0 = rel * B + (int)((long long)B * j / gridDim.x), b1 = rel * B + (int)((long long)B * (j + 1) / gridDim.x);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int mo = min(m, To - 1), mt = min(m, T - 1);
  const float omask = m < To ? 1.f : 0.f, tmask = m < T ? 1.f : 0.f;
  constexpr int kWaves = kBlock / kWave;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int b = b0; b < b1; ++b) {
    const float* dsrc = dout + (size_t)b * N * To + mo;
    const float* xsrc = x + ((size_t)b * C + c) * N * T + mt;
    for (int s0 = 4 * wave * kTdaUnroll; s0 < N; s0 += 4 * kWaves * kTdaUnroll) {
      float av[kTdaUnroll], bv[kTdaUnroll];
#pragma unroll
      for (int u = 0; u < kTdaUnroll; ++u) {
        const int n = s0 + 4 * u + kq;
        const int nc = min(n, N - 1);
        const float keep = n < N ? 1.f : 0.f;
        av[u] = dsrc[(size_t)nc * To] * (keep * omask);
        bv[u] = xsrc[(size_t)nc * T] * (keep * tmask);
      }
#pragma unroll
      for (int u = 0; u < kTdaUnroll; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
  }
  // D[o = 4 * (lane >> 4) + reg][t = lane & 15]; the four waves' tiles are added in a fixed order
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) red[wave][(4 * kq + reg) * 16 + m] = acc[reg];
  __syncthreads();
  for (int i = threadIdx.x; i < To * T; i += kBlock) {
    const int o = i / T, t = i - o * T;
    const int e = o * 16 + t;
    part[((((size_t)rel * gridDim.y + c) * gridDim.x + j) * To * T) + i] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
}

// the buffer holds the pre-laid-out weights [R,C,T,16] (R <= B, T <= 16); never smaller than the channel-chunk partials
// of the VALU form it was first sized for
size_t head_fwd_partial_floats(int B, int C, int N, int To) {
  const size_t old = (size_t)B * cdiv(C, kHeadCC) * N * To, wp = (size_t)B * C * 16 * kHeadTo;
  return old > wp ? old : wp;
}
size_t head_dw_partial_floats(int C, int T, int To, int R) { return (size_t)R * C * kHeadChunks * To * T; }

#define MSGAT_T_SWITCH(T, CALL)                 \
  switch (T) {                                  \
    case 4: { constexpr int TT = 4; CALL; } break;   \
    case 8: { constexpr int TT = 8; CALL; } break;   \
    case 12: { constexpr int TT = 12; CALL; } break; \
    case 16: { constexpr int TT = 16; CALL; } break; \
    default: return MSGAT_ERR_UNSUPPORTED;      \
  }

int launch_head_fwd(const float* x, const float* W, const float* bias, float* out, float* part, int B, int C, int N,
                    int T, int To, int R, hipStream_t s, int ln, const float* lnw, const float* lnb, float eps, float* xn) {
  // the matrix-core form needs no channel-chunk partials; the buffer holds the weights in staging order
  const int Bg = B / R;
  const int total = R * C * T * kHeadTo;
  if ((size_t)total > head_fwd_partial_floats(B, C, N, To)) return MSGAT_ERR_WORKSPACE;
  hipLaunchKernelGGL(k_head_wperm, dim3(cdiv(total, kBlock)), dim3(kBlock), 0, s, W, part, C, T, To, total);
  MSGAT_CHECK_LAUNCH();
  dim3 grid(cdiv(N, 16 * (kBlock / kWave)), B);
  if (ln) {
    MSGAT_T_SWITCH(T, hipLaunchKernelGGL((k_head_fwd<TT, true>), grid, dim3(kBlock), 0, s, x, part, bias, out, C, N, To, Bg, lnw, lnb, eps, xn));
  } else {
    MSGAT_T_SWITCH(T, hipLaunchKernelGGL((k_head_fwd<TT, false>), grid, dim3(kBlock), 0, s, x, part, bias, out, C, N, To, Bg, lnw, lnb, eps, nullptr));
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_head_dx(const float* dout, const float* W, float* dx, int B, int C, int N, int T, int To, int R,
                   hipStream_t s) {
  const int Bg = B / R;
  dim3 grid(cdiv(N, kBlock), cdiv(C, kHeadCC), B);
  MSGAT_T_SWITCH(T, hipLaunchKernelGGL(k_head_dx<TT>, grid, dim3(kBlock), 0, s, dout, W, dx, C, N, To, Bg));
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

size_t lnhead_partial_floats(int B, int C, int N, int T) {
  return (size_t)B * cdiv(C, kHeadCC) * cdiv(N, kBlock) * 2 * T;
}

// dx [B,C,N,T] = LayerNorm backward of (head input gradient); dlnw / dlnb [R,T]
int launch_lnhead_bwd(const float* dout, const float* W, const float* x, const float* lnw, float* dx, float* dlnw,
                      float* dlnb, float* part, int B, int C, int N, int T, int To, int R, float eps, int relu_mask,
                      hipStream_t s) {
  const int Bg = B / R;
  dim3 grid(cdiv(N, kBlock), cdiv(C, kHeadCC), B);
  MSGAT_T_SWITCH(T, hipLaunchKernelGGL(k_lnhead_bwd<TT>, grid, dim3(kBlock), 0, s, dout, W, x, lnw, dx, part, C, N, To, Bg,
                                       eps, relu_mask));
  MSGAT_CHECK_LAUNCH();
  // a relation's blocks are contiguous (b-major): J = Bg * channel chunks * node blocks partials of 2T columns each
  return launch_reduce_split(part, R, Bg * (int)grid.y * (int)grid.x, 2 * T, dlnw, T, dlnb, T, s);
}

// dWc[c][o][t] (the caller permutes to the convolution's [To][T][1][C] layout)
int launch_head_dW(const float* dout, const float* x, float* dWc, float* part, int B, int C, int N, int T, int To,
                   int R, hipStream_t s) {
  dim3 grid(kHeadChunks, C, R);
  MSGAT_T_SWITCH(T, hipLaunchKernelGGL(k_head_dW<TT>, grid, dim3(kBlock), 0, s, dout, x, part, B / R, C, N, To));
  MSGAT_CHECK_LAUNCH();
  return launch_reduce_groups(part, R * C, kHeadChunks, To * T, dWc, s);   // dWc [R,C,To,T]
}

}  // namespace msgat
