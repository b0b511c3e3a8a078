// LayerNorm over the timestep axis: the producer of every GACN input.
//
// Reference: nn.LayerNorm([n_timesteps]) at /root/reference/src/models/msgat.py:114 (MEAM.ln, applied
// at :122 right before the graph branch) and :152 (TPC.ln, applied at :158).  The normalised axis is
// T = 12 contiguous floats; a [B,C,N,T] activation is B*C*N such rows (2 M rows at PEMSD7, B = 32,
// C = 72).  PyTorch's LayerNorm kernels give every row its own thread group and run at ~80 GB/s here
// (1.26 ms per call, 28 ms of a 61 ms training step in profiles/r01); one lane per row streams the
// tensor at HBM speed instead.
//
//   forward   y = (x - mean) * rstd * w + b,   mean/var over the T values of a row (biased variance)
//   backward  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w
//             dw[t] = sum_rows dy * xhat,  db[t] = sum_rows dy     (per-block partials, fixed-order sum)
#include "common.hpp"
#include "rowtile.hpp"

namespace msgat {

template <int T>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&v)[T]) {
#pragma unroll
  for (int t4 = 0; t4 < T / 4; ++t4) {
    const float4 a = reinterpret_cast<const float4*>(p)[t4];
    v[4 * t4 + 0] = a.x; v[4 * t4 + 1] = a.y; v[4 * t4 + 2] = a.z; v[4 * t4 + 3] = a.w;
  }
}
template <int T>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&v)[T]) {
#pragma unroll
  for (int t4 = 0; t4 < T / 4; ++t4)
    reinterpret_cast<float4*>(p)[t4] = make_float4(v[4 * t4], v[4 * t4 + 1], v[4 * t4 + 2], v[4 * t4 + 3]);
}
// Centres a row in place and returns rstd.  The mean is taken of the differences to the row's first
// value (exact for nearby floats), so a large common offset costs no accuracy -- the two-pass
// formula on shifted data.
template <int T>
__device__ __forceinline__ float centre_row(float (&x)[T], float eps) {
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

// Sum over the 64 lanes of a wave, result in lane 63 (six DPP adds, fixed order: project.hip's wave_sum_to_lane63)
__device__ __forceinline__ float ln_wave_sum63(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return v;
}

// pool_w / pool_part (optional, N >= 64): the node pooling of the OUTPUT, p[s,t] = sum_n pool_w[n] y[s,n,t] over its
// [N,T] slabs (attention.py:89 on MEAM's normalised input), out of this pass: a wave's 64 consecutive rows lie in at
// most two slabs, so every trip leaves two T-vectors -- its rows' share of the slab its first row lies in, and of the
// next one -- in pool_part[relation][trip][2][T]; k_pool_trips adds a slab's ~N/64 shares in trip order.  The pooling
// as its own launch read the 293 MB it had just been written: 76 us at PEMSD7 size.
template <int T>
__global__ __launch_bounds__(kBlock) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ y,
                                                   long long rows, float eps, const float* __restrict__ pool_w = nullptr,
                                                   float* __restrict__ pool_part = nullptr, int N = 1) {
  // blockIdx.y = relation (parameter set): its `rows` rows are contiguous, its weight / bias are row y of [R,T]
  x += (size_t)blockIdx.y * rows * T;
  y += (size_t)blockIdx.y * rows * T;
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const int wave = threadIdx.x >> 6;
  const RowTile<T> rt(tiles[wave], threadIdx.x & (kWave - 1));
  float wv[T], bv[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { wv[t] = w ? w[blockIdx.y * T + t] : 1.f; bv[t] = b ? b[blockIdx.y * T + t] : 0.f; }
  // a wave owns 64 consecutive rows per trip (lane = row), moved in flat order: rowtile.hpp
  for (long long r0 = ((long long)blockIdx.x * (kBlock / kWave) + wave) * kWave; r0 < rows;
       r0 += (long long)gridDim.x * kBlock) {
    const int nf = (int)min((long long)kWave, rows - r0) * (T / 4);
    float4 in[T / 4];
    rt.fetch(x + r0 * T, nf, in);
    float v[T];
    rt.to_row(in, v);
    const float rstd = centre_row<T>(v, eps);
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = fmaf(v[t] * rstd, wv[t], bv[t]);
    rt.store(y + r0 * T, nf, v);
    if (pool_w != nullptr) {   // kernel-uniform
      const int lane = threadIdx.x & (kWave - 1);
      const int ri = (int)min(r0 + lane, rows - 1);
      const bool live = r0 + lane < rows;
      const int slab = ri / N, n = ri - slab * N;
      const bool inA = slab == (int)(r0 / N);
      const float pw = live ? pool_w[(size_t)blockIdx.y * N + n] : 0.f;
      float* out = pool_part + ((size_t)blockIdx.y * (size_t)((rows + kWave - 1) / kWave) + (size_t)(r0 / kWave)) * 2 * T;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float c = pw * v[t];
        const float sa = ln_wave_sum63(inA ? c : 0.f), sb = ln_wave_sum63(inA ? 0.f : c);
        if (lane == kWave - 1) { out[t] = sa; out[T + t] = sb; }
      }
    }
  }
}

// pooled[rel, s, t] = the shares of slab s in trip order (see k_ln_fwd): one lane per (slab, t)
template <int T>
__global__ __launch_bounds__(kBlock) void k_pool_trips(const float* __restrict__ part, float* __restrict__ pooled,
                                                       long long rows, int N) {
  const int slabs = (int)(rows / N);
  const long long ntrip = (rows + kWave - 1) / kWave;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= slabs * T) return;
  const int s = i / T, t = i - s * T;
  const long long k0 = ((long long)s * N) / kWave, k1 = ((long long)(s + 1) * N - 1) / kWave;
  const float* p = part + (size_t)blockIdx.y * (size_t)ntrip * 2 * T;
  float acc = 0.f;
  for (long long k = k0; k <= k1; ++k) {
    const int first = (int)((k * kWave) / N);          // the slab trip k's first row lies in
    acc += p[(size_t)k * 2 * T + (first == s ? 0 : T) + t];
  }
  pooled[((size_t)blockIdx.y * slabs + s) * T + t] = acc;
}

// pool_w / dpooled (optional): the LayerNorm output also fed a node pooling p[s,t] = sum_n pool_w[n] y[s,n,t] over its
// [N,T] slabs (ChannelAttention's pooled signal, attention.py:89, on MEAM's normalised input) -- that consumer's gradient
// pool_w[n] dpooled[s,t] is rank one, so it is added to dy here, per row, instead of in a pass of its own over the
// activation (k_node_pool_dx: a 293 MB read and a 293 MB write per block at PEMSD7 size).
template <int T>
__global__ __launch_bounds__(kBlock) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ dy, const float* __restrict__ add,
                                                   float* __restrict__ dx, float* __restrict__ part, long long rows,
                                                   float eps, int relu_mask, const float* __restrict__ pool_w = nullptr,
                                                   const float* __restrict__ dpooled = nullptr, int N = 1,
                                                   const float* __restrict__ lnb = nullptr,
                                                   float* __restrict__ dpw_rows = nullptr) {
  __shared__ float red[kBlock / kWave][2 * T];
  x += (size_t)blockIdx.y * rows * T;
  dy += (size_t)blockIdx.y * rows * T;
  dx += (size_t)blockIdx.y * rows * T;
  if (add != nullptr) add += (size_t)blockIdx.y * rows * T;
  __shared__ float4 tiles[kBlock / kWave][RowTile<T>::kFloat4s];
  const RowTile<T> rt(tiles[threadIdx.x >> 6], threadIdx.x & (kWave - 1));
  float wv[T], dw[T], db[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { wv[t] = w ? w[blockIdx.y * T + t] : 1.f; dw[t] = 0.f; db[t] = 0.f; }
  for (long long r0 = ((long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kWave; r0 < rows;
       r0 += (long long)gridDim.x * kBlock) {
    const int nf = (int)min((long long)kWave, rows - r0) * (T / 4);
    const bool live = (long long)(threadIdx.x & (kWave - 1)) < rows - r0;   // lanes past the last row contribute nothing
    // the three input streams are read row-wise (a lane's own row, clamped): routing them through the tile as well
    // costs 36 more registers in flight and halves the occupancy (52 -> 108 VGPRs: 125 -> 137 us in the step); the
    // gradient leaves in flat order
    const long long r = min(r0 + (long long)(threadIdx.x & (kWave - 1)), rows - 1);
    float xv[T], gv[T];
    load_row<T>(x + r * T, xv);
    load_row<T>(dy + r * T, gv);
    if (pool_w != nullptr) {   // kernel-uniform: + pool_w[n] dpooled[slab,:] (row r of this relation = slab r / N, node r % N)
      const int ri = (int)r, slab = ri / N, n = ri - slab * N;
      const float pw = pool_w[(size_t)blockIdx.y * N + n];
      float pv[T];
      load_row<T>(dpooled + ((size_t)blockIdx.y * (size_t)(rows / N) + slab) * T, pv);
#pragma unroll
      for (int t = 0; t < T; ++t) gv[t] = fmaf(pw, pv[t], gv[t]);
      if (dpw_rows != nullptr) {
        // ... and the pooling weights' gradient dpool_w[n] = sum_slabs y[slab,n,:] . dpooled[slab,:]: this row's term, with
        // y = xhat w + b rebuilt from x (one float per row; summed over the slabs by a reduction launch) instead of
        // k_node_pool_dw's pass over the stored y
        float xc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) xc[t] = xv[t];
        const float rs = centre_row<T>(xc, eps);
        float dotp = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t)
          dotp = fmaf(fmaf(xc[t] * rs, wv[t], lnb ? lnb[blockIdx.y * T + t] : 0.f), pv[t], dotp);
        if (live) dpw_rows[(size_t)blockIdx.y * rows + ri] = dotp;
      }
    }
    if (!live) {
#pragma unroll
      for (int t = 0; t < T; ++t) { xv[t] = 0.f; gv[t] = 0.f; }
    }
    unsigned positive = 0;  // x > 0, per element: x is a ReLU output when relu_mask is set
#pragma unroll
    for (int t = 0; t < T; ++t) positive |= (xv[t] > 0.f ? 1u : 0u) << t;
    const float rstd = centre_row<T>(xv, eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      xv[t] *= rstd;                  // xhat
      db[t] += gv[t];
      dw[t] = fmaf(gv[t], xv[t], dw[t]);
      gv[t] *= wv[t];                 // g = dy * w
      s1 += gv[t];
      s2 = fmaf(gv[t], xv[t], s2);
    }
    s1 *= (1.0f / T);
    s2 *= (1.0f / T);
#pragma unroll
    for (int t = 0; t < T; ++t) gv[t] = rstd * (gv[t] - s1 - xv[t] * s2);
    if (add != nullptr) {  // the gradient that reached x along its other path (MEAM's residual convolution reads x too)
      float av[T];
      load_row<T>(add + r * T, av);
#pragma unroll
      for (int t = 0; t < T; ++t) gv[t] += av[t];
    }
    if (relu_mask) {  // the ReLU that produced x (MEAM's tail, msgat.py:131): its backward, applied where x is at hand
#pragma unroll
      for (int t = 0; t < T; ++t) gv[t] = ((positive >> t) & 1u) ? gv[t] : 0.f;
    }
    rt.store(dx + r0 * T, nf, gv);
  }
  // block partial of (dw, db): wave shuffle tree, then the 4 waves in a fixed order
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float a = dw[t], c = db[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); }
    if (lane == 0) { red[wave][t] = a; red[wave][T + t] = c; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * T)
    part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * T + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static int ln_blocks(long long rows) {
  const long long need = (rows + kBlock - 1) / kBlock;
  return (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);  // 8 blocks per CU, grid-stride beyond
}

size_t layernorm_partial_floats(long long rows, int T, int R) { return (size_t)R * ln_blocks(rows / R) * 2 * T; }

size_t layernorm_pool_partial_floats(long long rows, int T, int R) {   // rows = all relations' rows
  return (size_t)R * (size_t)((rows / R + kWave - 1) / kWave) * 2 * T;
}

int launch_layernorm_fwd(const float* x, const float* w, const float* b, float* y, long long rows, int T,
                         float eps, int R, hipStream_t s, const float* pool_w, float* pool_part, float* pooled, int N) {
  rows /= R;  // per relation
  if (pool_w != nullptr && (pool_part == nullptr || pooled == nullptr || N < kWave || rows % N != 0 || rows > 0x7fffffffLL))
    return MSGAT_ERR_SHAPE;
  const int nb = ln_blocks(rows);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_ln_fwd<4>, dim3(nb, R), dim3(kBlock), 0, s, x, w, b, y, rows, eps, pool_w, pool_part, N); break;
    case 8: hipLaunchKernelGGL(k_ln_fwd<8>, dim3(nb, R), dim3(kBlock), 0, s, x, w, b, y, rows, eps, pool_w, pool_part, N); break;
    case 12: hipLaunchKernelGGL(k_ln_fwd<12>, dim3(nb, R), dim3(kBlock), 0, s, x, w, b, y, rows, eps, pool_w, pool_part, N); break;
    case 16: hipLaunchKernelGGL(k_ln_fwd<16>, dim3(nb, R), dim3(kBlock), 0, s, x, w, b, y, rows, eps, pool_w, pool_part, N); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  if (pool_w != nullptr) {
    const dim3 grid(cdiv((int)(rows / N) * T, kBlock), R);
    switch (T) {
      case 4: hipLaunchKernelGGL(k_pool_trips<4>, grid, dim3(kBlock), 0, s, pool_part, pooled, rows, N); break;
      case 8: hipLaunchKernelGGL(k_pool_trips<8>, grid, dim3(kBlock), 0, s, pool_part, pooled, rows, N); break;
      case 12: hipLaunchKernelGGL(k_pool_trips<12>, grid, dim3(kBlock), 0, s, pool_part, pooled, rows, N); break;
      default: hipLaunchKernelGGL(k_pool_trips<16>, grid, dim3(kBlock), 0, s, pool_part, pooled, rows, N); break;
    }
    MSGAT_CHECK_LAUNCH();
  }
  return MSGAT_OK;
}

int launch_layernorm_bwd(const float* x, const float* w, const float* dy, const float* add, float* dx, float* dw,
                         float* db, float* part, long long rows, int T, float eps, int R, int relu_mask, hipStream_t s,
                         const float* pool_w, const float* dpooled, int N, const float* lnb, float* dpw_rows, float* dpw) {
  rows /= R;  // per relation
  if (pool_w != nullptr && (dpooled == nullptr || N <= 0 || rows % N != 0 || rows > 0x7fffffffLL)) return MSGAT_ERR_SHAPE;
  const int nb = ln_blocks(rows);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_ln_bwd<4>, dim3(nb, R), dim3(kBlock), 0, s, x, w, dy, add, dx, part, rows, eps, relu_mask, pool_w, dpooled, N, lnb, dpw_rows); break;
    case 8: hipLaunchKernelGGL(k_ln_bwd<8>, dim3(nb, R), dim3(kBlock), 0, s, x, w, dy, add, dx, part, rows, eps, relu_mask, pool_w, dpooled, N, lnb, dpw_rows); break;
    case 12: hipLaunchKernelGGL(k_ln_bwd<12>, dim3(nb, R), dim3(kBlock), 0, s, x, w, dy, add, dx, part, rows, eps, relu_mask, pool_w, dpooled, N, lnb, dpw_rows); break;
    case 16: hipLaunchKernelGGL(k_ln_bwd<16>, dim3(nb, R), dim3(kBlock), 0, s, x, w, dy, add, dx, part, rows, eps, relu_mask, pool_w, dpooled, N, lnb, dpw_rows); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  ReduceJobs jobs{};   // both sums in one launch
  if (int st = launch_reduce_split(part, R, nb, 2 * T, dw, T, db, T, s, &jobs)) return st;
  if (pool_w != nullptr && dpw_rows != nullptr && dpw != nullptr)   // dpool_w [R,N]: the column sums of a relation's [rows / N, N] terms
    if (int st = launch_reduce_groups_defer(dpw_rows, R, (int)(rows / N), N, dpw, s, &jobs)) return st;
  return launch_reduce_jobs(jobs, s);
}

}  // namespace msgat
