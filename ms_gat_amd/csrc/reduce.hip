// Parameter-gradient reductions of the backward pass.
//
//   dWg[r,t,s]   = sum_{g in r, n}  q[g,n,t] dkW[g,n,s]
//   dW[r,o,c]    = sum_{g in r, p}  A[g,o,p] B[g,c,p]     (A,B) = (dz, y) or (du, x)
//   dalpha[r,c]  = sum_{g in r, p}  dq[g,p]  x[g,c,p]
//
// All three are "channel-pair" contractions over a long position axis.  Each block reduces
// a span of positions to a [Ca x Cb] partial, written with plain stores; a second small
// kernel sums the partials in a fixed order, so the gradients are bitwise reproducible
// run to run (no float atomics).
#include "common.hpp"

namespace msgat {

constexpr int kNS4 = 4;

// ---- sum of partials: out[r,i] = sum_j part[r,j,i], split over two destinations -----------------
__global__ __launch_bounds__(kBlock) void k_reduce_partials(const float* __restrict__ part, int J,
                                                            int Wd, float* __restrict__ dst0, int n0,
                                                            float* __restrict__ dst1, int n1) {
  // block = 64 columns x 4 waves; wave w sums partials j = w, w+4, ... (8 loads in flight),
  // the 4 wave sums are combined in a fixed order through LDS
  __shared__ float red[kNS4][kWave];
  const int r = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;
  const int i = blockIdx.x * kWave + lane;
  float acc = 0.f;
  if (i < Wd) {
    const float* p = part + (size_t)r * J * Wd + i;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    int j = w;
    for (; j + 28 < J; j += 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += p[(size_t)(j + 4 * k) * Wd];
    }
    for (; j < J; j += 4) a[0] += p[(size_t)j * Wd];
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w != 0 || i >= Wd) return;
  const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  if (i < n0) {
    if (dst0 != nullptr) dst0[(size_t)r * n0 + i] = v;
  } else if (dst1 != nullptr && i - n0 < n1) {
    dst1[(size_t)r * n1 + (i - n0)] = v;
  }
}

static int launch_reduce_partials(const float* part, int R, int J, int Wd, float* dst0, int n0,
                                  float* dst1, int n1, hipStream_t s) {
  dim3 grid(cdiv(Wd, kWave), R);
  hipLaunchKernelGGL(k_reduce_partials, grid, dim3(kBlock), 0, s, part, J, Wd, dst0, n0, dst1, n1);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- dWg -------------------------------------------------------------------------------------------
constexpr int kRB = 128;  // rows per block

template <int T>
__global__ __launch_bounds__(kBlock) void k_dwg(const float* __restrict__ q,
                                                const float* __restrict__ dkW,
                                                float* __restrict__ part, int N, int nblk) {
  __shared__ float qs[kRB * T];
  __shared__ float ds[kRB * T];
  const int g = blockIdx.y;
  const int n0 = blockIdx.x * kRB;
  const int rows = min(kRB, N - n0);
  const float* qsrc = q + ((size_t)g * N + n0) * T;
  const float* dsrc = dkW + ((size_t)g * N + n0) * T;
  for (int i = threadIdx.x; i < rows * T; i += kBlock) {
    qs[i] = qsrc[i];
    ds[i] = dsrc[i];
  }
  __syncthreads();
  if (threadIdx.x < T * T) {
    const int t = threadIdx.x / T, s = threadIdx.x - t * T;
    float acc = 0.f;
    for (int row = 0; row < rows; ++row) acc = fmaf(qs[row * T + t], ds[row * T + s], acc);
    part[((size_t)g * nblk + blockIdx.x) * (T * T) + threadIdx.x] = acc;
  }
}

size_t dwg_partial_floats(int G, int N, int T) { return (size_t)G * cdiv(N, kRB) * T * T; }

int launch_dwg(const float* q, const float* dkW, float* part, float* dWg, int G, int Bg, int N,
               int T, hipStream_t s) {
  const int nblk = cdiv(N, kRB);
  dim3 grid(nblk, G);
  switch (T) {
    case 4: hipLaunchKernelGGL(k_dwg<4>, grid, dim3(kBlock), 0, s, q, dkW, part, N, nblk); break;
    case 8: hipLaunchKernelGGL(k_dwg<8>, grid, dim3(kBlock), 0, s, q, dkW, part, N, nblk); break;
    case 12: hipLaunchKernelGGL(k_dwg<12>, grid, dim3(kBlock), 0, s, q, dkW, part, N, nblk); break;
    case 16: hipLaunchKernelGGL(k_dwg<16>, grid, dim3(kBlock), 0, s, q, dkW, part, N, nblk); break;
    default: return MSGAT_ERR_UNSUPPORTED;
  }
  MSGAT_CHECK_LAUNCH();
  return launch_reduce_partials(part, G / Bg, Bg * nblk, T * T, dWg, T * T, nullptr, 0, s);
}

// ---- channel-pair contraction over positions --------------------------------------------------------
// Block = one span of kPB positions of one group.  Per step, kPT positions of all Ca + Cb
// channels are staged TRANSPOSED in LDS ([position][channel], channel padded to x4) so a
// lane reads its 4 A-channels and 4 B-channels with one ds_read_b128 each and does a 4x4
// outer product.  Lanes beyond the (Ca/4)x(Cb/4) tile grid take other positions of the
// step (`nh` position phases) and their partials are summed by k_reduce_partials.
constexpr int kPT = 32;
constexpr int kPB = 2048;

__global__ __launch_bounds__(kBlock) void k_chanpair(
    const float* __restrict__ A, const float* __restrict__ Aextra, const float* __restrict__ B,
    float* __restrict__ part, int Ca, int Cb, int P, int nblk, int nh) {
  extern __shared__ float lds[];
  const int CaP = (Ca + 3) & ~3, CbP = (Cb + 3) & ~3;
  float* At = lds;              // [kPT][CaP]
  float* Bt = lds + kPT * CaP;  // [kPT][CbP]
  const int g = blockIdx.y;
  const int CaMain = (Aextra != nullptr) ? Ca - 1 : Ca;
  const int na = CaP / 4, nb = CbP / 4;
  const int ntile = na * nb;
  const int tile = threadIdx.x % ntile;
  const int phase = threadIdx.x / ntile;  // position phase; lanes with phase >= nh idle in the math
  const int ta = tile / nb, tb = tile - ta * nb;
  const int pbeg = blockIdx.x * kPB;
  const int pend = min(P, pbeg + kPB);

  for (int i = threadIdx.x; i < kPT * (CaP + CbP); i += kBlock) lds[i] = 0.f;  // zero the channel padding once

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int p0 = pbeg; p0 < pend; p0 += kPT) {
    __syncthreads();
    // stage: one float4 (4 positions of one channel) per lane-iteration, stored transposed
    constexpr int Q = kPT / 4;
    for (int i = threadIdx.x; i < (Ca + Cb) * Q; i += kBlock) {
      const int chn = i / Q, pq4 = i - chn * Q;
      const int p = p0 + 4 * pq4;
      float4 v = f4zero();
      if (p < pend) {  // P is a multiple of 4 and so are pbeg/pend: a float4 never straddles the end
        const float* src;
        if (chn < CaMain) src = A + ((size_t)g * CaMain + chn) * P + p;
        else if (chn < Ca) src = Aextra + (size_t)g * P + p;
        else src = B + ((size_t)g * Cb + (chn - Ca)) * P + p;
        v = *reinterpret_cast<const float4*>(src);
      }
      float* dst = (chn < Ca) ? (At + (4 * pq4) * CaP + chn) : (Bt + (4 * pq4) * CbP + (chn - Ca));
      const int stride = (chn < Ca) ? CaP : CbP;
      dst[0] = v.x;
      dst[stride] = v.y;
      dst[2 * stride] = v.z;
      dst[3 * stride] = v.w;
    }
    __syncthreads();
    if (phase < nh) {
      for (int pp = phase; pp < kPT; pp += nh) {
        const float4 a = *reinterpret_cast<const float4*>(At + pp * CaP + 4 * ta);
        const float4 b = *reinterpret_cast<const float4*>(Bt + pp * CbP + 4 * tb);
        const float av[4] = {a.x, a.y, a.z, a.w};
        const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
      }
    }
  }
  if (phase < nh) {
    float* out = part + (((size_t)g * nblk + blockIdx.x) * nh + phase) * ((size_t)Ca * Cb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int a = 4 * ta + i;
      if (a >= Ca) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * tb + j;
        if (c < Cb) out[(size_t)a * Cb + c] = acc[i][j];
      }
    }
  }
}

static void chanpair_geometry(int Ca, int Cb, int* ntile, int* nh) {
  *ntile = ((Ca + 3) / 4) * ((Cb + 3) / 4);
  int h = kBlock / *ntile;
  if (h < 1) h = 0;  // tile grid larger than a block: unsupported
  if (h > kPT) h = kPT;
  *nh = h;
}

size_t chanpair_partial_floats(int G, int Ca, int Cb, int P) {
  int ntile, nh;
  chanpair_geometry(Ca, Cb, &ntile, &nh);
  if (nh == 0) return 0;
  return (size_t)G * cdiv(P, kPB) * nh * Ca * Cb;
}

int launch_chanpair(const float* A, const float* Aextra, const float* B, float* part, float* dst0,
                    int n0, float* dst1, int n1, int G, int Bg, int Ca, int Cb, int P,
                    hipStream_t s) {
  int ntile, nh;
  chanpair_geometry(Ca, Cb, &ntile, &nh);
  if (nh == 0) return MSGAT_ERR_UNSUPPORTED;
  const int nblk = cdiv(P, kPB);
  const int CaP = (Ca + 3) & ~3, CbP = (Cb + 3) & ~3;
  const size_t lds = (size_t)kPT * (CaP + CbP) * sizeof(float);
  dim3 grid(nblk, G);
  hipLaunchKernelGGL(k_chanpair, grid, dim3(kBlock), lds, s, A, Aextra, B, part, Ca, Cb, P, nblk, nh);
  MSGAT_CHECK_LAUNCH();
  return launch_reduce_partials(part, G / Bg, Bg * nblk * nh, Ca * Cb, dst0, n0, dst1, n1, s);
}

}  // namespace msgat
