// Parameter-gradient reductions of the backward pass.
//
//   dWg[r,t,s]   = sum_{g in r, n}  q[g,n,t] dkW[g,n,s]
//   dW[r,o,c]    = sum_{g in r, p}  A[g,o,p] B[g,c,p]     (A,B) = (dz, y) or (du, x)
//   dalpha[r,c]  = sum_{g in r, p}  dq[g,p]  x[g,c,p]
//
// All three are "channel-pair" contractions over a long position axis.  Each wave reduces
// a span of positions to a [Ca x Cb] partial (MFMA, mfma.hip), written with plain stores; a
// second small kernel sums the partials in a fixed order, so the gradients are bitwise
// reproducible run to run (no float atomics).
#include "common.hpp"

namespace msgat {


// ---- sum of partials: out[r,i] = sum_j part[r,j,i], split over two destinations -----------------
// blockIdx.z picks the job; block = 16 waves.  A wave covers Wc = min(Wd, 64) columns; when Wd < 64 its lanes also
// split the partials: lane = (sub-slice, column), 64 / Wc sub-slices (Wc a power of two).  (wave w, sub-slice s) sums
// partials j = w S + s, + 16 S, ... (8 loads in flight); the sub-slices are combined by a fixed xor butterfly, the 16
// waves in wave order through LDS: a fixed order, bitwise reproducible.  (LayerNorm's backward leaves 2048 partials of 24 columns per
// relation: with 4 waves and one lane per column that sum took 34 us of dependent loads, twice per training step.)
constexpr int kRedWaves = 16;
constexpr int kRedBlock = 64 * kRedWaves;

// columns per wave: Wd rounded up to a power of two, at most 64 (so that the sub-slices combine by xor shuffles)
__host__ __device__ static inline int red_cols(int Wd) {
  int c = 1;
  while (c < Wd && c < kWave) c <<= 1;
  return c;
}

__global__ __launch_bounds__(kRedBlock) void k_reduce_partials(ReduceJobs jobs) {
  __shared__ float red[kRedWaves][kWave];
  const ReduceJob& jb = jobs.job[blockIdx.z];
  const int r = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;
  const int Wd = jb.Wd, J = jb.J;
  const int Wc = red_cols(Wd);          // columns per wave
  const int S = kWave / Wc;             // sub-slices of the partials per wave
  const int col = lane % Wc, sub = lane / Wc;
  const int i = blockIdx.x * Wc + col;
  if (r >= jb.R || blockIdx.x * Wc >= Wd) return;  // the grid covers the largest job
  float acc = 0.f;
  if (i < Wd) {
    const float* p = jb.part + (size_t)r * J * Wd + i;
    const int step = kRedWaves * S;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    int j = w * S + sub;
    for (; j + 7 * step < J; j += 8 * step) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += p[(size_t)(j + k * step) * Wd];
    }
    for (; j < J; j += step) a[0] += p[(size_t)j * Wd];
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  for (int off = Wc; off < kWave; off <<= 1) acc += __shfl_xor(acc, off);   // sub-slices of this wave
  red[w][lane] = acc;
  __syncthreads();
  if (w != 0 || sub != 0 || i >= Wd) return;
  float v = 0.f;
  for (int ww = 0; ww < kRedWaves; ++ww) v += red[ww][col];
  if (jb.rw > 0) {
    const int row = i / jb.rw, c = i - row * jb.rw;
    if (c < jb.rw - 1) jb.dst0[(size_t)r * jb.n0 + row * (jb.rw - 1) + c] = v;
    else jb.dst1[(size_t)r * jb.n1 + row] = v;
  } else if (i < jb.n0) {
    if (jb.dst0 != nullptr) jb.dst0[(size_t)r * jb.n0 + i] = v;
  } else if (jb.dst1 != nullptr && i - jb.n0 < jb.n1) {
    jb.dst1[(size_t)r * jb.n1 + (i - jb.n0)] = v;
  }
}

// Few partials of many columns (the per-sample matrix gradient of the merged channel mixing: 2 partials of 98 x 73 sums for each
// of 96 groups): a lane per column.  k_reduce_partials gives such a job 16 waves per 64 columns of which J work -- 10 656 blocks
// of 1024 lanes, 25 us; here 2 700 blocks of 256.  Same order of additions (p0 + p1 + ... from wave 0 upwards), same bits.
constexpr int kFewPartials = 16;   // = kRedWaves: up to here k_reduce_partials gives every partial its own wave (S = 1)
__global__ __launch_bounds__(256) void k_reduce_few(ReduceJobs jobs) {
  const ReduceJob& jb = jobs.job[blockIdx.z];
  const int r = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int Wd = jb.Wd, J = jb.J;
  if (r >= jb.R || i >= Wd) return;
  const float* p = jb.part + (size_t)r * J * Wd + i;
  float a[kFewPartials];
#pragma unroll
  for (int j = 0; j < kFewPartials; ++j) a[j] = j < J ? p[(size_t)j * Wd] : 0.f;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < kFewPartials; ++j) v += a[j];
  if (jb.rw > 0) {
    const int row = i / jb.rw, c = i - row * jb.rw;
    if (c < jb.rw - 1) jb.dst0[(size_t)r * jb.n0 + row * (jb.rw - 1) + c] = v;
    else jb.dst1[(size_t)r * jb.n1 + row] = v;
  } else if (i < jb.n0) {
    if (jb.dst0 != nullptr) jb.dst0[(size_t)r * jb.n0 + i] = v;
  } else if (jb.dst1 != nullptr && i - jb.n0 < jb.n1) {
    jb.dst1[(size_t)r * jb.n1 + (i - jb.n0)] = v;
  }
}

int launch_reduce_jobs(const ReduceJobs& jobs, hipStream_t s) {
  if (jobs.n <= 0) return MSGAT_OK;
  bool few = true;
  int fx = 1, fy = 1;
  for (int k = 0; k < jobs.n; ++k) {
    few = few && jobs.job[k].J <= kFewPartials && jobs.job[k].Wd >= kWave;   // Wd >= 64: one sub-slice per wave over there
    fx = max(fx, cdiv(jobs.job[k].Wd, 256));
    fy = max(fy, jobs.job[k].R);
  }
  if (few) {
    hipLaunchKernelGGL(k_reduce_few, dim3(fx, fy, jobs.n), dim3(256), 0, s, jobs);
    MSGAT_CHECK_LAUNCH();
    return MSGAT_OK;
  }
  int bx = 1, by = 1;
  for (int k = 0; k < jobs.n; ++k) {
    bx = max(bx, cdiv(jobs.job[k].Wd, red_cols(jobs.job[k].Wd)));
    by = max(by, jobs.job[k].R);
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3(bx, by, jobs.n), dim3(kRedBlock), 0, s, jobs);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// queue the reduction when the caller collects them, run it now otherwise
static int reduce_or_defer(const ReduceJob& jb, ReduceJobs* defer, hipStream_t s) {
  if (defer != nullptr && defer->n < kMaxReduceJobs) {
    defer->job[defer->n++] = jb;
    return MSGAT_OK;
  }
  ReduceJobs one{};
  one.job[0] = jb;
  one.n = 1;
  return launch_reduce_jobs(one, s);
}

static int launch_reduce_partials(const float* part, int R, int J, int Wd, float* dst0, int n0,
                                  float* dst1, int n1, hipStream_t s, ReduceJobs* defer = nullptr) {
  return reduce_or_defer(ReduceJob{part, R, J, Wd, dst0, n0, dst1, n1}, defer, s);
}

int launch_reduce_groups_defer(const float* part, int R, int J, int Wd, float* dst, hipStream_t s, ReduceJobs* defer) {
  return launch_reduce_partials(part, R, J, Wd, dst, Wd, nullptr, 0, s, defer);
}

int launch_reduce_rows(const float* part, int J, int Wd, float* dst0, int n0, float* dst1, int n1,
                       hipStream_t s) {
  return launch_reduce_partials(part, 1, J, Wd, dst0, n0, dst1, n1, s);
}

int launch_reduce_split(const float* part, int R, int J, int Wd, float* dst0, int n0, float* dst1, int n1,
                        hipStream_t s, ReduceJobs* defer) {
  return launch_reduce_partials(part, R, J, Wd, dst0, n0, dst1, n1, s, defer);
}

int launch_reduce_lastcol(const float* part, int R, int J, int rows, int rw, float* dst, hipStream_t s, ReduceJobs* defer) {
  ReduceJob jb{part, R, J, rows * rw, dst, rows * (rw - 1), dst + (size_t)R * rows * (rw - 1), rows};
  jb.rw = rw;
  return reduce_or_defer(jb, defer, s);
}

int launch_reduce_groups(const float* part, int R, int J, int Wd, float* dst, hipStream_t s) {
  return launch_reduce_partials(part, R, J, Wd, dst, Wd, nullptr, 0, s);
}

// ---- channel-pair contraction over positions --------------------------------------------------------
// k_chanpair_mfma (mfma.hip) leaves one [Ca x Cb] partial per block of its persistent grid; the
// fixed-order sum over a relation's blocks happens here.
size_t chanpair_partial_floats(int G, int Bg, int Ca, int Cb) {
  const int R = G / Bg;
  return (size_t)R * chanpair_mfma_blocks(R) * Ca * Cb + 260;   // + the dump words of the fused contraction-and-mix form
}

int launch_chanpair_seg(const SegList& A, const float* B, float* part, float* dst0, int n0, float* dst1, int n1,
                        int G, int Bg, int Cb, int P, hipStream_t s, int b_ones, ReduceJobs* defer, int lastcol_apart) {
  const int R = G / Bg;
  int nblk = 0;
  const int st = launch_chanpair_mfma(A, B, part, R, Bg, Cb, P, chanpair_mfma_blocks(R), b_ones, s, &nblk);
  if (st) return st;
  if (lastcol_apart) return launch_reduce_lastcol(part, R, nblk, A.total(), Cb, dst0, s, defer);
  return launch_reduce_partials(part, R, nblk, A.total() * Cb, dst0, n0, dst1, n1, s, defer);
}

// dW | dalpha = [du | dq] x^T and dx = W^T du + alpha (x) dq in one pass over du, dq and x; *done = 0 when the fused
// form does not cover the shape (nothing launched)
int launch_chanpair_mix(const float* du, const float* dq, const float* x, const float* W, const float* alpha, float* dx,
                        float* part, float* dW, float* dalpha, int G, int Bg, int Co, int C, int P, hipStream_t s,
                        ReduceJobs* defer, int* done) {
  const int R = G / Bg;
  int nblk = 0;
  const int st = launch_chanpair_mix(seg_pair(du, Co, dq, 1), x, part, R, Bg, C, P, chanpair_mfma_blocks(R), W, alpha, dx,
                                     s, &nblk, done);
  if (st || !*done) return st;
  return launch_reduce_partials(part, R, nblk, (Co + 1) * C, dW, Co * C, dalpha, C, s, defer);
}

int launch_chanpair(const float* A, const float* Aextra, const float* B, float* part, float* dst0,
                    int n0, float* dst1, int n1, int G, int Bg, int Ca, int Cb, int P,
                    hipStream_t s, ReduceJobs* defer) {
  // channel axis [A | Aextra]: Aextra, when given, supplies the last row
  const SegList seg = (Aextra == nullptr) ? seg_single(A, Ca)
                      : (Ca == 1 ? seg_single(Aextra, 1) : seg_pair(A, Ca - 1, Aextra, 1));
  return launch_chanpair_seg(seg, B, part, dst0, n0, dst1, n1, G, Bg, Cb, P, s, 0, defer);
}

}  // namespace msgat
