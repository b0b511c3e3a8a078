// The tail of a training step: loss + metric sums in one pass over the prediction, and
// Adam over flat state buffers in one launch.
//
//   reference            engine.py:56 HuberLoss (loss.py:51-52: mean of 0.5 e^2 inside delta, delta |e| - 0.5 delta^2
//                        beyond), engine.py:66-70 + metrics.py:20-35: per batch `loss.item()` and three more `.item()`
//                        sums (|e|, 100 |e / y| where y > mask, e^2) -- four host syncs and ~10 elementwise / reduce
//                        launches; engine.py:106 optim.Adam(lr 1e-3, weight_decay 5e-4).
//   here                 k_huber_metrics: every block folds its elements into four double partials (fixed order inside
//                        the block), k_huber_finish adds the partials in block order: deterministic, nothing read back.
//                        k_huber_grad: the loss gradient.  k_adam: torch.optim.Adam's update (L2 decay folded into the
//                        gradient, bias correction, eps outside the square root) for every parameter tensor at once,
//                        driven by a chunk table; the step counter and the learning rate live in device memory so the
//                        launch is capturable in a HIP graph.
#include "common.hpp"

namespace msgat {

constexpr int kTailBlock = 256;
constexpr int kTailPerThread = 8;  // elements per lane per block trip

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(kTailBlock) void k_huber_metrics(const float* __restrict__ pred,
                                                              const float* __restrict__ truth, long long n,
                                                              float delta, float mask_value,
                                                              double* __restrict__ part) {
  __shared__ double red[4][kTailBlock / 64];
  double s[4] = {0.0, 0.0, 0.0, 0.0};  // huber, |e|, 100 |e / y| (y > mask), e^2
  const long long stride = (long long)gridDim.x * kTailBlock;
  for (long long i = (long long)blockIdx.x * kTailBlock + threadIdx.x; i < n; i += stride) {
    const float p = pred[i], y = truth[i];
    const float e = p - y, a = fabsf(e);
    s[0] += (a <= delta) ? 0.5f * a * a : delta * a - 0.5f * delta * delta;
    s[1] += a;
    if (y > mask_value) s[2] += 100.0 * (double)fabsf(e / y);
    s[3] += (double)e * (double)e;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double w = wave_sum(s[k]);
    if (lane == 0) red[k][wave] = w;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kTailBlock / 64; ++w) t += red[threadIdx.x][w];
    part[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
  }
}

// loss[0] = mean Huber loss (fp32, what the reference's loss tensor holds); sums[0..2] += AE, APE, SE and
// sums[3] += loss_weight * that mean (the epoch's running total of batch losses; a rank that holds n_r of a global
// batch's n_b samples passes n_r / n_b, so the sum over ranks is the global batch's mean loss), all double
__global__ void k_huber_finish(const double* __restrict__ part, int nblocks, long long n, float* __restrict__ loss,
                               double* __restrict__ sums, float loss_weight) {
  // wave k adds sum k: lane l takes blocks l, l + 64, ... in order, then the lanes are combined by the fixed butterfly
  // of wave_sum (a serial loop over up to 1024 partials by one lane took 19 us)
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double t = 0.0;
  for (int b = lane; b < nblocks; b += 64) t += part[(size_t)b * 4 + k];
  t = wave_sum(t);
  if (lane != 0) return;
  if (k == 0) {
    loss[0] = (float)(t / (double)n);
    if (sums != nullptr) sums[3] += (t / (double)n) * (double)loss_weight;
  } else if (sums != nullptr) {
    sums[k - 1] += t;
  }
}

__global__ __launch_bounds__(kTailBlock) void k_huber_grad(const float* __restrict__ pred,
                                                           const float* __restrict__ truth,
                                                           const float* __restrict__ dloss, long long n, float delta,
                                                           float* __restrict__ dpred) {
  const long long i = (long long)blockIdx.x * kTailBlock + threadIdx.x;
  if (i >= n) return;
  const float e = pred[i] - truth[i];
  const float g = fminf(fmaxf(e, -delta), delta);  // d/de of the Huber function
  dpred[i] = g * (dloss[0] / (float)n);
}

static int huber_blocks(long long n) {
  const long long want = (n + (long long)kTailBlock * kTailPerThread - 1) / ((long long)kTailBlock * kTailPerThread);
  return (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
}

size_t huber_partial_doubles(long long n) { return (size_t)huber_blocks(n) * 4; }

int launch_huber_metrics(const float* pred, const float* truth, long long n, float delta, float mask_value,
                         double* part, float* loss, double* sums, float loss_weight, hipStream_t s) {
  const int nb = huber_blocks(n);
  hipLaunchKernelGGL(k_huber_metrics, dim3(nb), dim3(kTailBlock), 0, s, pred, truth, n, delta, mask_value, part);
  MSGAT_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_huber_finish, dim3(1), dim3(256), 0, s, part, nb, n, loss, sums, loss_weight);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_huber_grad(const float* pred, const float* truth, const float* dloss, long long n, float delta,
                      float* dpred, hipStream_t s) {
  hipLaunchKernelGGL(k_huber_grad, dim3((unsigned)((n + kTailBlock - 1) / kTailBlock)), dim3(kTailBlock), 0, s, pred,
                     truth, dloss, n, delta, dpred);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- the gated sum over the components (msgat.py:203-205, embeddings.py:36-39) --------------------------------------
//   out[b,e] = sum_r pred[r,b,e] * (h_w[H[b]][r,e] + d_w[D[b]][r,e])        e = (node, output step)
// The reference runs two embedding gathers, an add, R multiplies and R - 1 adds forward and, backward, R + R multiplies,
// a transposing copy and two dense embedding gradients (a zero fill and a scatter each): 17 launches for 4 MB tensors.
// Here one launch each way.  Indices are clamped into the tables (torch's gather would trap on the device instead).
// Static gate (msgat.py:189): H = D = nullptr, h_w = W[R,E] as a one-row table, d_w = nullptr.
constexpr int kGateMaxB = 1024;   // samples whose table rows a block of the backward keeps in LDS

__device__ __forceinline__ int gate_row(const long long* idx, int b, int rows) {
  if (idx == nullptr) return 0;
  const long long v = idx[b];
  return (int)(v < 0 ? 0 : (v >= rows ? rows - 1 : v));
}

__global__ __launch_bounds__(kTailBlock) void k_gate_sum(const float* __restrict__ pred, const long long* __restrict__ H,
                                                         const long long* __restrict__ D, const float* __restrict__ h_w,
                                                         const float* __restrict__ d_w, float* __restrict__ out, int R,
                                                         int B, int E, int nh, int nd) {
  const long long i = (long long)blockIdx.x * kTailBlock + threadIdx.x;
  if (i >= (long long)B * E) return;
  const int b = (int)(i / E), e = (int)(i - (long long)b * E);
  const float* hr = h_w + (size_t)gate_row(H, b, nh) * R * E + e;
  const float* dr = d_w ? d_w + (size_t)gate_row(D, b, nd) * R * E + e : nullptr;
  float acc = 0.f;
  {
#pragma clang fp contract(off)   // the reference's ops round the product and the sum separately (hipcc would fuse them)
    for (int r = 0; r < R; ++r) {   // and add in this order: term_0 + term_1 + ...
      const float gate = dr ? hr[(size_t)r * E] + dr[(size_t)r * E] : hr[(size_t)r * E];
      const float term = pred[((size_t)r * B + b) * E + e] * gate;
      acc = r == 0 ? term : acc + term;
    }
  }
  out[i] = acc;
}

// blocks [0, nblkA): dpred[r,b,e] = dout[b,e] * gate; the rest: a table row's gradient, summed over the samples that
// selected it in sample order (thread per (row, r, e); rows [0, nh) of h_w, then [nh, nh + nd) of d_w)
__global__ __launch_bounds__(kTailBlock) void k_gate_sum_bwd(const float* __restrict__ dout, const float* __restrict__ pred,
                                                             const long long* __restrict__ H, const long long* __restrict__ D,
                                                             const float* __restrict__ h_w, const float* __restrict__ d_w,
                                                             float* __restrict__ dpred, float* __restrict__ dh_w,
                                                             float* __restrict__ dd_w, int R, int B, int E, int nh, int nd,
                                                             int nblkA) {
  if ((int)blockIdx.x < nblkA) {
    const long long i = (long long)blockIdx.x * kTailBlock + threadIdx.x;
    if (dpred == nullptr || i >= (long long)R * B * E) return;
    const int e = (int)(i % E), b = (int)((i / E) % B), r = (int)(i / ((long long)E * B));
    float gate = h_w[((size_t)gate_row(H, b, nh) * R + r) * E + e];
    if (d_w) gate += d_w[((size_t)gate_row(D, b, nd) * R + r) * E + e];
    dpred[i] = dout[(size_t)b * E + e] * gate;
    return;
  }
  // the samples' table rows, once per block in LDS (read from global inside the loop below -- B dependent round trips per
  // thread -- this half of the launch took 25 us for 4 MB tensors)
  __shared__ int rows_h[kGateMaxB], rows_d[kGateMaxB];
  const bool staged = B <= kGateMaxB;   // kernel-uniform
  if (staged) {
    for (int b = threadIdx.x; b < B; b += kTailBlock) {
      rows_h[b] = gate_row(H, b, nh);
      rows_d[b] = D ? gate_row(D, b, nd) : 0;
    }
    __syncthreads();
  }
  const long long i = (long long)(blockIdx.x - nblkA) * kTailBlock + threadIdx.x;
  const long long RE = (long long)R * E;
  if (i >= (long long)(nh + nd) * RE) return;
  int row = (int)(i / RE);
  const long long re = i - (long long)row * RE;
  const int r = (int)(re / E), e = (int)(re - (long long)r * E);
  const bool day = row >= nh;
  row -= day ? nh : 0;
  float* dst = day ? dd_w : dh_w;
  if (dst == nullptr) return;
  const long long* idx = day ? D : H;
  const int rows = day ? nd : nh;
  const int* staged_rows = day ? rows_d : rows_h;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const int rb = staged ? staged_rows[b] : gate_row(idx, b, rows);
    if (rb == row) acc = fmaf(dout[(size_t)b * E + e], pred[((size_t)r * B + b) * E + e], acc);
  }
  dst[(size_t)row * RE + re] = acc;
}

int launch_gate_sum(const float* pred, const long long* H, const long long* D, const float* h_w, const float* d_w, float* out,
                    int R, int B, int E, int nh, int nd, hipStream_t s) {
  const long long n = (long long)B * E;
  hipLaunchKernelGGL(k_gate_sum, dim3((unsigned)((n + kTailBlock - 1) / kTailBlock)), dim3(kTailBlock), 0, s, pred, H, D, h_w,
                     d_w, out, R, B, E, nh, nd);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_gate_sum_bwd(const float* dout, const float* pred, const long long* H, const long long* D, const float* h_w,
                        const float* d_w, float* dpred, float* dh_w, float* dd_w, int R, int B, int E, int nh, int nd,
                        hipStream_t s) {
  const long long nA = dpred ? (long long)R * B * E : 0, nB = (long long)(nh + (d_w ? nd : 0)) * R * E;
  const int nblkA = (int)((nA + kTailBlock - 1) / kTailBlock);
  const long long nblk = nblkA + (nB + kTailBlock - 1) / kTailBlock;
  if (nblk == 0) return MSGAT_OK;
  hipLaunchKernelGGL(k_gate_sum_bwd, dim3((unsigned)nblk), dim3(kTailBlock), 0, s, dout, pred, H, D, h_w, d_w, dpred, dh_w,
                     dd_w, R, B, E, nh, d_w ? nd : 0, nblkA);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- Adam ---------------------------------------------------------------------------------------------------------
// Chunk c covers chunk_len[c] <= kAdamChunk elements of parameter tensor chunk_tensor[c], starting at chunk_param[c];
// its gradient and moments sit at chunk_off[c] of the flat buffers.  steps[t] = updates tensor t has received (torch
// keeps one step count per parameter: a parameter without a gradient is skipped and its bias correction lags);
// lr[0] = learning rate.  Both live in device memory, so a captured launch follows the scheduler.
constexpr int kAdamChunk = 2048;

__global__ void k_adam_advance(float* __restrict__ steps, const int* __restrict__ active, int n_active) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_active) steps[active[i]] += 1.0f;
}

__global__ __launch_bounds__(kTailBlock) void k_adam(float* const* __restrict__ chunk_param,
                                                     const long long* __restrict__ chunk_off,
                                                     const int* __restrict__ chunk_len,
                                                     const int* __restrict__ chunk_tensor,
                                                     const float* __restrict__ grad, float* __restrict__ m,
                                                     float* __restrict__ v, const float* __restrict__ steps,
                                                     const float* __restrict__ lrp, double beta1d, double beta2d,
                                                     float eps, float weight_decay,
                                                     const float* __restrict__ grad_divisor) {
  const int c = blockIdx.x;
  float* p = chunk_param[c];
  const long long off = chunk_off[c];
  const int len = chunk_len[c];
  const float t = steps[chunk_tensor[c]], lr = lrp[0];
  // data-parallel step: the flat buffer holds sum_ranks(w_r g_r) and *grad_divisor = sum_ranks(w_r) (the last element
  // of the same all-reduced buffer); the division that used to be a pass of its own happens on the way in
  const bool scaled = grad_divisor != nullptr;
  const float div = scaled ? grad_divisor[0] : 1.f;
  // torch.optim.Adam (_single_tensor_adam): bias_correction = 1 - beta^t; step_size = lr / bc1;
  // denom = sqrt(v) / sqrt(bc2) + eps; p -= step_size * m / denom
  // (in double, as the host-side Python floats of the reference optimizer are: 1 - 0.999^t cancels badly in fp32)
  const double bc1 = 1.0 - pow(beta1d, (double)t), bc2 = 1.0 - pow(beta2d, (double)t);
  const float beta2 = (float)beta2d, omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);  // 1 - beta rounded once
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  for (int i = threadIdx.x; i < len; i += kTailBlock) {
    const float w = p[i];
    const float gin = scaled ? grad[off + i] / div : grad[off + i];
    const float g = fmaf(weight_decay, w, gin);
    float mi = m[off + i], vi = v[off + i];
    mi = fmaf(g - mi, omb1, mi);                      // exp_avg.lerp_(grad, 1 - beta1)
    vi = fmaf(g * g, omb2, vi * beta2);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[off + i] = mi;
    v[off + i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = w - step_size * (mi / denom);
  }
}

int launch_adam(float* const* chunk_param, const long long* chunk_off, const int* chunk_len, const int* chunk_tensor,
                int nchunks, const int* active, int n_active, const float* grad, float* m, float* v, float* steps,
                const float* lr, double beta1, double beta2, double eps, double weight_decay, const float* grad_divisor,
                hipStream_t s) {
  if (n_active > 0) {
    hipLaunchKernelGGL(k_adam_advance, dim3(cdiv(n_active, kTailBlock)), dim3(kTailBlock), 0, s, steps, active, n_active);
    MSGAT_CHECK_LAUNCH();
  }
  if (nchunks > 0) {
    hipLaunchKernelGGL(k_adam, dim3(nchunks), dim3(kTailBlock), 0, s, chunk_param, chunk_off, chunk_len, chunk_tensor,
                       grad, m, v, steps, lr, beta1, beta2, (float)eps, (float)weight_decay, grad_divisor);
    MSGAT_CHECK_LAUNCH();
  }
  return MSGAT_OK;
}

int adam_chunk_elems() { return kAdamChunk; }

// ---- gather of the gradients into the flat buffer a data-parallel step all-reduces -----------------------------------
// flat[chunk_off[c] + i] = scale * chunk_src[c][i]; block 0 also writes flat[weight_index] = scale: the rank's weight
// (its sample count) rides in the buffer's last element, so sum(w g) and sum(w) come out of ONE collective.  Replaces
// a multi-tensor copy + mul_ + fill_ over the 7.8 MB buffer by one pass.
__global__ __launch_bounds__(kTailBlock) void k_gather_scaled(const float* const* __restrict__ chunk_src,
                                                              const long long* __restrict__ chunk_off,
                                                              const int* __restrict__ chunk_len, float scale,
                                                              float* __restrict__ flat, long long weight_index) {
  const int c = blockIdx.x;
  const float* __restrict__ src = chunk_src[c];
  float* __restrict__ dst = flat + chunk_off[c];
  const int len = chunk_len[c];
  for (int i = threadIdx.x; i < len; i += kTailBlock) dst[i] = scale * src[i];
  if (c == 0 && threadIdx.x == 0 && weight_index >= 0) flat[weight_index] = scale;
}

int launch_gather_scaled(const float* const* chunk_src, const long long* chunk_off, const int* chunk_len, int nchunks,
                         float scale, float* flat, long long weight_index, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_scaled, dim3(nchunks), dim3(kTailBlock), 0, s, chunk_src, chunk_off, chunk_len, scale, flat,
                     weight_index);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

}  // namespace msgat
