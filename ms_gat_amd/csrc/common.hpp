// Shared helpers for the gfx950 kernels of libmsgat_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "msgat_hip.h"

#define MSGAT_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e__; \
  } while (0)

namespace msgat {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kMaxT = 16;          // timesteps supported: 4, 8, 12, 16
constexpr int kMaxC = 256;         // channels supported per side
constexpr int kLdsBudget = 64 * 1024;  // per-block LDS the slab kernels aim for (>= 2 blocks/CU)
constexpr int kLdsMax = 160 * 1024;

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4fma(float a, const float4& x, float4& acc) {
  acc.x = fmaf(a, x.x, acc.x);
  acc.y = fmaf(a, x.y, acc.y);
  acc.z = fmaf(a, x.z, acc.z);
  acc.w = fmaf(a, x.w, acc.w);
}
__device__ __forceinline__ float f4dot(const float4& a, const float4& b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}
// v_exp_f32: 2^x, 1 ulp, no denormal results (flushes to 0 below 2^-126) -- what the
// softmax needs; exp2f() from the device library adds range scaling we do not want.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// Host-side queries a launcher would otherwise repeat on every call (`hipGetDevice` + `hipDeviceGetAttribute` +
// `hipFuncSetAttribute` per launch are part of what makes a PEMSD4-sized step host-bound).  Both caches are per
// device and lock-free; a lost race only repeats an idempotent driver call.
constexpr int kMaxDevices = 32;
// the current device's slot of the per-device caches, or -1 when it has none (query failed, or an ordinal >= 32):
// such a device is never aliased to another one's slot -- its launches repeat the driver calls every time
inline int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  return dev;
}
// compute units of the current device (256 on MI355X, also the fallback when the query fails)
inline int device_cu_count() {
  static std::atomic<int> cached[kMaxDevices];
  const int slot = current_device_slot();
  int ncu = slot >= 0 ? cached[slot].load(std::memory_order_relaxed) : 0;
  if (ncu <= 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
      ncu = 256;
    if (slot >= 0) cached[slot].store(ncu, std::memory_order_relaxed);
  }
  return ncu;
}
// A kernel's dynamic-LDS ceiling (hipFuncAttributeMaxDynamicSharedMemorySize) has to be raised above 64 KiB once per
// device, not once per launch: `granted` is the call site's own (static, zero-initialised) record of what it has
// asked for so far.  Returns a status code.
struct LdsGrant {
  std::atomic<int> bytes[kMaxDevices];
};
template <typename K>
inline int grant_dynamic_lds(K kernel, size_t lds, LdsGrant& granted) {
  if (lds <= 64 * 1024) return MSGAT_OK;
  const int slot = current_device_slot();
  if (slot >= 0 && (int)lds <= granted.bytes[slot].load(std::memory_order_acquire)) return MSGAT_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return MSGAT_ERR_HIP_BASE - (int)e;
  if (slot >= 0) granted.bytes[slot].store((int)lds, std::memory_order_release);
  return MSGAT_OK;
}

// the MFMA projection's matrix (+ row-pointer table) may take this much LDS: two blocks per CU still fit 160 KiB
constexpr int kProjLdsMax = 78 * 1024;

inline bool t_supported(int T) { return T == 4 || T == 8 || T == 12 || T == 16; }

// how many channels of an [N,T] fp32 slab fit the per-block LDS budget
inline int slab_channels(int N, int T, int Cu, int budget_bytes) {
  const int64_t slab = (int64_t)N * T * 4;
  int ch = (int)(budget_bytes / slab);
  if (ch < 1) ch = (slab <= kLdsMax - 1024) ? 1 : 0;  // 0 => slab does not fit LDS at all
  if (ch > Cu) ch = Cu;
  return ch;
}

// the SELL kernels hold one float4 per node in LDS; they take over when no whole [N,T] slab fits (or on request)
inline bool sell_usable(const msgat_sell_t& j, int nnz, int N, int T) {
  if (j.n_slices <= 0 || nnz <= 0 || !j.slice_off || !j.lane_row || !j.idx || !j.src) return false;
  if ((size_t)N * 16 > (size_t)kLdsMax - 1024) return false;  // <= 10176 nodes = 159 slices = 10 per wave (kSellMaxSlices)
  return j.prefer != 0 || slab_channels(N, T, 1, kLdsBudget) == 0;
}

#if defined(__HIPCC__)
// Workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() makes hipcc drain vmcnt(0)
// first, which would serialise a register prefetch -- or an LDS-DMA ring -- against the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Counted wait: until all but the wave's N youngest vector-memory operations (loads, LDS-DMA and stores, in issue order)
// are done.  Safe as long as at least N operations were really issued behind the one that is waited for.
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif

// Diagnostic builds only (tools/*_stamps.hip compile a kernel file with -DMSGAT_STAMPS): s_memtime
// stamps of the phases of a block, written to a buffer nothing else reads.  The product library is
// built without the macro and contains no stamp.
#ifdef MSGAT_STAMPS
__device__ unsigned long long g_stamps[8 * 4096];
__device__ unsigned g_hwid[4096];  // (XCC_ID << 16) | HW_ID of the block's first wave
#define MSGAT_STAMP(i)                                                                             \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.z == 0) {                                                     \
      const unsigned b__ = blockIdx.y * gridDim.x + blockIdx.x;                                    \
      if (b__ < 4096) {                                                                            \
        g_stamps[b__ * 8 + (i)] = __builtin_amdgcn_s_memtime();                                    \
        if ((i) == 0) {                                                                            \
          g_stamps[b__ * 8 + 6] = __builtin_amdgcn_s_memrealtime();                                \
          g_hwid[b__] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16) |      \
                        (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) & 0xffff);    \
        }                                                                                          \
        if ((i) == 5) g_stamps[b__ * 8 + 7] = __builtin_amdgcn_s_memrealtime();                    \
      }                                                                                            \
    }                                                                                              \
  } while (0)
#else
#define MSGAT_STAMP(i)
#endif

// A channel axis assembled from up to kMaxSeg tensors ("segments"): channel c of group g lives in the
// segment k with begin[k] <= c < begin[k+1], at ptr[k] + (g * gstride[k] + c - begin[k]) * P.  gstride is
// the channel count of the tensor the segment is a slice of (>= its own width), so a segment may be a
// channel slice [:, a:b] of a wider contiguous [G,C,N,T] tensor.  This is what lets one channel-mixing
// pass read cat(...) without the concatenation and write its outputs into separate tensors.
constexpr int kMaxSeg = 6;
struct SegList {
  const float* ptr[kMaxSeg];
  int begin[kMaxSeg + 1];
  int gstride[kMaxSeg];
  int n;
  __host__ __device__ int total() const { return begin[n]; }
#if defined(__HIPCC__)
  // SEGS = false: the caller guarantees n == 1 and the lookup folds to one multiply-add
  template <bool SEGS = true>
  __device__ __forceinline__ const float* row(int g, int c, int P) const {
    if (!SEGS) return ptr[0] + ((size_t)g * gstride[0] + c) * P;
    const float* p = ptr[0];
    int b = begin[0], gs = gstride[0];
#pragma unroll
    for (int i = 1; i < kMaxSeg; ++i)
      if (i < n && c >= begin[i]) { p = ptr[i]; b = begin[i]; gs = gstride[i]; }
    return p + ((size_t)g * gs + (c - b)) * P;
  }
#endif
};
inline SegList seg_none() {
  SegList s{};
  return s;
}
inline SegList seg_single(const float* p, int C) {
  SegList s{};
  s.ptr[0] = p; s.begin[0] = 0; s.begin[1] = C; s.gstride[0] = C; s.n = 1;
  return s;
}
inline SegList seg_pair(const float* a, int Ca, const float* b, int Cb) {  // [a | b]
  SegList s = seg_single(a, Ca);
  if (b != nullptr && Cb > 0) { s.ptr[1] = b; s.begin[2] = Ca + Cb; s.gstride[1] = Cb; s.n = 2; }
  return s;
}

// Epilogue of the channel-mixing kernels: v + bias[r,co] + add[g,co,p], then ReLU -- what turns
// "W x" into a 1x1 convolution with bias, and the residual convolution of MEAM (msgat.py:130-131)
// into one pass: relu(cat(branches) + res(x)).
struct MixEpilogue {
  const float* bias = nullptr;   // [R?, Co]: element r * bias_rstride + co
  int bias_rstride = 0;
  SegList add = {};              // channel axis of the output, possibly several tensors (n = 0: none)
  int relu = 0;
#if defined(__HIPCC__)
  // same, with the add operand's address already resolved (nullptr: none)
  __device__ __forceinline__ float4 apply_rows(float4 v, int r, int co, const float4* addp) const {
    if (bias != nullptr) {
      const float b = bias[(size_t)r * bias_rstride + co];
      v.x += b; v.y += b; v.z += b; v.w += b;
    }
    if (addp != nullptr) {
      const float4 a = *addp;
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
  }
  template <bool SEGS = true>
  __device__ __forceinline__ float4 apply(float4 v, int r, int g, int co, int p4, int P4) const {
    if (bias != nullptr) {
      const float b = bias[(size_t)r * bias_rstride + co];
      v.x += b; v.y += b; v.z += b; v.w += b;
    }
    if (add.n > 0) {
      const float4 a = reinterpret_cast<const float4*>(add.template row<SEGS>(g, co, 4 * P4))[p4];
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
  }
#endif
};

// ---- kernel launchers (defined in the .hip files) ---------------------------------
int launch_qonly(const float* x, const float* alpha, float* q, int G, int Bg, int C, int P,
                 hipStream_t s);
// out[g,co,p] = sum_ci M(r; ci->co) in[g,ci,p] (+ addvec[r,co]*extra[g,p]);  q[g,p] = sum_ci qvec[r,ci] in[g,ci,p]
int launch_project(const float* in, const float* M, int m_in_major, const float* qvec,
                   const float* addvec, const float* extra, float* out, float* q, int G, int Bg,
                   int Ci, int Co, int P, hipStream_t s);
int launch_project_epi(const float* in, const float* M, int m_in_major, const float* qvec,
                       const float* addvec, const float* extra, float* out, float* q, int G, int Bg,
                       int Ci, int Co, int P, const MixEpilogue& epi, hipStream_t s);
// the general form: input and output channel axes are segment lists (Ci = in.total(), Co = out.total())
int launch_project_seg(const SegList& in, const float* M, int m_in_major, const float* qvec, const float* addvec,
                       const float* extra, const SegList& out, float* q, int G, int Bg, int P,
                       const MixEpilogue& epi, hipStream_t s);
// MFMA forms (mfma.hip); launch_project / launch_chanpair dispatch to them
size_t project_mfma_lds_bytes(int Ci, int Co, bool has_extra);
int launch_project_mfma(const SegList& in, const float* M, int m_in_major, const float* qvec,
                        const float* addvec, const float* extra, const SegList& out, float* q, int G, int Bg,
                        int P, const MixEpilogue& epi, hipStream_t s);
int chanpair_mfma_blocks(int R);  // blocks (= partials) per relation
// b_ones: B's last channel (Cb counts it) is a virtual row of ones: part[a, Cb-1] = sum_p A[a,p]
// nblk: partial blocks per relation the buffer has room for; *nblk_used: how many the launch wrote (fewer when the
// channel matrix is cut into several z-blocks, which all run at once)
int launch_chanpair_mix(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                        const float* Mw, const float* Mlast, float* mixout, hipStream_t s, int* nblk_used, int* done);
int launch_chanpair_mix_wide(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                             int b_ones, const float* M, float* mixout, hipStream_t s, int* nblk_used, int* done);
int launch_chanpair_mfma(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                         int b_ones, hipStream_t s, int* nblk_used);
int launch_chanpair_shifted(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                            int b_ones, int dilation, int T, hipStream_t s, int* nblk_used);
bool project_taps_supported(int Cr, int Co);
int launch_project_taps(const float* in, int in_gstride, const float* taps, int m_in_major, const float* bias,
                        int bias_rstride, float* out, int G, int Bg, int Cr, int Co, int P, int T, int tshift, hipStream_t s);
int contract_form_name(int Ca, int Cb, int b_ones, int P, int with_mix, char* buf, int buflen, int* one_pass, int* nza,
                       int* nzb);
// x / alpha / C / qout given (scores_take_x): q = alpha . x is computed inside the kernel from x[G,C,N,T] and written to
// qout -- no k_qonly launch; `q` is ignored then.  apW / apCo / apZ given as well (AGG_FIRST, W[R,apCo,C]): the kernel
// also finishes the layer for its rows, y = E x (to apY[G,C,N,T] when not NULL) and z = W y (to apZ[G,apCo,N,T]) -- no
// k_agg_proj launch.
bool scores_take_x(int C, bool with_tail);
int launch_scores(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse,
                  float* pq, float* E, float* Ec, int G, int Bg, int N, int T, hipStream_t s,
                  const float* x = nullptr, const float* alpha = nullptr, int C = 0, float* qout = nullptr,
                  const float* apW = nullptr, int apCo = 0, float* apY = nullptr, float* apZ = nullptr, void* scratch = nullptr);
// The dense passes (launch_scores, launch_bwd_dense_col) of large graphs run on the bf16 / fp16 matrix core with split
// operands (dense_bf16.hip) and then need `scratch` (dense_scratch_bytes() > 0): the operand images of one pass.
bool dense_split_selected(int N, int T);
size_t dense_scratch_bytes(int G, int N, int T);
// v[g,c,n,:] = sum_{e in ptr[n]..ptr[n+1]} E[g,e] u[g,c,idx[e],:] (+ addvec[r,c]*extra[g,n,:])
// sell != nullptr: E is in that layout's position order (permuted by sell->src, row stride sell->n_pos, with
// MSGAT_SELL_SLACK readable floats behind the last row) and the SELL kernel runs
// xdot/dap (both or neither; Cu <= kAggDotMaxC, addvec given): the slab kernel also leaves dap[g,c] =
// sum_p extra[g,p] xdot[g,c,p] -- dalpha = dq . x of the AGG_FIRST / PLAIN backward -- when it runs (returns
// MSGAT_OK and sets *dot_done to the number of partials per group it left: dap[(g * dot_done + i) * Cu + c]); other
// kernel forms leave *dot_done = 0 and the caller contracts separately
constexpr int kAggDotMaxC = 4;
constexpr int kAggMaxSplit = 3;   // blocks that may share one slab's output rows (k_agg_lds)
int launch_aggregate(const int* ptr, const int* idx, int nnz, const msgat_sell_t* sell, const float* u,
                     const float* E, const float* addvec, const float* extra, float* v, int G, int Bg, int Cu,
                     int N, int T, hipStream_t s, const float* xdot = nullptr, float* dap = nullptr,
                     int* dot_done = nullptr);
// Eo[g,k] = E[g, perm[k]] for k < n (0 where perm[k] < 0): edge coefficients in CSC or SELL position order
int launch_permute_edges(const float* E, const int* perm, float* Eo, int G, int nnz, int n, hipStream_t s);
int launch_aggregate_project(const msgat_graph_t& gr, const float* x, const float* E,
                             const float* W, float* y, float* z, int G, int Bg, int C, int Co,
                             int N, int T, hipStream_t s);
// dEp[g,k,e] = sum over channel chunk k of <dv[g,c,erow[e],:], u[g,c,col[e],:]>
// (with gr.sell_rows usable the partials are [G,chunks,n_pos] in SELL position order + MSGAT_SELL_SLACK floats
// of slack, and launch_bwd_rows reads them through sell.pos)
int sddmm_chunks(int G, int Cu, int N, int T, const msgat_sell_t* sell);
int launch_sddmm(const msgat_graph_t& gr, const float* u, const float* dv, float* dEp, int G,
                 int Cu, int N, int T, hipStream_t s);
// PROJ_FIRST backward, one pass over dv: du = E^T dv on the CSC (Ec = E in CSC order) AND the SDDMM partials, which
// come out in CSC order (launch_bwd_rows: partials_in_csc).  Usable when agg_sddmm_fusable(); chunks = sddmm_chunks().
bool agg_sddmm_fusable(const msgat_graph_t& gr, int N, int T, int Cu);
int launch_agg_sddmm(const msgat_graph_t& gr, const float* dv, const float* Ec, const float* u, float* du, float* dEp,
                     int G, int Cu, int N, int T, hipStream_t s, int dv_group_channels = 0);
int launch_bwd_dense_col(const msgat_graph_t& gr, const float* q, const float* kW,
                         const float* lse, const float* delta, const float* gE, float* dq, int G,
                         int N, int T, hipStream_t s, void* scratch = nullptr);
// A fixed-order sum of partials, out[r,i] = sum_j part[r,j,i] split over dst0 (first n0 columns) and dst1 (next n1),
// that may wait for the end of a backward pass: the launchers below queue their reduction in `defer` when given one,
// and launch_reduce_jobs runs the queue as ONE launch (each reduction alone is a 4-5 us launch of a few blocks).
struct ReduceJob {
  const float* part;
  int R, J, Wd;
  float* dst0;
  int n0;
  float* dst1;
  int n1;
  int rw = 0;   // > 0: the Wd sums are rows of rw columns; a row's first rw - 1 go to dst0 [R, Wd/rw, rw - 1], its last to
                // dst1 [R, Wd/rw] (a bias gradient riding as the last column of a matrix gradient, delivered apart)
};
constexpr int kMaxReduceJobs = 4;
struct ReduceJobs {
  ReduceJob job[kMaxReduceJobs];
  int n;
};
int launch_reduce_jobs(const ReduceJobs& jobs, hipStream_t s);
// The edge and row work of backward (scores.hip): g_e = E_e dE_e (an edge-parallel pass over the SDDMM partials --
// in CSC order with E = Ecsc when Ecsc != nullptr -- or, direct_c > 0, computed inside the row pass from u / dv with
// direct_c <= bwd_rows_direct_max_channels() channels: no SDDMM launch at all), then in ONE launch delta, dkW, the row-local part of dq, and the per-block partials of
// dWg[r,t,s] = sum_{g in r, n} q[g,n,t] dkW[g,n,s] (dwg_part: dwg_partial_floats() floats; reduction queued in `defer`).
size_t dwg_partial_floats(int G, int N, int T);
int bwd_rows_direct_max_channels();
int launch_bwd_rows(const msgat_graph_t& gr, const float* dEp, int nchunks, const float* Ecsc, int direct_c,
                    const float* u, const float* dv, const float* E, const float* q, const float* pq, const float* Wg,
                    float* gE, float* delta, float* dkW, float* dq, float* dwg_part, float* dWg, int G, int Bg, int N,
                    int T, hipStream_t s, ReduceJobs* defer = nullptr);
// out[r,a,c] = sum_{g in r, p} A(g,a,p) B[g,c,p];  channel a == Ca-1 comes from Aextra[g,p] when given
size_t chanpair_partial_floats(int G, int Bg, int Ca, int Cb);
int launch_chanpair(const float* A, const float* Aextra, const float* B, float* part, float* dst0,
                    int n0, float* dst1, int n1, int G, int Bg, int Ca, int Cb, int P,
                    hipStream_t s, ReduceJobs* defer = nullptr);
// dW | dalpha = [du | dq] x^T and dx = W^T du + alpha (x) dq in ONE pass over du, dq and x (PROJ_FIRST backward); *done = 0
// and nothing launched when the fused form does not cover the shape
int launch_chanpair_mix(const float* du, const float* dq, const float* x, const float* W, const float* alpha, float* dx,
                        float* part, float* dW, float* dalpha, int G, int Bg, int Co, int C, int P, hipStream_t s,
                        ReduceJobs* defer, int* done);
int launch_chanpair_seg(const SegList& A, const float* B, float* part, float* dst0, int n0, float* dst1, int n1,
                        int G, int Bg, int Cb, int P, hipStream_t s, int b_ones = 0, ReduceJobs* defer = nullptr,
                        int lastcol_apart = 0);
// AGG_FIRST backward with few input channels (C <= kAggFirstMaxC): dy = W^T dz and the partials of dW = dz y^T in ONE
// pass over dz (project.hip); partials [G * aggfirst_blocks(P)][Co*C], summed per relation by the queued job
constexpr int kAggFirstMaxC = 4;
int aggfirst_blocks(int P);
int launch_aggfirst_bwd(const float* dz, const float* W, const float* y, float* dy, float* part, float* dW, int G,
                        int Bg, int C, int Co, int P, hipStream_t s, ReduceJobs* defer, int dz_group_channels = 0, int ones = 0);
// out[i] = sum_j part[j,i], i < Wd, split over dst0 (first n0) and dst1 (next n1); fixed order
int launch_reduce_rows(const float* part, int J, int Wd, float* dst0, int n0, float* dst1, int n1,
                       hipStream_t s);
// out[r,i] = sum_j part[r,j,i]
int launch_reduce_groups(const float* part, int R, int J, int Wd, float* dst, hipStream_t s);
// out[r] = sum_j part[r,j] of [rows, rw] matrices, written as [R, rows, rw - 1] at dst followed by [R, rows] (the last column)
int launch_reduce_lastcol(const float* part, int R, int J, int rows, int rw, float* dst, hipStream_t s, ReduceJobs* defer = nullptr);
int launch_reduce_groups_defer(const float* part, int R, int J, int Wd, float* dst, hipStream_t s, ReduceJobs* defer);
// temporal / channel branch kernels (branches.hip)
int launch_tmix(const float* src, const float* A, int per_group, const float* bias, float* dst, int G, int Co,
                int K, int N, int T, int backward, int R, hipStream_t s, int src_gs = 0);
size_t tmix_partial_floats(int G, int K, int T);
int launch_tmix_dA(const float* dout, const float* y, float* dA, float* part, int G, int Co, int K, int N, int T,
                   hipStream_t s, int dout_gs = 0);
int launch_node_pool(const float* x, const float* w, float* pooled, long long slabs, int N, int T, int R,
                     hipStream_t s, int C = 0, int gs = 0);
int launch_node_pool_dx(const float* w, const float* dp, const float* add, float* dx, long long slabs, int N, int T,
                        int R, hipStream_t s);
size_t node_pool_partial_floats(int G, int C, int N);
int launch_node_pool_dw(const float* x, const float* dp, float* dw, float* part, int G, int C, int N, int T, int R,
                        hipStream_t s);
size_t head_fwd_partial_floats(int B, int C, int N, int To);
size_t head_dw_partial_floats(int C, int T, int To, int R);
size_t lnhead_partial_floats(int B, int C, int N, int T);
int launch_lnhead_bwd(const float* dout, const float* W, const float* x, const float* lnw, float* dx, float* dlnw,
                      float* dlnb, float* part, int B, int C, int N, int T, int To, int R, float eps, int relu_mask,
                      hipStream_t s);
int launch_head_fwd(const float* x, const float* W, const float* bias, float* out, float* part, int B, int C, int N,
                    int T, int To, int R, hipStream_t s,
                    int ln = 0, const float* lnw = nullptr, const float* lnb = nullptr, float eps = 0.f, float* xn = nullptr);
int launch_head_dx(const float* dout, const float* W, float* dx, int B, int C, int N, int T, int To, int R,
                   hipStream_t s);
int launch_head_dW(const float* dout, const float* x, float* dWc, float* part, int B, int C, int N, int T, int To,
                   int R, hipStream_t s);
// LayerNorm over the last axis of [rows, T] (layernorm.hip)
size_t layernorm_partial_floats(long long rows, int T, int R);
int launch_reduce_split(const float* part, int R, int J, int Wd, float* dst0, int n0, float* dst1, int n1,
                        hipStream_t s, ReduceJobs* defer = nullptr);
int launch_layernorm_fwd(const float* x, const float* w, const float* b, float* y, long long rows, int T,
                         float eps, int R, hipStream_t s,
                         const float* pool_w = nullptr, float* pool_part = nullptr, float* pooled = nullptr, int N = 1);
size_t layernorm_pool_partial_floats(long long rows, int T, int R);
int launch_layernorm_bwd(const float* x, const float* w, const float* dy, const float* add, float* dx, float* dw,
                         float* db, float* part, long long rows, int T, float eps, int R, int relu_mask, hipStream_t s,
                         const float* pool_w = nullptr, const float* dpooled = nullptr, int N = 1, const float* lnb = nullptr,
                         float* dpw_rows = nullptr, float* dpw = nullptr);

// the tiny attention matrices of a MEAM block (smallatt.hip)
size_t chanatt_partial_floats(int G, int C, int cb, int T);
int launch_chanatt_fwd(const float* pooled, const float* Wc, const float* conv, float* att, float* Mc, int G, int R,
                       int C, int cb, int T, hipStream_t s);
int launch_chanatt_bwd(const float* dMc, const float* att, const float* pooled, const float* Wc, const float* conv,
                       float* dpooled, float* dWc, float* dconv, float* part, int G, int R, int C, int cb, int T,
                       hipStream_t s);
size_t tempatt_partial_floats(int G, int K, int N);
int launch_tempatt_fwd(const float* pooled, const float* Wt1, const float* Wt2, float* lr, float* att, float* taps,
                       int G, int R, int N, int K, int T, int dil, hipStream_t s);
int launch_tempatt_bwd(const float* dtaps, const float* att, const float* lr, const float* pooled, const float* Wt1,
                       const float* Wt2, float* dpooled, float* dWt1, float* dWt2, float* part, int G, int R, int N,
                       int K, int T, int dil, hipStream_t s);
// step tail (tail.hip)
int launch_gate_sum(const float* pred, const long long* H, const long long* D, const float* h_w, const float* d_w, float* out,
                    int R, int B, int E, int nh, int nd, hipStream_t s);
int launch_gate_sum_bwd(const float* dout, const float* pred, const long long* H, const long long* D, const float* h_w,
                        const float* d_w, float* dpred, float* dh_w, float* dd_w, int R, int B, int E, int nh, int nd,
                        hipStream_t s);
size_t huber_partial_doubles(long long n);
int launch_huber_metrics(const float* pred, const float* truth, long long n, float delta, float mask_value,
                         double* part, float* loss, double* sums, float loss_weight, hipStream_t s);
int launch_huber_grad(const float* pred, const float* truth, const float* dloss, long long n, float delta,
                      float* dpred, hipStream_t s);
int adam_chunk_elems();
int launch_adam(float* const* chunk_param, const long long* chunk_off, const int* chunk_len, const int* chunk_tensor,
                int nchunks, const int* active, int n_active, const float* grad, float* m, float* v, float* steps,
                const float* lr, double beta1, double beta2, double eps, double weight_decay, const float* grad_divisor,
                hipStream_t s);
int launch_gather_scaled(const float* const* chunk_src, const long long* chunk_off, const int* chunk_len, int nchunks,
                         float scale, float* flat, long long weight_index, hipStream_t s);

}  // namespace msgat
