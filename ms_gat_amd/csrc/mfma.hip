// The two GEMM-shaped pieces of the hot path on the matrix cores (exact fp32 MFMA,
// v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, so the 1e-4 bar is untouched).
//
//   k_project_mfma   out[g,co,p] = sum_ci M(co,ci) in[g,ci,p] (+ addvec[co] extra[g,p]),  q = alpha . in
//                    forward: u = W x (msgat.py:27 applied before the aggregation), q (attention.py:33)
//                    backward: dy = W^T dz, dx = W^T du + alpha (x) dq
//   k_chanpair_mfma  part[a,c] = sum_p A[g,a,p] B[g,c,p]   (dW = du x^T, dalpha = dq . x, dW = dz y^T)
//
// Both stream their operands from HBM in the reference's [B,C,N,T] layout (a (group, channel)
// slab is contiguous over p = n*T + t) and are HBM-bound by design.  What limits such a kernel
// on CDNA4 is bytes in flight per SIMD, so the small 16x16 tile (4 accumulator registers) is used:
// it keeps 3-4 waves per SIMD resident, each with 8 x 1 KiB loads outstanding.
//   - projection: positions ride on the MFMA's N axis.  Lane (j = lane & 15, kq = lane >> 4)
//     loads one float4 = 4 consecutive positions of channel 4k + kq; its 4 components feed 4
//     independent 16x16 tiles, so one 16-B load per lane drives 4 MFMAs per 16 output channels
//     and the 4 result tiles re-assemble into float4 stores.  The matrix is the A operand (LDS).
//   - contraction: positions ride on K and channels on the lanes, the worst case for global
//     loads, so 256-position tiles of all rows are staged through LDS in 1-KiB row pieces by a
//     persistent split-K kernel (see k_chanpair_mfma).
#include "common.hpp"
#include <cstdio>
#ifdef MSGAT_LAB
#include <cstdlib>
#endif

namespace msgat {

// msgat_contract_form_name(): the launchers below run as usual up to the point where they would touch the device, and
// the leaf that would launch writes its kernel's name here instead.  Thread-private, set only by that query.
struct FormProbe {
  char name[96];
  int nza, nzb;
};
static thread_local FormProbe* g_form_probe = nullptr;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// ---------------------------------------------------------------------------------------------
// projection
// ---------------------------------------------------------------------------------------------
constexpr int kKC = 8;  // k-steps (groups of 4 input channels) per register buffer
constexpr int kProjMaxMG = 7;

__host__ __device__ static inline int proj_kpad(int Kx) {
  const int K4 = (Kx + 3) & ~3;
  return K4 + 2;  // Kpad/2 odd: the 16 rows x 2 k-quarters of a half-wave fragment read hit 32 banks
}

// MG = output-channel tiles (of 16) held in accumulators per pass over the input channels.
// The k-loop keeps kKC loads in flight per wave in a register ring: slot i is re-issued for k-step
// k + kKC right after k-step k consumed it.  Every load is unconditional (addresses are clamped,
// padding is neutralised by zero matrix entries): a load inside a branch makes hipcc fall back
// to s_waitcnt vmcnt(0), which serialises the prefetch against the MFMAs.
// Global-address-space views of pointers that reach the kernel through LDS (the row-pointer table) or through a
// select between kernel arguments: hipcc cannot prove such a pointer global and falls back to flat_load, which counts
// on BOTH vmcnt and lgkmcnt -- every LDS operand read of the MFMAs then waits for the whole register ring
// (s_waitcnt vmcnt(0) lgkmcnt(0) at the top of each chunk: load burst, drain, compute, instead of a ring).
typedef const f32x4 __attribute__((address_space(1)))* gf4_in;   // (a builtin vector: float4 is a class, whose
typedef f32x4 __attribute__((address_space(1)))* gf4_out;         //  copy constructor only binds generic references)
__device__ __forceinline__ float4 load_global(const float* p) {
  const f32x4 v = *(gf4_in)(const f32x4*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_global(const float* p, const float4& v) {
  const f32x4 w = {v.x, v.y, v.z, v.w};
#ifdef MSGAT_PROJ_NT
  __builtin_nontemporal_store(w, (gf4_out)(f32x4*)const_cast<float*>(p));
#else
  *(gf4_out)(f32x4*)const_cast<float*>(p) = w;
#endif
}

#ifndef MSGAT_PROJ_LB
#define MSGAT_PROJ_LB 2   // waves per SIMD the register allocation must leave room for (= blocks per CU)
#endif
// TAPS: a causal dilated [1,2] convolution (msgat.py:69-74 behind a Chomp; its autograd with tshift > 0) in ONE pass:
//   out[co,n,t] = sum_ci W0[co,ci] in[ci,n,t + tshift] + W1[co,ci] in[ci,n,t]     (terms outside 0 <= t + tshift < T are zero)
// The input's Cr rows appear as 2 Cr virtual channels -- the first Cr shifted along time, the second Cr plain -- under
// the matrix [W0 | W1]; a lane's float4 is 4 consecutive timesteps of ONE row of T (T % 4 == 0), so the shifted
// operand is one unaligned 16-B load of the same row plus a lane-constant element mask.  Replaces the channel mixing
// Cr -> 2 Co, the [G,2Co,N,T] intermediate and the time-mixing pass of every TACN layer whose taps are constant shifts.
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef const f32x4_a4 __attribute__((address_space(1)))* gf4_in_a4;
__device__ __forceinline__ float4 load_global_a4(const float* p) {   // dword-aligned 16-B load
  const f32x4_a4 v = *(gf4_in_a4)(const f32x4_a4*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}

template <int MG, bool DO_Q, bool ONEPASS, bool SEGS, bool HAS_ADD, bool TAPS = false>
__global__ __launch_bounds__(kBlock, MSGAT_PROJ_LB) void k_project_mfma(
    SegList in, const float* __restrict__ M, int m_in_major,
    const float* __restrict__ qvec, const float* __restrict__ addvec,
    const float4* __restrict__ extra4, SegList out, float4* __restrict__ q4, int Bg,
    int P4, MixEpilogue epi, int tshift = 0, int T = 4) {
  extern __shared__ float lds[];
  const int Cr = in.total();                  // rows the input really has
  const int Ci = TAPS ? 2 * Cr : Cr, Co = out.total();
  const bool has_extra = addvec != nullptr;
  const int Kx = Ci + (has_extra ? 1 : 0);  // the extra "channel" carries addvec (x) extra
  const int K4 = (Kx + 3) >> 2;             // k-steps
  const int Kpad = proj_kpad(Kx);
  const int Mt = cdiv(Co, 16);
  const int Mrows = cdiv(Mt, MG) * MG * 16;  // rows padded to whole passes (zeros)
  float* Wl = lds;                           // [Mrows][Kpad]
  float* ql = lds + Mrows * Kpad;            // [4*K4]
  float* bl = ql + 4 * K4;                   // [Mrows] bias of every output row (0 without one)
  const int g = blockIdx.y;
  const int r = g / Bg;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int p4 = (blockIdx.x * 4 + wave) * 16 + j;
  const bool pvalid = p4 < P4;
  const int p4c = min(p4, P4 - 1);  // out-of-range lanes re-read the last position; their stores are masked
  const float4* ex = has_extra ? extra4 + (size_t)g * P4 + p4c : nullptr;   // already at this lane's position

  // SEGS: the segment lists (3 x 26 scalars) are resolved ONCE per block into a row-pointer table in LDS
  // -- entry c = address of channel c of this group.  Looked up per load from the kernel arguments they
  // spilled the scalar registers and halved the occupancy (203 us for a pass that takes 66 us unsegmented).
  const float** rowtab = reinterpret_cast<const float**>(lds + ((Mrows * Kpad + 4 * K4 + Mrows + 1) & ~1));  // [Ci | Co | Co]
  if (SEGS) {
    for (int c = threadIdx.x; c < Ci; c += kBlock) rowtab[c] = in.row(g, c, 4 * P4);
    for (int c = threadIdx.x; c < Co; c += kBlock) {
      rowtab[Ci + c] = out.row(g, c, 4 * P4);
      rowtab[Ci + Co + c] = HAS_ADD ? epi.add.row(g, c, 4 * P4) : nullptr;
    }
    __syncthreads();
  }

  // channel 4*kk + kq of this lane's 4 positions; padding channels alias a real one (their matrix
  // column is zero) -- never a branch
  // TAPS: lane constants of the shifted operand.  Element e of the lane's float4 is timestep t0 + e of its row; the
  // shifted value in[t0 + e + tshift] exists iff 0 <= t0 + e + tshift < T.  At the two ends of a channel row the
  // unaligned load would leave the row (and, for the first / last row of the tensor, the allocation): there the lane
  // loads its own (aligned) float4 instead and moves the elements in registers.
  const int t0 = TAPS ? (4 * p4c) % T : 0;
  const int spos = 4 * p4c + tshift;                                  // first element of the shifted float4
  const int sfix = TAPS ? (spos < 0 ? 1 : (spos > 4 * P4 - 4 ? 2 : 0)) : 0;
  float4 km = make_float4(1.f, 1.f, 1.f, 1.f);
  if (TAPS) {
    km.x = (t0 + 0 + tshift >= 0 && t0 + 0 + tshift < T) ? 1.f : 0.f;
    km.y = (t0 + 1 + tshift >= 0 && t0 + 1 + tshift < T) ? 1.f : 0.f;
    km.z = (t0 + 2 + tshift >= 0 && t0 + 2 + tshift < T) ? 1.f : 0.f;
    km.w = (t0 + 3 + tshift >= 0 && t0 + 3 + tshift < T) ? 1.f : 0.f;
  }
  const int sabs = tshift < 0 ? -tshift : tshift;
  auto loadB = [&](int kk) -> float4 {
    const int ci = 4 * min(kk, K4 - 1) + kq;
    if (TAPS) {   // raw load only: what the value still needs (tap_fix) happens when it is consumed, not in front of the ring
      const int cv = min(ci, Ci - 1);                // virtual channel: < Cr shifted, >= Cr plain
      const bool sh = cv < Cr;
      const float* base = in.template row<false>(g, sh ? cv : cv - Cr, 4 * P4);
      return load_global_a4(base + ((sh && sfix == 0) ? spos : 4 * p4c));
    }
    const float* base = SEGS ? rowtab[min(ci, Ci - 1)] : in.template row<false>(g, min(ci, Ci - 1), 4 * P4);
    const float* p = (ci == Ci && has_extra) ? reinterpret_cast<const float*>(ex) : base + 4 * (size_t)p4c;
    return load_global(p);
  };
  // TAPS, at consume time (branch-free: every lane runs the selects): a shifted channel's float4 gets its row-end
  // correction -- the own float4 moved by |tshift| elements where the unaligned load would have left the row -- and the
  // lane's timestep mask
  auto tap_fix = [&](int kk, float4 v) -> float4 {
    const bool sh = min(4 * min(kk, K4 - 1) + kq, Ci - 1) < Cr;
    const float4 right = make_float4(0.f, sabs == 1 ? v.x : 0.f, sabs == 1 ? v.y : (sabs == 2 ? v.x : 0.f),
                                     sabs == 1 ? v.z : (sabs == 2 ? v.y : (sabs == 3 ? v.x : 0.f)));
    const float4 left = make_float4(sabs == 1 ? v.y : (sabs == 2 ? v.z : (sabs == 3 ? v.w : 0.f)),
                                    sabs == 1 ? v.z : (sabs == 2 ? v.w : 0.f), sabs == 1 ? v.w : 0.f, 0.f);
    const bool r1 = sh && sfix == 1, r2 = sh && sfix == 2;
    v.x = r1 ? right.x : (r2 ? left.x : v.x); v.y = r1 ? right.y : (r2 ? left.y : v.y);
    v.z = r1 ? right.z : (r2 ? left.z : v.z); v.w = r1 ? right.w : (r2 ? left.w : v.w);
    v.x = (sh && km.x == 0.f) ? 0.f : v.x; v.y = (sh && km.y == 0.f) ? 0.f : v.y;   // selects, not products: what is
    v.z = (sh && km.z == 0.f) ? 0.f : v.z; v.w = (sh && km.w == 0.f) ? 0.f : v.w;   // masked belongs to another row
    return v;
  };

  // ONEPASS (all output tiles fit the accumulators, the usual case): the first chunk of the stream is
  // requested BEFORE the matrix is staged, so the block's prologue (matrix -> LDS, barrier) runs under
  // the HBM latency of those loads instead of ahead of it.  With several passes the ring is (re)loaded
  // at the top of each pass -- never inside a branch, which would cost the counted vmcnt waits.
  float4 ring[kKC];
  if (ONEPASS) {
#pragma unroll
    for (int i = 0; i < kKC; ++i) ring[i] = loadB(i);
  }

  // matrix -> LDS: 4 independent, unconditional (clamped) loads per trip, padding zeroed by select
  const int total = Mrows * Kpad;
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * kBlock) {
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, total - 1);
      const int co = i / Kpad, k = i - co * Kpad;
      const int coc = min(co, Co - 1), kc = min(k, Ci - 1);
      // TAPS, row-major taps [R, 2 Co, Cr] = [W0; W1]: M'(co, k) = W_{k / Cr}[co, k % Cr]; in-major (the backward's
      // transposed pass over [R, 2 Co', Cr'] with Co' = Cr here) is the plain in-major indexing of that same array
      const float m = m_in_major ? M[((size_t)r * Ci + kc) * Co + coc]
                      : (TAPS ? M[((size_t)r * 2 * Co + (size_t)(kc / Cr) * Co + coc) * Cr + kc % Cr]
                              : M[((size_t)r * Co + coc) * Ci + kc]);
      const float a = has_extra ? addvec[r * Co + coc] : 0.f;
      w[u] = (co < Co) ? ((k < Ci) ? m : ((k == Ci) ? a : 0.f)) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * kBlock < total) Wl[i0 + u * kBlock] = w[u];
  }
  if (DO_Q)
    for (int i = threadIdx.x; i < 4 * K4; i += kBlock) ql[i] = (i < Ci) ? qvec[r * Ci + i] : 0.f;
  for (int i = threadIdx.x; i < Mrows; i += kBlock)
    bl[i] = (epi.bias != nullptr && i < Co) ? epi.bias[(size_t)r * epi.bias_rstride + i] : 0.f;
  __syncthreads();
  const float relu_floor = epi.relu ? 0.f : -3.4e38f;   // kernel-uniform: max(v, floor) is the ReLU or nothing

  float4 qa = f4zero();
  for (int m0 = 0; m0 < Mt; m0 += MG) {
    f32x4 acc[MG][4];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mg][i] = zero4();
    const float* wrow = Wl + (m0 * 16 + j) * Kpad + kq;  // A fragment: row co = tile*16 + j, column 4k + kq
    if (!ONEPASS) {
#pragma unroll
      for (int i = 0; i < kKC; ++i) ring[i] = loadB(i);
    }
    auto step = [&](int kk, const float4& braw) {
      const float4 b = TAPS ? tap_fix(kk, braw) : braw;
#pragma unroll
      for (int mg = 0; mg < MG; ++mg) {
        const float a = wrow[mg * 16 * Kpad + 4 * kk];
        acc[mg][0] = mfma16(a, b.x, acc[mg][0]);
        acc[mg][1] = mfma16(a, b.y, acc[mg][1]);
        acc[mg][2] = mfma16(a, b.z, acc[mg][2]);
        acc[mg][3] = mfma16(a, b.w, acc[mg][3]);
      }
      if (DO_Q && m0 == 0) f4fma(ql[4 * kk + kq], b, qa);
    };
    // whole chunks: one straight-line basic block per trip, so hipcc emits counted vmcnt waits
    int k0 = 0;
    for (; k0 + kKC <= K4; k0 += kKC) {
#pragma unroll
      for (int i = 0; i < kKC; ++i) {
        step(k0 + i, ring[i]);
        ring[i] = loadB(k0 + i + kKC);
      }
    }
    // tail (< kKC k-steps, already in the ring): wave-uniform branches, no loads inside
#pragma unroll
    for (int i = 0; i < kKC; ++i)
      if (k0 + i < K4) step(k0 + i, ring[i]);
    // D tile i: column = lane & 15 <-> position 4*p4 + i;  row = 4*(lane >> 4) + reg.
    // Epilogue = bias (LDS) + add operand + ReLU + store.  The add operand's 4 float4 of tile mg + 1 are requested
    // before tile mg is finished and stored, every load unconditional (clamped row and position): a load inside the
    // `co < Co` branch made hipcc emit load, s_waitcnt vmcnt(0), store -- 2 x 20 dependent round trips per wave.
    float4 addv[2][4];
    auto load_add = [&](int mg, float4 (&dst)[4]) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int coc = min((m0 + mg) * 16 + 4 * kq + reg, Co - 1);
        const float* arow = SEGS ? rowtab[Ci + Co + coc] : epi.add.template row<false>(g, coc, 4 * P4);
        dst[reg] = load_global(arow + 4 * (size_t)p4c);
      }
    };
    if (HAS_ADD) load_add(0, addv[0]);
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      if (HAS_ADD && mg + 1 < MG) load_add(mg + 1, addv[(mg + 1) & 1]);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = (m0 + mg) * 16 + 4 * kq + reg;
        const float b = bl[co];
        float4 v = make_float4(acc[mg][0][reg] + b, acc[mg][1][reg] + b, acc[mg][2][reg] + b, acc[mg][3][reg] + b);
        if (HAS_ADD) {
          const float4 a = addv[mg & 1][reg];
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
        if (co < Co && pvalid) {
          const float* orow = SEGS ? rowtab[Ci + min(co, Co - 1)] : out.template row<false>(g, co, 4 * P4);
          store_global(orow + 4 * (size_t)p4, v);
        }
      }
    }
  }
  if (DO_Q) {  // the four lane quarters hold the 4k + kq channels' share of q
    qa.x += __shfl_xor(qa.x, 16); qa.y += __shfl_xor(qa.y, 16); qa.z += __shfl_xor(qa.z, 16); qa.w += __shfl_xor(qa.w, 16);
    qa.x += __shfl_xor(qa.x, 32); qa.y += __shfl_xor(qa.y, 32); qa.z += __shfl_xor(qa.z, 32); qa.w += __shfl_xor(qa.w, 32);
    if (kq == 0 && pvalid) q4[(size_t)g * P4 + p4] = qa;
  }
}

static int proj_passes_mg(int Co, int* mg_out) {
  const int Mt = cdiv(Co, 16);
  // all output tiles in one pass when they fit the accumulator registers: up to 7 x 16 channels (188 registers, two
  // waves per SIMD -- forced by the kernel's launch bounds, left alone the allocator took more than 256 and one wave).
  // A 72 -> 98 channel mixing in ONE pass over its input: 288 -> 252 us against two passes of 4 and 3 tiles.
  // Several passes: at most 6 tiles each -- the ring is re-loaded per pass there, and with 7 tiles that form needs
  // 2-4 registers more than the 256 a wave has (it spilled).
  const int passes = Mt <= kProjMaxMG ? 1 : cdiv(Mt, kProjMaxMG - 1);
  *mg_out = cdiv(Mt, passes);
  return passes;
}

size_t project_mfma_lds_bytes(int Ci, int Co, bool has_extra) {  // Ci, Co: totals over the segments
  const int Kx = Ci + (has_extra ? 1 : 0);
  int MG;
  const int passes = proj_passes_mg(Co, &MG);
  // matrix + q vector + bias per output row (+ 1 float of alignment) + the row-pointer table of the segmented form
  return (size_t)(passes * MG * 16 * proj_kpad(Kx) + 4 * ((Kx + 3) / 4) + passes * MG * 16 + 1) * sizeof(float) +
         (size_t)(Ci + 2 * Co) * sizeof(float*);
}

template <int MG>
static int launch_project_mg(const SegList& in, const float* M, int m_in_major, const float* qvec,
                             const float* addvec, const float* extra, const SegList& out, float* q, int G, int Bg,
                             int P4, const MixEpilogue& epi, hipStream_t s) {
  const int Ci = in.total(), Co = out.total();
  const size_t lds = project_mfma_lds_bytes(Ci, Co, addvec != nullptr);
  dim3 grid(cdiv(P4, 64), G);
  const bool one = cdiv(Co, 16) <= MG;
  const bool segs = in.n > 1 || out.n > 1 || epi.add.n > 1;
  const bool has_add = epi.add.n > 0;
  // (matrices beyond 64 KiB of LDS -- 96 -> 130 channels, the merged mixing of msgat96 -- need the kernel's dynamic-LDS
  // ceiling raised once per device; up to kProjLdsMax two blocks still share a CU)
#define MSGAT_PROJ(Q, ONE, SG, AD)                                                                                      \
  do {                                                                                                                  \
    static LdsGrant granted;                                                                                            \
    if (int st_ = grant_dynamic_lds(&k_project_mfma<MG, Q, ONE, SG, AD>, lds, granted)) return st_;                      \
    hipLaunchKernelGGL((k_project_mfma<MG, Q, ONE, SG, AD>), grid, dim3(kBlock), lds, s, in, M, m_in_major, qvec,        \
                       addvec, (const float4*)extra, out, (float4*)q, Bg, P4, epi);                                     \
  } while (0)
#define MSGAT_PROJ2(Q, ONE)                                                          \
  do {                                                                               \
    if (segs) { if (has_add) MSGAT_PROJ(Q, ONE, true, true); else MSGAT_PROJ(Q, ONE, true, false); }   \
    else { if (has_add) MSGAT_PROJ(Q, ONE, false, true); else MSGAT_PROJ(Q, ONE, false, false); }      \
  } while (0)
  if constexpr (MG == kProjMaxMG) {   // proj_passes_mg() hands out 7 tiles only when one pass covers the output
    if (!one) return MSGAT_ERR_UNSUPPORTED;
    if (qvec != nullptr) MSGAT_PROJ2(true, true); else MSGAT_PROJ2(false, true);
  } else {
    if (qvec != nullptr) { if (one) MSGAT_PROJ2(true, true); else MSGAT_PROJ2(true, false); }
    else { if (one) MSGAT_PROJ2(false, true); else MSGAT_PROJ2(false, false); }
  }
#undef MSGAT_PROJ2
#undef MSGAT_PROJ
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_project_mfma(const SegList& in, const float* M, int m_in_major, const float* qvec,
                        const float* addvec, const float* extra, const SegList& out, float* q, int G, int Bg,
                        int P, const MixEpilogue& epi, hipStream_t s) {
  const int P4 = P / 4;
  int MG;
  proj_passes_mg(out.total(), &MG);
  switch (MG) {
    case 1: return launch_project_mg<1>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
    case 2: return launch_project_mg<2>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
    case 3: return launch_project_mg<3>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
    case 4: return launch_project_mg<4>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
    case 5: return launch_project_mg<5>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
    case 6: return launch_project_mg<6>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);
    default: return launch_project_mg<7>(in, M, m_in_major, qvec, addvec, extra, out, q, G, Bg, P4, epi, s);

  }
}

#ifdef MSGAT_LAB
static int lab_env(const char* name, int dflt) {   // lab builds only: a launch parameter from the environment
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#endif

// ---- the same convolution with ONE load per input float4 (Cr % 4 == 0: the widths of the reference's models) ---------
// k_project_mfma<.., TAPS> loads every input row twice (the plain and the shifted operand): 12 loads per lane for 24
// channels, half of them unaligned -- it ran at 2.9 TB/s of its algorithmic bytes.  Here a wave covers GP4 = the largest
// multiple of T/4 that fits 16 lanes (15 float4 = 5 whole rows at T = 12, the 16th lane idle), so every row of T lies
// inside ONE 16-lane group and the shifted operand is a lane shuffle of the loaded values: float4 index f of a row holds
// timesteps 4f .. 4f+3, and in[t + s] for s = 4a + b comes from the float4s a and a + 1 lanes away (zero where the row
// ends).  No unaligned load, no second load, nothing that could leave the tensor.
template <int MG, int kMaxK>   // kMaxK: k-steps the registers hold (8: Cr <= 32, 16: Cr <= 64)
__global__ __launch_bounds__(kBlock, 2) void k_causal_conv(
    const float* __restrict__ in, int in_gstride, const float* __restrict__ taps, int m_in_major,
    const float* __restrict__ bias, int bias_rstride, float* __restrict__ out, int Bg, int P4, int Cr, int Co,
    int tshift, int T, int tiles, int per_block
#ifdef MSGAT_LAB
    , int lab   // MSGAT_LAB_CC: 1 = no lane shuffles (shifted operand = plain), 2 = only the plain half's MFMAs, 4 = no stores
#endif
    ) {
  extern __shared__ float lds[];
  const int Ci = 2 * Cr;                      // virtual channels [shifted | plain]
  const int K4r = Cr >> 2;                    // k-steps over the REAL channels (Cr % 4 == 0, host-checked)
  const int Kpad = proj_kpad(Ci);
  constexpr int Mrows = MG * 16;
  float* Wl = lds;                            // [Mrows][Kpad]
  float* bl = lds + Mrows * Kpad;             // [Mrows]
  const int g = blockIdx.y, r = g / Bg;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int F = T >> 2;                       // float4s per row of T
  const int GP4 = (16 / F) * F;               // float4s a wave covers: whole rows only
  const float* gbase = in + (size_t)g * in_gstride * (4 * (size_t)P4);
  // a block works on `per_block` consecutive tiles (4 waves x GP4 float4s each); the next tile's input is requested before
  // the current one is multiplied, so only the first tile's round trip and one matrix staging are exposed per block (one
  // tile per block, 4320 blocks of 46 KB each at PEMSD7 size, ran at 2.9 TB/s)
  // the second register set fits beside the accumulators up to here (3 x 16 and 4 x 16 spill: those take one tile per
  // block, host-side per_block = 1, and a loop the compiler sees through)
  constexpr bool kPrefetch = kMaxK == 8 || MG <= 2;
  const int tile0 = blockIdx.x * (kPrefetch ? per_block : 1), tile1 = kPrefetch ? min(tile0 + per_block, tiles) : tile0 + 1;

  float4 own[kMaxK], nxt[kPrefetch ? kMaxK : 1];
  auto fetch = [&](int tile, auto& dst) {
    const int p4c = min((tile * 4 + wave) * GP4 + j, P4 - 1);
    const float* base = gbase + 4 * (size_t)p4c;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k)
      if (k < K4r) dst[k] = load_global(base + (size_t)(4 * k + kq) * (4 * (size_t)P4));   // kernel-uniform guard
  };
  fetch(tile0, own);                          // requested before the matrix is staged

  for (int i = threadIdx.x; i < Mrows * Kpad; i += kBlock) {
    const int co = i / Kpad, k = i - co * Kpad;
    const int coc = min(co, Co - 1), kc = min(k, Ci - 1);
    const float m = m_in_major ? taps[((size_t)r * Ci + kc) * Co + coc]
                               : taps[((size_t)r * 2 * Co + (size_t)(kc / Cr) * Co + coc) * Cr + kc % Cr];
    Wl[i] = (co < Co && k < Ci) ? m : 0.f;
  }
  for (int i = threadIdx.x; i < Mrows; i += kBlock)
    bl[i] = (bias != nullptr && i < Co) ? bias[(size_t)r * bias_rstride + i] : 0.f;
  __syncthreads();

  // lane constants of the shift: s = +-(4 a + b); source float4s a and a + 1 lanes away in the direction of the shift
  const int f = j % F;
  const int sabs = tshift < 0 ? -tshift : tshift, dir = tshift < 0 ? -1 : 1;
  const int a = sabs >> 2, b = sabs & 3;
  const bool vA = dir < 0 ? (f - a >= 0) : (f + a < F);
  const bool vB = dir < 0 ? (f - a - 1 >= 0) : (f + a + 1 < F);
  const int laneA = lane + dir * a, laneB = lane + dir * (a + 1);     // same kq group whenever the source is valid
  auto shifted = [&](const float4& v) -> float4 {
    float4 A = make_float4(__shfl(v.x, laneA), __shfl(v.y, laneA), __shfl(v.z, laneA), __shfl(v.w, laneA));
    float4 B = make_float4(__shfl(v.x, laneB), __shfl(v.y, laneB), __shfl(v.z, laneB), __shfl(v.w, laneB));
    A = vA ? A : f4zero();
    B = vB ? B : f4zero();
    if (dir < 0) {   // element e <- in[t0 + e - (4a + b)]: e >= b from A[e - b], else B[e - b + 4]
      return make_float4(b == 0 ? A.x : (b == 1 ? B.w : (b == 2 ? B.z : B.y)),
                         b == 0 ? A.y : (b == 1 ? A.x : (b == 2 ? B.w : B.z)),
                         b == 0 ? A.z : (b == 1 ? A.y : (b == 2 ? A.x : B.w)),
                         b == 0 ? A.w : (b == 1 ? A.z : (b == 2 ? A.y : A.x)));
    }
    // element e <- in[t0 + e + 4a + b]: e + b < 4 from A[e + b], else B[e + b - 4]
    return make_float4(b == 0 ? A.x : (b == 1 ? A.y : (b == 2 ? A.z : A.w)),
                       b == 0 ? A.y : (b == 1 ? A.z : (b == 2 ? A.w : B.x)),
                       b == 0 ? A.z : (b == 1 ? A.w : (b == 2 ? B.x : B.y)),
                       b == 0 ? A.w : (b == 1 ? B.x : (b == 2 ? B.y : B.z)));
  };

  const float* wrow = Wl + j * Kpad + kq;     // row co = tile*16 + j; column 4k + kq (shifted half), Cr + 4k + kq (plain half)
  for (int tile = tile0; tile < tile1; ++tile) {
    if constexpr (kPrefetch) {
      if (tile + 1 < tile1) fetch(tile + 1, nxt);   // block-uniform
    }
    f32x4 acc[MG][4];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mg][i] = zero4();
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      if (k < K4r) {                            // kernel-uniform; MFMAs, shuffles and LDS reads only
        const float4 pl = own[k];
#ifdef MSGAT_LAB
        const float4 sh = (lab & 1) ? pl : shifted(pl);
#else
        const float4 sh = shifted(pl);
#endif
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
          const float a0 = wrow[mg * 16 * Kpad + 4 * k];
          const float a1 = wrow[mg * 16 * Kpad + Cr + 4 * k];
#ifdef MSGAT_LAB
          if (!(lab & 2)) {
#endif
          acc[mg][0] = mfma16(a0, sh.x, acc[mg][0]); acc[mg][1] = mfma16(a0, sh.y, acc[mg][1]);
          acc[mg][2] = mfma16(a0, sh.z, acc[mg][2]); acc[mg][3] = mfma16(a0, sh.w, acc[mg][3]);
#ifdef MSGAT_LAB
          }
#endif
          acc[mg][0] = mfma16(a1, pl.x, acc[mg][0]); acc[mg][1] = mfma16(a1, pl.y, acc[mg][1]);
          acc[mg][2] = mfma16(a1, pl.z, acc[mg][2]); acc[mg][3] = mfma16(a1, pl.w, acc[mg][3]);
        }
      }
    }
    const int p4 = (tile * 4 + wave) * GP4 + j;
    const bool pvalid = j < GP4 && p4 < P4;
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = mg * 16 + 4 * kq + reg;
        const float bb = bl[co];
        const float4 v = make_float4(acc[mg][0][reg] + bb, acc[mg][1][reg] + bb, acc[mg][2][reg] + bb, acc[mg][3][reg] + bb);
#ifdef MSGAT_LAB
        if ((lab & 4) && v.x != 12345.678f) continue;
#endif
        if (co < Co && pvalid) store_global(out + ((size_t)g * Co + co) * (4 * (size_t)P4) + 4 * (size_t)p4, v);
      }
    if constexpr (kPrefetch) {
#pragma unroll
      for (int k = 0; k < kMaxK; ++k) own[k] = nxt[k];
    }
  }
}

// The causal dilated convolution in one pass (k_project_mfma<.., TAPS>): in [G,Cr,P] (a channel slice of a wider
// tensor when in_gstride > Cr), taps [R, 2 Co', Cr'] row-major; m_in_major = 0: forward (Co' = Co, Cr' = Cr, tshift =
// -dilation); m_in_major = 1: its input gradient (Co' = Cr, Cr' = Co: the same array read in-major, tshift = +dilation).
bool project_taps_supported(int Cr, int Co) {
  int MG;
  return Cr > 0 && Co >= 8 && proj_passes_mg(Co, &MG) == 1 && project_mfma_lds_bytes(2 * Cr, Co, false) <= (size_t)kProjLdsMax;
}

int launch_project_taps(const float* in, int in_gstride, const float* taps, int m_in_major, const float* bias,
                        int bias_rstride, float* out, int G, int Bg, int Cr, int Co, int P, int T, int tshift, hipStream_t s) {
  if (!project_taps_supported(Cr, Co) || T % 4 != 0 || P % T != 0 || tshift < -T || tshift > T) return MSGAT_ERR_UNSUPPORTED;
  SegList sin = seg_single(in, Cr);
  if (in_gstride > Cr) sin.gstride[0] = in_gstride;
  const SegList sout = seg_single(out, Co);
  MixEpilogue epi;
  epi.bias = bias;
  epi.bias_rstride = bias_rstride;
  const int P4 = P / 4;
  int MG;
  proj_passes_mg(Co, &MG);
  const size_t lds = project_mfma_lds_bytes(2 * Cr, Co, false);
#ifndef MSGAT_TAPS_TWO_LOADS
  if (Cr % 4 == 0 && Cr <= 64 && MG <= 4) {   // one load per input float4, the shifted operand by lane shuffles
    const int F = T / 4, GP4 = (16 / F) * F;
    const int tiles = cdiv(P4, 4 * GP4);
    // consecutive tiles per block (lab builds: MSGAT_LAB_CCPB).  Cold operands, G = 96, 24 -> 24 channels, tools/
    // causal_conv_time.py: N = 883 (4320 tiles) 98.6 / 85.8 / 83.3 / 92.1 us at 1 / 2 / 3 / 4; N = 307 (1536 tiles) 38.4 /
    // 35.0 / 40.8 / 39.5
    int per_block = std::max(1, std::min(3, (int)((long long)tiles * G / 768)));
    if (Cr > 32 && MG > 2) per_block = 1;      // no room for the prefetched tile's registers (kPrefetch)
#ifdef MSGAT_LAB
    per_block = lab_env("MSGAT_LAB_CCPB", per_block);
#endif
    const dim3 grid1(cdiv(tiles, per_block), G);
    const int gs = in_gstride > Cr ? in_gstride : Cr;
#ifdef MSGAT_LAB
#define MSGAT_CC_LAB , lab_env("MSGAT_LAB_CC", 0)
#else
#define MSGAT_CC_LAB
#endif
#define MSGAT_CC(mg, kk)                                                                                               \
  {                                                                                                                   \
    static LdsGrant granted;                                                                                          \
    if (int st_ = grant_dynamic_lds(&k_causal_conv<mg, kk>, lds, granted)) return st_;                                 \
    hipLaunchKernelGGL((k_causal_conv<mg, kk>), grid1, dim3(kBlock), lds, s, in, gs, taps, m_in_major, bias,           \
                       bias_rstride, out, Bg, P4, Cr, Co, tshift, T, tiles, per_block MSGAT_CC_LAB);                  \
  }
#define MSGAT_CC2(mg) case mg: if (Cr <= 32) MSGAT_CC(mg, 8) else MSGAT_CC(mg, 16) break;
    switch (MG) { MSGAT_CC2(1) MSGAT_CC2(2) MSGAT_CC2(3) MSGAT_CC2(4) }
#undef MSGAT_CC2
#undef MSGAT_CC
    MSGAT_CHECK_LAUNCH();
    return MSGAT_OK;
  }
#endif
  const dim3 grid(cdiv(P4, 64), G);
#define MSGAT_TAPS(mg)                                                                                                   \
  case mg: {                                                                                                            \
    static LdsGrant granted;                                                                                            \
    if (int st_ = grant_dynamic_lds(&k_project_mfma<mg, false, true, false, false, true>, lds, granted)) return st_;     \
  }                                                                                                                     \
    hipLaunchKernelGGL((k_project_mfma<mg, false, true, false, false, true>), grid, dim3(kBlock), lds, s, sin, taps,     \
                       m_in_major, (const float*)nullptr, (const float*)nullptr, (const float4*)nullptr, sout,           \
                       (float4*)nullptr, Bg, P4, epi, tshift, T);                                                       \
    break;
  switch (MG) {
    MSGAT_TAPS(1) MSGAT_TAPS(2) MSGAT_TAPS(3) MSGAT_TAPS(4) MSGAT_TAPS(5) MSGAT_TAPS(6) MSGAT_TAPS(7)
    default: return MSGAT_ERR_UNSUPPORTED;
  }
#undef MSGAT_TAPS
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---------------------------------------------------------------------------------------------
// channel-pair contraction over positions
// ---------------------------------------------------------------------------------------------
// part[a,c] = sum_{g in r} sum_p A[g,a,p] B[g,c,p]: positions are the MFMA's K axis, channels ride
// on the lanes, so an operand fragment is "one word per channel row" -- the worst possible global
// access.  A block (8 waves, one block per CU) therefore stages 256-position tiles of all its
// Ca + Cb rows through LDS:
//   - global side: one wave-instruction reads 1 KiB contiguous of ONE row.  A row of the [N,T]
//     slab starts at an arbitrary 16-B offset (N*T*4 is not a multiple of 128), so every piece
//     straddles one extra 128-B line: 1-KiB pieces cost 9 lines per 8, 64-B pieces 2 per 1
//     (measured with 64-B pieces: 2.5 TB/s).  Every byte of A and B is read once.
//   - LDS side: rows are padded by 16 B so the 16 rows x 4 words of a fragment read
//     (ds_read_b32, row = lane & 15, word = lane >> 4 of a 16-B chunk) are 2-way banked;
//   - wave w multiplies positions [32w, 32w+32) of the tile: 8 k-steps of 4 positions.
// Persistent split-K: the tiles of a relation (its Bg groups back to back) form one stream and each
// block owns a contiguous run of it, so the load pipeline (two tiles in flight per wave, in two
// register sets) never drains between groups and there is ONE reduction per block: the 8 waves'
// accumulators are summed in a fixed order into the block's partial (no atomics).  In-kernel
// stamps showed the alternative -- one block per (group, 1024 positions) -- spending 45% of a
// block in its exposed prologue, first-tile wait and reduction.

// lds_barrier() (common.hpp): the workgroup barrier for LDS hand-offs that leaves global loads in flight.

constexpr int kCpWaves = 8;
constexpr int kCpBlock = 64 * kCpWaves;
constexpr int kTile = 32 * kCpWaves;   // positions per staged tile (default): 32 (8 k-steps) per wave

// TWO = false: one register set in flight instead of two -- what the [64 x 80] block (MA = 4) has registers for.  It
// exists for operands with 49..64 A channels per z-block: a 98-channel gradient against 73 channels (the merged
// channel mixing of a MEAM block) takes 2 z-blocks instead of 3, i.e. reads B twice instead of three times.
// TILE = 128: half-length tiles (512-B row pieces, two rows per wave-instruction, 4 k-steps per wave).  Half the LDS
// per row, so [80 x 80] and [112 x 80] channel blocks fit: a 72- or 98-channel gradient against 73 channels is ONE
// pass over both operands instead of two z-blocks that each re-read B.
// SHIFT: the A operand's Cr real rows appear as 2 Cr virtual rows -- row a < Cr is row a read `ashift` timesteps LATER
// (A[a, n, t + ashift], zero where t + ashift >= T), row a >= Cr is row a - Cr as it stands: [dout[t+d]; dout], the
// gradient at the two taps of a causal dilated convolution (msgat.py:69-74), without a pass that writes it out.  The
// shifted float4 is an unaligned load of the same row of T (a float4 never straddles rows: T % 4 == 0); the last
// float4 of a slab, where that load would leave the tensor, is its own float4 moved in registers.
struct TimeShift {
  int d = 0;   // 0: no virtual rows
  int T = 4;
};

template <int MA, int NB, bool TWO = true, int TILE = kTile, bool SHIFT = false>
__global__ __launch_bounds__(kCpBlock) void k_chanpair_mfma(
    SegList A, const float* __restrict__ B, float* __restrict__ part, int Cb, int P, int Bg, int nzb, int b_ones,
    int nza, int nblk, int R, TimeShift ts) {
  // b_ones: B's last channel (index Cb-1) is a virtual row of ones, so part[a, Cb-1] = sum_p A[a,p] -- the bias
  // gradient of a 1x1 convolution comes out of the contraction that computes its weight gradient
  const int Cbr = Cb - b_ones;  // channels B really has
  const int Cr = A.total();     // rows A really has
  const int Ca = SHIFT ? 2 * Cr : Cr;
  extern __shared__ float4 lds4[];
  constexpr int kLPR = TILE / 4;        // lanes per row piece (64: one row per wave-instruction)
  constexpr int kRPI = 64 / kLPR;        // rows per wave-instruction
  constexpr int kRowF4 = TILE / 4 + 1;  // float4s per LDS row (piece + 16 B pad)
  constexpr int kPPW = TILE / kCpWaves;  // positions per wave and tile: 4 per k-step
  constexpr int RPW = ((MA + NB) * 16 + kCpWaves * kRPI - 1) / (kCpWaves * kRPI);  // load instructions per wave per tile
  // Block -> (relation r, run bx of the relation's tile stream, z-block zb of the channel matrix).  A channel matrix
  // larger than one [MA*16 x NB*16] block is cut into nza x nzb z-blocks that each stream their own A rows and B
  // columns -- the z-blocks of one A split all re-read the SAME B tiles.  With several z-blocks the grid is
  // one-dimensional and XCD-aware: workgroups are dealt round-robin over the 8 XCDs (block b on XCD b % 8), so the
  // z-blocks of one run take consecutive slots of ONE XCD, start together, do equal work (A rows split evenly: 49 + 49
  // of 98, not 64 + 34) and stay in step -- the second reader of a B tile finds it in that XCD's L2.  (As a 3-D grid
  // with z slowest, all z = 0 blocks ran first and z = 1 re-read B from HBM 170 us later.)
  const int nz = nza * nzb;
  int bx, r, zb;
  if (nz > 1) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    zb = slot % nz;
    const int k = (slot / nz) * 8 + xcd;
    if (k >= nblk * R) return;  // the grid is padded to a multiple of 8 runs
    r = k / nblk;
    bx = k - r * nblk;
  } else {
    bx = blockIdx.x;
    r = blockIdx.y;
    zb = 0;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int a_per = cdiv(Ca, nza);  // A rows per z-block (<= MA * 16, host-checked)
  const int a0 = (zb / nzb) * a_per;
  const int c0 = (zb % nzb) * (NB * 16);
  const int ca = min(a_per, Ca - a0), cb = min(NB * 16, Cb - c0);
  const int rows = ca + cb;  // rows [0,ca) = A channels, [ca,rows) = B channels
  constexpr int kZeroRow = (MA + NB) * 16;  // an all-zero row for absent channels
  constexpr int kOnesRow = kZeroRow + 1;    // an all-ones row for the virtual channel
  // this block's run of the relation's tile stream
  const int tpg = cdiv(P, TILE);  // tiles per group (the last one is partial)
  const long long ntot = (long long)Bg * tpg;
  const int t0 = (int)(ntot * bx / nblk), t1 = (int)(ntot * (bx + 1) / nblk);
  const int ntile = t1 - t0;

  MSGAT_STAMP(0);
  if (threadIdx.x < kRowF4) {
    lds4[kZeroRow * kRowF4 + threadIdx.x] = f4zero();
    lds4[kOnesRow * kRowF4 + threadIdx.x] = make_float4(1.f, 1.f, 1.f, 1.f);
  }

  // staging plan: instruction k of this wave covers rows (wave + kCpWaves*k)*kRPI + (lane / kLPR); a
  // lane fetches float4 (lane % kLPR) of the tile.  Rows past the last one alias row 0 (always a
  // valid address: no load sits in a branch, see k_project_mfma) and land in scratch rows nobody reads.
  const int lrow = lane / kLPR, lcol = lane % kLPR;
  const float* src[RPW];  // row pointer for group 0 of the relation
  int gstride[RPW];       // elements between consecutive groups of that row
  unsigned shifted = 0;   // SHIFT: bit k = instruction k of this lane stages a time-shifted row
  static_assert(RPW <= 32, "one flag bit per staging instruction");
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int rr = (wave + kCpWaves * k) * kRPI + lrow;
    const int row = rr < rows ? rr : 0;
    const size_t g0 = (size_t)r * Bg;
    const float* p;
    if (row < ca) {  // A channel a0 + row: its segment's tensor and group stride
      const int av = a0 + row;                       // virtual row
      const int a = SHIFT ? (av < Cr ? av : av - Cr) : av;
      if (SHIFT && av < Cr) shifted |= 1u << k;
      int sk = 0;
#pragma unroll
      for (int i = 1; i < kMaxSeg; ++i) sk += (i < A.n && a >= A.begin[i]) ? 1 : 0;
      p = A.row((int)g0, a, P);
      int gs = A.gstride[0];
#pragma unroll
      for (int i = 1; i < kMaxSeg; ++i) gs = (i == sk) ? A.gstride[i] : gs;
      gstride[k] = gs * P;
    } else {
      p = B + (g0 * Cbr + min(c0 + row - ca, Cbr - 1)) * P;  // the virtual row aliases a real one; nobody reads its copy
      gstride[k] = Cbr * P;
    }
    src[k] = p + 4 * lcol;
  }
  const int plast = P - 4 - 4 * lcol;  // clamp so that the float4 stays inside the row (P % 4 == 0)
  auto fetch = [&](int t, float4 (&regs)[RPW], int& pos) {  // t relative to t0, clamped to the run
    const int tau = t0 + min(t, ntile - 1);
    const int b = tau / tpg;
    const int p0 = (tau - b * tpg) * TILE;
    const float keep = (p0 + 4 * lcol < P) ? 1.f : 0.f;
    const int poff = min(p0, plast);
    pos = poff + 4 * lcol;                                   // SHIFT: the lane's first position in the slab, for stash()
    const int soff = (SHIFT && pos + ts.d <= P - 4) ? ts.d : 0;   // the slab's last float4 stays put (moved in stash())
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      // tail positions are zeroed by a multiply, not a select: hipcc sinks a load that only feeds a
      // select into a branch, and a load inside a branch costs the counted vmcnt waits (the
      // clamped address re-reads finite in-row data, so x * 0 is exact)
      const float* gp = src[k] + (size_t)b * gstride[k] + poff;
      float4 v;
      if (SHIFT) v = load_global_a4(gp + (((shifted >> k) & 1u) ? soff : 0));
      else v = *reinterpret_cast<const float4*>(gp);
      regs[k] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
    }
  };
  auto stash = [&](const float4 (&regs)[RPW], int pos) {
    // SHIFT (when the values are consumed, not in front of the loads): element e of a shifted row's float4 is timestep
    // tq + e + d of its row of T and exists iff that is < T; at the slab's last float4 the own values move left by d
    const int tq = SHIFT ? pos % ts.T : 0;
    const bool atend = SHIFT && pos + ts.d > P - 4;
    const int d = ts.d;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int rr = (wave + kCpWaves * k) * kRPI + lrow;
      float4 v = regs[k];
      if (SHIFT) {
        const bool sh = (shifted >> k) & 1u;
        const float4 left = make_float4(d == 1 ? v.y : (d == 2 ? v.z : (d == 3 ? v.w : 0.f)),
                                        d == 1 ? v.z : (d == 2 ? v.w : 0.f), d == 1 ? v.w : 0.f, 0.f);
        const bool mv = sh && atend;
        v.x = mv ? left.x : v.x; v.y = mv ? left.y : v.y; v.z = mv ? left.z : v.z; v.w = mv ? left.w : v.w;
        v.x = (sh && tq + 0 + d >= ts.T) ? 0.f : v.x; v.y = (sh && tq + 1 + d >= ts.T) ? 0.f : v.y;
        v.z = (sh && tq + 2 + d >= ts.T) ? 0.f : v.z; v.w = (sh && tq + 3 + d >= ts.T) ? 0.f : v.w;
      }
      if (rr < kZeroRow) lds4[rr * kRowF4 + lcol] = v;
    }
  };

  // fragment words of this lane: word kq of chunk (8*wave + qq) of its row
  const float* ldsw = reinterpret_cast<const float*>(lds4);
  int aw[MA], bw[NB];
#pragma unroll
  for (int ma = 0; ma < MA; ++ma) {
    const int row = (ma * 16 + j < ca) ? ma * 16 + j : kZeroRow;
    aw[ma] = row * (kRowF4 * 4) + kPPW * wave + kq;
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cl = nb * 16 + j;
    const int row = (cl < cb) ? ((b_ones && c0 + cl == Cbr) ? kOnesRow : ca + cl) : kZeroRow;
    bw[nb] = row * (kRowF4 * 4) + kPPW * wave + kq;
  }

  f32x4 acc[MA][NB];
#pragma unroll
  for (int ma = 0; ma < MA; ++ma)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = zero4();

  auto multiply = [&]() {
#pragma unroll
    for (int qq = 0; qq < kPPW / 4; ++qq) {
      float av[MA], bv[NB];
#pragma unroll
      for (int ma = 0; ma < MA; ++ma) av[ma] = ldsw[aw[ma] + 4 * qq];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bv[nb] = ldsw[bw[nb] + 4 * qq];
#pragma unroll
      for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = mfma16(av[ma], bv[nb], acc[ma][nb]);
    }
  };
  // two tiles in flight per wave (two register sets): while tile t is multiplied the loads of
  // t+1 and t+2 are outstanding.  All fetches are unconditional (clamped tile index) to keep
  // hipcc's vmcnt waits counted.
  if (TWO) {
    if (ntile > 0) {
      float4 ra[RPW], rb[RPW];
      int pa, pb;
      fetch(0, ra, pa);
      fetch(1, rb, pb);
      MSGAT_STAMP(1);
      for (int t = 0; t < ntile; t += 2) {
        lds_barrier();  // every wave is done reading the previous tile
        stash(ra, pa);
        lds_barrier();
        if (t == 0) MSGAT_STAMP(2);
        fetch(t + 2, ra, pa);
        multiply();
        if (t == 0) MSGAT_STAMP(3);
        lds_barrier();
        stash(rb, pb);
        lds_barrier();
        fetch(t + 3, rb, pb);
        if (t + 1 < ntile) multiply();  // wave-uniform; LDS reads and MFMAs only
      }
    }
  } else if (ntile > 0) {
    float4 ra[RPW];
    int pa;
    fetch(0, ra, pa);
    for (int t = 0; t < ntile; ++t) {
      lds_barrier();
      stash(ra, pa);
      lds_barrier();
      fetch(t + 1, ra, pa);  // clamped to the run: the last trip re-reads its own tile
      multiply();
    }
  }

  // sum the 8 waves' accumulators in a fixed order, kRedTiles 16x16 tiles at a time (what the tile buffer holds:
  // 8 waves x 1 KiB per tile): element e = (tile * 4 + reg) * 64 + lane
  MSGAT_STAMP(4);
  constexpr int kTiles = MA * NB;
  constexpr int kBufTiles = (((MA + NB) * 16 + 2) * kRowF4 * 16) / (kCpWaves * 1024);  // tiles the staging buffer holds
  constexpr int kRedTiles = kBufTiles >= kTiles ? kTiles : kBufTiles;
  static_assert(kRedTiles >= 1, "tile buffer too small for the reduction");
  float* red = reinterpret_cast<float*>(lds4) + (size_t)wave * (kRedTiles * 256);
  const float* all = reinterpret_cast<const float*>(lds4);
  float* out = part + ((size_t)r * nblk + bx) * ((size_t)Ca * Cb);
#pragma unroll
  for (int t0r = 0; t0r < kTiles; t0r += kRedTiles) {
    __syncthreads();  // the tile buffer (or the previous pass) is no longer read
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int tile = ma * NB + nb;
        if (tile >= t0r && tile < t0r + kRedTiles) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) red[((tile - t0r) * 4 + reg) * 64 + lane] = acc[ma][nb][reg];
        }
      }
    __syncthreads();
    const int ntl = (kTiles - t0r < kRedTiles) ? kTiles - t0r : kRedTiles;
    for (int e = threadIdx.x; e < ntl * 256; e += kCpBlock) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kCpWaves; ++w) v += all[w * (kRedTiles * 256) + e];
      const int el = e & 63, reg = (e >> 6) & 3, tile = t0r + (e >> 8);
      const int ma = tile / NB, nb = tile - ma * NB;
      const int a = a0 + ma * 16 + 4 * (el >> 4) + reg;  // D row = 4*(lane >> 4) + reg
      const int c = c0 + nb * 16 + (el & 15);            // D column = lane & 15
      if (a < a0 + ca && c < c0 + cb) out[(size_t)a * Cb + c] = v;   // this z-block's rows and columns only
    }
  }
  MSGAT_STAMP(5);
}

// ---- the same contraction with LDS-DMA staging ---------------------------------------------------------------------
// k_chanpair_mfma keeps a tile in flight in REGISTERS, and its wide channel blocks have registers for one tile only:
// the next fetch cannot be issued before the previous tile has been written to LDS, so each tile pays its whole load
// time (5-7 us under load) plus the write pass in series with nothing -- 269 us for the [98 x 73] contraction whose
// load side alone takes 160 us and whose multiply side alone 180 us (profiles/r03/contraction_lab.txt).  Here the
// tiles go from global memory straight into LDS (global_load_lds_dwordx4: no destination registers) into a ring of
// NBUF buffers of TILE positions: while tile t is multiplied, tiles t+1 .. t+NBUF-2 are landing, and the accumulators
// are all the registers the kernel needs -- a [112 x 80] channel block fits, so the 98-channel gradient is ONE pass.
//   - one wave-instruction moves 1 KiB = the 4*TILE-byte pieces of 1024 / (4 TILE) consecutive rows (a "row group").
//     The LDS image of an instruction is lane-linear (base in M0 + 16 B x lane), so the group's pieces are adjacent;
//     a 16-B pad follows each group.  Rows of a group would hit the same banks: the float4 slots of row r are
//     XOR-swizzled on the SOURCE side (lane l of the piece fetches float4 l ^ swz(r)), and the fragment reads apply
//     the same XOR -- conflict-free;
//   - order (NBUF >= 3): counted s_waitcnt vmcnt (own pieces of tile t landed, later tiles may still fly), barrier
//     (everyone's pieces landed, everyone is done with tile t-1), re-issue into tile t-1's buffer, multiply.  With
//     two buffers a second barrier separates the multiply from the re-issue.  Raw s_barrier with lgkmcnt(0) only:
//     __syncthreads() would drain vmcnt;
//   - positions past the end of a row cannot be zeroed on the way (no registers): their lanes re-read in-row data and
//     the A fragment of those k-steps is zeroed instead (partial tiles only: the last tile of a group);
//   - row groups past the block's last row are not staged: their instructions (kept, so that every wave's vmcnt
//     arithmetic is the same) fetch one 16-B word into a dump group.
constexpr int kGGroupF4 = 64 + 1;  // float4s per row group: 1 KiB + 16 B

// MIX: the pass also writes mix[g,c,p] = sum_a Mx[r,a,c] A[g,a,p] -- with A = [du | dq] and Mx = [W | alpha] that is
// dx = W^T du + alpha (x) dq of the PROJ_FIRST backward (msgat.py:27's autograd), computed from the A tile the
// contraction has in LDS anyway: du and dq are read once for dW, dalpha AND dx.  Wave w owns positions 16 w .. 16 w + 15
// of a tile for all channels: D[i = position][j = channel] = sum_a A[a][position] Mx[a][channel], Mx fragments held in
// registers for the whole run (positions as D's rows: a lane ends up with four consecutive positions of a channel, one
// 16-B store; with channels as rows it was four 4-B stores: the [98 x 73] pass 456 instead of 415 us in the step).  The stores count on vmcnt like the LDS-DMA loads (in issue order), so EVERY lane
// stores every time -- lanes without a valid (channel, position) into `dump` -- and the waits are counted over both.
struct ChanMix {
  const float* Mw = nullptr;     // [R, Ca - 1, Cb]
  const float* Mlast = nullptr;  // [R, Cb]: row Ca - 1 of the matrix
  float* out = nullptr;          // [G, Cb, P]
  float* dump = nullptr;         // >= 256 floats (16-B aligned) nobody reads
};

// With several z-blocks over B (nzb > 1; the mix forms never cut A: all its rows must be in LDS) each block writes the
// mix output of ITS columns c0 .. c0 + cb.
// MODE 0: the contraction.  MODE 1 (MIX): every wave also computes the mix output of its 16 positions.  MODE 2 (SPLIT, for
// the wide channel blocks whose accumulators leave no registers for the matrix fragments): waves 0-3 contract (16
// positions of a 64-position tile each), waves 4-7 compute the mix output from the same LDS tiles -- two roles with
// equal MFMA counts, one wave of each per SIMD; all eight stage.
template <int MA, int NB, int TILE, int NBUF, int MODE = 0>
__global__ __launch_bounds__(kCpBlock) void k_chanpair_glds(
    SegList A, const float* __restrict__ B, float* __restrict__ part, int Cb, int P, int Bg, int nzb, int b_ones,
    int nza, int nblk, int R, ChanMix mix) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  static_assert(TILE == 64 || TILE == 128, "row pieces of 256 or 512 bytes");
  constexpr bool MIX = MODE != 0, SPLIT = MODE == 2;
  constexpr int kRPI = 256 / TILE;                 // rows per wave-instruction (row group)
  constexpr int kLPR = TILE / 4;                   // lanes (float4s) per row piece
  constexpr int kCWaves = SPLIT ? kCpWaves / 2 : kCpWaves;   // waves that contract
  constexpr int kPPW = TILE / kCWaves;             // positions per (contracting) wave and tile
  constexpr int kMaxGroups = (MA + NB) * 16 / kRPI;
  constexpr int RPW = (kMaxGroups + kCpWaves - 1) / kCpWaves;  // LDS-DMA instructions per wave and tile
  const int Cbr = Cb - b_ones;
  const int Ca = A.total();
  extern __shared__ float4 lds4[];
  const int nz = nza * nzb;
  int bx, r, zb;
  if (nz > 1) {  // XCD-paired one-dimensional grid: see k_chanpair_mfma
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    zb = slot % nz;
    const int k = (slot / nz) * 8 + xcd;
    if (k >= nblk * R) return;
    r = k / nblk;
    bx = k - r * nblk;
  } else {
    bx = blockIdx.x;
    r = blockIdx.y;
    zb = 0;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int a_per = cdiv(Ca, nza);
  const int c_per = cdiv(Cb, nzb);      // even split (<= 16 NB: the host picked nzb = ceil(Cb / 16 NB))
  const int a0 = (zb / nzb) * a_per;
  const int c0 = (zb % nzb) * c_per;
  const int ca = min(a_per, Ca - a0), cb = min(c_per, Cb - c0);
  const int rows = ca + cb;
  // buffer = the groups of the block's largest z-block (the host sized LDS for that), a constant group (zero row,
  // ones row), a dump group
  const int ngroups = cdiv(min(a_per, Ca) + min(c_per, Cb), kRPI);
  const int bufF4 = (ngroups + 2) * kGGroupF4;
  const int zero_row = ngroups * kRPI, ones_row = zero_row + 1;
  const int tpg = cdiv(P, TILE);
  const long long ntot = (long long)Bg * tpg;
  const int t0 = (int)(ntot * bx / nblk), t1 = (int)(ntot * (bx + 1) / nblk);
  const int ntile = t1 - t0;
  float* out = part + ((size_t)r * nblk + bx) * ((size_t)Ca * Cb);
  if (ntile <= 0) {  // (never with the launcher's block counts) this block's partial is zero
    for (int e = threadIdx.x; e < ca * cb; e += kCpBlock) out[(size_t)(a0 + e / cb) * Cb + c0 + e % cb] = 0.f;
    return;
  }

  if (threadIdx.x < 2 * kLPR) {  // the constant rows, in every buffer
    const float4 v = (int)threadIdx.x < kLPR ? f4zero() : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int b = 0; b < NBUF; ++b) lds4[b * bufF4 + ngroups * kGGroupF4 + threadIdx.x] = v;
  }

  // staging plan: instruction k of this wave fills group wave + 8k; lane l fetches float4 ((l % kLPR) ^ swz(row)) of
  // row group * kRPI + l / kLPR
  auto swz = [](int row) { return (row % kRPI) * (16 / kRPI); };   // in float4 slots: row i of a group sits 64 i / kRPI banks further
  const int lrow = lane / kLPR, lcol = lane % kLPR;
  const float* src[RPW];
  int gstride[RPW], lcs[RPW], grp[RPW];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int g = wave + kCpWaves * k;
    const int rr = g * kRPI + lrow;
    const bool live = rr < rows;
    const int row = live ? rr : 0;
    const size_t g0 = (size_t)r * Bg;
    const float* p;
    if (row < ca) {
      const int a = a0 + row;
      int sk = 0;
#pragma unroll
      for (int i = 1; i < kMaxSeg; ++i) sk += (i < A.n && a >= A.begin[i]) ? 1 : 0;
      p = A.row((int)g0, a, P);
      int gs = A.gstride[0];
#pragma unroll
      for (int i = 1; i < kMaxSeg; ++i) gs = (i == sk) ? A.gstride[i] : gs;
      gstride[k] = gs * P;
    } else {
      p = B + (g0 * Cbr + min(c0 + row - ca, Cbr - 1)) * P;
      gstride[k] = Cbr * P;
    }
    src[k] = p;
    lcs[k] = live ? 4 * (lcol ^ swz(rr)) : -1;              // float offset inside the piece; -1: fetch one word only
    grp[k] = g * kRPI < rows ? g : ngroups + 1;             // wave-uniform: groups without a live row go to the dump
  }
  auto issue = [&](int t) {  // t relative to t0, clamped to the run; buffer t % NBUF
    const int tau = t0 + min(t, ntile - 1);
    const int b = tau / tpg;
    const int p0 = (tau - b * tpg) * TILE;
    float4* buf = lds4 + (t % NBUF) * bufF4;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int poff = lcs[k] < 0 ? 0 : min(p0 + lcs[k], P - 4);   // inside the row (P % 4 == 0); masked in multiply()
      const float* gp = src[k] + (size_t)b * gstride[k] + poff;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)gp, (lds_ptr_t)(buf + grp[k] * kGGroupF4), 16, 0, 0);
    }
  };

  // fragment words: row `row`, positions kPPW * wave + 4 qq + kq -> float4 slot ((kPPW / 4) * wave ^ swz) + qq
  const float* ldsw = reinterpret_cast<const float*>(lds4);
  const int rwave = SPLIT ? (wave & (kCWaves - 1)) : wave;   // index within the wave's role: its slice of the tile
  auto frag_word = [&](int row) {
    return (row / kRPI) * (kGGroupF4 * 4) + (row % kRPI) * TILE + 4 * (((kPPW / 4) * rwave) ^ swz(row)) + kq;
  };
  int aw[MA], bw[NB];
#pragma unroll
  for (int ma = 0; ma < MA; ++ma) aw[ma] = frag_word((ma * 16 + j < ca) ? ma * 16 + j : zero_row);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cl = nb * 16 + j;
    bw[nb] = frag_word((cl < cb) ? ((b_ones && c0 + cl == Cbr) ? ones_row : ca + cl) : zero_row);
  }

  // The two roles of SPLIT are the two arms of ONE if: the accumulators exist only in the first, the matrix fragments
  // only in the second -- as two independent conditions hipcc kept both sets alive everywhere (256 VGPRs + 142 spilled).
  f32x4 acc[MA][NB];
  const bool contracts = !SPLIT || wave < kCWaves;   // wave-uniform
  auto zero_acc = [&]() {
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = zero4();
  };

  // MIX: matrix fragments A2[i = channel 16 nb + j][k = a = 4 s + kq], and the words of the B2 operand
  // B2[k = a][j = position kPPW w + j] in the A rows of the tile
  constexpr int KS = MIX ? MA * 4 : 1;
  constexpr int kStores = MIX ? NB : 0;       // per lane and tile
  static_assert(!MIX || kPPW == 16, "one 16-position tile per wave");
  float mfrag[KS][NB];
  // B2 word of k-step s: row 4 s + kq of the tile = group (4 s + kq) / kRPI -- linear in s, so one register and an
  // immediate offset per read.  Rows a >= Ca of the last k-steps hold other operands' data (B = x): those lanes read the
  // tile's zero row instead, so a non-finite x cannot reach the mix output as 0 * Inf (the two-pass fallback never reads
  // x for it either).  The launchers only pick a block shape for Ca > kCaMin, so the earlier k-steps need no select.
  constexpr int kMStep = (4 / kRPI) * (kGGroupF4 * 4);
  constexpr int kCaMin = MA <= 2 ? (MA - 1) * 16 : (MA - 2) * 16;
  const int zword = ngroups * (kGGroupF4 * 4) + j;
  const int mbase = (kq / kRPI) * (kGGroupF4 * 4) + (kq % kRPI) * TILE + 4 * (((kPPW / 4) * rwave + (j >> 2)) ^ swz(kq)) + (j & 3);
  auto load_mfrag = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int a = 4 * s + kq;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int c = c0 + nb * 16 + j;     // this z-block's columns of the matrix
        const int cc = min(c, Cbr - 1);
        float v;
        if (mix.Mlast != nullptr) {   // kernel-uniform: the matrix's last row lives in another array
          const float wv = mix.Mw[((size_t)r * (Ca - 1) + max(min(a, Ca - 2), 0)) * Cbr + cc];
          const float lv = mix.Mlast[(size_t)r * Cbr + cc];
          v = a == Ca - 1 ? lv : wv;
        } else {
          v = mix.Mw[((size_t)r * Ca + min(a, Ca - 1)) * Cbr + cc];
        }
        mfrag[s][nb] = (nb * 16 + j < cb && c < Cbr && a < Ca) ? v : 0.f;
      }
    }
  };
  auto mix_tile = [&](int t) {
    const int tau = t0 + t;
    const int b = tau / tpg;
    const int p0 = (tau - b * tpg) * TILE;
    const float* w = ldsw + (t % NBUF) * (bufF4 * 4);
    f32x4 d[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) d[nb] = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      int word = mbase + s * kMStep;
      if (4 * s + 3 >= kCaMin) word = (4 * s + kq < Ca) ? word : zword;
      const float bv = w[word];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) d[nb] = mfma16(bv, mfrag[s][nb], d[nb]);
    }
    // D[i = position 4 kq + reg][j = channel]: a lane holds four consecutive positions of one channel -- one 16-B store
    const int pos = p0 + kPPW * rwave + 4 * kq;
    float* og = mix.out + ((size_t)r * Bg + b) * Cbr * P + pos;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int c = c0 + nb * 16 + j;
      float* dst = (nb * 16 + j < cb && c < Cbr && pos < P) ? og + (size_t)c * P : mix.dump + 4 * lane;   // a select on the address: no branch
      *reinterpret_cast<f32x4*>(dst) = d[nb];
    }
  };

  auto multiply = [&](int t) {
    const int tau = t0 + t;
    const int p0 = (tau - (tau / tpg) * tpg) * TILE;
    const bool partial = p0 + TILE > P;   // wave-uniform: the last tile of a group
    const float* w = ldsw + (t % NBUF) * (bufF4 * 4);
#pragma unroll
    for (int qq = 0; qq < kPPW / 4; ++qq) {
      float av[MA], bv[NB];
#pragma unroll
      for (int ma = 0; ma < MA; ++ma) av[ma] = w[aw[ma] + 4 * qq];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bv[nb] = w[bw[nb] + 4 * qq];
      if (partial) {
        const float keep = (p0 + kPPW * rwave + 4 * qq + kq < P) ? 1.f : 0.f;
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) av[ma] *= keep;
      }
#pragma unroll
      for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = mfma16(av[ma], bv[nb], acc[ma][nb]);
    }
  };

#pragma unroll
  for (int t = 0; t < NBUF - 1; ++t) issue(t);
  if (NBUF == 2) {
    zero_acc();
    issue(1);
    for (int t = 0; t < ntile; ++t) {
      wait_vmcnt<RPW>();   // tile t landed, tile t + 1 may be in flight
      lds_barrier();
      multiply(t);
      lds_barrier();       // the buffer is free
      issue(t + 2);
    }
  } else {
    static_assert(!MIX || NBUF == 3, "the counted waits of the MIX forms are written for three buffers");
    constexpr int kSteady = (NBUF - 2) * RPW + (NBUF - 1) * kStores;   // younger than tile t's loads: see below
    static_assert(kSteady < 64, "vmcnt is a 6-bit counter");
    // Per trip: this wave's pieces of tile t are in LDS (later tiles and -- waves that mix -- the stores of the last
    // NBUF - 1 trips, issued after tile t's loads, may be in flight; the first trips have fewer operations behind
    // them); barrier: ... and every other wave's, and nobody reads tile t-1's buffer any more; re-issue into that
    // buffer (clamped index: the last trips re-read the last tile); the wave's role(s).  Both roles pass the same
    // barriers.
    if (contracts) {
      zero_acc();
      if (MODE == 1) load_mfrag();
      for (int t = 0; t < ntile; ++t) {
        if (MODE == 1 && t == 0) wait_vmcnt<RPW>();
        else if (MODE == 1 && t == 1) wait_vmcnt<RPW + kStores>();
        else wait_vmcnt<MODE == 1 ? kSteady : (NBUF - 2) * RPW>();
        lds_barrier();
        issue(t + NBUF - 1);
        multiply(t);
        if (MODE == 1) mix_tile(t);
      }
    } else {   // SPLIT, waves that mix
      load_mfrag();
      for (int t = 0; t < ntile; ++t) {
        if (t == 0) wait_vmcnt<RPW>();
        else if (t == 1) wait_vmcnt<RPW + kStores>();
        else wait_vmcnt<kSteady>();
        lds_barrier();
        issue(t + NBUF - 1);
        mix_tile(t);
      }
    }
  }
  wait_vmcnt<0>();

  // sum the 8 waves' accumulators in a fixed order, kRedTiles 16x16 tiles at a time (the launcher checks that
  // kRedTiles * 8 KiB fit the staging buffers)
  constexpr int kTiles = MA * NB;
  constexpr int kRedTiles = kTiles < 8 ? kTiles : 8;
  float* red = reinterpret_cast<float*>(lds4) + (size_t)wave * (kRedTiles * 256);
  const float* all = reinterpret_cast<const float*>(lds4);
#pragma unroll
  for (int t0r = 0; t0r < kTiles; t0r += kRedTiles) {
    __syncthreads();
    if (contracts) {
#pragma unroll
      for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int tile = ma * NB + nb;
          if (tile >= t0r && tile < t0r + kRedTiles) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[((tile - t0r) * 4 + reg) * 64 + lane] = acc[ma][nb][reg];
          }
        }
    }
    __syncthreads();
    const int ntl = (kTiles - t0r < kRedTiles) ? kTiles - t0r : kRedTiles;
    for (int e = threadIdx.x; e < ntl * 256; e += kCpBlock) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kCWaves; ++w) v += all[w * (kRedTiles * 256) + e];
      const int el = e & 63, reg = (e >> 6) & 3, tile = t0r + (e >> 8);
      const int ma = tile / NB, nb = tile - ma * NB;
      const int a = a0 + ma * 16 + 4 * (el >> 4) + reg;
      const int c = c0 + nb * 16 + (el & 15);
      if (a < a0 + ca && c < c0 + cb) out[(size_t)a * Cb + c] = v;
    }
  }
}

// Blocks per relation.  The kernels are built for ONE resident block per CU, so the grid runs in rounds of `ncu` blocks and
// what counts is how full the last round is.  Few relations: ncu / R blocks each fill one round (R = 3: 255 of 256 CUs).
// Many relations -- a per-sample matrix makes every group its own relation: R = 96 at PEMSD7 size, 160 with five components
// -- leave ncu / R = 2 or 1 blocks each, i.e. 192 or 160 busy CUs of 256 (the merged convolution backward ran 415 us at
// R = 96 where the same bytes take 345 us at R = 3, and 796 us at R = 160): there, the smallest count up to 16 whose rounds
// are at least 95 % full (8 at R = 96 and R = 160: three and five full rounds), else the fullest: 415 -> 363 us, 796 -> 613.
int chanpair_mfma_blocks(int R) {
  const int ncu = device_cu_count();
  const int kmin = max(1, ncu / R);
  if (kmin >= 16) return kmin;
  int best = kmin;
  double best_fill = 0.0;
  for (int k = kmin; k <= 16; ++k) {
    const long long blocks = (long long)R * k;
    const double fill = (double)blocks / (double)(((blocks + ncu - 1) / ncu) * ncu);
    if (fill >= 0.95) return k;
    if (fill > best_fill + 1e-9) { best_fill = fill; best = k; }
  }
  return best;
}

static thread_local TimeShift g_time_shift;   // set by launch_chanpair_mfma for the duration of its dispatch cascade

template <int MA, int NB, bool TWO = true, int TILE = kTile>
static int launch_chanpair_t(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk_max,
                             int b_ones, hipStream_t s, int* nblk_used) {
  const TimeShift ts = g_time_shift;
  const int Ca = ts.d ? 2 * A.total() : A.total();
  const int nza = cdiv(Ca, MA * 16), nzb = cdiv(Cb, NB * 16);
  const int nz = nza * nzb;
  // several z-blocks: all of them resident at once (one block per CU), so fewer, longer runs per relation
  const int nblk = nz > 1 ? max(1, nblk_max / nz) : nblk_max;
  *nblk_used = nblk;
  // tile rows + the zero row + the ones row; the reduction re-uses the buffer a few tiles at a time
  const size_t lds = sizeof(float4) * (size_t)(((MA + NB) * 16 + 2) * (TILE / 4 + 1));
  if (g_form_probe) {
    snprintf(g_form_probe->name, sizeof g_form_probe->name, "k_chanpair_mfma<%d,%d,%s,%d%s>", MA, NB, TWO ? "true" : "false", TILE,
             ts.d ? ",shift" : "");
    g_form_probe->nza = nza; g_form_probe->nzb = nzb;
    return MSGAT_OK;
  }
  const dim3 grid = nz > 1 ? dim3((unsigned)cdiv(nblk * R, 8) * 8 * nz) : dim3(nblk, R, 1);
  if (ts.d) {
    static LdsGrant granted;
    if (int st = grant_dynamic_lds(&k_chanpair_mfma<MA, NB, TWO, TILE, true>, lds, granted)) return st;
    hipLaunchKernelGGL((k_chanpair_mfma<MA, NB, TWO, TILE, true>), grid, dim3(kCpBlock), lds, s, A, B, part, Cb, P, Bg, nzb,
                       b_ones, nza, nblk, R, ts);
  } else {
    static LdsGrant granted;
    if (int st = grant_dynamic_lds(&k_chanpair_mfma<MA, NB, TWO, TILE, false>, lds, granted)) return st;
    hipLaunchKernelGGL((k_chanpair_mfma<MA, NB, TWO, TILE, false>), grid, dim3(kCpBlock), lds, s, A, B, part, Cb, P, Bg, nzb,
                       b_ones, nza, nblk, R, ts);
  }
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// LDS bytes of the LDS-DMA form for a channel matrix cut into nza x nzb z-blocks
template <int MA, int NB, int TILE, int NBUF>
static size_t chanpair_glds_lds(int Ca, int Cb) {
  const int nza = cdiv(Ca, MA * 16), nzb = cdiv(Cb, NB * 16);
  const int rows = min(cdiv(Ca, nza), Ca) + min(cdiv(Cb, nzb), Cb);
  return sizeof(float4) * (size_t)NBUF * (cdiv(rows, 256 / TILE) + 2) * kGGroupF4;
}

template <int MA, int NB, int TILE, int NBUF, int MODE = 0>
static int launch_chanpair_glds_t(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P,
                                  int nblk_max, int b_ones, hipStream_t s, int* nblk_used, ChanMix mix = ChanMix()) {
  const int Ca = A.total();
  const int nza = cdiv(Ca, MA * 16), nzb = cdiv(Cb, NB * 16);
  const int nz = nza * nzb;
  int nblk = nz > 1 ? max(1, nblk_max / nz) : nblk_max;
#ifdef MSGAT_LAB
  nblk = min(nblk_max, max(1, lab_env("MSGAT_LAB_BPC", 1) * nblk_max / nz));   // resident blocks per CU
#endif
  *nblk_used = nblk;
  const size_t lds = chanpair_glds_lds<MA, NB, TILE, NBUF>(Ca, Cb);
  if (lds > (size_t)kLdsMax || lds < (size_t)(MA * NB < 8 ? MA * NB : 8) * kCpWaves * 1024) return MSGAT_ERR_UNSUPPORTED;
  if (g_form_probe) {
    snprintf(g_form_probe->name, sizeof g_form_probe->name, "k_chanpair_glds<%d,%d,%d,%d,%d>", MA, NB, TILE, NBUF, MODE);
    g_form_probe->nza = nza; g_form_probe->nzb = nzb;
    return MSGAT_OK;
  }
  {
    static LdsGrant granted;
    if (int st = grant_dynamic_lds(&k_chanpair_glds<MA, NB, TILE, NBUF, MODE>, lds, granted)) return st;
  }
  const dim3 grid = nz > 1 ? dim3((unsigned)cdiv(nblk * R, 8) * 8 * nz) : dim3(nblk, R, 1);
  hipLaunchKernelGGL((k_chanpair_glds<MA, NB, TILE, NBUF, MODE>), grid, dim3(kCpBlock), lds, s, A, B, part, Cb, P, Bg,
                     nzb, b_ones, nza, nblk, R, mix);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

// ---- which LDS-DMA form for which channel block -----------------------------------------------------------------
// X(MA, NB, TILE, MIXMODE): a [16 MA x 16 NB] channel block staged in three buffers of TILE positions; MIXMODE says how
// the one-pass form (contraction AND mix output) runs: 1 = every wave does both (128-position tiles), 2 = waves 0-3
// contract and waves 4-7 mix (64-position tiles: what three buffers of that many rows leave room for), 0 = there is
// none.  The list covers the channel counts of the three models of the reference's registry (main.py:17,
// msgat.py:220-229: 48 / 72 / 96 hidden channels, 16 / 24 / 32 per branch):
//                       GACN projection backward     merged channel mixing          residual convolution
//                       [Co + 1 x C]                 [4 Co + 2 x C + 1]             [C x C + 1]
//   msgat48             17 x 48   <2,3,128> mix 1    66 x 49   <5,4,64> mix 2       48 x 49   <3,4,128> mix 1
//   msgat72             25 x 72   <2,5,128> mix 1    98 x 73   <7,5,64> mix 2       72 x 73   <5,5,64>  mix 2
//   msgat96             33 x 96   <3,6,64>  mix 2    130 x 97  <9,4,64> mix 2       96 x 97   <6,4,64>  mix 2
//                                                    two z-blocks over B, 49 + 48 columns each with ALL rows of A
//                                                    (a [96 x 112] accumulator block spills; A is read twice, B once)
#define MSGAT_GLDS_FORMS(X) \
  X(2, 3, 128, 1) X(2, 4, 128, 1) X(2, 5, 128, 1) X(3, 4, 128, 1) X(3, 6, 64, 2) X(5, 4, 64, 2) X(5, 5, 64, 2) \
  X(7, 5, 64, 2) X(6, 4, 64, 2) X(9, 4, 64, 2) MSGAT_GLDS_LAB_FORMS(X)
// lab builds (-DMSGAT_LAB): narrow blocks over B -- several z-blocks per run that each stage all of A (re-read from the
// XCD's L2) and write FEW mix channels over the run's whole position range
#ifdef MSGAT_LAB
#define MSGAT_GLDS_LAB_FORMS(X) \
  X(2, 1, 128, 1) X(2, 2, 128, 1) X(7, 1, 64, 2) X(7, 2, 64, 2) X(7, 3, 64, 2) X(5, 1, 64, 2) X(5, 2, 64, 2) X(5, 3, 64, 2) \
  X(6, 5, 64, 2)   /* 96 x 73: what the merged gradient would cost without its two single-channel rows */
#else
#define MSGAT_GLDS_LAB_FORMS(X)
#endif

// Launches the form [MA x NB] if the list has it (and its buffers fit LDS); *handled = 0 and nothing launched otherwise.
static int launch_glds_form(int MA, int NB, bool with_mix, const SegList& A, const float* B, float* part, int R, int Bg,
                            int Cb, int P, int nblk, int b_ones, hipStream_t s, int* nblk_used, const ChanMix& mix,
                            int* handled) {
  const int Ca = A.total();
  *handled = 0;
#define MSGAT_GLDS_TRY(ma, nb, tile, mode)                                                                            \
  if (MA == ma && NB == nb) {                                                                                         \
    if ((with_mix && mode == 0) || chanpair_glds_lds<ma, nb, tile, 3>(Ca, Cb) > (size_t)kLdsMax) return MSGAT_OK;      \
    *handled = 1;                                                                                                     \
    if (with_mix) return launch_chanpair_glds_t<ma, nb, tile, 3, mode>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used, mix); \
    return launch_chanpair_glds_t<ma, nb, tile, 3, 0>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);          \
  }
  MSGAT_GLDS_FORMS(MSGAT_GLDS_TRY)
#undef MSGAT_GLDS_TRY
  return MSGAT_OK;
}

static bool glds_form_exists(int MA, int NB, bool with_mix) {
#define MSGAT_GLDS_HAS(ma, nb, tile, mode) \
  if (MA == ma && NB == nb) return !with_mix || mode != 0;
  MSGAT_GLDS_FORMS(MSGAT_GLDS_HAS)
#undef MSGAT_GLDS_HAS
  return false;
}

// rows of >= 512 positions in whole float4s: every group spans several tiles (shorter rows: the register-staged kernel)
static bool glds_rows_ok(int P) { return P % 4 == 0 && P >= 512; }

// The contraction AND mixout = M^T A in one pass (k_chanpair_glds, MODE 1 / 2): part[a, c] (c < Cb; with b_ones a virtual
// last channel of ones in B) and mixout[g, c, p] = sum_a M[r, a, c] A[g, a, p] over B's Cb - b_ones real channels, with
// M = [R, Ca, Cb - b_ones] -- or, Mlast given, [Mw | Mlast] with Mw = [R, Ca - 1, .] and Mlast = [R, .] its last row
// ([W | alpha] of the GACN projection).  All Ca rows must be in LDS at once (one z-block); a block one tile taller than
// the operand is fine (the kernels mask rows >= Ca and read the zero row for them).  *done = 0 (nothing launched) when
// no form covers the shape: the caller runs the two passes.
static int launch_glds_mix(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                           int b_ones, const float* Mw, const float* Mlast, float* mixout, hipStream_t s, int* nblk_used,
                           int* done) {
  const int Ca = A.total();
  *done = 0;
  if (!glds_rows_ok(P) || Ca <= 16) return MSGAT_OK;
  ChanMix mix;
  mix.Mw = Mw; mix.Mlast = Mlast; mix.out = mixout;
  mix.dump = part + (((size_t)R * nblk * Ca * Cb + 3) & ~(size_t)3);   // chanpair_partial_floats() leaves 260 floats behind the partials
  int nzb_lo = 1, nzb_hi = 2;
#ifdef MSGAT_LAB
  if (lab_env("MSGAT_LAB_NZB", 0) > 0) nzb_lo = nzb_hi = lab_env("MSGAT_LAB_NZB", 0);
#endif
  for (int nzb = nzb_lo; nzb <= nzb_hi; ++nzb) {           // z-blocks over B: each stages all of A
    // the narrowest block that covers the z-block's columns, or one tile wider (columns past the operand read the zero
    // row): the list is written for the widths WITH a bias column (49 / 73 / 97), and the same mixing without one
    // (48 / 72 / 96 columns: the merged channel mixing of the stacked schedule) must not fall back to two passes
    for (int NB = cdiv(cdiv(Cb, nzb), 16); NB <= cdiv(cdiv(Cb, nzb), 16) + 1; ++NB) {
      if (cdiv(Cb, NB * 16) != nzb) continue;      // the kernel derives nzb from the block width
      for (int MA = cdiv(Ca, 16); MA <= cdiv(Ca, 16) + 1; ++MA) {
        if (!glds_form_exists(MA, NB, true)) continue;
        const int st = launch_glds_form(MA, NB, true, A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used, mix, done);
        if (st || *done) return st;
      }
    }
  }
  return MSGAT_OK;
}

int launch_chanpair_mix(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                        const float* Mw, const float* Mlast, float* mixout, hipStream_t s, int* nblk_used, int* done) {
  return launch_glds_mix(A, B, part, R, Bg, Cb, P, nblk, 0, Mw, Mlast, mixout, s, nblk_used, done);
}

int launch_chanpair_mix_wide(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                             int b_ones, const float* M, float* mixout, hipStream_t s, int* nblk_used, int* done) {
  return launch_glds_mix(A, B, part, R, Bg, Cb, P, nblk, b_ones, M, nullptr, mixout, s, nblk_used, done);
}

// A with time-shifted virtual rows (k_chanpair_mfma<.., SHIFT>): part is [2 Cr x Cb] for A's Cr real rows
int launch_chanpair_shifted(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                            int b_ones, int dilation, int T, hipStream_t s, int* nblk_used) {
  if (dilation <= 0 || T % 4 != 0 || P % T != 0) return MSGAT_ERR_SHAPE;
  g_time_shift.d = dilation < T ? dilation : T;
  g_time_shift.T = T;
  const int st = launch_chanpair_mfma(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
  g_time_shift = TimeShift();
  return st;
}

int launch_chanpair_mfma(const SegList& A, const float* B, float* part, int R, int Bg, int Cb, int P, int nblk,
                         int b_ones, hipStream_t s, int* nblk_used) {
  const bool shift = g_time_shift.d != 0;     // the LDS-DMA forms cannot shift: register-staged kernel
  const int Ca = shift ? 2 * A.total() : A.total();
#ifndef MSGAT_NO_GLDS
  // LDS-DMA staging where the list above has a block for the shape: the fewest z-blocks over A (each re-reads B) whose
  // row count some form of this width covers, the smallest such form
  if (!shift && glds_rows_ok(P) && Ca > 16) {
    // fewest rows staged in total: nzb z-blocks over B each stage A, nza z-blocks over A each stage B
    int best_ma = 0, best_nb = 0;
    long best_cost = -1;
    for (int nzb = 1; nzb <= 2; ++nzb)
      for (int nza = 1; nza <= 3; ++nza) {
        const int NBg = cdiv(cdiv(Cb, nzb), 16);
        if (cdiv(Cb, NBg * 16) != nzb) continue;
        for (int ma = max(cdiv(cdiv(Ca, nza), 16), 2); ma <= 9; ++ma) {
          if (!glds_form_exists(ma, NBg, false) || cdiv(Ca, ma * 16) != nza) continue;
          const long cost = (long)nzb * Ca + (long)nza * Cb;
          if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_ma = ma; best_nb = NBg; }
          break;   // the smallest block of this width that covers the rows
        }
      }
    if (best_cost >= 0) {
      int handled = 0;
      const int st = launch_glds_form(best_ma, best_nb, false, A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used, ChanMix(), &handled);
      if (st || handled) return st;
    }
  }
#endif
  // the register-staged kernel.  Block shapes whose accumulators + two register sets in flight exceed the 256
  // registers of a 2-waves-per-SIMD block are not offered ([48 x 80], [48 x 96] and [32 x 96] spilled 18 / 93 / 14
  // registers): a wider B is cut into z-blocks, which re-read A
  const int MA = min(cdiv(Ca, 16), 3), NB = min(cdiv(Cb, 16), MA == 3 ? 4 : (MA == 2 ? 5 : 6));
  // 65..80 A channels against 17..80 B channels: half-length tiles hold all of A and B in LDS at once -- ONE pass over
  // both operands where the [48 x 96] blocks take two z-blocks that each re-read B (72 x 73, the residual tail's weight
  // gradient: 286 -> 174 us).  Not for wider A (a [112 x 80] block spills and ran at 372 us against 344 for the two
  // [64 x 80] blocks below) nor for a B of one tile (re-reading it is cheap: 64 -> 84 us).
  // (49..64 rows go to the [64 x 16 NB] block below instead: 256-position tiles, 1-KiB row pieces -- the shifted
  // [64 x 33] weight gradient of msgat96's convolutions 125 -> 110 us)
  if (NB >= 2 && NB <= 5 && Ca > 64 && Ca <= 80) {
    switch (NB) {
      case 2: return launch_chanpair_t<5, 2, false, 128>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      case 3: return launch_chanpair_t<5, 3, false, 128>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      case 4: return launch_chanpair_t<5, 4, false, 128>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      default: return launch_chanpair_t<5, 5, false, 128>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
    }
  }
  // 64 A channels per z-block where that saves a pass over B (and the [64 + 16 NB] rows fit LDS: NB <= 5)
  if (NB <= 5 && cdiv(Ca, 64) < cdiv(Ca, 48)) {
    switch (NB) {
      case 1: return launch_chanpair_t<4, 1, false>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      case 2: return launch_chanpair_t<4, 2, false>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      case 3: return launch_chanpair_t<4, 3, false>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      case 4: return launch_chanpair_t<4, 4, false>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
      default: return launch_chanpair_t<4, 5, false>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
    }
  }
#define MSGAT_CP(ma, nb) \
  if (MA == ma && NB == nb) return launch_chanpair_t<ma, nb>(A, B, part, R, Bg, Cb, P, nblk, b_ones, s, nblk_used);
  MSGAT_CP(1, 1) MSGAT_CP(1, 2) MSGAT_CP(1, 3) MSGAT_CP(1, 4) MSGAT_CP(1, 5) MSGAT_CP(1, 6)
  MSGAT_CP(2, 1) MSGAT_CP(2, 2) MSGAT_CP(2, 3) MSGAT_CP(2, 4) MSGAT_CP(2, 5)
  MSGAT_CP(3, 1) MSGAT_CP(3, 2) MSGAT_CP(3, 3) MSGAT_CP(3, 4)
#undef MSGAT_CP
  return MSGAT_ERR_UNSUPPORTED;
}

// Which kernel a [Ca x Cb] channel-pair contraction over rows of P positions takes (with_mix: the one-pass form that also
// writes the mix output; *one_pass = 0 there means "no fused form: the contraction below plus a projection pass").
// Runs the launchers' own selection code with the probe set: nothing is launched, no device is touched.
int contract_form_name(int Ca, int Cb, int b_ones, int P, int with_mix, char* buf, int buflen, int* one_pass, int* nza,
                       int* nzb) {
  FormProbe probe{};
  SegList A = seg_single(reinterpret_cast<const float*>(16), Ca);   // never dereferenced
  float* fake = reinterpret_cast<float*>(16);
  int nblk = 0, done = 0, st = MSGAT_OK;
  g_form_probe = &probe;
  if (with_mix) st = launch_glds_mix(A, fake, fake, 1, 1, Cb, P, 256, b_ones, fake, nullptr, fake, nullptr, &nblk, &done);
  if (!st && !done) st = launch_chanpair_mfma(A, fake, fake, 1, 1, Cb, P, 256, b_ones, nullptr, &nblk);
  g_form_probe = nullptr;
  if (st) return st;
  if (one_pass) *one_pass = done;
  if (nza) *nza = probe.nza;
  if (nzb) *nzb = probe.nzb;
  if (buf && buflen > 0) snprintf(buf, (size_t)buflen, "%s", probe.name);
  return MSGAT_OK;
}

}  // namespace msgat
