// The dense side of the attention on the bf16 matrix core, at fp32 accuracy (round 6).
//
// dense.hip runs both matrix products of a 16x16 tile as v_mfma_f32_16x16x4_f32.  That instruction issues at the fp32
// VECTOR rate and shares the issue port with the VALU (tools/mfma_rate.hip): 7 of them are 224 of a tile's ~330 clocks.
// v_mfma_f32_16x16x32_bf16 does 16x the multiply-adds per clock.  Here every fp32 operand is written as THREE bf16 terms,
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)          (exact: 8 + 8 + 8 significand bits)
// and a product a.b is the six partial products  ah bh, am bh, ah bm, am bm, ah bl, al bh  (what is dropped is below
// 2^-24 |a b|), stacked on the K axis of the bf16 MFMA and accumulated in fp32.  bf16 x bf16 is exact in fp32, so the only
// roundings are the accumulator's: tools/mfma_split_lab.hip measures 2.1e-8 mean / 2.7e-7 max error per unit of sum|a b|
// for K = 12, against 2.0e-8 / 2.0e-7 for the fp32 fmaf chain, and no dependence of any result bit on the tile position of
// an operand or on the side (A or B) it is on -- which is what lets backward re-create the forward's scores bit for bit.
//
// Score product (K = T = 12 -> 72 slots in three MFMAs of K = 32).  A "run" is 4 consecutive timesteps of one term
// (8 bytes); a lane's MFMA fragment is two runs.  Per instruction one PAIR of terms per side:
//     instr 0:  streamed side (h | m)  x  own side (h | h)        q_h k_h + q_m k_h
//     instr 1:  streamed side (h | m)  x  own side (m | m)        q_h k_m + q_m k_m
//     instr 2:  streamed side (h | l)  x  own side (l | h)        q_h k_l + q_l k_h
// a pair (X | Y) is the six runs X0 X1 X2 Y0 Y1 Y2 = three fragments, one per quad (lanes 16q .. 16q+15) 0..2; quad 3
// supplies zeros.  "Role A" is the (h|m),(h|l) side -- the columns' q in the forward, the own columns' q in the backward;
// "role B" the (h|h),(m|m),(l|h) side -- the own rows' kW in the forward, the streamed rows' kW in the backward: slot by
// slot both passes multiply the same two bf16 numbers.
//
// Payload product (K = 16 streamed items), fp16 x fp16 -> fp32.  P = 2^(S - m) comes out of the score tile in the B layout
// already (lane (own j, quad) holds items 4 quad + r) and must be split PER TILE on the VALU, which is what these passes
// are bound by (a bf16 three-term split costs 24 instructions per tile: 25 % of the pass, lab build).  P is bounded, so it
// goes to fp16 (11 significand bits per term) in TWO terms: four instructions per pair of values -- v_cvt_pk_f16_f32,
// two v_fma_mix_f32 (residual p - h with the half-precision operand read in place), v_cvt_pk_f16_f32.  P is carried times
// 2^7 (forward: P <= 2^8 under the deferred maximum) or 2^14 (backward: P <= 1) so that what fp16 drops (below 2^-25
// absolute) is below 2^-32 of the row sum.  The payload matrix (q and a ones row in the forward, delta kW in the
// backward) is two fp16 terms as well, times a power of two chosen per group from its largest entry (fp16 has no range
// to spare: the scale puts that entry at 2^13 .. 2^14 and the dropped part below 2^-38 of it):
//     instr 0:  payload (h | m)  x  P (h | m)  =  q_h P_h + q_m P_m        instr 1:  payload (h | m)  x  P (m | h)
// (one payload fragment from LDS serves both; the second P fragment is the first with its halves exchanged).
//
// The streamed side is prepared ONCE per group by k_dense_images (the split, and the 4-item transposition the payload
// operand needs) into "tile records" whose bytes are exactly the LDS image a block wants:
//     record = [S planes][P planes][backward: lse2 of the 16 rows], a plane = 16 x 16 bytes,
//     S planes role A (5):  [h0|h1] [h2|m0] [m1|m2] [h2|l0] [l1|l2]                      indexed [plane][item]
//     S planes role B (8):  [h0|h1] [h2|h0] [h1|h2] [m0|m1] [m2|m0] [m1|m2] [l0|l1] [l2|h0]
//     P planes (4):         [quad][row s]   8 fp16 = the term pair (h | m) for items 4 quad .. 4 quad + 3
// so a lane's fragment is one ds_read_b128 at slot (item or s) of a 256-byte plane: conflict-free for the four 16-lane
// groups of that instruction, and a chunk of records lands in LDS by LDS-DMA with no register or VALU in between.
#include "common.hpp"
#include "halfsplit.hpp"

namespace msgat {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_bf(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// The operand split.  EVERY score operand of both passes goes through this one function (image builder, own rows, own
// columns), so the two passes see the same bf16 numbers.  Terms are returned in fp32 format (low 16 bits zero).
__device__ __forceinline__ uint32_t bf16_hi(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = bf16_hi(x);
  const float r1 = x - __uint_as_float(h);
  m = bf16_hi(r1);
  const float r2 = r1 - __uint_as_float(m);
  l = bf16_hi(r2);
}
__device__ __forceinline__ uint32_t pack_hi(uint32_t lo, uint32_t hi) { return (lo >> 16) | hi; }   // both in fp32 format

constexpr int kBT = 12;                       // the form exists for T = 12 (every model of the reference's registry)
constexpr int kPlane = 256;                   // bytes: 16 items (or payload rows) x 16
constexpr int kRecSA = 5 * kPlane, kRecSB = 8 * kPlane, kRecP = 4 * kPlane;
constexpr int kRecA = kRecSA + kRecP;         // 2304 B: forward record (16 columns)
constexpr int kRecB = kRecSB + kRecP + kPlane; // 3328 B: backward record (16 rows) + their lse2 - 14 (64 B, padded to a plane)
constexpr int kChunkTiles = 4;                // records per staged chunk: 9 / 13 KiB, whole 1-KiB LDS-DMA pieces
constexpr int kBWaves = 8;
constexpr int kBBlock = 64 * kBWaves;
constexpr int kBRows = 16 * kBWaves;

// ---- the images ----------------------------------------------------------------------------------------------------
// Two small launches per dense pass.  k_dense_absmax: per-block maxima of |payload| (the power-of-two scale of the fp16
// payload comes from the largest entry of the GROUP).  k_dense_images: 256 threads = two tiles per block, one thread per
// unit of a tile -- units 0..47 = (item, 4 timesteps) of the score planes, 64..127 = (quad, payload row) of the payload
// planes; the records are assembled in LDS and leave as whole, coalesced 16-byte rows.
// BWD = false: values q (score planes role A; payload q with a ones row at s = T).  BWD = true: score planes of kW log2e
// (role B), payload delta[n] kW[n], lse2 - 14 of the rows (+inf past the end).  Items past N are zeros (forward: masked by
// the column count; backward: lse = +inf makes P = 0).  Block 0 of a group records (scale, 1 / scale).
// (One launch -- every block finding the group's maximum itself from the group's 42 KB -- was latency bound at 10-12 us for
// PEMSD7: three dependent phases in 384 blocks of 1024 threads.)
constexpr int kImgBlock = 256;
constexpr int kAbsF4 = 1024;                  // float4s of the source per block of k_dense_absmax
static inline int absmax_blocks(int N) { return cdiv(N * (kBT / 4), kAbsF4); }

template <bool BWD>
__global__ __launch_bounds__(kImgBlock) void k_dense_absmax(const float* __restrict__ src, const float* __restrict__ delta,
                                                            float* __restrict__ gmaxp, int N) {
  constexpr int T4 = kBT / 4;
  __shared__ float red[kImgBlock / 64];
  const int g = blockIdx.y;
  const float4* sg = reinterpret_cast<const float4*>(src + (size_t)g * N * kBT);
  const float* dg = BWD ? delta + (size_t)g * N : nullptr;
  const int n4 = N * T4;
  float mx = 0.f;
#pragma unroll
  for (int k = 0; k < kAbsF4 / kImgBlock; ++k) {
    const int i = blockIdx.x * kAbsF4 + k * kImgBlock + threadIdx.x;
    const int ic = min(i, n4 - 1);                      // clamped, unconditional: the four loads fly together
    const float4 v = sg[ic];
    float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (BWD) a *= fabsf(dg[ic / T4]);
    mx = fmaxf(mx, a);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kImgBlock / 64; ++w) mx = fmaxf(mx, red[w]);
    gmaxp[(size_t)g * gridDim.x + blockIdx.x] = mx;
  }
}
// the power of two that puts `absmax` at 2^13 .. 2^14 (1 for an all-zero group)
__device__ __forceinline__ float payload_scale(float absmax) { return absmax == 0.f ? 1.f : pow2i(payload_scale_exp(absmax)); }

template <bool BWD>
__global__ __launch_bounds__(kImgBlock) void k_dense_images(const float* __restrict__ src, const float* __restrict__ delta,
                                                            const float* __restrict__ lse, const float* __restrict__ gmaxp,
                                                            int nmax, uint4* __restrict__ img, float2* __restrict__ scales, int N,
                                                            int NTp) {
  constexpr int T = kBT;
  constexpr int kRec = BWD ? kRecB : kRecA;
  constexpr int kSOff = BWD ? kRecSB : kRecSA;
  __shared__ uint4 stage[2 * kRec / 16];
  const int g = blockIdx.y;
  const float* sg = src + (size_t)g * N * T;
  const float* dg = BWD ? delta + (size_t)g * N : nullptr;
  const int tile0 = blockIdx.x * 2;
  const int tile = tile0 + (threadIdx.x >> 7);
  const int u = threadIdx.x & 127;
  char* rec = reinterpret_cast<char*>(stage) + (threadIdx.x >> 7) * kRec;

  // the unit's source values first (they do not need the scale), then the group's maximum
  float4 v4 = f4zero();
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int j = 0, tq = 0, qd = 0, s = 0;
  if (u < 48) {
    j = u / 3; tq = u - 3 * j;
    const int n = tile * 16 + j;
    if (tile < NTp && n < N) v4 = reinterpret_cast<const float4*>(sg + (size_t)n * T)[tq];
  } else if (u >= 64) {
    qd = (u - 64) >> 4; s = (u - 64) & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = tile * 16 + 4 * qd + r;
      if (tile < NTp && n < N && s < T) {
        v[r] = sg[(size_t)n * T + s];
        if (BWD) v[r] *= dg[n];
      }
    }
  }
  float mx = 0.f;
  for (int i = 0; i < nmax; ++i) mx = fmaxf(mx, gmaxp[(size_t)g * nmax + i]);
  const float scale = payload_scale(mx);
  if (blockIdx.x == 0 && threadIdx.x == 0) scales[g] = make_float2(scale, 1.0f / scale);

  if (u < 48) {
    if (BWD) v4 = make_float4(v4.x * kLog2e, v4.y * kLog2e, v4.z * kLog2e, v4.w * kLog2e);
    uint32_t h[4], m[4], l[4];
    split3(v4.x, h[0], m[0], l[0]); split3(v4.y, h[1], m[1], l[1]); split3(v4.z, h[2], m[2], l[2]); split3(v4.w, h[3], m[3], l[3]);
    const uint2 H = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
    const uint2 M = make_uint2(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]));
    const uint2 L = make_uint2(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]));
    auto put = [&](int plane, int half, const uint2& r) { *reinterpret_cast<uint2*>(rec + plane * kPlane + j * 16 + half * 8) = r; };
    if (!BWD) {   // [h0|h1] [h2|m0] [m1|m2] [h2|l0] [l1|l2]
      if (tq == 0) { put(0, 0, H); put(1, 1, M); put(3, 1, L); }
      else if (tq == 1) { put(0, 1, H); put(2, 0, M); put(4, 0, L); }
      else { put(1, 0, H); put(3, 0, H); put(2, 1, M); put(4, 1, L); }
    } else {      // [h0|h1] [h2|h0] [h1|h2] [m0|m1] [m2|m0] [m1|m2] [l0|l1] [l2|h0]
      if (tq == 0) { put(0, 0, H); put(1, 1, H); put(7, 1, H); put(3, 0, M); put(4, 1, M); put(6, 0, L); }
      else if (tq == 1) { put(0, 1, H); put(2, 0, H); put(3, 1, M); put(5, 0, M); put(6, 1, L); }
      else { put(1, 0, H); put(2, 1, H); put(4, 0, M); put(5, 1, M); put(7, 0, L); }
    }
  } else if (u >= 64) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] *= scale;
      if (!BWD && s == T && tile * 16 + 4 * qd + r < N) v[r] = 1.f;   // the ones row (unscaled): carries the row sum of P
    }
    uint32_t H0, H1, M0, M1;
    split2_f16(v, 1.0f, H0, H1, M0, M1);
    *reinterpret_cast<uint4*>(rec + kSOff + qd * kPlane + s * 16) = make_uint4(H0, H1, M0, M1);   // (h | m)
  } else if (BWD) {   // u = 48 .. 63: the rows' lse2, less the exponent P is carried with
    const int jj = u - 48;
    const int n = tile * 16 + jj;
    reinterpret_cast<float*>(rec + kRecSB + kRecP)[jj] = (tile < NTp && n < N) ? lse[(size_t)g * N + n] - kPOffB : INFINITY;
  }
  __syncthreads();
  const int nrec = min(2, NTp - tile0);
  uint4* dst = img + ((size_t)g * NTp + tile0) * (kRec / 16);
  for (int i = threadIdx.x; i < nrec * (kRec / 16); i += kImgBlock) dst[i] = stage[i];
}

// ---- shared pieces of the two passes -------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One LDS-DMA piece: lane i's 16 bytes at gptr -> LDS byte (lds_base + 16 i), no registers.  As inline asm on purpose:
// hipcc tracks __builtin_amdgcn_global_load_lds as a write to LDS and puts `s_waitcnt vmcnt(0)` in front of the next
// ds_read whose memory operand it still knows -- the NEXT chunk's flight then ends before THIS chunk is multiplied
// (tools/mfma_split_lab.hip's model: 206 clocks per tile, the first build of these kernels: 256).  The kernels order the
// two themselves (wait_vmcnt + lds_barrier before a buffer is read, a barrier before it is refilled).  M0 is restored.
__device__ __forceinline__ void lds_dma16(const void* gptr, void* lds_base) {
  const unsigned la = (unsigned)(size_t)(lds_ptr_t)lds_base;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(la), "v"(gptr) : "memory");
}

// the role-A / role-B fragments of ONE node's T values for this lane's quad (own rows / own columns; registers)
struct FragA { uint4 hm, hl; };          // (h | m), (h | l)
struct FragB { uint4 hh, mm, lh; };      // (h | h), (m | m), (l | h)
__device__ __forceinline__ void runs_of(const float* v, uint2 H[3], uint2 M[3], uint2 L[3]) {
#pragma unroll
  for (int tq = 0; tq < 3; ++tq) {
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3(v[4 * tq + i], h[i], m[i], l[i]);
    H[tq] = make_uint2(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]));
    M[tq] = make_uint2(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]));
    L[tq] = make_uint2(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]));
  }
}
__device__ __forceinline__ uint4 two_runs(const uint2& a, const uint2& b) { return make_uint4(a.x, a.y, b.x, b.y); }
__device__ __forceinline__ FragA make_frag_a(const float* v, int quad) {
  uint2 H[3], M[3], L[3];
  runs_of(v, H, M, L);
  FragA f;
  const uint4 z = make_uint4(0, 0, 0, 0);
  f.hm = quad == 0 ? two_runs(H[0], H[1]) : quad == 1 ? two_runs(H[2], M[0]) : quad == 2 ? two_runs(M[1], M[2]) : z;
  f.hl = quad == 0 ? two_runs(H[0], H[1]) : quad == 1 ? two_runs(H[2], L[0]) : quad == 2 ? two_runs(L[1], L[2]) : z;
  return f;
}
__device__ __forceinline__ FragB make_frag_b(const float* v, int quad) {
  uint2 H[3], M[3], L[3];
  runs_of(v, H, M, L);
  FragB f;
  const uint4 z = make_uint4(0, 0, 0, 0);
  f.hh = quad == 0 ? two_runs(H[0], H[1]) : quad == 1 ? two_runs(H[2], H[0]) : quad == 2 ? two_runs(H[1], H[2]) : z;
  f.mm = quad == 0 ? two_runs(M[0], M[1]) : quad == 1 ? two_runs(M[2], M[0]) : quad == 2 ? two_runs(M[1], M[2]) : z;
  f.lh = quad == 0 ? two_runs(L[0], L[1]) : quad == 1 ? two_runs(L[2], H[0]) : quad == 2 ? two_runs(H[1], H[2]) : z;
  return f;
}

// Blocks of one group on one XCD (workgroups are dealt round-robin over the 8 XCDs, each with its own L2): the group's
// image -- N x 272 or 336 bytes -- is then fetched into ONE L2 instead of eight.  Grid = 8 * nb * ceil(G / 8) blocks.
__device__ __forceinline__ bool xcd_group_block(int nb, int G, int& g, int& b) {
  const int id = blockIdx.x;
  const int xcd = id & 7, k = id >> 3;
  g = xcd + 8 * (k / nb);
  b = k - (k / nb) * nb;
  return g < G;
}

constexpr float kBDefer = 8.f;

// ---- forward ---------------------------------------------------------------------------------------------------------
// Wave w owns rows n0 + 16w .. +15; all columns stream through LDS in chunks of kChunkTiles records, two LDS buffers,
// the next chunk in flight (LDS-DMA) while this one is multiplied; one barrier per chunk.
// Score tile D[i = column][j = row]: lane (row j, quad) holds columns 4 quad + r.  Payload tile D2[i = s][j = row].
template <bool WITH_PQ>
__global__ __launch_bounds__(kBBlock) void k_scores_b(
    const float* __restrict__ q, const float* __restrict__ Wg, const uint4* __restrict__ img,
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const int* __restrict__ erow, float* __restrict__ kW, float* __restrict__ lse, float* __restrict__ pq,
    float* __restrict__ E, const int* __restrict__ cpos, float* __restrict__ Ec, const float2* __restrict__ scales, float one,
    int G, int Bg, int N, int nnz, int NTp, int nb) {
  constexpr int T = kBT, T4 = T / 4;
  constexpr int kRecU = kRecA / 16;                    // uint4 per record
  constexpr int kChunkU = kChunkTiles * kRecU;         // 832
  constexpr int kPieces = kChunkU / 64;                // 13 LDS-DMA pieces of 1 KiB
  constexpr float kPOff = WITH_PQ ? kPOffF : 0.f;      // P is carried times 2^kPOff (see split_p)
  extern __shared__ uint4 ring[];                      // 2 x kChunkU
  __shared__ float kw2s[kBRows][T];

  int g, bx;
  if (!xcd_group_block(nb, G, g, bx)) return;
  const int r = g / Bg;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 15, quad = lane >> 4;
  const int n0 = bx * kBRows;
  const int n = n0 + 16 * wave + j;
  const bool valid = n < N;
  const uint4* imgg = img + (size_t)g * NTp * kRecU;
  const int nchunk = NTp / kChunkTiles;

  auto issue = [&](int c) {
    const uint4* src = imgg + (size_t)c * kChunkU;
    uint4* buf = ring + (c & 1) * kChunkU;
#pragma unroll
    for (int k = 0; k < (kPieces + kBWaves - 1) / kBWaves; ++k) {
      const int piece = wave + kBWaves * k;   // wave-uniform
      if (piece < kPieces)
        lds_dma16(src + piece * 64 + lane, buf + piece * 64);
    }
  };
  issue(0);

  // the rows' kW (stored unscaled for backward) and their role-B fragments of kW log2e
  const float* wg = Wg + (size_t)r * T * T;
  {
    float qr[T];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      float4 v = f4zero();
      if (valid) v = reinterpret_cast<const float4*>(q + ((size_t)g * N + n) * T)[t4];
      qr[4 * t4 + 0] = v.x; qr[4 * t4 + 1] = v.y; qr[4 * t4 + 2] = v.z; qr[4 * t4 + 3] = v.w;
    }
#pragma unroll
    for (int kk = 0; kk < T4; ++kk) {
      const int s = 4 * kk + quad;
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) a = fmaf(qr[t], wg[t * T + s], a);
      if (valid) kW[((size_t)g * N + n) * T + s] = a;
      kw2s[16 * wave + j][s] = a * kLog2e;
    }
  }
  __syncthreads();
  FragB fb;
  {
    float kr[T];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      const float4 v = reinterpret_cast<const float4*>(&kw2s[16 * wave + j][0])[t4];
      kr[4 * t4 + 0] = v.x; kr[4 * t4 + 1] = v.y; kr[4 * t4 + 2] = v.z; kr[4 * t4 + 3] = v.w;
    }
    fb = make_frag_b(kr, quad);
  }

  // per-lane offsets (uint4 units) inside a record
  const int offHM = (quad == 1 ? 1 : quad == 2 ? 2 : 0) * 16 + j;
  const int offHL = (quad == 1 ? 3 : quad == 2 ? 4 : 0) * 16 + j;
  const int offP = kRecSA / 16 + quad * 16 + j;

  float m = -3.0e38f, mo = m;   // running max of the row (finite floor, not -inf); mo = m - kPOff is what the exponent subtracts
  float lsum = 0.f;    // only without the payload product (inference): the row sum on the VALU
  f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da;

  for (int c = 0; c < nchunk; ++c) {
    wait_vmcnt<0>();
    lds_barrier();           // everyone's pieces of chunk c have landed; nobody reads chunk c-1's buffer any more
    if (c + 1 < nchunk) issue(c + 1);
    const uint4* buf = ring + (c & 1) * kChunkU;
#pragma unroll
    for (int tp = 0; tp < kChunkTiles; tp += 2) {
      const int m0 = (c * kChunkTiles + tp) * 16;   // first column of the trip
      if (m0 >= N) break;
      const uint4* ra = buf + tp * kRecU;
      const uint4* rb = ra + kRecU;
      const uint4 a_hm0 = ra[offHM], a_hl0 = ra[offHL], a_hm1 = rb[offHM], a_hl1 = rb[offHL];
      // (Starting the accumulator at -mo would give S - mo for free, four VALU instructions per tile less -- and costs accuracy:
      // the bf16 MFMA does not round a large C plus small products like an fmaf chain does; with scores ~1e2 dx was 3.3e-4
      // from float64 against 2e-5 with the subtraction on the VALU, tests/test_gpu_parity.py::test_large_scores_need_the_running_max.)
      f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
      S0 = mfma_bf(a_hm0, fb.hh, S0); S0 = mfma_bf(a_hm0, fb.mm, S0); S0 = mfma_bf(a_hl0, fb.lh, S0);
      S1 = mfma_bf(a_hm1, fb.hh, S1); S1 = mfma_bf(a_hm1, fb.mm, S1); S1 = mfma_bf(a_hl1, fb.lh, S1);
      const int mq = m0 + 4 * quad;
      float sv[8];
      if (m0 + 32 > N) {   // wave-uniform: the last trip may hold padding columns
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          sv[rr] = (mq + rr < N) ? S0[rr] : -3.0e38f;
          sv[4 + rr] = (mq + 16 + rr < N) ? S1[rr] : -3.0e38f;
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { sv[rr] = S0[rr]; sv[4 + rr] = S1[rr]; }
      }
      const float cm = fmaxf(fmaxf(fmaxf(fmaxf(sv[0], sv[1]), sv[2]), fmaxf(fmaxf(sv[3], sv[4]), sv[5])), fmaxf(sv[6], sv[7]));
      if (__any(cm > m + kBDefer)) {   // rare (deferred re-base); the row's 4 quads must agree on m
        float cx = fmaxf(cm, __shfl_xor(cm, 16));
        cx = fmaxf(cx, __shfl_xor(cx, 32));
        const float mn = fmaxf(m, cx);
        const float sc = fast_exp2(m - mn);  // m at its floor on the first tile -> 0
        m = mn;
        mo = mn - kPOff;
        lsum *= sc;
        if (WITH_PQ) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) { da[rr] *= sc; db[rr] *= sc; }
        }
      }
      float p[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) p[rr] = fast_exp2(sv[rr] - mo);
      if (!WITH_PQ) lsum += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      if (WITH_PQ) {
        const uint4 Fa = split_p(p, one), A2a = ra[offP];
        da = mfma_h(A2a, Fa, da); da = mfma_h(A2a, swap_halves(Fa), da);
        const uint4 Fb = split_p(p + 4, one), A2b = rb[offP];
        db = mfma_h(A2b, Fb, db); db = mfma_h(A2b, swap_halves(Fb), db);
      }
    }
  }

  if (WITH_PQ) {   // D2[s = T][row]: register 0 of the lanes with quad == 3
    lsum = __shfl(da[0] + db[0], j + 48);
  } else {
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
  }
  const float lse2 = mo + fast_log2(lsum);   // lsum carries 2^kPOff
  if (quad == 0 && valid) lse[(size_t)g * N + n] = lse2;
  if (WITH_PQ && valid && quad < T4) {
    const float inv = scales[g].y / lsum;     // the payload's scale and P's 2^kPOff (in lsum as well) both leave here
    reinterpret_cast<float4*>(pq + ((size_t)g * N + n) * T)[quad] =
        make_float4((da[0] + db[0]) * inv, (da[1] + db[1]) * inv, (da[2] + db[2]) * inv, (da[3] + db[3]) * inv);
  }

  // Edge coefficients of the wave's 16 rows.  The score of an edge must be THE tile's score (backward re-creates the tile and
  // cancels against the sparse term on saturated rows), so it is the same three MFMAs on the same fragments: 16 edges per
  // step on the A side (their columns' fragments gathered from the image in memory), the wave's rows on the B side, and edge
  // i's score is D[i][row of edge i] -- held by the lane of that row.
  const int nw = n0 + 16 * wave;
  const int e0 = rowptr[min(nw, N)], e1 = rowptr[min(nw + 16, N)];
  for (int et = e0; et < e1; et += 16) {
    const int ea = min(et + j, e1 - 1);
    const int ca = col[ea];
    const uint4* rec = imgg + (size_t)(ca >> 4) * kRecU;
    const int jj = ca & 15;
    const uint4 a_hm = rec[(quad == 1 ? 1 : quad == 2 ? 2 : 0) * 16 + jj];
    const uint4 a_hl = rec[(quad == 1 ? 3 : quad == 2 ? 4 : 0) * 16 + jj];
    int er[4], ee[4];
    float ev[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      ee[rr] = min(et + 4 * quad + rr, e1 - 1);
      er[rr] = erow[ee[rr]];
      ev[rr] = val[ee[rr]];
    }
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    S = mfma_bf(a_hm, fb.hh, S); S = mfma_bf(a_hm, fb.mm, S); S = mfma_bf(a_hl, fb.lh, S);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (et + 4 * quad + rr < e1 && er[rr] - nw == j) {
        const float x = fast_exp2(S[rr] - lse2) * ev[rr];
        E[(size_t)g * nnz + ee[rr]] = x;
        if (Ec != nullptr) Ec[(size_t)g * nnz + cpos[ee[rr]]] = x;
      }
    }
  }
}

// ---- backward: dense column pass ------------------------------------------------------------------------------------
// Wave w owns columns m0 + 16w .. +15 (B side: role-A fragments of their q); the rows stream through LDS as records of
// [kW log2e role B | delta kW payload | lse2].  Score tile D[i = row][j = column]; payload tile D2[i = s][j = column].
//   dq[m] += sum_{e into m} g_e kW[row_e]  -  sum_n 2^(S[n,m] - lse2[n]) delta[n] kW[n]
// the sparse in-edge term (as dense.hip's in_edge_term; T = 12)
__device__ __forceinline__ float4 in_edge_term_b(const int* __restrict__ colptr, const int* __restrict__ crow,
                                                 const int* __restrict__ cperm, const float* __restrict__ gEg,
                                                 const float* __restrict__ kWg, int mcol, int quad) {
  constexpr int T = kBT;
  const int c0 = colptr[mcol], c1 = colptr[mcol + 1];
  float4 sp = f4zero();
  for (int k = c0; k < c1; k += 4) {
    int ep[4], er[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = min(k + i, c1 - 1);
      ep[i] = cperm[kk];
      er[i] = crow[kk];
    }
    float ge[4];
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float raw = gEg[ep[i]];
      ge[i] = (k + i < c1) ? raw : 0.f;
      v[i] = reinterpret_cast<const float4*>(kWg + (size_t)er[i] * T)[quad];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) f4fma(ge[i], v[i], sp);
  }
  return sp;
}

__global__ __launch_bounds__(kBBlock) void k_bwd_dense_col_b(
    const float* __restrict__ q, const float* __restrict__ kW, const uint4* __restrict__ img,
    const float* __restrict__ gE, const int* __restrict__ colptr, const int* __restrict__ crow,
    const int* __restrict__ cperm, float* __restrict__ dq, const float2* __restrict__ scales, float one, int G, int N, int nnz,
    int NTp, int nb) {
  constexpr int T = kBT, T4 = T / 4;
  constexpr int kRecU = kRecB / 16;                    // 272
  constexpr int kChunkU = kChunkTiles * kRecU;         // 1088
  constexpr int kPieces = kChunkU / 64;                // 17
  extern __shared__ uint4 ring[];                      // 2 x kChunkU

  int g, bx;
  if (!xcd_group_block(nb, G, g, bx)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 15, quad = lane >> 4;
  const int mcol = bx * kBRows + 16 * wave + j;
  const bool valid = mcol < N;
  const uint4* imgg = img + (size_t)g * NTp * kRecU;
  const int nchunk = NTp / kChunkTiles;

  auto issue = [&](int c) {
    const uint4* src = imgg + (size_t)c * kChunkU;
    uint4* buf = ring + (c & 1) * kChunkU;
#pragma unroll
    for (int k = 0; k < (kPieces + kBWaves - 1) / kBWaves; ++k) {
      const int piece = wave + kBWaves * k;
      if (piece < kPieces)
        lds_dma16(src + piece * 64 + lane, buf + piece * 64);
    }
  };
  issue(0);

  FragA fa;
  {
    float qr[T];
#pragma unroll
    for (int t4 = 0; t4 < T4; ++t4) {
      float4 v = f4zero();
      if (valid) v = reinterpret_cast<const float4*>(q + ((size_t)g * N + mcol) * T)[t4];
      qr[4 * t4 + 0] = v.x; qr[4 * t4 + 1] = v.y; qr[4 * t4 + 2] = v.z; qr[4 * t4 + 3] = v.w;
    }
    fa = make_frag_a(qr, quad);
  }
  const int off0 = (quad == 1 ? 1 : quad == 2 ? 2 : 0) * 16 + j;   // (h | h)
  const int off1 = (quad == 1 ? 4 : quad == 2 ? 5 : 3) * 16 + j;   // (m | m)
  const int off2 = (quad == 1 ? 7 : quad == 2 ? 2 : 6) * 16 + j;   // (l | h)
  const int offP = kRecSB / 16 + quad * 16 + j;
  const int offL = (kRecSB + kRecP) / 16 + quad;                   // lse2 - 14 of rows 4 quad .. +3

  f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da;
  for (int c = 0; c < nchunk; ++c) {
    wait_vmcnt<0>();
    lds_barrier();
    if (c + 1 < nchunk) issue(c + 1);
    const uint4* buf = ring + (c & 1) * kChunkU;
#pragma unroll
    for (int tp = 0; tp < kChunkTiles; tp += 2) {
      if ((c * kChunkTiles + tp) * 16 >= N) break;
      const uint4* ra = buf + tp * kRecU;
      const uint4* rb = ra + kRecU;
      f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
      S0 = mfma_bf(ra[off0], fa.hm, S0); S0 = mfma_bf(ra[off1], fa.hm, S0); S0 = mfma_bf(ra[off2], fa.hl, S0);
      S1 = mfma_bf(rb[off0], fa.hm, S1); S1 = mfma_bf(rb[off1], fa.hm, S1); S1 = mfma_bf(rb[off2], fa.hl, S1);
      const float4 l4 = *reinterpret_cast<const float4*>(ra + offL), l5 = *reinterpret_cast<const float4*>(rb + offL);
      float p[8];
      p[0] = fast_exp2(S0[0] - l4.x); p[1] = fast_exp2(S0[1] - l4.y); p[2] = fast_exp2(S0[2] - l4.z); p[3] = fast_exp2(S0[3] - l4.w);
      p[4] = fast_exp2(S1[0] - l5.x); p[5] = fast_exp2(S1[1] - l5.y); p[6] = fast_exp2(S1[2] - l5.z); p[7] = fast_exp2(S1[3] - l5.w);
      const uint4 Fa = split_p(p, one), A2a = ra[offP];
      da = mfma_h(A2a, Fa, da); da = mfma_h(A2a, swap_halves(Fa), da);
      const uint4 Fb = split_p(p + 4, one), A2b = rb[offP];
      db = mfma_h(A2b, Fb, db); db = mfma_h(A2b, swap_halves(Fb), db);
    }
  }
  if (!valid || quad >= T4) return;
  const float unscale = scales[g].y * 6.103515625e-05f;   // 1 / (payload scale * 2^14)
  const float* kWg = kW + (size_t)g * N * T;
  const float4 sp = in_edge_term_b(colptr, crow, cperm, gE + (size_t)g * nnz, kWg, mcol, quad);
  float4* dst = reinterpret_cast<float4*>(dq + ((size_t)g * N + mcol) * T) + quad;
  float4 v = *dst;
  v.x += sp.x - (da[0] + db[0]) * unscale;
  v.y += sp.y - (da[1] + db[1]) * unscale;
  v.z += sp.z - (da[2] + db[2]) * unscale;
  v.w += sp.w - (da[3] + db[3]) * unscale;
  *dst = v;
}

// ---- host --------------------------------------------------------------------------------------------------------------
// scratch = [G][padded tiles] records (sized for the larger, backward record) + [G] (scale, 1 / scale) + [G][blocks] partial maxima
static inline int padded_tiles(int N) { return cdiv(cdiv(N, 16), kChunkTiles) * kChunkTiles; }
static inline size_t image_bytes(int G, int N) { return ((size_t)G * padded_tiles(N) * kRecB + 255) & ~(size_t)255; }
size_t dense_split_scratch_bytes(int G, int N, int T) {
  if (T != kBT) return 0;
  return image_bytes(G, N) + (((size_t)G * (8 + 4 * absmax_blocks(N)) + 255) & ~(size_t)255);
}

template <bool BWD>
static int launch_images(const float* src, const float* delta, const float* lse, void* scratch, int G, int N, hipStream_t s) {
  const int NTp = padded_tiles(N);
  uint4* img = reinterpret_cast<uint4*>(scratch);
  float2* scales = reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch) + image_bytes(G, N));
  float* gmaxp = reinterpret_cast<float*>(scales + G);
  const int nmax = absmax_blocks(N);
  hipLaunchKernelGGL(k_dense_absmax<BWD>, dim3(nmax, G), dim3(kImgBlock), 0, s, src, delta, gmaxp, N);
  MSGAT_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_dense_images<BWD>, dim3(cdiv(NTp, 2), G), dim3(kImgBlock), 0, s, src, delta, lse, gmaxp, nmax, img, scales, N,
                     NTp);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_scores_b(const msgat_graph_t& gr, const float* q, const float* Wg, float* kW, float* lse, float* pq, float* E,
                    float* Ec, int G, int Bg, int N, hipStream_t s, void* scratch) {
  const int st = launch_images<false>(q, nullptr, nullptr, scratch, G, N, s);
  if (st) return st;
  const int NTp = padded_tiles(N);
  const uint4* img = reinterpret_cast<const uint4*>(scratch);
  const float2* scales = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(scratch) + image_bytes(G, N));
  const int nb = cdiv(N, kBRows);
  const dim3 grid(8 * nb * cdiv(G, 8));
  const size_t lds = 2 * (size_t)kChunkTiles * kRecA;
  if (pq != nullptr)
    hipLaunchKernelGGL(k_scores_b<true>, grid, dim3(kBBlock), lds, s, q, Wg, img, gr.rowptr, gr.col, gr.val, gr.erow, kW, lse, pq,
                       E, gr.cpos, Ec, scales, 1.0f, G, Bg, N, gr.nnz, NTp, nb);
  else
    hipLaunchKernelGGL(k_scores_b<false>, grid, dim3(kBBlock), lds, s, q, Wg, img, gr.rowptr, gr.col, gr.val, gr.erow, kW, lse,
                       pq, E, gr.cpos, Ec, scales, 1.0f, G, Bg, N, gr.nnz, NTp, nb);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

int launch_bwd_dense_col_b(const msgat_graph_t& gr, const float* q, const float* kW, const float* lse, const float* delta,
                           const float* gE, float* dq, int G, int N, hipStream_t s, void* scratch) {
  const int st = launch_images<true>(kW, delta, lse, scratch, G, N, s);
  if (st) return st;
  const int NTp = padded_tiles(N);
  const uint4* img = reinterpret_cast<const uint4*>(scratch);
  const float2* scales = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(scratch) + image_bytes(G, N));
  const int nb = cdiv(N, kBRows);
  const dim3 grid(8 * nb * cdiv(G, 8));
  const size_t lds = 2 * (size_t)kChunkTiles * kRecB;
  hipLaunchKernelGGL(k_bwd_dense_col_b, grid, dim3(kBBlock), lds, s, q, kW, img, gE, gr.colptr, gr.crow, gr.cperm, dq, scales, 1.0f,
                     G, N, gr.nnz, NTp, nb);
  MSGAT_CHECK_LAUNCH();
  return MSGAT_OK;
}

}  // namespace msgat
