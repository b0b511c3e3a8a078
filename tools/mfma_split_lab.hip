// Scratch experiment (GPU box), round 6, step 0 of the "dense products on the bf16 matrix core" question:
//   A. does v_mfma_f32_16x16x32_bf16 issue beside VALU work on one SIMD (max of the two pipes) where
//      v_mfma_f32_16x16x4_f32 does not (tools/mfma_rate.hip: sum)?  Bare rates, then a register-only model of one
//      trip of the dense passes (two 16x16 score tiles: score chain -> max / exp2 -> payload product) in five forms.
//   B. what a K = 12 dot product costs in accuracy when every fp32 operand is three bf16 terms (hi, mid, lo; exact)
//      and the products hh, hm, mh, hl, lh, mm (6) or all but ll (8) are stacked on the K axis of the bf16 MFMA with
//      fp32 accumulate: error against float64 next to the fp32 fmaf chain's; and whether the result depends on the
//      tile position of an operand or on which side (A or B) it is on -- forward and backward must agree bit for bit.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_split_lab.hip -o build/mfma_split_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_bf(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- A1: bare issue rates ---------------------------------------------------------------------------------
template <int NV, bool EXP>
__global__ void k_bf_mix(float* out, int iters, float seed) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = {seed, seed, seed, seed};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  const uint32_t w = 0x3f803f80u + threadIdx.x;   // two bf16 near 1
  const uint4 a = {w, w, w, w}, b = {w + 1, w, w + 1, w};
  const float fb = seed * 0.5f, fc = seed * 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u] = mfma_bf(a, b, acc[u]);
#pragma unroll
      for (int k = 0; k < NV / 8; ++k) v[(u + k) & 7] = EXP ? __builtin_amdgcn_exp2f(v[(u + k) & 7]) : fmaf(v[(u + k) & 7], fb, fc);
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  if (r == 1234.5f) out[0] = r;
}

// ---- A2: one trip of the dense passes, operands in registers ----------------------------------------------
// MODE 0: fp32 score chains (2 x 3) + fp32 payload (8)             -- k_scores7 / k_bwd_dense_col7 today
// MODE 1: bf16 score products (2 x 3 of 16x16x32) + fp32 payload   -- scores split, P not
// MODE 2: bf16 scores + bf16 payload: P split into three truncated bf16 terms per tile (6 of 16x16x32 per trip)
// MODE 3: fp32 scores, no payload (the inference forward)    MODE 4: bf16 scores, no payload
template <int MODE>
__global__ void k_trip(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  const float fa[3] = {seed + lane * 1e-3f, seed * 0.5f, seed * 0.25f};
  const float fb[3] = {0.01f * seed, 0.02f, 0.03f};
  const uint32_t w = 0x3c003c00u + lane;
  const uint4 a0 = {w, w + 1, w + 2, w + 3}, a1 = {w + 4, w, w + 5, w}, a2 = {w + 6, w + 7, w, w};
  const uint4 b0 = {w + 1, w, w, w + 2}, b1 = {w, w + 3, w, w}, b2 = {w + 2, w, w + 1, w};
  float m = 0.f;
  f32x4 da = {0, 0, 0, 0}, db = da, dc = da, dd = da;
  float lsum = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x4 S0 = {0, 0, 0, 0}, S1 = S0;
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) S0 = mfma_f32(fa[kk], fb[kk], S0);
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) S1 = mfma_f32(fa[2 - kk], fb[kk], S1);
    } else {
      S0 = mfma_bf(a0, b0, S0); S0 = mfma_bf(a1, b1, S0); S0 = mfma_bf(a2, b2, S0);
      S1 = mfma_bf(a1, b0, S1); S1 = mfma_bf(a2, b1, S1); S1 = mfma_bf(a0, b2, S1);
    }
    float sv[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { sv[r] = S0[r]; sv[4 + r] = S1[r]; }
    const float cm = fmaxf(fmaxf(fmaxf(fmaxf(sv[0], sv[1]), sv[2]), fmaxf(fmaxf(sv[3], sv[4]), sv[5])), fmaxf(sv[6], sv[7]));
    if (__any(cm > m + 8.f)) {
      const float mn = fmaxf(m, cm);
      const float sc = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      lsum *= sc;
#pragma unroll
      for (int r = 0; r < 4; ++r) { da[r] *= sc; db[r] *= sc; dc[r] *= sc; dd[r] *= sc; }
    }
    float p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) p[r] = __builtin_amdgcn_exp2f(sv[r] - m);
    if (MODE == 3 || MODE == 4) lsum += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    if (MODE == 0 || MODE == 1) {
      da = mfma_f32(fa[0], p[0], da); db = mfma_f32(fa[1], p[1], db); dc = mfma_f32(fa[2], p[2], dc); dd = mfma_f32(fa[0], p[3], dd);
      da = mfma_f32(fa[1], p[4], da); db = mfma_f32(fa[2], p[5], db); dc = mfma_f32(fa[0], p[6], dc); dd = mfma_f32(fa[1], p[7], dd);
    }
    if (MODE == 2) {
      // three truncated bf16 terms per P value (exact: 8 + 8 + 8 bits), packed two to a register by v_perm_b32
      uint32_t h[8], mm[8], l[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t pu = __float_as_uint(p[r]);
        h[r] = pu;
        const float r1 = p[r] - __uint_as_float(pu & 0xffff0000u);
        const uint32_t r1u = __float_as_uint(r1);
        mm[r] = r1u;
        l[r] = __float_as_uint(r1 - __uint_as_float(r1u & 0xffff0000u));
      }
      auto pk = [](uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };  // upper halves of both
#pragma unroll
      for (int t = 0; t < 2; ++t) {   // per 16-column tile: B slots Ph Ph Pm Ph Pl Pm for each of the lane's 4 columns
        const int o = 4 * t;
        const uint4 B0 = {pk(h[o], h[o + 1]), pk(h[o + 2], h[o + 3]), pk(h[o], h[o + 1]), pk(h[o + 2], h[o + 3])};
        const uint4 B1 = {pk(mm[o], mm[o + 1]), pk(mm[o + 2], mm[o + 3]), B0.x, B0.y};
        const uint4 B2 = {pk(l[o], l[o + 1]), pk(l[o + 2], l[o + 3]), B1.x, B1.y};
        if (t == 0) { da = mfma_bf(a0, B0, da); db = mfma_bf(a1, B1, db); dc = mfma_bf(a2, B2, dc); }
        else        { dd = mfma_bf(a0, B0, dd); da = mfma_bf(a1, B1, da); db = mfma_bf(a2, B2, db); }
      }
    }
  }
  float r = lsum + m;
  for (int i = 0; i < 4; ++i) r += da[i] + db[i] + dc[i] + dd[i];
  if (r == 1234.5f) out[0] = r;
}

// ---- B: accuracy and position independence ------------------------------------------------------------------
// 16 "column" vectors a[i][0..T) and 16 "row" vectors b[j][0..T), T = 12; S[i][j] = sum_t a[i][t] b[j][t].
constexpr int T = 12;
__host__ __device__ inline uint16_t bf16_rne(float x) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(x);
#else
  memcpy(&u, &x, 4);
#endif
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f; memcpy(&f, &u, 4); return f;
#endif
}
__host__ __device__ inline void split3(float x, uint16_t t[3]) {
  t[0] = bf16_rne(x);
  const float r1 = x - bf16_f(t[0]);
  t[1] = bf16_rne(r1);
  const float r2 = r1 - bf16_f(t[1]);
  t[2] = bf16_rne(r2);
}
// slot s of the stacked K axis: product s / T, timestep s % T; terms of the A side / B side per product
__constant__ int kTa[8] = {0, 0, 1, 0, 2, 1, 1, 2};
__constant__ int kTb[8] = {0, 1, 0, 2, 0, 1, 2, 1};

// NP = 6 or 8 products.  a, b: [16][T] floats.  perm: the tile position operand i is placed at.  out[i][j].
// SWAP: the b vectors go to the A side and the a vectors to the B side (the backward's roles); out still [i][j].
template <int NP, bool SWAP>
__global__ void k_split_tile(const float* a, const float* b, const int* perm, float* out) {
  const int lane = threadIdx.x, j = lane & 15, quad = lane >> 4;
  // the vector this lane supplies on each side
  const float* va = SWAP ? b + j * T : a + perm[j] * T;     // A side: row index of D = lane & 15
  const float* vb = SWAP ? a + perm[j] * T : b + j * T;     // B side: column index of D = lane & 15
  uint16_t ta[T][3], tb[T][3];
  for (int t = 0; t < T; ++t) { split3(va[t], ta[t]); split3(vb[t], tb[t]); }
  f32x4 acc = {0, 0, 0, 0};
  constexpr int NS = NP * T;
  for (int i = 0; i < (NS + 31) / 32; ++i) {
    uint16_t fa[8], fbv[8];
    for (int jj = 0; jj < 8; ++jj) {
      const int s = 32 * i + 8 * quad + jj;
      const int p = s / T, t = s % T;
      const bool live = s < NS;
      // A side always carries the "a-role" term table when !SWAP; when SWAP the A side holds b vectors, which
      // take the B-role terms, so that every slot multiplies the same pair of terms as in the other orientation
      fa[jj] = live ? ta[t][SWAP ? kTb[p] : kTa[p]] : 0;
      fbv[jj] = live ? tb[t][SWAP ? kTa[p] : kTb[p]] : 0;
    }
    uint4 A, B;
    A.x = fa[0] | (fa[1] << 16); A.y = fa[2] | (fa[3] << 16); A.z = fa[4] | (fa[5] << 16); A.w = fa[6] | (fa[7] << 16);
    B.x = fbv[0] | (fbv[1] << 16); B.y = fbv[2] | (fbv[3] << 16); B.z = fbv[4] | (fbv[5] << 16); B.w = fbv[6] | (fbv[7] << 16);
    acc = mfma_bf(A, B, acc);
  }
  // D[row = 4 quad + r][col = j]
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * quad + r, colm = j;
    if (!SWAP) {   // row <-> a vector at position row (= a[perm[row]]), col <-> b[j]
      // find which a vector sits at tile position `row`: perm[row]
      out[perm[row] * 16 + colm] = acc[r];
    } else {       // row <-> b[row], col <-> a[perm[colm]]
      out[perm[colm] * 16 + row] = acc[r];
    }
  }
}

__global__ void k_fma_tile(const float* a, const float* b, float* out) {
  const int i = threadIdx.x / 16, j = threadIdx.x % 16;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s = fmaf(a[i * T + t], b[j * T + t], s);
  out[i * 16 + j] = s;
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  const int iters = 4000;
  printf("== A1: v_mfma_f32_16x16x32_bf16 beside VALU work (cycles at 2.4 GHz per trip of 8 MFMAs per wave slot) ==\n");
  for (int wps : {1, 2, 4}) {
    dim3 grid(256 * wps), block(256);
    const char* nm[] = {"8 MFMA alone", "8 MFMA + 16 v_fma", "8 MFMA + 32 v_fma", "8 MFMA + 64 v_fma", "8 MFMA + 8 v_exp", "8 MFMA + 16 v_exp", "8 MFMA + 32 v_exp"};
    for (int mode = 0; mode < 7; ++mode) {
      const float ms = time_ms([&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL((k_bf_mix<0, false>), grid, block, 0, 0, out, iters, 1.0001f); break;
          case 1: hipLaunchKernelGGL((k_bf_mix<16, false>), grid, block, 0, 0, out, iters, 1.0001f); break;
          case 2: hipLaunchKernelGGL((k_bf_mix<32, false>), grid, block, 0, 0, out, iters, 1.0001f); break;
          case 3: hipLaunchKernelGGL((k_bf_mix<64, false>), grid, block, 0, 0, out, iters, 1.0001f); break;
          case 4: hipLaunchKernelGGL((k_bf_mix<8, true>), grid, block, 0, 0, out, iters, 1.0001f); break;
          case 5: hipLaunchKernelGGL((k_bf_mix<16, true>), grid, block, 0, 0, out, iters, 1.0001f); break;
          default: hipLaunchKernelGGL((k_bf_mix<32, true>), grid, block, 0, 0, out, iters, 1.0001f); break;
        }
      });
      printf("%-20s %d waves/SIMD: %7.1f cycles per trip per wave slot\n", nm[mode], wps, ms * 1e6 / ((double)iters * wps) * 2.4);
    }
  }
  printf("== A2: register-only model of one trip (two 16x16 tiles) of the dense passes ==\n");
  const char* tn[] = {"fp32 scores + fp32 payload (today)", "bf16x3 scores + fp32 payload", "bf16x3 scores + bf16x3 payload (P split per tile)",
                      "fp32 scores, no payload (inference)", "bf16x3 scores, no payload"};
  for (int wps : {1, 2, 4, 6}) {
    dim3 grid(256 * wps), block(256);
    for (int mode = 0; mode < 5; ++mode) {
      const float ms = time_ms([&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_trip<0>, grid, block, 0, 0, out, iters, 1.0001f); break;
          case 1: hipLaunchKernelGGL(k_trip<1>, grid, block, 0, 0, out, iters, 1.0001f); break;
          case 2: hipLaunchKernelGGL(k_trip<2>, grid, block, 0, 0, out, iters, 1.0001f); break;
          case 3: hipLaunchKernelGGL(k_trip<3>, grid, block, 0, 0, out, iters, 1.0001f); break;
          default: hipLaunchKernelGGL(k_trip<4>, grid, block, 0, 0, out, iters, 1.0001f); break;
        }
      });
      printf("%-52s %d waves/SIMD: %7.1f cycles per trip per wave slot = %6.1f per 16x16 tile\n", tn[mode], wps,
             ms * 1e6 / ((double)iters * wps) * 2.4, ms * 1e6 / ((double)iters * wps) * 1.2);
    }
  }

  printf("== B: K = 12 dot products, three bf16 terms per operand, fp32 accumulate ==\n");
  float *da, *db, *dout; int* dperm;
  (void)hipMalloc(&da, 16 * T * 4); (void)hipMalloc(&db, 16 * T * 4); (void)hipMalloc(&dout, 256 * 4); (void)hipMalloc(&dperm, 64);
  srand(7);
  auto gauss = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
  for (double scale : {1.0, 30.0, 1e-3}) {
    double e_fma = 0, e6 = 0, e8 = 0, e_fma_max = 0, e6_max = 0, e8_max = 0; int mism_pos = 0, mism_swap = 0, n = 0;
    for (int trial = 0; trial < 200; ++trial) {
      std::vector<float> a(16 * T), b(16 * T);
      for (auto& x : a) x = (float)(gauss() * scale);
      for (auto& x : b) x = (float)(gauss() * scale);
      int id[16], pm[16];
      for (int i = 0; i < 16; ++i) { id[i] = i; pm[i] = i; }
      for (int i = 15; i > 0; --i) { int k = rand() % (i + 1); int t = pm[i]; pm[i] = pm[k]; pm[k] = t; }
      (void)hipMemcpy(da, a.data(), 16 * T * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(db, b.data(), 16 * T * 4, hipMemcpyHostToDevice);
      std::vector<float> r_fma(256), r6(256), r8(256), r6p(256), r6s(256);
      hipLaunchKernelGGL(k_fma_tile, dim3(1), dim3(256), 0, 0, da, db, dout);
      (void)hipMemcpy(r_fma.data(), dout, 1024, hipMemcpyDeviceToHost);
      (void)hipMemcpy(dperm, id, 64, hipMemcpyHostToDevice);
      hipLaunchKernelGGL((k_split_tile<6, false>), dim3(1), dim3(64), 0, 0, da, db, dperm, dout);
      (void)hipMemcpy(r6.data(), dout, 1024, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL((k_split_tile<8, false>), dim3(1), dim3(64), 0, 0, da, db, dperm, dout);
      (void)hipMemcpy(r8.data(), dout, 1024, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL((k_split_tile<6, true>), dim3(1), dim3(64), 0, 0, da, db, dperm, dout);
      (void)hipMemcpy(r6s.data(), dout, 1024, hipMemcpyDeviceToHost);
      (void)hipMemcpy(dperm, pm, 64, hipMemcpyHostToDevice);
      hipLaunchKernelGGL((k_split_tile<6, false>), dim3(1), dim3(64), 0, 0, da, db, dperm, dout);
      (void)hipMemcpy(r6p.data(), dout, 1024, hipMemcpyDeviceToHost);
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = 0, mag = 0;
          for (int t = 0; t < T; ++t) { ref += (double)a[i * T + t] * b[j * T + t]; mag += fabs((double)a[i * T + t] * b[j * T + t]); }
          const double f = fabs(r_fma[i * 16 + j] - ref) / mag, s6 = fabs(r6[i * 16 + j] - ref) / mag, s8 = fabs(r8[i * 16 + j] - ref) / mag;
          e_fma += f; e6 += s6; e8 += s8; ++n;
          e_fma_max = fmax(e_fma_max, f); e6_max = fmax(e6_max, s6); e8_max = fmax(e8_max, s8);
          mism_pos += memcmp(&r6[i * 16 + j], &r6p[i * 16 + j], 4) != 0;
          mism_swap += memcmp(&r6[i * 16 + j], &r6s[i * 16 + j], 4) != 0;
        }
    }
    printf("scale %g: error / sum|a b|  fp32 fmaf chain mean %.3e max %.3e | 6 products mean %.3e max %.3e | 8 products mean %.3e max %.3e | "
           "bit mismatches of %d: tile position %d, A/B sides swapped %d\n", scale, e_fma / n, e_fma_max, e6 / n, e6_max, e8 / n, e8_max, n, mism_pos, mism_swap);
  }
  return 0;
}
