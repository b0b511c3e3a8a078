// fp16 two-term operands for the payload product of the dense attention passes (dense.hip, dense_bf16.hip).
//
// The payload product of a 16x16 score tile, D2[s][own] += sum_items payload[item][s] P[own][item], is bound by what it costs
// to turn P (fresh out of v_exp_f32, fp32, in the B layout of the MFMA already) into matrix-core operands.  P is bounded, so
// it is written as TWO fp16 terms (11 significand bits each; what is dropped is below 2^-23 of the value or 2^-25 absolute):
// v_cvt_pk_f16_f32, two v_fma_mix_f32 (the residual p - h with the half-precision operand read in place),
// v_cvt_pk_f16_f32 -- four instructions per pair of values.  The payload matrix is two fp16 terms as well, times a power of
// two that puts its largest entry at 2^13 .. 2^14 (fp16 has no range to spare).  Two v_mfma_f32_16x16x32_f16 per tile:
//     instr 0:  payload (h | m)  x  P (h | m)  =  q_h P_h + q_m P_m        instr 1:  payload (h | m)  x  P (m | h)
// one payload fragment serving both (the second P fragment is the first with its halves exchanged), fp32 accumulate.
#pragma once
#include "common.hpp"

namespace msgat {

typedef float f32x4h __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2h __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4h mfma_h(const uint4& a, const uint4& b, f32x4h c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two fp32 -> two fp16 (round to nearest even), first argument in the low half: v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi) {
  const f32x2h v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// x = h + m in fp16 (x already scaled into range): the pair of terms of four values as (h0 h1 | h2 h3), (m0 m1 | m2 m3).
// `one` is 1.0f that the compiler cannot see through: fma(float(h), -one, x) then selects v_fma_mix_f32, which reads the
// half-precision operand in place (a plain x - float(h) is a v_cvt_f32_f16 and a v_sub per value).
__device__ __forceinline__ void split2_f16(const float* x, float one, uint32_t& H01, uint32_t& H23, uint32_t& M01, uint32_t& M23) {
  H01 = cvt_pk_f16(x[0], x[1]);
  H23 = cvt_pk_f16(x[2], x[3]);
  const f16x2 a = __builtin_bit_cast(f16x2, H01), b = __builtin_bit_cast(f16x2, H23);
  const float r0 = __builtin_fmaf((float)a[0], -one, x[0]), r1 = __builtin_fmaf((float)a[1], -one, x[1]);
  const float r2 = __builtin_fmaf((float)b[0], -one, x[2]), r3 = __builtin_fmaf((float)b[1], -one, x[3]);
  M01 = cvt_pk_f16(r0, r1);
  M23 = cvt_pk_f16(r2, r3);
}
// P of one tile (4 values per lane, fp32, already times 2^7 or 2^14) -> the B fragment of the payload MFMAs: (Ph | Pm)
__device__ __forceinline__ uint4 split_p(const float* p, float one) {
  uint4 F;
  split2_f16(p, one, F.x, F.y, F.z, F.w);
  return F;
}
__device__ __forceinline__ uint4 swap_halves(const uint4& f) { return make_uint4(f.z, f.w, f.x, f.y); }   // (h | m) -> (m | h)

constexpr float kPOffF = 7.f, kPOffB = 14.f;  // P is carried times 2^7 (forward: P <= 2^8 under the deferred maximum) / 2^14 (backward: P <= 1)

// the exponent e with absmax * 2^e in [2^13, 2^14) (a huge one for an all-zero set: no constraint; clamped for absurd magnitudes)
__device__ __forceinline__ int payload_scale_exp(float absmax) {
  int e = (int)((__float_as_uint(absmax) >> 23) & 0xffu);   // biased exponent: absmax in [2^(e-127), 2^(e-126))
  if (absmax == 0.f) return 100;
  e = min(max(e, 30), 240);
  return 13 - (e - 127);
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((uint32_t)(127 + min(max(e, -126), 127)) << 23); }

}  // namespace msgat
