// Lane-contiguous access to rows of T floats for kernels that work one LANE PER ROW.
//
// A lane that loads its own row touches 16 B of every 4T-byte row per instruction: three instructions (T = 12) share
// each 128-B line, and on tensors beyond the 256 MB cache that pattern streams ~12 % slower than flat accesses
// (LayerNorm forward at G = 96: 127 -> 111 us, 4.6 -> 5.3 TB/s; profiles/r03/rowtile_lab.txt).  Here a wave's 64 rows
// -- 64 * T/4 float4 of consecutive bytes -- are moved in FLAT order (lane l takes float4 number 64 i + l: every
// instruction covers 1 KiB of consecutive bytes) and change hands through a wave-private LDS tile: written flat, read
// row-wise (stride T/4 float4: conflict-free at T = 12) and back.  LDS operations of one wave execute in order, so a
// wave-private tile needs no workgroup barrier, only that the compiler keeps the order (the "memory" clobbers).
#pragma once
#include "common.hpp"

namespace msgat {

template <int T>
struct RowTile {
  static constexpr int T4 = T / 4;
  static constexpr int kFloat4s = kWave * T4;   // LDS float4s per wave
  float4* tile;                                  // this wave's kFloat4s float4
  int lane;

  __device__ __forceinline__ RowTile(float4* wave_tile, int lane_) : tile(wave_tile), lane(lane_) {}

  static __device__ __forceinline__ void fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

  // global -> registers, flat: nf = float4s of the wave's valid rows (the load is clamped into them, never predicated)
  __device__ __forceinline__ void fetch(const float* __restrict__ base, int nf, float4 (&in)[T4]) const {
    const float4* src = reinterpret_cast<const float4*>(base);
#pragma unroll
    for (int i = 0; i < T4; ++i) in[i] = src[min(kWave * i + lane, nf - 1)];
  }
  // flat registers -> this lane's row
  __device__ __forceinline__ void to_row(const float4 (&in)[T4], float (&v)[T]) const {
#pragma unroll
    for (int i = 0; i < T4; ++i) tile[kWave * i + lane] = in[i];
    fence();
#pragma unroll
    for (int j = 0; j < T4; ++j) {
      const float4 a = tile[lane * T4 + j];
      v[4 * j + 0] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w;
    }
    fence();   // every lane holds its row before the tile is written again
  }
  // this lane's row -> global, flat
  __device__ __forceinline__ void store(float* __restrict__ base, int nf, const float (&v)[T]) const {
#pragma unroll
    for (int j = 0; j < T4; ++j) tile[lane * T4 + j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    fence();
    float4* dst = reinterpret_cast<float4*>(base);
#pragma unroll
    for (int i = 0; i < T4; ++i) {
      const float4 o = tile[kWave * i + lane];
      if (kWave * i + lane < nf) dst[kWave * i + lane] = o;
    }
    fence();
  }
};

}  // namespace msgat
