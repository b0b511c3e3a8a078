// Diagnostic (GPU box): where one block of k_chanpair_mfma spends its cycles.  Separate build with
// -DMSGAT_STAMPS; the product library never contains the stamps.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -Ims_gat_amd/csrc tools/contract_stamps.hip -o build/contract_stamps
#define MSGAT_STAMPS 1
#include "../ms_gat_amd/csrc/mfma.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

int main() {
  const int G = 96, Ca = 25, Cb = 72, N = 883, T = 12, P = N * T;
  float *A, *Ax, *B, *part;
  hipMalloc(&A, (size_t)G * 24 * P * 4); hipMalloc(&Ax, (size_t)G * P * 4); hipMalloc(&B, (size_t)G * Cb * P * 4);
  const int R = 3, Bg = G / R;
  const int nspan = msgat::chanpair_mfma_blocks(R);
  hipMalloc(&part, (size_t)R * nspan * Ca * Cb * 4);
  hipMemset(A, 0, (size_t)G * 24 * P * 4); hipMemset(Ax, 0, (size_t)G * P * 4); hipMemset(B, 0, (size_t)G * Cb * P * 4);
  {
    int nblk = -1;
    const size_t lds = sizeof(float4) * (size_t)std::max(((2 + 5) * 16 + 1) * msgat::kRowF4, msgat::kCpWaves * 2 * 5 * 64);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)msgat::k_chanpair_mfma<2, 5>, msgat::kCpBlock, lds);
    printf("occupancy API: %d blocks/CU at %zu B LDS (%s)\n", nblk, lds, hipGetErrorString(e));
    for (size_t l : {0, 16384, 32768, 40960, 49152, 53248, 57344, 65536, 81920}) {
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)msgat::k_chanpair_mfma<2, 5>, msgat::kCpBlock, l);
      printf("  LDS %6zu -> %d blocks/CU;", l, nblk);
    }
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)msgat::k_chanpair_mfma<1, 1>, msgat::kCpBlock, 16384);
    printf("\n  <1,1> at 16 KB -> %d blocks/CU\n", nblk);
  }
  for (int i = 0; i < 3; ++i) msgat::launch_chanpair_mfma(A, Ax, B, part, R, Bg, Ca, Cb, P, nspan, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(8 * 4096);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(msgat::g_stamps), st.size() * 8);
  const int nb = std::min(4096, R * nspan);
  const char* names[] = {"entry->first fetches issued", "->tile 0 stashed", "->tile 0 multiplied", "->loop done (remaining tiles)", "->reduced+written"};
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int b = 0; b < nb; ++b) { tmin = std::min(tmin, st[b * 8]); tmax = std::max(tmax, st[b * 8 + 5]); }
  printf("blocks %d, kernel span %.1f us (100 MHz ticks?) raw ticks %llu\n", nb, (tmax - tmin) / 100.0, tmax - tmin);
  for (int ph = 0; ph < 5; ++ph) {
    std::vector<double> d;
    for (int b = 0; b < nb; ++b) d.push_back((double)(st[b * 8 + ph + 1] - st[b * 8 + ph]));
    std::sort(d.begin(), d.end());
    printf("%-34s median %8.0f  p10 %8.0f  p90 %8.0f ticks\n", names[ph], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
  }
  std::vector<double> tot;
  for (int b = 0; b < nb; ++b) tot.push_back((double)(st[b * 8 + 5] - st[b * 8]));
  std::sort(tot.begin(), tot.end());
  printf("block total median %8.0f ticks\n", tot[tot.size() / 2]);
  std::vector<double> clk;
  unsigned long long r0 = ~0ull, r1 = 0;
  for (int b = 0; b < nb; ++b) {
    clk.push_back((double)(st[b * 8 + 5] - st[b * 8]) / (double)(st[b * 8 + 7] - st[b * 8 + 6]) * 100.0);
    r0 = std::min(r0, st[b * 8 + 6]); r1 = std::max(r1, st[b * 8 + 7]);
  }
  std::sort(clk.begin(), clk.end());
  printf("in-kernel clock median %.0f MHz; kernel wall (memrealtime, 100 MHz) %.1f us\n", clk[clk.size() / 2], (r1 - r0) / 100.0);
  // concurrency profile from the realtime stamps (100 MHz)
  const int NB_ = 24;
  double span = (double)(r1 - r0);
  printf("concurrent blocks over time:");
  for (int q = 0; q < NB_; ++q) {
    const double tq = r0 + span * (q + 0.5) / NB_;
    int live = 0;
    for (int b = 0; b < nb; ++b) live += (st[b * 8 + 6] <= tq && tq < st[b * 8 + 7]);
    printf(" %d", live);
  }
  printf("\n");
  return 0;
}
