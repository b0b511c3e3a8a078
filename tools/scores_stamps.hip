// Diagnostic (GPU box): where one block of k_scores spends its cycles (-DMSGAT_STAMPS build).
#define MSGAT_STAMPS 1
#include "../ms_gat_amd/csrc/dense.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
int main(int argc, char** argv) {   // scores_stamps [G N nnz]  (default: the PEMSD7 workload; PEMSD4: 64 307 987)
  const int G = argc > 1 ? atoi(argv[1]) : 96, N = argc > 2 ? atoi(argv[2]) : 883, T = 12;
  const int nnz = argc > 3 ? atoi(argv[3]) : 2615, Bg = G / (argc > 1 ? 1 : 3);
  float *q, *Wg, *kW, *lse, *pq, *E, *val; int *rowptr, *col, *erow;
  hipMalloc(&q, (size_t)G * N * T * 4); hipMalloc(&kW, (size_t)G * N * T * 4); hipMalloc(&pq, (size_t)G * N * T * 4);
  hipMalloc(&Wg, 3 * T * T * 4); hipMalloc(&lse, G * N * 4); hipMalloc(&E, (size_t)G * nnz * 4); hipMalloc(&val, nnz * 4);
  hipMalloc(&rowptr, (N + 1) * 4); hipMalloc(&col, nnz * 4); hipMalloc(&erow, nnz * 4);
  std::vector<int> rp(N + 1), cl(nnz), er(nnz);
  for (int i = 0; i <= N; ++i) rp[i] = (int)((long long)i * nnz / N);
  for (int i = 0; i < N; ++i) for (int e = rp[i]; e < rp[i + 1]; ++e) { er[e] = i; cl[e] = (i * 7 + e) % N; }
  hipMemcpy(rowptr, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(col, cl.data(), nnz * 4, hipMemcpyHostToDevice);
  hipMemcpy(erow, er.data(), nnz * 4, hipMemcpyHostToDevice);
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hq((size_t)G * N * T), hw(3 * T * T), hv(nnz, 0.3f);
  for (auto& v : hq) v = nd(rng); for (auto& v : hw) v = 0.3f * nd(rng);
  hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Wg, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(val, hv.data(), nnz * 4, hipMemcpyHostToDevice);
  msgat_graph_t gr{};
  gr.n_nodes = N; gr.nnz = nnz; gr.rowptr = rowptr; gr.col = col; gr.val = val; gr.erow = erow;
  gr.colptr = rowptr; gr.crow = erow; gr.cperm = col; gr.cpos = col;
  for (size_t l : {0, 8192, 13056, 16384, 32768}) {
    int nblk = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)msgat::k_scores<12, true>, msgat::kDBlock, l);
    printf("occupancy API: dynamic LDS %zu -> %d blocks/CU\n", l, nblk);
  }
  for (int i = 0; i < 3; ++i) msgat::launch_scores(gr, q, Wg, kW, lse, pq, E, nullptr, G, Bg, N, T, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(8 * 4096);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(msgat::g_stamps), st.size() * 8);
  const bool seven = (N == 883 || (argc > 4 && atoi(argv[4]) == 7));   // the 7 + 1 wave form stamps its own phases
  const int nb = std::min(4096, G * ((N + (seven ? msgat::kHRows : msgat::kDRows) - 1) / (seven ? msgat::kHRows : msgat::kDRows)));
  const char* names8[] = {"prologue (row fragment, kW)", "first chunk (stage + tiles)", "remaining chunks", "merge + stores", "edge pass"};
  const char* names7[] = {"helper columns staged + planes, rows' kW", "first chunk (stage, planes, tiles)", "remaining chunks", "wait for helper + fold + stores", "edge pass"};
  const char** names = seven ? names7 : names8;
  for (int ph = 0; ph < 5; ++ph) {
    std::vector<double> d;
    for (int b = 0; b < nb; ++b) d.push_back((double)(st[b * 8 + ph + 1] - st[b * 8 + ph]));
    std::sort(d.begin(), d.end());
    printf("%-30s median %8.0f  p10 %8.0f  p90 %8.0f cycles\n", names[ph], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
  }
  unsigned long long r0 = ~0ull, r1 = 0; std::vector<double> tot;
  for (int b = 0; b < nb; ++b) { tot.push_back((double)(st[b * 8 + 5] - st[b * 8])); r0 = std::min(r0, st[b * 8 + 6]); r1 = std::max(r1, st[b * 8 + 7]); }
  std::sort(tot.begin(), tot.end());
  printf("block total median %.0f cycles; kernel wall %.1f us; concurrency:", tot[tot.size() / 2], (r1 - r0) / 100.0);
  for (int qd = 0; qd < 16; ++qd) { const double tq = r0 + (double)(r1 - r0) * (qd + 0.5) / 16; int live = 0; for (int b = 0; b < nb; ++b) live += (st[b * 8 + 6] <= tq && tq < st[b * 8 + 7]); printf(" %d", live); }
  printf("\n");
  printf("block cycles p0 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f\n", tot[0], tot[tot.size()/10], tot[tot.size()/2], tot[tot.size()*9/10], tot[tot.size()*99/100], tot.back());
  // by XCD guess (linear block id % 8) and by start time
  for (int x = 0; x < 8; ++x) {
    double s0 = 0, s1 = 0, dur = 0; int cnt = 0;
    for (int b = x; b < nb; b += 8) { s0 += (st[b*8+6]-r0)/100.0; s1 += (st[b*8+7]-r0)/100.0; dur += (double)(st[b*8+5]-st[b*8]); ++cnt; }
    printf("  id%%8=%d: n=%d mean start %.1f us, mean end %.1f us, mean cycles %.0f\n", x, cnt, s0/cnt, s1/cnt, dur/cnt);
  }
  {
    std::vector<unsigned> hw(4096);
    hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(msgat::g_hwid), hw.size() * 4);
    // HW_ID: cu_id bits [11:8], sh_id [12], se_id [15:13]; key = (xcc, se, sh, cu)
    std::vector<int> perkey(1 << 16, 0); std::vector<double> durkey(1 << 16, 0.0);
    for (int b = 0; b < nb; ++b) { const unsigned key = ((hw[b] >> 16) & 0xf) << 8 | ((hw[b] >> 8) & 0xff); perkey[key]++; durkey[key] += (double)(st[b*8+5]-st[b*8]); }
    int hist[16] = {0}; double dsum[16] = {0}; int ncu_used = 0;
    for (int k = 0; k < (1 << 16); ++k) if (perkey[k]) { ++ncu_used; const int c = std::min(perkey[k], 15); hist[c]++; dsum[c] += durkey[k] / perkey[k]; }
    printf("CUs used: %d; blocks-per-CU histogram:", ncu_used);
    for (int c = 1; c < 16; ++c) if (hist[c]) printf("  %d blocks: %d CUs (mean block cycles %.0f)", c, hist[c], dsum[c] / hist[c]);
    printf("\n");
  }
  // how many blocks start late
  int late = 0; for (int b = 0; b < nb; ++b) late += ((st[b*8+6]-r0)/100.0 > 5.0);
  printf("blocks starting later than 5 us after the first: %d\n", late);
  return 0;
}
