// Host side of the boundary: dense [N,N] adjacency -> CSR + CSC.
//
// The reference keeps the adjacency dense (built at data_loader.py:59-66, held as a
// frozen parameter at msgat.py:190) and applies it as a mask `att * adjacency`
// (attention.py:36).  Only its non-zeros matter to that product, so the kernels walk
// them: CSR for the forward gather, CSC (+ the CSC->CSR permutation) for the transposed
// gather of the backward pass.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "msgat_hip.h"

static inline bool is_edge(float v) { return !(v == 0.0f); }  // NaN is an edge (it would poison the dense product too)

extern "C" int msgat_graph_count(const float* adj, int32_t n, int64_t ld, int32_t* nnz_out) {
  if (!adj || !nnz_out) return MSGAT_ERR_NULL;
  if (n <= 0 || ld < n) return MSGAT_ERR_SHAPE;
  int64_t cnt = 0;
  for (int32_t i = 0; i < n; ++i) {
    const float* row = adj + (int64_t)i * ld;
    for (int32_t j = 0; j < n; ++j) cnt += is_edge(row[j]) ? 1 : 0;
  }
  if (cnt > INT32_MAX) return MSGAT_ERR_SHAPE;
  *nnz_out = (int32_t)cnt;
  return MSGAT_OK;
}

extern "C" int msgat_graph_build(const float* adj, int32_t n, int64_t ld, int32_t nnz,
                                 int32_t* rowptr, int32_t* col, float* val, int32_t* erow,
                                 int32_t* colptr, int32_t* crow, int32_t* cperm) {
  if (!adj || !rowptr || !colptr) return MSGAT_ERR_NULL;
  if (nnz > 0 && (!col || !val || !erow || !crow || !cperm)) return MSGAT_ERR_NULL;
  if (n <= 0 || ld < n || nnz < 0) return MSGAT_ERR_SHAPE;

  // CSR, rows in order, columns ascending inside a row
  std::vector<int32_t> colcount((size_t)n + 1, 0);
  int64_t e = 0;
  for (int32_t i = 0; i < n; ++i) {
    rowptr[i] = (int32_t)e;
    const float* row = adj + (int64_t)i * ld;
    for (int32_t j = 0; j < n; ++j) {
      if (!is_edge(row[j])) continue;
      if (e >= nnz) return MSGAT_ERR_SHAPE;  // caller's count is stale
      col[e] = j;
      val[e] = row[j];
      erow[e] = i;
      colcount[(size_t)j + 1]++;
      ++e;
    }
  }
  if (e != nnz) return MSGAT_ERR_SHAPE;
  rowptr[n] = nnz;

  // CSC by counting sort over the CSR edges: rows ascending inside a column
  colptr[0] = 0;
  for (int32_t j = 0; j < n; ++j) colptr[j + 1] = colptr[j] + colcount[(size_t)j + 1];
  std::vector<int32_t> cursor(colptr, colptr + n);
  for (int32_t k = 0; k < nnz; ++k) {
    const int32_t pos = cursor[col[k]]++;
    crow[pos] = erow[k];
    cperm[pos] = k;
  }
  return MSGAT_OK;
}

extern "C" int msgat_graph_validate(const msgat_graph_t* g) {
  if (!g || !g->rowptr || !g->colptr) return MSGAT_ERR_NULL;
  const int32_t n = g->n_nodes, nnz = g->nnz;
  if (n <= 0 || nnz < 0) return MSGAT_ERR_SHAPE;
  if (nnz > 0 && (!g->col || !g->val || !g->erow || !g->crow || !g->cperm)) return MSGAT_ERR_NULL;
  if (g->rowptr[0] != 0 || g->rowptr[n] != nnz || g->colptr[0] != 0 || g->colptr[n] != nnz)
    return MSGAT_ERR_GRAPH;
  for (int32_t i = 0; i < n; ++i) {
    if (g->rowptr[i + 1] < g->rowptr[i] || g->colptr[i + 1] < g->colptr[i]) return MSGAT_ERR_GRAPH;
    for (int32_t k = g->rowptr[i]; k < g->rowptr[i + 1]; ++k) {
      if (g->col[k] < 0 || g->col[k] >= n || g->erow[k] != i) return MSGAT_ERR_GRAPH;
      if (k > g->rowptr[i] && g->col[k] <= g->col[k - 1]) return MSGAT_ERR_GRAPH;
    }
  }
  std::vector<char> seen((size_t)nnz, 0);
  for (int32_t j = 0; j < n; ++j) {
    for (int32_t k = g->colptr[j]; k < g->colptr[j + 1]; ++k) {
      const int32_t p = g->cperm[k];
      if (p < 0 || p >= nnz || seen[(size_t)p]) return MSGAT_ERR_GRAPH;
      seen[(size_t)p] = 1;
      if (g->col[p] != j || g->erow[p] != g->crow[k]) return MSGAT_ERR_GRAPH;
    }
  }
  // SELL forms: every CSR edge sits at exactly one position, in the lane that owns the row the edge starts from
  // (rows form) or the column it ends in (columns form), as that lane's k-th entry in CSR / CSC order, and holds
  // the node at the edge's other end; everything else is padding
  for (int form = 0; form < 2; ++form) {
    const msgat_sell_t& j = form == 0 ? g->sell_rows : g->sell_cols;
    if (j.n_slices == 0) continue;
    if (j.n_slices != (n + 63) / 64 || j.n_pos < nnz || !j.slice_off || !j.lane_row) return MSGAT_ERR_GRAPH;
    if (j.n_pos > 0 && (!j.idx || !j.src)) return MSGAT_ERR_GRAPH;
    if (j.slice_off[0] != 0 || j.slice_off[j.n_slices] != j.n_pos) return MSGAT_ERR_GRAPH;
    const int32_t* ptr = form == 0 ? g->rowptr : g->colptr;
    std::vector<char> used((size_t)nnz, 0), seen_row((size_t)n, 0);
    int32_t prev_deg = INT32_MAX;
    for (int32_t s = 0; s < j.n_slices; ++s) {
      const int32_t span = j.slice_off[s + 1] - j.slice_off[s];
      if (span < 0 || span % 256 != 0) return MSGAT_ERR_GRAPH;  // 64 lanes x a multiple of 4 columns
      const int32_t width = span / 64;
      for (int32_t l = 0; l < 64; ++l) {
        const int32_t owner = j.lane_row[64 * s + l];
        if (owner < -1 || owner >= n) return MSGAT_ERR_GRAPH;
        int32_t deg = 0;
        if (owner >= 0) {
          if (seen_row[(size_t)owner]) return MSGAT_ERR_GRAPH;
          seen_row[(size_t)owner] = 1;
          deg = ptr[owner + 1] - ptr[owner];
          if (deg > prev_deg || deg > width) return MSGAT_ERR_GRAPH;  // sorted by degree, descending
          prev_deg = deg;
        }
        for (int32_t k = 0; k < width; ++k) {
          const int32_t p = j.slice_off[s] + 256 * (k >> 2) + 4 * l + (k & 3), e = j.src[p];
          if (k >= deg) {
            if (e != -1 || j.idx[p] >= n) return MSGAT_ERR_GRAPH;
            continue;
          }
          if (e < 0 || e >= nnz || used[(size_t)e]) return MSGAT_ERR_GRAPH;
          used[(size_t)e] = 1;
          const int32_t want = form == 0 ? ptr[owner] + k : g->cperm[ptr[owner] + k];
          const int32_t to = form == 0 ? g->col[e] : g->erow[e];
          if (e != want || j.idx[p] != to) return MSGAT_ERR_GRAPH;
          if (j.pos && j.pos[e] != p) return MSGAT_ERR_GRAPH;
        }
      }
    }
    int32_t pair = 0;
    for (int32_t i = 0; 2 * i < j.n_slices; ++i) {
      const int32_t o = j.n_slices - 1 - i;
      pair = std::max(pair, (j.slice_off[i + 1] - j.slice_off[i]) / 256 + (o > i ? (j.slice_off[o + 1] - j.slice_off[o]) / 256 : 0));
    }
    if (j.pair_trips != pair) return MSGAT_ERR_GRAPH;
    for (int32_t i = 0; i < n; ++i)
      if (!seen_row[(size_t)i]) return MSGAT_ERR_GRAPH;
    for (int32_t e = 0; e < nnz; ++e)
      if (!used[(size_t)e]) return MSGAT_ERR_GRAPH;
    for (int32_t p = j.n_pos; p < j.n_pos + MSGAT_SELL_SLACK; ++p)
      if (j.idx[p] >= n) return MSGAT_ERR_GRAPH;  // the slack is read (and ignored) by the kernels
  }
  return MSGAT_OK;
}

// ---- sliced ELLPACK (SELL-64, degree-sorted) layout of one sparse structure: see msgat_sell_t ---------------------
// order = rows by degree, descending, stable (counting sort); slice s = order[64 s .. 64 s + 63]; its width is the
// degree of its first row rounded up to a multiple of 4; a "trip" = 4 consecutive columns, stored lane-interleaved
// (lane l owns entries 4 l .. 4 l + 3 of the trip's 256) so one 16-B-per-lane load fetches a lane's 4 edges.
static void sell_order(const int32_t* ptr, int32_t n, std::vector<int32_t>& order) {
  int32_t maxdeg = 0;
  for (int32_t i = 0; i < n; ++i) maxdeg = std::max(maxdeg, ptr[i + 1] - ptr[i]);
  std::vector<int32_t> start((size_t)maxdeg + 2, 0);
  for (int32_t i = 0; i < n; ++i) start[(size_t)(maxdeg - (ptr[i + 1] - ptr[i])) + 1]++;
  for (int32_t d = 0; d <= maxdeg; ++d) start[(size_t)d + 1] += start[(size_t)d];
  order.assign((size_t)n, 0);
  for (int32_t i = 0; i < n; ++i) order[(size_t)start[(size_t)(maxdeg - (ptr[i + 1] - ptr[i]))]++] = i;
}

extern "C" int msgat_graph_sell_count(const int32_t* ptr, int32_t n, int32_t* n_slices_out, int32_t* n_pos_out,
                                      int32_t* pair_trips_out) {
  if (!ptr || !n_slices_out || !n_pos_out || !pair_trips_out) return MSGAT_ERR_NULL;
  if (n <= 0) return MSGAT_ERR_SHAPE;
  for (int32_t i = 0; i < n; ++i)
    if (ptr[i + 1] < ptr[i]) return MSGAT_ERR_GRAPH;
  std::vector<int32_t> order;
  sell_order(ptr, n, order);
  const int32_t ns = (n + 63) / 64;
  int64_t total = 0;
  std::vector<int32_t> trips((size_t)ns, 0);
  for (int32_t s = 0; s < ns; ++s) {
    const int32_t first = order[(size_t)64 * s];
    trips[(size_t)s] = (ptr[first + 1] - ptr[first] + 3) / 4;
    total += (int64_t)256 * trips[(size_t)s];
  }
  if (total > INT32_MAX - MSGAT_SELL_SLACK) return MSGAT_ERR_SHAPE;
  int32_t pair = 0;
  for (int32_t i = 0; 2 * i < ns; ++i)
    pair = std::max(pair, trips[(size_t)i] + (ns - 1 - i > i ? trips[(size_t)(ns - 1 - i)] : 0));
  *n_slices_out = ns;
  *n_pos_out = (int32_t)total;
  *pair_trips_out = pair;
  return MSGAT_OK;
}

extern "C" int msgat_graph_sell_build(const int32_t* ptr, const int32_t* idx, const int32_t* perm, int32_t n,
                                      int32_t nnz, int32_t n_slices, int32_t n_pos, int32_t* slice_off,
                                      int32_t* lane_row, uint16_t* sidx, int32_t* ssrc, int32_t* spos) {
  if (!ptr || !slice_off || !lane_row || !sidx) return MSGAT_ERR_NULL;
  if (n_pos > 0 && !ssrc) return MSGAT_ERR_NULL;
  if (nnz > 0 && !idx) return MSGAT_ERR_NULL;
  if (n <= 0 || n > 65535 || nnz < 0 || n_slices != (n + 63) / 64 || ptr[n] != nnz || n_pos < nnz) return MSGAT_ERR_SHAPE;
  std::vector<int32_t> order;
  sell_order(ptr, n, order);
  int64_t off = 0;
  for (int32_t s = 0; s < n_slices; ++s) {
    const int32_t first = order[(size_t)64 * s];
    const int32_t width = (ptr[first + 1] - ptr[first] + 3) / 4 * 4;
    if (off + (int64_t)64 * width > n_pos) return MSGAT_ERR_SHAPE;  // caller's count is stale
    slice_off[s] = (int32_t)off;
    for (int32_t l = 0; l < 64; ++l) {
      const int32_t i = 64 * s + l;
      const int32_t row = i < n ? order[(size_t)i] : -1;
      lane_row[i] = row;
      const int32_t deg = row >= 0 ? ptr[row + 1] - ptr[row] : 0;
      for (int32_t k = 0; k < width; ++k) {
        const int32_t p = (int32_t)off + 256 * (k >> 2) + 4 * l + (k & 3);
        if (k < deg) {
          const int32_t e = ptr[row] + k;
          const int32_t id = perm ? perm[e] : e;
          sidx[p] = (uint16_t)idx[e];
          ssrc[p] = id;
          if (spos) spos[id] = p;
        } else {
          sidx[p] = 0;  // padding: coefficient 0 times node 0
          ssrc[p] = -1;
        }
      }
    }
    off += (int64_t)64 * width;
  }
  if (off != n_pos) return MSGAT_ERR_SHAPE;
  slice_off[n_slices] = n_pos;
  for (int32_t p = n_pos; p < n_pos + MSGAT_SELL_SLACK; ++p) sidx[p] = 0;
  return MSGAT_OK;
}
