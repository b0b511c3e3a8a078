// Host side of the boundary: dense [N,N] adjacency -> CSR + CSC.
//
// The reference keeps the adjacency dense (built at data_loader.py:59-66, held as a
// frozen parameter at msgat.py:190) and applies it as a mask `att * adjacency`
// (attention.py:36).  Only its non-zeros matter to that product, so the kernels walk
// them: CSR for the forward gather, CSC (+ the CSC->CSR permutation) for the transposed
// gather of the backward pass.
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
  // JDS forms: every position belongs to exactly one CSR edge, owned by the lane whose row the edge starts from
  // (rows form) or ends in (columns form), and holds the node at the edge's other end
  for (int form = 0; form < 2; ++form) {
    const msgat_jds_t& j = form == 0 ? g->jds_rows : g->jds_cols;
    if (j.n_slices == 0) continue;
    if (j.n_slices != (n + 63) / 64 || j.n_cols < 0 || !j.slice || !j.colstart || !j.lane_row) return MSGAT_ERR_GRAPH;
    if (nnz > 0 && (!j.idx || !j.src)) return MSGAT_ERR_GRAPH;
    if (j.slice[0] != 0 || j.slice[j.n_slices] != j.n_cols || j.colstart[0] != 0) return MSGAT_ERR_GRAPH;
    for (int32_t c = j.n_cols; c <= j.n_cols + MSGAT_JDS_PAD; ++c)
      if (j.colstart[c] != nnz) return MSGAT_ERR_GRAPH;
    std::vector<char> used((size_t)nnz, 0);
    for (int32_t s = 0; s < j.n_slices; ++s) {
      if (j.slice[s + 1] < j.slice[s]) return MSGAT_ERR_GRAPH;
      int32_t prev = 64;
      for (int32_t c = j.slice[s]; c < j.slice[s + 1]; ++c) {
        const int32_t cnt = j.colstart[c + 1] - j.colstart[c];
        if (cnt <= 0 || cnt > prev) return MSGAT_ERR_GRAPH;  // active lanes form a shrinking prefix
        prev = cnt;
        for (int32_t l = 0; l < cnt; ++l) {
          const int32_t p = j.colstart[c] + l, e = j.src[p], owner = j.lane_row[64 * s + l];
          if (e < 0 || e >= nnz || used[(size_t)e] || owner < 0 || owner >= n) return MSGAT_ERR_GRAPH;
          used[(size_t)e] = 1;
          const int32_t from = form == 0 ? g->erow[e] : g->col[e], to = form == 0 ? g->col[e] : g->erow[e];
          if (owner != from || j.idx[p] != to) return MSGAT_ERR_GRAPH;
          if (j.pos && j.pos[e] != p) return MSGAT_ERR_GRAPH;
        }
      }
    }
    for (int32_t e = 0; e < nnz; ++e)
      if (!used[(size_t)e]) return MSGAT_ERR_GRAPH;
  }
  return MSGAT_OK;
}

// ---- sliced jagged-diagonal (JDS) layout of one sparse structure ---------------------------------------------
// For graphs whose [N,T] slab exceeds LDS the aggregate keeps ONE 4-timestep column of a slab in LDS per pass,
// so the edge lists are re-read once per (group, channel, column) -- 8x more bytes than the features themselves
// at the N = 8192 / degree 16 stress graph.  Read through the CSR (lane = row, a window of the row's edges per
// trip) those reads are dependent round trips of 16-B pieces at a ~68-B stride.  The JDS form makes them
// contiguous and address-independent: rows are cut into slices of 64 (one wavefront), sorted by degree inside
// the slice (descending, stable), and the k-th edges of all rows of a slice that have one are stored back to
// back ("jagged column" k).  Lane l of the wave then reads entry colstart[k] + l for l < count_k: one coalesced
// 4-B-per-lane load per array and k, every byte read once, and all addresses known up front.
//
//   slice[s] .. slice[s+1]        jagged columns of slice s (width = its largest degree)
//   colstart[c] .. colstart[c+1]  positions of jagged column c
//   lane_row[64 s + l]            row handled by lane l of slice s (-1 past the last row)
//   idx[p]                        neighbour of position p;  src[p] = edge id (in the caller's edge numbering,
//                                 through `perm` when given);  pos[e] = inverse of src (optional)
extern "C" int msgat_graph_jds_count(const int32_t* ptr, int32_t n, int32_t* n_slices_out, int32_t* n_cols_out) {
  if (!ptr || !n_slices_out || !n_cols_out) return MSGAT_ERR_NULL;
  if (n <= 0) return MSGAT_ERR_SHAPE;
  const int32_t ns = (n + 63) / 64;
  int64_t cols = 0;
  for (int32_t s = 0; s < ns; ++s) {
    int32_t w = 0;
    for (int32_t i = 64 * s; i < n && i < 64 * s + 64; ++i) {
      const int32_t d = ptr[i + 1] - ptr[i];
      if (d < 0) return MSGAT_ERR_GRAPH;
      if (d > w) w = d;
    }
    cols += w;
  }
  if (cols > INT32_MAX) return MSGAT_ERR_SHAPE;
  *n_slices_out = ns;
  *n_cols_out = (int32_t)cols;
  return MSGAT_OK;
}

extern "C" int msgat_graph_jds_build(const int32_t* ptr, const int32_t* idx, const int32_t* perm, int32_t n,
                                     int32_t nnz, int32_t n_slices, int32_t n_cols, int32_t* slice,
                                     int32_t* colstart, int32_t* lane_row, int32_t* jidx, int32_t* jsrc,
                                     int32_t* jpos) {
  if (!ptr || !slice || !colstart || !lane_row) return MSGAT_ERR_NULL;
  if (nnz > 0 && (!idx || !jidx || !jsrc)) return MSGAT_ERR_NULL;
  if (n <= 0 || nnz < 0 || n_slices != (n + 63) / 64 || ptr[n] != nnz) return MSGAT_ERR_SHAPE;
  int32_t c = 0, p = 0;
  for (int32_t s = 0; s < n_slices; ++s) {
    const int32_t r0 = 64 * s, rows = (n - r0 < 64) ? n - r0 : 64;
    int32_t order[64], deg[64];
    for (int32_t i = 0; i < rows; ++i) { order[i] = i; deg[i] = ptr[r0 + i + 1] - ptr[r0 + i]; }
    // stable insertion sort by degree, descending: the active lanes of every jagged column are a prefix
    for (int32_t i = 1; i < rows; ++i) {
      const int32_t o = order[i];
      int32_t j = i;
      while (j > 0 && deg[order[j - 1]] < deg[o]) { order[j] = order[j - 1]; --j; }
      order[j] = o;
    }
    for (int32_t l = 0; l < 64; ++l) lane_row[64 * s + l] = (l < rows) ? r0 + order[l] : -1;
    slice[s] = c;
    const int32_t width = rows > 0 ? deg[order[0]] : 0;
    for (int32_t k = 0; k < width; ++k) {
      if (c >= n_cols) return MSGAT_ERR_SHAPE;
      colstart[c++] = p;
      for (int32_t l = 0; l < rows && deg[order[l]] > k; ++l) {
        const int32_t e = ptr[r0 + order[l]] + k;
        if (p >= nnz) return MSGAT_ERR_SHAPE;
        jidx[p] = idx[e];
        const int32_t id = perm ? perm[e] : e;
        jsrc[p] = id;
        if (jpos) jpos[id] = p;
        ++p;
      }
    }
  }
  if (c != n_cols || p != nnz) return MSGAT_ERR_SHAPE;
  slice[n_slices] = c;
  for (int32_t i = 0; i <= MSGAT_JDS_PAD; ++i) colstart[c + i] = p;  // end marker + padding, all = nnz
  return MSGAT_OK;
}
