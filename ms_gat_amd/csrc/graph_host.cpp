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
  return MSGAT_OK;
}
