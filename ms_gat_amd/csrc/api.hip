// extern "C" entry points of libmsgat_hip.so (see include/msgat_hip.h).
// Argument checks happen on the host before anything is enqueued: a bad shape must come
// back as a status code, never as a faulting kernel.
#include "common.hpp"
#include <cstdio>

using namespace msgat;

namespace {

int check_shape(const msgat_shape_t* sh) {
  if (!sh) return MSGAT_ERR_NULL;
  if (sh->R <= 0 || sh->Bg <= 0 || sh->C <= 0 || sh->Co < 0 || sh->N <= 0 || sh->T <= 0) return MSGAT_ERR_SHAPE;
  if (!t_supported(sh->T)) return MSGAT_ERR_UNSUPPORTED;
  if (sh->C > kMaxC || sh->Co > kMaxC) return MSGAT_ERR_UNSUPPORTED;
  if ((int64_t)sh->R * sh->Bg > 65535) return MSGAT_ERR_UNSUPPORTED;  // groups ride on gridDim.y/z
  if ((int64_t)sh->N * sh->T > (1 << 28)) return MSGAT_ERR_UNSUPPORTED;
  // one group's [C,N,T] block is indexed with 32-bit element offsets
  if ((int64_t)(sh->C > sh->Co ? sh->C : sh->Co + 1) * sh->N * sh->T >= (1ll << 31)) return MSGAT_ERR_UNSUPPORTED;
  return MSGAT_OK;
}

int check_graph(const msgat_shape_t* sh, const msgat_graph_t* gr) {
  if (!gr) return MSGAT_ERR_NULL;
  if (gr->n_nodes != sh->N || gr->nnz < 0) return MSGAT_ERR_SHAPE;
  if (!gr->rowptr || !gr->colptr) return MSGAT_ERR_NULL;
  if (gr->nnz > 0 && (!gr->col || !gr->val || !gr->erow || !gr->crow || !gr->cperm || !gr->cpos)) return MSGAT_ERR_NULL;
  for (const msgat_sell_t* j : {&gr->sell_rows, &gr->sell_cols}) {
    if (j->n_slices == 0) continue;
    if (j->n_slices != cdiv(sh->N, 64) || j->n_pos < gr->nnz) return MSGAT_ERR_GRAPH;
    if (!j->slice_off || !j->lane_row) return MSGAT_ERR_NULL;
    if (gr->nnz > 0 && (!j->idx || !j->src)) return MSGAT_ERR_NULL;
  }
  if (gr->sell_rows.n_slices != 0 && gr->nnz > 0 && !gr->sell_rows.pos) return MSGAT_ERR_NULL;
  return MSGAT_OK;
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// backward workspace layout, shared by the size query and the run
struct BwdPlan {
  int mode, G, Cu, nch;
  size_t off_dEp, off_gE, off_Ec, off_delta, off_dkW, off_dq, off_dv, off_dwg, off_cp, off_cp2, off_dense, total;
};

// the graph decides the edge layout (CSR, or SELL for large N) and with it the size of the per-edge buffers
BwdPlan plan_bwd(const msgat_shape_t& sh, const msgat_graph_t& gr) {
  BwdPlan p{};
  const int nnz = gr.nnz;
  p.mode = msgat_gacn_mode(sh.C, sh.Co);
  p.G = sh.R * sh.Bg;
  p.Cu = (p.mode == MSGAT_MODE_PROJ_FIRST) ? sh.Co : sh.C;
  const bool sell_r = sell_usable(gr.sell_rows, nnz, sh.N, sh.T), sell_c = sell_usable(gr.sell_cols, nnz, sh.N, sh.T);
  p.nch = sddmm_chunks(p.G, p.Cu, sh.N, sh.T, sell_r ? &gr.sell_rows : nullptr);
  const size_t G = p.G, N = sh.N, T = sh.T, P = N * T;
  size_t off = 0;
  auto take = [&](size_t floats) {
    const size_t o = off;
    off += align256(floats * sizeof(float));
    return o;
  };
  p.off_dEp = take(sell_r ? G * p.nch * (size_t)gr.sell_rows.n_pos + MSGAT_SELL_SLACK : G * p.nch * (size_t)nnz);
  p.off_gE = take(G * (size_t)nnz);
  p.off_Ec = take(sell_c ? G * (size_t)gr.sell_cols.n_pos + MSGAT_SELL_SLACK : G * (size_t)nnz);
  p.off_delta = take(G * N);
  p.off_dkW = take(G * P);
  p.off_dq = take(G * P);
  // AGG_FIRST: dy = W^T dz [G,C,P];  PROJ_FIRST: du = E^T dz [G,Co,P];  PLAIN: none
  p.off_dv = take(p.mode == MSGAT_MODE_PLAIN ? 0 : G * (size_t)p.Cu * P);
  p.off_dwg = take(dwg_partial_floats(p.G, sh.N, sh.T));
  // partials of the parameter gradients; cp2: dalpha of AGG_FIRST, whose dW partials (cp) wait for the same
  // end-of-pass reduction
  size_t cp = 0, cp2 = 0;
  if (p.mode == MSGAT_MODE_PLAIN) {
    cp = chanpair_partial_floats(p.G, sh.Bg, 1, sh.C);
    if (G * (size_t)sh.C * kAggMaxSplit > cp) cp = G * (size_t)sh.C * kAggMaxSplit;
  } else if (p.mode == MSGAT_MODE_AGG_FIRST) {
    cp = chanpair_partial_floats(p.G, sh.Bg, sh.Co, sh.C);
    const size_t fused = G * (size_t)aggfirst_blocks((int)P) * sh.Co * sh.C;
    if (sh.C <= kAggFirstMaxC && fused > cp) cp = fused;
    cp2 = chanpair_partial_floats(p.G, sh.Bg, 1, sh.C);
    if (G * (size_t)sh.C * kAggMaxSplit > cp2) cp2 = G * (size_t)sh.C * kAggMaxSplit;
  } else {
    cp = chanpair_partial_floats(p.G, sh.Bg, sh.Co + 1, sh.C);
  }
  p.off_cp = take(cp);
  p.off_cp2 = take(cp2);
  p.off_dense = take((dense_scratch_bytes(p.G, sh.N, sh.T) + 3) / 4);   // operand images of the dense column pass (large N)
  p.total = off;
  return p;
}

}  // namespace

extern "C" int msgat_abi_version(void) { return MSGAT_ABI_VERSION; }

extern "C" const char* msgat_status_string(int status) {
  switch (status) {
    case MSGAT_OK: return "ok";
    case MSGAT_ERR_NULL: return "required pointer is NULL";
    case MSGAT_ERR_SHAPE: return "bad or inconsistent dimension";
    case MSGAT_ERR_UNSUPPORTED: return "unsupported size (T must be 4/8/12/16, channels <= 256)";
    case MSGAT_ERR_WORKSPACE: return "workspace too small";
    case MSGAT_ERR_GRAPH: return "malformed CSR/CSC graph";
  }
  if (status <= MSGAT_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(MSGAT_ERR_HIP_BASE - status));
  return "unknown status";
}

extern "C" int msgat_gacn_mode(int32_t C, int32_t Co) {
  if (Co <= 0) return MSGAT_MODE_PLAIN;
  return (C > Co) ? MSGAT_MODE_PROJ_FIRST : MSGAT_MODE_AGG_FIRST;
}

// ---- stages -------------------------------------------------------------------------------------
extern "C" int msgat_stage_project(const msgat_shape_t* sh, const float* x, const float* alpha,
                                   const float* W, float* q, float* u, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  if (!x || !alpha || !q) return MSGAT_ERR_NULL;
  const int G = sh->R * sh->Bg, P = sh->N * sh->T;
  hipStream_t s = (hipStream_t)stream;
  if (u == nullptr) return launch_qonly(x, alpha, q, G, sh->Bg, sh->C, P, s);
  if (!W || sh->Co <= 0) return MSGAT_ERR_NULL;
  return launch_project(x, W, 0, alpha, nullptr, nullptr, u, q, G, sh->Bg, sh->C, sh->Co, P, s);
}

extern "C" int msgat_stage_scores(const msgat_shape_t* sh, const msgat_graph_t* gr, const float* q,
                                  const float* Wg, float* kW, float* lse, float* pq, float* E, float* Ec,
                                  void* dense_scratch, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!q || !Wg || !kW || !lse) return MSGAT_ERR_NULL;
  if (gr->nnz > 0 && !E) return MSGAT_ERR_NULL;
  return launch_scores(*gr, q, Wg, kW, lse, pq, E, Ec, sh->R * sh->Bg, sh->Bg, sh->N, sh->T, (hipStream_t)stream, nullptr, nullptr, 0,
                       nullptr, nullptr, 0, nullptr, nullptr, dense_scratch);
}

// forward aggregate over the CSR: on the SELL layout when the graph carries a usable one (E re-ordered into
// `scratch` first), on the CSR otherwise
static int aggregate_rows(const msgat_shape_t* sh, const msgat_graph_t* gr, int Cu, const float* u, const float* E,
                          float* v, float* scratch, hipStream_t s) {
  const int G = sh->R * sh->Bg;
  if (sell_usable(gr->sell_rows, gr->nnz, sh->N, sh->T)) {
    if (!scratch) return MSGAT_ERR_WORKSPACE;
    int st = launch_permute_edges(E, gr->sell_rows.src, scratch, G, gr->nnz, gr->sell_rows.n_pos, s);
    if (st) return st;
    return launch_aggregate(gr->rowptr, gr->col, gr->nnz, &gr->sell_rows, u, scratch, nullptr, nullptr, v, G, sh->Bg,
                            Cu, sh->N, sh->T, s);
  }
  return launch_aggregate(gr->rowptr, gr->col, gr->nnz, nullptr, u, E, nullptr, nullptr, v, G, sh->Bg, Cu, sh->N,
                          sh->T, s);
}

// transposed aggregate over the CSC (backward): E goes to CSC order -- or to the SELL order of the CSC -- in Ec
// (Ec_ready: E already in CSC order, as the forward's score kernel left it -- used when the CSC itself is walked)
static int aggregate_cols(const msgat_shape_t* sh, const msgat_graph_t* gr, int Cu, const float* dv, const float* E,
                          float* Ec, const float* addvec, const float* extra, float* out, hipStream_t s,
                          const float* xdot = nullptr, float* dap = nullptr, int* dot_done = nullptr,
                          const float* Ec_ready = nullptr) {
  const int G = sh->R * sh->Bg;
  const bool sell = sell_usable(gr->sell_cols, gr->nnz, sh->N, sh->T);
  const float* Eo = Ec;
  if (!sell && Ec_ready != nullptr) {
    Eo = Ec_ready;
  } else {
    int st = launch_permute_edges(E, sell ? gr->sell_cols.src : gr->cperm, Ec, G, gr->nnz,
                                  sell ? gr->sell_cols.n_pos : gr->nnz, s);
    if (st) return st;
  }
  return launch_aggregate(gr->colptr, gr->crow, gr->nnz, sell ? &gr->sell_cols : nullptr, dv, Eo, addvec, extra, out,
                          G, sh->Bg, Cu, sh->N, sh->T, s, xdot, dap, dot_done);
}

extern "C" size_t msgat_dense_scratch_bytes(const msgat_shape_t* sh) {
  if (check_shape(sh) != MSGAT_OK) return 0;
  return dense_scratch_bytes(sh->R * sh->Bg, sh->N, sh->T);
}

extern "C" size_t msgat_edge_scratch_floats(const msgat_shape_t* sh, const msgat_graph_t* gr) {
  if (check_shape(sh) != MSGAT_OK || check_graph(sh, gr) != MSGAT_OK) return 0;
  if (!sell_usable(gr->sell_rows, gr->nnz, sh->N, sh->T)) return 0;
  return (size_t)sh->R * sh->Bg * gr->sell_rows.n_pos + MSGAT_SELL_SLACK;
}

extern "C" int msgat_stage_aggregate(const msgat_shape_t* sh, const msgat_graph_t* gr, int32_t Cu,
                                     const float* u, const float* E, float* v, float* edge_scratch,
                                     void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!u || !v || (gr->nnz > 0 && !E)) return MSGAT_ERR_NULL;
  if (Cu <= 0 || Cu > kMaxC) return MSGAT_ERR_SHAPE;
  return aggregate_rows(sh, gr, Cu, u, E, v, edge_scratch, (hipStream_t)stream);
}

extern "C" int msgat_stage_aggregate_project(const msgat_shape_t* sh, const msgat_graph_t* gr,
                                             const float* x, const float* E, const float* W, float* y,
                                             float* z, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!x || !W || !z || (gr->nnz > 0 && !E)) return MSGAT_ERR_NULL;
  if (sh->Co <= 0) return MSGAT_ERR_SHAPE;
  return launch_aggregate_project(*gr, x, E, W, y, z, sh->R * sh->Bg, sh->Bg, sh->C, sh->Co, sh->N, sh->T,
                                  (hipStream_t)stream);
}

extern "C" int msgat_stage_mix(const msgat_shape_t* sh, int32_t Ci, int32_t Co, const float* in,
                               const float* M, int32_t m_in_major, const float* addvec, const float* extra,
                               float* out, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  if (!in || !M || !out || (addvec && !extra)) return MSGAT_ERR_NULL;
  if (Ci <= 0 || Co <= 0 || Ci > kMaxC || Co > kMaxC) return MSGAT_ERR_SHAPE;
  return launch_project(in, M, m_in_major, nullptr, addvec, extra, out, nullptr, sh->R * sh->Bg, sh->Bg, Ci, Co,
                        sh->N * sh->T, (hipStream_t)stream);
}

extern "C" size_t msgat_contract_partial_floats(const msgat_shape_t* sh, int32_t Ca, int32_t Cb) {
  if (check_shape(sh) != MSGAT_OK || Ca <= 0 || Cb <= 0) return 0;
  return chanpair_partial_floats(sh->R * sh->Bg, sh->Bg, Ca, Cb);
}

extern "C" int msgat_stage_contract(const msgat_shape_t* sh, int32_t Ca, int32_t Cb, const float* A,
                                    const float* Aextra, const float* B, float* partials, float* dst0,
                                    int32_t n0, float* dst1, int32_t n1, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  if (!B || !partials || (!A && !(Aextra && Ca == 1))) return MSGAT_ERR_NULL;
  if (Ca <= 0 || Cb <= 0 || Ca > kMaxC + 1 || Cb > kMaxC || n0 < 0 || n1 < 0 || n0 + n1 > Ca * Cb) return MSGAT_ERR_SHAPE;
  return launch_chanpair(A, Aextra, B, partials, dst0, n0, dst1, n1, sh->R * sh->Bg, sh->Bg, Ca, Cb,
                         sh->N * sh->T, (hipStream_t)stream);
}

extern "C" int msgat_stage_project_backward(const msgat_shape_t* sh, const float* du, const float* dq, const float* x,
                                            const float* W, const float* alpha, float* partials, float* dW,
                                            float* dalpha, float* dx, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  if (!du || !dq || !x || !W || !alpha || !partials || !dW || !dalpha || !dx) return MSGAT_ERR_NULL;
  if (sh->Co <= 0) return MSGAT_ERR_SHAPE;
  const int G = sh->R * sh->Bg, Bg = sh->Bg, C = sh->C, Co = sh->Co, P = sh->N * sh->T;
  hipStream_t s = (hipStream_t)stream;
  ReduceJobs jobs{};
  int both = 0;
  st = launch_chanpair_mix(du, dq, x, W, alpha, dx, partials, dW, dalpha, G, Bg, Co, C, P, s, &jobs, &both);
  if (st) return st;
  if (!both) {
    st = launch_chanpair(du, dq, x, partials, dW, Co * C, dalpha, C, G, Bg, Co + 1, C, P, s, &jobs);
    if (st) return st;
    st = launch_project(du, W, 1, nullptr, alpha, dq, dx, nullptr, G, Bg, Co, C, P, s);
    if (st) return st;
  }
  return launch_reduce_jobs(jobs, s);
}

// ---- temporal / channel branches -------------------------------------------------------------------------
extern "C" int msgat_stage_mix_epilogue(const msgat_shape_t* sh, int32_t Ci, int32_t Co, const float* in,
                                        const float* M, int32_t m_in_major, const float* bias,
                                        int32_t bias_per_relation, const float* add, int32_t relu, float* out,
                                        void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  if (!in || !M || !out) return MSGAT_ERR_NULL;
  if (Ci <= 0 || Co <= 0 || Ci > kMaxC || Co > kMaxC) return MSGAT_ERR_SHAPE;
  MixEpilogue epi;
  epi.bias = bias;
  epi.bias_rstride = bias_per_relation ? Co : 0;
  if (add != nullptr) epi.add = seg_single(add, Co);
  epi.relu = relu;
  return launch_project_epi(in, M, m_in_major, nullptr, nullptr, nullptr, out, nullptr, sh->R * sh->Bg, sh->Bg, Ci,
                            Co, sh->N * sh->T, epi, (hipStream_t)stream);
}

static int check_rgnt(int32_t R, int32_t Bg, int32_t N, int32_t T);
static int check_rows(int32_t G, int32_t Co, int32_t K, int32_t N, int32_t T) {
  if (G <= 0 || Co <= 0 || N <= 0) return MSGAT_ERR_SHAPE;
  if (!t_supported(T) || K < 1 || K > 2) return MSGAT_ERR_UNSUPPORTED;
  if ((int64_t)G * K * Co * N * T >= ((int64_t)1 << 40) || (int64_t)Co * N > 0x7fffffff || G > 65535) return MSGAT_ERR_SHAPE;
  return MSGAT_OK;
}

extern "C" int msgat_time_mix(const float* src, const float* A, int32_t a_per_group, const float* bias,
                              float* dst, int32_t G, int32_t Co, int32_t K, int32_t N, int32_t T,
                              int32_t backward, int32_t R, int32_t src_group_stride, void* stream) {
  if (!src || !A || !dst) return MSGAT_ERR_NULL;
  int st = check_rows(G, Co, K, N, T);
  if (st) return st;
  if (R <= 0 || G % R) return MSGAT_ERR_SHAPE;
  if (src_group_stride != 0 && (!backward || src_group_stride < Co || src_group_stride > 4 * kMaxC)) return MSGAT_ERR_SHAPE;
  return launch_tmix(src, A, a_per_group, backward ? nullptr : bias, dst, G, Co, K, N, T, backward, R,
                     (hipStream_t)stream, src_group_stride);
}

extern "C" int msgat_causal_conv_fused(int32_t Ci, int32_t Co) {
  return (Ci > 0 && Co > 0 && project_taps_supported(Ci, Co) && project_taps_supported(Co, Ci)) ? 1 : 0;
}

extern "C" int msgat_causal_conv(const float* src, const float* taps, const float* bias, int32_t bias_per_relation,
                                 float* dst, int32_t R, int32_t Bg, int32_t Ci, int32_t Co, int32_t N, int32_t T,
                                 int32_t dilation, int32_t backward, int32_t src_group_stride, void* stream) {
  if (!src || !taps || !dst) return MSGAT_ERR_NULL;
  int st = check_rgnt(R, Bg, N, T);
  if (st) return st;
  if (Ci <= 0 || Co <= 0 || Ci > kMaxC || Co > kMaxC || dilation <= 0) return MSGAT_ERR_SHAPE;
  const int Cs = backward ? Co : Ci;                 // channels of src
  if (src_group_stride != 0 && (src_group_stride < Cs || src_group_stride > 4 * kMaxC)) return MSGAT_ERR_SHAPE;
  if (!msgat_causal_conv_fused(Ci, Co)) return MSGAT_ERR_UNSUPPORTED;
  const int G = R * Bg, P = N * T, d = dilation < T ? dilation : T;   // a dilation of T or more: the shifted tap sees nothing
  hipStream_t s = (hipStream_t)stream;
  if (!backward)
    return launch_project_taps(src, src_group_stride, taps, 0, bias, bias_per_relation ? Co : 0, dst, G, Bg, Ci, Co, P, T, -d, s);
  return launch_project_taps(src, src_group_stride, taps, 1, nullptr, 0, dst, G, Bg, Co, Ci, P, T, d, s);
}

extern "C" int msgat_causal_conv_grad_weight(const float* dout, int32_t dout_group_stride, const float* h, float* partials,
                                             float* dtaps, int32_t R, int32_t Bg, int32_t Ci, int32_t Co, int32_t N,
                                             int32_t T, int32_t dilation, int32_t with_ones, void* stream) {
  if (!dout || !h || !partials || !dtaps) return MSGAT_ERR_NULL;
  int st = check_rgnt(R, Bg, N, T);
  if (st) return st;
  if (Ci <= 0 || Co <= 0 || Ci > kMaxC || 2 * Co > kMaxC || dilation <= 0 || with_ones < 0 || with_ones > 2)
    return MSGAT_ERR_SHAPE;
  if (dout_group_stride != 0 && (dout_group_stride < Co || dout_group_stride > 4 * kMaxC)) return MSGAT_ERR_SHAPE;
  SegList A = seg_single(dout, Co);
  if (dout_group_stride > Co) A.gstride[0] = dout_group_stride;
  const int apart = with_ones == 2;
  with_ones = with_ones != 0;
  const int Cbx = Ci + with_ones;
  hipStream_t s = (hipStream_t)stream;
  int nblk = 0;
  st = launch_chanpair_shifted(A, h, partials, R, Bg, Cbx, N * T, chanpair_mfma_blocks(R), with_ones, dilation, T, s, &nblk);
  if (st) return st;
  if (apart) return launch_reduce_lastcol(partials, R, nblk, 2 * Co, Cbx, dtaps, s);
  return launch_reduce_groups(partials, R, nblk, 2 * Co * Cbx, dtaps, s);
}

extern "C" size_t msgat_time_mix_partial_floats(int32_t G, int32_t K, int32_t T) {
  if (G <= 0 || K < 1 || K > 2 || !t_supported(T)) return 0;
  return tmix_partial_floats(G, K, T);
}

extern "C" int msgat_time_mix_grad_matrix(const float* dout, const float* y, float* dA, float* partials,
                                          int32_t G, int32_t Co, int32_t K, int32_t N, int32_t T,
                                          int32_t dout_group_stride, void* stream) {
  if (!dout || !y || !dA || !partials) return MSGAT_ERR_NULL;
  int st = check_rows(G, Co, K, N, T);
  if (st) return st;
  if (dout_group_stride != 0 && (dout_group_stride < Co || dout_group_stride > 4 * kMaxC)) return MSGAT_ERR_SHAPE;
  return launch_tmix_dA(dout, y, dA, partials, G, Co, K, N, T, (hipStream_t)stream, dout_group_stride);
}

extern "C" int msgat_node_pool(const float* x, const float* w, float* pooled, int64_t slabs, int32_t N, int32_t T,
                               int32_t R, int32_t channels, int32_t group_stride, void* stream) {
  if (!x || !w || !pooled) return MSGAT_ERR_NULL;
  if (slabs <= 0 || slabs > 0x7fffffff || N <= 0 || R <= 0 || slabs % R) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  if (channels < 0 || (channels > 0 && (slabs % channels || group_stride < channels))) return MSGAT_ERR_SHAPE;
  return launch_node_pool(x, w, pooled, slabs, N, T, R, (hipStream_t)stream, channels, group_stride);
}

extern "C" int msgat_node_pool_grad_signal(const float* w, const float* dpooled, const float* dx_add, float* dx,
                                           int64_t slabs, int32_t N, int32_t T, int32_t R, void* stream) {
  if (!w || !dpooled || !dx) return MSGAT_ERR_NULL;
  if (slabs <= 0 || slabs > 0x7fffffff || N <= 0 || R <= 0 || slabs % R) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  return launch_node_pool_dx(w, dpooled, dx_add, dx, slabs, N, T, R, (hipStream_t)stream);
}

extern "C" size_t msgat_node_pool_partial_floats(int32_t G, int32_t C, int32_t N) {
  if (G <= 0 || C <= 0 || N <= 0) return 0;
  return node_pool_partial_floats(G, C, N);
}

extern "C" int msgat_node_pool_grad_weight(const float* x, const float* dpooled, float* dw, float* partials,
                                           int32_t G, int32_t C, int32_t N, int32_t T, int32_t R, void* stream) {
  if (!x || !dpooled || !dw || !partials) return MSGAT_ERR_NULL;
  if (G <= 0 || G > 65535 || C <= 0 || N <= 0 || R <= 0 || G % R) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  return launch_node_pool_dw(x, dpooled, dw, partials, G, C, N, T, R, (hipStream_t)stream);
}

// ---- segment lists ----------------------------------------------------------------------------------------
static int to_seglist(const msgat_seg_t* segs, int32_t n, SegList* out) {
  SegList s{};
  if (n < 0 || n > kMaxSeg) return MSGAT_ERR_SHAPE;
  if (n > 0 && !segs) return MSGAT_ERR_NULL;
  int c = 0;
  for (int i = 0; i < n; ++i) {
    if (!segs[i].ptr) return MSGAT_ERR_NULL;
    if (segs[i].channels <= 0 || (segs[i].group_stride != 0 && segs[i].group_stride < segs[i].channels)) return MSGAT_ERR_SHAPE;
    s.ptr[i] = segs[i].ptr;
    s.begin[i] = c;
    s.gstride[i] = segs[i].group_stride ? segs[i].group_stride : segs[i].channels;
    if (s.gstride[i] > 4 * kMaxC) return MSGAT_ERR_UNSUPPORTED;
    c += segs[i].channels;
  }
  s.begin[n] = c;
  s.n = n;
  if (c > kMaxC) return MSGAT_ERR_UNSUPPORTED;
  *out = s;
  return MSGAT_OK;
}

static int check_rgnt(int32_t R, int32_t Bg, int32_t N, int32_t T) {
  if (R <= 0 || Bg <= 0 || N <= 0) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  if ((int64_t)R * Bg > 65535 || (int64_t)4 * kMaxC * N * T >= (1ll << 31)) return MSGAT_ERR_UNSUPPORTED;
  return MSGAT_OK;
}

extern "C" int msgat_mix_segments(int32_t R, int32_t Bg, int32_t N, int32_t T, const msgat_seg_t* in, int32_t n_in,
                                  const float* M, int32_t m_in_major, const float* bias, int32_t bias_per_relation,
                                  const msgat_seg_t* add, int32_t n_add, int32_t relu, const msgat_seg_t* out,
                                  int32_t n_out, void* stream) {
  int st = check_rgnt(R, Bg, N, T);
  if (st) return st;
  if (!M) return MSGAT_ERR_NULL;
  SegList si, so;
  MixEpilogue epi;
  if ((st = to_seglist(in, n_in, &si)) || (st = to_seglist(out, n_out, &so)) || (st = to_seglist(add, n_add, &epi.add))) return st;
  if (si.n == 0 || so.n == 0) return MSGAT_ERR_SHAPE;
  if (epi.add.n > 0 && epi.add.total() != so.total()) return MSGAT_ERR_SHAPE;
  epi.bias = bias;
  epi.bias_rstride = bias_per_relation ? so.total() : 0;
  epi.relu = relu;
  return launch_project_seg(si, M, m_in_major, nullptr, nullptr, nullptr, so, nullptr, R * Bg, Bg, N * T, epi,
                            (hipStream_t)stream);
}

extern "C" size_t msgat_contract_segments_partial_floats(int32_t R, int32_t Ca, int32_t Cb) {
  if (R <= 0 || Ca <= 0 || Cb <= 0) return 0;
  return chanpair_partial_floats(R, 1, Ca, Cb);
}

extern "C" int msgat_contract_segments(int32_t R, int32_t Bg, int32_t N, int32_t T, const msgat_seg_t* A, int32_t n_a,
                                       const float* B, int32_t Cb, int32_t with_ones, float* partials, float* dst,
                                       void* stream) {
  int st = check_rgnt(R, Bg, N, T);
  if (st) return st;
  if (!B || !partials || !dst) return MSGAT_ERR_NULL;
  if (Cb <= 0 || Cb > kMaxC || with_ones < 0 || with_ones > 2) return MSGAT_ERR_SHAPE;
  SegList sa;
  if ((st = to_seglist(A, n_a, &sa))) return st;
  if (sa.n == 0) return MSGAT_ERR_SHAPE;
  const int ones = with_ones != 0, Cbx = Cb + ones;
  return launch_chanpair_seg(sa, B, partials, dst, sa.total() * Cbx, nullptr, 0, R * Bg, Bg, Cbx, N * T,
                             (hipStream_t)stream, ones, nullptr, with_ones == 2);
}

extern "C" int msgat_contract_form_name(int32_t Ca, int32_t Cb, int32_t with_ones, int32_t n_positions, int32_t with_mix,
                                        char* buf, int32_t buflen) {
  if (!buf || buflen <= 0) return MSGAT_ERR_NULL;
  if (Ca <= 0 || Cb <= 0 || Ca > kMaxC + 2 || Cb > kMaxC || n_positions <= 0 || (with_ones != 0 && with_ones != 1))
    return MSGAT_ERR_SHAPE;
  char name[96];
  int one = 0, nza = 1, nzb = 1;
  const int st = contract_form_name(Ca, Cb + with_ones, with_ones, n_positions, with_mix, name, (int)sizeof name, &one, &nza, &nzb);
  if (st) return st;
  char za[24] = "", zb[24] = "";
  if (nza > 1) snprintf(za, sizeof za, " nza=%d", nza);
  if (nzb > 1) snprintf(zb, sizeof zb, " nzb=%d", nzb);
  snprintf(buf, (size_t)buflen, "%s%s%s%s", name, za, zb, (with_mix && !one) ? " + projection pass" : "");
  return MSGAT_OK;
}

extern "C" size_t msgat_contract_mix_partial_floats(int32_t R, int32_t Bg, int32_t N, int32_t T, int32_t Ca, int32_t Cb) {
  if (R <= 0 || Bg <= 0 || N <= 0 || T <= 0 || Ca <= 0 || Cb <= 0) return 0;
  const size_t general = chanpair_partial_floats(R, 1, Ca, Cb);
  const size_t tiny = Cb <= kAggFirstMaxC ? (size_t)R * Bg * aggfirst_blocks(N * T) * Ca * Cb : 0;
  return general > tiny ? general : tiny;
}

extern "C" int msgat_contract_mix_segments(int32_t R, int32_t Bg, int32_t N, int32_t T, const msgat_seg_t* A, int32_t n_a,
                                           const float* B, int32_t Cb, int32_t with_ones, const float* M, float* partials,
                                           float* dst, float* mixout, void* stream) {
  int st = check_rgnt(R, Bg, N, T);
  if (st) return st;
  if (!B || !partials || !dst || !M || !mixout) return MSGAT_ERR_NULL;
  if (Cb <= 0 || Cb > kMaxC || with_ones < 0 || with_ones > 2) return MSGAT_ERR_SHAPE;
  SegList sa;
  if ((st = to_seglist(A, n_a, &sa))) return st;
  if (sa.n == 0) return MSGAT_ERR_SHAPE;
  const int apart = with_ones == 2;          // the ones column's sums delivered behind the matrix, not inside it
  with_ones = with_ones != 0;
  const int Cbx = Cb + with_ones, Ca = sa.total(), P = N * T;
  hipStream_t s = (hipStream_t)stream;
  // a convolution with 1..3 input channels (the first block of every component: 1 or 3 features): ONE pass over the
  // Ca-channel gradient gives dM (| dbias) and dx (k_aggfirst_bwd) -- the caller sized `partials` with
  // msgat_contract_mix_partial_floats, which knows this form
  if (Cbx <= kAggFirstMaxC && sa.n == 1 && P % 4 == 0)
    return launch_aggfirst_bwd(sa.ptr[0], M, B, mixout, partials, dst, R * Bg, Bg, Cb, Ca, P, s, nullptr, sa.gstride[0],
                               with_ones + apart);
  int nblk = 0, both = 0;
  st = launch_chanpair_mix_wide(sa, B, partials, R, Bg, Cbx, P, chanpair_mfma_blocks(R), with_ones, M, mixout, s, &nblk, &both);
  if (st) return st;
  if (both) return apart ? launch_reduce_lastcol(partials, R, nblk, Ca, Cbx, dst, s) : launch_reduce_groups(partials, R, nblk, Ca * Cbx, dst, s);
  // no fused form for this shape: the two passes
  st = launch_chanpair_seg(sa, B, partials, dst, Ca * Cbx, nullptr, 0, R * Bg, Bg, Cbx, P, s, with_ones, nullptr, apart);
  if (st) return st;
  return launch_project_seg(sa, M, 1, nullptr, nullptr, nullptr, seg_single(mixout, Cb), nullptr, R * Bg, Bg, P, MixEpilogue(), s);
}

// ---- attention core on projected features: backward --------------------------------------------------------
static msgat_shape_t plain_shape(const msgat_shape_t* sh) {
  msgat_shape_t s = *sh;
  s.Co = 0;
  return s;
}

extern "C" size_t msgat_attention_bwd_workspace_bytes(const msgat_shape_t* sh, const msgat_graph_t* gr) {
  if (check_shape(sh) != MSGAT_OK || check_graph(sh, gr) != MSGAT_OK) return 0;
  return plan_bwd(plain_shape(sh), *gr).total;
}

extern "C" int msgat_attention_bwd_accepts_strided_dv(const msgat_shape_t* sh, const msgat_graph_t* gr) {
  if (check_shape(sh) != MSGAT_OK || check_graph(sh, gr) != MSGAT_OK) return 0;
  return agg_sddmm_fusable(*gr, sh->N, sh->T, sh->C) ? 1 : 0;
}

extern "C" int msgat_attention_backward(const msgat_shape_t* shp, const msgat_graph_t* gr, const float* u,
                                        const float* dv, int32_t dv_group_channels, const float* q, const float* kW,
                                        const float* lse, const float* pq, const float* E, const float* Ec_in,
                                        const float* Wg, float* du, float* dq, float* dWg, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  int st = check_shape(shp);
  if (st) return st;
  const msgat_shape_t shv = plain_shape(shp);
  const msgat_shape_t* sh = &shv;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!u || !dv || !q || !kW || !lse || !pq || !Wg || !du || !dq || !dWg) return MSGAT_ERR_NULL;
  if (gr->nnz > 0 && !E) return MSGAT_ERR_NULL;
  const BwdPlan p = plan_bwd(*sh, *gr);
  if (p.total > 0 && (!workspace || workspace_bytes < p.total)) return MSGAT_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  float* dEp = (float*)(ws + p.off_dEp);
  float* gE = (float*)(ws + p.off_gE);
  float* Ec = (float*)(ws + p.off_Ec);
  float* delta = (float*)(ws + p.off_delta);
  float* dkW = (float*)(ws + p.off_dkW);
  float* dwgp = (float*)(ws + p.off_dwg);
  const int G = p.G, Bg = sh->Bg, N = sh->N, T = sh->T;
  // du = E^T dv and the SDDMM walk the same (column, edge) pairs over the same dv slabs: one pass when the graph allows
  const bool fused = agg_sddmm_fusable(*gr, N, T, sh->C);
  if (dv_group_channels != 0 && (dv_group_channels < sh->C || !fused)) return MSGAT_ERR_SHAPE;  // see ..._accepts_strided_dv
  const float* Ecsc = Ec;  // E in CSC order: the forward's, or re-ordered here
  const int direct_c = (!fused && sh->C <= bwd_rows_direct_max_channels()) ? sh->C : 0;  // few channels: no SDDMM launch
  if (fused) {
    if (Ec_in != nullptr) {
      Ecsc = Ec_in;
    } else {
      st = launch_permute_edges(E, gr->cperm, Ec, G, gr->nnz, gr->nnz, s);
      if (st) return st;
    }
    st = launch_agg_sddmm(*gr, dv, Ecsc, u, du, dEp, G, sh->C, N, T, s, dv_group_channels);
  } else if (direct_c == 0) {
    st = launch_sddmm(*gr, u, dv, dEp, G, sh->C, N, T, s);
  }
  if (st) return st;
  st = launch_bwd_rows(*gr, dEp, p.nch, fused ? Ecsc : nullptr, direct_c, u, dv, E, q, pq, Wg, gE, delta, dkW, dq, dwgp, dWg, G, Bg, N,
                       T, s);
  if (st) return st;
  st = launch_bwd_dense_col(*gr, q, kW, lse, delta, gE, dq, G, N, T, s, ws + p.off_dense);
  if (st) return st;
  return fused ? MSGAT_OK : aggregate_cols(sh, gr, sh->C, dv, E, Ec, nullptr, nullptr, du, s, nullptr, nullptr, nullptr, Ec_in);
}

// The dense column pass of the backward alone (what msgat_attention_backward / msgat_gacn_backward enqueue after the
// edge gradients): dq[g,m] -= sum_n P[n,m] delta[n] kW[n] (+ the CSC edge term from gE).  For profiling and bench.py.
extern "C" int msgat_stage_dense_column_pass(const msgat_shape_t* sh, const msgat_graph_t* gr, const float* q,
                                             const float* kW, const float* lse, const float* delta, const float* gE,
                                             float* dq, void* dense_scratch, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!q || !kW || !lse || !delta || !dq || (gr->nnz > 0 && !gE)) return MSGAT_ERR_NULL;
  return launch_bwd_dense_col(*gr, q, kW, lse, delta, gE, dq, sh->R * sh->Bg, sh->N, sh->T, (hipStream_t)stream, dense_scratch);
}

// ---- prediction head ----------------------------------------------------------------------------------
static int check_head(int32_t B, int32_t C, int32_t N, int32_t T, int32_t To) {
  if (B <= 0 || B > 65535 || C <= 0 || C > 65535 || N <= 0 || To <= 0) return MSGAT_ERR_SHAPE;
  if (!t_supported(T) || To > 16) return MSGAT_ERR_UNSUPPORTED;
  return MSGAT_OK;
}

extern "C" size_t msgat_head_forward_partial_floats(int32_t B, int32_t C, int32_t N, int32_t To) {
  if (B <= 0 || C <= 0 || N <= 0 || To <= 0) return 0;
  return head_fwd_partial_floats(B, C, N, To);
}

extern "C" int msgat_head_forward(const float* x, const float* W, const float* bias, float* out, float* partials,
                                  int32_t B, int32_t C, int32_t N, int32_t T, int32_t To, int32_t R, void* stream) {
  if (!x || !W || !out || !partials) return MSGAT_ERR_NULL;
  int st = check_head(B, C, N, T, To);
  if (st) return st;
  if (R <= 0 || B % R) return MSGAT_ERR_SHAPE;
  return launch_head_fwd(x, W, bias, out, partials, B, C, N, T, To, R, (hipStream_t)stream);
}

extern "C" int msgat_head_forward_ln(const float* x, const float* ln_weight, const float* ln_bias, float eps, const float* W,
                                     const float* bias, float* out, float* normalised, float* partials, int32_t B, int32_t C,
                                     int32_t N, int32_t T, int32_t To, int32_t R, void* stream) {
  if (!x || !W || !out || !partials) return MSGAT_ERR_NULL;
  int st = check_head(B, C, N, T, To);
  if (st) return st;
  if (R <= 0 || B % R || !(eps >= 0.f)) return MSGAT_ERR_SHAPE;
  return launch_head_fwd(x, W, bias, out, partials, B, C, N, T, To, R, (hipStream_t)stream, 1, ln_weight, ln_bias, eps, normalised);
}

extern "C" int msgat_head_grad_signal(const float* dout, const float* W, float* dx, int32_t B, int32_t C, int32_t N,
                                      int32_t T, int32_t To, int32_t R, void* stream) {
  if (!dout || !W || !dx) return MSGAT_ERR_NULL;
  int st = check_head(B, C, N, T, To);
  if (st) return st;
  if (R <= 0 || B % R) return MSGAT_ERR_SHAPE;
  return launch_head_dx(dout, W, dx, B, C, N, T, To, R, (hipStream_t)stream);
}

extern "C" size_t msgat_layernorm_head_backward_partial_floats(int32_t B, int32_t C, int32_t N, int32_t T) {
  if (B <= 0 || C <= 0 || N <= 0 || !t_supported(T)) return 0;
  return lnhead_partial_floats(B, C, N, T);
}

extern "C" int msgat_layernorm_head_backward(const float* dout, const float* W, const float* x, const float* ln_weight,
                                             float* dx, float* dln_weight, float* dln_bias, float* partials, int32_t B,
                                             int32_t C, int32_t N, int32_t T, int32_t To, int32_t R, float eps,
                                             int32_t relu_mask, void* stream) {
  if (!dout || !W || !x || !dx || !partials) return MSGAT_ERR_NULL;
  int st = check_head(B, C, N, T, To);
  if (st) return st;
  if (R <= 0 || B % R || !(eps >= 0.f)) return MSGAT_ERR_SHAPE;
  return launch_lnhead_bwd(dout, W, x, ln_weight, dx, dln_weight, dln_bias, partials, B, C, N, T, To, R, eps,
                           relu_mask != 0, (hipStream_t)stream);
}

extern "C" size_t msgat_head_grad_weight_partial_floats(int32_t C, int32_t T, int32_t To, int32_t R) {
  if (C <= 0 || To <= 0 || R <= 0 || !t_supported(T)) return 0;
  return head_dw_partial_floats(C, T, To, R);
}

extern "C" int msgat_head_grad_weight(const float* dout, const float* x, float* dWc, float* partials, int32_t B,
                                      int32_t C, int32_t N, int32_t T, int32_t To, int32_t R, void* stream) {
  if (!dout || !x || !dWc || !partials) return MSGAT_ERR_NULL;
  int st = check_head(B, C, N, T, To);
  if (st) return st;
  if (R <= 0 || B % R) return MSGAT_ERR_SHAPE;
  return launch_head_dW(dout, x, dWc, partials, B, C, N, T, To, R, (hipStream_t)stream);
}

// ---- LayerNorm over T (the producer of the GACN inputs) ----------------------------------------------
extern "C" int msgat_layernorm_forward(const float* x, const float* weight, const float* bias, float* y,
                                       int64_t rows, int32_t T, float eps, int32_t R, void* stream) {
  if (!x || !y) return MSGAT_ERR_NULL;
  if (rows < 0 || !(eps >= 0.f) || R <= 0 || R > 65535 || rows % R) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  if (rows == 0) return MSGAT_OK;
  return launch_layernorm_fwd(x, weight, bias, y, rows, T, eps, R, (hipStream_t)stream);
}

extern "C" size_t msgat_layernorm_partial_floats(int64_t rows, int32_t T, int32_t R) {
  if (rows <= 0 || R <= 0 || rows % R || !t_supported(T)) return 0;
  return layernorm_partial_floats(rows, T, R);
}

extern "C" int msgat_layernorm_backward(const float* x, const float* weight, const float* dy, const float* dx_add,
                                        float* dx, float* dweight, float* dbias, float* partials, int64_t rows,
                                        int32_t T, float eps, int32_t R, int32_t relu_mask, void* stream) {
  if (!x || !dy || !dx || !partials) return MSGAT_ERR_NULL;
  if (rows <= 0 || !(eps >= 0.f) || R <= 0 || R > 65535 || rows % R) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  return launch_layernorm_bwd(x, weight, dy, dx_add, dx, dweight, dbias, partials, rows, T, eps, R, relu_mask != 0,
                              (hipStream_t)stream);
}

extern "C" size_t msgat_layernorm_pool_partial_floats(int64_t rows, int32_t T, int32_t R) {
  if (rows <= 0 || R <= 0 || rows % R || !t_supported(T)) return 0;
  return layernorm_pool_partial_floats(rows, T, R);
}

extern "C" int msgat_layernorm_forward_pooled(const float* x, const float* weight, const float* bias, float* y,
                                              const float* pool_w, int32_t N, float* pooled, float* partials, int64_t rows,
                                              int32_t T, float eps, int32_t R, void* stream) {
  if (!x || !y || !pool_w || !pooled || !partials) return MSGAT_ERR_NULL;
  if (rows <= 0 || !(eps >= 0.f) || R <= 0 || R > 65535 || rows % R || N <= 0 || (rows / R) % N) return MSGAT_ERR_SHAPE;
  if (!t_supported(T) || N < kWave) return MSGAT_ERR_UNSUPPORTED;
  return launch_layernorm_fwd(x, weight, bias, y, rows, T, eps, R, (hipStream_t)stream, pool_w, partials, pooled, N);
}

extern "C" int msgat_layernorm_backward_pooled(const float* x, const float* weight, const float* bias, const float* dy,
                                               const float* dx_add, const float* pool_w, const float* dpooled, int32_t N,
                                               float* dx, float* dweight, float* dbias, float* dpool_w, float* dpool_rows,
                                               float* partials, int64_t rows, int32_t T, float eps, int32_t R,
                                               int32_t relu_mask, void* stream) {
  if (!x || !dy || !dx || !partials || !pool_w || !dpooled) return MSGAT_ERR_NULL;
  if ((dpool_w == nullptr) != (dpool_rows == nullptr)) return MSGAT_ERR_NULL;
  if (rows <= 0 || !(eps >= 0.f) || R <= 0 || R > 65535 || rows % R || N <= 0 || (rows / R) % N) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  return launch_layernorm_bwd(x, weight, dy, dx_add, dx, dweight, dbias, partials, rows, T, eps, R, relu_mask != 0,
                              (hipStream_t)stream, pool_w, dpooled, N, bias, dpool_rows, dpool_w);
}

// ---- fused forward ---------------------------------------------------------------------------------
extern "C" int msgat_gacn_forward(const msgat_shape_t* sh, const msgat_graph_t* gr,
                                  const msgat_fwd_t* io, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!io) return MSGAT_ERR_NULL;
  if (!io->x || !io->alpha || !io->Wg || !io->z || !io->q || !io->kW || !io->lse) return MSGAT_ERR_NULL;
  if (gr->nnz > 0 && !io->E) return MSGAT_ERR_NULL;
  const int mode = msgat_gacn_mode(sh->C, sh->Co);
  if (mode != MSGAT_MODE_PLAIN && !io->W) return MSGAT_ERR_NULL;
  if (mode == MSGAT_MODE_PROJ_FIRST && !io->u) return MSGAT_ERR_NULL;
  if (io->need_bwd && !io->pq) return MSGAT_ERR_NULL;
  if (io->need_bwd && mode == MSGAT_MODE_AGG_FIRST && !io->u) return MSGAT_ERR_NULL;

  hipStream_t s = (hipStream_t)stream;
  const int G = sh->R * sh->Bg, P = sh->N * sh->T;
  float* pq = io->need_bwd ? io->pq : nullptr;

  // q = alpha . x: out of the projection pass (PROJ_FIRST), inside the score kernel (one input channel; three when the
  // kernel also finishes an AGG_FIRST layer), or from its own small pass
  const bool split = dense_split_selected(sh->N, sh->T);   // the split-operand score kernel reads q from memory
  if (split && !io->dense_scratch) return MSGAT_ERR_WORKSPACE;
  const bool tail = !split && mode == MSGAT_MODE_AGG_FIRST && sh->Co <= 64 && scores_take_x(sh->C, true) &&
                    !sell_usable(gr->sell_rows, gr->nnz, sh->N, sh->T);
  const bool q_in_scores = !split && mode != MSGAT_MODE_PROJ_FIRST && scores_take_x(sh->C, tail);
  if (mode == MSGAT_MODE_PROJ_FIRST)
    st = launch_project(io->x, io->W, 0, io->alpha, nullptr, nullptr, io->u, io->q, G, sh->Bg, sh->C, sh->Co, P, s);
  else if (!q_in_scores)
    st = launch_qonly(io->x, io->alpha, io->q, G, sh->Bg, sh->C, P, s);
  if (st) return st;

  if (q_in_scores)
    st = launch_scores(*gr, nullptr, io->Wg, io->kW, io->lse, pq, io->E, io->Ec, G, sh->Bg, sh->N, sh->T, s, io->x, io->alpha,
                       sh->C, io->q, tail ? io->W : nullptr, tail ? sh->Co : 0, (tail && io->need_bwd) ? io->u : nullptr,
                       tail ? io->z : nullptr);
  else
    st = launch_scores(*gr, io->q, io->Wg, io->kW, io->lse, pq, io->E, io->Ec, G, sh->Bg, sh->N, sh->T, s, nullptr, nullptr, 0,
                       nullptr, nullptr, 0, nullptr, nullptr, io->dense_scratch);
  if (st) return st;
  if (tail) return MSGAT_OK;   // z (and y) came out of the score kernel

  switch (mode) {
    case MSGAT_MODE_PLAIN:
      return aggregate_rows(sh, gr, sh->C, io->x, io->E, io->z, io->edge_scratch, s);
    case MSGAT_MODE_AGG_FIRST:
      return launch_aggregate_project(*gr, io->x, io->E, io->W, io->need_bwd ? io->u : nullptr, io->z, G,
                                      sh->Bg, sh->C, sh->Co, sh->N, sh->T, s);
    default:
      return aggregate_rows(sh, gr, sh->Co, io->u, io->E, io->z, io->edge_scratch, s);
  }
}

// ---- fused backward --------------------------------------------------------------------------------
extern "C" size_t msgat_bwd_workspace_bytes(const msgat_shape_t* sh, const msgat_graph_t* gr) {
  if (check_shape(sh) != MSGAT_OK || check_graph(sh, gr) != MSGAT_OK) return 0;
  return plan_bwd(*sh, *gr).total;
}

extern "C" int msgat_bwd_accepts_strided_dz(const msgat_shape_t* sh, const msgat_graph_t* gr) {
  if (check_shape(sh) != MSGAT_OK || check_graph(sh, gr) != MSGAT_OK) return 0;
  const int mode = msgat_gacn_mode(sh->C, sh->Co);
  if (mode == MSGAT_MODE_AGG_FIRST) return 1;  // dz feeds channel-mixing passes only, which take a group stride
  if (mode == MSGAT_MODE_PROJ_FIRST) return agg_sddmm_fusable(*gr, sh->N, sh->T, sh->Co) ? 1 : 0;
  return 0;
}

extern "C" int msgat_gacn_backward(const msgat_shape_t* sh, const msgat_graph_t* gr,
                                   const msgat_bwd_t* io, void* stream) {
  int st = check_shape(sh);
  if (st) return st;
  st = check_graph(sh, gr);
  if (st) return st;
  if (!io) return MSGAT_ERR_NULL;
  if (!io->x || !io->alpha || !io->Wg || !io->q || !io->kW || !io->lse || !io->pq || !io->dz ||
      !io->dx || !io->dalpha || !io->dWg || !io->workspace)
    return MSGAT_ERR_NULL;
  if (gr->nnz > 0 && !io->E) return MSGAT_ERR_NULL;
  const BwdPlan p = plan_bwd(*sh, *gr);
  if (p.mode != MSGAT_MODE_PLAIN && (!io->W || !io->dW || !io->u)) return MSGAT_ERR_NULL;
  if (io->workspace_bytes < p.total) return MSGAT_ERR_WORKSPACE;
  if (((uintptr_t)io->workspace & 255) != 0) return MSGAT_ERR_WORKSPACE;

  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)io->workspace;
  float* dEp = (float*)(ws + p.off_dEp);
  float* gE = (float*)(ws + p.off_gE);
  float* Ec = (float*)(ws + p.off_Ec);
  float* delta = (float*)(ws + p.off_delta);
  float* dkW = (float*)(ws + p.off_dkW);
  float* dq = (float*)(ws + p.off_dq);
  float* dvb = (float*)(ws + p.off_dv);
  float* dwgp = (float*)(ws + p.off_dwg);
  float* cpp = (float*)(ws + p.off_cp);
  const int G = p.G, Bg = sh->Bg, C = sh->C, Co = sh->Co, N = sh->N, T = sh->T, P = N * T;

  // the fixed-order sums of the parameter-gradient partials (dW, dalpha, dWg) are queued and run as ONE launch at the
  // end of the pass
  ReduceJobs jobs{};
  float* cpp2 = (float*)(ws + p.off_cp2);

  // dz as a channel slice of a wider tensor: group stride in channels (its own width when contiguous)
  const int zc = (p.mode == MSGAT_MODE_PLAIN) ? C : Co;
  const bool strided = io->dz_group_channels != 0 && io->dz_group_channels != zc;
  if (io->dz_group_channels != 0 && io->dz_group_channels < zc) return MSGAT_ERR_SHAPE;
  const int dzgs = strided ? io->dz_group_channels : zc;

  // features the attention acted on (u) and the gradient arriving at its output (dv)
  const float* u = (p.mode == MSGAT_MODE_PROJ_FIRST) ? io->u : io->x;
  const float* dv = io->dz;
  if (p.mode == MSGAT_MODE_AGG_FIRST) {
    // z = W y:  dy = W^T dz,  dW = dz y^T -- one pass over dz when the input has few channels
    if (C <= kAggFirstMaxC) {
      st = launch_aggfirst_bwd(io->dz, io->W, io->u, dvb, cpp, io->dW, G, Bg, C, Co, P, s, &jobs, dzgs);
    } else {
      SegList dzs = seg_single(io->dz, Co);
      dzs.gstride[0] = dzgs;
      st = launch_project_seg(dzs, io->W, 1, nullptr, nullptr, nullptr, seg_single(dvb, C), nullptr, G, Bg, P,
                              MixEpilogue{}, s);
      if (st) return st;
      st = launch_chanpair_seg(dzs, io->u, cpp, io->dW, Co * C, nullptr, 0, G, Bg, C, P, s, 0, &jobs);
    }
    if (st) return st;
    dv = dvb;
  }

  // PROJ_FIRST: du = E^T dz does not wait for dq, and it walks the same (column, edge) pairs over the same dz slabs
  // as the SDDMM -- one pass does both when the graph allows (CSC, slab + edge shares within half the LDS)
  const bool fused = p.mode == MSGAT_MODE_PROJ_FIRST && agg_sddmm_fusable(*gr, N, T, p.Cu);
  if (strided && p.mode != MSGAT_MODE_AGG_FIRST && !fused) return MSGAT_ERR_SHAPE;  // see msgat_bwd_accepts_strided_dz
  const float* Ecsc = Ec;  // E in CSC order: the forward's (io->Ec), or re-ordered here
  // attention over few channels (the first MEAM of every component): dE is computed inside the row pass, no SDDMM
  const int direct_c = (!fused && p.Cu <= bwd_rows_direct_max_channels()) ? p.Cu : 0;
  if (fused) {
    if (io->Ec != nullptr) {
      Ecsc = io->Ec;
    } else {
      st = launch_permute_edges(io->E, gr->cperm, Ec, G, gr->nnz, gr->nnz, s);
      if (st) return st;
    }
    st = launch_agg_sddmm(*gr, dv, Ecsc, u, dvb, dEp, G, p.Cu, N, T, s, dzgs);
  } else if (direct_c == 0) {
    st = launch_sddmm(*gr, u, dv, dEp, G, p.Cu, N, T, s);
  }
  if (st) return st;
  st = launch_bwd_rows(*gr, dEp, p.nch, fused ? Ecsc : nullptr, direct_c, u, dv, io->E, io->q, io->pq, io->Wg, gE, delta, dkW, dq, dwgp,
                       io->dWg, G, Bg, N, T, s, &jobs);
  if (st) return st;
  st = launch_bwd_dense_col(*gr, io->q, io->kW, io->lse, delta, gE, dq, G, N, T, s, ws + p.off_dense);
  if (st) return st;

  if (p.mode == MSGAT_MODE_PROJ_FIRST) {
    // du = E^T dz;  dx = W^T du + alpha (x) dq;  dW = du x^T;  dalpha = dq . x
    if (!fused) {
      st = aggregate_cols(sh, gr, Co, dv, io->E, Ec, nullptr, nullptr, dvb, s, nullptr, nullptr, nullptr, io->Ec);
      if (st) return st;
    }
    // the contraction (reads only) goes first: behind the projection it would stream x while the 72-channel dx the
    // projection has just written is still draining to HBM
    // ... and where its channel block leaves the registers (LDS-DMA staging), the same pass also writes dx: du and dq
    // are read once
    int both = 0;   // (msgat_stage_project_backward is this stage on its own)
    st = launch_chanpair_mix(dvb, dq, io->x, io->W, io->alpha, io->dx, cpp, io->dW, io->dalpha, G, Bg, Co, C, P, s, &jobs,
                             &both);
    if (st) return st;
    if (!both) {
      st = launch_chanpair(dvb, dq, io->x, cpp, io->dW, Co * C, io->dalpha, C, G, Bg, Co + 1, C, P, s, &jobs);
      if (st) return st;
      st = launch_project(dvb, io->W, 1, nullptr, io->alpha, dq, io->dx, nullptr, G, Bg, Co, C, P, s);
      if (st) return st;
    }
    return launch_reduce_jobs(jobs, s);
  }
  // PLAIN / AGG_FIRST:  dx = E^T dv + alpha (x) dq;  dalpha = dq . x -- from the same kernel (it holds dq) when the
  // slab form runs and the input has few channels, from a contraction otherwise
  float* dap = (p.mode == MSGAT_MODE_AGG_FIRST) ? cpp2 : cpp;
  int dot_done = 0;
  st = aggregate_cols(sh, gr, C, dv, io->E, Ec, io->alpha, dq, io->dx, s, C <= kAggDotMaxC ? io->x : nullptr, dap,
                      &dot_done, io->Ec);
  if (st) return st;
  if (dot_done)
    st = launch_reduce_groups_defer(dap, sh->R, Bg * dot_done, C, io->dalpha, s, &jobs);
  else
    st = launch_chanpair(nullptr, dq, io->x, dap, nullptr, 0, io->dalpha, C, G, Bg, 1, C, P, s, &jobs);
  if (st) return st;
  return launch_reduce_jobs(jobs, s);
}

// ---- the gated sum over the components (msgat.py:203-205) --------------------------------------------------------
static int check_gate(const float* h_w, const int64_t* H, const int64_t* D, const float* d_w, int32_t R, int32_t B, int64_t E,
                      int32_t nh, int32_t nd) {
  if (!h_w) return MSGAT_ERR_NULL;
  if (R <= 0 || B <= 0 || E <= 0 || nh <= 0 || (long long)R * B * E > 0x7fffffffLL * 64) return MSGAT_ERR_SHAPE;
  if (E > 0x7fffffff / (R > nh + nd ? R : nh + nd + 1)) return MSGAT_ERR_SHAPE;   // 32-bit (row, r, e) arithmetic
  if ((d_w != nullptr) != (D != nullptr) || (d_w && nd <= 0)) return MSGAT_ERR_SHAPE;
  if (!H && nh != 1) return MSGAT_ERR_SHAPE;                                      // static gate: one row
  return MSGAT_OK;
}

extern "C" int msgat_gate_sum(const float* pred, const int64_t* H, const int64_t* D, const float* h_w, const float* d_w,
                              float* out, int32_t R, int32_t B, int64_t E, int32_t nh, int32_t nd, void* stream) {
  if (!pred || !out) return MSGAT_ERR_NULL;
  if (int st = check_gate(h_w, H, D, d_w, R, B, E, nh, nd)) return st;
  return launch_gate_sum(pred, (const long long*)H, (const long long*)D, h_w, d_w, out, R, B, (int)E, nh, nd,
                         (hipStream_t)stream);
}

extern "C" int msgat_gate_sum_backward(const float* dout, const float* pred, const int64_t* H, const int64_t* D,
                                       const float* h_w, const float* d_w, float* dpred, float* dh_w, float* dd_w,
                                       int32_t R, int32_t B, int64_t E, int32_t nh, int32_t nd, void* stream) {
  if (!dout || !pred) return MSGAT_ERR_NULL;
  if (int st = check_gate(h_w, H, D, d_w, R, B, E, nh, nd)) return st;
  return launch_gate_sum_bwd(dout, pred, (const long long*)H, (const long long*)D, h_w, d_w, dpred, dh_w, dd_w, R, B, (int)E,
                             nh, nd, (hipStream_t)stream);
}

// ---- step tail: fused Huber loss + metric sums, flat Adam ------------------------------------------------------
extern "C" size_t msgat_huber_partial_doubles(int64_t n) { return n > 0 ? huber_partial_doubles(n) : 0; }

extern "C" int msgat_huber_metrics(const float* pred, const float* truth, int64_t n, float delta, float mask_value,
                                   double* partials, float* loss, double* sums, float loss_weight, void* stream) {
  if (!pred || !truth || !partials || !loss) return MSGAT_ERR_NULL;
  if (n <= 0 || !(delta > 0.f) || !(loss_weight >= 0.f)) return MSGAT_ERR_SHAPE;
  return launch_huber_metrics(pred, truth, n, delta, mask_value, partials, loss, sums, loss_weight, (hipStream_t)stream);
}

extern "C" int msgat_huber_grad(const float* pred, const float* truth, const float* dloss, int64_t n, float delta,
                                float* dpred, void* stream) {
  if (!pred || !truth || !dloss || !dpred) return MSGAT_ERR_NULL;
  if (n <= 0 || !(delta > 0.f)) return MSGAT_ERR_SHAPE;
  return launch_huber_grad(pred, truth, dloss, n, delta, dpred, (hipStream_t)stream);
}

extern "C" int msgat_adam_chunk_elems(void) { return adam_chunk_elems(); }

extern "C" int msgat_adam_step(float* const* chunk_param, const int64_t* chunk_off, const int32_t* chunk_len,
                               const int32_t* chunk_tensor, int32_t n_chunks, const int32_t* active_tensors,
                               int32_t n_active, const float* grad, float* exp_avg, float* exp_avg_sq, float* steps,
                               const float* lr, double beta1, double beta2, double eps, double weight_decay,
                               const float* grad_divisor, void* stream) {
  if (n_chunks < 0 || n_active < 0) return MSGAT_ERR_SHAPE;
  if (!steps || !lr) return MSGAT_ERR_NULL;
  if (n_active > 0 && !active_tensors) return MSGAT_ERR_NULL;
  if (n_chunks > 0 && (!chunk_param || !chunk_off || !chunk_len || !chunk_tensor || !grad || !exp_avg || !exp_avg_sq)) return MSGAT_ERR_NULL;
  if (!(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && weight_decay >= 0.)) return MSGAT_ERR_SHAPE;
  return launch_adam(chunk_param, (const long long*)chunk_off, chunk_len, chunk_tensor, n_chunks, active_tensors, n_active,
                     grad, exp_avg, exp_avg_sq, steps, lr, beta1, beta2, eps, weight_decay, grad_divisor, (hipStream_t)stream);
}

extern "C" int msgat_gather_scaled(const float* const* chunk_src, const int64_t* chunk_off, const int32_t* chunk_len,
                                   int32_t n_chunks, float scale, float* flat, int64_t weight_index, void* stream) {
  if (n_chunks <= 0) return MSGAT_ERR_SHAPE;
  if (!chunk_src || !chunk_off || !chunk_len || !flat) return MSGAT_ERR_NULL;
  return launch_gather_scaled(chunk_src, (const long long*)chunk_off, chunk_len, n_chunks, scale, flat, weight_index,
                              (hipStream_t)stream);
}

// ---- the tiny attention matrices of a MEAM block ----------------------------------------------------------------
static int check_small(int32_t G, int32_t R, int32_t T) {
  if (G <= 0 || R <= 0 || G % R) return MSGAT_ERR_SHAPE;
  if (!t_supported(T)) return MSGAT_ERR_UNSUPPORTED;
  return MSGAT_OK;
}

extern "C" int msgat_channel_attention_forward(const float* pooled, const float* Wc, const float* conv, float* att,
                                               float* Mc, int32_t G, int32_t R, int32_t C, int32_t cb, int32_t T,
                                               void* stream) {
  if (!pooled || !Wc || !conv || !att || !Mc) return MSGAT_ERR_NULL;
  int st = check_small(G, R, T);
  if (st) return st;
  if (C <= 0 || C > kMaxC || cb <= 0 || cb > kMaxC) return MSGAT_ERR_SHAPE;
  return launch_chanatt_fwd(pooled, Wc, conv, att, Mc, G, R, C, cb, T, (hipStream_t)stream);
}

extern "C" size_t msgat_channel_attention_partial_floats(int32_t G, int32_t C, int32_t cb, int32_t T) {
  if (G <= 0 || C <= 0 || cb <= 0 || !t_supported(T)) return 0;
  return chanatt_partial_floats(G, C, cb, T);
}

extern "C" int msgat_channel_attention_backward(const float* dMc, const float* att, const float* pooled,
                                                const float* Wc, const float* conv, float* dpooled, float* dWc,
                                                float* dconv, float* partials, int32_t G, int32_t R, int32_t C,
                                                int32_t cb, int32_t T, void* stream) {
  if (!dMc || !att || !pooled || !Wc || !conv || !dpooled || !dWc || !dconv || !partials) return MSGAT_ERR_NULL;
  int st = check_small(G, R, T);
  if (st) return st;
  if (C <= 0 || C > kMaxC || cb <= 0 || cb > kMaxC) return MSGAT_ERR_SHAPE;
  return launch_chanatt_bwd(dMc, att, pooled, Wc, conv, dpooled, dWc, dconv, partials, G, R, C, cb, T, (hipStream_t)stream);
}

extern "C" int msgat_temporal_attention_forward(const float* pooled, const float* Wt1, const float* Wt2, float* lr,
                                                float* att, float* taps, int32_t G, int32_t R, int32_t N, int32_t K,
                                                int32_t T, int32_t dilation, void* stream) {
  if (!pooled || !Wt1 || !Wt2 || !lr || !att || !taps) return MSGAT_ERR_NULL;
  int st = check_small(G, R, T);
  if (st) return st;
  if (N <= 0 || K <= 0 || dilation < 0) return MSGAT_ERR_SHAPE;
  return launch_tempatt_fwd(pooled, Wt1, Wt2, lr, att, taps, G, R, N, K, T, dilation, (hipStream_t)stream);
}

extern "C" size_t msgat_temporal_attention_partial_floats(int32_t G, int32_t K, int32_t N) {
  if (G <= 0 || K <= 0 || N <= 0) return 0;
  return tempatt_partial_floats(G, K, N);
}

extern "C" int msgat_temporal_attention_backward(const float* dtaps, const float* att, const float* lr,
                                                 const float* pooled, const float* Wt1, const float* Wt2,
                                                 float* dpooled, float* dWt1, float* dWt2, float* partials, int32_t G,
                                                 int32_t R, int32_t N, int32_t K, int32_t T, int32_t dilation,
                                                 void* stream) {
  if (!dtaps || !att || !lr || !pooled || !Wt1 || !Wt2 || !dpooled || !dWt1 || !dWt2 || !partials) return MSGAT_ERR_NULL;
  int st = check_small(G, R, T);
  if (st) return st;
  if (N <= 0 || K <= 0 || dilation < 0) return MSGAT_ERR_SHAPE;
  return launch_tempatt_bwd(dtaps, att, lr, pooled, Wt1, Wt2, dpooled, dWt1, dWt2, partials, G, R, N, K, T, dilation,
                            (hipStream_t)stream);
}
