/*
 * msgat_hip.h -- C ABI of libmsgat_hip.so: the MI355X (gfx950) implementation of the
 * MS-GAT graph-attention hot path.
 *
 * The reference (luokn/ms-gat) has no FFI: its hot path is a sequence of PyTorch eager
 * ops.  Each entry point below names the reference lines it replaces (paths under
 * /root/reference/).  INTEGRATION.md shows the ctypes binding a maintainer adds to call
 * them from `GraphAttention.forward` / `GACN.forward`.
 *
 * Conventions
 *   - plain C, PODs only: raw device pointers, ints, a hipStream_t passed as void*.
 *   - every device entry point only ENQUEUES kernels on `stream` and returns; it never
 *     allocates, frees or synchronises (re-entrant; safe under hipGraph capture).  All
 *     buffers, including the tensors saved for backward and the workspace, are owned by
 *     the caller.  The only process-wide state is a set of per-device, write-once caches
 *     inside the launchers (csrc/common.hpp: the device's compute-unit count, and per
 *     kernel the dynamic-LDS ceiling already granted by hipFuncSetAttribute): lock-free,
 *     idempotent -- a lost race repeats a driver call -- and indexed by the CURRENT HIP
 *     device, which the caller makes the device of `stream` before the call.  Device
 *     ordinals >= 32 (or a failed hipGetDevice) are never aliased to another device's
 *     entry: for them the driver calls are simply repeated on every launch.
 *   - return value: 0 (MSGAT_OK) or a negative MSGAT_ERR_* code; hipError_t e from a
 *     launch is reported as MSGAT_ERR_HIP_BASE - e.  No exceptions cross the ABI.
 *   - all floating point is IEEE fp32; tensors are dense, contiguous, row-major.  (From N = 1536 nodes the two dense
 *     passes of the attention multiply on the bf16 / fp16 matrix core with every fp32 operand split into three bf16 or two
 *     scaled fp16 terms and fp32 accumulation -- fp32-class accuracy, see msgat_dense_scratch_bytes.)
 *
 * Vocabulary
 *   relation r in [0,R)  one independent parameter set (alpha_r, Wg_r, W_r).  The
 *                        reference evaluates R = n_components such sets over one shared
 *                        adjacency (msgat.py:191-199,204); a single GACN call has R = 1.
 *   group    g in [0,G)  one (relation, sample) pair, G = R*Bg, g = r*Bg + b.  A group is
 *                        one [N,N] attention problem (attention is per sample,
 *                        attention.py:34).
 *   signals  x[G,C,N,T]  the reference's [batch, channels, nodes, timesteps] layout
 *                        (attention.py:21), relations stacked on the batch axis.
 *   graph                CSR + CSC of the non-zeros of the [N,N] adjacency
 *                        (data_loader.py:59-66 builds it dense; msgat.py:190 holds it).
 */
#ifndef MSGAT_HIP_H
#define MSGAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7 (round 6): + msgat_dense_scratch_bytes; msgat_fwd_t.dense_scratch (new last field); msgat_stage_scores and
 * msgat_stage_dense_column_pass take a dense_scratch pointer in front of the stream.  The one process-wide environment switch,
 * MSGAT_DENSE_SPLIT = 0 | 1, forces the arithmetic of the dense passes for A/B runs and tests (read once).
 * 6 (round 5): + msgat_contract_form_name, msgat_causal_conv{,_fused,_grad_weight}, msgat_layernorm_head_backward
 * {,_partial_floats}, msgat_head_forward_ln, msgat_gate_sum{,_backward}, msgat_layernorm_{forward,backward}_pooled, msgat_layernorm_pool_partial_floats,
 * msgat_contract_mix_partial_floats; no existing signature or structure changed since 5. */
#define MSGAT_ABI_VERSION 7

enum {
  MSGAT_OK = 0,
  MSGAT_ERR_NULL = -1,        /* a required pointer is NULL                         */
  MSGAT_ERR_SHAPE = -2,       /* a dimension is <= 0 or inconsistent                 */
  MSGAT_ERR_UNSUPPORTED = -3, /* T not in {4,8,12,16}, C/Co beyond the built limits  */
  MSGAT_ERR_WORKSPACE = -4,   /* workspace smaller than msgat_bwd_workspace_bytes()  */
  MSGAT_ERR_GRAPH = -5,       /* malformed CSR/CSC                                   */
  MSGAT_ERR_HIP_BASE = -1000  /* -1000 - hipError_t                                  */
};

/* How a GACN call orders aggregation and channel projection (they commute: the
 * aggregation acts on the node axis, W on the channel axis). */
enum {
  MSGAT_MODE_PLAIN = 0,   /* Co == 0: GraphAttention only (attention.py:32-36)               */
  MSGAT_MODE_AGG_FIRST = 1, /* C <= Co: aggregate C channels, then project (msgat.py order)  */
  MSGAT_MODE_PROJ_FIRST = 2 /* C >  Co: project to Co channels, then aggregate (3x less gather) */
};

/* Sliced ELLPACK form (SELL-64, rows sorted by degree) of one sparse structure -- the rows of the
 * CSR, or the columns of the CSC -- used by the aggregate / SDDMM kernels when an [N,T] slab does
 * not fit LDS (N > ~3400 at T = 12; the N = 8192 stress graph).  Those kernels keep ONE 4-timestep
 * column of a slab in LDS per pass, so the edge lists are re-read once per (group, channel, column):
 * several times the bytes of the features themselves, and through the CSR that is dependent round
 * trips of 16-B pieces at a ~68-B stride plus a dozen address instructions per edge.  Here the rows
 * are sorted by degree (descending, stable), cut into slices of 64 (one wavefront) and every slice
 * is padded to its largest degree rounded up to a multiple of 4 and stored in "trips" of 4 edges
 * per row, lane-interleaved:
 *     entry (slice s, k-th edge, lane l)  at position  slice_off[s] + 256 (k / 4) + 4 l + k % 4
 * so lane l reads its edges 4t .. 4t+3 with ONE coalesced 16-B-per-lane load per array (1 KiB per
 * wave instruction): no dependent address, no mask (padding entries carry coefficient 0).
 * Sorting makes the degrees inside a slice nearly equal, so the padding is a few percent.  WHICH of a row's
 * edges is its k-th is the builder's choice: it fills every column so that the 16 lanes a ds_read_b128
 * serves together gather from different LDS bank quads where the graph allows (a bipartite matching per
 * column; 11.5 -> 6.5 LDS cycles per gather instruction at the stress graph).
 * Built on the host by msgat_graph_sell_build().  n_slices == 0: absent (the CSR kernels run). */
#define MSGAT_SELL_SLACK 512 /* idx (and every per-position buffer) carries this many readable trailing entries */
typedef struct msgat_sell {
  int32_t n_slices;         /* ceil(N / 64), or 0 when the layout is absent                          */
  int32_t n_pos;            /* positions in total = slice_off[n_slices] (>= nnz: padding included)    */
  const int32_t* slice_off; /* [n_slices+1]  first position of each slice; width = difference / 64    */
  const int32_t* lane_row;  /* [64 n_slices] row handled by each lane of a slice, -1 past N           */
  const uint16_t* idx;      /* [n_pos+SLACK] neighbour node of each position (0 at padding); the layout
                               is only built for N <= 10176 (one float4 per node must fit LDS)         */
  const int32_t* src;       /* [n_pos]       position -> CSR edge index, -1 at padding                */
  const int32_t* pos;       /* [nnz]         CSR edge index -> position (row form only)               */
  int32_t prefer;           /* != 0: use the SELL kernels even when a whole slab fits LDS (tests)     */
  int32_t pair_trips;       /* max over i of trips(slice i) + trips(slice n_slices-1-i), a trip = 4
                               columns: the SDDMM keeps a wave's slice PAIR in registers when it fits    */
} msgat_sell_t;

/* Device-resident sparse adjacency.  Built on the host by msgat_graph_build() and copied
 * to the device by the caller. */
typedef struct msgat_graph {
  int32_t n_nodes;
  int32_t nnz;
  const int32_t* rowptr; /* [N+1]  CSR row starts                                   */
  const int32_t* col;    /* [nnz]  column of each CSR edge                          */
  const float* val;      /* [nnz]  adjacency weight of each CSR edge                */
  const int32_t* erow;   /* [nnz]  row of each CSR edge (COO companion of col)      */
  const int32_t* colptr; /* [N+1]  CSC column starts                                */
  const int32_t* crow;   /* [nnz]  row of each CSC entry                            */
  const int32_t* cperm;  /* [nnz]  CSC position -> CSR edge index                   */
  const int32_t* cpos;   /* [nnz]  CSR edge index -> CSC position (inverse of cperm): lets the forward leave the
                                   edge coefficients in CSC order too, so backward starts without a re-ordering pass */
  msgat_sell_t sell_rows; /* SELL of the CSR (forward aggregate, SDDMM); optional     */
  msgat_sell_t sell_cols; /* SELL of the CSC (transposed aggregate of backward); opt. */
} msgat_graph_t;

typedef struct msgat_shape {
  int32_t R;  /* relations (parameter sets)                      */
  int32_t Bg; /* groups per relation; G = R*Bg                    */
  int32_t C;  /* input channels                                   */
  int32_t Co; /* output channels of W; 0 = no projection (PLAIN)  */
  int32_t N;  /* nodes                                            */
  int32_t T;  /* timesteps (the score contraction axis)           */
} msgat_shape_t;

/* Forward buffers.  `need_bwd` != 0 makes the call fill everything backward needs. */
typedef struct msgat_fwd {
  const float* x;     /* in  [G,C,N,T]                                                    */
  const float* alpha; /* in  [R,C]        attention.py:30                                 */
  const float* Wg;    /* in  [R,T,T]      attention.py:29                                 */
  const float* W;     /* in  [R,Co,C]     msgat.py:23 (NULL iff Co == 0)                  */
  float* z;           /* out [G,Co,N,T]   (PLAIN: [G,C,N,T])                              */
  float* q;           /* out [G,N,T]      q = k = sum_c alpha_c x_c   (attention.py:33)   */
  float* kW;          /* out [G,N,T]      k @ Wg                      (attention.py:34)   */
  float* lse;         /* out [G,N]        row log-sum-exp of the scores in log2 units:
                             log2 sum_m 2^(S[n,m] log2 e); natural log = lse * ln 2        */
  float* pq;          /* out [G,N,T]      softmax(S) @ q, only written when need_bwd      */
  float* E;           /* out [G,nnz]      softmax(S)*adj at the edges (attention.py:36)   */
  float* u;           /* out AGG_FIRST: y [G,C,N,T] when need_bwd; PROJ_FIRST: W x [G,Co,N,T];
                             PLAIN: unused (may be NULL)                                  */
  int32_t need_bwd;
  float* edge_scratch; /* tmp msgat_edge_scratch_floats() floats (0 unless the graph carries a SELL
                             layout the aggregate will use: E re-ordered for it); may be NULL then */
  float* Ec;          /* out [G,nnz]      E in CSC order (Ec[g, cpos[e]] = E[g, e]), written by the score kernel
                             itself; optional (NULL: not written).  Hand it to msgat_bwd_t.Ec */
  void* dense_scratch; /* tmp msgat_dense_scratch_bytes() bytes, 256-byte aligned (0 for the PEMS-sized graphs: may be
                             NULL then): the operand images of the score pass on large graphs (ABI v7) */
} msgat_fwd_t;

typedef struct msgat_bwd {
  /* forward inputs and what forward saved */
  const float* x;
  const float* alpha;
  const float* Wg;
  const float* W;
  const float* q;
  const float* kW;
  const float* lse;
  const float* pq;
  const float* E;
  const float* u;
  /* incoming gradient, same shape as z; contiguous unless dz_group_channels says otherwise (below) */
  const float* dz;
  /* gradients out (overwritten, not accumulated) */
  float* dx;     /* [G,C,N,T]                       */
  float* dalpha; /* [R,C]                           */
  float* dWg;    /* [R,T,T]                         */
  float* dW;     /* [R,Co,C]   (NULL iff Co == 0)   */
  /* scratch, >= msgat_bwd_workspace_bytes() bytes, 256-byte aligned */
  void* workspace;
  size_t workspace_bytes;
  /* 0, or the channel count of the tensor dz is a channel slice of: group g of dz then starts at
   * dz + g * dz_group_channels * N * T (a gradient that arrives as dout[:, a:b] of a wider [G,C',N,T] tensor is read in
   * place instead of being copied).  Only where msgat_bwd_accepts_strided_dz() says so; 0 everywhere else. */
  int32_t dz_group_channels;
  /* E in CSC order as the forward left it (msgat_fwd_t.Ec); NULL: backward re-orders E itself (one more launch) */
  const float* Ec;
} msgat_bwd_t;

/* ---- library ------------------------------------------------------------------- */
int msgat_abi_version(void);
const char* msgat_status_string(int status);
/* MSGAT_MODE_* the library uses for (C, Co); the caller sizes `u` from it. */
int msgat_gacn_mode(int32_t C, int32_t Co);

/* ---- host: dense adjacency -> CSR/CSC ---------------------------------------------
 * Replaces the dense mask `att * adjacency` of attention.py:36 by its non-zeros.
 * `adj` is a HOST pointer to the row-major [n,n] fp32 matrix with leading dimension ld.
 * An entry is an edge iff it is != 0 (NaN counts as an edge). */
int msgat_graph_count(const float* adj, int32_t n, int64_t ld, int32_t* nnz_out);
int msgat_graph_build(const float* adj, int32_t n, int64_t ld, int32_t nnz,
                      int32_t* rowptr, int32_t* col, float* val, int32_t* erow,
                      int32_t* colptr, int32_t* crow, int32_t* cperm, int32_t* cpos);
/* Host-side structural check of a (host-resident) graph (its SELL forms too, when present). */
int msgat_graph_validate(const msgat_graph_t* host_graph);
/* SELL form of a CSR (ptr = rowptr, idx = col, perm = NULL) or CSC (ptr = colptr, idx = crow,
 * perm = cperm) structure, all HOST pointers: first the sizes, then the arrays (see msgat_sell_t;
 * spos may be NULL; sidx must have n_pos + MSGAT_SELL_SLACK entries; n <= 65535).  Pure index work: replaces
 * nothing in the reference, it re-orders the non-zeros of the mask of attention.py:36 for
 * coalesced reads. */
int msgat_graph_sell_count(const int32_t* ptr, int32_t n, int32_t* n_slices_out, int32_t* n_pos_out,
                           int32_t* pair_trips_out);
int msgat_graph_sell_build(const int32_t* ptr, const int32_t* idx, const int32_t* perm, int32_t n,
                           int32_t nnz, int32_t n_slices, int32_t n_pos, int32_t* slice_off,
                           int32_t* lane_row, uint16_t* sidx, int32_t* ssrc, int32_t* spos);

/* ---- device: fused entry points --------------------------------------------------
 * msgat_gacn_forward replaces attention.py:33-36 (+ msgat.py:27-28 when Co > 0).
 * msgat_gacn_backward replaces the autograd of those lines (adj gets no gradient,
 * msgat.py:190). */
size_t msgat_edge_scratch_floats(const msgat_shape_t* shape, const msgat_graph_t* graph);
/* Scratch of the two dense passes over all N columns of a row (the softmax denominator of attention.py:34 and its
 * backward).  From N = 1536 nodes (T = 12) they run on the bf16 / fp16 matrix core with every fp32 operand split into
 * bf16 / fp16 terms (fp32 accuracy, fp32 accumulate) and first write those operand "images" here; 0 below that.
 * msgat_gacn_backward / msgat_attention_backward carry their share inside their workspace. */
size_t msgat_dense_scratch_bytes(const msgat_shape_t* shape);
int msgat_gacn_forward(const msgat_shape_t* shape, const msgat_graph_t* graph,
                       const msgat_fwd_t* io, void* stream);
size_t msgat_bwd_workspace_bytes(const msgat_shape_t* shape, const msgat_graph_t* graph);
/* 1 when msgat_gacn_backward reads a channel-sliced dz in place for this shape and graph (dz_group_channels), 0 when
 * the caller must pass a contiguous copy. */
int msgat_bwd_accepts_strided_dz(const msgat_shape_t* shape, const msgat_graph_t* graph);
int msgat_gacn_backward(const msgat_shape_t* shape, const msgat_graph_t* graph,
                        const msgat_bwd_t* io, void* stream);

/* ---- device: the individual stages (exposed for tests, profiling and bench.py) ----
 * Each is what the fused entry points enqueue, in order. */

/* attention.py:33 (and msgat.py:27 when projecting first): q = sum_c alpha_c x_c and,
 * if u != NULL, u[g,o] = sum_c W[r,o,c] x[g,c]. */
int msgat_stage_project(const msgat_shape_t* shape, const float* x, const float* alpha,
                        const float* W, float* q, float* u, void* stream);
/* attention.py:34 + the edge part of :36: kW = q Wg, lse = row log-sum-exp over ALL N
 * columns of kW q^T (log2 units), pq = softmax @ q (optional), E = softmax * adj at the edges (Ec, optional: the
 * same values in CSC order).  dense_scratch: msgat_dense_scratch_bytes() bytes (NULL when that is 0). */
int msgat_stage_scores(const msgat_shape_t* shape, const msgat_graph_t* graph,
                       const float* q, const float* Wg, float* kW, float* lse, float* pq,
                       float* E, float* Ec, void* dense_scratch, void* stream);
/* attention.py:36: v[g,c,n,:] = sum_{e in row n} E[g,e] u[g,c,col_e,:] over Cu channels.
 * edge_scratch: msgat_edge_scratch_floats() floats (NULL when that is 0). */
int msgat_stage_aggregate(const msgat_shape_t* shape, const msgat_graph_t* graph,
                          int32_t Cu, const float* u, const float* E, float* v, float* edge_scratch,
                          void* stream);
/* attention.py:36 + msgat.py:27-28 for C <= Co: y = aggregate(x) (stored if y != NULL),
 * z[g,o] = sum_c W[r,o,c] y[g,c]. */
int msgat_stage_aggregate_project(const msgat_shape_t* shape, const msgat_graph_t* graph,
                                  const float* x, const float* E, const float* W, float* y,
                                  float* z, void* stream);

/* The dense column pass of the backward (autograd of the softmax of attention.py:34): given the row sums
 * delta[G,N] of the edge gradients and the edge gradients gE[G,nnz] (CSR order),
 *     dq[g,m,:] += sum_{e -> m} gE_e kW[row_e,:] - sum_n P[n,m] delta[n] kW[n,:],   P = 2^(S log2e - lse)
 * re-creating P with the forward's operands and k order (bit-identical to the forward's).  dq is updated in place.
 * dense_scratch: msgat_dense_scratch_bytes() bytes (NULL when that is 0). */
int msgat_stage_dense_column_pass(const msgat_shape_t* shape, const msgat_graph_t* graph, const float* q,
                                  const float* kW, const float* lse, const float* delta, const float* gE,
                                  float* dq, void* dense_scratch, void* stream);

/* Backward building blocks (also what msgat_gacn_backward enqueues).
 * msgat_stage_mix:      out[g,co,p] = sum_ci M[r,..] in[g,ci,p] (+ addvec[r,co] extra[g,p]);
 *                       M is [R,Co,Ci] (m_in_major = 0) or [R,Ci,Co] (m_in_major = 1, i.e. W^T applied).
 *                       P = N*T positions.  dy = W^T dz and dx = W^T du + alpha (x) dq use it.
 * msgat_stage_contract: dst[r,a,c] = sum_{g in r, p} A[g,a,p] B[g,c,p]; if Aextra != NULL it supplies
 *                       row a = Ca-1 (shape [G,P]) and A holds the other Ca-1 rows.  The first n0
 *                       outputs of a relation go to dst0, the next n1 to dst1 (dW and dalpha).
 *                       `partials` needs msgat_contract_partial_floats() floats. */
int msgat_stage_mix(const msgat_shape_t* shape, int32_t Ci, int32_t Co, const float* in, const float* M,
                    int32_t m_in_major, const float* addvec, const float* extra, float* out, void* stream);
size_t msgat_contract_partial_floats(const msgat_shape_t* shape, int32_t Ca, int32_t Cb);
int msgat_stage_contract(const msgat_shape_t* shape, int32_t Ca, int32_t Cb, const float* A,
                         const float* Aextra, const float* B, float* partials, float* dst0, int32_t n0,
                         float* dst1, int32_t n1, void* stream);
/* msgat_stage_project_backward: the last stage of the PROJ_FIRST backward as msgat_gacn_backward enqueues it --
 *   dW[r,o,c] = sum du[g,o,p] x[g,c,p],  dalpha[r,c] = sum dq[g,p] x[g,c,p],  dx = W^T du + alpha (x) dq
 * (the autograd of msgat.py:27 and attention.py:33) in ONE pass over du, dq and x where the shape has a fused form,
 * as msgat_stage_contract + msgat_stage_mix otherwise.  partials: msgat_contract_partial_floats(shape, Co + 1, C). */
int msgat_stage_project_backward(const msgat_shape_t* shape, const float* du, const float* dq, const float* x,
                                 const float* W, const float* alpha, float* partials, float* dW, float* dalpha,
                                 float* dx, void* stream);

/* ---- device: the producer of every GACN input -- LayerNorm over the timestep axis ----
 * Replaces nn.LayerNorm([n_timesteps]) of the callers (src/models/msgat.py:114 applied at :122,
 * :152 applied at :158): x is [rows, T] contiguous (rows = B*C*N), weight / bias are [T] or NULL
 * (identity), eps as torch's (1e-5 in the reference), biased variance.
 * Backward re-derives mean / rstd from x (nothing is saved but x): dx [rows,T], dweight / dbias [T]
 * (either may be NULL) summed in a fixed order through `partials`
 * (msgat_layernorm_partial_floats() floats).  dx_add (may be NULL) is added to dx in the same pass: the
 * gradient that reached x along its other path -- MEAM's residual convolution reads the block input too
 * (msgat.py:130) -- so autograd's separate accumulation pass over the activation disappears.
 * relu_mask != 0: x is the output of a ReLU (MEAM's tail, msgat.py:131) whose backward is applied here,
 * dx = 0 where x <= 0, instead of in a pass of its own (the caller then skips that ReLU's own mask).
 * T in {4, 8, 12, 16}.
 *
 * R ("relations") in this and the following entry points: the number of parameter sets evaluated in one
 * launch.  The reference loops over its components (src/models/msgat.py:204), each with its own weights;
 * here the components ride on the leading (batch) axis, relation-major -- rows / slabs / groups of relation r
 * are contiguous and R divides their count -- and every parameter gets a leading [R] axis (weight [R,T],
 * w [R,N], bias [R,Co], W [R,T_out,T,1,C]).  R = 1 is the single-module case. */
int msgat_layernorm_forward(const float* x, const float* weight, const float* bias, float* y,
                            int64_t rows, int32_t T, float eps, int32_t R, void* stream);
size_t msgat_layernorm_partial_floats(int64_t rows, int32_t T, int32_t R);
int msgat_layernorm_backward(const float* x, const float* weight, const float* dy, const float* dx_add,
                             float* dx, float* dweight, float* dbias, float* partials, int64_t rows,
                             int32_t T, float eps, int32_t R, int32_t relu_mask, void* stream);
/* msgat_layernorm_forward_pooled: msgat_layernorm_forward that also leaves the node pooling of its OUTPUT,
 * pooled[s,t] = sum_n pool_w[n] y[s,n,t] over y's [N,T] slabs (ChannelAttention's pooled signal, attention.py:89, on MEAM's
 * normalised input, msgat.py:122-125): a wave's 64 rows lie in at most two slabs (N >= 64, else MSGAT_ERR_UNSUPPORTED), every
 * 64-row trip leaves its share of both and a small second launch adds a slab's shares in trip order -- instead of
 * msgat_node_pool reading back the tensor that was just written.  pool_w [R,N], pooled [rows / N, T];
 * partials: msgat_layernorm_pool_partial_floats(rows, T, R) floats. */
size_t msgat_layernorm_pool_partial_floats(int64_t rows, int32_t T, int32_t R);
int msgat_layernorm_forward_pooled(const float* x, const float* weight, const float* bias, float* y, const float* pool_w,
                                   int32_t N, float* pooled, float* partials, int64_t rows, int32_t T, float eps, int32_t R,
                                   void* stream);
/* msgat_layernorm_backward_pooled: msgat_layernorm_backward for a LayerNorm whose output y [.., N, T] ALSO fed a node
 * pooling p[s,t] = sum_n pool_w[n] y[s,n,t] over its [N,T] slabs (ChannelAttention's pooled signal, attention.py:89, taken
 * of MEAM's normalised input, msgat.py:122-125): that consumer's gradient pool_w[n] dpooled[s,t] is rank one and is added
 * to dy row by row inside this pass, instead of by msgat_node_pool_grad_signal in a pass of its own over the activation.
 * pool_w [R,N] (one set per relation), dpooled [rows / N, T]; everything else as msgat_layernorm_backward.
 * dpool_w [R,N] (optional, with dpool_rows: `rows` floats of scratch): the pooling weights' gradient
 * sum_s y[s,n,:] . dpooled[s,:] out of the same pass (y is rebuilt from x, weight and `bias`, the LayerNorm's own), instead
 * of msgat_node_pool_grad_weight's pass over the stored y. */
int msgat_layernorm_backward_pooled(const float* x, const float* weight, const float* bias, const float* dy,
                                    const float* dx_add, const float* pool_w, const float* dpooled, int32_t N, float* dx,
                                    float* dweight, float* dbias, float* dpool_w, float* dpool_rows, float* partials,
                                    int64_t rows, int32_t T, float eps, int32_t R, int32_t relu_mask, void* stream);

/* ---- device: the temporal and channel branches of MEAM ----
 * Building blocks for TACN (src/models/msgat.py:57-80, TemporalAttention attention.py:58-66) and CACN
 * (msgat.py:83-100, ChannelAttention attention.py:88-94) and for the block's residual tail
 * (msgat.py:130-131).  The tiny [T,T] / [C,C] attention matrices are built by the caller from the
 * pooled signals; these entry points are the passes over the [B,C,N,T] activations.
 *
 * msgat_stage_mix_epilogue: msgat_stage_mix plus out = relu?(... + bias[r*bias_per_relation*Co + co]
 *     + add[g,co,p]): a 1x1 convolution with bias when R = 1; a per-sample matrix (CACN's
 *     Wconv att_b) when R = batch, Bg = 1; the residual tail relu(cat(branches) + res(x)) with `add`.
 * msgat_time_mix: forward (backward = 0)
 *         dst[g,o,n,t] = bias[o] + sum_k sum_i A[ga,k,t,i] src[g,k*Co+o,n,i]      src [G,K*Co,N,T]
 *     backward (backward = 1): dst[g,k*Co+o,n,i] = sum_t A[ga,k,t,i] src[g,o,n,t]  src [G,Co,N,T]
 *     A is [G,K,T,T] (a_per_group = 1) or [K,T,T] shared; K in {1, 2}.  With A_1 = att_b and
 *     A_0 = att_b shifted down by the dilation this is conv_[1,2],dilated(TemporalAttention(x)) after the
 *     channel mixing src = [W_0; W_1] x; with constant shift matrices it is a plain causal convolution.
 * msgat_time_mix_grad_matrix: dA[g,k,t,i] = sum_{o,n} dout[g,o,n,t] y[g,k*Co+o,n,i];
 *     `partials` needs msgat_time_mix_partial_floats() floats.
 * Gradients arriving as channel slices [:, a:b] of a wider contiguous [G, group_stride, N, T] tensor (the block's
 * concatenated gradient) are read in place: src_group_stride / dout_group_stride / (channels, group_stride) name the
 * wider tensor's channel count (0 = the operand is contiguous by itself).
 * msgat_node_pool: pooled[s,t] = sum_n w[n] x[s,n,t] over `slabs` (sample, channel) slabs (attention.py:89);
 *     msgat_node_pool_grad_signal: dx[s,n,t] = w[n] dpooled[s,t] (+ dx_add[s,n,t] when given: the gradient that
 *     reached x along its other path, so autograd's separate accumulation pass over the activation disappears);
 *     msgat_node_pool_grad_weight: dw[n] = sum_{g,c,t} x[g,c,n,t] dpooled[g,c,t] (partials:
 *     msgat_node_pool_partial_floats()).
 * The channel pooling sum_c alpha_c x[b,c] (attention.py:59) is msgat_stage_project with W = u = NULL. */
int msgat_stage_mix_epilogue(const msgat_shape_t* shape, int32_t Ci, int32_t Co, const float* in, const float* M,
                             int32_t m_in_major, const float* bias, int32_t bias_per_relation, const float* add,
                             int32_t relu, float* out, void* stream);
int msgat_time_mix(const float* src, const float* A, int32_t a_per_group, const float* bias, float* dst,
                   int32_t G, int32_t Co, int32_t K, int32_t N, int32_t T, int32_t backward, int32_t R,
                   int32_t src_group_stride, void* stream);
size_t msgat_time_mix_partial_floats(int32_t G, int32_t K, int32_t T);
/* msgat_causal_conv: a causal dilated [1,2] convolution -- nn.Conv2d(Ci, Co, [1,2], padding=[0,d], dilation=[1,d])
 * followed by Chomp(d), msgat.py:69-74: every TACN layer after the first -- in ONE pass over its input, with
 * taps [R, 2*Co, Ci] = [W0; W1] (W0 acts on in[t-d], W1 on in[t]; model.TACN.stacked_taps):
 *   backward = 0:  dst[g,co,n,t] = bias[co] + sum_ci W0[co,ci] src[g,ci,n,t-d] + W1[co,ci] src[g,ci,n,t]   src [G,Ci,N,T]
 *   backward = 1:  dst[g,ci,n,t] = sum_co W0[co,ci] src[g,co,n,t+d] + W1[co,ci] src[g,co,n,t]              src [G,Co,N,T]
 * (terms whose timestep falls outside [0,T) are zero).  It replaces msgat_mix_segments (Ci -> 2*Co), the [G,2*Co,N,T]
 * intermediate and msgat_time_mix with constant shift matrices: the shifted operand is an unaligned 16-byte load of
 * the same row of T plus a per-lane element mask.  bias [Co] or (bias_per_relation) [R,Co], forward only.
 * src_group_stride: src is a channel slice of a [G, src_group_stride, N, T] tensor (0: contiguous by itself).
 * msgat_causal_conv_fused(Ci, Co) = 1 when both directions have a one-pass form for these widths (else the entry point
 * returns MSGAT_ERR_UNSUPPORTED and the caller runs the two passes). */
/* msgat_causal_conv_grad_weight: dtaps[r, k*Co + co, ci] = sum_{g in r, n, t} dout[g,co,n,t + (k == 0 ? d : 0)] h[g,ci,n,t]
 * ([R, 2*Co, Ci + with_ones]; with_ones: column Ci = the sums of the gradient rows, whose tap-1 half [Co, 2*Co) is the
 * bias gradient) -- the gradient rows [dout[t+d]; dout] are virtual (time-shifted reads of dout inside the contraction's
 * staging loads), nothing [G,2*Co,N,T] is written.  partials: msgat_contract_segments_partial_floats(R, 2*Co, Ci + with_ones).
 * dout may be a channel slice of a [G, dout_group_stride, N, T] tensor (0: contiguous). */
int msgat_causal_conv_grad_weight(const float* dout, int32_t dout_group_stride, const float* h, float* partials, float* dtaps,
                                  int32_t R, int32_t Bg, int32_t Ci, int32_t Co, int32_t N, int32_t T, int32_t dilation,
                                  int32_t with_ones, void* stream);
int msgat_causal_conv_fused(int32_t Ci, int32_t Co);
int msgat_causal_conv(const float* src, const float* taps, const float* bias, int32_t bias_per_relation, float* dst,
                      int32_t R, int32_t Bg, int32_t Ci, int32_t Co, int32_t N, int32_t T, int32_t dilation,
                      int32_t backward, int32_t src_group_stride, void* stream);

int msgat_time_mix_grad_matrix(const float* dout, const float* y, float* dA, float* partials, int32_t G,
                               int32_t Co, int32_t K, int32_t N, int32_t T, int32_t dout_group_stride, void* stream);
int msgat_node_pool(const float* x, const float* w, float* pooled, int64_t slabs, int32_t N, int32_t T,
                    int32_t R, int32_t channels, int32_t group_stride, void* stream);
int msgat_node_pool_grad_signal(const float* w, const float* dpooled, const float* dx_add, float* dx,
                                int64_t slabs, int32_t N, int32_t T, int32_t R, void* stream);
size_t msgat_node_pool_partial_floats(int32_t G, int32_t C, int32_t N);
int msgat_node_pool_grad_weight(const float* x, const float* dpooled, float* dw, float* partials, int32_t G,
                                int32_t C, int32_t N, int32_t T, int32_t R, void* stream);

/* ---- device: channel axes assembled from several tensors ("segments") ----
 * A segment is `channels` channels of a [G, group_stride, N, T] tensor starting at `ptr` (group_stride = 0
 * means `channels`: the whole tensor; a larger value makes it a channel slice of a wider tensor).  The
 * segments of a list are concatenated along the channel axis -- without a concatenation pass:
 *   msgat_mix_segments:      out_segments = relu?(M cat(in_segments) + bias + cat(add_segments)); M is
 *       [R, Co_total, Ci_total] (or [R, Ci_total, Co_total] when m_in_major).  One pass reads cat(...) and
 *       writes each output range to its own tensor: MEAM's tail (msgat.py:123-131) without the cat, and all
 *       channel mixings of one activation (CACN's per-sample matrix, TACN's first convolution, GACN's
 *       projection and the two alpha poolings) in ONE pass; m_in_major = 1 is its backward (one dx).
 *   msgat_contract_segments: dst[r,a,c] = sum_{g in r, p} cat(A_segments)[g,a,p] B[g,c,p]; with_ones = 1 appends a
 *       virtual channel of ones to B: dst is [R, Ca, Cb+1] and dst[r,a,Cb] = sum_{g,p} A[g,a,p] -- the bias gradient of
 *       a 1x1 convolution out of the pass that computes its weight gradient (size the partials for Cb+1).
 *       with_ones = 2 (ABI 6; also msgat_contract_mix_segments and msgat_causal_conv_grad_weight): the same sums delivered
 *       apart -- dst holds the plain [R, Ca, Cb] matrix followed by the ones column as [R, Ca], R*Ca*(Cb+1) floats in all
 *       -- so weight and bias gradient are two contiguous tensors without a copy each.
 * At most 6 segments per list. */
typedef struct {
  float* ptr;
  int32_t channels;
  int32_t group_stride;
} msgat_seg_t;
int msgat_mix_segments(int32_t R, int32_t Bg, int32_t N, int32_t T, const msgat_seg_t* in, int32_t n_in,
                       const float* M, int32_t m_in_major, const float* bias, int32_t bias_per_relation,
                       const msgat_seg_t* add, int32_t n_add, int32_t relu, const msgat_seg_t* out,
                       int32_t n_out, void* stream);
size_t msgat_contract_segments_partial_floats(int32_t R, int32_t Ca, int32_t Cb);
int msgat_contract_segments(int32_t R, int32_t Bg, int32_t N, int32_t T, const msgat_seg_t* A, int32_t n_a,
                            const float* B, int32_t Cb, int32_t with_ones, float* partials, float* dst, void* stream);
/* msgat_contract_mix_segments: the whole backward of a 1x1 convolution y = M x (+ bias) in ONE pass over its
 * operands -- the autograd of the reference's nn.Conv2d(kernel_size=1) / GACN projection calls, msgat.py:27,63-66,
 * 92,116.  With A = cat(A_segments) the gradient at y [G,Ca,N,T], B = x [G,Cb,N,T], M [R,Ca,Cb]:
 *   dst[r,a,c]     = sum_{g in r, p} A[g,a,p] B[g,c,p]      (dM; with_ones: column Cb = sum A, the bias gradient)
 *   mixout[g,c,p]  = sum_a M[r,a,c] A[g,a,p]                (dx, [G,Cb,N,T] contiguous)
 * i.e. msgat_contract_segments + msgat_mix_segments(m_in_major = 1) with A streamed once; shapes without a fused
 * form run those two passes.  partials: msgat_contract_mix_partial_floats(R, Bg, N, T, Ca, Cb + with_ones). */
/* partials of msgat_contract_mix_segments for Cb = channels of B INCLUDING the virtual one (round 5: with Cb <= 4 and one
 * A segment the call is ONE pass over A -- dM | dbias and dx together, the form of the AGG_FIRST backward -- whose
 * per-block partials outnumber those of the general form; msgat_contract_segments_partial_floats stays valid for
 * msgat_contract_segments) */
size_t msgat_contract_mix_partial_floats(int32_t R, int32_t Bg, int32_t N, int32_t T, int32_t Ca, int32_t Cb);
int msgat_contract_mix_segments(int32_t R, int32_t Bg, int32_t N, int32_t T, const msgat_seg_t* A, int32_t n_a,
                                const float* B, int32_t Cb, int32_t with_ones, const float* M, float* partials,
                                float* dst, float* mixout, void* stream);
/* Host only, touches no device: the name of the kernel form a [Ca x Cb (+1 with_ones)] contraction over rows of
 * n_positions = N*T takes -- with_mix = 1: as msgat_contract_mix_segments / msgat_stage_project_backward would run it,
 * " + projection pass" appended when no one-pass form covers the shape; with_mix = 0: msgat_contract_segments.
 * E.g. "k_chanpair_glds<7,5,64,3,2>" (LDS-DMA ring, [112 x 80] block, 64-position tiles, wave roles split),
 * "k_chanpair_glds<9,4,64,3,2> nzb=2" (two z-blocks over B), "k_chanpair_mfma<3,2,true,256>" (register-staged).
 * The string comes from the launchers' own selection code (csrc/mfma.hip), so it cannot drift from what runs; the forms
 * exist for the channel counts of the reference's three models (main.py:17, msgat.py:220-229), other shapes fall back
 * to the register-staged kernels.  buf: at least 128 bytes. */
int msgat_contract_form_name(int32_t Ca, int32_t Cb, int32_t with_ones, int32_t n_positions, int32_t with_mix,
                             char* buf, int32_t buflen);

/* ---- device: the attention core on already projected features ----
 * Forward = msgat_stage_scores(q) + msgat_stage_aggregate(u) (above).  Backward of exactly that pair, for
 * callers that produced u and q themselves (e.g. in a merged channel-mixing pass):
 *   du = E^T dv  [G,Cu,N,T],   dq = total gradient at q  [G,N,T],   dWg [R,T,T].
 * shape->C is Cu (the channels of u), shape->Co is ignored.  du must not overlap u or dv (du and the edge gradients
 * come out of one pass that reads both). */
size_t msgat_attention_bwd_workspace_bytes(const msgat_shape_t* shape, const msgat_graph_t* graph);
/* dv_group_channels: as msgat_bwd_t.dz_group_channels, for dv (0 = contiguous); non-zero only where
 * msgat_attention_bwd_accepts_strided_dv() returns 1.  Ec: E in CSC order from msgat_stage_scores, or NULL. */
int msgat_attention_bwd_accepts_strided_dv(const msgat_shape_t* shape, const msgat_graph_t* graph);
int msgat_attention_backward(const msgat_shape_t* shape, const msgat_graph_t* graph, const float* u,
                             const float* dv, int32_t dv_group_channels, const float* q, const float* kW,
                             const float* lse, const float* pq, const float* E, const float* Ec, const float* Wg,
                             float* du, float* dq, float* dWg, void* workspace, size_t workspace_bytes, void* stream);

/* ---- device: the prediction head of a component ----
 * TPC's Conv2d(T_in -> T_out, kernel [1, C]) over the transposed activation (src/models/msgat.py:153,
 * applied :158-159):  out[b,n,o] = bias[o] + sum_c sum_t W[o,t,0,c] x[b,c,n,t].
 * x [B,C,N,T]; W in the convolution's own layout [T_out,T,1,C]; out [B,N,T_out] (what the reference has after
 * its squeeze + transpose); T_out <= 16; `partials` (msgat_head_forward_partial_floats floats) is scratch: the weights
 * re-ordered to [R,C,T,16] by a small pre-pass, so that the kernel stages them with contiguous copies.  Backward: dx [B,C,N,T]; dWc [R,C,T_out,T] (the caller permutes to
 * the convolution layout); partial buffers sized by the *_partial_floats queries. */
size_t msgat_head_forward_partial_floats(int32_t B, int32_t C, int32_t N, int32_t T_out);
int msgat_head_forward(const float* x, const float* W, const float* bias, float* out, float* partials,
                       int32_t B, int32_t C, int32_t N, int32_t T, int32_t T_out, int32_t R, void* stream);
int msgat_head_grad_signal(const float* dout, const float* W, float* dx, int32_t B, int32_t C, int32_t N,
                           int32_t T, int32_t T_out, int32_t R, void* stream);
size_t msgat_head_grad_weight_partial_floats(int32_t C, int32_t T, int32_t T_out, int32_t R);
int msgat_head_grad_weight(const float* dout, const float* x, float* dWc, float* partials, int32_t B, int32_t C,
                           int32_t N, int32_t T, int32_t T_out, int32_t R, void* stream);
/* msgat_head_forward_ln: msgat_head_forward on LayerNorm(x) with x the LayerNorm's INPUT (ln_weight, ln_bias [R,T] or
 * NULL): a row's T values lie in neighbouring lanes of the kernel's load layout, so it normalises what it loads (two lane
 * sums per row) -- one pass over x where msgat_layernorm_forward + msgat_head_forward made three (read, write, read).
 * normalised [B,C,N,T] or NULL: receives LayerNorm(x) when something else needs it (training: msgat_head_grad_weight
 * reads it); with NULL -- inference -- the LayerNorm output, an activation only the head reads (msgat.py:158-159), is never
 * written.  Other buffers as for msgat_head_forward. */
int msgat_head_forward_ln(const float* x, const float* ln_weight, const float* ln_bias, float eps, const float* W,
                          const float* bias, float* out, float* normalised, float* partials, int32_t B, int32_t C,
                          int32_t N, int32_t T, int32_t T_out, int32_t R, void* stream);
/* msgat_layernorm_head_backward: the head's input gradient (msgat_head_grad_signal) AND the backward of the LayerNorm in
 * front of it (msgat.py:158-159: fc(ln(x).transpose(1, 3))) in ONE pass over x: the [B,C,N,T] gradient between the two is
 * built row by row in registers and consumed on the spot.  x is the LayerNorm's INPUT; relu_mask as in
 * msgat_layernorm_backward; dln_weight / dln_bias [R,T] (either may be NULL);
 * partials: msgat_layernorm_head_backward_partial_floats(B, C, N, T) floats. */
size_t msgat_layernorm_head_backward_partial_floats(int32_t B, int32_t C, int32_t N, int32_t T);
int msgat_layernorm_head_backward(const float* dout, const float* W, const float* x, const float* ln_weight, float* dx,
                                  float* dln_weight, float* dln_bias, float* partials, int32_t B, int32_t C, int32_t N,
                                  int32_t T, int32_t To, int32_t R, float eps, int32_t relu_mask, void* stream);

/* ---- device: the tiny attention matrices of a MEAM block, one launch each way ----
 * In the reference each is a chain of batched [T x T] / [C x C] matmuls, a softmax, transposes, pads and stacks
 * (~20 eager launches forward, ~30 backward per block).  One workgroup per (relation, sample) group here;
 * parameter gradients are summed over a relation's groups in a fixed order through `partials`.
 *   channel attention (attention.py:88-94) folded with CACN's 1x1 convolution (msgat.py:93-94):
 *       att[g] = softmax_rows((p_g Wc_r) p_g^T) [C,C],  Mc[g] = conv_r att[g] [cb,C];   p = pooled [G,C,T]
 *       (the node-weighted sums of msgat_node_pool), Wc [R,T,T], conv [R,cb,C].  Mc is the per-sample channel
 *       matrix msgat_mix_segments applies.  Backward: dpooled [G,C,T], dWc [R,T,T], dconv [R,cb,C] from dMc.
 *   temporal attention (attention.py:58-66) as the taps of TACN's first causal convolution (msgat.py:66-74):
 *       left = q_g^T Wt1_r^T, right = q_g^T Wt2_r^T [T,K] (saved in lr [G,2,T,K]),  att[g] = softmax_rows(left right^T),
 *       taps[g,1] = att[g],  taps[g,0] = att[g] shifted down by `dilation` rows (rows < dilation zero);
 *       q = pooled [G,N,T] (the alpha-weighted channel sums), Wt1 / Wt2 [R,K,N], 2 T K <= 256.  taps [G,2,T,T] is what
 *       msgat_time_mix applies (K = 2).  Backward: dpooled [G,N,T], dWt1 / dWt2 [R,K,N] from dtaps. */
int msgat_channel_attention_forward(const float* pooled, const float* Wc, const float* conv, float* att, float* Mc,
                                    int32_t G, int32_t R, int32_t C, int32_t cb, int32_t T, void* stream);
size_t msgat_channel_attention_partial_floats(int32_t G, int32_t C, int32_t cb, int32_t T);
int msgat_channel_attention_backward(const float* dMc, const float* att, const float* pooled, const float* Wc,
                                     const float* conv, float* dpooled, float* dWc, float* dconv, float* partials,
                                     int32_t G, int32_t R, int32_t C, int32_t cb, int32_t T, void* stream);
int msgat_temporal_attention_forward(const float* pooled, const float* Wt1, const float* Wt2, float* lr, float* att,
                                     float* taps, int32_t G, int32_t R, int32_t N, int32_t K, int32_t T,
                                     int32_t dilation, void* stream);
size_t msgat_temporal_attention_partial_floats(int32_t G, int32_t K, int32_t N);
int msgat_temporal_attention_backward(const float* dtaps, const float* att, const float* lr, const float* pooled,
                                      const float* Wt1, const float* Wt2, float* dpooled, float* dWt1, float* dWt2,
                                      float* partials, int32_t G, int32_t R, int32_t N, int32_t K, int32_t T,
                                      int32_t dilation, void* stream);

/* ---- device: the tail of a training step ----
 * msgat_huber_metrics: one pass over the prediction replaces engine.py:56 (HuberLoss, loss.py:51-52) and the
 *     metric sums of engine.py:70 / metrics.py:20-35, with nothing read back to the host:
 *       loss[0]  = mean over n of (|e| <= delta ? e^2 / 2 : delta |e| - delta^2 / 2),  e = pred - truth   (fp32)
 *       sums[0] += sum |e|;  sums[1] += 100 sum_{truth > mask_value} |e / truth|;  sums[2] += sum e^2;
 *       sums[3] += loss_weight * loss[0]                                           (fp64 [4], running totals
 *     the caller zeroes per epoch; may be NULL).  loss_weight = 1 for a whole batch; a rank holding n_r of a global
 *     batch's n_b samples passes n_r / n_b, so the sum over ranks is the reference's per-batch mean loss
 *     (engine.py:66-67).  `partials`: msgat_huber_partial_doubles(n) doubles.  Fixed summation order: bitwise
 *     reproducible.
 * msgat_huber_grad: dpred = dloss[0] * clamp(e, -delta, delta) / n  -- the backward of the loss above.
 * msgat_adam_step: torch.optim.Adam's update (engine.py:106: lr 1e-3, weight_decay 5e-4 as L2 on the gradient,
 *     betas, eps outside the square root, bias correction) for ALL parameter tensors in one launch.  The
 *     gradients and both moments are flat fp32 buffers; the parameters stay where they are and are reached
 *     through a chunk table (device arrays): chunk c = chunk_len[c] <= msgat_adam_chunk_elems() elements of
 *     tensor chunk_tensor[c] at chunk_param[c], whose gradient / moments start at element chunk_off[c] of the
 *     flat buffers.  steps[t] = updates tensor t has received so far, advanced by this call for the n_active
 *     tensors listed in active_tensors (torch keeps one step count per parameter and skips parameters without a
 *     gradient); lr[0] = learning rate.  Both are device memory, so a captured launch follows the scheduler.
 *     The hyper-parameters are doubles, like the Python floats the reference's optimizer holds: 1 - beta2 must
 *     be rounded to fp32 once, not computed from an fp32 beta2.
 *     grad_divisor (device, may be NULL): when given, every gradient is divided by grad_divisor[0] on the way in.
 * msgat_gather_scaled: the front half of a data-parallel step (replaces nn.DataParallel's gradient reduce-add,
 *     main.py:52-55): flat[chunk_off[c] + i] = scale * chunk_src[c][i] for chunk_len[c] elements per chunk (device
 *     table, as above, but of GRADIENT pointers) and flat[weight_index] = scale (weight_index < 0: not written).
 *     With scale = the rank's sample count and weight_index = the buffer's last element, ONE all-reduce of `flat`
 *     yields sum_r(w_r g_r) and sum_r(w_r); msgat_adam_step(grad_divisor = flat + weight_index) finishes the mean. */
/* msgat_gate_sum: the model's last line, out = sum_r pred_r * gate_r (msgat.py:203-205) with the gate built on the fly
 *     from the two embedding tables (embeddings.py:36-39: gate[b] = h_ebd(H[b]) + d_ebd(D[b]) viewed [R,N,T_out]):
 *       out[b,e] = sum_r pred[r,b,e] * (h_w[H[b]][r,e] + d_w[D[b]][r,e]),   e < E = N * T_out, summed in the order of r.
 *     pred [R,B,E] (the R components' predictions, relation-major); H, D [B] int64 (torch's index type); h_w [nh, R*E],
 *     d_w [nd, R*E] row-major as nn.Embedding keeps them.  Indices are clamped into [0, nh) / [0, nd) (torch's gather
 *     traps on the device instead).  Static gate (msgat.py:189): H = D = d_w = NULL, nh = 1, h_w = W [R,E].
 * msgat_gate_sum_backward: dpred [R,B,E] = dout * gate, and the DENSE table gradients dh_w [nh, R*E], dd_w [nd, R*E]
 *     (every row written; a row's samples are added in sample order).  Any output may be NULL.
 *     Reference: 5 launches forward, 12 backward (gathers, multiplies, a transposing copy, two zero-fill + scatter
 *     embedding gradients); one each here. */
int msgat_gate_sum(const float* pred, const int64_t* H, const int64_t* D, const float* h_w, const float* d_w, float* out,
                   int32_t R, int32_t B, int64_t E, int32_t nh, int32_t nd, void* stream);
int msgat_gate_sum_backward(const float* dout, const float* pred, const int64_t* H, const int64_t* D, const float* h_w,
                            const float* d_w, float* dpred, float* dh_w, float* dd_w, int32_t R, int32_t B, int64_t E,
                            int32_t nh, int32_t nd, void* stream);
size_t msgat_huber_partial_doubles(int64_t n);
int msgat_huber_metrics(const float* pred, const float* truth, int64_t n, float delta, float mask_value,
                        double* partials, float* loss, double* sums, float loss_weight, void* stream);
int msgat_huber_grad(const float* pred, const float* truth, const float* dloss, int64_t n, float delta,
                     float* dpred, void* stream);
int msgat_adam_chunk_elems(void);
int msgat_adam_step(float* const* chunk_param, const int64_t* chunk_off, const int32_t* chunk_len,
                    const int32_t* chunk_tensor, int32_t n_chunks, const int32_t* active_tensors, int32_t n_active,
                    const float* grad, float* exp_avg, float* exp_avg_sq, float* steps, const float* lr,
                    double beta1, double beta2, double eps, double weight_decay, const float* grad_divisor,
                    void* stream);
int msgat_gather_scaled(const float* const* chunk_src, const int64_t* chunk_off, const int32_t* chunk_len,
                        int32_t n_chunks, float scale, float* flat, int64_t weight_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSGAT_HIP_H */
