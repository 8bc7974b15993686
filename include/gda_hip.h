/*
 * gda_hip.h -- C ABI of libgda_hip.so: the MI355X (gfx950) kernels behind pygda's
 * message-passing / domain-adaptation-loss hot path.
 *
 * The reference (pygda-team/pygda v1.2.1) is pure Python and has no FFI of its own;
 * the arithmetic on this path lives in third-party wheels it calls (PyG
 * MessagePassing.propagate, torch_scatter.scatter_add, add_remaining_self_loops) and
 * in torch elementwise chains (pygda/utils/mmd.py).  Each entry point below names the
 * reference call site(s) whose work it replaces.  pygda_amd/ binds these with ctypes
 * (INTEGRATION.md shows the stub a pygda maintainer would add).
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = invalid argument (GDA_E_*),
 *     >0 = hipError_t of the failing runtime call.  Nothing throws across the ABI.
 *   - all pointers are caller-owned DEVICE pointers unless the name ends in _host;
 *     kernels never allocate: scratch is passed in (size from gda_*_workspace_bytes).
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all
 *     work is enqueued asynchronously on it; the library keeps no mutable state, so
 *     calls are re-entrant and hipGraph-capturable.
 *   - indices inside the library are int32 (N, nnz < 2^31); features are fp32,
 *     row-major with an explicit leading dimension in elements.
 */
#ifndef GDA_HIP_H
#define GDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDA_OK 0
#define GDA_E_NULL (-1)      /* required pointer is NULL */
#define GDA_E_SIZE (-2)      /* negative / overflowing / inconsistent size */
#define GDA_E_WORKSPACE (-3) /* workspace too small */
#define GDA_E_UNSUPPORTED (-4)
#define GDA_E_ALIAS (-5)     /* output aliases an input that must stay intact */
#define GDA_E_RCCL (-100)    /* RCCL failure: status = GDA_E_RCCL - ncclResult_t */

typedef void* gda_stream_t;

int gda_abi_version(void);
/* Human-readable text for a status returned by any function here. */
const char* gda_status_string(int status);

/* ------------------------------------------------------------------------------
 * Graph ingestion: COO edge list -> self-loop merge -> symmetric normalisation ->
 * CSR by destination (forward) + CSR by source (transpose, for the backward pass).
 *
 * Replaces gcn_norm (pygda/nn/prop_gcn_conv.py:64-81: add_remaining_self_loops :72,
 * scatter_add degree over col :78, deg^-1/2 with inf->0 :79-80, w' = dis[row]*w*dis[col]
 * :81) and CachedGCNConv.norm (pygda/nn/cached_gcn_conv.py:88-103, degree over row),
 * plus the edge ordering PyG's propagate scatter relies on.  Both CSRs keep the
 * edges of a row in their original edge order (stable sort), so a sequential row
 * sum reproduces the CPU scatter-add order.
 *
 *   src,dst   [E] int64  edge_index[0], edge_index[1]  (message flows src -> dst)
 *   w         [E] fp32 or NULL (all ones)
 *   fill_value     self-loop weight for nodes without one (1, or 2 for improved)
 *   add_self_loops 0 = edges as given; 1 = PyG add_remaining_self_loops; 2 = existing loops dropped,
 *                  none appended (get_laplacian's remove_self_loops, pygda/nn/dgsda_base.py:128)
 *   normalize 0/1 (0: values are the raw weights)
 *   degree_side    0 = over dst/col (PropGCNConv, GCNConv), 1 = over src/row (CachedGCNConv)
 *   rowptr  [N+1], colidx/val [E+N]  : by-destination CSR, colidx = source node
 *   t_rowptr[N+1], t_colidx/t_val [E+N] : by-source CSR, t_colidx = destination node
 * nnz (= kept edges + N) is rowptr[N].
 * ---------------------------------------------------------------------------- */
size_t gda_graph_workspace_bytes(int64_t E, int64_t N);
int gda_build_csr_norm(const int64_t* src, const int64_t* dst, const float* w,
                       int64_t E, int64_t N, float fill_value, int add_self_loops,
                       int normalize, int degree_side,
                       int32_t* rowptr, int32_t* colidx, float* val,
                       int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                       void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* Same, and additionally t_to_fwd [E+N]: for entry k' of the by-source CSR, the position of the
 * same edge in the by-destination CSR -- lets a backward pass that walks the transpose read
 * per-edge quantities (attention coefficients) stored in forward order. */
int gda_build_csr_norm_map(const int64_t* src, const int64_t* dst, const float* w,
                           int64_t E, int64_t N, float fill_value, int add_self_loops,
                           int normalize, int degree_side,
                           int32_t* rowptr, int32_t* colidx, float* val,
                           int32_t* t_rowptr, int32_t* t_colidx, float* t_val, int32_t* t_to_fwd,
                           void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* Expand a CSR back to the COO edge_index the reference returns from gcn_norm, in
 * CSR order; nnz_cap >= rowptr[N] bounds the launch (entries past rowptr[N] untouched). */
int gda_csr_to_coo(const int32_t* rowptr, const int32_t* colidx, int64_t N, int64_t nnz_cap,
                   int64_t* src_out, int64_t* dst_out, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Neighbour aggregation  y = A_hat * x  (CSR SpMM, fp32).
 *
 * Replaces MessagePassing.propagate + message + scatter-add aggregate
 * (pygda/nn/prop_gcn_conv.py:208-210,238; pygda/nn/cached_gcn_conv.py:138,156):
 *   y[i,:] = sum_{k in rowptr[i]..rowptr[i+1]} val[k] * x[colidx[k],:]   (+ bias)
 * accumulated in CSR (= edge) order with separately rounded multiply and add, i.e.
 * the CPU result bit for bit for rows processed by one lane group.
 * `bias` ([d], may be NULL) is added once to the final output (prop_gcn_conv.py:212-213,
 * cached_gcn_conv.py:172-174).  y must not alias x.
 *
 * gda_spmm_csr_kstep_f32 applies the operator K>=1 times (the prop_nums loop
 * prop_gcn_conv.py:208-210) ping-ponging between y and tmp ([n_rows, ldy], needed
 * when K>1); x is left intact.  The backward of K steps is the same call on the
 * by-source CSR with the incoming gradient.
 * ---------------------------------------------------------------------------- */
/* Optional load balancing for power-law graphs: rows with more than `threshold` entries are
 * processed as chunks of `threshold` entries (partials in `scratch [n_chunks, d]`, then an
 * ordered per-row reduce).  Built once per graph by gda_row_split_build; all pointers device. */
typedef struct gda_row_split {
    int32_t threshold;
    int32_t n_long, n_chunks;
    const int32_t* long_rows;       /* [n_long]     */
    const int32_t* long_chunk_ptr;  /* [n_long + 1] */
    const int32_t* chunk_long;      /* [n_chunks]   */
    float* scratch;                 /* [n_chunks, d] caller-owned, reused across calls */
    const int32_t* counts_dev;      /* NULL, or the device {n_long, n_chunks} written by gda_row_split_build:
                                       n_long / n_chunks above then hold the CAPACITIES (grid bounds) and
                                       the kernels read the live counts themselves -- no host read-back,
                                       for graphs that live one step (sampled mini-batches) */
} gda_row_split;

/* Find the long rows of a CSR and lay out their chunks.  Outputs must hold the upper bounds
 * cap_long = nnz_cap/threshold + 1 and cap_chunks = 2*nnz_cap/threshold + 2 entries;
 * counts_out [2] (device) receives {n_long, n_chunks}: the one data-dependent pair the host
 * reads back, once per graph. */
size_t gda_row_split_workspace_bytes(int64_t n_rows);
int gda_row_split_build(const int32_t* rowptr, int64_t n_rows, int32_t threshold,
                        int32_t* long_rows, int32_t* long_chunk_ptr, int32_t* chunk_long,
                        int32_t* counts_out, void* workspace, size_t workspace_bytes,
                        gda_stream_t stream);

/* K-step aggregation with optional row splitting (split may be NULL). */
int gda_spmm_csr_split_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                           int64_t n_rows, int64_t d, int K, const float* x, int64_t ldx,
                           float* y, int64_t ldy, float* tmp, const float* bias,
                           const gda_row_split* split, gda_stream_t stream);

int gda_spmm_csr_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                     int64_t n_rows, int64_t d, const float* x, int64_t ldx,
                     float* y, int64_t ldy, const float* bias, gda_stream_t stream);
int gda_spmm_csr_kstep_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                           int64_t n_rows, int64_t d, int K, const float* x, int64_t ldx,
                           float* y, int64_t ldy, float* tmp, const float* bias,
                           gda_stream_t stream);
/* One step with the result written TRANSPOSED: yT[c * ldyT + i] = (A x)[i][c] (+ bias[c]), ldyT >= n_rows -- the same
 * row walk and sums as gda_spmm_csr_f32 (no hub-row split: callers with rows beyond the split threshold use
 * gda_spmm_csr_split_f32 + gda_transpose_f32).  The projection of sparse input features (pygda/nn/prop_gcn_conv.py:205
 * on a bag-of-words matrix) hands its result to gda_kstep_lds_colmajor_f32 in that kernel's own layout. */
int gda_spmm_csr_tout_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                          int64_t n_rows, int64_t d, const float* x, int64_t ldx,
                          float* yT, int64_t ldyT, const float* bias, gda_stream_t stream);

/* K aggregation steps on a sampled sub-graph (a NeighborLoader batch, pygda/models/a2gnn.py:260-277, numbered seeds
 * first then in discovery order): the nodes found in the last hop are never expanded, so rows [n_int, n_rows) of the
 * normalised adjacency hold their unit self loop only and K steps leave them unchanged.  Only rows [0, n_int) are
 * recomputed per step (transposed = 0: leaf columns are read from x; same values as gda_spmm_csr_kstep_f32, signed
 * zeros aside); the transposed operator (backward) runs its K steps on the interior rows, sums their inputs in `sacc`
 * and finishes the leaves in one pass, y_L = x_L + A_IL^T (h_0 + ... + h_{K-1}) -- K full steps up to fp32 summation
 * order.  Rows must be short (<= 128 entries: no hub splitting here; sampled batches hold fan-out + 1 per row).
 * x, y: [n_rows, d] contiguous; tmp [n_int, d] (K > 1), sacc [n_int, d] (transposed). */
int gda_spmm_csr_interior_kstep_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                    int64_t n_rows, int64_t n_int, int64_t d, int K, int transposed,
                                    const float* x, float* y, float* tmp, float* sacc, const float* bias,
                                    gda_stream_t stream);

/* The same K steps with ONE launch for the step loop (round 5; csrc/gda_interior.inc): per feature column the interior
 * block (<= gda_interior_max_rows() = 16,384 rows: 1024 seeds at fan-out 15) is a 64 KB vector that stays in the LDS of one workgroup for all K
 * steps,  y_I <- A_II y_I + c  with the leaf columns' contribution c = A_IL x_L formed once; a call is 3 launches forward
 * (c + transposition | step loop | rows back + leaf copy) and 4 transposed, whatever K is, against K + 2 above.
 *   gda_interior_plan_build: compiles ONE direction of the batch's CSR (rowptr / colidx / val = the by-destination arrays
 *     for the forward operator, the by-source arrays for the transposed one) into the step loop's per-lane register
 *     program ON THE DEVICE -- three small capacity-sized launches, typically on the sampler's stream right after
 *     gda_dsampler_sample; n_int_dev = device int64 (the sampler's counts + 4); plan = gda_interior_plan_bytes() device
 *     bytes, 16-byte aligned; status_dev = device int64[2]: [0] <- > 0: the plan is valid (entries per lane), 0: the batch
 *     does not fit (too many interior rows, more than 1024 x 26 off-diagonal interior entries): keep the call above;
 *     [1] <- the number of off-diagonal interior entries.
 *   gda_interior_kstep_lds_f32: x, y [n_rows, d] contiguous, 16-byte aligned; d % 4 == 0, d <= gda_interior_max_width();
 *     workspace = gda_interior_kstep_lds_workspace_bytes(n_int, d) device bytes.  Same sums as the call above with `sacc`
 *     given (bit for bit when the diagonal entry is the last of its row, as in every CSR this library builds). */
int gda_interior_max_rows(void);
int gda_interior_max_width(void);
size_t gda_interior_plan_bytes(void);
size_t gda_interior_kstep_lds_workspace_bytes(int64_t n_int, int64_t d);
int gda_interior_plan_build(const int32_t* rowptr, const int32_t* colidx, const float* val,
                            const int64_t* n_int_dev, void* plan, size_t plan_bytes, int64_t* status_dev,
                            gda_stream_t stream);
int gda_interior_kstep_lds_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                               int64_t n_rows, int64_t n_int, int64_t d, int K, int transposed,
                               const void* plan, const float* x, float* y, const float* bias,
                               void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* The forward call with the conv layer's activation (pygda/nn/a2gnn_base.py:135-138: relu, then inverted dropout) in its
 * epilogue: y0 = dropout_{site0}(relu(A^K x + bias)); y1 (may be NULL) = dropout_{site1}(relu(the same values)) -- a second,
 * independent draw for the trainer's second pass over a shared layer-0 output.  The values gda_relu_dropout_fwd_f32 would
 * produce from gda_interior_kstep_lds_f32's result with the same (seed, step, site); the pre-activation is not stored
 * (gda_relu_dropout_bwd_f32 needs y > 0 only).  p <= 0: relu only. */
int gda_interior_kstep_lds_act_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                                   int64_t n_rows, int64_t n_int, int64_t d, int K, const void* plan,
                                   const float* x, float* y0, float* y1, const float* bias,
                                   float p, uint64_t seed, const int64_t* step_dev, uint32_t site0, uint32_t site1,
                                   void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * K-step aggregation in ONE launch for graphs whose feature columns fit a CU's LDS
 * (n_rows <= gda_kstep_max_rows() = 16320; the citation-graph regime), csrc/gda_kstep.hip.
 *
 * Replaces the same prop_nums loop (pygda/nn/prop_gcn_conv.py:208-210) as gda_spmm_csr_kstep_f32,
 * with the same results bit for bit (per-row sums in CSR order, separately rounded multiply and
 * add): A_hat^K x is independent per feature column, so a workgroup keeps one column of all rows in
 * LDS across the K steps and the graph is compiled once into a per-lane register program.
 * On a graph with a row of more than 48 entries (power-law hubs) the rows longer than 4*S entries are cut into
 * segments of 4*S entries whose sequential
 * sums are added as a fixed balanced tree by a second phase of each step: deterministic, within fp32
 * summation tolerance of the sequential order (the rows of up to 4*S entries stay bit-exact).
 *
 * gda_kstep_plan_host compiles a CSR given as HOST arrays into `plan_host` (caller-owned host
 * memory of gda_kstep_plan_bytes(12) bytes; the first gda_kstep_plan_bytes(slots) are meaningful) and
 * returns `slots` = S | hub_waves << 8 (S = 6, 8, 10 or 12 slots per thread; hub_waves = 0 for a graph
 * without long rows) -- the value the launch entry points take --, 0 when the graph is not eligible (too
 * many rows / a row longer than 64 segments / more slots than one workgroup holds / nodes + segments beyond
 * the 16320 words of an LDS buffer) -- callers then use gda_spmm_csr_kstep_f32 --, or a negative status.
 * The plan is copied to the device by the caller.
 * gda_kstep_plan_host_ex takes flags: bit 0 = bank-aware placement -- WHERE a node's word lives in LDS is
 * chosen such that the 32 words a lane group gathers (or stores) in one instruction fall on different LDS
 * banks as far as possible (the step loop is bound by exactly those gathers); the sums and their order do
 * not change.  gda_kstep_plan_host = flags 1.
 *
 * gda_kstep_lds_f32: K >= 1 steps; x is row-major [n_rows, ldx] (x_colmajor = 0) or column-major
 * [d, ldx] (x_colmajor = 1: ldx >= round_up(n_rows, 4), 16-byte aligned columns), y likewise; bias ([d] or
 * NULL) is added once at the end; colsum ([d] or NULL) receives the column sums of the INPUT over the
 * real rows in a fixed order (the bias gradient when the call is the backward pass of the layer);
 * scratchT holds 2 * d * round_up(n_rows, 4) floats for the transposes of row-major operands.
 * gda_kstep_lds_colmajor_f32: the kernel alone on column-major operands, K >= 0.
 * gda_transpose_f32: out [cols, ldo] = in [rows, ldi]^T.
 * ---------------------------------------------------------------------------- */
int gda_kstep_max_rows(void);
size_t gda_kstep_plan_bytes(int slots);
int gda_kstep_plan_host(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                        int64_t n_rows, void* plan_host, size_t plan_bytes);
int gda_kstep_plan_host_ex(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                           int64_t n_rows, int flags, void* plan_host, size_t plan_bytes);
int gda_kstep_lds_f32(const void* plan, int slots, int64_t n_rows, int64_t d, int K,
                      const float* x, int64_t ldx, int x_colmajor, float* y, int64_t ldy, int y_colmajor,
                      const float* bias, float* colsum, float* scratchT, gda_stream_t stream);
int gda_kstep_lds_colmajor_f32(const void* plan, int slots, int64_t n_rows, int64_t d, int K,
                               const float* xT, int64_t ldx, float* yT, int64_t ldy,
                               const float* bias, float* colsum, gda_stream_t stream);
int gda_transpose_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols,
                      gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * GAT aggregation (single head): edge-softmax attention + weighted neighbour sum, fused.
 *
 * Replaces PyG GATConv(heads=1, concat=False) as GNNBase(gnn='gat') uses it
 * (pygda/nn/gnn_base.py:80-87): for destination i over its incoming edges k (self loop
 * included by the ingestion),  e_k = LeakyReLU_slope(a_src[col_k] + a_dst[i]),
 * alpha = softmax_k(e),  out[i,:] = sum_k alpha_k h[col_k,:].   h = x W^T and the two
 * attention logits a_src = h.att_src, a_dst = h.att_dst are formed by the caller (dense).
 *   forward : out [N,d]; alpha [nnz] (by-destination order, kept for backward)
 *   backward: given gout [N,d] ->  gh [N,d] (the aggregation path only), ga_src [N], ga_dst [N];
 *             the caller adds the dense chain  gh += ga_src att_src + ga_dst att_dst.
 *             needs the by-source CSR and t_to_fwd from gda_build_csr_norm_map;
 *             dpre [nnz] is caller-owned scratch.
 * ---------------------------------------------------------------------------- */
int gda_gat_fwd_f32(const int32_t* rowptr, const int32_t* colidx, int64_t n_rows, int64_t d,
                    const float* h, const float* a_src, const float* a_dst, float slope,
                    float* out, float* alpha, gda_stream_t stream);
int gda_gat_bwd_f32(const int32_t* rowptr, const int32_t* colidx,
                    const int32_t* t_rowptr, const int32_t* t_colidx, const int32_t* t_to_fwd,
                    int64_t n_rows, int64_t d, const float* h, const float* a_src, const float* a_dst,
                    float slope, const float* alpha, const float* gout,
                    float* gh, float* ga_src, float* ga_dst, float* dpre, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Multi-kernel Gaussian MMD over sampled rows, forward and backward, with no
 * [n,n,d] temporary.
 *
 * Replaces MMD / get_MMD / guassian_kernel (pygda/utils/mmd.py:4-159) as called from
 * A2GNN.forward_model (pygda/models/a2gnn.py:208) and GRADE (pygda/models/grade.py:182).
 * For each of `times` resamples t: rows r<n of `total` are src[src_idx[t,r]], rows
 * n<=r<2n are tgt[tgt_idx[t,r-n]] (idx NULL = the rows as given, stacked [times, n, d] per
 * domain; times = 1 is get_MMD).  L2[i,j] = sum_k (total[j,k]-total[i,k])^2 (direct
 * difference form, mmd.py:43-46); bandwidth = (sum L2 + 1e-6)/(m^2-m) / kernel_mul^(kernel_num/2),
 * m = 2n (mmd.py:50-51; fix_sigma>0 overrides the data-dependent value); K = sum_q
 * exp(-L2/(bandwidth*kernel_mul^q)) (mmd.py:52-55); loss_t = mean(XX+YY-XY-YX) over the
 * n x n blocks (mmd.py:100-106); loss = (sum_t loss_t)/times (mmd.py:152-157).
 * The reference's block arithmetic needs equally many source and target rows; so does this.
 *
 *   loss        [1] fp32 (device)
 *   bandwidth   [times] fp32 (device, saved for backward; gradient does not flow
 *               through it -- mmd.py:50 uses .data)
 *   l2_saved    [times, 2n, 2n] fp32 (device; produced by fwd, consumed by bwd -- on return it
 *               holds d K / d L2 with the block signs applied, not the distances themselves;
 *               opaque to the caller)
 * Backward: grad_rows [times, 2n, d] = d loss / d total rows, scaled by *grad_loss.
 * The caller scatters them onto feature rows (a CSR SpMM with the selection matrix).
 * ---------------------------------------------------------------------------- */
size_t gda_mmd_workspace_bytes(int times, int64_t n, int64_t d);
int gda_mmd_fwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                    int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                    int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                    float* loss, float* bandwidth, float* l2_saved,
                    void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_mmd_bwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                    int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                    int times, int64_t n, float kernel_mul, int kernel_num,
                    const float* bandwidth, const float* l2_saved, const float* grad_loss,
                    float* grad_rows, void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* The same two calls with the glue of the trainer's loss line folded in (`loss = CE + MMD(...) * weight`,
 * pygda/models/a2gnn.py:207-209): forward returns add[0] + scale * mmd (add may be NULL); backward scales the
 * incoming gradient by `scale`, and -- given the 0/1 selection CSRs of the sampled rows (rows = feature rows of a
 * domain, columns = positions t*2n + i of the [times, 2n, d] row-gradient array: gda_selection_csr_host) -- sums the
 * row gradients straight onto the feature rows: gsrc [n_src_rows, d], gtgt [n_tgt_rows, d] (grad_rows unused then;
 * the same values as grad_rows followed by the selection-matrix SpMM, bit for bit). */
int gda_mmd_fwd_ex_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                       int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                       int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                       float scale, const float* add, float* loss, float* bandwidth, float* l2_saved,
                       void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* _gather: with row indices, rows_src / rows_tgt ([times*n, d] each, both or neither) receive the sampled rows
 * (`source_feat[source_samples]` mmd.py:153-154) as a by-product of the statistics pass, and the later kernels --
 * and the backward call, which is then handed these buffers without indices -- read them instead of chasing the
 * index again: no separate gather launches. */
int gda_mmd_fwd_gather_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                           int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                           int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                           float scale, const float* add, float* rows_src, float* rows_tgt,
                           float* loss, float* bandwidth, float* l2_saved,
                           void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_mmd_bwd_ex_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                       int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                       int times, int64_t n, float kernel_mul, int kernel_num,
                       const float* bandwidth, const float* l2_saved, const float* grad_loss, float scale,
                       float* grad_rows,
                       const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                       const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                       void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* The same loss and gradients in ONE pass over the row pairs (round 4): the kernel weights d K / d L2 are consumed where
 * they are produced -- multiplied into the rows on the 16-bit matrix cores -- instead of travelling through a
 * [times, 2n, 2n] matrix.  Both products run on fp16 MFMAs with SPLIT operands (a = hi + lo, three products per pair,
 * fp32 accumulation: 22 significant bits per operand, an error of the order of an fp32 sum in another order; the rows
 * are scaled by one power of two per resample first, so fp16's exponent range never matters).  Covered:
 * kernel_num = 5 with kernel_mul = 2 (every pygda call: mmd.py:57-63 defaults), d in {32, 64, 96, 128}, rows on
 * 16-byte boundaries; gda_mmd_fused_nseg() returns 0 for anything else (use gda_mmd_fwd_gather_f32 / gda_mmd_bwd_ex_f32).
 *   grad_part   [times, nseg, 2n, d] fp32 (device; nseg = gda_mmd_fused_nseg(...)): UNSCALED row-gradient partials,
 *               produced by fwd, consumed by bwd (which multiplies by 4 * grad_loss * scale / (n^2 times), folds the
 *               segments in order and, given the selection CSRs, scatters onto the feature rows).  Opaque.
 *   workspace   gda_mmd_workspace_bytes(times, n, d) bytes, as for the two-pass calls
 * src_idx / tgt_idx need rows_src / rows_tgt (the gathered copy) here. */
int gda_mmd_fused_nseg(int times, int64_t n, int64_t d, float kernel_mul, int kernel_num);
/* Host arithmetic only (no device): what the fused pass lays out for (times, n, d), for tests and debuggers.
 *   out[16] = { nb, ntiles, njb, nseg, workgroups, image bytes,
 *               image offsets: rows lo, columns hi, columns lo, norms; float index of the tile scale in the norm block,
 *               floats reserved for that block (32 norms + scale);
 *               workspace offsets: images, part_max, kpartial; workspace bytes }
 * GDA_E_UNSUPPORTED when gda_mmd_fused_nseg() would return 0. */
int gda_mmd_fused_layout(int times, int64_t n, int64_t d, int64_t* out, int n_out);
int gda_mmd_fused_fwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                          int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                          int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                          float scale, const float* add, float* rows_src, float* rows_tgt,
                          float* loss, float* bandwidth, float* grad_part, int nseg,
                          void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_mmd_fused_bwd_f32(const float* grad_part, int nseg, int times, int64_t n, int64_t d,
                          const float* grad_loss, float scale, float* grad_rows,
                          const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                          const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                          gda_stream_t stream);
/* _mask: mask_src / mask_tgt (either may be NULL) = the OUTPUT y [n_*_rows, d] of the activation dropout(relu(.), p_*) that
 * produced that domain's feature rows: the scattered row gradient is stored as y > 0 ? g / (1 - p) : 0, i.e. already
 * through that activation's backward (gda_relu_dropout_bwd_f32's values) -- its launch and one [rows, d] round trip
 * disappear.  Scatter form only (selection CSRs given). */
int gda_mmd_fused_bwd_mask_f32(const float* grad_part, int nseg, int times, int64_t n, int64_t d,
                               const float* grad_loss, float scale, float* grad_rows,
                               const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                               const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                               const float* mask_src, float p_src, const float* mask_tgt, float p_tgt,
                               gda_stream_t stream);

/* The one-pass MMD for ANY feature width d <= 1024 (round 6; pygda/models/grade.py:177-182 calls MMD() on 645-wide rows):
 * the feature dimension is cut into chunks of 32 | 64 | 96 | 128 columns -- the chunk width that pads d least --, a
 * workgroup keeps the 32 x 32 distance blocks of up to eight row tiles in accumulators over all chunks, takes the
 * exponentials once, and forms the gradient chunk by chunk (csrc/gda_mmd_chunked.inc).  Rows of any alignment; same
 * split-fp16 arithmetic and the same error class as gda_mmd_fused_fwd_f32.  kernel_num = 5, kernel_mul = 2, n <= 1024.
 *   gda_mmd_chunked_plan   host arithmetic only: out[8] = { nseg, padded width dp, nb (32-column blocks per chunk), nc
 *                          (chunks), ntiles, njb, workgroups, image bytes }; GDA_E_UNSUPPORTED outside the envelope
 *   rows_src / rows_tgt    [times * n, ld_rows] fp32 scratch, ld_rows = dp, 16-byte aligned: the gathered (src_idx given)
 *                          or copied (NULL: rows stacked [times, n, d]) rows, zero padded -- REQUIRED
 *   grad_part              [times, nseg, 2n, ld_part] fp32, ld_part = dp: unscaled row-gradient partials for
 *                          gda_mmd_fused_bwd_ld_f32 (= gda_mmd_fused_bwd_mask_f32 with the partials' row stride)
 *   workspace              gda_mmd_chunked_workspace_bytes(times, n, d) */
int gda_mmd_chunked_plan(int times, int64_t n, int64_t d, float kernel_mul, int kernel_num, int64_t* out, int n_out);
size_t gda_mmd_chunked_workspace_bytes(int times, int64_t n, int64_t d);
int gda_mmd_chunked_fwd_f32(const float* src, int64_t ld_src, const float* tgt, int64_t ld_tgt,
                            int64_t d, const int64_t* src_idx, const int64_t* tgt_idx,
                            int times, int64_t n, float kernel_mul, int kernel_num, float fix_sigma,
                            float scale, const float* add, float* rows_src, float* rows_tgt, int64_t ld_rows,
                            float* loss, float* bandwidth, float* grad_part, int64_t ld_part, int nseg,
                            void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_mmd_fused_bwd_ld_f32(const float* grad_part, int64_t ld_part, int nseg, int times, int64_t n, int64_t d,
                             const float* grad_loss, float scale, float* grad_rows,
                             const int32_t* sel_s_rowptr, const int32_t* sel_s_col, int64_t n_src_rows, float* gsrc,
                             const int32_t* sel_t_rowptr, const int32_t* sel_t_col, int64_t n_tgt_rows, float* gtgt,
                             const float* mask_src, float p_src, const float* mask_tgt, float p_tgt,
                             gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Gradient-reversal + linear domain discriminator + softmax cross-entropy, fused.
 *
 * Replaces GradReverse.apply (pygda/nn/reverse_layer.py:39,65-66) -> Linear(h, C) ->
 * F.cross_entropy over cat(source, target) rows (pygda/models/a2gnn.py:197-205,
 * pygda/models/grade.py:170-176).  Rows [0, n_src) of `feat_src` carry label 0,
 * rows of `feat_tgt` label 1 (C = 2 in the reference; C <= 8 supported, labels may
 * instead be given per row with `labels` != NULL over the concatenation).
 *   forward : loss[0] = mean_r ( logsumexp(z_r) - z_r[label_r] ),  z = f W^T + b
 *             also writes probs [n_src+n_tgt, C] (softmax, saved for backward)
 *   backward: dz = (probs - onehot)/n * grad_loss ; gW [C,h] = dz^T f ; gb [C] = sum dz ;
 *             gfeat = -alpha * dz W   (the reversal), written for both domains.
 * ---------------------------------------------------------------------------- */
int gda_grl_disc_ce_fwd_f32(const float* feat_src, int64_t ld_src, int64_t n_src,
                            const float* feat_tgt, int64_t ld_tgt, int64_t n_tgt,
                            int64_t h, int C, const float* W, const float* b,
                            const int64_t* labels, float* probs, float* loss,
                            void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_grl_disc_ce_bwd_f32(const float* feat_src, int64_t ld_src, int64_t n_src,
                            const float* feat_tgt, int64_t ld_tgt, int64_t n_tgt,
                            int64_t h, int C, const float* W, const int64_t* labels,
                            const float* probs, const float* grad_loss, float alpha,
                            float* gfeat_src, float* gfeat_tgt, float* gW, float* gb,
                            void* workspace, size_t workspace_bytes, gda_stream_t stream);
size_t gda_grl_disc_workspace_bytes(int64_t n_rows, int64_t h, int C);

/* ------------------------------------------------------------------------------
 * Gradient-reversal + TWO-LAYER domain discriminator + per-domain softmax cross-entropy, fused
 * (csrc/gda_disc_mlp.hip).  Replaces UDAGCN's domain branch, pygda/models/udagcn.py:176-190 with the
 * discriminator built at pygda/nn/udagcn_base.py:157-162:
 *     D(x) = W2 dropout_p(relu(W1 GradReverse(x) + b1)) + b2,   W1 [a, h], b1 [a], W2 [2, a], b2 [2] (row-major)
 *     losses[0] = mean over the n_s source rows of CE(D(x), 0),  losses[1] = mean over the n_t target rows of CE(D(x), 1),
 *     losses[2] = their sum (what the reference adds to its loss; the two means also come back separately so that a
 *     data-parallel caller can weight them by node counts).  h <= 128, a <= 64.  Dropout keep-bits: Philox keyed on (seed, *step, site + domain, row, unit) --
 * forward and backward regenerate them, nothing of size [rows, a] is stored.
 *   backward: grad_losses = upstream gradients (device): of the two means at [0] and [grad_stride] (grad_stride 1), or
 *             of their sum (grad_stride 0); g_es / g_et (may be NULL) receive
 *             -alpha * dD/dx (the reversal; alpha_dev != NULL: alpha read from the device), gW1 [a, h], gb1 [a],
 *             gW2 [2, a], gb2 [2] the parameter gradients (fixed-order sums: deterministic).
 * ---------------------------------------------------------------------------- */
size_t gda_grl_mlp_ce_workspace_bytes(int64_t h, int64_t a);
int gda_grl_mlp_ce_fwd_f32(const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                           int64_t h, int64_t a, const float* W1, const float* b1, const float* W2, const float* b2,
                           float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                           float* losses, void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_grl_mlp_ce_bwd_f32(const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                           int64_t h, int64_t a, const float* W1, const float* b1, const float* W2, const float* b2,
                           float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                           const float* grad_losses, int grad_stride, float alpha, const float* alpha_dev,
                           float* g_es, float* g_et, float* gW1, float* gb1, float* gW2, float* gb2,
                           void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* The same row kernels with the head given: head = 0 is gda_grl_mlp_ce_*; head = 1 -- ONE logit per row through a sigmoid, the
 * loss term of a domain the MEAN of its rows' sigmoids: losses[0] = mean_s D(x), losses[1] = mean_t D(x), D = Linear(h, a) -
 * ReLU - Dropout(p) - Linear(a, 1) - Sigmoid, i.e. the critic of pygda/models/adagcn.py:264-270 inside the encoder's loss
 * (:190-193, `torch.mean(D(source)) - torch.mean(D(target))`), W2 [1, a], b2 [1], gW2 [1, a], gb2 [1]; no reversal: pass
 * alpha = -1 (the input gradient is multiplied by -alpha). */
int gda_mlp_head_fwd_f32(int head, const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                         int64_t h, int64_t a, const float* W1, const float* b1, const float* W2, const float* b2,
                         float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                         float* losses, void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_mlp_head_bwd_f32(int head, const float* es, int64_t ld_s, int64_t n_s, const float* et, int64_t ld_t, int64_t n_t,
                         int64_t h, int64_t a, const float* W1, const float* b1, const float* W2, const float* b2,
                         float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                         const float* grad_losses, int grad_stride, float alpha, const float* alpha_dev,
                         float* g_es, float* g_et, float* gW1, float* gb1, float* gW2, float* gb2,
                         void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * DANE's LSGAN discriminator head (csrc/gda_disc_mlp.hip).  Replaces what follows the first layer of
 * domain_discriminator = Linear(h, h) - ReLU - Linear(h, 1) (pygda/models/dane.py:241-247) in the LSGAN terms
 * `(pre ** 2).mean()` / `((pre - 1) ** 2).mean()` (dane.py:339-350, 468-470) and their autograd graph, on
 * Z = x W1^T + b1 ([rows, a] fp32, the first layer: a GEMM, gda_gemm_ex_f32):
 *   forward : pre[r] = sum_k relu(Z[r,k]) w2[k] + b2,  loss[0] = mean_r (pre[r] - target)^2
 *   backward: gZ[r,k] = 2 (pre[r] - target) / rows * grad_loss[0] * w2[k] * [Z[r,k] > 0],  gw2 [a], gb2 [1]
 * (gW1, gb1 and the input gradient follow from gZ by gda_gemm_ex_f32).  a <= 256; fixed-order sums.
 * ---------------------------------------------------------------------------- */
size_t gda_lsgan_head_workspace_bytes(int64_t a);
int gda_lsgan_head_fwd_f32(const float* Z, int64_t ldz, int64_t rows, int64_t a, const float* w2, const float* b2,
                           float target, float* pre, float* loss, void* workspace, size_t workspace_bytes,
                           gda_stream_t stream);
int gda_lsgan_head_bwd_f32(const float* Z, int64_t ldz, int64_t rows, int64_t a, const float* w2, const float* pre,
                           float target, const float* grad_loss, float* gZ, int64_t ldg, float* gw2, float* gb2,
                           void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Wasserstein critic update with gradient penalty (WGAN-GP), loss and parameter gradients in closed form
 * (csrc/gda_critic.hip).  Replaces the body of AdaGCN's critic loop, pygda/models/adagcn.py:169-183 with
 * gradient_penalty (:387-454), for the critic built at :264-270:
 *     D(x) = sigmoid(w2 . dropout_p(relu(W1 x + b1)) + b2),   W1 [a, h] row-major, b1 [a], w2 [a], b2 [1]
 *     loss = -| mean D(es) - mean D(et) | + gp_weight * mean_i (|| grad_x D(x_i) ||_2 - 1)^2,
 *     x_i over cat(es, et, interpolates),  interpolates_i = et[idx_t[i]] + alpha[i] * (es[idx_s[i]] - et[idx_t[i]])
 * es [n_s, h], et [n_t, h] contiguous; idx_s / idx_t int32 [n_i], alpha [n_i] (the reference draws it from the
 * host generator).  The three critic evaluations (on es, on et, on the penalty rows) draw independent dropout
 * masks from Philox keyed on (seed, step[0], site + {0, 1, 2}); dropout_p = 0 needs no step.  Outputs: loss [1]
 * and the gradients gW1 [a, h], gb1 [a], gw2 [a], gb2 [1] of that loss (deterministic reductions).
 * h <= 256, h % 4 == 0, a <= 64.
 * ---------------------------------------------------------------------------- */
size_t gda_wgan_critic_workspace_bytes(int64_t n_s, int64_t n_t, int64_t n_i, int h, int a);
int gda_wgan_critic_f32(const float* es, int64_t n_s, const float* et, int64_t n_t, int h,
                        const int32_t* idx_s, const int32_t* idx_t, const float* alpha, int64_t n_i,
                        const float* W1, const float* b1, const float* w2, const float* b2, int a,
                        float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                        float gp_weight, float* loss, float* gW1, float* gb1, float* gw2, float* gb2,
                        void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * ReLU + inverted dropout, fused (the activation after every conv layer:
 * pygda/nn/a2gnn_base.py:135-138, grade_base.py:146-148, gnn_base.py:166-168).
 *   forward : y = (x > 0 && keep) ? x/(1-p) : 0, keep-bits from Philox-4x32-10 keyed on `seed`
 *             with counter (step[0], site, element/4): `step` is a device int64 the trainer
 *             increments once per training step (so hipGraph replays draw fresh masks), `site`
 *             numbers the call sites inside a step.
 *   backward: gx = (y > 0) ? gy/(1-p) : 0   -- needs only the saved output.
 * x, y, gy, gx: n contiguous fp32 values, 16-byte aligned.
 * ---------------------------------------------------------------------------- */
int gda_relu_dropout_fwd_f32(const float* x, float* y, int64_t n, float p, uint64_t seed,
                             const int64_t* step, uint32_t site, gda_stream_t stream);
int gda_relu_dropout_bwd_f32(const float* gy, const float* y, float* gx, int64_t n, float p,
                             gda_stream_t stream);
/* gda_relu_dropout_fwd_f32 of `copies` stacked copies of x [period] (period % 4 == 0) without materialising them:
 * y [copies * period], element i = the activation of x[i % period] with element i's keep-bit -- what
 * `F.dropout(F.relu(x.repeat(copies, 1)), p)` draws.  AdaGCN's critic loop re-encodes both domains `critic_steps` times
 * with an encoder that does not change inside the loop (pygda/models/adagcn.py:169-171): the passes differ by their
 * dropout draws only and run as one stacked pass.  Forward only (the loop runs under no_grad). */
int gda_relu_dropout_tiled_fwd_f32(const float* x, int64_t period, int64_t copies, float* y, float p, uint64_t seed,
                                   const int64_t* step, uint32_t site, gda_stream_t stream);
/* The same activation across a layout change, next to the LDS-resident K-step kernel (which works on
 * column-major activations): forward reads xT [d, ldT] column-major and writes y [n, d] row-major, backward
 * reads gy, y [n, d] row-major and writes gxT [d, ldT] column-major (rows n..ldT-1 of a column untouched) --
 * the transposition rides through LDS tiles, no separate pass.  Same keep-bits as the plain kernels (keyed on
 * the row-major element index); d % 4 == 0. */
int gda_relu_dropout_fwd_cm_f32(const float* xT, int64_t ldT, float* y, int64_t n, int64_t d, float p,
                               uint64_t seed, const int64_t* step, uint32_t site, gda_stream_t stream);
int gda_relu_dropout_bwd_cm_f32(const float* gy, const float* y, float* gxT, int64_t ldT, int64_t n, int64_t d,
                               float p, gda_stream_t stream);

/* Two independent dropout draws of one activation, stacked: y [2n, d], y[i] = drop_a(relu(x[i])),
 * y[n + i] = drop_b(relu(x[i])) -- A2GNN's source feature pass and source logits pass (pygda/models/a2gnn.py:181,192:
 * the same layers on the same input, separate dropout draws) continue as ONE pass over 2n rows.  Backward sums the
 * halves: gx[i] = ((y[i] > 0) gy[i] + (y[n+i] > 0) gy[n+i]) / (1 - p).  d % 4 == 0.
 * gda_stack2_f32: out = [a ; b] of half_elems floats each (a NULL half reads as zeros) -- the gradient of such a
 * pair whose halves went to different consumers (MMD rows / classifier). */
int gda_relu_dropout_pair_fwd_f32(const float* x, float* y, int64_t n, int64_t d, float p, uint64_t seed,
                                  const int64_t* step, uint32_t site_a, uint32_t site_b, gda_stream_t stream);
/* colsum (may be NULL) [d]: column sums of gx as a by-product (the bias gradient of the layer that produced x);
 * needs workspace of gda_relu_dropout_pair_workspace_bytes(d) and d <= 1024. */
size_t gda_relu_dropout_pair_workspace_bytes(int64_t d);
int gda_relu_dropout_pair_bwd_f32(const float* gy, const float* y, float* gx, int64_t n, int64_t d, float p,
                                  float* colsum, void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_stack2_f32(const float* a, const float* b, float* out, int64_t half_elems, gda_stream_t stream);
/* The backward of `relu_dropout` followed by a split into halves, in one pass: gx [2 * half_elems] = mask(y) * [ga ; gb]
 * / (1 - p) with mask = (y > 0) -- gda_stack2_f32 and gda_relu_dropout_bwd_f32 without the unmasked stack in between
 * (a NULL half reads as zeros). */
int gda_relu_dropout_bwd2_f32(const float* ga, const float* gb, const float* y, float* gx, int64_t half_elems,
                              float p, gda_stream_t stream);

/* Column sums of a row-major [n, d] matrix (ld = ldx): the bias gradient `gy.sum(0)` of a conv layer
 * (out += self.bias, pygda/nn/prop_gcn_conv.py:212-213) as a deterministic two-stage sum; d <= 1024. */
size_t gda_colsum_workspace_bytes(int64_t n, int64_t d);
int gda_colsum_f32(const float* x, int64_t ldx, int64_t n, int64_t d, float* out, void* workspace,
                   size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Layer epilogue of StruRW's mixup backbone -- replaces the elementwise tail of the three
 * MixUpGCNConv calls per layer in pygda/nn/mixup_base.py:146-196 (out = Agg(lin(x)) + lin_cen(x_cen)
 * + bias, pygda/nn/mixup_gcnconv.py:231-236).  With P = Agg(lin(x)) [n, h] (one aggregation: the
 * shuffled graph of pygda/models/strurw.py:735-758 is a renumbering, so its aggregate is P[perm]):
 *     XX[i]     = drop(relu(P[i] + C[i]  + bias))                                       (x')
 *     XX[n + i] = drop(lam relu(P[i] + Cm[i] + bias) + (1-lam) relu(Pb[i] + Cm[i] + bias))   (x_mix')
 * Pb = P[perm[i]] when the pointer is NULL, else an explicit [n, h] aggregate (foreign edge_index_b).
 *   first != 0: CC = C = lin_cen(x) is [n, h] and Cm[i] = lam C[i] + (1-lam) C[perm[i]]  (x_mix of the
 *               input features is never formed);   first == 0: CC = [C ; Cm] is [2n, h] = lin_cen(XX_prev).
 * mask [n, h] bytes (relu bits of the two mixed terms + keep bit) is written for the backward pass.
 * Keep-bits as in gda_relu_dropout_fwd_f32, two call sites (plain row, mixed row).  h % 4 == 0, h <= 1024.
 * Backward: gXX [2n, h] -> gP [n, h] (incl. the P[perm] route, gathered through inv_perm), gPb [n, h] iff the
 * forward had an explicit Pb (else NULL), gCC ([n, h] if first else [2n, h]), gbias [h] (deterministic
 * two-stage column sum; workspace from gda_mixup_combine_workspace_bytes).
 * ---------------------------------------------------------------------------- */
size_t gda_mixup_combine_workspace_bytes(int64_t n, int64_t h);
int gda_mixup_combine_fwd_f32(const float* P, const float* Pb, const float* CC, int first, const float* bias,
                              const int64_t* perm, int64_t n, int64_t h, float lam, float p, uint64_t seed,
                              const int64_t* step, uint32_t site_x, uint32_t site_m, float* XX, uint8_t* mask,
                              gda_stream_t stream);
int gda_mixup_combine_bwd_f32(const float* gXX, const float* XX, const uint8_t* mask, const int64_t* inv_perm,
                              int first, int64_t n, int64_t h, float lam, float p, float* gP, float* gPb,
                              float* gCC, float* gbias, void* workspace, size_t workspace_bytes,
                              gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Feature-row gather  out[r,:] = x[idx[r],:]  (mini-batch assembly: the x[n_id]
 * slice PyG's NeighborLoader performs, pygda/models/a2gnn.py:260-277).
 * ---------------------------------------------------------------------------- */
int gda_gather_rows_f32(const float* x, int64_t ldx, int64_t d, const int64_t* idx,
                        int64_t n_out, float* out, int64_t ldo, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Graph-level readout (mode='graph'): PyG's global_mean_pool(x, batch) as pygda/nn/a2gnn_base.py:140-141 calls
 * it on the collated batch of a DataLoader -- `batch` sorted, graph g = rows seg_ptr[g] .. seg_ptr[g+1].
 * out[g] = (rows added in node order) / max(count, 1); backward gx[i] = gout[batch[i]] / max(count, 1).
 * seg_ptr int64 [G+1], batch int64 [n] (device).
 * ---------------------------------------------------------------------------- */
int gda_segment_mean_fwd_f32(const float* x, int64_t ldx, const int64_t* seg_ptr, int64_t G, int64_t d,
                             float* out, int64_t ldo, gda_stream_t stream);
int gda_segment_mean_bwd_f32(const float* gout, int64_t ldg, const int64_t* seg_ptr, const int64_t* batch,
                             int64_t n, int64_t d, float* gx, int64_t ldx, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Host neighbour sampler (mini-batch assembly; HOST pointers throughout).
 *
 * Replaces the C++ sampler behind PyG's NeighborLoader as pygda's trainers construct it
 * (pygda/models/a2gnn.py:260-277; grade.py, udagcn.py, adagcn.py likewise): L-hop in-neighbour
 * sampling without replacement from seed nodes, fan-out list `fanouts` (-1 = all), seeds first
 * then newly reached nodes in discovery order, sampled edges relabelled to local ids.
 * A handle is not thread-safe; one per loader.  gda_sampler_sample keeps the batch inside the
 * handle and reports its sizes; gda_sampler_fetch copies it out into caller buffers of those
 * sizes (nodes: global ids [n_nodes]; esrc/edst: local ids [n_edges], message esrc -> edst).
 * ---------------------------------------------------------------------------- */
typedef struct gda_sampler gda_sampler;
int gda_sampler_create(const int64_t* src_host, const int64_t* dst_host, int64_t E, int64_t N,
                       gda_sampler** out);
void gda_sampler_destroy(gda_sampler* s);
/* Worker threads for the neighbour picks of a hop (default 1).  The batch does not depend on it. */
int gda_sampler_set_threads(gda_sampler* s, int workers);
int gda_sampler_sample(gda_sampler* s, const int64_t* seeds_host, int64_t n_seeds,
                       const int32_t* fanouts, int L, uint64_t rng_seed,
                       int64_t* n_nodes_out, int64_t* n_edges_out);
int gda_sampler_fetch(const gda_sampler* s, int64_t* nodes_out, int64_t* esrc_out,
                      int64_t* edst_out);
/* The GCN-normalised adjacency of the last sampled batch as the two CSRs of the aggregation kernels, built on
 * the host where the structure is known: exactly the arrays gda_build_csr_norm(esrc, edst, NULL, n_edges,
 * n_nodes, 1.0, add_self_loops = 1, normalize = 1, degree_side = 0) writes (gcn_norm with unit weights,
 * pygda/nn/prop_gcn_conv.py:64-81), without the device sorts.  rowptr / t_rowptr: n_nodes + 1 entries;
 * colidx / val / t_colidx / t_val: n_edges + n_nodes entries (HOST pointers). */
int gda_sampler_csr_norm(const gda_sampler* s, int32_t* rowptr, int32_t* colidx, float* val,
                         int32_t* t_rowptr, int32_t* t_colidx, float* t_val);

/* ------------------------------------------------------------------------------
 * Device neighbour sampler (csrc/gda_dsampler.hip; DEVICE pointers, fan-outs on the host).
 *
 * The same NeighborLoader call sites (pygda/models/a2gnn.py:260-277) with the graph resident in HBM: the same
 * contract as the host sampler above and, bit for bit, the same batches (same counter-based generator keyed on
 * (seed, hop, node), same discovery order) -- plus the batch's GCN-normalised CSR pair exactly as
 * gda_build_csr_norm(esrc, edst, NULL, n_edges, n_nodes, 1.0, 1, 1, 0) would write it, without its two sorts.
 * Stateless: the caller owns every array.
 *
 * gda_dsampler_build_graph: in-neighbour lists (in_ptr int64 [N+1], in_src int32 [E], edge order kept inside a
 *   list) from a COO edge list; status = device int32[2] {edges with an endpoint outside [0, N), largest in-degree}.
 * gda_dsampler_caps: node_cap / edge_cap of a batch of n_seeds seeds (every output array is sized by them);
 *   GDA_E_UNSUPPORTED for a fan-out of 0 or above 64, or a batch beyond the int32 range (use the host sampler).
 * gda_dsampler_sample: nodes int64 [node_cap] (global ids, seeds first), esrc / edst int64 [edge_cap] (local ids),
 *   rowptr / t_rowptr int32 [node_cap + 1], colidx / val / t_colidx / t_val [edge_cap + node_cap] (all six NULL:
 *   no CSR), counts = device int64[5] {n_nodes, n_edges, nnz, status: 0 ok / 1 capacity exceeded / 2 seed out of
 *   range, n_interior: nodes before the last hop's discoveries -- the rows from there on are never expanded and hold
 *   their self loop only (gda_spmm_csr_interior_kstep_f32)}; rowptr[i] == nnz for i >= n_nodes.  All launches are capacity sized; nothing returns to the host.
 * ---------------------------------------------------------------------------- */
size_t gda_dsampler_graph_workspace_bytes(int64_t E, int64_t N);
int gda_dsampler_build_graph(const int64_t* src, const int64_t* dst, int64_t E, int64_t N,
                             int64_t* in_ptr, int32_t* in_src, int32_t* status,
                             void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_dsampler_caps(int64_t n_seeds, const int32_t* fanouts_host, int L, int64_t max_in_degree, int64_t E,
                      int64_t N, int64_t* node_cap, int64_t* edge_cap);
size_t gda_dsampler_workspace_bytes(int64_t n_seeds, const int32_t* fanouts_host, int L, int64_t max_in_degree,
                                    int64_t E, int64_t N);
int gda_dsampler_sample(const int64_t* in_ptr, const int32_t* in_src, int64_t N, int64_t E, int64_t max_in_degree,
                        const int64_t* seeds, int64_t n_seeds, const int32_t* fanouts_host, int L, uint64_t rng_seed,
                        int64_t* nodes, int64_t* esrc, int64_t* edst,
                        int32_t* rowptr, int32_t* colidx, float* val,
                        int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                        int64_t* counts, void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* Host-to-device copy of a PINNED host block (hipHostMalloc / hipHostRegister, 16-byte aligned both sides) as a kernel on
 * `stream` that reads the host memory over the bus: for the 0.1 - 1 MB blocks a captured step is fed with per replay (MMD row
 * samples, interpolation weights), where a DMA-engine copy queued behind a running graph idles the device for the engine
 * hand-over.  GDA_E_UNSUPPORTED when the block is not device-mapped or misaligned (callers fall back to hipMemcpyAsync). */
int gda_copy_from_pinned(void* dst, const void* src_host, size_t bytes, gda_stream_t stream);

/* One foreign call per batch for a loader that RECYCLES its batch blocks (round 5; pygda_amd/sampler.py, the producer
 * thread of pygda_amd/data.py's NeighborLoader -- the reference's NeighborLoader workers, pygda/models/a2gnn.py:260-277):
 *   [stream waits for wait_event] -> seeds (host OR device, int64 [n_seeds]) copied to seeds_dev -> gda_dsampler_sample
 *   -> gda_interior_plan_build for both directions when plan_fwd / plan_bwd are given (both or neither; their {q, T} land
 *   in counts[5:7] / counts[7:9]) -> counts (device int64[12], zeroed first) copied to counts_host (PINNED int64[12]) ->
 *   [done_event recorded].  Every array may be a block an earlier batch used: wait_event (recorded by the consumer on its
 *   stream once it has moved on) orders the overwrite.
 * gda_event_*: plain HIP events without timing, owned by the library, so that neither side of that hand-over builds
 *   framework objects per batch.  gda_event_synchronize blocks the calling host thread. */
int gda_event_create(void** event_out);
int gda_event_destroy(void* event);
int gda_event_record(void* event, gda_stream_t stream);
int gda_event_synchronize(void* event);
int gda_stream_wait_event(gda_stream_t stream, void* event);
int gda_dsampler_batch(const int64_t* in_ptr, const int32_t* in_src, int64_t N, int64_t E, int64_t max_in_degree,
                       const int64_t* seeds, int64_t n_seeds, int64_t* seeds_dev,
                       const int32_t* fanouts_host, int L, uint64_t rng_seed,
                       int64_t* nodes, int64_t* esrc, int64_t* edst,
                       int32_t* rowptr, int32_t* colidx, float* val,
                       int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                       int64_t* counts, void* plan_fwd, void* plan_bwd, size_t plan_bytes,
                       int64_t* counts_host, void* wait_event, void* done_event,
                       void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* _ex: interior_rows > 0 declares the batch's first min(interior_rows, n_nodes) rows "interior" (counts[4] is raised to it
 * before the plans are built): for a consumer that runs every batch at ONE static shape (the captured sampled step).  The
 * rows between the sampled n_interior and interior_rows hold their unit self loop only, so the interior-rows kernels
 * leave them as the leaf copy would. */
int gda_dsampler_batch_ex(const int64_t* in_ptr, const int32_t* in_src, int64_t N, int64_t E, int64_t max_in_degree,
                          const int64_t* seeds, int64_t n_seeds, int64_t* seeds_dev,
                          const int32_t* fanouts_host, int L, uint64_t rng_seed,
                          int64_t* nodes, int64_t* esrc, int64_t* edst,
                          int32_t* rowptr, int32_t* colidx, float* val,
                          int32_t* t_rowptr, int32_t* t_colidx, float* t_val,
                          int64_t* counts, void* plan_fwd, void* plan_bwd, size_t plan_bytes,
                          int64_t* counts_host, void* wait_event, void* done_event, int64_t interior_rows,
                          void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Host construction of the PPMI graph (HOST pointers).
 *
 * Replaces the Python random-walk loop of PPMIConv.norm (pygda/nn/ppmi_conv.py:98-172):
 * `passes` walks (40 in the reference) of length ~U{1..path_len} from every node over the
 * symmetrised neighbour sets, row-normalised visit counts, PPMI weights
 * max(log(p / colsum * |targets| / path_len), 0).  Own counter-based generator (`seed`):
 * statistical, not bit-wise, parity with the reference's np.random stream.  The weighted edge
 * list (sorted by (src, dst), zero weights kept, no self loops added) is held by the returned
 * handle; fetch it into buffers of gda_edge_list_size() entries, then feed it to
 * gda_build_csr_norm(add_self_loops=1, normalize=1, degree_side=1) (ppmi_conv.py:174-184).
 * ---------------------------------------------------------------------------- */
/* Host helper for the MMD backward: CSR (rowptr [num_rows+1], colidx [times*n], int32) of the
 * 0/1 selection matrix that sums the per-sample row gradients back onto the sampled feature
 * rows; idx_host [times, n] are the row samples torch.randint drew on the host
 * (pygda/utils/mmd.py:148-149); column ids are t*m + offset + r into the [times, m, d] buffer. */
int gda_selection_csr_host(const int64_t* idx_host, int times, int64_t n, int64_t num_rows,
                           int64_t offset, int64_t m, int32_t* rowptr_out, int32_t* colidx_out);

typedef struct gda_edge_list gda_edge_list;
int gda_ppmi_build_host(const int64_t* src_host, const int64_t* dst_host, int64_t E, int64_t N,
                        int path_len, int passes, uint64_t seed, gda_edge_list** out);
int64_t gda_edge_list_size(const gda_edge_list* l);
int gda_edge_list_fetch(const gda_edge_list* l, int64_t* src_out, int64_t* dst_out, float* w_out);
void gda_edge_list_destroy(gda_edge_list* l);

/* Aggregation with an affine epilogue, for polynomial graph filters:
 *     y[i] = alpha * x[i] + beta * sum_j A[i,j] x[j] + gamma * z[i]
 * gamma = gamma_imm * (gamma_dev ? *gamma_dev : 1); z may be NULL (then gamma is ignored).
 * Replaces the propagate calls of BernProp.forward (pygda/nn/dgsda_base.py:128-148), where
 * the operators are L = I - A_sym (get_laplacian 'sym', :128) and 2I - L = I + A_sym
 * (add_self_loops(-norm, fill 2), :130): one launch per operator application, the Horner
 * accumulation of the filter terms riding in the same pass.  gamma_dev lets a learnable filter
 * coefficient stay on the device (hipGraph-capturable, no host sync). */
int gda_spmm_csr_axpby_f32(const int32_t* rowptr, const int32_t* colidx, const float* val,
                           int64_t n_rows, int64_t d, const float* x, int64_t ldx,
                           float* y, int64_t ldy, float alpha, float beta,
                           const float* z, int64_t ldz, float gamma_imm, const float* gamma_dev,
                           const gda_row_split* split, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * TDSS smoothness term (rank 1 of the "next" rows): Laplacian loss over a smoothing graph.
 *
 * gda_laplacian_fwd_f32 / _bwd_f32 replace TDSS.compute_laplacian_loss
 * (pygda/models/tdss.py:435-454) and its autograd graph:
 *     loss = 1/2 sum_e || f[row_e] dinv[row_e] - f[col_e] dinv[col_e] ||^2,
 *     dinv[i] = (#edges with row == i)^-1/2 (inf -> 0).
 *   rowptr_r/colidx_r : CSR whose rows are the `row` (= edge_index[0]) nodes, entries the `col` nodes
 *                       (the by-source CSR of gda_build_csr_norm(add_self_loops=0, normalize=0))
 *   rowptr_c/colidx_c : CSR whose rows are the `col` nodes, entries the `row` nodes (by-destination)
 *   f [N, d] fp32 (ld = ldf); loss [1]; dinv [N] is written by fwd and read by bwd;
 *   grad_loss [1] DEVICE scalar (upstream gradient); grad_f [N, d] overwritten.
 * ---------------------------------------------------------------------------- */
size_t gda_laplacian_workspace_bytes(int64_t N);
int gda_laplacian_fwd_f32(const int32_t* rowptr_r, const int32_t* colidx_r, int64_t N, int d,
                          const float* f, int64_t ldf, float* loss, float* dinv,
                          void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_laplacian_bwd_f32(const int32_t* rowptr_r, const int32_t* colidx_r,
                          const int32_t* rowptr_c, const int32_t* colidx_c, int64_t N, int d,
                          const float* f, int64_t ldf, const float* dinv, const float* grad_loss,
                          float* grad_f, int64_t ldg, gda_stream_t stream);

/* Host builders of the TDSS smoothing graphs (pygda/models/tdss.py:314-388); results come back as a
 * gda_edge_list sorted by (row, col) without duplicates (w is empty: pass NULL to _fetch).
 *   gda_two_hop_host   : `rounds` applications of TwoHopNeighbor (tdss.py:67-87):
 *                        E <- coalesce(E U pattern(A.A) without self loops)
 *   gda_walk_smooth_host: one uniform random walk of walk_len steps from every node along
 *                        row -> col (torch_cluster.random_walk semantics); edge (visited, start)
 *                        for every visited node (tdss.py:367-373).  Own generator (`seed`):
 *                        statistical parity. */
int gda_two_hop_host(const int64_t* src_host, const int64_t* dst_host, int64_t E, int64_t N,
                     int rounds, int threads, gda_edge_list** out);
int gda_walk_smooth_host(const int64_t* src_host, const int64_t* dst_host, int64_t E, int64_t N,
                         int walk_len, uint64_t seed, int threads, gda_edge_list** out);

/* ------------------------------------------------------------------------------
 * Data-parallel exchange steps over RCCL (xGMI inside a node), on a communicator owned by this
 * library and on the caller's stream.  The reference is single-GPU; these are the
 * collectives SURVEY 8e assigns to the path: ONE flat gradient all-reduce per step and the
 * all-gather of the MMD sample rows.
 *   gda_rccl_load       bind librccl at run time (dlopen `path`; NULL/"" = "librccl.so"); pass the
 *                       copy the host framework already loaded so one RCCL lives in the process.
 *                       Without it every function below returns GDA_E_UNSUPPORTED.
 *   gda_comm_unique_id  rank 0 fills 128 bytes (ncclUniqueId) and ships them to the other ranks
 *                       out of band (the host uses its rendezvous store / a broadcast)
 *   gda_comm_init_rank  collective over all ranks; returns the communicator handle
 *   gda_allreduce_f32   in-place sum of buf[count] over the ranks
 *   gda_allgather_f32   recv[rank * count_per_rank ...] = send of that rank
 * ---------------------------------------------------------------------------- */
typedef void* gda_comm_t;
int gda_rccl_load(const char* path);
int gda_comm_unique_id(void* id_out, size_t bytes);
int gda_comm_init_rank(const void* id, size_t bytes, int nranks, int rank, gda_comm_t* comm_out);
int gda_comm_destroy(gda_comm_t comm);
int gda_allreduce_f32(float* buf, int64_t count, gda_comm_t comm, gda_stream_t stream);
int gda_allgather_f32(const float* send, float* recv, int64_t count_per_rank, gda_comm_t comm,
                      gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * View attention of UDAGCN's dual-view encoder (pygda/nn/attention.py:51-54):
 *   stacked = stack(inputs, dim=1); weights = softmax(dense_weight(stacked), dim=1); out = sum(stacked * weights, dim=1)
 * fused: s_k = x_k . w + b, a = softmax_k(s), out = sum_k a_k x_k, one pass each way, no [N, K, h] temporary.
 *   x     HOST array of n_views (2..4) device pointers, view k = [n, h] rows with leading dimension ld[k]
 *   w [h], b [1]   dense_weight.weight / .bias (device);  att [n, n_views] receives the weights (kept for backward)
 * Backward: gx[k] ([n, h] contiguous, or NULL for a view that needs no gradient), gw [h], gb [1]; fixed-order sums
 * (scratch from gda_attention_workspace_bytes).  h % 4 == 0, h <= 512, 16-byte aligned rows, else GDA_E_UNSUPPORTED.
 * ---------------------------------------------------------------------------- */
size_t gda_attention_workspace_bytes(int64_t n, int64_t h);
int gda_attention_fuse_fwd_f32(int n_views, const float* const* x /* HOST array */, const int64_t* ld /* HOST array */,
                               int64_t n, int64_t h, const float* w, const float* b, float* out, int64_t ldo,
                               float* att, gda_stream_t stream);
int gda_attention_fuse_bwd_f32(int n_views, const float* const* x /* HOST array */, const int64_t* ld /* HOST array */,
                               int64_t n, int64_t h, const float* w, const float* att, const float* gout, int64_t ldg,
                               float* const* gx /* HOST array */, float* gw, float* gb,
                               void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Step epilogue: the Adam update.
 *
 * gda_adam_multi_f32: one torch.optim.Adam step (amsgrad off, maximize off) over up to
 *   GDA_ADAM_MAX_TENSORS fp32 tensors in one launch -- the optimiser of every trainer
 *   (pygda/models/a2gnn.py:290-294).  grad += weight_decay * param; m = lerp(m, grad, 1-beta1);
 *   v = beta2 v + (1-beta2) grad^2; param -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps).
 *   `step` of every tensor is a device float holding t-1 (torch keeps one per parameter); it is
 *   incremented first (capturable: no host state).  Tensors with numel 0 are skipped.
 * ---------------------------------------------------------------------------- */
#define GDA_ADAM_MAX_TENSORS 48
typedef struct gda_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    float* step;
    int64_t numel;
} gda_adam_tensor;
int gda_adam_multi_f32(const gda_adam_tensor* tensors /* HOST array */, int n_tensors, float lr, float beta1,
                       float beta2, float eps, float weight_decay, gda_stream_t stream);
/* The same update in TWO calls, so that the counter bump leaves the tail of the step (it sat between the last
 * gradient kernel and the update, 8 us + a launch gap on the critical path of a replayed step):
 *   gda_step_bump          at the START of a training step: `*counter += 1` (the device step counter the fused
 *                          dropout kernels key their Philox streams on; may be NULL) and `*steps[k] += 1` for the
 *                          n_steps Adam step counters (device floats; HOST array of device pointers, n_steps <=
 *                          GDA_ADAM_MAX_TENSORS) -- one launch;
 *   gda_adam_multi_ex_f32  flags & GDA_ADAM_STEPS_BUMPED: the step counters already hold t (no increment launch). */
#define GDA_ADAM_STEPS_BUMPED 1
int gda_step_bump(int64_t* counter, float* const* steps /* HOST array */, int n_steps, gda_stream_t stream);
int gda_adam_multi_ex_f32(const gda_adam_tensor* tensors /* HOST array */, int n_tensors, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int flags, gda_stream_t stream);
/* ... with a gradient that arrives as TWO contributions: grad2[k] (HOST array of device pointers, entries or the array
 * itself may be NULL) is added to tensors[k].grad inside the update -- `g = grad + grad2`, one rounding, the value
 * autograd's accumulation of the two would have stored.  A parameter used by two branches of a step (A2GNN's source
 * and target passes share every layer, pygda/models/a2gnn.py:181-193) otherwise costs one elementwise launch per
 * tensor between the last gradient kernel and the update. */
int gda_adam_multi_sum_f32(const gda_adam_tensor* tensors /* HOST array */, const float* const* grad2 /* HOST array */,
                           int n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay, int flags,
                           gda_stream_t stream);
/* gda_wgan_critic_f32 (above) AND the critic optimiser's step in the same two launches: the body of one iteration of
 * AdaGCN's critic loop, pygda/models/adagcn.py:169-183 (`loss.backward(); self.c_optimizer.step()`, the optimiser built
 * at :271-275).  params[0..3] (HOST array) = W1 [a, h], b1 [a], w2 [a], b2 [1] with their torch.optim.Adam state; the
 * gradients are written to params[k].grad (as gda_wgan_critic_f32 writes gW1 ...), every step counter is incremented
 * and the parameters are updated by the thread that formed the gradient -- same arithmetic as gda_adam_multi_f32, same
 * bits.  Only where gda_wgan_critic_f32 takes its two-launch matrix-core path (h in {64, 96, 128}, a % 4 == 0, 16-byte
 * aligned es / et / W1): GDA_E_UNSUPPORTED otherwise, nothing launched. */
int gda_wgan_critic_adam_f32(const float* es, int64_t n_s, const float* et, int64_t n_t, int h,
                             const int32_t* idx_s, const int32_t* idx_t, const float* alpha, int64_t n_i,
                             int a, float dropout_p, uint64_t seed, const int64_t* step, uint32_t site,
                             float gp_weight, float* loss, const gda_adam_tensor* params /* HOST array of 4 */,
                             float lr, float beta1, float beta2, float eps, float weight_decay,
                             void* workspace, size_t workspace_bytes, gda_stream_t stream);

/* ------------------------------------------------------------------------------
 * Tall-skinny fp32 GEMMs on the matrix cores: the dense projection of the hidden / classifier
 * layers and its two gradient products (`self.lin(x)`, pygda/nn/prop_gcn_conv.py:204,
 * cached_gcn_conv.py:129, and autograd's dgrad / wgrad).  One operand is tall (rows = nodes).
 *   GDA_GEMM_NT  C[M,N] = A[M,K] * B[N,K]^T      forward  y  = x W^T
 *   GDA_GEMM_NN  C[M,N] = A[M,K] * B[K,N]        dgrad    gx = gy W
 *   GDA_GEMM_TN  C[M,N] = A[K,M]^T * B[K,N]      wgrad    gW = gy^T x  (K = nodes; deterministic split
 *                                                over row slabs, scratch from gda_gemm_workspace_bytes)
 * Row-major with leading dimensions in elements; C must not alias A or B.
 * The same three modes serve activations kept COLUMN-MAJOR (hT [out, ld] next to the LDS-resident K-step
 * kernel): forward hT = W x^T is NT with A = W, B = x; dgrad gx = gT^T W is TN with A = gT; wgrad
 * gW = gT x is NN whose K is the node count -- NN with K >= 1024 and M, N <= 512 takes the deterministic
 * row-slab split as well (gda_gemm_workspace_bytes says how much scratch that needs).
 * ---------------------------------------------------------------------------- */
#define GDA_GEMM_NT 0
#define GDA_GEMM_NN 1
#define GDA_GEMM_TN 2
size_t gda_gemm_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K);
int gda_gemm_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                 const float* B, int64_t ldb, float* C, int64_t ldc,
                 void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* The same with the two by-products of a biased layer `y = x W^T + b` (prop_gcn_conv.py:204,212-213 with
 * prop_nums = 0): `bias` [N] (NT only) is added in the forward epilogue; `colsum` [M] (TN only) receives
 * sum_k A[k, m] -- with A = gy that is the bias gradient, formed beside the weight gradient gy^T x from the tiles
 * the kernel stages anyway (deterministic row-slab partials like the product itself).  Either may be NULL. */
int gda_gemm_ex_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                    const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colsum,
                    void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* The same products for the shapes of sampled sub-graphs (10^5 .. 10^7 rows against a weight whose extents are 128 or
 * 256; pygda/nn/prop_gcn_conv.py:205 on a NeighborLoader batch): the weight lives in registers as ready MFMA operands,
 * only the tall operand streams through LDS (NT / NN), and the weight gradient (TN) reads both operands along their
 * rows with the whole output of a row slab in accumulators (csrc/gda_gemm.hip, "tall products").
 *   NT / NN: N in {128, 256}, K in {128, 256}, A 16-byte aligned with lda % 4 == 0;
 *   TN: M == 128 (the output rows = columns of A), N in {128, 256}, K = the node count; colsum as in gda_gemm_ex_f32.
 * GDA_E_UNSUPPORTED outside that envelope -- callers then use gda_gemm_ex_f32, which takes any shape. */
size_t gda_gemm_tall_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K);
/* ... and for the classifier projection h -> C (C <= 8 classes, `self.cls` of a2gnn_base.py:62 / grade_base.py:66) at
 * 10^5 rows, where a matrix-core tile is 92 % padding: memory-bound vector kernels, fixed-order sums.
 *   NT: N <= 8, K in {32, 64, 128, 256} (+ bias);  NN: K <= 8, N in {32, .., 256};  TN: M <= 8, N in {32, .., 256}
 *   (colsum as in gda_gemm_ex_f32).  GDA_E_UNSUPPORTED outside. */
size_t gda_gemm_skinny_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K);
int gda_gemm_skinny_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                        const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colsum,
                        void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* Data gradient with the upstream activation's backward in its epilogue: C[M, N] = (y > 0) * (A[M, K] B[K, N]) / (1 - p) with
 * y [M, N] (ld = ldm) the output of dropout(relu(.), p) -- gda_relu_dropout_bwd_f32(A B, y) in one launch, same values.
 * Envelopes: tall (N, K in {128, 256}, the split-fp16 kernel) and skinny (K <= 8, N in {32, 64, 128, 256});
 * GDA_E_UNSUPPORTED elsewhere. */
int gda_gemm_nn_mask_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, const float* y, int64_t ldm, float p, gda_stream_t stream);
/* The sampled batch's first projection without its gather pass, and projections with the conv layer's activation
 * (pygda/nn/a2gnn_base.py:135-138) in the epilogue.  Forward (NT): C = act(A[arow] B^T + bias).
 *   arow (may be NULL): int64 [M] device row ids -- row i of the tall operand is A[arow[i]] (x[n_id] of a sampled batch).
 *   act_mode 0: none; 1: C [M, N] = dropout_site0(relu(.)); 2: C [2M, N], rows i / M + i = two independent draws of the same
 *   pre-activation (gda_relu_dropout_pair_fwd_f32's stacked pair).  Keep-bits as gda_relu_dropout_fwd_f32 / _pair_fwd_f32
 *   would draw them on the [M, N] pre-activation; p <= 0: relu only.
 * Envelope: N, K in {128, 256}, ldc == N, 16-byte aligned operands, the split-fp16 kernel; GDA_E_UNSUPPORTED elsewhere.
 * gda_gemm_tall_wgrad_gather_f32: the TN form of gda_gemm_tall_f32 (C[128, N] = A[Krows, 128]^T X[xrow], colsum[128]) reading
 * x through the batch's node ids; same workspace (gda_gemm_tall_workspace_bytes(TN, 128, N, Krows)). */
int gda_gemm_tall_fwd_ex_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const int64_t* arow,
                             const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                             int act_mode, float p, uint64_t seed, const int64_t* step_dev, uint32_t site0,
                             uint32_t site1, gda_stream_t stream);
int gda_gemm_tall_wgrad_gather_f32(int64_t N, int64_t Krows, const float* A, int64_t lda, const float* X, int64_t ldx,
                                   const int64_t* xrow, float* C, int64_t ldc, float* colsum,
                                   void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_gemm_tall_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                      const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colsum,
                      void* workspace, size_t workspace_bytes, gda_stream_t stream);


/* C = A * A of a CSR operator on the host (threaded, deterministic order), for STATIC full-batch
 * graphs: a K-step propagation (pygda/nn/prop_gcn_conv.py:208-210) then takes K/2 dependent
 * launches -- at citation-graph sizes a launch costs its latency, not its edges.  rowptr / colidx /
 * val are HOST arrays.  The result comes back as an edge list (src = row, dst = column, w = value),
 * rows ascending, columns ascending inside a row; EMPTY when it would hold more than max_nnz
 * entries (max_nnz < 0: no limit). */
int gda_csr_square_host(const int32_t* rowptr_host, const int32_t* colidx_host, const float* val_host,
                        int64_t N, int threads, int64_t max_nnz, gda_edge_list** out);

/* ------------------------------------------------------------------------------
 * Source classification loss: mean_i( -log_softmax(logits_i)[labels_i] ), forward and backward.
 * Replaces F.nll_loss(F.log_softmax(source_logits, dim=1), source_data.y)
 * (pygda/models/a2gnn.py:182 and the same line of every trainer).  logits [N, C] fp32 (C <= 64),
 * labels [N] int64 in [0, C); loss [1]; grad_loss [1] DEVICE scalar; grad_logits [N, C].
 * Labels are trusted (no ignore_index): the trainers pass dataset labels.
 * ---------------------------------------------------------------------------- */
size_t gda_softmax_nll_workspace_bytes(void);
/* Mean entropy of the clamped softmax: loss = mean_i sum_c -q_ic log q_ic, q = clamp(softmax(logits_i), clamp_min, 1) --
 * UDAGCN's target term (pygda/models/udagcn.py:193-197: softmax, clamp(min=1e-9, max=1.0), -p log p, sum, mean: seven library
 * launches forward and their autograd twins) as one row kernel + the loss kernels' final fold each way.  The clamp's
 * gradient is the reference's (passes where clamp_min <= p <= 1).  workspace: gda_softmax_nll_workspace_bytes(). */
int gda_softmax_entropy_fwd_f32(const float* logits, int64_t ld, int64_t N, int C, float clamp_min, float* loss,
                                void* workspace, size_t workspace_bytes, gda_stream_t stream);
int gda_softmax_entropy_bwd_f32(const float* logits, int64_t ld, int64_t N, int C, float clamp_min,
                                const float* grad_loss, float* grad_logits, int64_t ldg, gda_stream_t stream);
int gda_softmax_nll_fwd_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                            float* loss, void* workspace, size_t workspace_bytes, gda_stream_t stream);
/* _ex: stats (may be NULL) [2] doubles = {loss, number of rows whose argmax (first maximum, as torch.argmax) is the
 * label}: the per-epoch log line of every trainer (loss.item(), source micro-F1 = accuracy of the single-label
 * predictions, pygda/models/a2gnn.py:326-343) as a by-product of the pass that reads the logits anyway. */
int gda_softmax_nll_fwd_ex_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                               float* loss, double* stats, void* workspace, size_t workspace_bytes,
                               gda_stream_t stream);
int gda_softmax_nll_bwd_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                            const float* grad_loss, float* grad_logits, int64_t ldg, gda_stream_t stream);
/* _nv: n_valid (DEVICE int64[1], or NULL = all N rows): only rows [0, *n_valid) are real -- the loss is THEIR mean, the
 * count is theirs, the backward pass writes exact zeros into the rows behind them.  For batches padded to a static
 * capacity (the captured sampled step): the row count is read on the device, the launch is the same for every batch. */
int gda_softmax_nll_fwd_nv_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                               const int64_t* n_valid, float* loss, double* stats, void* workspace,
                               size_t workspace_bytes, gda_stream_t stream);
int gda_softmax_nll_bwd_nv_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t N, int C,
                               const int64_t* n_valid, const float* grad_loss, float* grad_logits, int64_t ldg,
                               gda_stream_t stream);

/* PPMI graph construction on the DEVICE: the estimator and the walks of gda_ppmi_build_host (same
 * counter-based generator, so both builders count the same visits), as sorts + run-length counts.
 *   src/dst [E] int64 device; out_src/out_dst [cap] int64, out_w [cap] fp32 with
 *   cap = N * passes * path_len (upper bound of distinct (start, visited) pairs); out_count [1] int64
 *   DEVICE: the number of pairs written, sorted by (src, dst).
 * Returns GDA_E_UNSUPPORTED when 2E or N*passes*path_len does not fit int32 (use the host builder). */
size_t gda_ppmi_workspace_bytes(int64_t E, int64_t N, int path_len, int passes);
int gda_ppmi_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int path_len, int passes,
                   uint64_t seed, int64_t* out_src, int64_t* out_dst, float* out_w, int64_t* out_count,
                   void* workspace, size_t workspace_bytes, gda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GDA_HIP_H */
