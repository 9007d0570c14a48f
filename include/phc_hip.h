/*
 * phc_hip.h — C ABI of libphc_hip.so, the MI355X (gfx950) engine behind the
 * PHC-GNN message-passing path.
 *
 * The reference (bayer-science-for-a-better-life/phc-gnn) has no FFI layer: its
 * hot path drops from Python straight into third-party ATen / torch-scatter
 * kernels.  Each entry point below replaces one such group of
 * call sites; the reference location is cited per function
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (HBM) unless a comment says "host";
 *   - matrices are dense row-major fp32; index arrays are int32 unless the
 *     parameter is named *_i64 (the reference's edge_index / batch are int64);
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.
 *     Work is only enqueued: no call synchronises, allocates or frees device
 *     memory, so every call is legal under hipGraph stream capture;
 *   - the caller owns every buffer, workspaces included (query *_ws_bytes);
 *   - return value: 0 = PHC_OK, negative = error (see phc_error_string); no
 *     C++ exception crosses the boundary; phc_last_error() returns a
 *     thread-local detail message for the last failing call on this thread;
 *   - thread-safe for calls on distinct streams; the only process-global state
 *     is the opt-in launch profiler (phc_prof_*).
 *
 * Feature layout is the reference's component-major one: column c*k + kappa
 * is hypercomplex component c, channel kappa (models.py:225, norm.py:32).
 */
#ifndef PHC_HIP_H
#define PHC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHC_ABI_VERSION 8

#define PHC_OK 0
#define PHC_ERR_INVALID_ARG (-1)   /* null pointer, negative size, bad enum        */
#define PHC_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels implement      */
#define PHC_ERR_WORKSPACE (-3)     /* workspace pointer null or too small           */
#define PHC_ERR_LAUNCH (-4)        /* HIP runtime reported an error at launch       */
#define PHC_ERR_NO_DEVICE (-5)     /* no gfx950 device visible to this process      */

/* message activation  phi  (phc/quaternion/activations.py:134-147) */
#define PHC_ACT_IDENTITY 0
#define PHC_ACT_RELU 1
#define PHC_ACT_LRELU 2            /* nn.LeakyReLU(), slope 0.01                    */
#define PHC_ACT_ELU 3              /* nn.ELU(), alpha 1                             */
#define PHC_ACT_SWISH 4            /* x * sigmoid(x), activations.py:13-35          */
#define PHC_ACT_SELU 5

int phc_version(void);                       /* PHC_ABI_VERSION of the loaded library */

/* Index validation.  The reference fails loudly on a bad index: nn.Embedding raises IndexError for a feature value
 * outside its vocabulary (phc/quaternion/encoder.py:43-51), torch-scatter / index_select for a node id outside
 * [0, N) (messagepassing.py:60,136).  The engine's kernels never dereference such a value: every kernel that reads a
 * caller-supplied index maps an out-of-range value to 0 (so no out-of-bounds access can happen), and the FIRST kernel
 * that sees it ORs one of the bits below into the int32 error word registered for the current device.  The word is
 * sticky and read lazily by the host (the Python layer raises IndexError at its next check point, as a device-side
 * assert would surface at the next synchronisation).  dev_word = NULL unregisters.  Opt-in handle: the only mutable
 * library state besides the launch profiler. */
#define PHC_IDX_ERR_EDGE 1         /* edge_index entry outside [0, num_nodes)                                  */
#define PHC_IDX_ERR_ATOM 2         /* integer node feature outside its vocabulary (phc_encode_fwd)             */
#define PHC_IDX_ERR_BOND 4         /* integer edge feature outside its vocabulary (packed codes / enc kernels) */
#define PHC_IDX_ERR_SEGMENT 8      /* segment id (batch vector) outside [0, num_segments)                      */
#define PHC_ERR_RENDEZVOUS 16      /* a fused launch's grid-wide exchange timed out (workgroups not co-resident):
                                    * its outputs are invalid (phc_downstream_fwd / _bwd poll loops are bounded)        */
int phc_set_error_word(int device, int32_t *dev_word);
const char *phc_error_string(int code);      /* host, static storage                  */
const char *phc_last_error(void);            /* host, thread-local storage            */
int phc_device_info(int *num_cu, int *lds_bytes_per_cu, char *arch, int arch_len);   /* host out-params */

/* ------------------------------------------------------------------------- *
 * PHM linear transform
 * ------------------------------------------------------------------------- */

/* H[a*kin+kap, c*kout+pi] = sum_i A[i,a,c] * W[i,kap,pi]     ([n*kin, n*kout])
 * Replaces kronecker_product_einsum_batched(...).sum(0)
 * (phc/hypercomplex/kronecker.py:35-48 + layers.py:214).  The forward/backward
 * kernels below never need this buffer (they assemble H tiles in LDS); it is
 * exported for inspection, tests and regularisers. */
int phc_phm_build_H(const float *A, const float *W, int n, int kin, int kout, float *H, void *stream);

/* y[N, n*kout] = x[N, n*kin] . H(A, W)  (+ b[n*kout])  (+ residual[N, n*kout])
 * Replaces matvec_product_new (phc/hypercomplex/layers.py:198-219); the optional
 * residual is PHMConv's `transform(agg) + x` (messagepassing.py:62-65).
 * A is staged in LDS, H tiles are assembled in LDS from W, fp32 MFMA.
 * b, residual may be NULL.  y must not alias x. */
int phc_phm_linear_fwd(const float *x, const float *A, const float *W, const float *b, const float *residual,
                       int64_t N, int n, int kin, int kout, float *y, void *stream);

/* Gradients of phc_phm_linear_fwd w.r.t. x, W, A, b given dy[N, n*kout]
 * (what autograd derives from layers.py:214-217; formulas in SURVEY.md §8 a4):
 *   dx = dy . H^T ; dH = x^T . dy ; dW[i] = sum_{a,c} A[i,a,c] dH[a,:,c,:] ;
 *   dA[i,a,c] = sum W[i] * dH[a,:,c,:] ; db = sum_N dy.
 * dx, dA, db may be NULL (not computed); dW may be NULL only if dA and db are.
 * Deterministic (fixed-order split-K slabs, no atomics). */
size_t phc_phm_linear_bwd_ws_bytes(int64_t N, int n, int kin, int kout);
int phc_phm_linear_bwd(const float *x, const float *A, const float *W, const float *dy,
                       int64_t N, int n, int kin, int kout,
                       float *dx, float *dW, float *dA, float *db,
                       void *ws, size_t ws_bytes, void *stream);

/* Operand matrices of MANY PHM layers in one launch: H = sum_i A_i (x) W_i  [n*kin, n*kout] and, when
 * HT is not NULL, its transpose [n*kout, n*kin].  A training step builds them once, right after the
 * weights changed, and runs every y = x.H (and dx = dy.H^T) of the step with phc_gemm_prebuilt: the
 * GEMM then loads its operand panel with one round of coalesced reads instead of assembling it from
 * A and W in every workgroup (3.6 us of a 13 us launch at molecule scale).  `jobs`: HOST array. */
typedef struct phc_phm_H_job {
    const float *A, *W;            /* device: [n,n,n], [n,kin,kout] */
    float *H, *HT;                 /* device outputs; HT may be NULL */
    int32_t n, kin, kout;
    int32_t flags;                 /* bit 0: n == 1 and W is an nn.Linear weight [kout, kin]: H = W^T, HT unused (it is W);
                                    * flags == 2: zero fill — H[0 .. kin * kout) = 0, n = 1, everything else NULL (the step's
                                    * statistics accumulators are cleared by the launch that opens the step) */
    /* Optional (ABI 4): the same matrices in "k4" form for the H-stationary GEMM — a [Kd, Nd] operand stored as
     * [Kd / 4][Nd] float4, slot (k / 4) * Nd + j = rows 4 (k / 4) .. +3 of column j (Kd % 4 == 0; PHM jobs only).  A lane of
     * v_mfma_f32_16x16x4_f32 then fetches its operands of four consecutive k-steps with ONE 16-byte access. */
    float *Hk4, *HTk4;
    /* flags == 3 (ABI 4): the read-out's gate as ONE linear map.  pooling.py:57-66 computes Linear_{h -> kg}(PHMLinear_{in -> h}(x))
     * with nothing between the two maps: x . M + c with M = H(A, W) . gate_w^T [in, kg] and c = phm_b . gate_w^T + gate_b is the
     * same function (fp32 round-off apart) with one small operand.  The job writes H := M, HT := M^T (may be NULL) and gate_c;
     * Hk4 / HTk4 must be NULL.  gate_w [kg, h] is the nn.Linear weight, gate_b [kg] / phm_b [h] may be NULL. */
    const float *gate_w, *gate_b, *phm_b;
    float *gate_c;
    int32_t gate_k;
} phc_phm_H_job;
int phc_phm_build_H_batched(const phc_phm_H_job *jobs, int num_jobs, void *stream);
/* ABI 5: the same launch with the step's ATOM ENCODER riding in it as passenger workgroups (encoder.py:31-34:
 * out[i, c k + kap] = sum_f T_{c,f}[idx[i, f]][kap], exactly phc_encode_fwd): both depend on the parameters and the batch only, and
 * both are latency-bound — one launch instead of two at the head of a training step.  tables: DEVICE array of n * F table pointers
 * ([c][f] -> [vocab_f, k] fp32), vocab: host array of F sizes, out [N, n k] 16-byte aligned; F <= 10, k % 4 == 0
 * (phc_encode_front_supported); index validation as phc_encode_fwd (PHC_IDX_ERR_ATOM).  enc == NULL: phc_phm_build_H_batched. */
typedef struct phc_encode_front {
    const int64_t *idx_i64;
    const float *const *tables;
    const int32_t *vocab;
    float *out;
    int64_t N;
    int32_t F, n, k;
} phc_encode_front;
int phc_encode_front_supported(int64_t N, int F, int n, int k);
int phc_phm_build_H_batched_ex(const phc_phm_H_job *jobs, int num_jobs, const phc_encode_front *enc, void *stream);
/* Parameter gradients of that gate map from dM^T = dg^T . x [kg, in] and dbg = sum dg [kg] (what a dense parameter-gradient job
 * with x and dg leaves behind): dH = dM . gate_w is re-evaluated where needed (never stored), dA / dW its contractions with W / A,
 * dWg = dM^T . H + dbg (x) phm_b, db = dbg . gate_w.  H [in, h] is the PHM layer's operand matrix (phc_phm_build_H*).  Outputs may be
 * NULL.  One launch, fixed summation order. */
int phc_gate_param_grads(const float *dMT, const float *dbg, const float *H, const float *gate_w, const float *phm_b, const float *A,
                         const float *W, int n, int kin, int kout, int kg, float *dA, float *dW, float *db, float *dWg, void *stream);

/* y[N, Nout] = x[N, K] . M[K, Nout] (+ b[Nout]) (+ residual[N, Nout]) with a prebuilt row-major M
 * (H for the forward, H^T for dx; is_dx only labels the launch for the profiler).  Streaming MFMA
 * kernel; K <= 256, K % 4 == 0, x and M 16-byte aligned — otherwise PHC_ERR_UNSUPPORTED and the
 * caller uses phc_phm_linear_fwd / _bwd.
 * Mk4 (ABI 4, may be NULL): the same operand in k4 form (phc_phm_H_job.Hk4 / HTk4).  With it, K in {128, 256} and Nout in
 * {128, 256} run on the H-stationary kernel (csrc/gemm_hs_device.h): operand columns resident in registers, x streamed through
 * an LDS ring by LDS-DMA, statistics / operand-transform variants in their accumulator forms.  Same results up to the fp32
 * summation order; every phc_gemm_prebuilt* entry point takes the pair (M, Mk4) the same way. */
int phc_gemm_prebuilt(const float *x, const float *M, const float *Mk4, const float *b, const float *residual, int64_t N, int K,
                      int Nout, float *y, int is_dx, void *stream);

/* The parameter gradients (dW, dA, db) of MANY PHM linear layers in three launches.
 * They are off the backward pass's critical path (only the optimizer reads them), and at
 * molecule scale one layer's dH = x^T.dy cannot fill 256 CUs: a training step runs
 * phc_phm_linear_bwd with dW = dA = db = NULL (dx only) per layer, keeps x and dy alive, and
 * submits all layers here once the backward pass is over (phc_gnn_amd/ops.py:
 * deferred_param_grads).  Same formulas and the same fixed-order split-K summation as
 * phc_phm_linear_bwd; `jobs` is a HOST array consumed at call time; every job needs N > 0. */
typedef struct phc_phm_grad_job {
    const float *x, *dy, *A, *W;   /* device: [N, n*kin], [N, n*kout], [n,n,n], [n,kin,kout] */
    float *dW, *dA, *db;           /* device outputs; dA / db may be NULL                     */
    int64_t N;
    int32_t n, kin, kout;
    int32_t flags;                 /* bit 0: dW is written as [n][kout][kin] (an nn.Linear weight when n == 1) */
    /* Optional: the layer's input was a batch-norm that phc_gemm_prebuilt_xf applied on the fly, so the
     * normalised tensor does not exist in HBM: x holds the norm's INPUT, and the kernel evaluates
     * act(x * scale + shift) per column (rows >= *xf_n_valid_dev read as 0) while staging the operand.
     * xf_stat: the norm's save_stat [6 * n*kin] (NULL: x is used as is). */
    const float *xf_stat;
    const int32_t *xf_n_valid_dev;
    int32_t xf_act, reserved;
} phc_phm_grad_job;
size_t phc_phm_param_grads_ws_bytes(const phc_phm_grad_job *jobs, int num_jobs);
int phc_phm_param_grads(const phc_phm_grad_job *jobs, int num_jobs, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Graph structure: destination-sorted (CSR) or source-sorted (CSC) edge lists
 * ------------------------------------------------------------------------- */

/* Stable counting sort of E items by key_i64[e] in [0, num_keys):
 *   rowptr[num_keys+1], perm[E] = item ids grouped by key, ascending id inside
 *   a group (so every later reduction has ONE fixed summation order),
 *   col[E] = other_i64[perm[e]] narrowed to int32 (col = perm if other is NULL).
 * CSR for the forward pass: key = edge_index[1] (targets), other = edge_index[0].
 * CSC for the backward pass: key = edge_index[0], other = edge_index[1].
 * Pooling segments: key = batch, other = NULL.
 * Replaces the index plumbing inside PyG 1.6.1 MessagePassing.propagate /
 * torch_scatter.scatter (call sites messagepassing.py:60,136; pooling.py:14).
 * Keys outside [0, num_keys) and `other` values outside [0, num_other) count as key / value 0 and raise
 * PHC_IDX_ERR_EDGE (other given) or PHC_IDX_ERR_SEGMENT (other NULL) in the registered error word. */
size_t phc_csr_ws_bytes(int64_t E, int64_t num_keys);
int phc_csr_build(const int64_t *key_i64, const int64_t *other_i64, int64_t E, int64_t num_keys, int64_t num_other,
                  int32_t *rowptr, int32_t *col, int32_t *perm, void *ws, size_t ws_bytes, void *stream);

/* The same result as phc_csr_build for LARGE key spaces (BASELINE config 5: 10 M keys, 50 M items), as a two-level
 * counting sort: items are partitioned by key >> shift into <= 16384 buckets of consecutive keys (per-chunk LDS histograms,
 * one global atomic per non-empty (chunk, bucket)), then one workgroup per bucket sorts its items with LDS counters and makes
 * every row ascending in item id.  Needs E >= 1 and num_keys <= 16384 * 4096; ws: phc_csr_bucketed_ws_bytes. */
size_t phc_csr_bucketed_ws_bytes(int64_t E, int64_t num_keys);
int phc_csr_build_bucketed(const int64_t *key_i64, const int64_t *other_i64, int64_t E, int64_t num_keys, int64_t num_other,
                           int32_t *rowptr, int32_t *col, int32_t *perm, void *ws, size_t ws_bytes, void *stream);

/* rowptr[num_keys+1] for keys that are already NON-DECREASING (PyG's collated `batch`
 * vector: graph id per node): rowptr[s] = first position with key >= s.  Segment members are
 * then simply positions rowptr[s] .. rowptr[s+1]-1 (identity order).  Undefined result for
 * unsorted keys — use phc_csr_build for those. */
int phc_segment_ptr_sorted(const int64_t *key_i64, int64_t n, int64_t num_keys, int32_t *rowptr, void *stream);

/* ------------------------------------------------------------------------- *
 * Neighbour aggregation (sum)
 * ------------------------------------------------------------------------- */

/* out[d, :] = sum_{e in CSR row d} phi( x[col[e], :] + ea[perm[e], :] )  (+ x[d, :] if add_self)
 * for d < n_dst; x has at least max(col)+1 rows (n_dst rows suffice for a
 * square graph), ea[E, h] is in ORIGINAL edge order and may be NULL.
 * Replaces index_select + message + scatter(reduce='add') + the self-loop add
 * (messagepassing.py:57-60,72-74 / :133-138,144-146).  One wavefront per
 * destination row, coalesced 16-byte-per-lane row gathers, no atomics.
 * add_self is a bit set: 1 = + x[d, :] (the self loop), 2 = + the previous contents of out[d, :] (accumulate: the
 * halo edges of a destination-range shard are added onto the local edges' sums when their rows have landed). */
int phc_aggregate_fwd(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col,
                      const int32_t *perm, int64_t n_dst, int h, int act, int add_self,
                      float *out, void *stream);

/* Backward of phc_aggregate_fwd, walking the SOURCE-sorted structure
 * (rowptr_t[n_src+1], col_t[e] = destination of the e-th out-edge, perm_t[e] =
 * its original edge id):
 *   g_e = phi'(x[s]+ea[e]) * dout[dst(e)] ;  dea[e] = g_e ;
 *   dx[s] = sum_{e out of s} g_e  (+ dout[s] if add_self and s < n_dst).
 * x and ea are only read when act != identity; dea may be NULL. */
int phc_aggregate_bwd(const float *dout, const float *x, const float *ea, const int32_t *rowptr_t,
                      const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int64_t n_dst, int h,
                      int act, int add_self, float *dx, float *dea, void *stream);

/* ------------------------------------------------------------------------- *
 * Softmax neighbour aggregation (the reference's default molhiv setting)
 *   m_e = phi(x[col[e]] + ea[perm[e]]);  per destination d and per column:
 *   out[d] = sum_e softmax_e(beta * m_e) * m_e        (empty rows: 0)
 * Replaces scatter_softmax + scatter_sum in PHMConvSoftmax / PHMGINEConvSoftmax.aggregate
 * (messagepassing.py:211-214, 297-300; torch-scatter 2.0.5: max-shifted, eps 1e-12 in the
 * denominator).  beta_dev is a DEVICE scalar (the learnable nn.Parameter).  lse[n_dst, h]
 * (max + log(sum + eps)) is written for the backward.
 * Backward (source-sorted walk): dx[n_src, h], dea[E, h] (may be NULL) and
 * phc_softmax_bwd_num_partials(n_src) partial sums of d/d beta (caller adds them up).
 * ------------------------------------------------------------------------- */
int phc_aggregate_softmax_fwd(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col,
                              const int32_t *perm, const float *beta_dev, int64_t n_dst, int h, int act, float *out,
                              float *lse, void *stream);
int phc_softmax_bwd_num_partials(int64_t n_src);
int phc_aggregate_softmax_bwd(const float *dout, const float *out, const float *lse, const float *x, const float *ea,
                              const float *beta_dev, const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t,
                              int64_t n_src, int h, int act, float *dx, float *dea, float *dbeta_partial, void *stream);

/* The same with the layer's self loop folded in (PHMConvSoftmax / PHMGINEConvSoftmax.forward add x to the aggregate,
 * messagepassing.py:222-231, 303-309): the forward also writes out_self = out + x[d] (x has >= n_dst rows; `out`
 * stays the plain aggregate for the backward), the backward adds dout[s] to dx[s] for s < self_n_dst.  One launch
 * each way instead of the aggregation plus an elementwise add. */
int phc_aggregate_softmax_fwd_self(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col,
                                   const int32_t *perm, const float *beta_dev, int64_t n_dst, int h, int act, float *out,
                                   float *out_self, float *lse, void *stream);
int phc_aggregate_softmax_bwd_self(const float *dout, const float *out, const float *lse, const float *x, const float *ea,
                                   const float *beta_dev, const int32_t *rowptr_t, const int32_t *col_t,
                                   const int32_t *perm_t, int64_t n_src, int h, int act, int64_t self_n_dst, float *dx,
                                   float *dea, float *dbeta_partial, void *stream);

/* ABI 6: the model-level norm of the layer in FRONT folded into the softmax aggregation (what phc_aggregate_norm_encp_fwd does
 * for the sum): x = act(batch_norm(y)) (+ skip) (models.py:208-215) is evaluated on the fly for every gathered row and for the row
 * itself from the statistics accumulator the producing GEMM filled, written once as x_out, save_stat / running statistics left
 * behind by the first workgroup; then messagepassing.py:211-214 / 297-300 as phc_aggregate_softmax_enc_fwd (identity message
 * function).  Bit-identical to phc_bn_fwd_from_partials followed by phc_aggregate_softmax_enc_fwd.  h % 4 == 0, h <= 512. */
int phc_aggregate_softmax_norm_supported(int64_t N, int h, int F, int vocab_rows);
int phc_aggregate_softmax_norm_enc_fwd(const float *y, const uint64_t *norm_acc, const float *const *weight, const float *const *bias,
                                       float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked,
                                       int n_comp, float momentum, float eps, int norm_act, const int32_t *n_valid_dev,
                                       const float *skip, float *save_stat, float *x_out, const uint32_t *codes_sorted,
                                       const float *const *tables, const int32_t *vocab, int F, int n, int k, const int32_t *rowptr,
                                       const int32_t *col, const float *beta_dev, int64_t n_dst, float *out, float *out_self,
                                       float *lse, void *stream);

/* The softmax aggregation with the bond embedding read in place (N1 on the N2 operators): instead of an [E, h] edge
 * tensor the kernels take the packed table rows of every edge (phc_pack_edge_codes) and the n*F table pointers, as
 * phc_aggregate_encp_fwd does for the sum.  Forward: codes_sorted in destination-sorted order (perm = the CSR's perm).
 * Backward: codes_src in SOURCE-sorted order (phc_pack_edge_codes with perm = perm_t); dea [E, n*k] (original order,
 * may be NULL) is the gradient of the embedded edge features, from which phc_encode_bwd[_batched] makes the table
 * gradients.  out_self / self_n_dst as in the *_self forms.  Needs F <= 4, sum(vocab) <= 256, k % 4 == 0. */
int phc_aggregate_softmax_enc_fwd(const float *x, const uint32_t *codes_sorted, const float *const *tables,
                                  const int32_t *vocab, int F, int n, int k, const int32_t *rowptr, const int32_t *col,
                                  const float *beta_dev, int64_t n_dst, int act, float *out, float *out_self, float *lse,
                                  void *stream);
int phc_aggregate_softmax_enc_bwd(const float *dout, const float *out, const float *lse, const float *x,
                                  const uint32_t *codes_src, const float *const *tables, const int32_t *vocab, int F,
                                  int n, int k, const float *beta_dev, const int32_t *rowptr_t, const int32_t *col_t,
                                  const int32_t *perm_t, int64_t n_src, int act, int64_t self_n_dst, float *dx,
                                  float *dea, float *dbeta_partial, void *stream);
/* ... with up to 8 further gradients of the same rows (residuals: HOST array of device pointers [n_src, n k]) added on the way out in
 * slot order: the skip connections' shares of dx (models.py:230-234) — the n-way sum launch behind the walk disappears. */
int phc_aggregate_softmax_enc_bwd_res(const float *dout, const float *out, const float *lse, const float *x, const uint32_t *codes_src,
                                      const float *const *tables, const int32_t *vocab, int F, int n, int k, const float *beta_dev,
                                      const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int act,
                                      int64_t self_n_dst, const float *const *residuals, int num_residuals, float *dx, float *dea,
                                      float *dbeta_partial, void *stream);

/* out[r, :] = src[idx[r], :]   (row gather; backward of a segment sum, e.g.
 * global_add_pool's gradient dx[i] = dout[batch[i]], pooling.py:14) */
int phc_gather_rows_i64(const float *src, const int64_t *idx_i64, int64_t n_rows, int h, float *out, void *stream);

/* ------------------------------------------------------------------------- *
 * Integer-feature encoders (atom / bond embeddings), read in place
 *
 * Reference: PHMEncoder = phm_dim independent IntegerEncoders, each the sum of
 * one nn.Embedding lookup per categorical feature
 * (phc/hypercomplex/encoder.py:31-34, phc/quaternion/encoder.py:43-51); the
 * model flattens the result to component-major rows (models.py:223-225,238-240).
 * `tables` is a DEVICE array of n*F device pointers, entry c*F+f -> table
 * T_{c,f} [vocab[f], k] row-major (the reference's separate Parameters, used
 * where they are); `vocab` is a HOST array of the F vocabulary sizes.
 * Gradients of all n*F tables are written to ONE flat buffer laid out
 * [c][f][vocab[f]][k]; the caller hands out per-table views of it.
 * ------------------------------------------------------------------------- */
#define PHC_MAX_ENC_FEATS 16

/* out[i, c*k + kap] = sum_f T_{c,f}[ idx[i,f] ][kap]        idx_i64 [N, F]
 * idx[i,f] outside [0, vocab[f]) reads row 0 and raises PHC_IDX_ERR_ATOM (see phc_set_error_word). */
int phc_encode_fwd(const int64_t *idx_i64, const float *const *tables, const int32_t *vocab, int64_t N, int F, int n, int k,
                   float *out, void *stream);

/* dT_{c,f}[v][kap] = sum_{i: idx[i,f]=v} dout[i, c*k+kap]    (replaces F*n embedding_dense_backward sorts) */
size_t phc_encode_bwd_ws_bytes(int64_t N, int F, const int32_t *vocab, int n, int k);
int phc_encode_bwd(const float *dout, const int64_t *idx_i64, int64_t N, int F, const int32_t *vocab, int n, int k,
                   float *dtables_flat, void *ws, size_t ws_bytes, void *stream);

/* The table gradients of SEVERAL encoders that were applied to one index tensor (e.g. the bond encoders of all
 * message-passing layers when the bond embedding is materialised, as the softmax aggregation needs it) in two
 * launches: jobs[q].dout [N, n*k] and jobs[q].dtables_flat [n][R][k] (x / tables are not read).  Same slabs and
 * fixed summation order as phc_encode_bwd.  `jobs` is declared with phc_table_grads_batched below. */
struct phc_table_grad_job;
size_t phc_encode_bwd_batched_ws_bytes(int64_t N, int F, const int32_t *vocab, int n, int k, int num_jobs);
int phc_encode_bwd_batched(const struct phc_table_grad_job *jobs, int num_jobs, const int64_t *idx_i64, int64_t N, int F,
                           const int32_t *vocab, int n, int k, void *ws, size_t ws_bytes, void *stream);

/* Aggregation with the edge features looked up from LDS-staged tables; the [E, h]
 * edge embedding of models.py:238-240 is never materialised:
 *   out[d] = sum_{e in CSR row d} phi( x[col[e]] + sum_f T_f[bond[perm[e], f]] ) (+ x[d])
 * bond_i64 [E, F] is in ORIGINAL edge order.  Needs k % 4 == 0 and
 * sum(vocab) * n * k * 4 <= 48 KiB (else PHC_ERR_UNSUPPORTED: use phc_encode_fwd +
 * phc_aggregate_fwd). */
int phc_aggregate_enc_fwd(const float *x, const int64_t *bond_i64, const float *const *tables, const int32_t *vocab,
                          const int32_t *rowptr, const int32_t *col, const int32_t *perm, int64_t n_dst, int F, int n, int k,
                          int act, int add_self, float *out, void *stream);

/* Packed variant (F <= 4 and sum(vocab) <= 256 — every molecule bond vocabulary): the F table rows of an
 * edge are packed one byte each into codes[e] (uint32, SORTED edge order; perm may be NULL = identity) once
 * per batch, and the aggregation reads 4 bytes per edge next to col[e]. */
int phc_pack_edge_codes(const int64_t *bond_i64, const int32_t *perm, const int32_t *vocab, int64_t E, int F,
                        uint32_t *codes, void *stream);
int phc_aggregate_encp_fwd(const float *x, const uint32_t *codes, const float *const *tables, const int32_t *vocab,
                           const int32_t *rowptr, const int32_t *col, int64_t n_dst, int F, int n, int k, int act,
                           int add_self, float *out, void *stream);

/* Backward of phc_aggregate_enc_fwd: dx (source-sorted walk, may be NULL) and the
 * flat table gradient (may be NULL).  The table gradient is a one-hot GEMM on the
 * matrix cores over the edges in ORIGINAL order, so it takes the raw int64 edge
 * list (edge_src_i64 / edge_dst_i64 = rows 0 / 1 of edge_index). */
size_t phc_aggregate_enc_bwd_ws_bytes(int64_t E, int F, const int32_t *vocab, int n, int k);
int phc_aggregate_enc_bwd(const float *dout, const float *x, const int64_t *bond_i64, const float *const *tables,
                          const int32_t *vocab, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E,
                          const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int64_t n_dst,
                          int F, int n, int k, int act, int add_self, float *dx, float *dtables_flat, void *ws,
                          size_t ws_bytes, void *stream);
/* dx of the sum aggregation for an IDENTITY message function, one source row per lane group (molecule batches; any size works, the
 * walk is latency-bound): dx[s] = sum over the out-edges of s of dout[dst] (+ dout[s] with add_self, s < n_dst) (+ residual[s]).
 * `residual` (may be NULL): another gradient of the same rows — the additive skip connection's share of d x (models.py:208-215) —
 * added on the way out instead of by a separate element-wise launch.  h % 4 == 0, 16-byte aligned rows. */
/* ABI 5: the model-level norm folded into the NEXT layer's aggregation.  x = act(norm(y)) (+ skip) (models.py:208-215) is evaluated
 * on the fly for every gathered row and for the row itself — statistics from the accumulator `acc` the producing GEMM filled (as
 * phc_bn_fwd_from_partials' accumulator form: save_stat [6 h], running statistics updated) — then  out[d] = sum_e (x[src_e] +
 * enc_e) (+ x[d])  as phc_aggregate_encp_fwd with an identity message function; x itself is written once to x_out [n_dst, h].
 * Same node set on both sides (n_src == n_dst); rows >= *n_valid_dev are padding (x = 0).  Bit-identical to the two launches it
 * replaces.  phc_aggregate_norm_supported: molecule-scale row counts (the one-row-per-lane-group walker), h <= 512, packed codes. */
int phc_aggregate_norm_supported(int64_t n_dst, int h, int F, int sum_vocab);
int phc_aggregate_norm_encp_fwd(const float *y, const uint64_t *acc, const float *const *weight, const float *const *bias,
                                float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked, int n_comp,
                                float momentum, float eps, int norm_act, const int32_t *n_valid_dev, const float *skip, float *save_stat,
                                float *x_out, const uint32_t *codes, const float *const *tables, const int32_t *vocab,
                                const int32_t *rowptr, const int32_t *col, int64_t n_dst, int F, int n, int k, int add_self, float *out,
                                void *stream);
int phc_aggregate_dx_rows(const float *dout, const int32_t *rowptr_t, const int32_t *col_t, int64_t n_src, int64_t n_dst, int h,
                          int add_self, const float *residual, float *dx, void *stream);
/* ABI 5: the same with up to 8 further gradients of the same rows added on the way out, in the order given (residuals: HOST array of
 * device pointers) — the shares of a skip tensor that feeds every layer (sc_type "first", models.py:230-234) meet in the first
 * layer's dx kernel instead of an n-way sum launch of their own. */
int phc_aggregate_dx_rows_multi(const float *dout, const int32_t *rowptr_t, const int32_t *col_t, int64_t n_src, int64_t n_dst, int h,
                                int add_self, const float *const *residuals, int num_residuals, float *dx, void *stream);

/* ------------------------------------------------------------------------- *
 * Batch normalisation of component-major activations, fused with the
 * activation and the skip add that follow it
 *
 * Reference: NaivePHMNorm = n_comp BatchNorm1d(h/n_comp) modules applied to a
 * reshaped copy (phc/hypercomplex/norm.py:30-35), followed in the model by
 * activation, dropout and "+ skip" (models.py:208-215) and in PHMMLP by the
 * activation (layers.py:351-353).  Per-column statistics make this one pass
 * over [N, h]:   z = (x - mean) * invstd * weight + bias ;  out = act(z) (+ residual)
 * training != 0: batch statistics (biased variance for z, unbiased for the
 *   running estimate), running = (1-momentum)*running + momentum*batch,
 *   num_batches_tracked += 1;  training == 0: running statistics.
 * weight / bias / running_mean / running_var / num_batches_tracked are HOST
 * arrays of n_comp DEVICE pointers (each [h/n_comp]; the reference's separate
 * tensors, used in place); any of the arrays may be NULL.
 * n_valid_dev: optional DEVICE int32 scalar; rows >= *n_valid_dev are padding of a
 *   fixed-shape batch: excluded from the statistics, written as 0, zero gradient.
 * save_stat [6h]: written by the forward, consumed by the backward:
 *   [mean | invstd | invstd*w | b - mean*invstd*w | w | b], w/b per column (1/0 without affine).
 * dwb_flat [2h]: dweight (component-major) followed by dbias.
 * ------------------------------------------------------------------------- */
#define PHC_MAX_COMPONENTS 16

size_t phc_bn_ws_bytes(int64_t N, int h);
int phc_bn_fwd(const float *x, const float *const *weight, const float *const *bias, float *const *running_mean,
               float *const *running_var, int64_t *const *num_batches_tracked, int n_comp, const int32_t *n_valid_dev,
               int64_t N, int h, int training, float momentum, float eps, int act, const float *residual, float *out,
               float *save_stat, void *ws, size_t ws_bytes, float drop_p, int drop_same_k, int drop_site,
               const uint64_t *rng_state, void *stream);
int phc_bn_bwd(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h,
               int training, int act, float *dx, float *dwb_flat, void *ws, size_t ws_bytes, float drop_p, int drop_same_k,
               int drop_site, const uint64_t *rng_state, void *stream);

/* Dropout behind the activation, inside the same pass (models.py:211-213: act -> phm_dropout -> + skip;
 * downstream.py:108-111).  phm_dropout (phc/hypercomplex/layers.py:31-55): an entry is dropped with probability drop_p,
 * kept entries are scaled by 1 / (1 - drop_p); drop_same_k > 0 = `same=True`: ONE mask per (row, channel) shared by the
 * h / drop_same_k hypercomplex components (drop_same_k = columns per component), 0 = independent entries.  The mask is a
 * pure function of (rng_state[0] = seed, rng_state[1] = step, drop_site, row, channel) — Philox4x32-10 — so phc_bn_bwd
 * regenerates it (same arguments) instead of reading a stored one.  rng_state is a DEVICE uint64[2]; phc_rng_advance
 * bumps the step (one tiny launch per training step, graph-capturable: every replay draws a new mask).  drop_p = 0 or
 * eval mode: off.  Needs h % 4 == 0 (and drop_same_k % 4 == 0). */
int phc_rng_advance(uint64_t *rng_state, void *stream);

/* Row-sharded batch-norm (one large graph split by destination-node range over the GPUs of a node;
 * the reference's BatchNorm1d sees ALL rows, norm.py:30-35, so the shards exchange 2h floats per norm and direction
 * to keep the single-GPU numbers).  Four phases with the host's two collectives between them:
 *   forward : phc_bn_moments (this rank's rows -> moments[2h] = sum | M2 around the rank's own mean, Chan-merged in a
 *             fixed order)  ->  all-gather of the pairs  ->  phc_bn_fwd_from_moments (every rank merges the same
 *             num_ranks pairs in rank order: identical mean / variance / running statistics on all ranks; then
 *             normalise + activation (+ residual) over the local rows).  counts_all [num_ranks] = rows per rank as
 *             DEVICE floats, N_total their sum.
 *   backward: phc_bn_bwd_sums (local dwb_local[2h] = sum g*x_hat | sum g, g = dy*act'(z); these ARE this rank's
 *             partial dweight | dbias)  ->  all-reduce(sum) of a copy  ->  phc_bn_bwd_apply (dx over the local rows
 *             with the global means).  ws: phc_bn_ws_bytes(N, h) for _moments / _sums, 2h floats for _apply. */
int phc_bn_moments(const float *x, int64_t N, int h, float *moments, void *ws, size_t ws_bytes, void *stream);
int phc_bn_fwd_from_moments(const float *x, int64_t N_local, const float *moments_all, const float *counts_all,
                            int num_ranks, int64_t N_total, const float *const *weight, const float *const *bias,
                            float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked,
                            int n_comp, int h, float momentum, float eps, int act, const float *residual, float *out,
                            float *save_stat, void *stream);
int phc_bn_bwd_sums(const float *dy, const float *x, const float *save_stat, int64_t N, int h, int act, float *dwb_local,
                    void *ws, size_t ws_bytes, void *stream);
int phc_bn_bwd_apply(const float *dy, const float *x, const float *save_stat, const float *dwb_global, int64_t N_total,
                     int64_t N_local, int h, int act, float *dx, void *ws, size_t ws_bytes, void *stream);

/* Batch-norm statistics out of the producing GEMM (training mode): phc_gemm_prebuilt_stats is
 * phc_gemm_prebuilt (no residual) whose workgroups also emit, per output column, the (sum, M2) of the
 * rows they produced below *n_valid_dev — stats_partial [phc_gemm_stats_max_partials()][2*Nout],
 * stats_counts [same] and *num_partials (host out).  phc_bn_fwd_from_partials merges them (same fixed-
 * order Chan merge as phc_bn_fwd) and applies the normalisation: the separate pass over y that computes
 * the statistics disappears.  Backward: phc_bn_bwd, unchanged.
 *
 * ACCUMULATOR FORM (ABI 3).  Every statistics hand-over of this header also exists without partial rows and without a
 * merge launch: the producing launch ADDS each workgroup's column sums into a caller-zeroed accumulator of
 * phc_stat_acc_words(h) 64-bit words (8-byte aligned) with integer atomics — two fixed-point words per value, so the
 * totals do not depend on the order of arrival and results stay bit-reproducible — and the consuming launch reads the
 * totals in its prologue.  It is selected by the arguments that have no meaning without partial rows:
 *   phc_gemm_prebuilt_stats    stats_counts = NULL (num_partials may be NULL): stats_partial is the accumulator
 *   phc_gemm_prebuilt_xf       stats_partial != NULL, stats_counts = NULL
 *   phc_bn_fwd_from_partials   stats_counts = NULL (num_partials ignored); needs h % 4 == 0 and 16-byte aligned rows
 *   phc_gemm_prebuilt_bwsums   num_partials = NULL: sums_partial is the accumulator
 *   phc_bn_bwd_from_partials   num_partials = 0: sums_partial is the accumulator (ws unused)
 *   phc_bn_bwd                 ws_bytes = 0: ws is the accumulator (one launch less: no merge between the two passes)
 * phc_bn_finalize_from_partials + phc_gemm_prebuilt_xf: phc_gemm_prebuilt_xf_acc reads the accumulator itself. */
int phc_gemm_stats_max_partials(void);
size_t phc_stat_acc_words(int h);             /* 64-bit words of one statistics accumulator of width h */
int phc_stat_acc_replicas(void);              /* rows of its [2 * replicas][2 * h] word layout (ops.py views it that way) */
/* An accumulator read back as floats (one tiny launch) — the bridge to host-side collectives of a row-sharded norm (dist.py):
 * count == 0: out[2h] = total of quantity 0 | total of quantity 1 (e.g. sum g | sum g x_hat of phc_bn_bwd_partial_acc);
 * count  > 0: the accumulator of a statistics epilogue (sum y | sum y^2) over `count` rows -> out[2h] = sum | M2 about the rows' own
 * mean, the pair phc_bn_moments produces (and phc_bn_fwd_from_moments merges across ranks). */
int phc_stat_acc_read(const uint64_t *acc, int h, int64_t count, float *out, void *stream);
int phc_gemm_prebuilt_stats(const float *x, const float *M, const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y,
                            const int32_t *n_valid_dev, float *stats_partial, float *stats_counts, int *num_partials,
                            void *stream);
/* ABI 7: the same with a residual added to the stored rows AND counted in the statistics — y = x . M + b + residual,
 * PHMConv's `transform(agg) + x` (messagepassing.py:62-65) in front of the model's norm (models.py:208-215). */
int phc_gemm_prebuilt_stats_res(const float *x, const float *M, const float *Mk4, const float *b, const float *residual, int64_t N,
                                int K, int Nout, float *y, const int32_t *n_valid_dev, float *stats_partial, float *stats_counts,
                                int *num_partials, void *stream);
int phc_bn_fwd_from_partials(const float *x, const float *stats_partial, const float *stats_counts, int num_partials,
                             const float *const *weight, const float *const *bias, float *const *running_mean,
                             float *const *running_var, int64_t *const *num_batches_tracked, int n_comp,
                             const int32_t *n_valid_dev, int64_t N, int h, float momentum, float eps, int act,
                             const float *residual, float *out, float *save_stat, float drop_p, int drop_same_k,
                             int drop_site, const uint64_t *rng_state, void *stream);

/* Bond-table gradients of SEVERAL message-passing layers in two launches.  The layers of one model share
 * the edge list and the integer bond features; only dout / x / tables differ.  Like the PHM parameter
 * gradients they are off the backward pass's critical path: a training step runs phc_aggregate_enc_bwd
 * with dtables_flat = NULL (dx only) per layer and submits the table gradients of all layers here at
 * the end.  Same one-hot MFMA GEMM and fixed-order slabs as phc_aggregate_enc_bwd.  `jobs`: HOST array. */
typedef struct phc_table_grad_job {
    const float *dout, *x;             /* device: [n_dst, n*k], [n_src, n*k] (x may be NULL when act = identity) */
    const float *const *tables;        /* device array of n*F table pointers                                     */
    float *dtables_flat;               /* device out [n][R][k]                                                   */
} phc_table_grad_job;
size_t phc_table_grads_ws_bytes(int64_t E, int F, const int32_t *vocab, int n, int k, int num_jobs);
int phc_table_grads_batched(const phc_table_grad_job *jobs, int num_jobs, const int64_t *bond_i64, const int32_t *vocab,
                            const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F, int n, int k,
                            int act, void *ws, size_t ws_bytes, void *stream);

/* The same gradients for an IDENTITY message function without touching the edges' feature rows: then
 * d enc[e] = dout[dst_e] for every in-edge e, so  dT = C^T . dout  with C[d][r] = number of in-edges of d whose
 * packed code (phc_pack_edge_codes, destination-sorted order; codes [E], rowptr [n_dst + 1]) names table row r.
 * Workgroups walk 64 destination nodes at a time, rebuild their count rows from rowptr / codes in LDS and run the
 * same MFMA tile: n_dst contiguous rows are read instead of E gathered ones.  jobs[].x and .tables are not read.  Needs F <= 4 and
 * sum(vocab) <= 256 (the packed-code limits).  Summation order differs from phc_table_grads_batched (fp32
 * rounding only). */
size_t phc_table_grads_counted_ws_bytes(int64_t n_dst, int F, const int32_t *vocab, int n, int k, int num_jobs);
int phc_table_grads_counted(const phc_table_grad_job *jobs, int num_jobs, const int32_t *rowptr, const uint32_t *codes,
                            int64_t n_dst, int64_t E, const int32_t *vocab, int F, int n, int k, void *ws,
                            size_t ws_bytes, void *stream);

/* Both sorted views of one edge list — destination-sorted (rowptr/col/perm, forward) and source-sorted
 * (rowptr_t/col_t/perm_t, backward) — side by side in FIVE launches instead of 22 (a training step needs
 * both).  Same result as two phc_csr_build calls (stable order).  At most 32767 rows per side (molecule
 * batches); larger graphs use phc_csr_build. */
size_t phc_csr_pair_ws_bytes(int64_t E, int64_t n_src, int64_t n_dst);
int phc_csr_build_pair(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src, int64_t n_dst,
                       int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t, int32_t *perm_t,
                       void *ws, size_t ws_bytes, void *stream);

/* The same, and the packed edge codes of phc_pack_edge_codes (destination-sorted order) out of the same five launches:
 * the last one knows where every edge ends up and packs its bond features there.  codes / bond_i64 / vocab may be NULL
 * (then this is phc_csr_build_pair).  F <= 4, sum(vocab) <= 256.
 * code_counts (optional, needs codes and sum(vocab) <= 32): [n_dst][CR] floats, CR = 16 when sum(vocab) <= 16 else 32 —
 * code_counts[d][t] = number of in-edges of node d whose code names table row t (columns >= sum(vocab) are 0): the bond
 * encoders' one-hot operand summed per destination, which phc_step_tail multiplies with each layer's dout on the matrix
 * cores.  Written by the same last launch. */
int phc_csr_build_pair_codes(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                             int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t,
                             int32_t *col_t, int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F,
                             uint32_t *codes, float *code_counts, void *ws, size_t ws_bytes, void *stream);
/* The same with a RESIDENT workspace: the caller zero-fills `ws_zeroed_once` ONCE (phc_csr_pair_ws_bytes bytes) and hands the same
 * buffer to every later call for this graph size; each call leaves the counter arrays zeroed again (the scan clears the counts it
 * has read, the canonicalising launch clears the fill cursors), so the per-call zero launch disappears — 4 launches instead of 5
 * in the per-batch preprocessing of a replayed training step.  A call that fails mid-way leaves the buffer undefined: zero it again. */
int phc_csr_build_pair_codes_resident(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                                      int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t,
                                      int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes,
                                      float *code_counts, void *ws_zeroed_once, size_t ws_bytes, void *stream);
/* ABI 8: ... and the first four in-edges of every destination row side by side, ell [n_dst][8] int32 (slots 0-3: source, 4-7: packed
 * code), and the first four out-edges of every source row, ell_t [n_src][4] (destination); either may be NULL; slots past a row's
 * degree name the row itself (code 0).  A kernel that owns a row reads its edges in the same round trip as its row pointers
 * (phc_mlp_norm_fwd_fused's aggregation).  Rows with more than four edges: the rest through rowptr / col / codes as before. */
int phc_csr_build_pair_codes_resident_ell(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                                          int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t,
                                          int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes,
                                          float *code_counts, int32_t *ell, int32_t *ell_t, void *ws_zeroed_once, size_t ws_bytes,
                                          void *stream);
/* ... and the packed codes in SOURCE-sorted order too (codes_src [E] or NULL): what phc_pack_edge_codes(perm_t) computes in a launch of
 * its own for phc_aggregate_softmax_enc_bwd's walk. */
int phc_csr_build_pair_codes_resident_ell_src(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                                              int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t,
                                              int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes,
                                              float *code_counts, int32_t *ell, int32_t *ell_t, uint32_t *codes_src, void *ws_zeroed_once,
                                              size_t ws_bytes, void *stream);

/* Per-batch preprocessing on the device: drop the nodes no edge touches and re-index,
 * i.e. the reference's CustomRemoveIsolatedNodes / PyG RemoveIsolatedNodes applied to every collated
 * batch (benchmarks/utils.py:36-46, train_hiv.py:171-172) — there on the CPU, before the batch is moved.
 * Kept nodes keep their order.  edge_*_out [E]: relabelled edge list; x_out [N, F] / batch_out [N] /
 * kept_index [N] (old index of every kept node): rows [0, *n_kept_dev) are written, the rest untouched
 * (fixed-shape batches keep their padding and read *n_kept_dev as the valid count).  x / batch /
 * kept_index outputs may be NULL.  F = number of int64 feature columns of x (0: no x). */
size_t phc_remove_isolated_ws_bytes(int64_t N);
int phc_remove_isolated_nodes(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t N,
                              const int64_t *x_i64, int F, const int64_t *batch_i64, int64_t *edge_src_out,
                              int64_t *edge_dst_out, int64_t *x_out, int64_t *batch_out, int64_t *kept_index,
                              int32_t *n_kept_dev, void *ws, size_t ws_bytes, void *stream);

/* Batch-norm + activation folded into the CONSUMING GEMM (linear -> norm -> act -> linear, layers.py:349-355):
 * phc_bn_finalize_from_partials merges the statistics the first GEMM emitted (save_stat, running statistics) and
 * phc_gemm_prebuilt_xf computes  y = act(x * scale + shift) . M + b  with the normalisation evaluated while each
 * 32-row slab of x is staged (rows >= *n_valid_dev read as 0): the normalised activation is never written to or
 * read from HBM.  Optionally it also emits the statistics partials of y (stats_partial != NULL), like
 * phc_gemm_prebuilt_stats.  Backward: dx through phc_gemm_prebuilt (is_dx), then phc_bn_bwd on (d act-input, x,
 * save_stat); the second layer's parameter gradients take x and the xf_* fields of phc_phm_grad_job. */
int phc_bn_finalize_from_partials(const float *stats_partial, const float *stats_counts, int num_partials,
                                  const float *const *weight, const float *const *bias, float *const *running_mean,
                                  float *const *running_var, int64_t *const *num_batches_tracked, int n_comp,
                                  const int32_t *n_valid_dev, int64_t N, int h, float momentum, float eps,
                                  float *save_stat, void *stream);
int phc_gemm_prebuilt_xf(const float *x, const float *xf_stat, int xf_act, const int32_t *n_valid_dev, const float *M,
                         const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y, float *stats_partial,
                         float *stats_counts, int *num_partials, void *stream);
/* The same with the input norm's statistics still in an ACCUMULATOR (xf_acc, width K, filled by phc_gemm_prebuilt_stats' accumulator
 * form): the launch derives scale / shift in its prologue — beside its operand panel load — writes save_stat [6 K] and updates the
 * running statistics: phc_bn_finalize_from_partials and its launch are not needed. */
int phc_gemm_prebuilt_xf_acc(const float *x, const uint64_t *xf_acc, const float *const *weight, const float *const *bias,
                             float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked,
                             int n_comp, float momentum, float eps, float *save_stat, int xf_act, const int32_t *n_valid_dev,
                             const float *M, const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y,
                             float *stats_partial, float *stats_counts, int *num_partials, void *stream);

/* Backward of that pair.  phc_gemm_prebuilt_bwsums computes dz = dy . M (M = H^T of the second layer) and, from
 * the same registers, the column sums the norm's backward needs — sum g and sum g*x_hat with
 * g = dz * act'(x_hat*w + b), x_hat = (bn_x - mean)*invstd, over the rows below *n_valid_dev — as
 * sums_partial [phc_gemm_stats_max_partials()][2*Nout] (*num_partials rows written).  phc_bn_bwd_from_partials
 * merges them (fixed order) and produces dx / dwb_flat exactly like phc_bn_bwd (training mode); ws: 2*h floats. */
int phc_gemm_prebuilt_bwsums(const float *dy, const float *M, const float *Mk4, int64_t N, int K, int Nout, float *dz, const float *bn_x,
                             const float *bn_stat, int act, const int32_t *n_valid_dev, float *sums_partial,
                             int *num_partials, void *stream);
/* The norm backward's ELEMENT-WISE pass folded into the dx GEMM behind it (ABI 4).  For  out = act(batch_norm(x)) (+ residual)
 * with the backward's column sums (sum g | sum g x_hat over the valid rows, g = dout * act'(.)) already in a statistics
 * accumulator (phc_bn_bwd_partial_acc, or the HS_BW epilogue of the GEMM in front), ONE launch computes
 *     dz = w invstd (g - mean g - x_hat mean(g x_hat))        (what phc_bn_bwd writes as dx; rows >= *n_valid_dev := 0)
 *     y  = dz . M                                             (M = H^T of the layer in front of the norm, k4 form)
 * with dz evaluated in LDS while the rows of dout and bn_x stream in, written out once (dz: the layer's parameter-gradient GEMM
 * reads it) and never re-read; dwb (dweight | dbias, [2 K]) is left behind by the first workgroup.  Optional second stage
 * (bw_sums_acc != NULL): y is itself the gradient of act(batch_norm(bw_x)) — the launch also adds THAT norm's column sums
 * into bw_sums_acc (the epilogue of phc_gemm_prebuilt_bwsums), so that the next phc_gemm_bwd_apply can follow directly.
 * One PHMGINEConv + model norm backward = phc_bn_bwd_partial_acc -> phc_gemm_bwd_apply (two stages) -> phc_gemm_bwd_apply:
 * 3 launches instead of 6.  K, Nout in {128, 256} (phc_gemm_bwd_apply_supported); otherwise PHC_ERR_UNSUPPORTED.
 * Reference ops: phc/hypercomplex/norm.py:30-35 (backward), layers.py:198-219, 349-355. */
typedef struct phc_gemm_ba_desc {
    const float *dout;             /* [N, K] upstream gradient                                        */
    const float *bn_x;             /* [N, K] the norm's input                                         */
    const float *bn_stat;          /* [6 K] its save_stat (mean | invstd | scale | shift | w | b)     */
    const uint64_t *sums_acc;      /* accumulator of width K: sum g | sum g x_hat                     */
    const int32_t *n_valid_dev;    /* device scalar or NULL                                           */
    float *dz;                     /* [N, K] out                                                      */
    float *dwb;                    /* [2 K] out or NULL                                               */
    const float *Mk4;              /* [K, Nout] operand in k4 form                                    */
    float *y;                      /* [N, Nout] out                                                   */
    int64_t N;
    int32_t K, Nout, act;
    int32_t bw_act;
    const float *bw_x;             /* second stage: [N, Nout] input of the norm in front, or NULL     */
    const float *bw_stat;          /* [6 Nout]                                                        */
    uint64_t *bw_sums_acc;         /* accumulator of width Nout (zeroed by the caller), or NULL       */
    int64_t count;                 /* rows sums_acc was taken over when that is MORE than this launch's rows: a row-sharded norm
                                      whose accumulators were all-reduced (integer sum: exact) divides by the total of all ranks;
                                      0 = the valid rows of this launch                                */
    /* ABI 6: the norm's output went through phm_dropout before it was used (models.py:211-213: act -> dropout -> + skip): dout is
     * multiplied by the forward's keep-and-scale mask inside the transform (arguments as for phc_bn_bwd; drop_p = 0: off).  Square
     * layers of width 128 / 256 / 180 / 200. */
    float drop_p;
    int32_t drop_same_k, drop_site;
    const uint64_t *rng_state;
} phc_gemm_ba_desc;
int phc_gemm_bwd_apply_supported(int K, int Nout);
/* Diagnostics only: when set (device buffer, 12 x 8 bytes per wavefront of the largest launch), every H-stationary GEMM launch
 * stamps its wavefronts' phase clocks there (scripts/dbg/hs_phases.py); NULL (the default) switches it off. */
int phc_debug_hs_stamps(unsigned long long *dev_words);
/* Run-time switch of the H-stationary kernel (ABI 5; host, process-wide): 0 = every phc_gemm_prebuilt* call takes the LDS-panel
 * kernel and phc_gemm_bwd_apply_supported answers 0 (the environment variable PHC_GEMM_NO_HS is the initial value).  The kernel's
 * correctness rests on inline-asm loads with hand-counted waits that hipcc cannot check (csrc/gemm_hs_device.h); the Python layer
 * compares it with the LDS-panel kernel once per process and device (phc_gnn_amd/selfcheck.py) and turns it off on a mismatch.
 * phc_gemm_hs_set_enabled returns the previous setting. */
int phc_gemm_hs_enabled(void);
int phc_gemm_hs_set_enabled(int on);

/* ABI 8: the backward of  act(norm_o(PHMMLP(s)))  at hidden 128 as ONE launch (csrc/mlp_fused.hip) instead of
 * phc_bn_bwd_partial_acc + phc_gemm_bwd_apply (outer, with the inner norm's sums) + phc_gemm_bwd_apply (inner) — autograd of
 * phc/hypercomplex/layers.py:349-355 behind undirectional/models.py:208-215 (norm.py:30-35).  A workgroup keeps its <= 32 rows in
 * LDS for the whole chain; the two grid-wide dependencies (a norm backward needs its column sums over all rows) are barriers
 * inside the launch, crossed only by the integer accumulators.  Outputs as the three launches': dy2 = the outer norm's dx (the dy of
 * linear2's parameter-gradient job), dz1 = the inner norm's dx (linear1's), ds = d loss / d s, dwb_* = dweight | dbias of the norms;
 * `da` never reaches memory.  Same results up to the fp32 summation order of the column sums; bit-reproducible.
 * acc_o / acc_i: phc_stat_acc_words(128) zeroed words each; barrier: phc_mlp_norm_bwd_fused_barrier_words() uint32, ZERO before the
 * first use, owned by the kernel (one buffer per stream).  phc_mlp_norm_bwd_fused_supported(N, h): h = 128 and ceil(N / 32) workgroups co-resident on this device. */
typedef struct phc_mlp_norm_bwd_desc {
    const float *dout;             /* [N, h] upstream gradient of act(norm_o(y2)) (+ skip)             */
    const float *y2, *z1;          /* [N, h] the outer / inner norm's input                            */
    const float *save_o, *save_i;  /* [6 h] their save_stat                                            */
    const float *H2Tk4, *H1Tk4;    /* k4 forms of linear2's / linear1's H^T                            */
    float *dy2, *dz1, *ds;         /* [N, h] out                                                       */
    float *dwb_o, *dwb_i;          /* [2 h] out or NULL                                                */
    uint64_t *acc_o, *acc_i;
    const int32_t *n_valid_dev;    /* device scalar or NULL                                            */
    int64_t N;
    int32_t h, act_o, act_i;
    uint32_t *barrier;
    /* agg_src != NULL: `dout` does not exist yet — it is the dx of the sum aggregation of the layer ABOVE (autograd of
     * undirectional/messagepassing.py:132-146, identity message function; what phc_aggregate_dx_rows computes in a launch of its own):
     *     dout[r] = sum over the out-edges (r -> d) of agg_src[d]  (+ agg_src[r] when agg_add_self)  (+ agg_residual[r]),
     * agg_src = that layer's ds.  The launch builds the rows itself (same edge order and additions as phc_aggregate_dx_rows: identical
     * rows for out-degrees <= 16) and writes them to dx_out once; `dout` is ignored.  agg_rowptr_t / agg_col_t: the source-sorted
     * view; agg_ell_t [N][4]: phc_csr_build_pair_codes_resident_ell.  n_src = n_dst = N. */
    const float *agg_src;          /* [N, h] or NULL                                                   */
    const float *agg_residual;     /* [N, h] or NULL                                                   */
    const int32_t *agg_rowptr_t, *agg_col_t, *agg_ell_t;
    float *dx_out;                 /* [N, h] out                                                       */
    int32_t agg_add_self;
} phc_mlp_norm_bwd_desc;
int phc_mlp_norm_bwd_fused_supported(int64_t N, int h);
int phc_mlp_norm_bwd_fused_agg_supported(int64_t N, int h);
size_t phc_mlp_norm_bwd_fused_barrier_words(void);        /* uint32 words of `barrier` */
int phc_mlp_norm_bwd_fused(const phc_mlp_norm_bwd_desc *desc, void *stream);
/* Diagnostics only: when set (32 device words), workgroups 0 and G - 1 of every launch stamp their phase clocks there; NULL: off. */
int phc_debug_mlp_bwd_stamps(unsigned long long *dev_words);
int phc_debug_mlp_all_stamps(unsigned long long *dev_words);   /* ... 8 words per workgroup: phases 0 .. 7 of every workgroup */

/* ABI 8: the forward of the same node behind the aggregation as ONE launch (csrc/mlp_fused.hip) instead of
 * phc_gemm_prebuilt_stats + phc_gemm_prebuilt_xf_acc:  z1 = s . H1 + b1;  a = act(batch_norm_train(z1));  y2 = a . H2 + b2
 * (phc/hypercomplex/layers.py:349-355, norm.py:30-35), the statistics (sum | sum of squares over the valid rows) of y2 left in acc2
 * for whoever applies the model-level norm.  z1's rows stay in registers across the launch's one grid-wide dependency, `a` never
 * reaches memory; z1 and y2 are bit-identical to the two launches' for the same operands.  The norm's tables as in phc_bn_fwd;
 * save_stat [6 h] (phc_bn_fwd's layout) and the running statistics are written by the launch.  acc1 / acc2: phc_stat_acc_words(128)
 * zeroed words each; barrier: as phc_mlp_norm_bwd_fused (the two may share one buffer per stream). */
typedef struct phc_mlp_norm_fwd_desc {
    const float *s;                /* [N, h] the layer's aggregated input                              */
    const float *H1k4, *b1, *H2k4, *b2;   /* k4 operands of linear1 / linear2; biases [h] or NULL     */
    float *z1, *y2;                /* [N, h] out                                                       */
    uint64_t *acc1, *acc2;
    const float *const *weight, *const *bias;            /* the inner norm: host arrays of n_comp device pointers (NULL: no affine) */
    float *const *running_mean, *const *running_var;
    int64_t *const *num_batches_tracked;
    int32_t n_comp;
    float momentum, eps;
    float *save_stat;              /* [6 h] out                                                        */
    int32_t act;                   /* the MLP's activation                                             */
    const int32_t *n_valid_dev;
    int64_t N;
    int32_t h;
    uint32_t *barrier;
    /* The layer's sum aggregation inside the same launch (codes != NULL; `s` is then ignored): the rows of
     *     s[d] = sum_{e: dst_e = d} (x[src_e] + sum_f table_{c,f}[code_f(e)]) (+ x[d])     messagepassing.py:132-146, encoder.py:31-34
     * are built by the workgroup that owns them (one row per 32-lane group, edge order and additions of phc_aggregate_encp_fwd: s is
     * bit-identical) and written to s_out once (the parameter-gradient GEMM reads it); fields as in phc_gemm_ag_desc.  With norm_acc
     * != NULL x is the un-applied  act(batch_norm(x)) (+ skip)  of the layer in front (models.py:208-215): scale / shift from its
     * accumulator in the prologue, every gathered row normalised on the fly, x_out written once, save_stat_o and the running
     * statistics left behind — what phc_aggregate_norm_encp_fwd does.  phc_mlp_norm_fwd_fused_agg_supported: k % 4 == 0, F <= 4,
     * sum(vocab) <= 32. */
    const float *x;
    const uint32_t *codes;
    const float *const *tables;
    const int32_t *vocab;          /* host [F] */
    const int32_t *rowptr, *col;
    const int32_t *ell;            /* [N][8] from phc_csr_build_pair_codes_resident_ell (required with codes)               */
    const float *const *tables_host;   /* HOST array [n F] of the same table pointers (n F <= 16; required with codes)      */
    int32_t F, n, add_self;
    float *s_out;
    const uint64_t *norm_acc;
    const float *const *weight_o, *const *bias_o;
    float *const *running_mean_o, *const *running_var_o;
    int64_t *const *num_batches_tracked_o;
    int32_t n_comp_o, norm_act;
    float momentum_o, eps_o;
    const float *skip;
    float *save_stat_o, *x_out;
    /* softmax_beta != NULL (with codes): the layer's SOFTMAX aggregation instead of the sum (PHMGINEConvSoftmax.aggregate,
     * messagepassing.py:211-214, 297-300: out = sum_e softmax_e(beta m_e) m_e per destination and column) — what
     * phc_aggregate_softmax_enc_fwd / phc_aggregate_softmax_norm_enc_fwd compute in a launch of their own, same online softmax in the
     * same edge order: agg_out [N, h] (the plain aggregate) and lse [N, h] (max + log(sum + 1e-12)) are written for
     * phc_aggregate_softmax_enc_bwd, s_out = agg_out (+ x when add_self) feeds the first GEMM; bit-identical for in-degrees <= 16. */
    const float *softmax_beta;     /* device scalar */
    float *agg_out, *lse;
} phc_mlp_norm_fwd_desc;
int phc_mlp_norm_fwd_fused_supported(int64_t N, int h);
int phc_mlp_norm_fwd_fused_agg_supported(int64_t N, int h, int n, int F, int vocab_rows);
int phc_mlp_norm_fwd_fused(const phc_mlp_norm_fwd_desc *desc, void *stream);

/* The layer's sum aggregation INSIDE its first GEMM (ABI 5, molecule batches).  One launch computes
 *     s[d] = sum_{e: dst_e = d} (x[src_e] + sum_f table_{c,f}[code_f(e)]) (+ x[d])     messagepassing.py:132-146 (GINE, identity
 *                                                                                      message function), encoder.py:31-34
 *     y    = s . M + b  and the column statistics of y (accumulator form)             layers.py:198-219, 349-355; norm.py:30-35
 * where x is either a tensor (norm_acc = NULL) or the layer in front's  act(batch_norm(x_in)) (+ skip)  that has not been applied
 * yet (norm_acc = the accumulator its producing GEMM filled; models.py:208-215): the launch then derives scale / shift in its
 * prologue, applies them to every row it gathers, writes x_out once and leaves save_stat / the running statistics behind — what
 * phc_aggregate_norm_encp_fwd does.  The workgroup that owns a 16-row step of the GEMM builds the step's rows itself (row-per-lane-
 * group walk, same edge order and additions as phc_aggregate_encp_fwd: s is bit-identical), writes s out once (the parameter-
 * gradient GEMM reads it) and multiplies from LDS; y and the accumulator are what phc_gemm_prebuilt_stats produces from that s.
 * Two launches (three with the stand-alone norm pass) become one.  h = Nout = 128, k % 4 == 0, F <= 4, sum(vocab) <= 256, codes as
 * written by phc_csr_build_pair_codes / phc_pack_edge_codes, 16-byte aligned rows (phc_gemm_aggregate_supported); otherwise
 * PHC_ERR_UNSUPPORTED and the caller runs the separate entry points. */
typedef struct phc_gemm_ag_desc {
    const float *x;                /* [N, h]: the gathered tensor, or (norm_acc != NULL) the pending norm's input            */
    const uint32_t *codes;         /* [E] packed bond codes in destination-sorted order                                      */
    const float *const *tables;    /* device array [n * F] of table pointers                                                 */
    const int32_t *vocab;          /* host [F]                                                                               */
    const int32_t *rowptr, *col;   /* destination-sorted view                                                                */
    int64_t N;                     /* rows (= destinations = sources)                                                        */
    int32_t F, n, k, add_self;
    float *s_out;                  /* [N, h] out: the aggregate                                                              */
    /* pending norm (all ignored when norm_acc == NULL) */
    const uint64_t *norm_acc;
    const float *const *weight, *const *bias;
    float *const *running_mean, *const *running_var;
    int64_t *const *num_batches_tracked;
    int32_t n_comp, norm_act;
    float momentum, eps;
    const float *skip;             /* [N, h] or NULL                                                                         */
    float *save_stat;              /* [6 h] out                                                                              */
    float *x_out;                  /* [N, h] out: the normalised rows                                                        */
    /* GEMM with the statistics epilogue */
    const float *Mk4;              /* [h, Nout] operand, k4 form                                                             */
    const float *b;                /* [Nout] or NULL                                                                         */
    int32_t Nout;
    float *y;                      /* [N, Nout] out                                                                          */
    const int32_t *n_valid_dev;    /* device scalar or NULL: rows below it count for both norms' statistics                  */
    uint64_t *stat_acc;            /* accumulator of width Nout (zeroed by the caller)                                       */
} phc_gemm_ag_desc;
int phc_gemm_aggregate_supported(int64_t N, int h, int Nout, int F, int vocab_rows);
int phc_gemm_aggregate_stats(const phc_gemm_ag_desc *d, void *stream);
int phc_gemm_bwd_apply(const phc_gemm_ba_desc *d, void *stream);
/* The sums pass alone: adds sum g | sum g x_hat of (dy, x, save_stat) over the rows below *n_valid_dev into `acc` (width h,
 * zeroed by the caller; h % 4 == 0, h <= 8192, 16-byte aligned rows) — phc_bn_bwd's first launch without its second. */
int phc_bn_bwd_partial_acc(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h,
                           int act, uint64_t *acc, void *stream);
/* ABI 6: the same sums with the forward's dropout mask on dy (arguments as for phc_bn_bwd; drop_p = 0: exactly the call above). */
int phc_bn_bwd_partial_acc_drop(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h,
                                int act, uint64_t *acc, float drop_p, int drop_same_k, int drop_site, const uint64_t *rng_state,
                                void *stream);
int phc_bn_bwd_from_partials(const float *dy, const float *x, const float *save_stat, const float *sums_partial,
                             int num_partials, const int32_t *n_valid_dev, int64_t N, int h, int act, float *dx,
                             float *dwb_flat, void *ws, size_t ws_bytes, void *stream);

/* The step's tail in one pass: every parameter gradient of a model — PHM / dense layers (jobs, as for
 * phc_phm_param_grads; <= 16), the bond tables of all message-passing layers
 * (tjobs, as for phc_table_grads_batched; <= 8) and the atom-encoder tables (atom_*, as phc_encode_bwd; N_atoms
 * = 0: none) — in three launches.  The first runs the table-slab workgroups and the MFMA-bound dH-slab workgroups
 * in ONE grid of 512 resident slots: the table families in front where they are a small part of the launch's work,
 * every family on its share of the slots from the start where the tables are a quarter of it or more (hidden 128;
 * round 6); the second all slab reductions, the third the PHM contraction.  Results are identical to the separate entry
 * points (same workgroup bodies, same fixed summation order).  Layers of 256 columns and more (round 4) ride in the
 * first launch as register-resident 128 x 64 tiles (128 x 32 for an odd output width) when their operand form allows
 * it (input width a multiple of 4, output width >= 32, a plain or relu-normalised operand); the rest of them take one
 * more launch of 128 x 128 LDS tiles in front.
 * Bond tables: bond_n_dst > 0 selects the counted form of phc_table_grads_counted (bond_rowptr / bond_codes, act
 * must be identity; bond_i64 / edge_* may be NULL), bond_n_dst = 0 the per-edge form of phc_table_grads_batched.
 * bond_counts (optional, counted form only): the code_counts of phc_csr_build_pair_codes — the bond tables then take
 * the register-resident form (dT = counts^T . dout straight from global memory, all layers of a group sharing the
 * counts operand); the atom tables take theirs whenever h is even and >= 32 (64-column tiles, the last one partial). */
size_t phc_step_tail_ws_bytes(const phc_phm_grad_job *jobs, int num_jobs, int num_tjobs, int64_t E, int64_t bond_n_dst,
                              int F_b, const int32_t *vocab_b, int64_t N_atoms, int F_a, const int32_t *vocab_a, int n,
                              int k);
int phc_step_tail(const phc_phm_grad_job *jobs, int num_jobs, const phc_table_grad_job *tjobs, int num_tjobs,
                  const int64_t *bond_i64, const int32_t *vocab_b, const int64_t *edge_src_i64,
                  const int64_t *edge_dst_i64, int64_t E, int F_b, int act, const int32_t *bond_rowptr,
                  const uint32_t *bond_codes, const float *bond_counts, int64_t bond_n_dst, const float *atom_dout,
                  const int64_t *atom_idx_i64, int64_t N_atoms, int F_a, const int32_t *vocab_a,
                  float *atom_dtables_flat, int n, int k, void *ws, size_t ws_bytes, void *stream);
/* ABI 5: the same call with two more families of workgroups riding in its last two launches (no launch of their own):
 *   - the gate map's parameter gradients (phc_gate_param_grads): `gate_job` names the dense job (x, dg) of `jobs` — its dW may be
 *     NULL, its db receives dbg — whose dH = x^T . dg is read where the reduce launch leaves it;
 *   - the explicit weight regulariser (phc_phm_weight_reg; regularization.py:15-23, train_hiv.py:181-196): reg_W[q] [reg_n[q],
 *     reg_cells[q]] are the regularised weights, reg_job[q] the job of `jobs` that writes that layer's dW (same W pointer; -2 = the
 *     gate map).  The gradient is ADDED to the freshly written dW, the value written to reg_value and added to *reg_loss_inout
 *     (either may be NULL).  reg_ws: phc_tail_extras_ws_floats(extras) floats, 16-byte aligned, contents irrelevant.
 * extras == NULL: exactly phc_step_tail.  gate_job < 0 / reg_num == 0 switch the families off individually. */
typedef struct phc_tail_extras {
    int32_t gate_job;
    int32_t gate_n, gate_kin, gate_kout, gate_kg;
    const float *gate_H, *gate_w, *gate_phm_b, *gate_A, *gate_W;
    float *gate_dA, *gate_dW, *gate_db, *gate_dWg;
    int32_t reg_num, reg_p;
    float reg_coef;
    const float *const *reg_W;                /* host arrays of reg_num entries */
    const int32_t *reg_n, *reg_cells, *reg_job;
    float *reg_value, *reg_loss_inout, *reg_ws;
} phc_tail_extras;
int phc_tail_reg_max_jobs(void);
size_t phc_tail_extras_ws_floats(const phc_tail_extras *extras);
int phc_step_tail_ex(const phc_phm_grad_job *jobs, int num_jobs, const phc_table_grad_job *tjobs, int num_tjobs, const int64_t *bond_i64,
                     const int32_t *vocab_b, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F_b, int act,
                     const int32_t *bond_rowptr, const uint32_t *bond_codes, const float *bond_counts, int64_t bond_n_dst,
                     const float *atom_dout, const int64_t *atom_idx_i64, int64_t N_atoms, int F_a, const int32_t *vocab_a,
                     float *atom_dtables_flat, int n, int k, void *ws, size_t ws_bytes, const phc_tail_extras *extras, void *stream);

/* ------------------------------------------------------------------------- *
 * Read-out
 * ------------------------------------------------------------------------- */

/* NaN-masked BCE-with-logits, mean over the labeled entries:
 *   mask = ~isnan(y); loss = BCEWithLogits(logits[mask], y[mask])   (benchmarks/train_hiv.py:174-178)
 * without the host-synchronising boolean gather.  logits, y: [total] device fp32 (y = NaN where
 * unlabeled); loss: device scalar; stat[2] = (sum of entry losses, labeled count), consumed by the
 * backward: dlogits = *dloss * (sigmoid(logits) - y) / count on labeled entries, 0 elsewhere.
 * Fixed-order reductions. */
size_t phc_bce_masked_ws_bytes(int64_t total);
int phc_bce_masked_fwd(const float *logits, const float *y, int64_t total, float *loss, float *stat, void *ws,
                       size_t ws_bytes, void *stream);
/* Single-workgroup form (total <= 4096, e.g. one task per graph): also writes dlogits_unit = d loss / d logits for an
 * upstream gradient of exactly 1, so the backward of a scalar training loss is a tensor that already exists. */
int phc_bce_masked_fwd_unit(const float *logits, const float *y, int64_t total, float *loss, float *stat,
                            float *dlogits_unit, void *stream);
int phc_bce_masked_bwd(const float *logits, const float *y, const float *dloss, const float *stat, int64_t total,
                       float *dlogits, void *stream);
/* The same three for the mean L1 loss over the labeled entries (benchmarks/train_zinc.py: torch.nn.L1Loss on the regression target;
 * d|d|/dd = sign(d) with sign(0) = 0 as in torch).  A NaN target (a padding graph of a fixed-shape batch) stays out of the mean. */
int phc_l1_masked_fwd(const float *pred, const float *y, int64_t total, float *loss, float *stat, void *ws, size_t ws_bytes,
                      void *stream);
int phc_l1_masked_fwd_unit(const float *pred, const float *y, int64_t total, float *loss, float *stat, float *dpred_unit,
                           void *stream);
int phc_l1_masked_bwd(const float *pred, const float *y, const float *dloss, const float *stat, int64_t total, float *dpred,
                      void *stream);

/* ------------------------------------------------------------------------- *
 * The downstream head in one launch each way
 *
 * PHMDownstreamNet.forward with two hidden layers (phc/hypercomplex/downstream.py:90-120) + RealTransformer 'linear'
 * (layers.py:399-414), training mode, dropout 0:
 *   z1 = x.H1 + b1;  a1 = act(norm1(z1));  z2 = a1.H2 + b2;  a2 = act(norm2(z2));  z3 = a2.H3 + b3;  logits = z3.Wr^T + br
 * on B rows (one per graph).  A workgroup keeps 32 rows on chip for the whole chain; the batch statistics of the two norms
 * (over ALL rows, as BatchNorm1d) are exchanged between the workgroups inside the launch (csrc/downstream.hip).
 * H* = prebuilt operand matrices [K][N], H*T = their transposes [N][K] (phc_phm_build_H_batched); Wr [T][d3] is the dense
 * nn.Linear weight, used in place.  Widths d0, d1, d2 in {32, 64, 128}, d3 <= 32, T <= 8, B <= 2048
 * (phc_downstream_supported; otherwise PHC_ERR_UNSUPPORTED and the caller stays on the per-layer entry points).
 * norm*: per-component pointer tables as in phc_bn_fwd (running statistics required; they are updated by the forward).
 * The forward writes z1 [B,d1], z2 [B,d2], z3 [B,d3], stat1 [6*d1], stat2 [6*d2] (phc_bn_fwd's save_stat layout) and
 * logits [B,T]; the backward reads them plus dlogits and writes dx [B,d0] (may be NULL), dz1, dz2, dz3 (= the dy of the
 * three PHM layers: their parameter gradients go through phc_phm_param_grads / phc_step_tail with x, (z1, stat1), (z2, stat2)
 * as inputs), dwb1 [2*d1], dwb2 [2*d2] (dweight | dbias of the norms) and dWr [T*d3 + T] (d Wr, then d br).
 * partial: phc_downstream_partial_floats(B) floats and sync: 4 uint32, both ZERO before the first use and never touched
 * by the caller afterwards (one pair of buffers per direction, per stream and per B: sync[0] is a launch
 * generation, partial holds 8-byte {value, generation tag} granules the workgroups exchange inside the launch).
 * n_valid_dev: optional device int32 — rows >= *n_valid_dev are padding graphs of a fixed-shape batch.
 * ------------------------------------------------------------------------- */
typedef struct phc_downstream_norm {
    const float *const *weight, *const *bias;            /* host arrays of n_comp device pointers (NULL: no affine) */
    float *const *running_mean, *const *running_var;
    int64_t *const *num_batches_tracked;                  /* may be NULL */
    int n_comp;
    float momentum, eps;
} phc_downstream_norm;
typedef struct phc_downstream_desc {
    int64_t B;
    int d0, d1, d2, d3, T, act;
    const int32_t *n_valid_dev;
    const float *x, *H1, *b1, *H2, *b2, *H3, *b3, *H1T, *H2T, *H3T, *Wr, *br;
    phc_downstream_norm norm1, norm2;
    float *z1, *z2, *z3, *logits, *stat1, *stat2;
    const float *dlogits;
    float *dx, *dz1, *dz2, *dz3, *dwb1, *dwb2, *dWr;
    float *partial;
    uint32_t *sync;
    /* ABI 5 (may all be NULL): the k4 forms of H1, H2 and of their transposes (phc_phm_H_job.Hk4 / HTk4).  With them the call runs
     * on the 16-row kernels (csrc/downstream.hip, v2): 16 rows per workgroup (32 above 1280 rows), the three wide GEMMs on
     * v_mfma_f32_16x16x4_f32 with the operand loaded straight into registers — no LDS staging pass in front of the first MFMA —
     * and d0, d1, d2 any multiple of 4 in [16, 256] (the 256-wide heads of configs[2], the reference's 180 -> 180 -> 80 head),
     * B <= 2048 (phc_downstream_supported_k4).  Same results up to the fp32 summation order; same partial / sync protocol. */
    const float *H1k4, *H2k4, *H1Tk4, *H2Tk4;
    /* ABI 7 (16-row kernels only; all zero / NULL: no dropout): phm_dropout behind the two activations (downstream.py:104-107,
     * layers.py:31-55), the masks of phc_bn_fwd's fused dropout — a pure function of (seed, step, site, row, column group), so the
     * backward launch re-derives them.  drop_same_k*: 0 = one mask value per element, else the component width (the mask is shared
     * by the components); rng_state: device {seed, step}.  a1 [B, d1], a2 [B, d2]: the dropped activations, written by the forward
     * when not NULL — with dropout the parameter-gradient jobs of layers 2 and 3 take them as x (the operand transform from
     * (z, stat) knows no mask). */
    float drop_p1, drop_p2;
    int32_t drop_same_k1, drop_same_k2, drop_site1, drop_site2;
    const uint64_t *rng_state;
    float *a1, *a2;
    /* ABI 8 (16-row kernels, forward only; loss_y NULL: off): the training loss of the logits inside the forward launch —
     * mean over the entries whose label is not NaN of BCE-with-logits (loss_kind 0: benchmarks/train_hiv.py:174-178,
     * mask = ~isnan(y); criterion(logits[mask], y[mask])) or of |logit - y| (loss_kind 1: train_zinc.py's L1Loss) — as
     * phc_bce_masked_fwd_unit / phc_l1_masked_fwd_unit compute it in a launch of their own: loss_out [3] = loss | sum | labeled
     * count, loss_dlogits [B, T] = d loss / d logits for an upstream gradient of 1.  loss_y [B, T]; B T <= 4096
     * (phc_downstream_loss_supported); loss_granules: B T 8-byte words, ZERO before the first use, owned like `partial`. */
    const float *loss_y;
    int32_t loss_kind;
    float *loss_out, *loss_dlogits;
    uint64_t *loss_granules;
    /* ABI 8 (with loss_y; phc_downstream_fused_bwd_supported): fused_bwd != 0 makes phc_downstream_fwd run the BACKWARD too, in the same
     * launch, for an upstream gradient d loss = 1 (autograd of downstream.py:90-120 + layers.py:399-414 behind the masked mean loss:
     * what phc_downstream_bwd computes from loss_dlogits): the backward's fields above (H1T, H2T, H3T, H1Tk4, H2Tk4, dx, dz1, dz2, dz3,
     * dwb1, dwb2, dWr) are read / written as phc_downstream_bwd does, `dlogits` is not read.  partial_bwd / sync_bwd: the buffers the
     * caller would hand to phc_downstream_bwd as partial / sync (distinct from the forward's).  Bit-identical to the two launches.
     * A caller whose upstream gradient turns out not to be 1 calls phc_downstream_bwd afterwards: it overwrites the outputs. */
    int32_t fused_bwd;
    float *partial_bwd;
    uint32_t *sync_bwd;
} phc_downstream_desc;
int phc_downstream_supported(int64_t B, int d0, int d1, int d2, int d3, int T);
int phc_downstream_supported_k4(int64_t B, int d0, int d1, int d2, int d3, int T);
int phc_downstream_dropout_supported(int64_t B, int d0, int d1, int d2, int d3, int T);   /* ABI 7: the drop_* fields are served */
int phc_downstream_loss_supported(int64_t B, int d0, int d1, int d2, int d3, int T);      /* ABI 8: the loss_* fields are served */
int phc_downstream_fused_bwd_supported(int64_t B, int d0, int d1, int d2, int d3, int T); /* ABI 8: fused_bwd is served */
size_t phc_downstream_partial_floats(int64_t B);
int phc_downstream_fwd(const phc_downstream_desc *desc, void *stream);
int phc_downstream_bwd(const phc_downstream_desc *desc, void *stream);

/* ABI 7: the head at widths beyond phc_downstream_* (d1 = 768, d3 = n * 128 targets, T = 128: BASELINE configs[3]'s molpcba head
 * 256 -> 768 -> 256 -> 512 -> 128, downstream.py:90-120 + layers.py:399-414) as ONE launch per direction (csrc/head_chain.hip).
 * A launch is a CHAIN of up to PHC_HEAD_MAX_STAGES stages over the B rows:
 *     y_s = in_s . M_s^T + bias_s            M_s [N_s][K_s] row-major (out column major: M_s[c][k] multiplies in_s[:, k])
 *     post 0:  out_s = y_s ;                                                   in_{s+1} = y_s
 *     post 1:  out_s = y_s (the pre-norm z), stat = save_stat (phc_bn_fwd's layout), running statistics updated ;
 *              in_{s+1} = act(batch_norm_train(y_s)) over all rows < *n_valid_dev  (norm.py:30-35; stages 0 and 1 only)
 *     post 2:  g = y_s * act'(x_hat w + b), x_hat from (z, stat) of the forward ; out_s = in_{s+1} = batch_norm_backward(g) ;
 *              dwb = dweight | dbias of the norm  (what autograd derives from norm.py:30-35)
 * Forward chain of the head:  M = (H1^T, H2^T, H3^T, Wr), post = (1, 1, 0, 0), in = x, out = (z1, z2, z3, logits).
 * Backward chain:             M = (Wr^T, H3, H2, H1),     post = (0, 2, 2, 0), in = dlogits, out = (dz3, dz2, dz1, dx);
 * the parameter gradients of the four layers are jobs of phc_step_tail / phc_phm_param_grads with (x, dz1), ((z1, stat1), dz2),
 * ((z2, stat2), dz3), (z3, dlogits) as before.  Every M is a matrix phc_phm_build_H_batched already holds (H, H^T, the dense weight
 * and its transpose), read in place.
 * K_s, N_s multiples of 16, K_s <= 1024, K_{s+1} = N_s; the grid is (B / 16 row blocks) x (CB column blocks) and all of it must be
 * resident at once (phc_head_chain_supported asks the device; 1024 valid rows at these widths on a 256-CU device).
 * work: phc_head_chain_work_floats floats, sync: 4 uint32 — both ZERO before the first use and never touched by the caller afterwards,
 * one pair per direction, per stream and per shape (same protocol as phc_downstream_*: generation-tagged 8-byte granules).
 * Same function as the per-layer entry points up to the fp32 summation order; bit-reproducible from launch to launch. */
#define PHC_HEAD_MAX_STAGES 4
typedef struct phc_head_stage {
    const float *M;                /* [N][K] */
    const float *bias;             /* [N] or NULL */
    int32_t K, N, post;
    float *out;                    /* [B][N] */
    const float *z;                /* post 2: the forward's pre-norm output [B][N] */
    float *stat;                   /* post 1: save_stat out [6 N]; post 2: save_stat in */
    float *dwb;                    /* post 2: [2 N] out, may be NULL */
    phc_downstream_norm norm;      /* post 1 */
    /* phm_dropout behind the activation of a post 1 stage and, with the same values, in the post 2 stage of the backward chain that
     * undoes it (drop_p = 0: none; drop_same_k: 0 or the component width of a mask shared by the components; the masks of
     * phc_bn_fwd's fused dropout — phc_head_chain_desc.rng_state).  a_out (post 1, may be NULL): the (dropped) activation [B][N],
     * the x of the next layer's parameter-gradient job when a mask is on. */
    float drop_p;
    int32_t drop_same_k, drop_site;
    float *a_out;
} phc_head_stage;
typedef struct phc_head_chain_desc {
    int64_t B;
    int32_t n_stages, act;
    const int32_t *n_valid_dev;    /* device scalar or NULL: rows at and beyond it are padding graphs */
    int64_t valid_rows_max;        /* an upper bound of *n_valid_dev the caller guarantees (0: B).  Row blocks at and beyond
                                    * *n_valid_dev exit at once (their output rows are zero) and need not be resident with the others:
                                    * a fixed-shape batch of 1024 graphs + 32 padding graphs runs on a 256-CU device. */
    const float *in;               /* [B][stage[0].K] */
    phc_head_stage stage[PHC_HEAD_MAX_STAGES];
    float *work;
    uint32_t *sync;
    const uint64_t *rng_state;     /* device {seed, step}; needed when a stage has drop_p > 0 */
} phc_head_chain_desc;
int phc_head_chain_supported(int64_t B, int64_t valid_rows_max, int n_stages, const int32_t *K, const int32_t *N);
size_t phc_head_chain_work_floats(int64_t B, int n_stages, const int32_t *N);
int phc_head_chain(const phc_head_chain_desc *desc, void *stream);

/* Gated per-graph sum of PHMSoftAttentionPooling (phc/hypercomplex/pooling.py:57-66):
 *   out[g, c*k + j] = sum_{r : batch[r] = g} x[r, c*k + j] * sigmoid(gate[r, j]),   k = h / n_comp
 * x [N, h], gate [N, k] (the pre-sigmoid gate), rows grouped by graph: seg_ptr [num_graphs + 1]
 * (phc_segment_ptr_sorted or phc_csr_build on `batch`).  Replaces sigmoid + broadcast multiply +
 * global_add_pool, and in the backward the row gather + two multiplies + component sum +
 * sigmoid_backward (row-parallel: the graph of a row comes from the int64 `batch` vector).
 * dx / dgate may be NULL. */
int phc_gated_pool_fwd(const float *x, const float *gate, const int32_t *seg_ptr, int64_t num_graphs, int h,
                       int n_comp, float *out, void *stream);
/* The same straight from the SORTED int64 `batch` vector of a collated mini-batch (graph id per row, non-decreasing):
 * every workgroup finds its graph's rows with two cooperative searches, so no segment-pointer launch is needed. */
int phc_gated_pool_fwd_sorted(const float *x, const float *gate, const int64_t *batch_sorted_i64, int64_t N,
                              int64_t num_graphs, int h, int n_comp, float *out, void *stream);
/* The soft-attention read-out's forward as ONE launch (ABI 5; h = 128 with k = kg = 32, h = 256 with k = kg = 32 or 64: phc_readout_fused_supported):
 *     x    = act(batch_norm(x_in)) (+ skip)   the last layer's norm, un-applied (norm_acc != NULL; as phc_aggregate_norm_encp_fwd), or x
 *     gate = x . M + c                        M [h, kg] row-major: the composed gate map (phc_phm_H_job.flags == 3), pooling.py:57-66
 *     out[g, c k + j] = sum_{r: batch[r] = g} x[r, c k + j] sigmoid(gate[r, j])              (phc_gated_pool_fwd_sorted)
 * A workgroup owns a graph (rows found in the sorted batch vector), builds 16-row tiles of x in LDS, multiplies them with the gate
 * operand on the matrix cores and adds the gated rows up; x_out (norm_acc != NULL) and gate are written once for the backward
 * kernels, which are unchanged.  Same function as phc_bn_fwd_from_partials + phc_gemm_prebuilt + phc_gated_pool_fwd_sorted up to
 * the fp32 summation order of the gate's contraction and of the per-graph sums (x itself is bit-identical). */
typedef struct phc_readout_desc {
    const float *x;                /* [N, h] rows, or the pending norm's input                                  */
    const int64_t *batch_sorted_i64;   /* [N] non-decreasing graph ids                                          */
    int64_t N, num_graphs;
    int32_t h, n_comp, kg;
    const float *M, *c;            /* gate operand [h, kg] and bias [kg] (or NULL)                              */
    float *gate;                   /* [N, kg] out (pre-sigmoid)                                                 */
    float *out;                    /* [num_graphs, h] out                                                       */
    const int32_t *n_valid_dev;    /* device scalar or NULL: rows at and beyond it are padding (x := 0)         */
    /* pending norm (ignored when norm_acc == NULL) */
    const uint64_t *norm_acc;
    const float *const *weight, *const *bias;
    float *const *running_mean, *const *running_var;
    int64_t *const *num_batches_tracked;
    int32_t norm_n_comp, norm_act;
    float momentum, eps;
    const float *skip;
    float *save_stat, *x_out;
} phc_readout_desc;
int phc_readout_fused_supported(int64_t N, int64_t num_graphs, int h, int n_comp, int kg);
int phc_readout_fused_fwd(const phc_readout_desc *d, void *stream);
/* Diagnostics only: 8 x 8 bytes per graph; every phc_readout_fused_fwd launch then stamps its workgroups' phase clocks there. */
int phc_debug_readout_stamps(unsigned long long *dev_words);
int phc_gated_pool_bwd(const float *dout, const float *x, const float *gate, const int64_t *batch, int64_t N,
                       int64_t num_graphs, int h, int n_comp, float *dx, float *dgate, void *stream);
/* ABI 7: the same with the gate layer's backward-data map folded in — gate = x . M + c (ops.gate_linear's composed operand,
 * pooling.py:57-66), MT = M^T [k, h] row-major:  dx[r, :] = dout[g(r), :] * sigmoid(gate[r, :]) (per component) + dgate[r, :] . MT.
 * One launch instead of phc_gated_pool_bwd + a K = k GEMM with the pooled share as residual.  k = h / n_comp = 32 (64 with
 * PHC_POOL_BWD_GATE=2: slower than the two launches there), n_comp <= 8 (phc_gated_pool_bwd_gate_supported); dx, dgate both required. */
int phc_gated_pool_bwd_gate_supported(int h, int n_comp);
int phc_gated_pool_bwd_gate(const float *dout, const float *x, const float *gate, const int64_t *batch, int64_t N, int64_t num_graphs,
                            int h, int n_comp, const float *MT, float *dx, float *dgate, void *stream);

/* Many small strided 2-D copies in one launch: dst[r * dst_stride + c] = src[r * src_stride + c], r < rows, c < cols, with
 * src = (job.src ? job.src : src_base) + job.src_off and dst likewise (element offsets).  `jobs_dev` is a DEVICE array, built once
 * (the pointers of a model's tables do not move; per-call buffers come in through the two base pointers); max_elems = the largest
 * rows * cols of any job.
 * Use: an encoder whose component size k is not a multiple of four (the reference's published 180- / 200-wide models, k = 45,
 * 50) keeps a flat [sum_f V_f, n*k] image of its n*F tables [V_f, k] — image[off_f + r, c*k + j] = table(c, f)[r, j] — that the
 * vector kernels read as the tables of a ONE-component encoder of width n*k (16-byte aligned rows whatever k is); a second job
 * table turns the image's gradient into the per-table gradients (encoder.py:30-75). */
typedef struct phc_copy2d_job {
    const float *src;              /* NULL: src_base */
    float *dst;                    /* NULL: dst_base */
    int64_t src_off, dst_off;      /* elements */
    int64_t src_stride, dst_stride;
    int32_t rows, cols;
} phc_copy2d_job;
int phc_copy2d_batched(const phc_copy2d_job *jobs_dev, int num_jobs, int64_t max_elems, const float *src_base, float *dst_base,
                       void *stream);

/* out = srcs[0] + srcs[1] + ... + srcs[n-1] (in that order), n <= 16, srcs a HOST array of device pointers:
 * the gradient of a tensor that fans out to several consumers — the encoded atoms feed the first conv and
 * every layer's skip connection (models.py:230-234) — in one pass instead of n-1 pairwise adds. */
int phc_sum_n(const float *const *srcs, int n, int64_t numel, float *out, void *stream);

/* *outs[j] = sum of srcs[j][0 .. lens[j])  for up to 16 vectors in one launch (one workgroup each, fixed order): the
 * d/d beta partial sums of every softmax-aggregation layer of a step (phc_aggregate_softmax_bwd writes
 * phc_softmax_bwd_num_partials values per layer).  srcs / lens / outs are HOST arrays of device pointers / lengths. */
int phc_vec_sums(const float *const *srcs, const int32_t *lens, float *const *outs, int num, void *stream);

/* The last dense map of the read-out (RealTransformer 'linear', layers.py:399-414: n*target -> target columns,
 * 4 -> 1 for ogbg-molhiv) on B rows: y = x.w^T + b with in*out <= phc_small_linear_max_weights(), one launch
 * forward and one backward (dx row-parallel; dw, db by one workgroup in a fixed order) instead of a BLAS
 * call, two backward GEMMs, a bias reduction and a fill.  w [out, in] as nn.Linear stores it. */
int phc_small_linear_max_weights(void);
int phc_small_linear_fwd(const float *x, const float *w, const float *b, int64_t N, int fi, int fo, float *y, void *stream);
int phc_small_linear_bwd(const float *dy, const float *x, const float *w, int64_t N, int fi, int fo, float *dx, float *dw,
                         float *db, void *stream);

/* ------------------------------------------------------------------------- *
 * Optimizer step of the training loop
 * ------------------------------------------------------------------------- */

/* clip_grad_norm_(parameters, max_norm, norm_type=2) + Adam(weight_decay=0).step() for ALL parameter
 * tensors of a model in two launches (benchmarks/train_hiv.py:198-201, :285; PyTorch needs ~21
 * multi-tensor launches for the ~230 small tensors of a PHC model).
 *   table_dev  [num_tensors][4] u64 on the DEVICE: {param ptr, exp_avg ptr, exp_avg_sq ptr, numel}
 *   grads      [num_tensors]   HOST array of device pointers, NULL = no gradient this step (tensor skipped);
 *              consumed at call time (they travel by value: at most phc_clip_adam_max_tensors() per call)
 *   chunks_dev [num_chunks][2] int32 on the DEVICE: {tensor, first element}, phc_clip_adam_chunk() elements each
 *   step_dev   [num_tensors] device floats: update counts per tensor (torch.optim.Adam counts per parameter),
 *              incremented for tensors that have a gradient; scal_dev[2] out: {total norm, clip coefficient}
 *   partial_dev [total_chunks] scratch; max_norm <= 0: no clipping; write_back_clipped: also scale the
 *   gradients in place, as clip_grad_norm_ does.
 *   phases: 1 = sums of squares of this group's chunks into partial_dev[chunk_base ..) (+ the step counters),
 *   4 = Adam update of this group — every workgroup derives the total norm / clip coefficient from all
 *   total_chunks partials in the same fixed order, the group with chunk_base = 0 publishes them in scal_dev;
 *   2 (without 4) = only publish norm and coefficient.  5 = a model that fits one group; larger models:
 *   phase 1 per group, then phase 4 per group.
 * Arithmetic as torch's fused Adam (fp32): m += (g-m)(1-b1); v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  Fixed-order norm reduction. */
int phc_clip_adam_max_tensors(void);
int phc_clip_adam_chunk(void);
int phc_clip_adam_step(const uint64_t *table_dev, const float *const *grads, int num_tensors, const int32_t *chunks_dev,
                       int num_chunks, int chunk_base, int total_chunks, int phases, float max_norm, float lr, float beta1,
                       float beta2, float eps, float *step_dev, float *scal_dev, float *partial_dev,
                       int write_back_clipped, void *stream);

/* The same pair as ONE launch (ABI 8): the sums of squares travel between the workgroups inside the launch (tagged write-through
 * granules, as the fused prediction head's batch statistics do), so the launch needs every chunk's workgroup co-resident:
 * phc_clip_adam_fused_supported() answers for the current device (1 = the model's chunk count fits; a CU mask in the environment
 * disqualifies), PHC_ERR_UNSUPPORTED otherwise — callers then use phc_clip_adam_step.  `ws`: phc_clip_adam_fused_ws_bytes(total_chunks)
 * bytes, 8-byte aligned, ZERO before the first use and owned by this optimizer (it holds the launch generation).  Results are
 * bit-identical to phc_clip_adam_step(phases = 5): same per-chunk reduction tree, same order of the chunk sums.  A rendezvous that
 * never completes raises PHC_ERR_RENDEZVOUS in the device error word and leaves NaN parameters, never a hang.
 * Replaces in the step: clip_grad_norm_ + Adam.step() (benchmarks/train_hiv.py:198-201). */
size_t phc_clip_adam_fused_ws_bytes(int total_chunks);
int phc_clip_adam_fused_supported(int num_tensors, int total_chunks);
int phc_clip_adam_fused(const uint64_t *table_dev, const float *const *grads, int num_tensors, const int32_t *chunks_dev, int total_chunks,
                        float max_norm, float lr, float beta1, float beta2, float eps, float *step_dev, float *scal_dev, void *ws,
                        size_t ws_bytes, int write_back_clipped, void *stream);

/* Data-parallel gradient bucket (ABI 4): every gradient tensor of the model -> ONE flat fp32 buffer in ONE launch, scaled by
 * `scale` (1 / world: the mean of the replicas' gradients folded in in front of the SUM all-reduce).  Tables as phc_clip_adam_step
 * (the flat offset of tensor t is the offset of its first-moment view from `moments_base`); a NULL gradient packs zeros.  The
 * optimizer then reads the reduced gradients in place (views of the bucket): no unpack pass. */
int phc_pack_grads(const uint64_t *table_dev, const float *const *grads, int num_tensors, const int32_t *chunks_dev, int num_chunks,
                   const float *moments_base, float *out_flat, float scale, void *stream);

/* The explicit weight regulariser of the training scripts (phc/hypercomplex/regularization.py:15-23, added to the loss at
 * benchmarks/train_hiv.py:181-196 with coef = lr * weight_decay; --weightdecay defaults to 0.001, --regularization to 2):
 *   reg = coef * sum_layers mean_{kappa,pi} || W_layer[:, kappa, pi] ||_p ,   p in {1, 2}
 * for up to phc_phm_weight_reg_max_jobs() PHM weights in ONE launch: W[j] is [n[j]][cells[j]] (cells = kin*kout);
 * d reg / d W is ADDED into dW[j] (may be NULL: value only), the value is written to *reg_out and, if loss_inout is given,
 * added to *loss_inout (the reference's `loss += reg_loss`).  W / dW / n / cells are HOST arrays.  n[j] <= 16, and
 * sum_j ceil(cells[j] / 256) <= phc_phm_weight_reg_max_blocks() (one workgroup per 256 cells; PHC_ERR_UNSUPPORTED beyond).
 * ws: phc_phm_weight_reg_ws_bytes() bytes, ZERO before the first use and owned by the kernel from then on (the
 * workgroups' shares of the value meet there as tagged granules; one buffer per stream). */
int phc_phm_weight_reg_max_jobs(void);
int phc_phm_weight_reg_max_blocks(void);
size_t phc_phm_weight_reg_ws_bytes(void);
int phc_phm_weight_reg(const float *const *W, float *const *dW, const int32_t *n, const int32_t *cells, int num_jobs, float coef,
                       int p, float *reg_out, float *loss_inout, void *ws, size_t ws_bytes, void *stream);


/* ------------------------------------------------------------------------- *
 * Opt-in launch profiler (process-global).  When enabled every kernel launch
 * made through this library is bracketed by two hipEvents on ITS stream;
 * phc_prof_read synchronises those events and returns per-kernel totals.
 * bench.py uses it for the "roofline" object; leave it off otherwise.
 * ------------------------------------------------------------------------- */
int phc_prof_enable(int on);
int phc_prof_reset(void);
int phc_prof_num_kernels(void);
int phc_prof_read(int kernel_id, const char **name, uint64_t *launches, double *total_ms);   /* host out-params */

#ifdef __cplusplus
}
#endif
#endif /* PHC_HIP_H */
