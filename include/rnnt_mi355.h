/*
 * rnnt_mi355.h — C ABI of librnnt_mi355.so, the MI355X (gfx950) RNN-Transducer loss.
 *
 * This is the drop-in boundary for the ONE native call the reference makes:
 *     /root/reference/src/rnnt/model.py:5    from warprnnt_pytorch import RNNTLoss
 *     /root/reference/src/rnnt/model.py:87   self.transducer_loss = RNNTLoss()
 *     /root/reference/src/rnnt/model.py:116  self.transducer_loss(logits, tokens.int(), out_len, tok_len.int())
 * warprnnt_pytorch (HawkAaron/warp-transducer, un-vendored, un-pinned:
 * /root/reference/README.md:6,47) binds a native `get_workspace_size` +
 * `compute_rnnt_loss` pair; the entry points below keep those two names and
 * their argument meaning, and add the forward/backward split this library
 * uses so that fwd+bwd costs exactly three passes over the logits.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++/torch types, no exceptions.
 *   - every pointer is a DEVICE pointer (hipMalloc'ed or torch-allocated);
 *     the caller owns every buffer, including the workspace.  The library
 *     allocates nothing, never synchronises the host, and keeps no state between
 *     calls (one exception, stated there: compute_rnnt_loss_static's device array):
 *     all work is enqueued on `options.stream`.
 *   - acts   [B, maxT, maxU, V] contiguous raw (un-normalised) joint logits,
 *            maxU = max label length + 1   (model.py:115 produces this layout)
 *     labels [B, maxU-1] int32, zero padded, WITHOUT the BOS (model.py:116)
 *     act_lens, label_lens [B] int32
 *   - maxU <= RNNT_MAX_LATTICE_COLUMNS (4095 labels per utterance); a wider lattice is INVALID_VALUE.  Up to 512
 *     columns the lattice pass keeps one column per lane; beyond that a slower kernel takes over (~1-2 us per
 *     anti-diagonal).  Upstream has no such limit; the reference's data never exceeds 150 labels.
 *   - degenerate numerics give NaN/Inf in the costs, not an error status
 *     (/root/reference/bin/train.py:36-41 consumes NaN/Inf losses).
 */
#ifndef RNNT_MI355_H_
#define RNNT_MI355_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNNT_MAX_LATTICE_COLUMNS 4096

typedef enum {
    RNNT_STATUS_SUCCESS = 0,
    RNNT_STATUS_MEMOPS_FAILED = 1,
    RNNT_STATUS_INVALID_VALUE = 2,
    RNNT_STATUS_EXECUTION_FAILED = 3,
    RNNT_STATUS_UNKNOWN_ERROR = 4,
    RNNT_STATUS_STATIC_WORKSPACE_EXCEEDED = 5 /* compute_rnnt_loss_static only */
} rnntStatus_t;

/* element type of acts and grads; the lattice (lse, alpha, beta, costs) is always fp32 */
typedef enum { RNNT_DTYPE_F32 = 0, RNNT_DTYPE_BF16 = 1, RNNT_DTYPE_F16 = 2 } rnntDtype_t;

typedef enum { RNNT_REDUCTION_NONE = 0, RNNT_REDUCTION_MEAN = 1, RNNT_REDUCTION_SUM = 2 } rnntReduction_t;

typedef struct {
    void *stream;     /* hipStream_t to enqueue on (NULL = the default stream)                     */
    int blank_label;  /* upstream RNNTLoss(blank=0) default; the reference never overrides it      */
    int maxT;         /* acts.shape[1]                                                             */
    int maxU;         /* acts.shape[2] = max label length + 1                                      */
    int dtype;        /* rnntDtype_t                                                               */
    /* Optional stage timing: if non-NULL, hipEvent_t handles recorded on `stream` at the stage
     * boundaries of rnnt_forward: [0] before K1, [1] after K1, [2] after K2 (+ cost reduction).
     * rnnt_backward records [0] before and [1] after K3.  Used by bench.py; NULL in production. */
    void *const *stage_events;
} rnntOptions;

/* Human-readable text for a status.  Mirrors upstream's rnntGetStatusString. */
const char *rnntGetStatusString(rnntStatus_t status);

/* Library/ABI version, bumped when a signature changes. */
int rnnt_mi355_abi_version(void);

/*
 * Bytes of device workspace for a minibatch.  Replaces upstream's
 * get_workspace_size(maxT, maxU, minibatch, gpu, &bytes) as bound by
 * warprnnt_pytorch for the call at model.py:116.  The workspace holds the
 * per-node denominators / blank / label log-probs written by K1, the alpha and
 * beta lattices written by K2 and the per-utterance log-likelihoods; it must
 * stay intact between rnnt_forward and rnnt_backward.  256-byte aligned.
 */
rnntStatus_t rnnt_get_workspace_size(int maxT, int maxU, int minibatch, size_t *size_bytes);
/* upstream-compatible alias: gpu must be non-zero (this library has no CPU path) */
rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, int gpu, size_t *size_bytes);

/*
 * Forward: K1 (log-softmax over V keeping lse / blank / next-label log-probs),
 * K2 (alpha and beta over the T+U anti-diagonals, one workgroup per utterance)
 * and the cost reduction.
 *   costs    [B] fp32     <- -log P(y_b | x_b)      (always written)
 *   loss_out fp32, may be NULL; [1] for MEAN (sum/B) / SUM, [B] for NONE
 * Reads acts exactly once.  Replaces the forward half of the native call under
 * RNNTLoss.forward (model.py:116).
 */
rnntStatus_t rnnt_forward(const void *acts, const int *labels, const int *act_lens,
                          const int *label_lens, int alphabet_size, int minibatch, float *costs,
                          float *loss_out, int reduction, void *workspace, rnntOptions options);

/*
 * Backward: K3, one fused pass that reads acts once more and writes every
 * element of grads once (zeros in padded nodes):
 *   grads[b] = grad_scale * grad_out[b * grad_out_stride] * d costs[b] / d acts[b]
 * grad_out is a DEVICE pointer to the upstream gradient (stride 0 for a scalar
 * loss, 1 for reduction NONE), so no host synchronisation is needed; NULL means 1.
 * grad_scale carries 1/B for MEAN.  Replaces upstream's gradient computation
 * plus the `grads * grad_output` / `grads / B` passes of its autograd glue
 * (the loss.backward() at /root/reference/bin/train.py:43).
 */
rnntStatus_t rnnt_backward(const void *acts, void *grads, const int *labels, const int *act_lens,
                           const int *label_lens, int alphabet_size, int minibatch,
                           const float *grad_out, int grad_out_stride, float grad_scale,
                           const void *workspace, rnntOptions options);

/*
 * Upstream-shaped single call: per-utterance costs and, when grads != NULL,
 * unscaled d costs[b] / d acts[b] — what warprnnt_pytorch's binding obtains
 * from the native library for the call at model.py:116.
 */
rnntStatus_t compute_rnnt_loss(const void *acts, void *grads, const int *labels,
                               const int *label_lens, const int *act_lens, int alphabet_size,
                               int minibatch, float *costs, void *workspace, rnntOptions options);

/*
 * compute_rnnt_loss without a workspace argument and without any allocation.  NO COUNTERPART IN THE
 * REFERENCE: it exists for callers that cannot allocate device memory — code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  The lattice workspace is a device array of
 * RNNT_STATIC_WORKSPACE_BYTES inside the library, one copy per GPU, held until the process ends;
 * its address is looked up once per device on the first call there (the device must be current, as for
 * every entry point).  A minibatch whose rnnt_get_workspace_size exceeds the array returns
 * RNNT_STATUS_STATIC_WORKSPACE_EXCEEDED and launches nothing.  All calls on one GPU share that GPU's
 * array: calls that overlap on one GPU (two streams, two threads) race; serialise them.  Nothing needs
 * clearing between calls (every value read is written earlier in the same call).  Same kernels, same
 * results bit for bit as compute_rnnt_loss.
 * Under HIP graph capture: make ONE call per device BEFORE capture begins (the first call on a device looks the
 * array's address up with hipGetDevice + hipGetSymbolAddress, a synchronous runtime call that may load the code
 * object and is not capturable); later calls only enqueue kernels.  Every replay of a captured graph uses the same
 * per-GPU array: replays of graphs that contain this call must not overlap on one GPU either.
 */
#define RNNT_STATIC_WORKSPACE_BYTES (32u << 20) /* 32 MiB: config 2 (B32 T400 U40) needs 21.7 MB */
rnntStatus_t compute_rnnt_loss_static(const void *acts, void *grads, const int *labels,
                                      const int *label_lens, const int *act_lens, int alphabet_size,
                                      int minibatch, float *costs, rnntOptions options);

/*
 * The cost reduction on its own: loss_out = sum(costs)/B (MEAN), sum (SUM) or a copy (NONE), summed in
 * index order so the result is bitwise reproducible.  Used by the batch-sharded harness
 * (rnnt_amd/parallel.py) after its all-reduce of per-utterance costs, so that the sharded scalar equals
 * the unsharded one bit for bit.  Counterpart of the `costs.sum() / B` in upstream's autograd glue.
 */
rnntStatus_t rnnt_reduce_costs(const float *costs, float *loss_out, int minibatch, int reduction,
                               void *stream);

/*
 * ---- next to the loss: the hidden layer of the joint network -----------------
 *
 * Replaces the repeat / cat / Linear(He+Hd -> H) / Tanh of the reference's JointNet.forward
 * (/root/reference/src/rnnt/model.py:46-54 with the mlp of :14-16).  Linear(cat(e,d)) = W_e e + W_d d + b,
 * so the caller runs two small GEMMs (enc_proj = enc W_e^T + b: [B,T,H]; dec_proj = dec W_d^T: [B,U1,H])
 * and this entry point produces what the reference's Tanh returns:
 *     hidden[b,t,u,:] = tanh(enc_proj[b,t,:] + dec_proj[b,u,:])        [B,T,U1,H], contiguous
 * joint="sum" (model.py:43-44) is the same kernel with W_e = W_d.  All three arrays have element type
 * `dtype`; arithmetic is fp32.  U1 = U+1 (the decoder saw BOS + U labels, model.py:106,114).
 */
rnntStatus_t rnnt_joint_hidden_forward(const void *enc_proj, const void *dec_proj, void *hidden, int B, int T,
                                       int U1, int H, rnntDtype_t dtype, void *stream);

/* Bytes of fp32 scratch rnnt_joint_hidden_backward needs (partial sums over frame chunks). */
rnntStatus_t rnnt_joint_hidden_workspace_size(int B, int T, int U1, int H, size_t *size_bytes);

/*
 * Backward of the above (what autograd runs through Tanh, Linear, cat and the two repeats in the
 * reference):  g = grad_hidden * (1 - hidden^2);  grad_enc_proj[b,t,:] = sum_u g;
 * grad_dec_proj[b,u,:] = sum_t g.  Sums are formed in a fixed order (no atomics): results are bitwise
 * reproducible.  The library allocates nothing; `workspace` comes from the caller.
 */
rnntStatus_t rnnt_joint_hidden_backward(const void *hidden, const void *grad_hidden, void *grad_enc_proj,
                                        void *grad_dec_proj, int B, int T, int U1, int H, rnntDtype_t dtype,
                                        void *workspace, void *stream);

/*
 * The pair above for a ragged batch whose consumer ignores the rows outside the utterances' lattices — the loss does
 * (/root/reference/src/rnnt/model.py:116 on the logits of :46-59), and rnnt_joint_loss_forward_compact /
 * _backward_compact never touch them: forward writes hidden[b,t,u,:] only for t < act_lens[b] and u <= label_lens[b]
 * (the other rows of `hidden` keep whatever they held); backward reads hidden and grad_hidden only there and takes the
 * gradient of every other row as zero (grad_enc_proj of frames past act_lens[b] and grad_dec_proj of positions past
 * label_lens[b] come out as zeros).  An utterance whose lengths do not fit the tensor has no rows.  Lengths are read
 * on the device.  Same workspace size.
 */
rnntStatus_t rnnt_joint_hidden_forward_ragged(const void *enc_proj, const void *dec_proj, void *hidden,
                                              const int *act_lens, const int *label_lens, int B, int T, int U1, int H,
                                              rnntDtype_t dtype, void *stream);
rnntStatus_t rnnt_joint_hidden_backward_ragged(const void *hidden, const void *grad_hidden, void *grad_enc_proj,
                                               void *grad_dec_proj, const int *act_lens, const int *label_lens, int B,
                                               int T, int U1, int H, rnntDtype_t dtype, void *workspace, void *stream);

/*
 * The same pair with `hidden` (and `grad_hidden`) in the COMPACTED node order: [hidden_rows, H], one row per lattice
 * node — utterance by utterance, frame by frame, label position fastest, the order of rnnt_compact_rows and of
 * rnnt_forward_compact's logits — instead of [B, T, U1, H].  For the caller that projects the nodes only (the
 * reference's Linear(512 -> V) of model.py:17,59 as a library GEMM over n rows, then rnnt_forward_compact): the hidden
 * vectors land where that GEMM reads them, without a gather, and their gradients are read from where its backward
 * leaves them, without a scatter into a zero-filled [B, T, U1, H] tensor.  The node numbering is derived on the device
 * from the two length tensors.  hidden_rows >= the number of nodes; frames the buffer has no room for are skipped
 * (forward) / taken as zero gradient (backward).  Same workspace size.
 */
rnntStatus_t rnnt_joint_hidden_forward_compact(const void *enc_proj, const void *dec_proj, void *hidden,
                                               long long hidden_rows, const int *act_lens, const int *label_lens, int B,
                                               int T, int U1, int H, rnntDtype_t dtype, void *stream);
rnntStatus_t rnnt_joint_hidden_backward_compact(const void *hidden, const void *grad_hidden, long long hidden_rows,
                                                void *grad_enc_proj, void *grad_dec_proj, const int *act_lens,
                                                const int *label_lens, int B, int T, int U1, int H, rnntDtype_t dtype,
                                                void *workspace, void *stream);

/*
 * ---- the joint's output projection fused with the loss's first pass ----------
 *
 * Replaces the pair  logits = Linear(H -> V)(hidden)  (/root/reference/src/rnnt/model.py:17,59) and the
 * forward half of the loss on those logits (model.py:116) for 16-bit operands: ONE kernel multiplies
 * hidden [B*T*U1, H] by weight [V, H] (nn.Linear's layout) + bias [V] (fp32) on the matrix cores, writes the
 * logits once to logits_out [B, T, U1, V] (rounded to `options.dtype`; rnnt_backward reads them) and keeps
 * each lattice node's max / sum-exp / blank and label log-probabilities without ever re-reading the logits:
 * the K1 pass of rnnt_forward disappears.  K2 and the cost reduction follow as in rnnt_forward; costs,
 * loss_out, reduction and workspace as there.  The softmax statistics are those of the ROUNDED logits, so
 * rnnt_backward(logits_out, ...) afterwards is exactly the backward of rnnt_forward(logits_out, ...).
 * options.dtype: RNNT_DTYPE_BF16 or RNNT_DTYPE_F16 (fp32 accumulation; fp32 operands are not taken: the
 * fp32 matrix rate is 1/16 of the 16-bit one); hidden_size: 128, 256 or 512; hidden and weight
 * 16-byte aligned; alphabet_size >= 64 (one tile), any value above, rows of logits_out need no alignment beyond
 * the element's.  INVALID_VALUE otherwise.  (Inside the call the alpha and beta planes of `workspace` hold partial
 * softmax statistics of row tiles whose vocabulary several workgroups share, until K2 writes the planes.)
 */
rnntStatus_t rnnt_joint_loss_forward(const void *hidden, const void *weight, const float *bias,
                                     void *logits_out, const int *labels, const int *act_lens,
                                     const int *label_lens, int hidden_size, int alphabet_size,
                                     int minibatch, float *costs, float *loss_out, int reduction,
                                     void *workspace, rnntOptions options);
/*
 * The same with padded rows: row r of logits_out starts logits_row_stride (>= alphabet_size) elements after row
 * r-1; the elements alphabet_size .. logits_row_stride-1 of a row are never touched.  A stride that is a multiple
 * of 64 elements puts every row on a 128-byte line: the kernel's 16-byte stores then never straddle two 64-byte
 * request blocks (V = 4232, bf16: 2.71 -> 2.35 ms for the forward at B32 T400 U40, profiles/r03_tuning.md).
 * rnnt_backward wants contiguous rows; rnnt_joint_loss_backward_strided takes the padded tensor.
 * (rnnt_joint_loss_forward is this call with logits_row_stride = alphabet_size.)
 */
rnntStatus_t rnnt_joint_loss_forward_strided(const void *hidden, const void *weight, const float *bias,
                                             void *logits_out, int logits_row_stride, const int *labels,
                                             const int *act_lens, const int *label_lens, int hidden_size,
                                             int alphabet_size, int minibatch, float *costs, float *loss_out,
                                             int reduction, void *workspace, rnntOptions options);

/*
 * Backward of the pair above with K3 as the PROLOGUE of the projection's two backward products: replaces what
 * autograd runs behind /root/reference/src/rnnt/model.py:17,59 and :116 when bin/train.py:43 calls
 * loss.backward() — the loss's gradient w.r.t. the logits (rnnt_backward: a [B,T,U1,V] tensor written once)
 * followed by grad_hidden = g W and grad_weight = g^T hidden (each reads that tensor back).  Here the gradient
 * w.r.t. the logits is never written: two MFMA kernels read the stored logits (logits_out of
 * rnnt_joint_loss_forward, or any logits rnnt_forward ran on with the same workspace), form the gradient tile in
 * registers exactly as rnnt_backward does (same operations, same rounding to options.dtype) and feed it to the
 * matrix cores.
 *   grad_hidden [B*T*U1, H] options.dtype   = g W                       (NULL: not computed)
 *   grad_weight [V, H]      options.dtype   = g^T hidden                (NULL: not computed)
 *   grad_bias   [V]         fp32            = column sums of g           (NULL: not computed)
 * where g = grad_scale * grad_out[b * grad_out_stride] * d costs[b] / d logits, rounded to options.dtype
 * (grad_out, grad_out_stride, grad_scale as rnnt_backward).  Sums over the lattice nodes are formed per node
 * range and added in range order: no atomics, results are bitwise reproducible.
 * workspace: the lattice workspace rnnt_joint_loss_forward / rnnt_forward filled.  backward_workspace: scratch of
 * rnnt_joint_loss_backward_workspace_size bytes, 256-byte aligned (node coefficients, partial sums of
 * grad_weight per node range; the size does not depend on the logits' row stride).  hidden_size 512 (the reference's
 * inner_dim), alphabet_size >= 128 and, for this contiguous call, a multiple of 8 (rows of the logits 16-byte
 * aligned: they are fetched by LDS-DMA; see the strided call below), 16-bit options.dtype; hidden, weight, logits,
 * grad_hidden 16-byte and grad_weight 8-byte aligned.  INVALID_VALUE otherwise: such callers run rnnt_backward and
 * two library GEMMs (rnnt_amd/fused.py does).
 * options.stage_events (optional): [0] start, [1] after the node coefficients, [2] after grad_hidden, [3] end.
 */
rnntStatus_t rnnt_joint_loss_backward_workspace_size(int hidden_size, int alphabet_size, int minibatch, int maxT,
                                                     int maxU, size_t *size_bytes);
rnntStatus_t rnnt_joint_loss_backward(const void *hidden, const void *weight, const void *logits, const int *labels,
                                      const int *act_lens, const int *label_lens, int hidden_size,
                                      int alphabet_size, int minibatch, const float *grad_out, int grad_out_stride,
                                      float grad_scale, void *grad_hidden, void *grad_weight, float *grad_bias,
                                      const void *workspace, void *backward_workspace, rnntOptions options);
/*
 * The same on logits with padded rows (logits_out of rnnt_joint_loss_forward_strided): logits_row_stride a multiple
 * of 8 elements.  With logits_row_stride >= alphabet_size rounded up to a multiple of 128 ANY alphabet_size >= 128 is
 * taken (the reference's 4234 and 5455 too): the kernels then read whole tiles, on into the padding, and mask the
 * entries from alphabet_size on (whatever the padding holds, NaN included).  With a smaller stride alphabet_size
 * must be a multiple of 8, as for the contiguous call (which is this call with logits_row_stride = alphabet_size).
 */
rnntStatus_t rnnt_joint_loss_backward_strided(const void *hidden, const void *weight, const void *logits,
                                              int logits_row_stride, const int *labels, const int *act_lens,
                                              const int *label_lens, int hidden_size, int alphabet_size, int minibatch,
                                              const float *grad_out, int grad_out_stride, float grad_scale,
                                              void *grad_hidden, void *grad_weight, float *grad_bias,
                                              const void *workspace, void *backward_workspace, rnntOptions options);

/*
 * The pair above in the COMPACTED node order: for ragged batches.  The reference's joint computes all B * maxT * U1
 * rows of the logits (/root/reference/src/rnnt/model.py:46-59) and its loss then ignores the rows outside the
 * utterances' lattices (t >= act_lens[b] or u > label_lens[b]); with the projection and the loss in one call nothing
 * outside the lattices has to exist.  These calls walk the nodes INSIDE the lattices only — utterance by utterance,
 * frame by frame, label position fastest, numbered on the device from the two length tensors (the host never reads
 * them) — and keep the logits in that order: row k of `logits` belongs to the k-th node, not to row k of the
 * [B, maxT, U1] tensor.  Matrix work, logits memory and HBM traffic then follow the sum of T_b (U_b + 1), not
 * B maxT U1 (a batch cut like BASELINE config 3 has 35 % of its rows inside lattices).  hidden and grad_hidden stay
 * [B, maxT, U1, H]: the kernels gather the nodes' hidden vectors and scatter their gradients; hidden's rows that are
 * no node are never read; grad_hidden's are set to zero (zero_padding_rows = 1) or left untouched (0: for a consumer
 * that skips them itself, rnnt_joint_hidden_backward_ragged).
 *   logits_rows        rows the logits buffer holds (logits_rows * logits_row_stride elements): an upper bound of the
 *                      number of nodes known on the host; minibatch * maxT * maxU always suffices.  The grids are cut
 *                      for it, workgroups past the last node leave at once.  Should the lattices hold MORE nodes, the
 *                      surplus nodes are not computed: the costs of their utterances come out NaN.
 *   compact_workspace  rnnt_joint_loss_compact_workspace_size bytes, 256-byte aligned: the node -> row map, written
 *                      by the forward call, read by the backward call (with the lattice workspace).
 *   backward_workspace rnnt_joint_loss_backward_compact_workspace_size(hidden_size, alphabet_size, logits_rows) bytes.
 * Everything else as rnnt_joint_loss_forward_strided / rnnt_joint_loss_backward_strided, whose results these calls
 * reproduce on a batch without padding; minibatch * maxT * maxU must fit 31 bits.  Utterances whose lengths do not
 * fit the tensor have no nodes (NaN cost, zero gradient, as everywhere in this library).
 */
rnntStatus_t rnnt_joint_loss_compact_workspace_size(int minibatch, int maxT, int maxU, size_t *size_bytes);
rnntStatus_t rnnt_joint_loss_forward_compact(const void *hidden, const void *weight, const float *bias,
                                             void *logits_out, int logits_row_stride, long long logits_rows,
                                             const int *labels, const int *act_lens, const int *label_lens,
                                             int hidden_size, int alphabet_size, int minibatch, float *costs,
                                             float *loss_out, int reduction, void *workspace, void *compact_workspace,
                                             rnntOptions options);
rnntStatus_t rnnt_joint_loss_backward_compact_workspace_size(int hidden_size, int alphabet_size, long long logits_rows,
                                                             size_t *size_bytes);
rnntStatus_t rnnt_joint_loss_backward_compact(const void *hidden, const void *weight, const void *logits,
                                              int logits_row_stride, long long logits_rows, const int *labels,
                                              const int *act_lens, const int *label_lens, int hidden_size,
                                              int alphabet_size, int minibatch, const float *grad_out, int grad_out_stride,
                                              float grad_scale, void *grad_hidden, int zero_padding_rows, void *grad_weight,
                                              float *grad_bias, const void *workspace, const void *compact_workspace,
                                              void *backward_workspace, rnntOptions options);

/*
 * ---- the loss itself on logits kept in the compacted node order (for the same ragged batches) --------------------
 * The operator the reference calls (/root/reference/src/rnnt/model.py:116) takes the logits of every row of the
 * [B, maxT, U1] tensor and ignores those outside the lattices; a caller that produces the logits itself (the joint's
 * output projection as a library GEMM: fp32 training, the reference's own precision) need not produce them.  These
 * calls are rnnt_forward / rnnt_backward on `acts` [acts_rows, V] holding one row per lattice NODE in the compacted
 * order above: utterance by utterance, frame by frame, label position fastest.  rnnt_compact_rows writes the map from
 * node number to tensor row (with which the caller gathers the hidden vectors it projects: row rowmap[k] of
 * hidden [B * maxT * U1, H] gives node k) — on the device, from the two length tensors.  Costs are those of
 * rnnt_forward on the padded tensor bit for bit; grads [acts_rows, V] holds the rows of rnnt_backward's gradient that
 * are nodes, bit for bit (rows of logits that are whole 128-byte lines and longer than 2 KiB; others start at another offset inside a line
 * in the two layouts, which changes the order of a row's sum of exponentials: last-bit differences of the log-sum),
 * and no zeros for padding are ever written (on a batch cut like BASELINE config 3 two thirds of
 * rnnt_backward's bytes are such zeros).  acts_rows: rows the two buffers hold, >= the number of nodes (an upper bound
 * known on the host; a smaller buffer gives NaN costs for the utterances it cuts, nothing is read or written past it).
 * rnnt_compact_rows: `capacity` = entries of rowmap; scratch = 16 * (minibatch + 1) bytes, 16-byte aligned;
 * minibatch * maxT * maxU must fit 31 bits.  Workspace, options, reduction, grad_out as rnnt_forward / rnnt_backward.
 */
rnntStatus_t rnnt_compact_rows(const int *act_lens, const int *label_lens, int minibatch, int maxT, int maxU,
                               long long capacity, int *rowmap, void *scratch, void *stream);
rnntStatus_t rnnt_forward_compact(const void *acts, long long acts_rows, const int *labels, const int *act_lens,
                                  const int *label_lens, int alphabet_size, int minibatch, float *costs, float *loss_out,
                                  int reduction, void *workspace, rnntOptions options);
rnntStatus_t rnnt_backward_compact(const void *acts, void *grads, long long acts_rows, const int *labels,
                                   const int *act_lens, const int *label_lens, int alphabet_size, int minibatch,
                                   const float *grad_out, int grad_out_stride, float grad_scale, const void *workspace,
                                   rnntOptions options);

/*
 * ---- greedy decoding --------------------------------------------------------
 * out[r] = first index of the maximum of row r of x[rows, V] (a NaN counts as the maximum: torch.argmax's
 * conventions).  Replaces the per-frame softmax -> argmax of the reference's greedy loop
 * (/root/reference/src/rnnt/model.py:152-153; softmax is monotone, so the argmax of the logits is taken)
 * for a whole window of frames in one launch; rnnt_amd/decode.py builds the loop of :145-165 on it.
 */
rnntStatus_t rnnt_row_argmax(const void *x, int rows, int V, rnntDtype_t dtype, int *out, void *stream);

/*
 * The same with the score of the prediction: logp[r] = max_v log_softmax(x[r, :])_v, the pair
 * `logp_targets, positions = torch.max(self.softmax(log_probs)..., dim=1)` of the reference's batched greedy
 * search (/root/reference/src/rnnt/beam_searcher.py:91-93) for all rows in one launch.  A row holding a NaN
 * or +inf gives NaN, as torch does.  logp may be NULL (then this is rnnt_row_argmax).
 * rnnt_amd/decode.py builds the loop of beam_searcher.py:65-127 on it.
 */
rnntStatus_t rnnt_row_max_logp(const void *x, int rows, int V, rnntDtype_t dtype, int *out, float *logp,
                               void *stream);

/* hipEvent helpers so a ctypes caller (bench.py) can time stages without a HIP binding. */
rnntStatus_t rnnt_event_create(void **event);
rnntStatus_t rnnt_event_destroy(void *event);
rnntStatus_t rnnt_event_record(void *event, void *stream);
rnntStatus_t rnnt_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */

#ifdef __cplusplus
}
#endif
#endif /* RNNT_MI355_H_ */
