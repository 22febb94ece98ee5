/*
 * neuralcx.h -- C ABI of libneuralcx_hip.so: the NeuralCX hot path on MI355X (gfx950).
 *
 * Drop-in boundary for gabegrand/VQA-Counterexamples.  Every entry point names the reference
 * code it replaces (paths relative to the reference root).  The reference is pure Python on
 * PyTorch, so "what its FFI would bind" is the body of vqa.models.cx.NeuralModel.forward and the
 * loss / optimiser calls of the training loop in counterexamples.py; the ctypes binding a
 * maintainer adds is shown in INTEGRATION.md and shipped in
 * vqa-counterexamples_amd/neuralcx/_lib.py.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes + a HIP stream (hipStream_t passed as void*); no
 *     torch types, no exceptions, no allocation inside, no global state on the product path (the
 *     only process-level state is the OPT-IN diagnostics at the end of this file -- ncx_profile_begin /
 *     _end and ncx_profile_stamps, the latter keyed by device -- and the lazily created per-device
 *     side stream, off unless NCX_SIDE_STREAM is set).  The caller owns every
 *     buffer (PyTorch's caching allocator in the shipped binding) and must keep them alive until
 *     the enqueued work has completed.
 *   - all floating point data is fp32, contiguous row-major; indices are int32.
 *   - weights use torch's nn.Linear layout [out, in]; linear_1.weight keeps the reference's
 *     concat column order (vqa/models/cx.py:309-320) so checkpoints are interchangeable.
 *   - return value: NCX_OK (0), a negative NCX_E_* for invalid arguments (nothing was enqueued),
 *     or a positive hipError_t passed through.
 *   - everything is enqueued on `stream`; nothing synchronises the host.
 *   - results are deterministic run to run (no floating point atomics anywhere).
 */
#ifndef NEURALCX_H
#define NEURALCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCX_OK            0
#define NCX_E_NULL       -1   /* required pointer is NULL */
#define NCX_E_DIMS       -2   /* a dimension is out of the supported range */
#define NCX_E_WORKSPACE  -3   /* workspace too small / misaligned */
#define NCX_E_FLAGS      -4   /* inconsistent flags / lesion inputs */
#define NCX_E_UNSUPPORTED -5  /* the RCCL library could not be loaded (ncx_comm_* / ncx_allreduce only) */
#define NCX_E_COMM       -6   /* RCCL reported an error */

/* model_spec switches of the reference (vqa/models/cx.py:265-307), 1 = feature present */
#define NCX_F_V_MULT   (1u << 0)   /* v_orig * v_other segment (cx.py:295-298), else zeros          */
#define NCX_F_V_DIST   (1u << 1)   /* pairwise_distance column (cx.py:299-302), else zero           */
#define NCX_F_V_RANK   (1u << 2)   /* one-hot candidate rank (cx.py:303-305), else inputs->v_rank   */
#define NCX_F_A_EMB    (1u << 3)   /* answer embeddings (cx.py:279-282), else noise blocks          */
#define NCX_F_ALL      (NCX_F_V_MULT | NCX_F_V_DIST | NCX_F_V_RANK | NCX_F_A_EMB)
/* BASELINE configs[4] ("bf16 weights"; net-new, the reference is fp32 only): the two dominant GEMMs -- candidate
 * segments of linear_1 forward, and their weight gradient -- and the three answer-embedding products (Gt forward,
 * d linear_1.weight[:, a_emb_other] and d answer_embedding backward) take bf16 operands (round-to-nearest-even
 * copies of the fp32 master weights, inputs and gradients) with fp32 accumulation on the bf16 MFMA path; everything
 * else, including Adam on the fp32 master weights, is unchanged.  Needs NCX_F_ALL (no lesions). */
#define NCX_F_BF16     (1u << 4)
/* Evaluation loops (eval_model, counterexamples.py:450-490): Gt = W1[:, a_emb_other] . E^T depends on the weights only.
 * With this bit ncx_forward trusts the Gt left in the workspace by an earlier ncx_forward on the SAME workspace and
 * weights and skips that GEMM.  The caller owns the invariant (the engine sets it from the second batch of an
 * evaluation pass on). */
#define NCX_F_REUSE_GT (1u << 5)
/* Training steps (counterexamples.py:330-339: forward, criterion, backward back to back): with this bit the out layer
 * (cx.py:327), the loss / Recall pass and the head of the backward run as ONE pass over the last hidden activations in
 * ncx_train_tail -- ncx_forward then leaves `scores` alone (it may be NULL) and ncx_backward[_phase] expects ncx_train_tail to
 * have run on the same workspace (its `dscores` may be NULL).  K <= 32 and H <= 256 only (ncx_train_tail says NCX_E_DIMS
 * otherwise: clear the bit and use the three calls). */
#define NCX_F_FUSED_TAIL (1u << 6)
/* "bf16 x 6": fp32-grade arithmetic on the bf16 matrix path.  NOT the default -- the headline path is fp32 MFMA, and this bit is never set unless the
 * caller sets it.  With it, the three big products of the training step take their fp32 operands as THREE bf16 planes each (x = x1 + x2 + x3
 * EXACTLY: x1 = x & 0xFFFF0000, x2 = (x - x1) & 0xFFFF0000, x3 = x - x1 - x2, cut when a tile is stored to LDS) and run the six plane products
 * that matter (a1 x1 + a1 x2 + a2 x1 + a1 x3 + a2 x2 + a3 x1) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; products of bf16 values are
 * exact in fp32, so what is dropped is 2^-24 relative: the rounding error of ONE fp32 operation (unlike the two-plane "bf16 x 3" form, whose
 * 2^-16 error fails the suite's ReLU-kink conditioning: DESIGN 5d).  Covered:
 *   - linear_1 forward on the 192-row form at K = 24 (per-triplet fold: the effective weight is formed in fp32, b = fma(v_o, W_m, W_k), then cut),
 *   - d linear_1.weight[:, v_other | v_orig*v_other] (the per-triplet fold pass, H % 256 == 0 and dv % 64 == 0),
 *   - dGt and every other column block of d linear_1.weight (the balanced TN launch, B <= 2048).
 * Shapes outside fall back to the fp32 kernels silently (same results to rounding).  Results agree with the fp32 kernels to fp32 rounding, not
 * bitwise (another summation order); every parity test of the suite passes at its unchanged tolerance with the bit set (NCX_X6=1 in the
 * environment of the Python binding sets it for a whole process). */
#define NCX_F_X6       (1u << 7)
/* (v_emb / q_emb / z_emb lesions replace INPUTS by uniform noise: the host does that before the call) */

typedef struct ncx_dims {
    int32_t B;        /* triplets in this (local) batch                                             */
    int32_t K;        /* candidates per triplet, knn_size (24 in the reference; 3..64 supported)     */
    int32_t dv;       /* dim_v  image feature width (2048)                                          */
    int32_t dq;       /* dim_q  question embedding width (2400)                                     */
    int32_t dz;       /* dim_mm multimodal fusion width (360)                                       */
    int32_t da;       /* dim_a  answer embedding width (2400, cx.py:235)                            */
    int32_t A;        /* ans_size, rows of answer_embedding (2000)                                  */
    int32_t H;        /* dim_h                                                                      */
    int32_t L;        /* n_layers, 1..3                                                             */
    int32_t n_img;    /* rows of the image feature table                                            */
    uint32_t flags;   /* NCX_F_*                                                                    */
    int32_t training; /* 1: apply dropout (cx.py:322-326), 0: eval                                  */
    float   drop_p;   /* nn.Dropout p (cx.py:259)                                                   */
    float   loss_scale; /* multiplies dscores/loss: 1/B_global (counterexamples.py:334); 0 => 1/B   */
    uint64_t seed;    /* per-step dropout seed (counter-based generator, see oracle/ncx_oracle.py)  */
} ncx_dims;

/* Inputs of NeuralModel.forward after vqa_forward (cx.py:261-285); all DEVICE pointers.
 * The reference gathers image_features[B,K+1,dv] on the host (counterexamples.py:540-541); here the
 * gather is folded into the kernels: `feats` is the resident table and img_idx the rows.  A caller
 * holding an already gathered [B,K+1,dv] block passes it as feats with img_idx = 0..B*(K+1)-1. */
typedef struct ncx_inputs {
    const float*   feats;        /* [n_img, dv]                                                    */
    const int32_t* img_idx;      /* [B, K+1]  column 0 = original image, 1..K = the K candidates   */
    const float*   q_emb;        /* [B, dq]                                                        */
    const float*   z_orig;       /* [B, dz]                                                        */
    const float*   z_knns;       /* [B, K, dz]                                                     */
    const float*   a_knns;       /* NCX_F_A_EMB: [B, K, A] answer LOGITS; else [B, K, da] noise    */
    const int32_t* answer_aids;  /* [B]   (NCX_F_A_EMB)                                            */
    const float*   a_emb_gt;     /* [B, da] noise block, only without NCX_F_A_EMB                  */
    const float*   v_rank;       /* [B, K, K] noise block, only without NCX_F_V_RANK               */
    const float*   keep_mask;    /* optional explicit dropout keep masks [L][B*K][H] (0/1 floats);
                                    NULL => counter-based generator keyed by dims->seed            */
} ncx_inputs;

/* Trainable tensors, state_dict names of the reference (cx.py:240-257). */
typedef struct ncx_params {
    const float* answer_embedding;  /* [A, da]                                   */
    const float* w1;  const float* b1;   /* linear_1.weight [H, Din], .bias [H]  */
    const float* w2;  const float* b2;   /* linear_2 [H,H],[H]   (L >= 2)        */
    const float* w3;  const float* b3;   /* linear_3 [H,H],[H]   (L >= 3)        */
    const float* w_out; const float* b_out; /* out.weight [1,H], out.bias [1]    */
} ncx_params;

typedef struct ncx_grads {          /* same shapes; OVERWRITTEN (not accumulated) by ncx_backward */
    float* answer_embedding;
    float* w1;  float* b1;
    float* w2;  float* b2;
    float* w3;  float* b3;
    float* w_out; float* b_out;
} ncx_grads;

/* Din = 3*dv + 2*da + 2*dz + dq + K + 1   (cx.py:245-251) */
int64_t ncx_input_size(const ncx_dims* d);

/* Bytes of scratch + saved activations ncx_forward/ncx_backward need for `d` (256-byte aligned base
 * required).  The same buffer must be passed to the ncx_backward that follows an ncx_forward. */
size_t ncx_workspace_bytes(const ncx_dims* d);

/* Replaces NeuralModel.forward (vqa/models/cx.py:261-333) from the answer-embedding lookups down:
 * K3/K4 softmax(a_knns) x answer_embedding and embedding(answer_aids) (cx.py:280-282), the per
 * candidate feature synthesis (cx.py:295-307), the concat (cx.py:309-320, never materialised),
 * linear_1..3 + ReLU + Dropout (cx.py:322-326) and `out` (cx.py:327).  scores: [B, K]. */
int ncx_forward(const ncx_dims* d, const ncx_inputs* in, const ncx_params* p,
                void* workspace, size_t workspace_bytes, float* scores, void* stream);

/* The same forward cut where the data ends and the weights begin (net-new; the reference is single-GPU: the cut exists for
 * the data-parallel step, counterexamples.py:334-339 being where a DP job sums gradients):
 *   NCX_FWD_PRELUDE  everything that is a function of the batch alone -- the feature-table row ids of every candidate row,
 *                    the pairwise distance (cx.py:300), the rank one-hot (cx.py:304-305), the softmax statistics of the
 *                    answer logits (cx.py:281); in the bf16 variant also the packed candidate rows.  Reads no weight
 *                    (`p` is validated like in ncx_forward but not dereferenced on the device; `scores` may be NULL).
 *   NCX_FWD_REST     everything that reads the weights (padded weight copies, Gt, Sh, linear_1..3, out).
 * PRELUDE then REST == ncx_forward bit for bit, on the same workspace.  A DP job enqueues step n + 1's PRELUDE before it waits
 * for step n's last gradient bucket and applies that bucket's Adam slice: the exchange hides under it. */
#define NCX_FWD_ALL     0
#define NCX_FWD_PRELUDE 1
#define NCX_FWD_REST    2
int ncx_forward_phase(const ncx_dims* d, const ncx_inputs* in, const ncx_params* p,
                      void* workspace, size_t workspace_bytes, float* scores, int32_t phase, void* stream);

/* Replaces nn.CrossEntropyLoss(size_average=False)(scores, comp_idxs) / len(batch)
 * (counterexamples.py:310,334) and recallAtK (counterexamples.py:501-506) in one pass.
 *   loss_rows[B]  per-triplet CE * scale          (nullable)
 *   loss[1]       sum of loss_rows                (nullable)
 *   dscores[B,K]  (softmax - onehot) * scale      (nullable)   == d loss / d scores
 *   rank[B]       #{k: s_k > s_gt} + #{k < gt: s_k == s_gt}    (nullable)
 *   hits[2]       OVERWRITTEN with #{rank < 1}, #{rank < 5}    (nullable)
 * scale = 1/B when scale <= 0.   K <= 64. */
int ncx_loss_rank(const float* scores, const int32_t* gt, int32_t B, int32_t K, float scale,
                  float* loss_rows, float* loss, float* dscores, int32_t* rank, int32_t* hits,
                  void* stream);

/* Replaces loss.backward() (counterexamples.py:338) for the tensors of ncx_params: given
 * dscores[B,K] = d loss / d scores, writes every gradient.  vqa_model is frozen in the reference
 * (cx.py:73-80), so no input gradients are produced. */
int ncx_backward(const ncx_dims* d, const ncx_inputs* in, const ncx_params* p,
                 void* workspace, size_t workspace_bytes, const float* dscores,
                 const ncx_grads* g, void* stream);

/* ncx_backward in halves, for overlapping the gradient exchange of a data-parallel job with compute
 * (net-new: the reference is single-GPU).  Two ways to cut it, each bit-identical to ncx_backward (phase 0) when
 * both halves run in order on the same stream and workspace:
 *   phase 1 | 2:  1 = out.*, linear_2/3.*, linear_1.bias and the complete answer_embedding gradient;
 *                 2 = linear_1.weight.
 *   phase 3 | 4:  3 = everything except the answer_embedding gradient; it leaves the block dGt | dGgt (the gradient
 *                 w.r.t. W1[:, a_emb_other] . E^T and the scattered dSh; layout: see NCX_WS_DGT below) in the workspace
 *                 region ncx_ws_region(NCX_WS_DGT);  4 = answer_embedding gradient = dGt^T . W1ak + dGgt^T . W1agt.
 *                 The embedding gradient is linear in that region, so a DP job sums the 4 MB region over ranks
 *                 between 3 and 4 instead of all-reducing the 19 MB [A, da] gradient (every rank then computes the
 *                 same, complete gradient).
 *   phase 5 | 2 | 4:  5 = phase 1 without the answer_embedding product: out.*, linear_2/3.*, linear_1.bias, and the
 *                 region dGt | dGgt complete (as after phase 3) -- a DP job starts summing the region here, runs
 *                 phase 2 (linear_1.weight: most of the backward) under that exchange, starts the exchange of the
 *                 remaining gradients, and runs phase 4 under it. */
int ncx_backward_phase(const ncx_dims* d, const ncx_inputs* in, const ncx_params* p,
                       void* workspace, size_t workspace_bytes, const float* dscores,
                       const ncx_grads* g, int32_t phase, void* stream);

/* Byte offset (from the 256-byte aligned workspace base) and size of a named workspace region.
 * NCX_WS_DGT: the block the answer_embedding gradient is linear in, in the form the library's own phases produce and
 * consume it -- fp32 path: dGt^T | dGgt^T, 2 x [A][pad4(H)] fp32 (reduction index contiguous: the embedding gradient runs
 * in NT form; 4.1 MB at A = 2000, H = 256); bf16 variant: dGt | dGgt, 2 x [H][A] fp32.  A DP job sums exactly
 * [offset, offset + bytes) over ranks between phase 5 (or 3) and phase 4; size 0 when the a_emb segment is lesioned. */
#define NCX_WS_DGT 1
#define NCX_WS_H1 2     /* diagnostics / tests: post-dropout activations of linear_1, [B*K, H] (valid after ncx_forward) */
#define NCX_WS_DPRE1 3  /* diagnostics / tests: gradient of linear_1's pre-activations, [B*K, H] (valid after ncx_backward) */
int ncx_ws_region(const ncx_dims* d, int32_t which, size_t* offset, size_t* bytes);

/* NCX_F_FUSED_TAIL: replaces, in one pass over h_L, the tail of ncx_forward (`out`, cx.py:327), ncx_loss_rank
 * (counterexamples.py:310,334,501-506) and the head of ncx_backward (d out.weight, d out.bias, the gradient of the last
 * pre-activations; for L == 1 also d linear_1.bias).  Call order: ncx_forward, ncx_train_tail, ncx_backward[_phase], all with the
 * same dims (flag set) and workspace.  Outputs as ncx_loss_rank (scale = dims.loss_scale, 1/B when <= 0) plus `scores`
 * [B, K]; dscores is optional.  Bit-identical to the three separate calls except d out.bias (zero in maths; its partial sums
 * are taken in another order). */
int ncx_train_tail(const ncx_dims* d, const ncx_params* p, void* workspace, size_t workspace_bytes, const int32_t* gt,
                   float* scores, float* loss_rows, float* loss, float* dscores, int32_t* rank, int32_t* hits,
                   const ncx_grads* g, void* stream);

/* Gradient exchange of a data-parallel job (net-new: the reference is single-GPU; /root/reference/counterexamples.py:334-339
 * is where a DP job sums gradients between loss.backward() and optimizer.step()).  An opaque handle around one RCCL
 * communicator -- the library's only other state; RCCL is loaded on first use (NCX_E_UNSUPPORTED when it is absent).
 *   rank 0:      ncx_comm_unique_id(id)          -> 128 bytes the host hands to every rank (file, socket, MPI ...)
 *   every rank:  hipSetDevice(local GPU); ncx_comm_create(id, nranks, rank, &comm)
 *   per step:    ncx_allreduce(comm, buf, n, stream)   in-place fp32 SUM over ranks, ordered on `stream`
 *                (the engine's two buckets: the dGt | dGgt workspace region and the flat gradient tail)
 * The Python host uses torch.distributed's RCCL backend for the same collective. */
#define NCX_COMM_ID_BYTES 128
typedef struct ncx_comm ncx_comm;
int ncx_comm_unique_id(void* id128);
int ncx_comm_create(const void* id128, int32_t nranks, int32_t rank, ncx_comm** out);
int ncx_comm_destroy(ncx_comm* comm);
int ncx_allreduce(ncx_comm* comm, float* buf, size_t n, void* stream);

/* Replaces torch.optim.Adam(...).step() (counterexamples.py:275-276,339) on a flat fp32 buffer:
 * defaults betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad.  `step` is the 1-based step
 * count; grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int ncx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                  float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale,
                  void* stream);

/* ---- the frozen VQA producer upstream of NeuralCX -----------------------------------------
 * Replaces CXModelBase.vqa_forward (vqa/models/cx.py:64-104) for the MutanNoAtt model in eval mode:
 * MutanFusion.forward (vqa/models/fusion.py:78-121: linear_v/linear_q + activation, R x {linear_hv_i, linear_hq_i,
 * hadamard}, sum) and AbstractNoAtt._classif (vqa/models/noatt.py:24-29) on the original + K candidate images.
 * The question branch is evaluated once per question (the reference duplicates q K+1 times first, cx.py:83-87), the
 * image gather is folded into linear_v, and the R rank-1 terms are folded inside ONE chained GEMM.  q_emb comes
 * from the question encoder (seq2vec), which stays on the PyTorch side. */
typedef struct ncx_mutan_params {
    const float* wv;  const float* bv;    /* fusion.linear_v.weight [dhv, dv], .bias [dhv]                       */
    const float* wq;  const float* bq;    /* fusion.linear_q.weight [dhq, dq], .bias [dhq]                       */
    const float* whv; const float* bhv;   /* fusion.list_linear_hv.{0..R-1} stacked: [R*dz, dhv], [R*dz]         */
    const float* whq; const float* bhq;   /* fusion.list_linear_hq.{0..R-1} stacked: [R*dz, dhq], [R*dz]         */
    const float* wc;  const float* bc;    /* linear_classif.weight [A, dz], .bias [A]                            */
    int32_t dhv, dhq, R;                  /* R <= 10                                                             */
    int32_t act_v, act_q;                 /* activation_v / activation_q: 0 none, 2 tanh                         */
} ncx_mutan_params;

size_t ncx_vqa_workspace_bytes(const ncx_dims* d, const ncx_mutan_params* m);

/* z_orig [B,dz], z_knns [B,K,dz], a_knns [B,K,A] (logits), a_orig [B,A] (nullable: NeuralModel never reads it).
 * Uses d->B, K, dv, dq, dz, A, n_img only. */
int ncx_vqa_forward(const ncx_dims* d, const float* feats, const int32_t* img_idx, const float* q_emb,
                    const ncx_mutan_params* m, void* workspace, size_t workspace_bytes,
                    float* z_orig, float* z_knns, float* a_knns, float* a_orig, void* stream);

/* ---- brute-force k nearest neighbours of feature rows --------------------------------------
 * Replaces knn.py:41-58 of the reference (sklearn NearestNeighbors(n_neighbors=k).fit(table).kneighbors(queries),
 * brute force, euclidean).  For each of the nq query rows: the k rows of `table` [n, dv] with the smallest
 * euclidean distance, ascending (ties by row index), as out_idx [nq, k] int64 and out_dist [nq, k] fp32.
 * One call handles one block of queries; the workspace holds -|x_j|^2/2 for the table (computed when
 * norms_ready == 0, reusable by later calls with the same table and workspace) and the nq x n product block.
 * 1 <= k <= min(n, 120); dv >= 4. */
size_t ncx_knn_workspace_bytes(int32_t n, int32_t block_rows);
int ncx_knn(const float* table, int32_t n, const float* queries, int32_t nq, int32_t dv, int32_t k,
            int32_t norms_ready, void* workspace, size_t workspace_bytes, int64_t* out_idx, float* out_dist,
            void* stream);

/* ---- diagnostics (bench.py / tests only; the only process-global state, off by default) --------------
 * GEMM ids: 0 Gt = W1[:,a_other].E^T, 1 Sh (shared segments), 2 MAIN (candidate segments, the dominant
 * forward kernel), 3 hidden layer l>=2 forward, 4 dW1 candidate columns (+dGt; the dominant backward
 * kernel; the shared columns ride in the same launch), 5 (unused: merged into 4), 6 dE, 7 dW1[:,a_other], 8 dA_gt, 9 dW_l (l>=2), 10 dX_l (l>=2).
 * ncx_profile_begin arms HIP-event timing (on the launch stream) of every launch of the gemm ids in the mask,
 * including a split-K fix-up; ncx_profile_end synchronises those events, writes up to `cap` (duration ms, id)
 * pairs and returns how many, then disarms.  Not thread safe; do not arm during graph capture. */
#define NCX_GEMM_GT 0
#define NCX_GEMM_SH 1
#define NCX_GEMM_MAIN 2
#define NCX_GEMM_FWD_L 3
#define NCX_GEMM_DW1C 4
#define NCX_GEMM_DW1S 5
#define NCX_GEMM_DE 6
#define NCX_GEMM_DW1AK 7
#define NCX_GEMM_DAGT 8
#define NCX_GEMM_DWL 9
#define NCX_GEMM_DXL 10
int ncx_profile_begin(uint32_t gemm_mask /* bit i = gemm id i */, int32_t max_launches);
int ncx_profile_end(float* ms, int32_t* ids, int32_t cap);
/* In-kernel clock diagnostics of the MAIN launch (the fused forward kernel of linear_1): while `stamps` is non-NULL
 * every fp32 MAIN launch makes thread 0 of each workgroup write 16 uint64 words to stamps[16 * workgroup id ...]:
 * word 0 = shader-cycle counter (s_memtime) at kernel entry, word 8 = the same at exit, words 14 / 15 = the 100 MHz
 * constant-rate counter (s_memrealtime) at entry / exit, word 13 = XCC id.  The clock a workgroup held is
 * (w8 - w0) / (w15 - w14) x 100 MHz.  `words` = capacity of the device buffer in uint64 (>= 16 x workgroups of the
 * launch, else the launch is not stamped).  Armed for the CURRENT device only (the buffer lives there; forwards on other
 * devices of the process never see it); ncx_profile_stamps(NULL, 0) disarms it.  The timed product path never arms it
 * (bench.py stamps a separate diagnostic pass after its timed region). */
int ncx_profile_stamps(unsigned long long* stamps, int64_t words);
/* out6 = {form (0 NT,1 TN,2 NN), M, N, 32-deep k-steps, tile cfg (0 64x64, 1 128x128, 2 96x128, 3 96x64, 4 128x64; the fused forward
 *         kernel: 5 48x128, 6 / 7 / 8 the per-triplet fold on 48x64 / 96x64 / 192x64 tiles), aligned k-chunks per output tile (1 = no split)} */
int ncx_plan_query(const ncx_dims* d, int32_t gemm_id, int32_t* out6);
/* Host-only self-check of the workgroup-id <-> (tile, k-chunk) maps (XCD-aware interleaved layout, chunk-per-XCD
 * layout with its padding ids): 0 when decode/encode are mutually inverse and cover every (tile, chunk) exactly once. */
int ncx_wgmap_check(int32_t tiles_m, int32_t tiles_n, int32_t S);

/* Library build id ("neuralcx-hip gfx950 <date>"). */
const char* ncx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NEURALCX_H */
