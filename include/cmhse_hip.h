/* cmhse_hip.h — C ABI of libcmhse_hip.so, the MI355X (gfx950) hot-path library.
 *
 * The reference (zbwglory/CMHSE) has no FFI layer: its hot path calls third-party PyTorch / NumPy
 * operators from Python.  Each entry point below replaces one of those call sites; the citation
 * (file:line into the reference tree) says which.  A maintainer of the reference binds these with
 * `ctypes`; cmhse_amd/_lib.py is that binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*; NULL = the null stream);
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - all float data is IEEE fp32, row-major, 16-byte aligned rows are the fast path (feature
 *     width and hidden size multiples of 4); indices are int32 unless stated (token ids: int64,
 *     as torch.LongTensor);
 *   - calls are asynchronous on `stream` and re-entrant; they allocate NO device memory: scratch
 *     is an explicit caller-owned workspace whose size the matching `*_workspace` function
 *     returns.  Host-side objects a call may take and give back before it returns: HIP events for
 *     stream fork / join (`tail_stream`, `stream`, `side_stream` of the job structs) and the event
 *     pairs owned by a cmhse_timer handle when the caller passes one (measurement only);
 *   - process-wide state, all of it host-side and mutex- or atomic-guarded: (o) 64 bytes of pinned
 *     host memory per device for cmhse_async_status, allocated at the first resident launch;
 *     (i) per-device free
 *     lists of those HIP events (creating an event while the GPU is busy can stall the host for
 *     tens of milliseconds, so events are recycled; an event returns to the list of the device
 *     that is current in the calling thread — use one device per thread across a call); (ii) the
 *     DEFAULT kernel-shape crossovers of cmhse_tune() below (a caller that must not share them
 *     enters a tuning context, cmhse_ctx_*: its own copy, current per thread).  Nothing else: no
 *     cached device memory, no cached streams, no environment variables;
 *   - return value: 0 = success, negative = error code (cmhse_strerror); no exceptions or aborts
 *     cross the ABI.
 */
#ifndef CMHSE_HIP_H_
#define CMHSE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  CMHSE_OK = 0,
  CMHSE_ERR_ARG = -1,        /* invalid argument (null pointer, non-positive size, bad mode) */
  CMHSE_ERR_WORKSPACE = -2,  /* workspace too small or misaligned */
  CMHSE_ERR_LAUNCH = -3,     /* HIP launch / runtime error (hipGetLastError text via strerror) */
  CMHSE_ERR_UNSUPPORTED = -4, /* shape outside what the kernels support */
  CMHSE_ERR_TIMEOUT = -5      /* an earlier resident-kernel launch on this device gave up at a grid
                                 barrier (see cmhse_async_status) */
};

/* pooling applied on top of the GRU hidden states (reference: model.py:27-34 `rnn_type`) */
enum {
  CMHSE_POOL_LAST = 0, /* layers.Seq2Seq.forward   layers.py:47-66   h at t = len-1            */
  CMHSE_POOL_ATTN = 1, /* layers.Attention.forward layers.py:93-119  masked exp-softmax pooling */
  CMHSE_POOL_MAX = 2,  /* layers.Maxout.forward    layers.py:185-204 max over valid steps       */
  CMHSE_POOL_ALL = 3,  /* decoder Seq2Seq_Decode + DecoderSequence.forward (decoder/layers.py:34-52,
                          decoder/model.py:36-45): every hidden state, sequence after sequence:
                          out[out_row[s] + t, :] = h_{s,t} */
  /* OR-ed into pool_mode for training: the forward also keeps gate activations (and the arg-max
   * step / tanh(lin(h))) in its workspace, which cmhse_gru_pool_bwd consumes. */
  CMHSE_SAVE_FOR_BACKWARD = 0x100,
  /* OR-ed into pool_mode: run the large GEMMs of the call on the bf16 matrix pipe with a 3-term
   * hi/lo split (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate): ~2^-17 relative error per
   * product, ~1e-6 on the embeddings (parity bar 1e-4); 2.7x less matrix time per product than exact
   * fp32 on the v_mfma_f32_32x32x8_bf16_1k pairs this library restricts itself to, 1.8x per validation
   * pass.  Inference calls only (ignored with CMHSE_SAVE_FOR_BACKWARD); applies to the steps with more
   * than `tiny_max_seqs` active sequences and to the attention projection of their rows; those
   * kernels stage pre-split operands by LDS-DMA and use 60 KB (steps) / 72 KB (projection) of LDS per
   * workgroup.  Outside the bit-identical-ranks contract.  Default (flag clear) is exact fp32. */
  CMHSE_MATH_BF16X3 = 0x200,
  /* OR-ed into the pool_mode of a cmhse_gru_job / cmhse_gru_bwd_job that has its own `stream`: the
   * call does NOT order its own stream argument behind that stream when it returns.  The job's
   * results are then ready on the job's stream only; a later job on the same stream may consume
   * them at once (level 2 of a tower right behind its level 1, without waiting for the other
   * tower), anything else must be ordered by the caller — a later call on the same job stream
   * without this flag does that for everything queued before it.  Buffers the job uses must stay
   * allocated until then. */
  CMHSE_NO_JOIN = 0x400
};

/* Weights of one encoder layer, laid out exactly as the reference's state-dict tensors
 * (layers.py:70-91; checkpoint keys rnn.rnn.weight_ih_l0 ...). */
typedef struct cmhse_gru_weights {
  const float* w_ih;  /* [3H, I]  rnn.rnn.weight_ih_l0, gate row order r,z,n (torch.nn.GRU) */
  const float* w_hh;  /* [3H, H]  rnn.rnn.weight_hh_l0 */
  const float* b_ih;  /* [3H]     rnn.rnn.bias_ih_l0 */
  const float* b_hh;  /* [3H]     rnn.rnn.bias_hh_l0 */
  const float* w_lin; /* [H, H]   rnn.lin.weight   (CMHSE_POOL_ATTN only, else NULL) */
  const float* b_lin; /* [H]      rnn.lin.bias */
  const float* w_att; /* [H]      rnn.att_w.weight ([1,H]) */
} cmhse_gru_weights;

/* A ragged batch of S sequences, already ordered by length, longest first — the order
 * torch.nn.utils.rnn.pack_padded_sequence imposes (layers.py:94-97).  The caller (host logic in
 * cmhse_amd/layers.py) sorts and fills these small arrays; the feature / token storage itself is
 * never copied or re-laid-out: each sequence is addressed through its own base pointer, so padded
 * [S,T,I] batches, several loader batches at once, and the consecutive-row inputs of
 * VSE.structure_emb (model.py:238-255) are all consumed in place. */
typedef struct cmhse_seq_batch {
  int32_t S;    /* sequences */
  int32_t Tmax; /* longest length (= lens[0]) */
  int32_t I;    /* input width: img_dim, word_dim or first-level size */
  int32_t H;    /* hidden size (embed_size) */
  const uint64_t* x_rows;   /* [S] address of step 0 of sequence s: fp32 rows, stride I floats;
                               NULL when tok_rows is used */
  int32_t x_step_floats;    /* floats between consecutive steps of a sequence in x_rows storage: I for
                               ordinary [T, I] rows, 0 for a time-constant input (the repeated
                               embedding a decoder is fed, model.py:261-265) */
  const uint64_t* tok_rows; /* [S] address of token 0 of sequence s (int64 ids, contiguous);
                               the embedding lookup of model.EncoderText.forward (model.py:94) is
                               fused into the operand load.  NULL when x_rows is used */
  const float* emb_table;   /* [vocab, I] embed.weight (tok_rows only) */
  int32_t vocab;
  const uint64_t* h0_rows;  /* [S] address of the initial hidden row of sequence s (H floats), or
                               NULL for h0 = 0 (layers.py:98-102 `hidden`) */
  const int32_t* lens;      /* [S] lengths, non-increasing, all >= 1 */
  const int32_t* out_row;   /* [S] row of `out` that receives sequence s (undoes the sort,
                               layers.py:116-117) */
  const int32_t* step_off;  /* [Tmax+1] step_off[t] = sum_{t'<t} #{s : lens[s] > t'}: row offset
                               of step t in the time-major packed hidden-state buffer */
  const int32_t* step_count_host; /* HOST [Tmax] #{s : lens[s] > t} (sizes the per-step grids) */
  void* step_timer;         /* optional cmhse_timer (or NULL): brackets the per-step GRU kernels of
                               this call on `stream` — measurement only, no effect on results */
  const void* const* step_events_host; /* optional HOST [Tmax] of hipEvent_t (NULL entries allowed), or
                               NULL: before step t's kernel is launched, its stream waits on entry t.
                               Lets the caller feed x rows chunk by chunk (cmhse_pull_steps on a copy
                               stream) while earlier steps compute */
  const int32_t* step_plan_host; /* optional HOST [Tmax], or NULL: #{s : lens[s] > t} over the WHOLE set of
                               sequences this batch is a share of (one rank's part of a validation
                               split, one super-batch of several: >= step_count_host[t], non-increasing).
                               Which kernel serves step t — the LDS-tiled one (one k-ordered sum over
                               [x_t | h_{t-1}]) or the small-batch one (hoisted input projection + K in
                               8 slices) — is then chosen from THIS count instead of the batch's own, so
                               a sequence is encoded bit for bit the same whatever share of the split
                               it is encoded with (parallel_eval: integer ranks identical for any
                               number of ranks).  NULL = the batch's own counts */
} cmhse_seq_batch;

/* Bytes of workspace cmhse_gru_pool_fwd needs for this batch: the time-major packed hidden states
 * hs[sumT, H] (kept for the attention pooling and for a later backward pass) plus the attention
 * energy partials. */
size_t cmhse_gru_pool_workspace(int32_t S, int32_t Tmax, int64_t sum_T, int32_t I, int32_t H,
                                int32_t pool_mode);

/* Where a named region of that workspace lives (tests and tools that inspect what a training
 * forward kept: the gate activations, the arg-max step of the max pooling ...).  name: "hs"
 * [sum_T, H] f32 | "gates" [sum_T, 4H] f32 (r, z, n, W_hn h + b_hn) | "argmax" [S, H] int32, rows in
 * SORTED sequence order (CMHSE_POOL_MAX with CMHSE_SAVE_FOR_BACKWARD) | "v" [sum_T, H] f32
 * (CMHSE_POOL_ATTN with CMHSE_SAVE_FOR_BACKWARD).  *bytes = 0 when the mode does not keep the
 * region.  Unknown name: CMHSE_ERR_ARG.  No effect on any computation. */
int cmhse_gru_pool_ws_region(int32_t S, int32_t Tmax, int64_t sum_T, int32_t I, int32_t H,
                             int32_t pool_mode, const char* name, size_t* offset, size_t* bytes);

/* Replaces the body of layers.{Seq2Seq,Attention,Maxout}.forward (layers.py:47-66, 93-119,
 * 185-204): 1-layer unidirectional GRU over the packed batch (nn.GRU, layers.py:31-34) followed
 * by the pooling `pool_mode`.  out[out_row[s], :] (row stride H) receives the un-normalised
 * embedding of sequence s.  The first sum_T*H floats of `workspace` hold, on return, the hidden
 * states in pack_padded_sequence's time-major packed order. */
int cmhse_gru_pool_fwd(const cmhse_seq_batch* seqs, const cmhse_gru_weights* w, int32_t pool_mode,
                       float* out, void* workspace, size_t workspace_bytes, void* stream);

#define CMHSE_MAX_JOBS 4
/* Several INDEPENDENT cmhse_gru_pool_fwd requests in one call (at most CMHSE_MAX_JOBS).  Results are bit-identical
 * to the separate calls; step t of every request shares one launch, so two encoders that do not
 * depend on each other — the clip and sentence encoders of VSE.forward_emb (model.py:222-236), the
 * video and paragraph encoders of structure_emb (model.py:238-255) — pay one launch latency and one
 * partially filled last wave of workgroups per time step instead of two — and, for inference calls,
 * the LDS-tiled steps of all requests that run them together are ONE launch altogether (the step
 * chain, "chain_min_steps" below: per-row-tile dependencies instead of a launch per time step).
 * The first request's step_timer (if set) brackets the step launches of the whole group. */
typedef struct cmhse_gru_job {
  const cmhse_seq_batch* seqs;
  const cmhse_gru_weights* weights;
  int32_t pool_mode;
  float* out;
  void* workspace;
  size_t workspace_bytes;
  void* tail_stream;   /* optional second hipStream_t (or NULL), ideally of higher priority.  When
                          this request's last time step is launched while other requests of the
                          call still have steps to go (a short chain beside a long one), the
                          REMAINING step launches of the call continue on tail_stream (ordered by
                          an event) and this request's attention projection + pooling start at
                          once on `stream`, beside that few-sequence, latency-bound tail.  The
                          call rejoins `stream` (event wait) before its last launches, so the
                          caller needs no extra ordering.  The same side stream also takes a
                          request whose chain has dropped to small-batch steps while another
                          request still launches LDS-tiled steps that do not fill the chip, so
                          that its short launches run beside those instead of between them. */
  void* stream;        /* optional hipStream_t of this request's OWN launches (or NULL = the call's
                          stream).  The towers of a training step are independent latency chains of
                          short dependent launches: given a stream each, they advance side by side
                          from their first step — the host queues step t of every request before
                          step t + 1 of any, so no chain waits for another chain's launches to be
                          queued.  The call forks the stream from, and joins it back into, the
                          call's stream.  Results do not depend on it. */
  void* side_stream;   /* optional hipStream_t (or NULL) for throughput work beside a chain that is
                          small-batch from its first step (a training batch): the hoisted input
                          projection x W_ih^T is then cut into chunks of time steps — only the first
                          stands in front of the chain, the others run on side_stream beside it, step
                          t waiting (event) for the chunk that holds its rows.  Results do not
                          depend on it. */
  void* out_ready_event; /* optional hipEvent_t (or NULL), recorded by the call where this request's
                          `out` becomes final: right behind its pooling pass when that is launched
                          early (tail_stream: its chain ended while other requests still step), else
                          at the end of the call on `stream`.  A consumer on another stream — the
                          device-to-host hand-over of evaluation.encode_data (evaluation.py:120-125) —
                          waits on it instead of on the whole call, and so runs beside the other
                          requests' remaining steps. */
} cmhse_gru_job;
int cmhse_gru_pool_fwd_multi(const cmhse_gru_job* jobs, int32_t n_jobs, void* stream);

/* Host -> HBM hand-over of the loader's feature tensors (the `.cuda()` of model.py:225-227,
 * evaluation.py:97-104), in the unit the step pipeline consumes: time steps [t0, t1) of the first
 * n_active sequences of a length-sorted batch.  src_rows_pinned[s] is the HOST address of step 0 of
 * sequence s inside page-locked memory (the DataLoader's pin_memory=True tensors,
 * activity_net/data.py:157-162; it must be device-readable, as hipHostMalloc / torch pinned memory
 * is), dst_rows[s] its device address; rows are row_floats floats apart in both.  Only valid steps
 * (t < lens[s]) are moved: zero padding never crosses PCIe.  Runs as a kernel on `stream` (give it a
 * stream of its own, record an event, hand it to the consumer via step_events_host). */
int cmhse_pull_steps(const uint64_t* src_rows_pinned, const uint64_t* dst_rows, const int32_t* lens,
                     int32_t n_active, int32_t row_floats, int32_t t0, int32_t t1, void* stream);

/* HBM -> host hand-over of finished embedding rows: what evaluation.encode_data does with
 * `.data.cpu()` + list.extend per batch (evaluation.py:120-125, 139-144).  `bytes` bytes from `src`
 * (device) to `dst_pinned` (page-locked host memory that is device-writable, as hipHostMalloc /
 * torch pinned memory is; both 16-byte aligned), as a kernel of `workgroups` workgroups (0 = 8) of
 * `waves` wavefronts (1..4, 0 = 1) on `stream`: small enough to run beside the level-2 encoders
 * and the ranking without taking their CUs, which the runtime's chip-wide blit copy does.
 * Asynchronous; record an event on `stream` (or synchronise it) before the host reads. */
int cmhse_push_rows(const void* src, void* dst_pinned, size_t bytes, int32_t workgroups, int32_t waves,
                    void* stream);

/* Do `bytes` bytes at `a` and at `b` differ?  `a`: device memory or page-locked host memory that the
 * device can read (read in place, over PCIe); `b`: device memory; both 16-byte aligned.  ORs 1 into
 * the device word *flag (zeroed by the caller) when they do.  evaluation.i2t / t2i (evaluation.py:
 * 160-213) use it to make sure the NumPy arrays they are handed still hold what encode_data wrote
 * before they report the ranking encode_data already queued on the device copies.  Asynchronous. */
int cmhse_rows_differ(const void* a, const void* b, size_t bytes, int32_t* flag, void* stream);

/* The padding half of the loader's collate_fn (activity_net/data.py:114-150, didemo_dev/data.py:
 * 133-165) as an index kernel: S ragged sequences stored back to back, row r of sequence s at
 * src + (first_row[s] + r) * row_bytes (frame features: row_bytes = 4 * img_dim; token ids: 8), are
 * written into the zero-padded block dst[S, Tmax, row_bytes].  The encoders do not need it (they
 * address ragged storage through cmhse_seq_batch.x_rows / tok_rows directly, so a packed batch is
 * uploaded without its padding and never padded); it exists for callers that want the reference's
 * padded tensors on the device.  row_bytes must be a multiple of 4. */
int cmhse_pad_rows(const void* src, const int64_t* first_row, const int32_t* lens, int32_t S,
                   int32_t Tmax, int32_t row_bytes, void* dst, void* stream);

/* torch.nn.functional.normalize(x) (p=2, dim=1, eps=1e-12) — call sites model.py:333-343,
 * evaluation.py:111-116.  y may alias x.  Rows have stride `ld` floats. */
int cmhse_l2norm_rows(const float* x, float* y, int32_t rows, int32_t cols, int64_t ld,
                      void* stream);

/* nn.Embedding lookup (model.EncoderText.forward, model.py:94): out[r,:] = table[ids[r],:] for
 * n ids (int64).  Only needed when the caller wants the word tensor itself (`return_word`,
 * model.py:233); the encoders fuse the lookup into cmhse_gru_pool_fwd. */
int cmhse_gather_rows(const float* table, const int64_t* ids, int64_t n, int32_t cols,
                      int32_t vocab, float* out, void* stream);

/* Replaces evaluation.i2t / t2i's `numpy.dot` + per-row `numpy.argsort` loop
 * (evaluation.py:164-171, 192-199) for the row stripe [row0, row0+nrows) of the N x M score
 * matrix d = A B^T (A [N,D], B [M,D], exact-fp32 MFMA).  The matrix is never materialised:
 *   rank[i - row0] = #{ j != i : d[i,j] > d[i,i] }   (position of i in the descending sort)
 *   top1[i - row0] = argmax_j d[i,j]                 (smallest j on exact ties)
 * for i in the stripe; the diagonal convention requires row0 + nrows <= M. */
size_t cmhse_sim_rank_workspace(int32_t nrows);
int cmhse_sim_rank(const float* A, const float* B, int32_t N, int32_t M, int32_t D, int32_t row0,
                   int32_t nrows, int32_t* rank, int32_t* top1, void* workspace,
                   size_t workspace_bytes, void* stream);

/* Same, with an optional cmhse_timer (or NULL) whose event pair brackets the counting pass (the
 * N x M MFMA contraction with the rank / arg-max epilogue) on `stream` — measurement only. */
int cmhse_sim_rank_ex(const float* A, const float* B, int32_t N, int32_t M, int32_t D,
                      int32_t row0, int32_t nrows, int32_t* rank, int32_t* top1, void* workspace,
                      size_t workspace_bytes, void* stream, void* timer);

/* Replaces loss.cosine_sim (loss.py:12-13): scores[n,m] = im s^T, exact fp32. */
int cmhse_cosine_sim(const float* im, const float* s, int32_t n, int32_t m, int32_t D,
                     float* scores, void* stream);

/* Replaces loss.ContrastiveLoss.forward (loss.py:86-117) for im [n,D], s [n,D]:
 *   cost_s = max(0, margin + S - diag_i), cost_im = max(0, margin + S - diag_j), diagonals cleared,
 *   max_violation: row/column maxima instead of all entries; norm: divide by n*n.
 * Writes the scalar loss to *loss (device).  `scores_out` ([n,n], may be NULL -> kept in the
 * workspace) receives the score matrix (needed by the backward pass). */
size_t cmhse_contrastive_workspace(int32_t n);
int cmhse_contrastive_fwd(const float* im, const float* s, int32_t n, int32_t D, float margin,
                          int32_t max_violation, int32_t norm, float* loss, float* scores_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Batched form of cmhse_contrastive_fwd for evaluation.encode_data's per-loader-batch 'Letest'
 * loss (evaluation.py:129 -> model.py:287-292): block b is the square problem on rows
 * [blk_off[b], blk_off[b+1]) of im and s (blk_off: device int32 [n_blocks+1], block sizes
 * <= max_n); losses[b] receives its loss.  One launch set for all blocks instead of one per
 * loader batch. */
size_t cmhse_contrastive_blocks_workspace(int32_t n_blocks, int32_t max_n);
int cmhse_contrastive_blocks_fwd(const float* im, const float* s, const int32_t* blk_off,
                                 int32_t n_blocks, int32_t max_n, int32_t D, float margin,
                                 int32_t max_violation, int32_t norm, float* losses,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Backward of cmhse_contrastive_blocks_fwd — what loss.backward() (model.py:367) computes through
 * the 4-7 ContrastiveLoss calls of a training step (model.py:333-343) when they are evaluated as
 * blocks of two row-blocked matrices: `scores` is the forward call's workspace (block b's stored
 * n_b x n_b scores at scores + b * max_n * max_n floats), grad_out[b] (device) the upstream gradient
 * of losses[b]; d_im / d_s ([rows, D], same row blocking as im / s) receive the gradients.  One
 * launch set (statistics, coefficient matrices, two batched TN products) for all blocks. */
size_t cmhse_contrastive_blocks_bwd_workspace(int32_t n_blocks, int32_t max_n);
int cmhse_contrastive_blocks_bwd(const float* im, const float* s, const float* scores,
                                 const int32_t* blk_off, int32_t n_blocks, int32_t max_n, int32_t D,
                                 float margin, int32_t max_violation, int32_t norm,
                                 const float* grad_out, float* d_im, float* d_s, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* The contrastive block of one training step (model.py:333-343) in one call: F.normalize of the
 * step's encoder outputs x[e] ([rows[e], D], device, contiguous) and the n_terms losses
 *   values[k] = ContrastiveLoss(normalize(x[term_a[k]]), normalize(x[term_b[k]]))   (loss.py:86-117)
 * plus their weighted total  *total = sum_k weight[k] * values[k]  (fp32, in term order) — the
 * reference's  loss_1 + loss_3 + (loss_5a + loss_5b) / 2 + ...  with weight 1 or 0.5.  Values are
 * those of cmhse_l2norm_rows + cmhse_contrastive_fwd per term, bit for bit (same kernels, the rows
 * normalised straight into the row-blocked operands).  The workspace keeps the normalised rows and
 * the stored scores for cmhse_step_losses_bwd, which must be given the same descriptor and
 * workspace: it writes dx[e] = d total / d x[e] * *grad_total for every e (dx: HOST array of n_emb
 * device pointers, each [rows[e], D]; an embedding no term uses gets zeros).  rows[term_a[k]] must
 * equal rows[term_b[k]].  Five launches each way on `stream`, no host synchronisation. */
#define CMHSE_STEP_LOSS_MAX 8
typedef struct cmhse_step_losses {
  int32_t n_emb, n_terms, D;
  const float* x[CMHSE_STEP_LOSS_MAX];
  int32_t rows[CMHSE_STEP_LOSS_MAX];
  int32_t term_a[CMHSE_STEP_LOSS_MAX], term_b[CMHSE_STEP_LOSS_MAX];
  float weight[CMHSE_STEP_LOSS_MAX];
  float margin;
  int32_t max_violation, norm;
} cmhse_step_losses;
size_t cmhse_step_losses_workspace(const cmhse_step_losses* d);
int cmhse_step_losses_fwd(const cmhse_step_losses* d, float* values, float* total, void* workspace,
                          size_t workspace_bytes, void* stream);
int cmhse_step_losses_bwd(const cmhse_step_losses* d, const float* grad_total, float* const* dx,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- backward pass (what loss.backward(), model.py:367, computes through the operators above) ---- */

/* Parameter gradients of one encoder layer, same shapes as cmhse_gru_weights; overwritten. */
typedef struct cmhse_gru_grads {
  float* dw_ih;  /* [3H, I] */
  float* dw_hh;  /* [3H, H] */
  float* db_ih;  /* [3H] */
  float* db_hh;  /* [3H] */
  float* dw_lin; /* [H, H]  (CMHSE_POOL_ATTN) */
  float* db_lin; /* [H] */
  float* dw_att; /* [H] */
} cmhse_gru_grads;

/* Backward of cmhse_gru_pool_fwd for the same `seqs` / `w` / pooling, given dout = d loss / d out
 * ([S,H], same row indexing as `out`) and the forward's workspace (run with
 * pool_mode | CMHSE_SAVE_FOR_BACKWARD).  Writes the parameter gradients to `grads` and, optionally,
 *   dx_rows      [S] device addresses: d loss / d x of step 0 of (sorted) sequence s goes there, rows
 *                of stride I floats (used for the level-2 encoders, whose inputs are level-1
 *                embeddings, model.py:252-253).  Ordinary rows are written exactly once; the
 *                row of a time-constant input (x_step_floats = 0) receives the SUM over its steps,
 *                accumulated with float atomics: zero it first;
 *   d_emb_table  [vocab, I]: gradient of embed.weight, ACCUMULATED with float atomics (zero it
 *                first); token batches only (model.py:94);
 *   dh0          [S, H] (row indexing of `out`): d loss / d hidden (layers.py:98-100).
 * Any of the three may be NULL; dx_rows and d_emb_table are mutually exclusive. */
size_t cmhse_gru_pool_bwd_workspace(int32_t S, int32_t Tmax, int64_t sum_T, int32_t I, int32_t H,
                                    int32_t pool_mode);
int cmhse_gru_pool_bwd(const cmhse_seq_batch* seqs, const cmhse_gru_weights* w, int32_t pool_mode,
                       const float* dout, const void* fwd_workspace, const cmhse_gru_grads* grads,
                       const uint64_t* dx_rows, float* d_emb_table, float* dh0, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Several INDEPENDENT cmhse_gru_pool_bwd requests in one call (at most CMHSE_MAX_JOBS), the
 * backward counterpart of cmhse_gru_pool_fwd_multi: the BPTT step of every request shares one
 * launch (chains are aligned at their LAST step), so the two towers of a training step
 * (model.py:319-321: clip_enc / txt_enc, then vid_seq_enc / txt_seq_enc) pay one dependent
 * launch per step instead of two.  Results are bit-identical to the separate calls. */
typedef struct cmhse_gru_bwd_job {
  const cmhse_seq_batch* seqs;
  const cmhse_gru_weights* weights;
  int32_t pool_mode;
  const float* dout;
  const void* fwd_workspace;
  const cmhse_gru_grads* grads;
  const uint64_t* dx_rows;
  float* d_emb_table;
  float* dh0;
  void* workspace;
  size_t workspace_bytes;
  void* stream;        /* optional hipStream_t of this request's own launches (see cmhse_gru_job.stream) */
  void* side_stream;   /* optional second hipStream_t (or NULL).  The BPTT chain of a training batch
                          is a sequence of short dependent launches that leaves most of the chip
                          idle; the weight-gradient products (dW_ih, dW_hh, the bias sums, dW_lin)
                          and d(input) are throughput work whose operands — the gate-derivative
                          rows of steps >= t — are final as soon as the chain has passed t.  With a
                          side stream they are launched there chunk of time steps by chunk, each
                          ordered behind the step that completed its rows by an event, and run
                          beside the rest of the chain; the call joins the side stream back into
                          `stream` before it returns.  Results are bit-identical with and without
                          it (the chunks accumulate in the same order either way). */
} cmhse_gru_bwd_job;
int cmhse_gru_pool_bwd_multi(const cmhse_gru_bwd_job* jobs, int32_t n_jobs, void* stream);

/* Backward of F.normalize (model.py:333-343): dx from x [rows, cols] (contiguous) and g = d/dy. */
int cmhse_l2norm_rows_bwd(const float* x, const float* g, float* dx, int32_t rows, int32_t cols,
                          void* stream);

/* Backward of ContrastiveLoss.forward (loss.py:86-117) from the score matrix the forward stored
 * (`scores_out`) and the upstream gradient *grad_out (device scalar): d_im, d_s [n, D]. */
size_t cmhse_contrastive_bwd_workspace(int32_t n);
int cmhse_contrastive_bwd(const float* im, const float* s, const float* scores, int32_t n, int32_t D,
                          float margin, int32_t max_violation, int32_t norm, const float* grad_out,
                          float* d_im, float* d_s, void* workspace, size_t workspace_bytes,
                          void* stream);

/* loss.GroupWiseContrastiveLoss.forward (loss.py:26-71, --weak_low_level_loss): clip x caption
 * scores [n, n], reduced per (video i, video j) block to their max (max_violation) or mean, then
 * the same hinge loss on the reduced [B, B] matrix (norm divides by B*B).  row_off / col_off:
 * device int32 [B+1] prefix sums of num_clips / num_caps.  `reduced` [B,B] and `arg` [B,B] are
 * kept by the caller for cmhse_groupwise_bwd. */
size_t cmhse_groupwise_workspace(int32_t n, int32_t B);
int cmhse_groupwise_fwd(const float* im, const float* s, int32_t n, int32_t D,
                        const int32_t* row_off, const int32_t* col_off, int32_t B, float margin,
                        int32_t max_violation, int32_t norm, float* loss, float* reduced,
                        int32_t* arg, float* scores_out, void* workspace, size_t workspace_bytes,
                        void* stream);
size_t cmhse_groupwise_bwd_workspace(int32_t n, int32_t B);
int cmhse_groupwise_bwd(const float* im, const float* s, int32_t n, int32_t D,
                        const int32_t* row_off, const int32_t* col_off, int32_t B, float margin,
                        int32_t max_violation, int32_t norm, const float* reduced,
                        const int32_t* arg, const float* grad_out, float* d_im, float* d_s,
                        void* workspace, size_t workspace_bytes, void* stream);

/* decoder/loss.py:17-26 EuclideanLoss: loss = mean_r (or sum_r) sqrt(sum_c (a[r,c] - b_r[c])^2).
 * a is [rows, cols] contiguous; row r of b is at b_rows[r] (device addresses) when b_rows != NULL,
 * else b + r*cols.  The backward gives d loss / d a (b is detached upstream, model.py:347,363). */
int cmhse_euclid_fwd(const float* a, const float* b, const uint64_t* b_rows, int32_t rows,
                     int32_t cols, int32_t norm, float* loss, float* row_dist /* [rows] scratch */,
                     void* stream);
int cmhse_euclid_bwd(const float* a, const float* b, const uint64_t* b_rows, int32_t rows,
                     int32_t cols, int32_t norm, const float* grad_out, float* d_a, void* stream);

/* Measurement aid (bench.py's roofline leg): a pair of HIP events owned by the handle.  A timer
 * passed in cmhse_seq_batch.step_timer is recorded before the first and after the last GRU step
 * kernel of that call; cmhse_timer_elapsed_ms waits for the stop event and returns the span. */
void* cmhse_timer_create(void);
void cmhse_timer_destroy(void* timer);
int cmhse_timer_elapsed_ms(void* timer, float* ms_host);
int32_t cmhse_timer_launches(void* timer);   /* step kernels launched inside the bracket */
/* The launches of the LDS-tiled step kernel inside the bracket, each between its own event pair:
 * summed duration, their algorithmic FLOPs (2*3H*(I+H) + 14H per sequence and step), their
 * algorithmic HBM bytes (4I + 8H per sequence and step, the weights once per launch) and count. */
int cmhse_timer_tiled(void* timer, float* ms_host, double* flops_host, double* bytes_host,
                      int32_t* launches_host);

/* The one configuration entry point: kernel-shape crossovers a caller (a test, a benchmark) may
 * move.  They change WHICH kernel shape serves a step, never a result beyond fp32 summation order
 * (and the shapes selected by mid_units / mid_waves are bit-identical to each other).  `value` >= 0
 * sets the tunable, < 0 only reads it; *old_value (if not NULL) receives the previous value.
 *   "tiny_max_seqs"     1024  active sequences at or below which a forward step runs on the
 *                             small-batch kernels instead of the LDS-tiled one
 *   "mid_max_seqs"      1024  ... at or below which it runs on the mid-size kernel (hoisted input
 *                             projection + split-K 16x16x4 tiles); 0 disables that kernel
 *   "mid_units"            0  16 | 8 | 4 forces the mid-size step's hidden units per workgroup
 *   "mid_waves"            0  4 | 8 forces its waves per workgroup
 *   "mid_tall_min_seqs"  129  active sequences from which the mid-size step of a training call
 *                             takes 64 sequences per workgroup instead of 32 (bit-identical)
 *   "tall_tile_min_wgs" 2048  64-row workgroups from which an LDS-tiled launch uses 128-row tiles
 *   "bwd_mid_max_seqs"   512  active sequences at or below which a BPTT step runs on the mid-size
 *                             backward kernel
 *   "bwd_split_min_seqs"  33  active sequences from which (up to bwd_mid_max_seqs) a BPTT step runs
 *                             as two launches with K split over the grid; 0 = never
 *   "bwd_tail_min_steps"   4  steps with at most 32 active sequences at the end of a chain from
 *                             which its BPTT runs them inside ONE resident kernel (a grid barrier
 *                             per step instead of a launch); 0 = never.  The kernel holds H / 16
 *                             workgroups resident (one per CU): at most four such chains at once
 *   "fwd_tail_min_steps"   4  the same for the forward chain of a training call
 *                             (CMHSE_SAVE_FOR_BACKWARD, job on its own stream)
 *   "resident_timeout_ms" 5000  wall time a grid barrier of a resident kernel / a dependency wait of
 *                             the step chain may take before the launch gives up (cmhse_async_status);
 *                             0 = give up at the second clock check of a wait (tests: walks the
 *                             abort path)
 *   "bwd_chunk_rows"    2048  packed rows a weight-gradient chunk spans before its products are
 *                             issued beside the chain (changes the order in which chunks are
 *                             accumulated, i.e. the gradients to fp32 rounding)
 *   "xproj_chunk_rows"  1536  packed rows per chunk of a training chain's hoisted input projection
 *                             beside the chain; 0 = one launch in front of it
 *   "chain_min_steps"      2  consecutive LDS-tiled steps of an inference call from which they run as
 *                             ONE launch (gru_step_chain_kernel: per-row-tile dependencies instead
 *                             of a launch per time step; bit-identical); 0 = never
 *   "early_xproj"          1  the hoisted input projection of an inference call's small-batch steps on
 *                             the call's side stream (tail_stream) before the first step, beside the
 *                             tiled steps; 0 = in order in front of those steps (results identical)
 *   "chain_tall_min_wgs" 256  64-row workgroups per step from which such a chain uses 128-row tiles
 *                             (four times that when a request of the chain has I < H)
 *   "tn_rows_bm"           0  tile height of the weight-gradient products: 128 | 192; 0 = 192 where
 *                             every product's row count (3H, H) is a whole number of them, else 128
 *                             (the row split into parts follows the tile count, i.e. the gradients
 *                             to fp32 rounding; profiles/r04_wgrad_rate.txt)
 *   "pull_waves"          32  single-wave workgroups of one cmhse_pull_steps launch
 *   "multi_step_off"       0  (not a crossover, and per DEVICE rather than per context) 1 while the
 *                             multi-step kernels are switched off on the calling thread's current device
 *                             after an acknowledged timeout (cmhse_async_status below); readable, and
 *                             settable either way
 * These are the PROCESS DEFAULTS (atomics): set them between calls, not while calls that size
 * workspaces with them (`*_workspace` reads mid_max_seqs) are in flight on other threads.  A caller
 * that wants its own values — two models tuned differently in one process, a library that must not
 * disturb its host's settings — uses a context instead (below).  Unknown name: CMHSE_ERR_ARG.
 * (Five opt-ins of rounds 4-5 that measured slower than the defaults — a resident BPTT chain, the
 * one-launch BPTT step, attention tiles as chain tasks, a second ticket map, a resident inference
 * tail — left the library in round 6; docs/HISTORY.md and profiles/r04_chain_resident.txt,
 * r05_bptt_one_launch.txt, r05_chain_attention.txt, r05_dual_column_tile.txt, r05_rank_share.txt keep
 * the measurements, the git history the code.) */
int cmhse_tune(const char* name, int32_t value, int32_t* old_value);

/* Tuning contexts (round 5; SURVEY 8b "re-entrant, no global state").  A context is a private copy
 * of the crossovers above, initialised from the process defaults at creation.  cmhse_ctx_enter(ctx)
 * makes it the CALLING THREAD's current context and returns the previous one (NULL = none): every
 * library call that thread makes until it enters another one (cmhse_ctx_enter(prev) to leave) —
 * workspace sizing and launches alike — reads its crossovers from the context, and cmhse_tune /
 * other threads / other contexts do not affect it.  The context must outlive the calls made inside
 * it; host-side only (a few hundred bytes).  cmhse_ctx_tune: cmhse_tune on a context. */
void* cmhse_ctx_create(void);
void cmhse_ctx_destroy(void* ctx);
int cmhse_ctx_tune(void* ctx, const char* name, int32_t value, int32_t* old_value);
void* cmhse_ctx_enter(void* ctx);

/* The kernels that stay resident over several time steps of a chain (the few-sequence tails of a
 * training chain: "*_tail_min_steps") synchronise their workgroups with a grid barrier, which needs
 * all of them on the chip at once; the step chain ("chain_min_steps") waits on per-row-tile counters
 * and needs its workgroups to start in index order.  The launchers check
 * the CU count; what they cannot see — another process holding CUs, a CU mask — ends in a timeout
 * ("resident_timeout_ms", default 5000): the kernel's workgroups all leave, the call's results are
 * invalid, and a word in pinned host memory is raised.  This function returns CMHSE_ERR_TIMEOUT
 * while that word is set for the calling thread's current device (CMHSE_OK otherwise) and clears
 * it when `clear` != 0; cmhse_gru_pool_fwd[_multi] / cmhse_gru_pool_bwd[_multi] check it on entry
 * and return CMHSE_ERR_TIMEOUT without launching.  Clearing a raised status also switches the
 * multi-step kernels off ON THAT DEVICE, for every tuning context and the process defaults alike
 * ("chain_min_steps", "*_tail_min_steps" read as 0 there: one launch per time step, which needs
 * neither co-residency nor in-order workgroup starts) — other devices keep them; setting one of those
 * knobs to a positive value (cmhse_tune / cmhse_ctx_tune) with the device current switches them back
 * on.  The status is asynchronous: it reflects launches that have RUN, so poll it after a stream
 * synchronisation (or once per step, one step late). */
int cmhse_async_status(int32_t clear);

/* Self-test of that barrier and of its failure path (tests): launches `workgroups` (<= 1024)
 * single-wave workgroups that each arrive at `rounds` grid barriers, every barrier expecting
 * `missing` more arrivals than there are workgroups.  workspace: 256 bytes, 256-byte aligned
 * (zeroed by the call); on completion uint32 word [2] = workgroups that passed every barrier, word
 * [3] = workgroups that left through the abort path.  missing = 0 never times out; missing > 0 always
 * does, after "resident_timeout_ms", and raises the status cmhse_async_status reports. */
int cmhse_selftest_grid_sync(void* workspace, int32_t workgroups, int32_t missing, int32_t rounds,
                             void* stream);

/* Text for an error code returned by the functions above (static storage). */
const char* cmhse_strerror(int code);

/* Library / target identification: "cmhse_hip <version> gfx950". */
const char* cmhse_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CMHSE_HIP_H_ */
