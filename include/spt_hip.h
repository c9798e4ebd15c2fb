/*
 * spt_hip.h -- C ABI of libspt_hip.so: SPT's PQ sparse-attention operators for
 * AMD MI355X (gfx950), hand-written HIP.
 *
 * This is the drop-in boundary for the hot path of ytgui/SPT-proto.  Each entry
 * point replaces one pybind11 export of the reference's `naive_gpt.ext`
 * (extension/entry.cpp:43-56).  The reference interface takes torch::Tensor; this
 * ABI takes what those tensors hold: raw DEVICE pointers, sizes and a HIP stream.
 * No allocation, no synchronisation and no global state inside any call, so calls
 * are safe from several threads/streams and can be captured into a hipGraph.
 *
 * Conventions
 *   - every pointer is device memory of the current device, contiguous, row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - float = IEEE fp32, indices = int32 (`index_t`, extension/common.h:59);
 *   - CSR: `indptr` [S+1] is shared by all B batches (batch stride 0), `indices` /
 *     values are [B, nnz] with nnz = indptr[S] (extension/sddmm.cpp:43-49);
 *   - return value: SPT_OK (0), a negative SPT_E* precondition code (the checks the
 *     reference makes with TORCH_CHECK, extension/common.h:13-21), or a positive
 *     hipError_t from the launch.  spt_strerror() names either.
 *   - outputs that the algorithm only partially covers are fully written by the
 *     kernels themselves (zeros included): callers pass uninitialised buffers.
 *   - `*_heads` arguments: a dense per-batch operand [B, S, E] may instead live inside the
 *     attention layers' [N, S, heads, E] tensor (batch b = n * heads + h, row stride
 *     heads * E); pass heads > 0 for that layout, 0 for plain [B, S, E].  This removes
 *     the transpose(1,2).contiguous() copies of naive_gpt/layers/sparse/attention.py:92-95.
 *     The head layout is implemented for d_head in {64, 128} with S * d_head * 4 <= 128 KiB
 *     (SPT_EUNSUP otherwise: callers then copy to [B, S, E] as the reference does).
 */
#ifndef SPT_HIP_H
#define SPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPT_OK 0
#define SPT_EINVAL (-1)   /* null pointer / non-positive size */
#define SPT_ESHAPE (-2)   /* shape precondition of the reference violated */
#define SPT_EUNSUP (-3)   /* combination the kernels do not implement */

/* ABI version; bump on any signature change. */
#define SPT_ABI_VERSION 39
int spt_abi_version(void);
const char *spt_strerror(int code);

/*
 * cdist_forward_cuda(query, table) -> [distance, indices]
 *   reference: extension/entry.cpp:7-9, extension/cdist.cu:185-250 (kernel :7-69)
 * query [M, NQ, D], table [M, C, D] -> distance [M, NQ, C] (may be NULL: the PQ
 * 'encode' mode of naive_gpt/layers/basic/quantizer.py:74-77 discards it),
 * indices [M, NQ].  L1 distance summed in fp32 in ascending i, argmin with strict
 * '<' in ascending c (bit-exact contract).  Requires D % 4 == 0, D <= 32.
 * (The reference additionally needs NQ % 16 == 0 and C % 16 == 0; not needed here.)
 */
int spt_cdist_forward(const float *query, const float *table, float *distance,
                      int32_t *indices, int n_subspaces, int n_queries,
                      int n_codewords, int d_code, void *stream);

/*
 * PQ 'encode' straight from the attention layout: z [batch, seq, heads, M * D] (the q / k
 * of naive_gpt/layers/sparse/attention.py:84-95) -> codes [batch * heads, seq, M], the
 * input of lookup.  Equivalent to the reference's transpose + PQBase.forward('encode')
 * (quantizer.py:44-48,64-77) without the three layout copies; same bit-exact
 * distance / argmin contract as spt_cdist_forward.
 */
int spt_pq_encode_heads(const float *z, const float *table, int32_t *codes,
                        int batch, int seq_length, int n_heads, int n_subspaces,
                        int n_codewords, int d_code, void *stream);
/* The same from bf16 storage: z holds raw bf16 patterns, widened exactly to fp32 before the
 * same distance / argmin arithmetic -- the codes equal those of spt_pq_encode_heads on the
 * widened values bit for bit. */
int spt_pq_encode_heads_bf16(const uint16_t *z, const float *table, int32_t *codes,
                             int batch, int seq_length, int n_heads, int n_subspaces,
                             int n_codewords, int d_code, void *stream);

/*
 * PQ codebook training loss, PQBase.forward(mode='train')[-1] of the reference
 * (naive_gpt/layers/basic/quantizer.py:80-111; armed every step by
 * script/4-sparse-tuning-0.py:71-78), as one forward and one backward pass over z:
 *   loss = mean((softmax(-log max(d, 1e-5)) . W - W[argmin d])^2) + mean((z - W[argmin d])^2)
 * with d the L1 distances of spt_cdist_forward.  z is any contiguous tensor of
 * n_vectors rows of M * D floats ([N,S,H,E] or [B,S,E] alike: the loss is a mean over all
 * sub-vectors); table [M, C, D]; loss / grad_loss are single device floats, grad_z has
 * the shape of z, grad_table the shape of table.  Supported: C == 16, D in {4, 8}, M a
 * power of two <= 32 (SPT_EUNSUP otherwise: callers compose the loss from spt_cdist_*).
 * workspace: spt_pq_loss_workspace_bytes() bytes, used by both passes.
 * backward, accumulate != 0: grad_z += the gradient (grad_z already holds what reached z over another
 * path -- the attention's grad_q / grad_k -- and no elementwise sum is needed).
 */
int64_t spt_pq_loss_workspace_bytes(int64_t n_vectors, int n_subspaces, int n_codewords,
                                    int d_code);
int spt_pq_loss_forward(const float *z, const float *table, float *loss, void *workspace,
                        int64_t n_vectors, int n_subspaces, int n_codewords, int d_code,
                        void *stream);
/* The same forward with the PQ codes as a by-product: z [batch, seq, heads, M * D] (the attention
 * layout) -> loss as above and codes [batch * heads, seq, M] -- bit for bit those of
 * spt_pq_encode_heads (the loss's argmin is the code): a training step that arms the loss needs no
 * encode pass. */
int spt_pq_loss_forward_codes(const float *z, const float *table, float *loss, void *workspace,
                              int32_t *codes, int batch, int seq_length, int n_heads,
                              int n_subspaces, int n_codewords, int d_code, void *stream);
int spt_pq_loss_backward(const float *z, const float *table, const float *grad_loss,
                         float *grad_z, float *grad_table, void *workspace,
                         int64_t n_vectors, int n_subspaces, int n_codewords, int d_code,
                         int accumulate, void *stream);
/* `parts` (1 .. 4) tensors back to back in z -- q and k of one attention, which the joint projection
 * writes into one buffer -- against the SAME table in ONE pass: loss = the SUM of the parts' losses
 * (each a mean over its own n_vectors = batch * seq * heads rows), codes [parts * batch * heads, seq, M],
 * grad_z / the accumulation target laid out like z, grad_table the sum over the parts (what the
 * reference's `loss_q + loss_k` of attention.py:98-104 differentiates to).  workspace:
 * spt_pq_loss_workspace_bytes(parts * n_vectors, ...) bytes. */
int spt_pq_loss_forward_codes_parts(const float *z, const float *table, float *loss, void *workspace,
                                    int32_t *codes, int parts, int batch, int seq_length, int n_heads,
                                    int n_subspaces, int n_codewords, int d_code, void *stream);
int spt_pq_loss_backward_parts(const float *z, const float *table, const float *grad_loss,
                               float *grad_z, float *grad_table, void *workspace,
                               int64_t n_vectors, int parts, int n_subspaces, int n_codewords,
                               int d_code, int accumulate, void *stream);

/*
 * cdist_backward_cuda(query, table, grad_output) -> [grad_query, grad_table]
 *   reference: extension/entry.cpp:11-14, extension/cdist.cu:252-333
 * grad_output [M, NQ, C] -> grad_query [M, NQ, D], grad_table [M, C, D].
 * `workspace` holds per-workgroup partial sums of grad_table:
 * spt_cdist_backward_workspace_bytes() bytes, uninitialised.
 */
int64_t spt_cdist_backward_workspace_bytes(int n_subspaces, int n_queries,
                                           int n_codewords, int d_code);
int spt_cdist_backward(const float *query, const float *table,
                       const float *grad_output, float *grad_query,
                       float *grad_table, void *workspace, int n_subspaces,
                       int n_queries, int n_codewords, int d_code, void *stream);

/*
 * lookup_forward_cuda(config, query, key) -> indices
 *   reference: extension/entry.cpp:16-19, extension/lookup.cu:87-174 (kernel :10-84)
 * query/key [B, S, M] PQ codes -> out [B, S, Z], Z = S / sparsity, where
 * `sparsity` is the reference's config.size(0) (lookup.cu:99).  Bit-exact with
 * the reference kernel's bucketed causal selection, including its cursor
 * saturation and zero padding.  Requires S % 16 == 0, S % sparsity == 0,
 * Z % 16 == 0 (lookup.cu:103-106), 4 <= M <= 16 (any Z: the reference only
 * instantiates M in {8,10,16} x Z in {32,64,128}, lookup.cu:113-169).
 */
int spt_lookup_forward(const int32_t *query, const int32_t *key, int32_t *out,
                       int batch_size, int seq_length, int n_subspaces,
                       int sparsity, void *stream);

/*
 * sddmm_forward_cuda(trans_lhs=false, trans_rhs=true, indptr, indices, query, key)
 *   reference: extension/entry.cpp:27-31, extension/sddmm.cpp:3-73 (cusparseSDDMM)
 * out[b, p] = sum_e query[b, row(p), e] * key[b, indices[b, p], e];
 * query/key [B, S, E], out [B, nnz].  Requires E % 4 == 0, E <= 256.
 * `scale`/`clamp`: out = clamp(scale * dot, -clamp, +clamp) when clamp > 0 (the
 * epilogue of naive_gpt/layers/sparse/attention.py:125-127); pass scale = 1,
 * clamp = 0 for the plain operator.
 *
 * Two forms (ABI 38).  Patterns as dense as lookup's (nnz >= S * S / 16; d_head 64, 128 <= S <= 1024, at
 * least 160 workgroups of 512 keys) are computed as DENSE 32-row stripes of scores on the matrix
 * cores (split-bf16, <= 2^-16 relative error per product), from which the CSR's entries are picked
 * (csrc/sddmm_tile.hip: 36 us against 51-56 at the configs[2] shape); everything else by the fp32
 * gather kernels (csrc/sddmm.hip).  spt_sddmm_form() tells which one a call would take
 * (1 = matrix cores, 0 = gather); the environment variable SPT_SDDMM_GATHER forces the gather form.
 */
int spt_sddmm_form(int batch_size, int seq_length, int d_head, int nnz);
int spt_sddmm_forward(const int32_t *indptr, const int32_t *indices,
                      const float *query, const float *key, float *out,
                      int batch_size, int seq_length, int d_head, int nnz,
                      float scale, float clamp, int query_heads, int key_heads,
                      void *stream);

/*
 * spmm_forward_cuda(trans_lhs, trans_rhs=false, indptr, indices, values, x)
 *   reference: extension/entry.cpp:21-25, extension/spmm.cpp:3-72 (cusparseSpMM)
 * trans_lhs == 0: y[b, r, :]             = sum_{p in row r} values[b,p] * x[b, indices[b,p], :]
 * trans_lhs != 0: y[b, indices[b,p], :] += values[b,p] * x[b, row(p), :]
 * x, y [B, S, E].  Requires E % 4 == 0, E <= 256.
 * The transposed product is computed as a gather over the transposed CSR structure,
 * which is built into `workspace` (spt_spmm_workspace_bytes() bytes, uninitialised;
 * may be NULL when trans_lhs == 0).
 *
 * Two forms of the non-transposed product (round 4).  The fp32 gather kernels (csrc/spmm.hip) are the
 * default.  With SPT_SPMM_MFMA=1 in the environment (read at every call), patterns as dense as lookup's
 * (nnz >= S * S / 16, mean row length <= 64, d_head 64 or 128, 64 <= S <= 2048) are multiplied as DENSE
 * 32 x 32 tiles on the matrix cores: a wave assembles the tile of its 32 rows in LDS from the CSR
 * entries (duplicates add) and contracts it with the bf16 image of x (split-bf16, <= 2^-16 relative
 * error per product; csrc/mfma_attention.hip: spmm_mfma_kernel).  Opt-in because it is SLOWER at the
 * benchmark shape (65.7 against 46.3 us, profiles/r04_ops_kernels.txt; the kernel's header says
 * where the time goes).  spt_spmm_form() tells which form a call would take (1 = matrix cores,
 * 0 = gather).
 */
int spt_spmm_form(int trans_lhs, int batch_size, int seq_length, int d_head, int nnz);
int64_t spt_spmm_workspace_bytes(int trans_lhs, int batch_size, int seq_length, int d_head,
                                 int nnz);
int spt_spmm_forward(int trans_lhs, const int32_t *indptr,
                     const int32_t *indices, const float *values,
                     const float *x, float *y, void *workspace, int batch_size,
                     int seq_length, int d_head, int nnz, int x_heads, int y_heads,
                     void *stream);

/*
 * The two halves of the transposed product, for callers that use one CSR pattern for
 * several products (the backward of naive_gpt/kernels/sddmm.py:25-51 and
 * spmm.py:23-49 needs A^T twice per attention layer):
 *   spt_csr_transpose   builds the transposed structure of (indptr, indices) into
 *                       `transposed` (spt_csr_transpose_workspace_bytes() bytes);
 *   spt_spmm_transposed y = A^T . x using that structure and the CSR-ordered `values`;
 *                       `workspace`: spt_spmm_transposed_workspace_bytes() bytes.
 * The structure is laid out for the product kernel of a given head size, so d_head is an
 * argument of all four and must be the same in the calls that share a buffer:
 *   flat form     (S * d_head * 4 <= 128 KiB, or d_head != 64): one transposed CSR per slice;
 *                 the workspace receives the values in transposed order (may be NULL: the
 *                 product then gathers them through the permutation);
 *   chunked form  (d_head == 64, S a multiple of 512 above 512): one transposed CSR per
 *                 512-row chunk of A, A^T x = sum_j A_j^T x_j with x_j in LDS; the workspace
 *                 holds the per-chunk partial outputs and is required.
 */
int64_t spt_csr_transpose_workspace_bytes(int batch_size, int seq_length, int nnz, int d_head);
int spt_csr_transpose(const int32_t *indptr, const int32_t *indices,
                      void *transposed, int batch_size, int seq_length, int nnz, int d_head,
                      void *stream);
int64_t spt_spmm_transposed_workspace_bytes(int batch_size, int seq_length, int d_head, int nnz);
int spt_spmm_transposed(const int32_t *indptr, const void *transposed, const float *values,
                        const float *x, float *y, void *workspace, int batch_size,
                        int seq_length, int d_head, int nnz, int x_heads,
                        int y_heads, void *stream);

/*
 * softmax_forward_cuda(indptr, indices, values) -> output
 *   reference: extension/entry.cpp:33-36, extension/softmax.cu:84-114 (kernel :7-47)
 * y = mask * exp(v) / max(1e-9, sum_row mask * exp(v)), mask = indices <= row.
 */
int spt_softmax_forward(const int32_t *indptr, const int32_t *indices,
                        const float *values, float *output, int batch_size,
                        int seq_length, int nnz, void *stream);

/*
 * softmax_backward_cuda(indptr, indices, output, grad_output) -> grad_values
 *   reference: extension/entry.cpp:38-41, extension/softmax.cu:116-148 (kernel :49-81)
 * c = max(1e-9, sum_row mask*y*dy); dv = mask * y * (dy - c).
 */
int spt_softmax_backward(const int32_t *indptr, const int32_t *indices,
                         const float *output, const float *grad_output,
                         float *grad_values, int batch_size, int seq_length,
                         int nnz, void *stream);

/*
 * softmax_backward chained through the attention layer's score epilogue
 * clamp(scale * raw, -clamp, clamp) (naive_gpt/layers/sparse/attention.py:125-127):
 * grad_scores = (|clamped_scores| < clamp) ? scale * softmax_backward(...) : 0, i.e. the
 * gradient wrt the RAW sddmm output, in one pass.
 */
int spt_softmax_backward_clamped(const int32_t *indptr, const int32_t *indices,
                                 const float *output, const float *grad_output,
                                 const float *clamped_scores, float scale, float clamp,
                                 float *grad_scores, int batch_size, int seq_length,
                                 int nnz, void *stream);

/*
 * The forward of the sparse attention core in one launch:
 *   scores = clamp(scale * sddmm(q, k), -clamp, clamp)      attention.py:119-127
 *   attn   = softmax(scores)  (masked, softmax.cu:7-47)      attention.py:128-130
 *   y      = spmm(attn, v)                                   attention.py:140
 * for UNIFORM CSR rows -- row r owns entries [r Z, (r+1) Z) of indices [B, nnz], Z = nnz / S,
 * which is what spt_lookup_forward produces -- so there is no indptr argument.
 * q, k, v: [batch, S, E] slices, or with heads > 0 the [N, S, heads, E] tensors themselves
 * (batch = N * heads).  scores and attn [batch, nnz] are both outputs (the backward needs the
 * clamp mask and the softmax VJP).  y is [batch, S, E], or with y_transposed != 0
 * [batch, E, S]: the memory layout of the reference's `y.transpose(1, 2).contiguous()`
 * (attention.py:141), saving that copy.
 * causal != 0 promises that every column id is <= its row (true for lookup's output): the
 * K / V slices are then staged 64 rows at a time just ahead of the rows that need them,
 * instead of whole before the first row; `scores` of entries with column > row are
 * unspecified in that mode (their probability is 0 either way).
 * Supported: d_head == 64, Z <= 64, Z % 4 == 0, S * 256 <= 128 KiB (SPT_EUNSUP otherwise:
 * use the separate operators).  Same arithmetic as the separate operators.
 */
int spt_sparse_attention_forward(const int32_t *indices, const float *q, const float *k,
                                 const float *v, float *scores, float *attn, float *y,
                                 int batch_size, int seq_length, int d_head, int nnz,
                                 float scale, float clamp, int heads, int y_transposed,
                                 int causal, void *stream);

/*
 * The row-wise half of the backward of spt_sparse_attention_forward in one launch:
 *   dP       = sddmm(grad_y, v)                          (kernels/spmm.py:25-40)
 *   grad_raw = clamp-mask(scale * softmax_backward(attn, dP))   -- wrt the RAW sddmm scores
 *              (softmax.cu:49-81 with its 1e-9 clamp, then attention.py:125-127)
 *   grad_q   = spmm(grad_raw, k)                         (kernels/sddmm.py:25-41)
 * grad_y is [batch, S, E], or with grad_y_transposed != 0 the [batch, E, S] memory that the
 * transposed forward output hands back; in that case the kernel also writes grad_y_rows
 * [batch, S, E] (may be NULL), the operand the grad_v product wants.  v, k, grad_q follow
 * `heads` as in the forward; scores / attn are the forward's outputs.  grad_k and grad_v are
 * the two spt_spmm_transposed calls on (grad_raw, q) and (attn, grad_y_rows).
 * Same shape limits and `causal` contract as the forward.
 */
int spt_sparse_attention_backward_rows(const int32_t *indices, const float *grad_y,
                                       const float *v, const float *k, const float *scores,
                                       const float *attn, float *grad_raw, float *grad_q,
                                       float *grad_y_rows, int batch_size, int seq_length,
                                       int d_head, int nnz, float scale, float clamp, int heads,
                                       int grad_y_transposed, int causal, void *stream);

/*
 * The same attention core on the matrix cores (mfma_attention.hip): the CSR rows select the
 * live cells of dense 32 x 32 score tiles computed with v_mfma_f32_32x32x16_bf16 on fp32
 * operands split into two bf16 halves (three MFMAs per product, fp32 accumulation, relative
 * error <= 2^-16 per product).  Nothing of size nnz is written: the backward recomputes the
 * tiles from q, k, v and the [batch, S] row sums.
 *   y[i]       = sum_p exp(s_p) v[col_p] / max(1e-9, row_sum[i]),
 *   s_p        = clamp(scale * q[i].k[col_p]),  entries with col_p > i take no part
 *   row_sum[i] = sum_p exp(s_p)                 (softmax.cu:17-30)
 * Any uniform-row CSR with Z = nnz / S <= 256, Z % 4 == 0, d_head 64 or 128, S <= 2048;
 * repeated columns count with their exact multiplicity (a row that holds ONE column Z = 256
 * times -- row 0 of a lookup pattern -- is kept as a byte count of 255 plus a per-row flag in
 * `tiles`; ABI 39: its row_sum is exact, no longer 255 / 256 of it).  SPT_EUNSUP otherwise.
 * Layouts of q, k, v (heads) and y (y_transposed) as in spt_sparse_attention_forward.
 *
 * ABI 39, `bounds`: the clamp of attention.py:125-127 passes gradient only inside (-clamp, clamp),
 * a DISCONTINUOUS function of the score, and a score formed from split operands can land on the
 * other side of the clamp than the fp32 score.  The forward therefore leaves slice-wide bounds on
 * the row norms of q and k in `bounds` (spt_attention_mfma_bounds_floats(batch) floats, device
 * memory, no initialisation needed), with which the backward finds the cells whose score lies
 * within the split's error of +-clamp and recomputes those exactly (fp64 dot, rounded to fp32,
 * times scale: the oracle's arithmetic, oracle/spt_oracle.c sddmm) -- the gradient mask is then
 * that of the fp32 scores.  `bounds` may be NULL in either call: the forward then skips the
 * norms and the backward decides the mask on its split-bf16 scores as before ABI 39.
 */
int spt_attention_mfma_supported(int seq_length, int d_head, int nnz);
/*
 * spt_attention_mfma_prepare buckets the CSR entries of every 32-row tile by 32-key tile (one
 * pass over `indices` [batch, nnz]) into `tiles`, an opaque device buffer of
 * spt_attention_mfma_tiles_bytes(...) bytes that the forward and the backward of the same
 * layer step share (it depends on `indices` only; layout: mfma_attention.hip, "cell tiles").
 * Two layouts, the same one to be named in all four calls:
 *   SPT_TILES_FULL     any uniform-row CSR;
 *   SPT_TILES_COMPACT  a quarter of the memory (S = 512: 66 instead of 272 KiB per slice) for
 *                      patterns in which only columns < 32 may repeat inside a row -- every
 *                      lookup pattern: its one repeated column is the padding column 0
 *                      (lookup.cu:107-109).  A pattern that breaks the promise is computed as
 *                      if each repeated column outside key tile 0 occurred once, and bit 0 of
 *                      the second 32-bit word of `tiles` is set.
 */
enum { SPT_TILES_FULL = 0, SPT_TILES_COMPACT = 1 };
int64_t spt_attention_mfma_tiles_bytes(int batch_size, int seq_length, int nnz, int layout);
int spt_attention_mfma_prepare(const int32_t *indices, void *tiles, int batch_size,
                               int seq_length, int nnz, int layout, void *stream);
int spt_attention_mfma_bounds_floats(int batch_size);
int spt_attention_mfma_forward(const void *tiles, int layout, const float *q, const float *k,
                               const float *v, float *y, float *row_sum, float *bounds,
                               int batch_size, int seq_length, int d_head, int nnz, float scale,
                               float clamp, int heads, int y_transposed, void *stream);

/*
 * The whole backward of spt_attention_mfma_forward: two launches, nothing of size nnz.
 *   delta[i]  = max(1e-9, grad_y[i] . y[i])            (= sum_p P_p dP_p, softmax.cu:69)
 *   dS_p      = scale * P_p * (dP_p - delta[i]) where |s_p| < clamp, else 0
 *   grad_q[i] = sum_p dS_p k[col_p];  grad_k[j] = sum_{p: col_p = j} dS_p q[row_p];
 *   grad_v[j] = sum_{p: col_p = j} P_p grad_y[row_p]
 * y, grad_y: [batch, S, E], or with transposed != 0 both [batch, E, S] (S % 4 == 0).
 * q, k, v and the three gradients follow `heads` as in the forward.  row_sum and bounds are the
 * forward's outputs for the same q, k (bounds may be NULL, see above); delta is scratch of
 * 2 * batch * S floats that the first launch fills for the second (the rows' 1 / row_sum, then
 * delta / row_sum).
 */
int spt_attention_mfma_backward(const void *tiles, int layout, const float *q, const float *k,
                                const float *v, const float *y, const float *grad_y,
                                const float *row_sum, const float *bounds, float *delta,
                                float *grad_q, float *grad_k, float *grad_v, int batch_size,
                                int seq_length, int d_head, int nnz, float scale, float clamp,
                                int heads, int transposed, void *stream);

/*
 * bf16 STORAGE variants (BASELINE configs[1] "bf16"; no reference counterpart -- the reference is
 * fp32 only): q, k, v, y, grad_y and the three gradients hold raw bf16 patterns in the same
 * layouts; row_sum and delta stay fp32; `tiles` is the same object.  A stored value is its own
 * matrix-core operand (no split, one MFMA where the fp32 path needs two or three); everything
 * computed on the way (scaled q, probabilities, dS, dY / row_sum) is fp32 split in two as in the
 * fp32 path, so the results are those of spt_attention_mfma_forward / _backward on the widened
 * inputs (to the split's 2^-16) rounded once, to nearest even, when stored.
 */
int spt_attention_mfma_forward_bf16(const void *tiles, int layout, const uint16_t *q,
                                    const uint16_t *k, const uint16_t *v, uint16_t *y,
                                    float *row_sum, float *bounds, int batch_size, int seq_length,
                                    int d_head, int nnz, float scale, float clamp, int heads,
                                    int y_transposed, void *stream);
int spt_attention_mfma_backward_bf16(const void *tiles, int layout, const uint16_t *q,
                                     const uint16_t *k, const uint16_t *v, const uint16_t *y,
                                     const uint16_t *grad_y, const float *row_sum,
                                     const float *bounds, float *delta,
                                     uint16_t *grad_q, uint16_t *grad_k, uint16_t *grad_v,
                                     int batch_size, int seq_length, int d_head, int nnz,
                                     float scale, float clamp, int heads, int transposed,
                                     void *stream);

/*
 * Routed FFN: token-bucketed grouped GEMM on the fp32 matrix cores.
 *   reference: the per-block loop of naive_gpt/layers/tuning/lora_ffn.py:87-111 and
 *   naive_gpt/layers/sparse/feedforward.py:66-85 (boolean-mask gather + cuBLAS per block;
 *   there is no native routed-FFN kernel in the reference).
 * Rows [offsets[g], offsets[g+1]) of the (block-sorted) row space belong to block g:
 *     out[p, n] = rowscale[p] * ( sum_k a[src(p), k] * W_g(n, k) + bias[g, n] )
 *     src(p) = gather ? gather[p] : p
 *     W_g(n, k) = w[g * w_group_stride + n * w_ldn + k * w_ldk]   (w_ldk == 1 or w_ldn == 1)
 * a [*, lda] (lda >= k), out [n_rows, n]; gather / bias [n_groups, n] / rowscale may be
 * NULL.  offsets is DEVICE memory (bucket sizes never visit the host).  Requires
 * k % 4 == 0, lda % 4 == 0 and 16-byte aligned weight rows.
 */
int spt_grouped_gemm(const float *a, const int32_t *gather, const float *w,
                     const float *bias, const float *rowscale,
                     const int32_t *offsets, float *out, int n_rows, int k, int n,
                     int n_groups, int lda, long long w_group_stride, int w_ldn,
                     int w_ldk, void *stream);

/*
 * The same product with the rest of a routed-FFN layer folded in (the torch glue around
 * the four block GEMMs cost as much as the GEMMs: DESIGN.md section 7):
 *
 *   v[p, n]  = rowscale[p] * ( sum_k a[src(p), k] * W_g(n, k) + bias[g, n] )
 *            + sum_{j < r} a2[src2(p), j] * b2[g * b2_group_stride + n * b2_ldn + j]
 *   epilogue SPT_EPI_PLAIN: out = v
 *            SPT_EPI_ACT  : out = act(v); out2 = v when out2 != NULL
 *            SPT_EPI_DACT : out = v * act'(s);
 *                           pdot_main[p, c] = sum_{n in c} v[p, n] * h[p, n]
 *                           pdot_act[p, c]  = sum_{n in c} out[p, n] * s[p, n]
 *              with c the 64-column half tile of n (pdot_ld >=
 *              spt_grouped_gemm_pdot_width(n) of them per row; the caller adds them up),
 *              s = s_in, h = act(s_in) -- or, when s_in == NULL (ReLU only), s = h = h_in.
 *   The kernel multiplies its accumulators by rowscale after the k-loop and runs the second
 *   term as one more k-step on top: nothing is ever divided by rowscale (0 is a legal router
 *   coefficient: 2 sigmoid(logit) underflows below logit -104).
 * The second term is the LoRA side path (lora_ffn.py:97-100,108-110): r <= 32 (SPT_EPI_PLAIN
 * through the fp32 operands: r <= 64), r % 4 == 0.
 * The pdot rows are the two inner products the gradient of the router coefficient needs,
 * <v, h> and <dS, s>: with c = rowscale, v = c (dY W2_g^T) + E and s = c (x W1_g^T + b1_g) + F,
 * d c = <dY W2_g^T, h> + <dS, x W1_g^T + b1_g> = (<v, h> - <E, h> + <dS, s> - <dS, F>) / c, and
 * the caller has <E, h> and <dS, F> as [*, r] dots from the LoRA side products.  Each would
 * otherwise cost a pass over [P, n].
 * activation: 0 ReLU, 1 GELU (erf), 2 SiLU.
 */
enum { SPT_EPI_PLAIN = 0, SPT_EPI_ACT = 1, SPT_EPI_DACT = 2 };
enum { SPT_ACT_RELU = 0, SPT_ACT_GELU = 1, SPT_ACT_SILU = 2 };
typedef struct SptGroupedGemm {
    const float *a;
    const int32_t *gather;
    const float *w;
    const float *bias;
    const float *rowscale;
    const int32_t *offsets;
    float *out;
    int32_t n_rows, k, n, n_groups, lda;
    int64_t w_group_stride;
    int32_t w_ldn, w_ldk;
    const float *a2;           /* NULL: no second term */
    const int32_t *gather2;
    const float *b2;
    int32_t lda2, r;
    int64_t b2_group_stride;
    int32_t b2_ldn;
    int32_t epilogue, activation;
    float *out2;
    const float *h_in;
    const float *s_in;
    float *pdot_main;
    float *pdot_act;
    int32_t pdot_ld;
    /* Pre-split operands (spt_split_bf16 below), both or neither: the image of the WHOLE
     * `a` matrix [*, k] (rows are addressed through `gather` exactly as rows of `a`) and of
     * the WHOLE weight `w` as it lies in memory (rows of w_ldn elements when w_ldk == 1, of
     * w_ldk elements when w_ldn == 1; the group offsets g * w_group_stride are resolved
     * inside it).  With images the operands go global -> LDS by LDS-DMA and the k-loop has
     * no conversion work; taken when k % 32 == 0 and the weight's row length and
     * w_group_stride are multiples of 32 -- otherwise the fp32 operands are used, which
     * therefore stay mandatory unless the caller knows the image path applies. */
    const void *a_image;
    const void *w_image;
    /* SPT_EPI_ACT with ReLU only (mandatory there): Euclidean norms of the rows of `a`
     * [rows of a] and of the vectors W_g(n, :) [n_groups, n] (upper bounds do).  A product of split
     * operands is off by <= 2^-16 |a| |w|; a pre-activation within that of ZERO would get
     * the wrong ReLU derivative -- an O(1) error in that token's gradients (measured: 6 of
     * 614 k elements of one FFN).  The epilogue finds those elements (4e-4 of them at
     * k = 1024) from the norms and recomputes each as a plain fp32 dot product, read from
     * the fp32 operands `a` and `w`, which therefore have to be given as well. */
    const float *a_norm;
    const float *w_norm;
    /* ... and, optionally, scratch for the queue of those elements: 16 KiB of counters +
     * 8 bytes per element (256 segments; n_rows * n / 64 elements are 25 x the expected
     * number).  With it the GEMM only queues them and a second launch recomputes all of them in
     * parallel; without it, or where a segment overflows, the epilogue recomputes them itself,
     * one after another (~2 us each).  16-byte aligned; contents on return are unspecified. */
    void *relu_queue;
    int64_t relu_queue_bytes;
    int64_t ldo;               /* row stride (floats) of out, out2, h_in, s_in; 0: n.  A padded
                                  stride keeps the rows of a ragged n (a vocabulary of 30522)
                                  16-byte aligned for the kernels that read them next */
    int32_t accumulate;        /* != 0 (SPT_EPI_PLAIN only): out += the product instead of out = --
                                  the sum of several products (dX of q / k / v) without a pass of its own */
    /* K in segments (0: none; SPT_EPI_PLAIN, fp32 operands only): the contraction walks k / a_seg_k
     * matrices [*, a_seg_k] that lie a_seg_stride floats apart,
     *   A(row, kk) = a[(kk / a_seg_k) * a_seg_stride + src(row) * lda + kk % a_seg_k],
     * so dX = dQ Wq + dK Wk + dV Wv is ONE product with k = 3 n against the three weights stacked
     * (a_seg_k % 32 == 0, k % a_seg_k == 0, a_seg_stride % 4 == 0, lda >= a_seg_k).  With it (and
     * SPT_EPI_PLAIN) the second term may be r <= 64 wide: the three rank-16 products side by side. */
    int32_t a_seg_k;
    int64_t a_seg_stride;
    /* != 0 (SPT_EPI_PLAIN, fp32 operands, r % 16 == 0): the second term's b2 in segments of 16 columns,
     *   b2(n, j) = b2[(j / 16) * b2_seg_stride + n * b2_ldn + j % 16]
     * -- rank-16 tables that lie b2_seg_stride floats apart (the adapters of q, k, v in a tuner's flat
     * parameter buffer) used as one [n, r] operand without a concatenated copy. */
    int64_t b2_seg_stride;
} SptGroupedGemm;
int spt_grouped_gemm_fused(const SptGroupedGemm *desc, void *stream);
int spt_grouped_gemm_pdot_width(int n);
/* How spt_grouped_gemm_fused would run `desc`: 1 = both operands from their images (a_image and
 * w_image given, k % 32 == 0); 2 = "A32": the weight from w_image, the activation from its fp32
 * rows `a` (a_image NULL; a 16-byte aligned, lda % 4 == 0), moved global -> LDS by LDS-DMA and split
 * into its bf16 parts inside the kernel -- no image of the activation is needed at all;
 * 0 = from the fp32 operands through registers (k % 32 != 0, or no w_image). */
int spt_grouped_gemm_image_path(const SptGroupedGemm *desc);

/*
 * The pre-split image of an fp32 matrix [rows, cols] (leading dimension ld, ld % 4 == 0,
 * 16-byte aligned): every element x as hi = bf16(x) (round to nearest even) and
 * lo = bf16(x - hi), laid out [row][ceil(cols / 32)][hi: 32 x bf16 | lo: 32 x bf16], i.e. one
 * 128-byte block per row and 32 columns (columns past `cols` in the last block are zero).
 * x = hi + lo to 2^-17 relative; three bf16 MFMAs (lo.hi + hi.lo + hi.hi) then give an fp32
 * product to 2^-16.  spt_split_bf16_bytes = rows * ceil(cols / 32) * 128.
 * No reference counterpart: the reference multiplies in fp32 on cuBLAS (lora_ffn.py:87-111).
 */
size_t spt_split_bf16_bytes(long long rows, int cols);
int spt_split_bf16(const float *src, void *image, long long rows, int cols, long long ld,
                   void *stream);

/*
 * The rank-r down product of the LoRA adapters (naive_gpt/layers/tuning/lora.py:70-80,
 * lora_ffn.py:87-111) as a single pass over the tall activation (lora_side.hip):
 *   spt_lora_down  u[rows, n] = x[rows, k] . l[k, n]   (x row stride ldx floats; l contiguous; u
 *                  with row stride ldu floats, 0 = n: a column slice of a wider matrix;
 *                  u_block_major != 0: u is n / 16 matrices [rows, 16] one after the other --
 *                  several adapters' tables side by side in l, each adapter's u contiguous);
 *                  optional by-products of the same read: `image` (spt_split_bf16's layout,
 *                  spt_split_bf16_bytes(rows, k) bytes) and `norms` [rows] (row 2-norms) -- NULL
 *                  to skip.  k % 32 == 0, n in {16, 32, 48, 64}: SPT_EUNSUP otherwise (callers
 *                  then use a library GEMM).
 * exact == 0: split-bf16 matrix-core products as in spt_grouped_gemm, <= 2^-16 relative per
 * product.  exact != 0: u in exact fp32 (the fp32-input MFMA, an fp32 FMA chain per element):
 * for the u that feeds the GEMM in front of a ReLU (lora_ffn.py:99-101), where a 2^-16 error
 * would decide the sign of pre-activations next to zero; 2-3 x the time at 48-64 columns.
 */
int spt_lora_down(const float *x, long long ldx, long long rows, int k, const float *l, int n,
                  float *u, long long ldu, int u_block_major, void *image, float *norms, int exact,
                  void *stream);
/* spt_lora_down_grouped with the groups' results side by side: row r of group g goes to
 * u[(r - offsets[g]) * ldu + g * n + c] -- for groups of equal size one matrix [rows / n_groups, n_groups * n]
 * (ldu >= n_groups * n): dU_q | dU_k | dU_v of a joint projection's backward from ONE launch over the
 * stacked gradients [dQ; dK; dV] (autograd of lora.py:70-80, three times). */
int spt_lora_down_grouped_cols(const float *x, long long ldx, long long rows, int k, const float *l,
                               long long l_group_stride, int n, const int32_t *offsets, int n_groups,
                               float *u, long long ldu, void *stream);
/* spt_lora_down for n_tables (<= 4) SEPARATE tables [k, 16] lying table_stride floats apart (table t at
 * l + t * table_stride): u as n_tables block-major matrices [rows, 16], exactly what one table
 * [k, 16 n_tables] of the concatenated columns gives -- without making that copy every step. */
int spt_lora_down_tables(const float *x, long long ldx, long long rows, int k, const float *l,
                         int n_tables, long long table_stride, float *u, void *image, float *norms,
                         int exact, void *stream);
/* spt_lora_down with a SECOND table: u gets one more block of 16 columns = x . l2^T for a row-major
 * l2 [n2 <= 16, k] (an nn.Linear weight as stored; 16-byte aligned); columns n2 .. 15 of that block are 0.
 * Exact form only (exact != 0; SPT_EUNSUP otherwise).  `n` counts the columns of `l` only (n + 16 <= 64); u_block_major as above: (n / 16 + 1) matrices
 * [rows, 16].  The routed FFN's router logits ride the pass that forms x . L1 (feedforward.py:22-25). */
int spt_lora_down2(const float *x, long long ldx, long long rows, int k, const float *l, int n,
                   const float *l2, int n2, float *u, long long ldu, int u_block_major, void *image,
                   float *norms, int exact, void *stream);
/*
 * The per-block tables of the routed FFN (lora_ffn.py:87-111: `h_i @ l2[i]`, and `ds_i @ r1[i]` in
 * its backward): rows offsets[g] .. offsets[g + 1] - 1 (device int32 [n_groups + 1], rows sorted by
 * block, offsets[n_groups] == rows) use the table l + g * l_group_stride.  u [rows, n] with row
 * stride ldu (0 = n); `image` / `norms` as above.  n_groups <= 64.
 */
int spt_lora_down_grouped(const float *x, long long ldx, long long rows, int k, const float *l,
                          long long l_group_stride, int n, const int32_t *offsets, int n_groups,
                          float *u, long long ldu, void *image, float *norms, void *stream);

/*
 * Softmax cross-entropy of the language-model head (script/4-sparse-tuning-0.py:45-59:
 * nn.CrossEntropyLoss() on logits [rows, n_classes]), loss and gradient in one pass, IN PLACE:
 *   loss[i]      = log(sum_j exp(z[i, j])) - z[i, target[i]]           (0 for an ignored row)
 *   z[i, j]     <- (softmax(z[i, :])[j] - [j == target[i]]) * *scale   (0 for an ignored row)
 * z has row stride ld >= n_classes (ld % 4 == 0, 16-byte aligned); columns n_classes .. ld-1 are
 * written as zeros.  target: int64 [rows]; a row whose target equals ignore_index (or lies
 * outside [0, n_classes)) is ignored.  scale: one DEVICE float (1 / number of counted rows for
 * the mean reduction); loss: [rows] -- the mean is sum(loss) * scale.
 */
int spt_cross_entropy_grad(float *logits, long long ld, long long rows, int n_classes,
                           const long long *target, const float *scale, float *loss,
                           long long ignore_index, void *stream);

/*
 * LayerNorm of the residual stream with the additions around it (the pre-norm wiring of
 * naive_gpt/layers/basic/transformer.py:46-52; nn.LayerNorm arithmetic: biased variance, eps
 * inside the root), or -- rms != 0 -- LLaMA's RMSNorm (utils.py:22-37: no mean, no beta; beta may be
 * NULL, mean is written as 0, dbeta as 0).  rows x d fp32, contiguous; d in {256, 512, 1024, 2048},
 * RMSNorm also 4096 (SPT_EUNSUP else).
 *   forward   s = x + r (r, s may be NULL: s = x, nothing written);  y = LN(s) gamma + beta;
 *             mean, rstd [rows] for the backward.
 *   backward  dx = dLN(dy) (+ dskip, may be NULL: the gradient arriving over the skip path);
 *             dgamma, dbeta = ONE buffer [2, d] (dbeta == dgamma + d); `partial`:
 *             spt_layernorm_partial_rows(rows) x 2 d floats of scratch.  Fixed summation order.
 */
int spt_layernorm_partial_rows(long long rows);
int spt_add_layernorm_forward(const float *x, const float *r, const float *gamma, const float *beta,
                              float *s, float *y, float *mean, float *rstd, long long rows, int d,
                              float eps, int rms, void *stream);
int spt_layernorm_backward(const float *s, const float *dy, const float *gamma, const float *mean,
                           const float *rstd, const float *dskip, float *dx, float *dgamma,
                           float *dbeta, float *partial, long long rows, int d, int rms, void *stream);

/*
 * The LoRA table gradients (autograd of lora.py:70-80 and of the per-block side products of
 * lora_ffn.py:87-111): out = wide^T . narrow for a tall activation (or activation gradient)
 * wide [rows, width] and the rank-sized side narrow [*, n], n = 4, 16 or 48,
 *   out[g][w][j] = sum over the rows r of group g of wide[r, w] * narrow[gather ? gather[r] : r, j]
 * plain fp32 FMAs (exact products), fixed summation order (chunks of 64 rows, then the chunks in
 * order: reproducible).  offsets == NULL: one group of all rows (n_groups must be 1); else
 * offsets [n_groups + 1] int32 on the device, rows offsets[g] .. offsets[g + 1] - 1 form group g
 * (rows sorted by block: the routed FFN's row space).  transposed: 0 out[g][w][j], 1 out[g][j][w],
 * 2 out[g][j / 16][w][j % 16] (n / 16 rank-16 tables side by side in narrow: each its own matrix).
 * out: n_groups * width * n floats, written whole.  workspace: spt_tall_tn_workspace_bytes(rows,
 * n_groups, width, n) bytes of device scratch.  width, ldw even; wide 8-byte aligned; ldn a
 * multiple of 4 and narrow 16-byte aligned.
 */
long long spt_tall_tn_workspace_bytes(long long rows, int n_groups, int width, int n);
int spt_tall_tn(const float *wide, long long ldw, const float *narrow, long long ldn,
                const int32_t *gather, const int32_t *offsets, int n_groups, long long rows,
                int width, int n, float *out, int transposed, void *workspace, void *stream);
/* `count` (1 .. 4) products of ONE shape, ONE pair of leading dimensions and ONE (gather, offsets)
 * in a single pair of launches: outs[i] = the spt_tall_tn result of (wides[i], narrows[i]) (host
 * arrays of `count` device pointers; the two table gradients of a LoRA linear, the three `right`
 * gradients of q / k / v, the two per-block tables of a routed FFN).  workspace:
 * count * spt_tall_tn_workspace_bytes(rows, n_groups, width, n) bytes. */
int spt_tall_tn_batch(int count, const float *const *wides, long long ldw, const float *const *narrows,
                      long long ldn, const int32_t *gather, const int32_t *offsets, int n_groups,
                      long long rows, int width, int n, float *const *outs, int transposed,
                      void *workspace, void *stream);

/*
 * Bucketing for the routed FFN: the k largest of the n_blocks router probabilities of every
 * token (feedforward.py:56-64: `topk(prob, k)`), as (token, block) pairs sorted by block and,
 * inside a block, by token -- the order of the reference's `x[mask]` (lora_ffn.py:93-95):
 *   token[p], block[p]   the p-th pair;           offsets[g]  first pair of block g
 *   pos[t * k + j]       the row of token t's j-th selected block (ascending block id)
 * prob [n_tokens, n_blocks] fp32; token / block [n_tokens * k], offsets [n_blocks + 1],
 * pos [n_tokens * k] int32.  Ties go to the lower block index.  One workgroup:
 * n_blocks <= 8 and n_tokens <= 65536 (SPT_EUNSUP otherwise).
 */
int spt_route_topk(const float *prob, int32_t *token, int32_t *block, int32_t *offsets,
                   int32_t *pos, int n_tokens, int n_blocks, int k, void *stream);

/*
 * The same launch with the by-products the routed LoRA FFN needs next (lora_ffn.py:87-98 reads
 * them off `topk` / boolean masks): token64 / block64 = token / block as int64 (torch's gather /
 * scatter index type), coeff[p] = scale * prob[token[p], block[p]] (the row's router coefficient;
 * the reference's `2.0 * prob` is scale = 2).  All three required.
 * spt_route_coeff_backward: the adjoint of `coeff`,
 *   dprob[t, g] = scale * dcoeff[p] if row p is token t's selection of block g, else 0
 * (dprob [n_tokens, n_blocks] is written whole).
 * spt_route_logit_backward: the same chained through the router's sigmoid (feedforward.py:22-25:
 * nn.Sequential(nn.Linear, nn.Sigmoid)): dlogit[t, g] = dprob[t, g] * (1 - prob[t, g]) * prob[t, g]
 * with prob [n_tokens, n_blocks] the sigmoid's outputs (torch's sigmoid backward).
 */
int spt_route_topk_coeff(const float *prob, int32_t *token, int32_t *block, int32_t *offsets,
                         int32_t *pos, long long *token64, long long *block64, float *coeff,
                         float scale, int n_tokens, int n_blocks, int k, void *stream);
/* The same from the router's LOGITS: prob[t, g] = sigmoid(logits[t * ld + g] + bias[g]) (the Linear +
 * Sigmoid of feedforward.py:22-25; `logits` = x . W_router^T without bias, e.g. the second-table block
 * of spt_lora_down2; 16-byte aligned, ld >= n_blocks) is formed inside the routing launch and written
 * to prob [n_tokens, n_blocks] -- no separate sigmoid pass, no library GEMM with 4 output columns. */
int spt_route_topk_logits(const float *logits, int ld, const float *bias, float *prob, int32_t *token,
                          int32_t *block, int32_t *offsets, int32_t *pos, long long *token64,
                          long long *block64, float *coeff, float scale, int n_tokens, int n_blocks,
                          int k, void *stream);
int spt_route_coeff_backward(const float *dcoeff, const int32_t *pos, const int32_t *block,
                             float scale, float *dprob, int n_tokens, int n_blocks, int k,
                             void *stream);
int spt_route_logit_backward(const float *dcoeff, const int32_t *pos, const int32_t *block,
                             float scale, const float *prob, float *dlogit, int n_tokens,
                             int n_blocks, int k, void *stream);

/*
 * Router-coefficient gradient of the routed LoRA FFN's backward (layers/sparse/grouped.py; the
 * reference differentiates c * (x W^T + b) by autograd, lora_ffn.py:96-98):
 *   out[p] = (sum_j dot_main[p, j] + sum_j dot_act[p, j] - <du[p], u[token[p]]> - <dzt[token[p]], z[p]>)
 *            / max(coeff[p], floor_value)
 * dot_main, dot_act [n_rows, width]: SptGroupedGemm.pdot_main / pdot_act of the EPI_DACT GEMM;
 * du, z [n_rows, rank]; u, dzt [tokens, rank]; rank % 4 == 0.
 */
int spt_ffn_coeff_grad(const float *dot_main, const float *dot_act, int width, const float *du,
                       const float *u, const float *dzt, const float *z, const int32_t *token,
                       const float *coeff, float floor_value, float *out, int n_rows, int rank,
                       void *stream);

/*
 * Rotary position embedding (position.py:24-34: y = cos[s] * x + sin[s] * rotate_half(x)) of up to
 * three tensors [batch, seq, heads, d_head] in one launch: x[0 .. n_rot - 1] are rotated, x[n_rot ..
 * n_parts - 1] copied, part p written to out + p * out_stride (>= batch * seq * heads * d_head floats).
 * cos_table / sin_table [>= seq, d_head]: row s = position s.  transpose != 0: the adjoint (the
 * backward).  Forward use: (q, k) -> one buffer [q'; k']; backward: (dq', dk', dv) -> one buffer of
 * three equally spaced gradients for the joint projection's backward (a_seg_k).  d_head % 8 == 0, all
 * pointers 16-byte aligned; `x` is a host array of n_parts device pointers.
 */
int spt_rotary(const float *const *x, int n_parts, int n_rot, float *out, long long out_stride,
               const float *cos_table, const float *sin_table, int batch, int seq_length, int n_heads,
               int d_head, int transpose, void *stream);

/*
 * The gated FFN's elementwise middle (feedforward.py:120-131: h = silu(gate) * side; routed + LoRA:
 * lora_ffn.py:196-222).  forward: h = silu(gate) * side over n_elements (% 4 == 0) contiguous floats.
 * backward, ONE pass over [rows, n] (n % 4 == 0) instead of ~10 elementwise / reduction launches:
 *   grad_gate = grad_h * side * silu'(gate),  grad_side = grad_h * silu(gate),
 *   dots [3, rows] = <grad_h, h>, <grad_gate, gate>, <grad_side, side> per row
 * (the three inner products of the router-coefficient gradient).  All pointers 16-byte aligned.
 */
int spt_swiglu_forward(const float *gate, const float *side, float *h, long long n_elements, void *stream);
int spt_swiglu_backward(const float *grad_h, const float *gate, const float *side, float *grad_gate,
                        float *grad_side, float *dots, long long rows, int n, void *stream);

/*
 * Gradient of table[ids] (the `left` table of a LoRA embedding, lora.py:118-126; autograd of
 * nn.Embedding): out [n_rows, width] = 0, then out[id] = sum of grad[order[t]] over the positions t of
 * the SORTED id list with sorted_ids[t] == id, added in ascending t in fp64 (deterministic; a
 * non-finite gradient row stays in its own id's row).  sorted_ids / order: int64, device memory, the
 * stable sort of the flattened ids and its permutation; grad [n_ids, width] fp32 (row stride ldg),
 * width % 4 == 0; workspace: spt_embedding_rows_backward_workspace_bytes(n_ids, width) bytes.  Launch
 * shapes depend on n_ids alone (replayable as part of a captured HIP graph); ids outside [0, n_rows)
 * are ignored.
 */
long long spt_embedding_rows_backward_workspace_bytes(long long n_ids, int width);
int spt_embedding_rows_backward(const float *grad, long long ldg, const long long *sorted_ids,
                                const long long *order, float *out, long long ldo, void *workspace,
                                long long n_ids, int width, long long n_rows, void *stream);

/*
 * Un-bucketing: out[t, :] = bias + sum_{j < k} rows[pos[t * k + j], :] (bias may be NULL).
 * Replaces the reference's per-block `y[mask] += ...` scatter (lora_ffn.py:107-111) with a
 * gather in a fixed order: deterministic, no atomics.  d % 4 == 0.
 */
int spt_rows_combine(const float *rows, const int32_t *pos, const float *bias, float *out,
                     int n_tokens, int k, int d, void *stream);
/* ... + sum_{j < n_side} side[t, j] * side_w[j, :] on top (side [n_tokens, n_side], side_w [n_side, d],
 * 16-byte aligned): the router's share of the routed FFN's input gradient, d logit . W_router
 * (autograd of feedforward.py:22-25), inside the un-bucketing pass. */
int spt_rows_combine_side(const float *rows, const int32_t *pos, const float *bias, const float *side,
                          const float *side_w, int n_side, float *out, int n_tokens, int k, int d,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPT_HIP_H */
