/*
 * spt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the seven native operators of ytgui/SPT-proto's
 * `naive_gpt.ext` (reference: extension/entry.cpp:43-56).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (spt-proto_amd/) never does.
 *
 * Parity status of each function (see DESIGN.md "Oracle"):
 *   cdist fwd/bwd  pinned: checked against the reference's own PQ-v1 path
 *                  (naive_gpt/layers/basic/quantizer.py:54-63, imported in the
 *                  build container) and its test formula test/kernel/test_cdist.py:24-52.
 *   softmax        pinned by the formula of test/kernel/test_softmax.py:66-92.
 *   sddmm / spmm   pinned by test/kernel/test_sddmm.py:57-85, test_spmm.py:55-82
 *                  (cuSPARSE itself is closed source, version unpinned).
 *   lookup         PARITY UNPINNED beyond the reference's recall property
 *                  (test/kernel/test_lookup.py:57-75): the reference holds no
 *                  golden vector for the CSR structure and its CUDA source cannot be
 *                  built here; this file is a literal sequential emulation of
 *                  extension/lookup.cu:10-84 and is the arbiter.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math: the
 * fp32 summation orders below are part of the contract).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ cdist */

/* extension/cdist.cu:7-69.  query [M,NQ,D], table [M,C,D] ->
 * distance [M,NQ,C] (may be NULL), indices [M,NQ].
 * d = sum_i |q_i - t_i| in fp32, i ascending (cdist.cu:47-51);
 * argmin with strict '<', c ascending, start value 1e13 (cdist.cu:28-29,52-54). */
void spt_oracle_cdist_forward(const float *query, const float *table,
                              float *distance, int32_t *indices,
                              int M, int NQ, int C, int D) {
    for (int m = 0; m < M; m++) {
        for (int q = 0; q < NQ; q++) {
            const float *qv = query + ((size_t)m * NQ + q) * D;
            int32_t min_index = 0;
            float min_distance = 1e13f;
            for (int c = 0; c < C; c++) {
                const float *tv = table + ((size_t)m * C + c) * D;
                float reduced = 0.0f;
                for (int i = 0; i < D; i++) {
                    reduced += fabsf(qv[i] - tv[i]);
                }
                int cond = reduced < min_distance;
                min_index = cond ? c : min_index;
                min_distance = cond ? reduced : min_distance;
                if (distance) {
                    distance[((size_t)m * NQ + q) * C + c] = reduced;
                }
            }
            indices[(size_t)m * NQ + q] = min_index;
        }
    }
}

/* extension/cdist.cu:71-131 (grad wrt query) and :133-182 (grad wrt table).
 * sign s = (q_i - t_ci) > 0 ? +1 : -1   (zero difference -> -1, cdist.cu:117,170-173)
 * gq[m,q,i] = sum_c go[m,q,c]*s   (c ascending)
 * gt[m,c,i] = -sum_q go[m,q,c]*s  (q ascending, `reduced -= grad_abs`) */
void spt_oracle_cdist_backward(const float *query, const float *table,
                               const float *grad_output, float *grad_query,
                               float *grad_table, int M, int NQ, int C, int D) {
    for (int m = 0; m < M; m++) {
        for (int q = 0; q < NQ; q++) {
            const float *qv = query + ((size_t)m * NQ + q) * D;
            const float *go = grad_output + ((size_t)m * NQ + q) * C;
            float *gq = grad_query + ((size_t)m * NQ + q) * D;
            for (int i = 0; i < D; i++) gq[i] = 0.0f;
            for (int c = 0; c < C; c++) {
                const float *tv = table + ((size_t)m * C + c) * D;
                float g = go[c];
                for (int i = 0; i < D; i++) {
                    gq[i] += (qv[i] - tv[i]) > 0 ? g : -g;
                }
            }
        }
        for (int c = 0; c < C; c++) {
            const float *tv = table + ((size_t)m * C + c) * D;
            float *gt = grad_table + ((size_t)m * C + c) * D;
            for (int i = 0; i < D; i++) {
                float reduced = 0.0f;
                for (int q = 0; q < NQ; q++) {
                    float g = grad_output[((size_t)m * NQ + q) * C + c];
                    float qi = query[((size_t)m * NQ + q) * D + i];
                    float grad_abs = (qi - tv[i]) > 0 ? g : -g;
                    reduced -= grad_abs;
                }
                gt[i] = reduced;
            }
        }
    }
}

/* ----------------------------------------------------------------- lookup */

/* Literal sequential emulation of lookup_forward_kernel, extension/lookup.cu:10-84,
 * for one thread block = 16 rows x 4 workers, run in CUDA lock-step order:
 * every thread executes loop iteration (window, j) together; when two workers
 * write the same shared-memory word in the same iteration the HIGHER lane wins
 * (the canonical rule fixed by SURVEY.md 8a-2; CUDA leaves it unspecified).
 *
 * left/right [B,S,M] int32 codes (compared after truncation to uint16,
 * lookup.cu:22,43), output [B,S,Z] int32, zero-initialised (lookup.cu:107-109).
 * Returns 0, or -1 if a precondition of lookup.cu:103-106 fails. */
#define LK_BLOCK 16
#define LK_WORKER 4
#define LK_SLOTS 4

int spt_oracle_lookup_forward(const int32_t *left, const int32_t *right,
                              int32_t *output, int B, int S, int M,
                              int sparsity) {
    if (S % LK_BLOCK != 0 || sparsity <= 0 || S % sparsity != 0) return -1;
    const int Z = S / sparsity;
    if (Z % LK_BLOCK != 0 || M < LK_SLOTS) return -1;
    memset(output, 0, (size_t)B * S * Z * sizeof(int32_t));

    uint16_t *indices = (uint16_t *)malloc((size_t)LK_BLOCK * LK_SLOTS * Z * sizeof(uint16_t));
    uint16_t *cache_lhs = (uint16_t *)malloc((size_t)LK_BLOCK * M * sizeof(uint16_t));
    uint16_t *cache_rhs = (uint16_t *)malloc((size_t)LK_BLOCK * M * sizeof(uint16_t));
    int cursors[LK_BLOCK][LK_WORKER][LK_SLOTS];

    for (int gz = 0; gz < B; gz++) {
        for (int by = 0; by < S / LK_BLOCK; by++) {
            /* per-block state; shared memory starts undefined, cursors = tx */
            memset(indices, 0, (size_t)LK_BLOCK * LK_SLOTS * Z * sizeof(uint16_t));
            for (int ty = 0; ty < LK_BLOCK; ty++) {
                int gy = by * LK_BLOCK + ty;
                for (int k = 0; k < M; k++)
                    cache_lhs[ty * M + k] = (uint16_t)left[((size_t)gz * S + gy) * M + k];
                for (int tx = 0; tx < LK_WORKER; tx++)
                    for (int s = 0; s < LK_SLOTS; s++) cursors[ty][tx][s] = tx;
            }
            /* window loop, lookup.cu:35-70 */
            for (int offset_x = 0; offset_x < S; offset_x += LK_BLOCK) {
                if (offset_x > by * LK_BLOCK) break; /* gy - ty == first row of block */
                for (int ty = 0; ty < LK_BLOCK; ty++)
                    for (int k = 0; k < M; k++)
                        cache_rhs[ty * M + k] =
                            (uint16_t)right[((size_t)gz * S + offset_x + ty) * M + k];
                /* lock-step: iteration j of every thread, lanes in ascending order */
                for (int j = 0; j < LK_BLOCK / LK_WORKER; j++) {
                    for (int ty = 0; ty < LK_BLOCK; ty++) {
                        int gy = by * LK_BLOCK + ty;
                        for (int tx = 0; tx < LK_WORKER; tx++) {
                            int local_x = tx + j * LK_WORKER;
                            if (offset_x + local_x > gy) continue; /* tril break */
                            int count = 0;
                            for (int k = 0; k < M; k++)
                                count += cache_lhs[ty * M + k] == cache_rhs[local_x * M + k];
                            int slot = count / (M / LK_SLOTS);
                            if (slot > LK_SLOTS - 1) slot = LK_SLOTS - 1;
                            int cursor = cursors[ty][tx][slot];
                            indices[((size_t)ty * LK_SLOTS + slot) * Z + cursor] =
                                (uint16_t)(offset_x + local_x);
                            int next = cursor + LK_WORKER;
                            int cap = Z - tx - 1;
                            cursors[ty][tx][slot] = next < cap ? next : cap;
                        }
                    }
                }
            }
            /* store, lookup.cu:72-83 */
            for (int ty = 0; ty < LK_BLOCK; ty++) {
                int gy = by * LK_BLOCK + ty;
                int limit = (gy + 1) < Z ? (gy + 1) : Z;
                for (int tx = 0; tx < LK_WORKER; tx++) {
                    int slot = LK_SLOTS - 1, cursor = tx;
                    for (int local_x = tx; local_x < limit; local_x += LK_WORKER) {
                        while (slot >= 0 && cursor >= cursors[ty][tx][slot]) {
                            slot = slot - 1;
                            cursor = tx;
                        }
                        if (slot < 0) break;
                        output[((size_t)gz * S + gy) * Z + local_x] =
                            indices[((size_t)ty * LK_SLOTS + slot) * Z + cursor];
                        cursor += LK_WORKER;
                    }
                }
            }
        }
    }
    free(indices);
    free(cache_lhs);
    free(cache_rhs);
    return 0;
}

/* ------------------------------------------------------------ sddmm / spmm */

/* extension/sddmm.cpp:27-69 (cusparseSDDMM, op_lhs=N, op_rhs=T, alpha=1, beta=0):
 * out[b,p] = sum_e Q[b,row(p),e] * K[b,indices[b,p],e].  indptr [S+1] is shared
 * by all batches, indices/out are [B,nnz].  cuSPARSE's summation order is
 * unspecified; the oracle accumulates in double and rounds once. */
void spt_oracle_sddmm_forward(const int32_t *indptr, const int32_t *indices,
                              const float *query, const float *key, float *out,
                              int B, int S, int E, int nnz) {
    for (int b = 0; b < B; b++) {
        for (int r = 0; r < S; r++) {
            const float *qv = query + ((size_t)b * S + r) * E;
            for (int p = indptr[r]; p < indptr[r + 1]; p++) {
                int col = indices[(size_t)b * nnz + p];
                const float *kv = key + ((size_t)b * S + col) * E;
                double acc = 0.0;
                for (int e = 0; e < E; e++) acc += (double)qv[e] * (double)kv[e];
                out[(size_t)b * nnz + p] = (float)acc;
            }
        }
    }
}

/* extension/spmm.cpp:27-69 (cusparseSpMM, alpha=1, beta=0):
 * trans_lhs == 0: Y[b,r,:]            = sum_{p in row r} values[b,p] * X[b,indices[b,p],:]
 * trans_lhs != 0: Y[b,indices[b,p],:] += values[b,p] * X[b,row(p),:]
 * Duplicate column entries accumulate.  Double accumulation, rounded once. */
void spt_oracle_spmm_forward(int trans_lhs, const int32_t *indptr,
                             const int32_t *indices, const float *values,
                             const float *x, float *y, int B, int S, int E,
                             int nnz) {
    double *acc = (double *)malloc((size_t)S * E * sizeof(double));
    for (int b = 0; b < B; b++) {
        memset(acc, 0, (size_t)S * E * sizeof(double));
        for (int r = 0; r < S; r++) {
            for (int p = indptr[r]; p < indptr[r + 1]; p++) {
                int col = indices[(size_t)b * nnz + p];
                double v = values[(size_t)b * nnz + p];
                int src = trans_lhs ? r : col;
                int dst = trans_lhs ? col : r;
                const float *xv = x + ((size_t)b * S + src) * E;
                for (int e = 0; e < E; e++) acc[(size_t)dst * E + e] += v * (double)xv[e];
            }
        }
        for (size_t i = 0; i < (size_t)S * E; i++) y[(size_t)b * S * E + i] = (float)acc[i];
    }
    free(acc);
}

/* ---------------------------------------------------------------- softmax */

/* extension/softmax.cu:7-47.  mask_p = indices[b,p] <= row; no max-subtraction;
 * cumulated in fp32 in entry order (softmax.cu:17-29); fmax(1e-9, .) (:30);
 * scale = 1.0 / cumulated evaluated in double then rounded (:33);
 * y = scale * expf(v) * mask (:41-44). */
void spt_oracle_softmax_forward(const int32_t *indptr, const int32_t *indices,
                                const float *values, float *output, int B,
                                int S, int nnz) {
    for (int b = 0; b < B; b++) {
        for (int gy = 0; gy < S; gy++) {
            float cumulated = 0.0f;
            for (int p = indptr[gy]; p < indptr[gy + 1]; p++) {
                size_t o = (size_t)b * nnz + p;
                cumulated += expf(values[o]) * (float)(indices[o] <= gy);
            }
            cumulated = (float)fmax(1e-9, (double)cumulated);
            float scale = (float)(1.0 / (double)cumulated);
            for (int p = indptr[gy]; p < indptr[gy + 1]; p++) {
                size_t o = (size_t)b * nnz + p;
                output[o] = scale * expf(values[o]) * (float)(indices[o] <= gy);
            }
        }
    }
}

/* extension/softmax.cu:49-81.  c = fmax(1e-9, sum mask*y*dy)  -- the clamp is a
 * reference quirk that parity keeps (:69); dv = y * (dy - c) * mask (:75-78). */
void spt_oracle_softmax_backward(const int32_t *indptr, const int32_t *indices,
                                 const float *output, const float *grad_output,
                                 float *grad_values, int B, int S, int nnz) {
    for (int b = 0; b < B; b++) {
        for (int gy = 0; gy < S; gy++) {
            float cumulated = 0.0f;
            for (int p = indptr[gy]; p < indptr[gy + 1]; p++) {
                size_t o = (size_t)b * nnz + p;
                cumulated += output[o] * grad_output[o] * (float)(indices[o] <= gy);
            }
            cumulated = (float)fmax(1e-9, (double)cumulated);
            for (int p = indptr[gy]; p < indptr[gy + 1]; p++) {
                size_t o = (size_t)b * nnz + p;
                grad_values[o] =
                    output[o] * (grad_output[o] - cumulated) * (float)(indices[o] <= gy);
            }
        }
    }
}
