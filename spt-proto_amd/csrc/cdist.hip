// cdist.hip -- PQ codebook distance + argmin, and its backward, for gfx950.
//
// Replaces extension/cdist.cu of the reference (forward kernel :7-69, backward
// :71-182), which runs one thread per query row in 16-thread blocks.  Here four
// lanes share a query row: lane `sub` owns codewords {4*sub .. 4*sub+3} (+16, +32 ...)
// so that one wave-wide store of `distance` is 64 consecutive float4 = 1 KiB, the
// query read is 16 rows x D floats contiguous, and the codebook sits in LDS where
// equal addresses broadcast.  HBM-bound: bytes = NQ*D*4 in, NQ*C*4 + NQ*4 out per
// subspace.
//
// Bit-exact contract: each distance is the fp32 sum in ascending
// i of |q_i - t_i| starting from 0 (cdist.cu:47-51); argmin uses strict '<' in
// ascending c from (index 0, 1e13) (cdist.cu:28-29,52-54).  A lane scans its own
// codewords in ascending order with strict '<', lanes are merged by (distance,
// index) lexicographic min, which selects the same codeword.
#include "spt_common.h"

namespace spt {

constexpr int CD_THREADS = 256;
constexpr int CD_QPB = CD_THREADS / 4;  // query rows per block per iteration

template <int D>
__global__ __launch_bounds__(CD_THREADS) void cdist_forward_kernel(
    const float *__restrict__ query, const float *__restrict__ table,
    float *__restrict__ distance, int32_t *__restrict__ indices, int NQ, int C,
    int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tab = reinterpret_cast<float *>(smem);  // [Cpad][D], Cpad = C rounded to 4
    const int m = blockIdx.y;
    const int tid = threadIdx.x;
    const int sub = tid & 3;
    const int Cpad = (C + 3) & ~3;

    const float *tsrc = table + (size_t)m * C * D;
    for (int i = tid; i < Cpad * D; i += CD_THREADS) tab[i] = (i < C * D) ? tsrc[i] : 0.0f;
    __syncthreads();

    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(NQ, row0 + rows_per_block);
    for (int base = row0; base < row1; base += CD_QPB) {
        const int q = base + (tid >> 2);
        const bool live = q < row1;
        const int qc = live ? q : (row1 - 1);
        float qv[D];
        const float4 *qp = reinterpret_cast<const float4 *>(query + ((size_t)m * NQ + qc) * D);
#pragma unroll
        for (int i = 0; i < D / 4; i++) {
            float4 t = qp[i];
            qv[4 * i + 0] = t.x; qv[4 * i + 1] = t.y; qv[4 * i + 2] = t.z; qv[4 * i + 3] = t.w;
        }
        int best_i = 0;
        float best_d = 1e13f;
        for (int c0 = 4 * sub; c0 < Cpad; c0 += 16) {
            float d[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 *tp = reinterpret_cast<const float4 *>(tab + (c0 + t) * D);
                float r = 0.0f;
#pragma unroll
                for (int i = 0; i < D / 4; i++) {
                    float4 tv = tp[i];
                    r += fabsf(qv[4 * i + 0] - tv.x);
                    r += fabsf(qv[4 * i + 1] - tv.y);
                    r += fabsf(qv[4 * i + 2] - tv.z);
                    r += fabsf(qv[4 * i + 3] - tv.w);
                }
                d[t] = r;
                const bool cond = (r < best_d) && (c0 + t < C);
                best_i = cond ? (c0 + t) : best_i;
                best_d = cond ? r : best_d;
            }
            if (distance != nullptr && live) {
                float *dp = distance + ((size_t)m * NQ + q) * C + c0;
                if (((C & 3) == 0)) {
                    *reinterpret_cast<float4 *>(dp) = make_float4(d[0], d[1], d[2], d[3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (c0 + t < C) dp[t] = d[t];
                }
            }
        }
        // merge the four lanes of this query: smaller distance, then smaller index
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {
            const float od = __shfl_xor(best_d, step, SPT_WAVE);
            const int oi = __shfl_xor(best_i, step, SPT_WAVE);
            const bool take = (od < best_d) || (od == best_d && oi < best_i);
            best_d = take ? od : best_d;
            best_i = take ? oi : best_i;
        }
        if (live && sub == 0) indices[(size_t)m * NQ + q] = best_i;
    }
}

// Backward.  grad_query[m,q,i] = sum_c s*go[m,q,c], grad_table[m,c,i] = -sum_q s*go[m,q,c]
// with s = (q_i - t_ci) > 0 ? +1 : -1 (cdist.cu:113-119,167-174).  Same 4-lanes-per-row
// layout; each lane keeps its 4 x D slice of grad_table in registers over all the
// rows of the block (requires C == 16), then the block reduces through LDS and writes
// one partial [C][D] slab; cdist_table_reduce_kernel sums the slabs (deterministic).
template <int D>
__global__ __launch_bounds__(CD_THREADS) void cdist_backward_kernel(
    const float *__restrict__ query, const float *__restrict__ table,
    const float *__restrict__ grad_output, float *__restrict__ grad_query,
    float *__restrict__ partial, int NQ, int C, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tab = reinterpret_cast<float *>(smem);  // [C][D]
    float *red = tab + C * D;                      // [C][D] block accumulator
    const int m = blockIdx.y;
    const int tid = threadIdx.x;
    const int sub = tid & 3;

    const float *tsrc = table + (size_t)m * C * D;
    for (int i = tid; i < C * D; i += CD_THREADS) {
        tab[i] = tsrc[i];
        red[i] = 0.0f;
    }
    __syncthreads();

    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(NQ, row0 + rows_per_block);
    for (int c0 = 4 * sub; c0 < C; c0 += 16) {
        float gt[4][D];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < D; i++) gt[t][i] = 0.0f;

        for (int base = row0; base < row1; base += CD_QPB) {
            const int q = base + (tid >> 2);
            const bool live = q < row1;
            const int qc = live ? q : (row1 - 1);
            float qv[D], gq[D];
            const float4 *qp =
                reinterpret_cast<const float4 *>(query + ((size_t)m * NQ + qc) * D);
#pragma unroll
            for (int i = 0; i < D / 4; i++) {
                float4 t = qp[i];
                qv[4 * i + 0] = t.x; qv[4 * i + 1] = t.y; qv[4 * i + 2] = t.z; qv[4 * i + 3] = t.w;
            }
            float4 g4 = *reinterpret_cast<const float4 *>(
                grad_output + ((size_t)m * NQ + qc) * C + c0);
            if (!live) g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int i = 0; i < D; i++) gq[i] = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float *tp = tab + (c0 + t) * D;
#pragma unroll
                for (int i = 0; i < D; i++) {
                    const float sg = (qv[i] - tp[i]) > 0.0f ? g[t] : -g[t];
                    gq[i] += sg;
                    gt[t][i] += sg;
                }
            }
            // grad_query: sum over the 4 lanes (and over the C/16 passes via +=)
#pragma unroll
            for (int i = 0; i < D; i++) gq[i] = group_sum<4>(gq[i]);
            if (live && sub == 0) {
                float *gp = grad_query + ((size_t)m * NQ + q) * D;
                if (c0 == 0) {
#pragma unroll
                    for (int i = 0; i < D / 4; i++)
                        reinterpret_cast<float4 *>(gp)[i] =
                            make_float4(gq[4 * i], gq[4 * i + 1], gq[4 * i + 2], gq[4 * i + 3]);
                } else {
#pragma unroll
                    for (int i = 0; i < D; i++) gp[i] += gq[i];
                }
            }
        }
        // block reduction of this lane's grad_table slice
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < D; i++) atomicAdd(&red[(c0 + t) * D + i], gt[t][i]);
    }
    __syncthreads();
    float *dst = partial + ((size_t)m * gridDim.x + blockIdx.x) * C * D;
    for (int i = tid; i < C * D; i += CD_THREADS) dst[i] = red[i];
}

__global__ void cdist_table_reduce_kernel(const float *__restrict__ partial,
                                          float *__restrict__ grad_table, int nblk,
                                          int CD_elems) {
    const int m = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CD_elems) return;
    const float *src = partial + (size_t)m * nblk * CD_elems + i;
    float acc = 0.0f;
    for (int b = 0; b < nblk; b++) acc += src[(size_t)b * CD_elems];
    grad_table[(size_t)m * CD_elems + i] = -acc;
}

static int fwd_rows_per_block(int NQ) {
    // enough blocks to fill 256 CUs several times over, few enough to amortise the
    // codebook load: 256..1024 rows per block
    int rpb = 256;
    while (rpb < 1024 && (NQ / rpb) > 4096) rpb <<= 1;
    return rpb;
}

static int bwd_blocks(int NQ) {
    int nblk = (NQ + 2047) / 2048;
    if (nblk < 1) nblk = 1;
    if (nblk > 256) nblk = 256;
    return nblk;
}

}  // namespace spt

using namespace spt;

extern "C" int spt_cdist_forward(const float *query, const float *table, float *distance,
                                 int32_t *indices, int n_subspaces, int n_queries,
                                 int n_codewords, int d_code, void *stream) {
    if (!query || !table || !indices) return SPT_EINVAL;
    if (n_subspaces <= 0 || n_queries <= 0 || n_codewords <= 0 || d_code <= 0) return SPT_EINVAL;
    if (d_code % 4 != 0) return SPT_ESHAPE;  // cdist.cu:200
    if (n_subspaces > 65535) return SPT_EUNSUP;
    const int rpb = fwd_rows_per_block(n_queries);
    dim3 grid((n_queries + rpb - 1) / rpb, n_subspaces);
    const int Cpad = (n_codewords + 3) & ~3;
    const size_t lds = (size_t)Cpad * d_code * sizeof(float);
    if (lds > 64 * 1024) return SPT_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
#define SPT_CD_FWD(DD)                                                                   \
    hipLaunchKernelGGL((cdist_forward_kernel<DD>), grid, dim3(CD_THREADS), lds, s, query, \
                       table, distance, indices, n_queries, n_codewords, rpb)
    switch (d_code) {  // same set as cdist.cu:213-245
        case 4: SPT_CD_FWD(4); break;
        case 8: SPT_CD_FWD(8); break;
        case 16: SPT_CD_FWD(16); break;
        case 24: SPT_CD_FWD(24); break;
        case 32: SPT_CD_FWD(32); break;
        default: return SPT_EUNSUP;  // "d_code not supported", cdist.cu:243-245
    }
#undef SPT_CD_FWD
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int64_t spt_cdist_backward_workspace_bytes(int n_subspaces, int n_queries,
                                                      int n_codewords, int d_code) {
    if (n_subspaces <= 0 || n_queries <= 0 || n_codewords <= 0 || d_code <= 0) return 0;
    return (int64_t)n_subspaces * bwd_blocks(n_queries) * n_codewords * d_code *
           (int64_t)sizeof(float);
}

extern "C" int spt_cdist_backward(const float *query, const float *table,
                                  const float *grad_output, float *grad_query,
                                  float *grad_table, void *workspace, int n_subspaces,
                                  int n_queries, int n_codewords, int d_code, void *stream) {
    if (!query || !table || !grad_output || !grad_query || !grad_table || !workspace)
        return SPT_EINVAL;
    if (n_subspaces <= 0 || n_queries <= 0 || n_codewords <= 0 || d_code <= 0) return SPT_EINVAL;
    if (d_code % 4 != 0) return SPT_ESHAPE;
    if (n_codewords % 16 != 0) return SPT_ESHAPE;  // cdist.cu:275 (float4 rows of grad_output)
    if (n_subspaces > 65535) return SPT_EUNSUP;
    const int nblk = bwd_blocks(n_queries);
    const int rpb = (((n_queries + nblk - 1) / nblk) + CD_QPB - 1) / CD_QPB * CD_QPB;
    const int CDe = n_codewords * d_code;
    const size_t lds = 2 * (size_t)CDe * sizeof(float);
    if (lds > 64 * 1024) return SPT_EUNSUP;
    dim3 grid(nblk, n_subspaces);
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
#define SPT_CD_BWD(DD)                                                                    \
    hipLaunchKernelGGL((cdist_backward_kernel<DD>), grid, dim3(CD_THREADS), lds, s, query, \
                       table, grad_output, grad_query, partial, n_queries, n_codewords, rpb)
    switch (d_code) {
        case 4: SPT_CD_BWD(4); break;
        case 8: SPT_CD_BWD(8); break;
        case 16: SPT_CD_BWD(16); break;
        case 24: SPT_CD_BWD(24); break;
        case 32: SPT_CD_BWD(32); break;
        default: return SPT_EUNSUP;
    }
#undef SPT_CD_BWD
    SPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(cdist_table_reduce_kernel, dim3((CDe + 255) / 256, n_subspaces),
                       dim3(256), 0, s, partial, grad_table, nblk, CDe);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

// ------------------------------------------------------------------ PQ encode, head layout
//
// The attention layers hold q / k as [N, S, H, E]; the reference first copies them to
// [N*H, S, E], then again to the [M, NQ, D] layout of cdist, and transposes the codes
// back (attention.py:92-95, quantizer.py:44-48,75-77).  This kernel reads the original
// layout and writes the codes where lookup wants them ([N*H, S, M]): one lane per
// (token, head, subspace), so a wave reads 8 whole 256-byte head vectors per
// instruction (E = 64) and writes 8 x 32 bytes of codes.  Same bit-exact distance /
// argmin contract as cdist_forward_kernel.
namespace spt {

// T: float, or uint16_t = bf16 storage (raw patterns, widened exactly: same arithmetic after)
template <int D, typename T>
__global__ __launch_bounds__(CD_THREADS) void pq_encode_heads_kernel(
    const T *__restrict__ z, const float *__restrict__ table, int32_t *__restrict__ codes,
    int n_vectors, int S, int H, int M, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [M][C*D + 4]: the 8 lanes of a token read the SAME codeword of 8 DIFFERENT subspaces;
    // unpadded, those rows are C*D*4 = 512 bytes apart = the same LDS banks (8-way conflict)
    float *tab = reinterpret_cast<float *>(smem);
    const int tstride = C * D + 4;
    for (int i = threadIdx.x; i < M * C * D; i += CD_THREADS) {
        const int mm = i / (C * D);
        tab[mm * tstride + (i - mm * C * D)] = table[i];
    }
    __syncthreads();
    // 32-bit index math throughout (the host guarantees n_vectors * M < 2^31): 64-bit
    // divisions by run-time values cost ~100 instructions each on this ISA
    const int total = n_vectors * M;
    for (int t = blockIdx.x * CD_THREADS + threadIdx.x; t < total;
         t += gridDim.x * CD_THREADS) {
        const int vec = t / M;                // index into [N, S, H]
        const int m = t - vec * M;
        float qv[D];
        const T *qp = z + (size_t)vec * (M * D) + m * D;
#pragma unroll
        for (int i = 0; i < D / 4; i++) {
            float4 v;
            if constexpr (sizeof(T) == 4) {
                v = reinterpret_cast<const float4 *>(qp)[i];
            } else {
                const uint2 w = reinterpret_cast<const uint2 *>(qp)[i];
                v = make_float4(__builtin_bit_cast(float, w.x << 16),
                                __builtin_bit_cast(float, w.x & 0xffff0000u),
                                __builtin_bit_cast(float, w.y << 16),
                                __builtin_bit_cast(float, w.y & 0xffff0000u));
            }
            qv[4 * i + 0] = v.x; qv[4 * i + 1] = v.y; qv[4 * i + 2] = v.z; qv[4 * i + 3] = v.w;
        }
        int best_i = 0;
        float best_d = 1e13f;
        const float *tm = tab + m * tstride;
        for (int c = 0; c < C; c++) {
            const float4 *tp = reinterpret_cast<const float4 *>(tm + c * D);
            float r = 0.0f;
#pragma unroll
            for (int i = 0; i < D / 4; i++) {
                const float4 tv = tp[i];
                r += fabsf(qv[4 * i + 0] - tv.x);
                r += fabsf(qv[4 * i + 1] - tv.y);
                r += fabsf(qv[4 * i + 2] - tv.z);
                r += fabsf(qv[4 * i + 3] - tv.w);
            }
            const bool cond = r < best_d;
            best_i = cond ? c : best_i;
            best_d = cond ? r : best_d;
        }
        // vec = (n * S + s) * H + h  ->  code row (n * H + h) * S + s
        const int ns = vec / H;
        const int h = vec - ns * H;
        const int n = ns / S;
        const int s = ns - n * S;
        codes[((size_t)(n * H + h) * S + s) * M + m] = best_i;
    }
}

}  // namespace spt

template <typename T>
static int pq_encode_heads_any(const T *z, const float *table, int32_t *codes, int batch,
                               int seq_length, int n_heads, int n_subspaces, int n_codewords,
                               int d_code, void *stream) {
    using namespace spt;
    if (!z || !table || !codes) return SPT_EINVAL;
    if (batch <= 0 || seq_length <= 0 || n_heads <= 0 || n_subspaces <= 0 || n_codewords <= 0 ||
        d_code <= 0)
        return SPT_EINVAL;
    if (d_code % 4 != 0) return SPT_ESHAPE;
    const size_t lds = (size_t)n_subspaces * (n_codewords * d_code + 4) * sizeof(float);
    if (lds > 64 * 1024) return SPT_EUNSUP;
    const long long n_vectors = (long long)batch * seq_length * n_heads;
    const long long total = n_vectors * n_subspaces;
    if (total >= 0x7FFFFFFFLL - 256 * 1024) return SPT_EUNSUP;
    long long nblk = (total + CD_THREADS - 1) / CD_THREADS;
    if (nblk > 256 * 8) nblk = 256 * 8;  // grid-stride: the codebook load is per block
    hipStream_t s = (hipStream_t)stream;
#define SPT_PQ(DD)                                                                       \
    hipLaunchKernelGGL((spt::pq_encode_heads_kernel<DD, T>), dim3((unsigned)nblk),       \
                       dim3(spt::CD_THREADS), lds, s, z, table, codes, (int)n_vectors,   \
                       seq_length, n_heads, n_subspaces, n_codewords)
    switch (d_code) {
        case 4: SPT_PQ(4); break;
        case 8: SPT_PQ(8); break;
        case 16: SPT_PQ(16); break;
        case 24: SPT_PQ(24); break;
        case 32: SPT_PQ(32); break;
        default: return SPT_EUNSUP;
    }
#undef SPT_PQ
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_pq_encode_heads(const float *z, const float *table, int32_t *codes,
                                   int batch, int seq_length, int n_heads, int n_subspaces,
                                   int n_codewords, int d_code, void *stream) {
    return pq_encode_heads_any(z, table, codes, batch, seq_length, n_heads, n_subspaces,
                               n_codewords, d_code, stream);
}

extern "C" int spt_pq_encode_heads_bf16(const uint16_t *z, const float *table, int32_t *codes,
                                        int batch, int seq_length, int n_heads, int n_subspaces,
                                        int n_codewords, int d_code, void *stream) {
    return pq_encode_heads_any(z, table, codes, batch, seq_length, n_heads, n_subspaces,
                               n_codewords, d_code, stream);
}
