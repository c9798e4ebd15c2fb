// softmax.hip -- masked CSR row softmax and its VJP for gfx950.
//
// Replaces extension/softmax.cu:7-81 of the reference (one thread per row, two
// serial passes, 16-thread blocks).  Here a group of G consecutive lanes owns one
// row, G chosen on the host from the mean row length so that one float4 per lane
// covers the row (Z = 64 -> G = 16: one wave handles 4 rows and every load / store
// instruction moves 1 KiB).  Row sums are DPP butterflies.  Pure streaming:
// 3 * nnz * 4 bytes forward, 4 * nnz * 4 bytes backward.
//
// Semantics kept from the reference: mask_p = indices[p] <= row, no max subtraction,
// denominator max(1e-9, sum) (softmax.cu:30), and in the backward the *clamped*
// c = max(1e-9, sum mask*y*dy) (softmax.cu:69).
#include "spt_common.h"

namespace spt {

constexpr int SM_THREADS = 256;

// MODE 0: forward, a = values.  MODE 1: backward, a = output (y), b = grad_output.
// MODE 2: backward chained through the layer's clamp(scale * raw, -clampv, clampv)
// (naive_gpt/layers/sparse/attention.py:125-127): `cin` holds the clamped scores that
// fed the softmax; the gradient is scaled by `scale` where |cin| < clampv and zeroed
// on the rails.  Saves four elementwise passes over the [B, nnz] tensor.
template <int G, int MODE>
__global__ __launch_bounds__(SM_THREADS) void softmax_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ a, const float *__restrict__ bgrad, float *__restrict__ out,
    int S, int nnz, long long total_rows, const float *__restrict__ cin, float scale,
    float clampv) {
    const int gl = threadIdx.x & (G - 1);
    const long long grow = ((long long)blockIdx.x * SM_THREADS + threadIdx.x) / G;
    // whole groups are in or out together, and G divides 64: reductions stay uniform
    if (grow >= total_rows) return;
    const int b = (int)(grow / S);
    const int row = (int)(grow - (long long)b * S);
    const int start = indptr[row], end = indptr[row + 1];
    const size_t base = (size_t)b * nnz;
    const bool vec = (((start | end | nnz) & 3) == 0);

    float sum = 0.0f;
    if (vec) {
        for (int p = start + 4 * gl; p < end; p += 4 * G) {
            const float4 v = *reinterpret_cast<const float4 *>(a + base + p);
            const int4 ix = *reinterpret_cast<const int4 *>(indices + base + p);
            if (MODE == 0) {
                sum += (ix.x <= row) ? expf(v.x) : 0.0f;
                sum += (ix.y <= row) ? expf(v.y) : 0.0f;
                sum += (ix.z <= row) ? expf(v.z) : 0.0f;
                sum += (ix.w <= row) ? expf(v.w) : 0.0f;
            } else {
                const float4 g = *reinterpret_cast<const float4 *>(bgrad + base + p);
                sum += (ix.x <= row) ? v.x * g.x : 0.0f;
                sum += (ix.y <= row) ? v.y * g.y : 0.0f;
                sum += (ix.z <= row) ? v.z * g.z : 0.0f;
                sum += (ix.w <= row) ? v.w * g.w : 0.0f;
            }
        }
    } else {
        for (int p = start + gl; p < end; p += G) {
            const float v = a[base + p];
            const bool keep = indices[base + p] <= row;
            if (MODE == 0) sum += keep ? expf(v) : 0.0f;
            else sum += keep ? v * bgrad[base + p] : 0.0f;
        }
    }
    sum = group_sum<G>(sum);
    sum = fmaxf(1e-9f, sum);

    if (MODE == 0) {
        const float scale = 1.0f / sum;
        if (vec) {
            for (int p = start + 4 * gl; p < end; p += 4 * G) {
                const float4 v = *reinterpret_cast<const float4 *>(a + base + p);
                const int4 ix = *reinterpret_cast<const int4 *>(indices + base + p);
                float4 y;
                y.x = (ix.x <= row) ? scale * expf(v.x) : 0.0f;
                y.y = (ix.y <= row) ? scale * expf(v.y) : 0.0f;
                y.z = (ix.z <= row) ? scale * expf(v.z) : 0.0f;
                y.w = (ix.w <= row) ? scale * expf(v.w) : 0.0f;
                *reinterpret_cast<float4 *>(out + base + p) = y;
            }
        } else {
            for (int p = start + gl; p < end; p += G) {
                const bool keep = indices[base + p] <= row;
                out[base + p] = keep ? scale * expf(a[base + p]) : 0.0f;
            }
        }
    } else {
        if (vec) {
            for (int p = start + 4 * gl; p < end; p += 4 * G) {
                const float4 v = *reinterpret_cast<const float4 *>(a + base + p);
                const float4 g = *reinterpret_cast<const float4 *>(bgrad + base + p);
                const int4 ix = *reinterpret_cast<const int4 *>(indices + base + p);
                float4 y;
                y.x = (ix.x <= row) ? v.x * (g.x - sum) : 0.0f;
                y.y = (ix.y <= row) ? v.y * (g.y - sum) : 0.0f;
                y.z = (ix.z <= row) ? v.z * (g.z - sum) : 0.0f;
                y.w = (ix.w <= row) ? v.w * (g.w - sum) : 0.0f;
                if (MODE == 2) {
                    const float4 c = *reinterpret_cast<const float4 *>(cin + base + p);
                    y.x = (fabsf(c.x) < clampv) ? y.x * scale : 0.0f;
                    y.y = (fabsf(c.y) < clampv) ? y.y * scale : 0.0f;
                    y.z = (fabsf(c.z) < clampv) ? y.z * scale : 0.0f;
                    y.w = (fabsf(c.w) < clampv) ? y.w * scale : 0.0f;
                }
                *reinterpret_cast<float4 *>(out + base + p) = y;
            }
        } else {
            for (int p = start + gl; p < end; p += G) {
                const bool keep = indices[base + p] <= row;
                float y = keep ? a[base + p] * (bgrad[base + p] - sum) : 0.0f;
                if (MODE == 2) y = (fabsf(cin[base + p]) < clampv) ? y * scale : 0.0f;
                out[base + p] = y;
            }
        }
    }
}

template <int MODE>
static int softmax_launch(const int32_t *indptr, const int32_t *indices, const float *a,
                          const float *bgrad, float *out, int B, int S, int nnz,
                          hipStream_t s, const float *cin = nullptr, float scale = 1.0f,
                          float clampv = 0.0f) {
    const long long rows = (long long)B * S;
    // lanes per row: one float4 per lane covers the mean row, at least 4, at most 64
    int G = pow2_ceil((nnz / S + 3) / 4);
    if (G < 4) G = 4;
    if (G > 64) G = 64;
    const long long threads = rows * G;
    const long long nblk = (threads + SM_THREADS - 1) / SM_THREADS;
    if (nblk > 0x7FFFFFFFLL) return SPT_EUNSUP;
    dim3 grid((unsigned)nblk);
#define SPT_SM(GG)                                                                     \
    hipLaunchKernelGGL((softmax_kernel<GG, MODE>), grid, dim3(SM_THREADS), 0, s, indptr, \
                       indices, a, bgrad, out, S, nnz, rows, cin, scale, clampv)
    switch (G) {
        case 4: SPT_SM(4); break;
        case 8: SPT_SM(8); break;
        case 16: SPT_SM(16); break;
        case 32: SPT_SM(32); break;
        default: SPT_SM(64); break;
    }
#undef SPT_SM
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

}  // namespace spt

using namespace spt;

extern "C" int spt_softmax_forward(const int32_t *indptr, const int32_t *indices,
                                   const float *values, float *output, int batch_size,
                                   int seq_length, int nnz, void *stream) {
    if (!indptr || !indices || !values || !output) return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || nnz < 0) return SPT_EINVAL;
    if (nnz == 0) return SPT_OK;
    return softmax_launch<0>(indptr, indices, values, nullptr, output, batch_size, seq_length,
                             nnz, (hipStream_t)stream);
}

extern "C" int spt_softmax_backward(const int32_t *indptr, const int32_t *indices,
                                    const float *output, const float *grad_output,
                                    float *grad_values, int batch_size, int seq_length,
                                    int nnz, void *stream) {
    if (!indptr || !indices || !output || !grad_output || !grad_values) return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || nnz < 0) return SPT_EINVAL;
    if (nnz == 0) return SPT_OK;
    return softmax_launch<1>(indptr, indices, output, grad_output, grad_values, batch_size,
                             seq_length, nnz, (hipStream_t)stream);
}

extern "C" int spt_softmax_backward_clamped(const int32_t *indptr, const int32_t *indices,
                                            const float *output, const float *grad_output,
                                            const float *clamped_scores, float scale, float clampv,
                                            float *grad_scores, int batch_size, int seq_length,
                                            int nnz, void *stream) {
    if (!indptr || !indices || !output || !grad_output || !clamped_scores || !grad_scores)
        return SPT_EINVAL;
    if (batch_size <= 0 || seq_length <= 0 || nnz < 0 || !(clampv > 0.0f)) return SPT_EINVAL;
    if (nnz == 0) return SPT_OK;
    return softmax_launch<2>(indptr, indices, output, grad_output, grad_scores, batch_size,
                             seq_length, nnz, (hipStream_t)stream, clamped_scores, scale, clampv);
}
