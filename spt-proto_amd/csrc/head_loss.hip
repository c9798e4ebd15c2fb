// head_loss.hip -- softmax cross-entropy of the language-model head, loss and gradient in one pass,
// in place.
//
// Reference: script/4-sparse-tuning-0.py:45-59 (`loss_fn = nn.CrossEntropyLoss()` on the
// flattened logits [tokens, vocabulary]); the step's last operator before the backward.  As
// separate library operators the logits [8192, 30522] (1 GB) are read and written five times --
// log_softmax (read, write), its backward (two reads, a write) and the zero fill of the gradient:
// 1.35 ms of the 73 ms BERT-large fine-tune step and three 1 GB tensors alive at the step's
// memory peak.  Here a workgroup owns a row: it reads the row once into registers (vocabularies
// up to 32768; twice above that), forms max, sum, the row's loss, and writes
//     z[i, :] <- (softmax(z[i, :]) - onehot(target_i)) * scale          (scale = 1 / #targets)
// over the logits.  Rows may be padded (ld >= n_classes, ld % 4 == 0): the pad columns are
// written as zeros, so that the buffer is directly the A operand of the head's backward GEMM
// (contraction over the vocabulary, 16-byte aligned rows).  fp32 throughout; exp and log are the
// fast intrinsics (__expf / __logf: v_exp_f32 / v_log_f32 with a pre-scale, ~2 ulp -- the library's
// log_softmax uses the precise expf / logf; tests bound the difference at 1e-5 of the loss, 1e-4 of a
// gradient element); the row reductions are fixed-order trees (reproducible).
// Two deliberate differences from nn.CrossEntropyLoss: a target outside [0, n_classes) that is not
// `ignore_index` counts as ignored here (torch: a device-side assert), and a row that is not
// counted gets a zero gradient even when NO row is counted (torch's mean then divides 0 by 0: the
// host side, layers/tuning/head_loss.py, returns torch's NaN loss in that case; the gradients stay
// zero instead of NaN).
#include "spt_common.h"

namespace spt {

constexpr int HL_THREADS = 256;
constexpr int HL_MAXJ = 32;          // float4 per thread held in registers: ld <= 32768

__device__ __forceinline__ float hl_block_max(float v, float *scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, SPT_WAVE));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    v = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float hl_block_sum(float v, float *scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, SPT_WAVE);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    v = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    __syncthreads();
    return v;
}

// RESIDENT: the row lives in registers between the reduction and the write (one read of z)
template <bool RESIDENT>
__global__ __launch_bounds__(HL_THREADS) void cross_entropy_grad_kernel(
    float *__restrict__ z, long long ld, int n_classes, const long long *__restrict__ target,
    const float *__restrict__ scale_ptr, float *__restrict__ loss, long long ignore_index) {
    __shared__ float scratch[4];
    const long long row = blockIdx.x;
    float *zr = z + row * ld;
    const long long t = target[row];
    const bool counted = t != ignore_index && t >= 0 && t < n_classes;
    const float zt = counted ? zr[t] : 0.0f;                  // (read before anything is written)
    const float scale = counted ? *scale_ptr : 0.0f;
    const int n4 = (int)(ld >> 2);                            // float4 per row, pad included
    const float ninf = -__builtin_huge_valf();
    auto masked = [&](float4 v, int c) {                      // columns >= n_classes: -inf
        if (c + 3 >= n_classes) {
            if (c + 0 >= n_classes) v.x = ninf;
            if (c + 1 >= n_classes) v.y = ninf;
            if (c + 2 >= n_classes) v.z = ninf;
            if (c + 3 >= n_classes) v.w = ninf;
        }
        return v;
    };
    float4 v[RESIDENT ? HL_MAXJ : 1];
    float mx = ninf;
    if (RESIDENT) {
#pragma unroll
        for (int j = 0; j < HL_MAXJ; j++) {
            const int i4 = threadIdx.x + HL_THREADS * j;
            v[j] = make_float4(ninf, ninf, ninf, ninf);
            if (i4 < n4) v[j] = masked(reinterpret_cast<const float4 *>(zr)[i4], 4 * i4);
            mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
        }
    } else {
        for (int i4 = threadIdx.x; i4 < n4; i4 += HL_THREADS) {
            const float4 a = masked(reinterpret_cast<const float4 *>(zr)[i4], 4 * i4);
            mx = fmaxf(mx, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        }
    }
    mx = hl_block_max(mx, scratch);
    float sum = 0.0f;
    if (RESIDENT) {
#pragma unroll
        for (int j = 0; j < HL_MAXJ; j++) {
            v[j].x = __expf(v[j].x - mx); v[j].y = __expf(v[j].y - mx);
            v[j].z = __expf(v[j].z - mx); v[j].w = __expf(v[j].w - mx);
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    } else {
        for (int i4 = threadIdx.x; i4 < n4; i4 += HL_THREADS) {
            const float4 a = masked(reinterpret_cast<const float4 *>(zr)[i4], 4 * i4);
            sum += (__expf(a.x - mx) + __expf(a.y - mx)) + (__expf(a.z - mx) + __expf(a.w - mx));
        }
    }
    sum = hl_block_sum(sum, scratch);
    if (threadIdx.x == 0) loss[row] = counted ? (__logf(sum) + mx) - zt : 0.0f;
    const float w = scale / sum;
    const int tc = counted ? (int)t : -1;
    auto store = [&](float4 e, int i4) {
        const int c = 4 * i4;
        float4 g = make_float4(e.x * w, e.y * w, e.z * w, e.w * w);   // (exp(-inf) = 0: pad columns)
        if (tc >= c && tc < c + 4) {
            if (tc == c) g.x -= scale;
            else if (tc == c + 1) g.y -= scale;
            else if (tc == c + 2) g.z -= scale;
            else g.w -= scale;
        }
        reinterpret_cast<float4 *>(zr)[i4] = g;
    };
    if (RESIDENT) {
#pragma unroll
        for (int j = 0; j < HL_MAXJ; j++) {
            const int i4 = threadIdx.x + HL_THREADS * j;
            if (i4 < n4) store(v[j], i4);
        }
    } else {
        for (int i4 = threadIdx.x; i4 < n4; i4 += HL_THREADS) {
            const float4 a = masked(reinterpret_cast<const float4 *>(zr)[i4], 4 * i4);
            store(make_float4(__expf(a.x - mx), __expf(a.y - mx), __expf(a.z - mx), __expf(a.w - mx)), i4);
        }
    }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_cross_entropy_grad(float *logits, long long ld, long long rows, int n_classes,
                                      const long long *target, const float *scale, float *loss,
                                      long long ignore_index, void *stream) {
    if (!logits || !target || !scale || !loss) return SPT_EINVAL;
    if (rows <= 0 || n_classes <= 0 || ld < n_classes) return SPT_EINVAL;
    if (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(logits) & 15) != 0) return SPT_ESHAPE;
    if (rows > 0x7FFFFFFFll || ld > 0x7FFFFFFFll) return SPT_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    if (ld <= 4ll * HL_THREADS * HL_MAXJ)
        hipLaunchKernelGGL(cross_entropy_grad_kernel<true>, dim3((unsigned)rows), dim3(HL_THREADS), 0, s,
                           logits, ld, n_classes, target, scale, loss, ignore_index);
    else
        hipLaunchKernelGGL(cross_entropy_grad_kernel<false>, dim3((unsigned)rows), dim3(HL_THREADS), 0, s,
                           logits, ld, n_classes, target, scale, loss, ignore_index);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
