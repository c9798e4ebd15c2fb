// Rotary position embedding of q and k in one pass (reference: naive_gpt/layers/basic/position.py:24-34,
// called twice per attention from attention.py:54-60):
//     y = cos[s] * x + sin[s] * rotate_half(x),   rotate_half(x) = [-x_hi | x_lo]
// over x [N, S, H, E] with tables [>= S, E] (row s: position s).  As library operators the two calls
// are ~8 launches forward (chunk, neg, cat, two products, a sum, twice) and as many backward, each
// over 33 MB at the LLaMA-7B block shape.  Here: up to three tensors per launch -- the first n_rot
// rotated, the rest copied -- written into ONE buffer `out_stride` floats apart:
//   forward   (q, k) -> [q' ; k'] back to back (what the PQ loss and the lookup read as a pair)
//   backward  (dq', dk', dv) -> [dq ; dk ; dv] equally spaced: the joint projection's backward then
//             contracts the three as one product (SptGroupedGemm.a_seg_k).  transpose != 0 applies
//             the adjoint:  dx_lo = cos_lo dy_lo + sin_hi dy_hi,  dx_hi = cos_hi dy_hi - sin_lo dy_lo.
// A thread owns a float4 of the low half and its partner in the high half.
#include "spt_common.h"

namespace spt {

struct RotaryParts {
    const float *x[3];
};

__global__ __launch_bounds__(256) void rotary_kernel(RotaryParts parts, float *__restrict__ out,
                                                     long long out_stride, const float *__restrict__ cos_t,
                                                     const float *__restrict__ sin_t, int n_rot, int n_parts,
                                                     long long rows, int S, int H, int E, int transpose) {
    const int q4 = E / 8;                                    // float4 pairs per row
    const long long total = rows * q4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / q4;                        // (n, s, h)
        const int c = (int)(i - row * q4) * 4;               // column in the low half
        const int s = (int)((row / H) % S);
        const size_t at = (size_t)row * E + c;
        const float4 cl = *reinterpret_cast<const float4 *>(cos_t + (size_t)s * E + c);
        const float4 ch = *reinterpret_cast<const float4 *>(cos_t + (size_t)s * E + E / 2 + c);
        const float4 sl = *reinterpret_cast<const float4 *>(sin_t + (size_t)s * E + c);
        const float4 sh = *reinterpret_cast<const float4 *>(sin_t + (size_t)s * E + E / 2 + c);
        for (int p = 0; p < n_parts; p++) {
            const float4 lo = *reinterpret_cast<const float4 *>(parts.x[p] + at);
            const float4 hi = *reinterpret_cast<const float4 *>(parts.x[p] + at + E / 2);
            float4 ol = lo, oh = hi;
            if (p < n_rot) {
                if (!transpose) {
                    ol = make_float4(cl.x * lo.x - sl.x * hi.x, cl.y * lo.y - sl.y * hi.y,
                                     cl.z * lo.z - sl.z * hi.z, cl.w * lo.w - sl.w * hi.w);
                    oh = make_float4(ch.x * hi.x + sh.x * lo.x, ch.y * hi.y + sh.y * lo.y,
                                     ch.z * hi.z + sh.z * lo.z, ch.w * hi.w + sh.w * lo.w);
                } else {
                    ol = make_float4(cl.x * lo.x + sh.x * hi.x, cl.y * lo.y + sh.y * hi.y,
                                     cl.z * lo.z + sh.z * hi.z, cl.w * lo.w + sh.w * hi.w);
                    oh = make_float4(ch.x * hi.x - sl.x * lo.x, ch.y * hi.y - sl.y * lo.y,
                                     ch.z * hi.z - sl.z * lo.z, ch.w * hi.w - sl.w * lo.w);
                }
            }
            float *o = out + (size_t)p * out_stride + at;
            *reinterpret_cast<float4 *>(o) = ol;
            *reinterpret_cast<float4 *>(o + E / 2) = oh;
        }
    }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_rotary(const float *const *x, int n_parts, int n_rot, float *out, long long out_stride,
                          const float *cos_table, const float *sin_table, int batch, int seq_length,
                          int n_heads, int d_head, int transpose, void *stream) {
    if (!x || !out || !cos_table || !sin_table) return SPT_EINVAL;
    if (n_parts <= 0 || n_parts > 3 || n_rot < 0 || n_rot > n_parts) return SPT_EINVAL;
    if (batch <= 0 || seq_length <= 0 || n_heads <= 0 || d_head <= 0) return SPT_EINVAL;
    if (d_head % 8 != 0) return SPT_ESHAPE;
    const long long rows = (long long)batch * seq_length * n_heads;
    if (out_stride < rows * d_head || out_stride % 4 != 0) return SPT_ESHAPE;
    RotaryParts parts = {};
    uintptr_t bits = reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(cos_table) |
                     reinterpret_cast<uintptr_t>(sin_table);
    for (int p = 0; p < n_parts; p++) {
        if (!x[p]) return SPT_EINVAL;
        parts.x[p] = x[p];
        bits |= reinterpret_cast<uintptr_t>(x[p]);
    }
    if (bits & 15) return SPT_ESHAPE;
    const long long total = rows * (d_head / 8);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(rotary_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, parts, out, out_stride, cos_table, sin_table, n_rot, n_parts, rows,
                       seq_length, n_heads, d_head, transpose);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
