// layernorm.hip -- LayerNorm of the residual stream, forward and backward, with the residual
// additions on either side of it folded in.
//
// Reference: the pre-norm wiring of naive_gpt/layers/basic/transformer.py:46-52
//     x = x + mha(norm1(x));  x = x + ffd(norm2(x))
// with nn.LayerNorm prototypes (models/opt.py).  Each norm sits between two elementwise
// additions -- forward: the sum it normalises; backward: the skip path's gradient joining the
// norm's -- and all of it is HBM-bound traffic over [tokens, d_model] fp32 tensors (33 MB each at
// BERT-large): as library operators 17 us forward, 35 + 21 us backward and 11 us per addition,
// 8.5 ms of a 71 ms fine-tune step.  Here:
//
//   forward   s = x (+ r);  mean, rstd over the row;  y = (s - mean) rstd gamma + beta
//             one read of x (and r), one write of y (and s)
//   backward  xhat = (s - mean) rstd;  g = dy gamma;
//             dx = rstd (g - mean(g) - xhat mean(g xhat)) (+ dskip)
//             dgamma = sum_rows dy xhat, dbeta = sum_rows dy: per-workgroup partial rows, summed in
//             a fixed order by a second small kernel (no atomics: the step is reproducible)
//
// A wave owns a row: d / 256 float4 per lane for every operand (coalesced 1 KiB per instruction),
// gamma / beta and the parameter-gradient accumulators live in registers for all the rows the wave
// walks.  Same arithmetic as torch.native_layer_norm (two-pass mean / variance, biased variance,
// rsqrt(var + eps)); row sums are butterfly trees, so results differ from the library's in the
// last bits only.
#include "spt_common.h"

namespace spt {

constexpr int LN_WAVES = 8;                       // rows in flight per workgroup
constexpr int LN_THREADS = LN_WAVES * SPT_WAVE;

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, SPT_WAVE);
    return v;
}

// NV = d / 256: float4 per lane
template <int NV, bool ADD>
__global__ __launch_bounds__(LN_THREADS) void add_layernorm_forward_kernel(
    const float *__restrict__ x, const float *__restrict__ r, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ s, float *__restrict__ y,
    float *__restrict__ mean, float *__restrict__ rstd, long long rows, float eps) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 gm[NV], bt[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        gm[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        bt[j] = reinterpret_cast<const float4 *>(beta)[lane + 64 * j];
    }
    for (long long row = (long long)blockIdx.x * LN_WAVES + wave; row < rows;
         row += (long long)gridDim.x * LN_WAVES) {
        const float4 *xp = reinterpret_cast<const float4 *>(x + row * D);
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = xp[lane + 64 * j];
        if (ADD) {
            const float4 *rp = reinterpret_cast<const float4 *>(r + row * D);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const float4 t = rp[lane + 64 * j];
                v[j].x += t.x; v[j].y += t.y; v[j].z += t.z; v[j].w += t.w;
            }
            float4 *sp = reinterpret_cast<float4 *>(s + row * D);
#pragma unroll
            for (int j = 0; j < NV; j++) sp[lane + 64 * j] = v[j];
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mu = ln_wave_sum(sum) * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, e = v[j].w - mu;
            sq += (a * a + b * b) + (c * c + e * e);
        }
        const float rs = rsqrtf(ln_wave_sum(sq) * (1.0f / D) + eps);
        float4 *yp = reinterpret_cast<float4 *>(y + row * D);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            float4 o;
            o.x = fmaf((v[j].x - mu) * rs, gm[j].x, bt[j].x);
            o.y = fmaf((v[j].y - mu) * rs, gm[j].y, bt[j].y);
            o.z = fmaf((v[j].z - mu) * rs, gm[j].z, bt[j].z);
            o.w = fmaf((v[j].w - mu) * rs, gm[j].w, bt[j].w);
            yp[lane + 64 * j] = o;
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// partial: [gridDim.x][2][D] (dgamma row, dbeta row) per workgroup
template <int NV, bool SKIP>
__global__ __launch_bounds__(LN_THREADS) void layernorm_backward_kernel(
    const float *__restrict__ s, const float *__restrict__ dy, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ dskip, float *__restrict__ dx, float *__restrict__ partial,
    long long rows) {
    constexpr int D = NV * 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *acc = reinterpret_cast<float4 *>(smem);           // [LN_WAVES][2][D / 4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 gm[NV], dg[NV], db[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        gm[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long row = (long long)blockIdx.x * LN_WAVES + wave; row < rows;
         row += (long long)gridDim.x * LN_WAVES) {
        const float4 *sp = reinterpret_cast<const float4 *>(s + row * D);
        const float4 *gp = reinterpret_cast<const float4 *>(dy + row * D);
        float4 xh[NV], g[NV], sk[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            xh[j] = sp[lane + 64 * j];
            g[j] = gp[lane + 64 * j];
            if (SKIP) sk[j] = reinterpret_cast<const float4 *>(dskip + row * D)[lane + 64 * j];
        }
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            xh[j].x = (xh[j].x - mu) * rs; xh[j].y = (xh[j].y - mu) * rs;
            xh[j].z = (xh[j].z - mu) * rs; xh[j].w = (xh[j].w - mu) * rs;
            // parameter gradients take dy itself, the input gradient dy * gamma
            dg[j].x = fmaf(g[j].x, xh[j].x, dg[j].x); dg[j].y = fmaf(g[j].y, xh[j].y, dg[j].y);
            dg[j].z = fmaf(g[j].z, xh[j].z, dg[j].z); dg[j].w = fmaf(g[j].w, xh[j].w, dg[j].w);
            db[j].x += g[j].x; db[j].y += g[j].y; db[j].z += g[j].z; db[j].w += g[j].w;
            g[j].x *= gm[j].x; g[j].y *= gm[j].y; g[j].z *= gm[j].z; g[j].w *= gm[j].w;
            c1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
            c2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
        }
        c1 = ln_wave_sum(c1) * (1.0f / D);
        c2 = ln_wave_sum(c2) * (1.0f / D);
        float4 *op = reinterpret_cast<float4 *>(dx + row * D);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            float4 o;
            o.x = rs * (g[j].x - c1 - xh[j].x * c2); o.y = rs * (g[j].y - c1 - xh[j].y * c2);
            o.z = rs * (g[j].z - c1 - xh[j].z * c2); o.w = rs * (g[j].w - c1 - xh[j].w * c2);
            if (SKIP) { o.x += sk[j].x; o.y += sk[j].y; o.z += sk[j].z; o.w += sk[j].w; }
            op[lane + 64 * j] = o;
        }
    }
    // the workgroup's eight waves -> one partial row pair, in a fixed order
#pragma unroll
    for (int j = 0; j < NV; j++) {
        acc[(wave * 2 + 0) * (D / 4) + lane + 64 * j] = dg[j];
        acc[(wave * 2 + 1) * (D / 4) + lane + 64 * j] = db[j];
    }
    __syncthreads();
    float4 *out = reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * 2 * D);
    for (int i = threadIdx.x; i < 2 * D / 4; i += LN_THREADS) {
        float4 t = acc[i];
#pragma unroll
        for (int w = 1; w < LN_WAVES; w++) {
            const float4 u = acc[w * 2 * (D / 4) + i];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        out[i] = t;
    }
}

// out[c] = sum over `nparts` rows of partial[., c], c < width (= 2 D): 16 columns x 16 row lanes
// per workgroup
__global__ __launch_bounds__(256) void layernorm_param_reduce_kernel(
    const float *__restrict__ partial, float *__restrict__ out, int width, int nparts) {
    __shared__ float red[16][17];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
    float a0 = 0.f, a1 = 0.f;
    if (col < width) {
        int p = rl;
        for (; p + 16 < nparts; p += 32) {
            a0 += partial[(size_t)p * width + col];
            a1 += partial[(size_t)(p + 16) * width + col];
        }
        if (p < nparts) a0 += partial[(size_t)p * width + col];
    }
    red[rl][threadIdx.x & 15] = a0 + a1;
    __syncthreads();
    if (rl == 0 && col < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) t += red[i][threadIdx.x & 15];
        out[col] = t;
    }
}

}  // namespace spt

using namespace spt;

static int ln_blocks(long long rows) {
    const long long want = (rows + LN_WAVES - 1) / LN_WAVES;
    return (int)(want < 512 ? want : 512);
}

extern "C" int spt_layernorm_partial_rows(long long rows) { return rows > 0 ? ln_blocks(rows) : 0; }

extern "C" int spt_add_layernorm_forward(const float *x, const float *r, const float *gamma,
                                         const float *beta, float *s, float *y, float *mean,
                                         float *rstd, long long rows, int d, float eps,
                                         void *stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || (r && !s)) return SPT_EINVAL;
    if (rows <= 0 || d <= 0) return SPT_EINVAL;
    if (d != 1024 && d != 2048 && d != 512 && d != 256) return SPT_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ln_blocks(rows)), block(LN_THREADS);
#define SPT_LNF(NV)                                                                              \
    do {                                                                                         \
        if (r) hipLaunchKernelGGL((add_layernorm_forward_kernel<NV, true>), grid, block, 0, st, x, r, \
                                  gamma, beta, s, y, mean, rstd, rows, eps);                      \
        else hipLaunchKernelGGL((add_layernorm_forward_kernel<NV, false>), grid, block, 0, st, x, r, \
                                gamma, beta, s, y, mean, rstd, rows, eps);                        \
    } while (0)
    switch (d / 256) {
        case 1: SPT_LNF(1); break;
        case 2: SPT_LNF(2); break;
        case 4: SPT_LNF(4); break;
        default: SPT_LNF(8); break;
    }
#undef SPT_LNF
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_layernorm_backward(const float *s, const float *dy, const float *gamma,
                                      const float *mean, const float *rstd, const float *dskip,
                                      float *dx, float *dgamma, float *dbeta, float *partial,
                                      long long rows, int d, void *stream) {
    if (!s || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !partial)
        return SPT_EINVAL;
    if (rows <= 0 || d <= 0) return SPT_EINVAL;
    if (d != 1024 && d != 2048 && d != 512 && d != 256) return SPT_EUNSUP;
    if (dbeta != dgamma + d) return SPT_EINVAL;        // one [2, d] buffer: reduced in one launch
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ln_blocks(rows);
    const dim3 grid(nblk), block(LN_THREADS);
    const size_t lds = (size_t)LN_WAVES * 2 * d * sizeof(float);
#define SPT_LNB(NV)                                                                              \
    do {                                                                                         \
        if (dskip) {                                                                             \
            SPT_HIP_TRY(hipFuncSetAttribute((const void *)layernorm_backward_kernel<NV, true>,   \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((layernorm_backward_kernel<NV, true>), grid, block, lds, st, s, dy, \
                               gamma, mean, rstd, dskip, dx, partial, rows);                     \
        } else {                                                                                 \
            SPT_HIP_TRY(hipFuncSetAttribute((const void *)layernorm_backward_kernel<NV, false>,  \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((layernorm_backward_kernel<NV, false>), grid, block, lds, st, s, dy, \
                               gamma, mean, rstd, dskip, dx, partial, rows);                     \
        }                                                                                        \
    } while (0)
    switch (d / 256) {
        case 1: SPT_LNB(1); break;
        case 2: SPT_LNB(2); break;
        case 4: SPT_LNB(4); break;
        default: SPT_LNB(8); break;
    }
#undef SPT_LNB
    SPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((2 * d + 15) / 16), dim3(256), 0, st,
                       partial, dgamma, 2 * d, nblk);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
