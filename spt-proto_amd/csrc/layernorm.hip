// layernorm.hip -- LayerNorm of the residual stream, forward and backward, with the residual
// additions on either side of it folded in.
//
// Reference: the pre-norm wiring of naive_gpt/layers/basic/transformer.py:46-52
//     x = x + mha(norm1(x));  x = x + ffd(norm2(x))
// with nn.LayerNorm prototypes (models/opt.py) or LLaMA's RMSNorm (models/llama.py; the RMS flag).  Each norm sits between two elementwise
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

// NV = d / 256: float4 per lane.  RMS: LLaMA's RMSNorm (naive_gpt/layers/basic/utils.py:22-37 of the
// reference): no mean, no beta -- y = s rsqrt(mean(s^2) + eps) gamma.  GREG: gamma (and beta) held in
// registers across the rows a wave walks (d <= 2048); re-read per row from L1 above that.
template <int NV, bool ADD, bool RMS>
__global__ __launch_bounds__(LN_THREADS) void add_layernorm_forward_kernel(
    const float *__restrict__ x, const float *__restrict__ r, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ s, float *__restrict__ y,
    float *__restrict__ mean, float *__restrict__ rstd, long long rows, float eps) {
    constexpr int D = NV * 256;
    constexpr bool GREG = NV <= 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 gm[GREG ? NV : 1], bt[GREG && !RMS ? NV : 1];
    if (GREG) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            gm[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
            if (!RMS) bt[j] = reinterpret_cast<const float4 *>(beta)[lane + 64 * j];
        }
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
        float mu = 0.f;
        if (!RMS) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NV; j++) sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            mu = ln_wave_sum(sum) * (1.0f / D);
        }
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
            const float4 g4 = GREG ? gm[j] : reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RMS) b4 = GREG ? bt[j] : reinterpret_cast<const float4 *>(beta)[lane + 64 * j];
            float4 o;
            o.x = fmaf((v[j].x - mu) * rs, g4.x, b4.x);
            o.y = fmaf((v[j].y - mu) * rs, g4.y, b4.y);
            o.z = fmaf((v[j].z - mu) * rs, g4.z, b4.z);
            o.w = fmaf((v[j].w - mu) * rs, g4.w, b4.w);
            yp[lane + 64 * j] = o;
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// partial: [gridDim.x][2][D] (dgamma row, dbeta row -- zeros for RMS) per workgroup
template <int NV, bool SKIP, bool RMS>
__global__ __launch_bounds__(LN_THREADS) void layernorm_backward_kernel(
    const float *__restrict__ s, const float *__restrict__ dy, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ dskip, float *__restrict__ dx, float *__restrict__ partial,
    long long rows) {
    constexpr int D = NV * 256;
    constexpr bool GREG = NV <= 8;
    __shared__ float4 acc[LN_WAVES][2][64];          // one 64-lane piece of the rows at a time
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 gm[GREG ? NV : 1], dg[NV], db[RMS ? 1 : NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        if (GREG) gm[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!RMS) db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long row = (long long)blockIdx.x * LN_WAVES + wave; row < rows;
         row += (long long)gridDim.x * LN_WAVES) {
        const float4 *sp = reinterpret_cast<const float4 *>(s + row * D);
        const float4 *gp = reinterpret_cast<const float4 *>(dy + row * D);
        const float mu = RMS ? 0.0f : mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
        // d <= 2048: the row's s and dy stay in registers between the reductions and the write;
        // above that (128 registers of row data beside the 64 of dgamma spill) they are read a
        // second time -- the row was just read, the second read is an L2 hit.
        constexpr bool KEEP = NV <= 8;
        float4 xh[KEEP ? NV : 1], g[KEEP ? NV : 1];
#pragma unroll
        for (int j = 0; j < NV; j++) {
            float4 x4 = sp[lane + 64 * j], g4r = gp[lane + 64 * j];
            const float4 gm4 = GREG ? gm[j] : reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
            x4.x = (x4.x - mu) * rs; x4.y = (x4.y - mu) * rs; x4.z = (x4.z - mu) * rs; x4.w = (x4.w - mu) * rs;
            // parameter gradients take dy itself, the input gradient dy * gamma
            dg[j].x = fmaf(g4r.x, x4.x, dg[j].x); dg[j].y = fmaf(g4r.y, x4.y, dg[j].y);
            dg[j].z = fmaf(g4r.z, x4.z, dg[j].z); dg[j].w = fmaf(g4r.w, x4.w, dg[j].w);
            if (!RMS) { db[j].x += g4r.x; db[j].y += g4r.y; db[j].z += g4r.z; db[j].w += g4r.w; }
            g4r.x *= gm4.x; g4r.y *= gm4.y; g4r.z *= gm4.z; g4r.w *= gm4.w;
            if (!RMS) c1 += (g4r.x + g4r.y) + (g4r.z + g4r.w);
            c2 += (g4r.x * x4.x + g4r.y * x4.y) + (g4r.z * x4.z + g4r.w * x4.w);
            if (KEEP) { xh[j] = x4; g[j] = g4r; }
            // (wide rows: keep the compiler from hoisting all 32 loads of the unrolled loop -- 128
            // registers -- in front of the arithmetic)
            if (!KEEP && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (!RMS) c1 = ln_wave_sum(c1) * (1.0f / D);
        c2 = ln_wave_sum(c2) * (1.0f / D);
        float4 *op = reinterpret_cast<float4 *>(dx + row * D);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            float4 x4, g4r;
            if (KEEP) {
                x4 = xh[j]; g4r = g[j];
            } else {
                x4 = sp[lane + 64 * j]; g4r = gp[lane + 64 * j];
                const float4 gm4 = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
                x4.x = (x4.x - mu) * rs; x4.y = (x4.y - mu) * rs; x4.z = (x4.z - mu) * rs; x4.w = (x4.w - mu) * rs;
                g4r.x *= gm4.x; g4r.y *= gm4.y; g4r.z *= gm4.z; g4r.w *= gm4.w;
            }
            float4 o;
            o.x = rs * (g4r.x - c1 - x4.x * c2); o.y = rs * (g4r.y - c1 - x4.y * c2);
            o.z = rs * (g4r.z - c1 - x4.z * c2); o.w = rs * (g4r.w - c1 - x4.w * c2);
            if (SKIP) {
                const float4 k4 = reinterpret_cast<const float4 *>(dskip + row * D)[lane + 64 * j];
                o.x += k4.x; o.y += k4.y; o.z += k4.z; o.w += k4.w;
            }
            op[lane + 64 * j] = o;
            if (!KEEP && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the workgroup's eight waves -> one partial row pair, in a fixed order, a 64-lane piece at a time
    float4 *out = reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * 2 * D);
#pragma unroll
    for (int j = 0; j < NV; j++) {
        acc[wave][0][lane] = dg[j];
        acc[wave][1][lane] = RMS ? make_float4(0.f, 0.f, 0.f, 0.f) : db[RMS ? 0 : j];
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, l = threadIdx.x & 63;
            float4 t = acc[0][which][l];
#pragma unroll
            for (int w = 1; w < LN_WAVES; w++) {
                const float4 u = acc[w][which][l];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            out[which * (D / 4) + l + 64 * j] = t;
        }
        __syncthreads();
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
        // (eight loads in flight: two at a time the 512 partial rows were 16 dependent round trips)
        for (; p + 112 < nparts; p += 128) {
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = partial[(size_t)(p + 16 * i) * width + col];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { a0 += t[i]; a1 += t[i + 1]; }
        }
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

static bool ln_width_ok(int d, int rms) {
    return d == 256 || d == 512 || d == 1024 || d == 2048 || (rms && d == 4096);
}

extern "C" int spt_add_layernorm_forward(const float *x, const float *r, const float *gamma,
                                         const float *beta, float *s, float *y, float *mean,
                                         float *rstd, long long rows, int d, float eps, int rms,
                                         void *stream) {
    if (!x || !gamma || (!rms && !beta) || !y || !mean || !rstd || (r && !s)) return SPT_EINVAL;
    if (rows <= 0 || d <= 0) return SPT_EINVAL;
    if (!ln_width_ok(d, rms)) return SPT_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ln_blocks(rows)), block(LN_THREADS);
#define SPT_LNF2(NV, RMS)                                                                        \
    do {                                                                                         \
        if (r) hipLaunchKernelGGL((add_layernorm_forward_kernel<NV, true, RMS>), grid, block, 0, st, \
                                  x, r, gamma, beta, s, y, mean, rstd, rows, eps);               \
        else hipLaunchKernelGGL((add_layernorm_forward_kernel<NV, false, RMS>), grid, block, 0, st, \
                                x, r, gamma, beta, s, y, mean, rstd, rows, eps);                 \
    } while (0)
#define SPT_LNF(NV)                                  \
    do {                                             \
        if (rms) SPT_LNF2(NV, true);                 \
        else SPT_LNF2(NV, false);                    \
    } while (0)
    switch (d / 256) {
        case 1: SPT_LNF(1); break;
        case 2: SPT_LNF(2); break;
        case 4: SPT_LNF(4); break;
        case 8: SPT_LNF(8); break;
        default: SPT_LNF2(16, true); break;
    }
#undef SPT_LNF
#undef SPT_LNF2
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_layernorm_backward(const float *s, const float *dy, const float *gamma,
                                      const float *mean, const float *rstd, const float *dskip,
                                      float *dx, float *dgamma, float *dbeta, float *partial,
                                      long long rows, int d, int rms, void *stream) {
    if (!s || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !partial)
        return SPT_EINVAL;
    if (rows <= 0 || d <= 0) return SPT_EINVAL;
    if (!ln_width_ok(d, rms)) return SPT_EUNSUP;
    if (dbeta != dgamma + d) return SPT_EINVAL;        // one [2, d] buffer: reduced in one launch
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ln_blocks(rows);
    const dim3 grid(nblk), block(LN_THREADS);
#define SPT_LNB2(NV, RMS)                                                                        \
    do {                                                                                         \
        if (dskip) hipLaunchKernelGGL((layernorm_backward_kernel<NV, true, RMS>), grid, block, 0, st, \
                                      s, dy, gamma, mean, rstd, dskip, dx, partial, rows);       \
        else hipLaunchKernelGGL((layernorm_backward_kernel<NV, false, RMS>), grid, block, 0, st, \
                                s, dy, gamma, mean, rstd, dskip, dx, partial, rows);             \
    } while (0)
#define SPT_LNB(NV)                                  \
    do {                                             \
        if (rms) SPT_LNB2(NV, true);                 \
        else SPT_LNB2(NV, false);                    \
    } while (0)
    switch (d / 256) {
        case 1: SPT_LNB(1); break;
        case 2: SPT_LNB(2); break;
        case 4: SPT_LNB(4); break;
        case 8: SPT_LNB(8); break;
        default: SPT_LNB2(16, true); break;
    }
#undef SPT_LNB
#undef SPT_LNB2
    SPT_LAUNCH_CHECK();
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((2 * d + 15) / 16), dim3(256), 0, st,
                       partial, dgamma, 2 * d, nblk);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
