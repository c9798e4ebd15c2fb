// grouped_gemm.hip -- token-bucketed grouped GEMM for the routed FFN, fp32 on MFMA.
//
// The reference evaluates a routed FFN with a Python loop over blocks, boolean-mask
// gathers (one device->host sync per block) and cuBLAS calls on the gathered rows
// (naive_gpt/layers/tuning/lora_ffn.py:87-111, layers/sparse/feedforward.py:66-85).
// Here the (token, block) pairs are sorted by block on the device; bucket g is the row
// range [offsets[g], offsets[g+1]) of the sorted order and ONE launch multiplies every
// bucket by its own weight block:
//
//     out[p, n] = rowscale[p] * ( sum_k A[src(p), k] * W_g(n, k)  +  bias[g, n] )
//     src(p) = gather ? gather[p] : p          (fuses the token gather)
//     W_g(n, k) = w[g * gstride + n * ldn + k * ldk]
//         ldk == 1 : "BT" weights, k contiguous  (forward: x.W1_g^T, h.W2_g^T)
//         ldn == 1 : "BN" weights, n contiguous  (backward: dY.W2_g, dH.W1_g)
//
// Bucket sizes never visit the host: the grid is sized for the worst case
// (ceil(P/128) + G row tiles) and every workgroup finds its bucket from `offsets`.
//
// This is the one place of the hot path where the contraction is dense, so it runs on
// the matrix cores: v_mfma_f32_32x32x2_f32, exact fp32 (a k-ordered fmaf chain), 157 TF
// peak (MI355X_MICROARCH "Matrix cores").  128 x 128 output tile per 256-thread
// workgroup, each wave a 64 x 64 quadrant (2 x 2 MFMA tiles, 64 accumulator registers);
// K is consumed in steps of 32 through LDS.  LDS image of an operand tile: [row][2][16]
// floats -- the k's of one parity contiguous -- because lane l of the MFMA needs
// k = k0 + (l >> 5): one ds_read_b128 then feeds four consecutive MFMAs.  Rows are
// padded by 16 bytes, which makes the b128 reads of 32 consecutive rows conflict-free.
#include "spt_common.h"

namespace spt {

constexpr int GG_THREADS = 256;
constexpr int GG_BM = 128;
constexpr int GG_BN = 128;
constexpr int GG_BK = 32;
constexpr int GG_ROW = GG_BK + 4;    // floats per LDS row of a k-contiguous tile (16-byte pad)
constexpr int GG_BNROW = GG_BN + 4;  // floats per LDS row of an n-contiguous weight tile

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GroupedArgs {
    const float *a;         // [*, K] row-major, leading dimension lda
    const int32_t *gather;  // [P] or null
    const float *w;
    const float *bias;      // [G, N] or null
    const float *rowscale;  // [P] or null
    const int32_t *offsets; // [G + 1]
    float *out;             // [P, N] row-major
    int P, K, N, G;
    int lda;
    long long gstride;
    int ldn, ldk;
};

template <bool BN_LAYOUT>
__global__ __launch_bounds__(GG_THREADS) void grouped_gemm_kernel(GroupedArgs g) {
    __shared__ __attribute__((aligned(16))) float As[GG_BM * GG_ROW];
    __shared__ __attribute__((aligned(16))) float Bs[GG_BN * GG_ROW];

    // ---- which bucket / row tile is this workgroup? ----
    int bucket = -1, row_lo = 0, row_hi = 0;
    {
        int tile = blockIdx.x;
        for (int i = 0; i < g.G; i++) {
            const int lo = g.offsets[i], hi = g.offsets[i + 1];
            const int tiles = (hi - lo + GG_BM - 1) / GG_BM;
            if (tile < tiles) {
                bucket = i;
                row_lo = lo + tile * GG_BM;
                row_hi = min(hi, row_lo + GG_BM);
                break;
            }
            tile -= tiles;
        }
    }
    if (bucket < 0) return;  // uniform for the workgroup
    const int n0 = blockIdx.y * GG_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * 64;  // quadrant origin inside the tile
    const int wn = (wave & 1) * 64;
    const float *wg = g.w + (size_t)bucket * g.gstride;

    // ---- staging assignment: tile = 128 rows x 8 float4 along k ----
    // A (and BT weights): thread -> row (tid >> 3) + 32 u, k-quad tid & 7
    const int s_row = tid >> 3, s_kq = tid & 7;
    const float *a_src[4];
    bool a_ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int p = row_lo + s_row + 32 * u;
        a_ok[u] = p < row_hi;
        const int src = a_ok[u] ? (g.gather ? g.gather[p] : p) : 0;
        a_src[u] = g.a + (size_t)src * g.lda;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // ---- software pipeline: the global loads of tile t+1 are in flight while the MFMAs
    // of tile t run; registers -> LDS happens at the top of the next iteration ----
    float4 av[4], bv[4];
    auto load_tile = [&](int k0) {
        const int k = k0 + 4 * s_kq;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok[u] && k < g.K) av[u] = *reinterpret_cast<const float4 *>(a_src[u] + k);
        }
        if (!BN_LAYOUT) {
            // W_g(n, k), k contiguous: same shape as A
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int n = n0 + s_row + 32 * u;
                bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < g.N && k < g.K)
                    bv[u] = *reinterpret_cast<const float4 *>(wg + (size_t)n * g.ldn + k);
            }
        } else {
            // W_g(n, k), n contiguous: thread -> k row (tid >> 5) + 8 u, n-quad tid & 31
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int kk = k0 + (tid >> 5) + 8 * u;
                const int n = n0 + 4 * (tid & 31);
                bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < g.K && n < g.N)
                    bv[u] = *reinterpret_cast<const float4 *>(wg + (size_t)kk * g.ldk + n);
            }
        }
    };
    load_tile(0);

    for (int k0 = 0; k0 < g.K; k0 += GG_BK) {
        __syncthreads();  // previous tile fully consumed
        // ---- registers -> LDS (parity-split rows) ----
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = s_row + 32 * u;
            // k = 4 kq .. 4 kq + 3  ->  parity 0: (x, z) at [2 kq, 2 kq + 1]; parity 1: (y, w)
            *reinterpret_cast<float2 *>(&As[r * GG_ROW + 2 * s_kq]) = make_float2(av[u].x, av[u].z);
            *reinterpret_cast<float2 *>(&As[r * GG_ROW + GG_BK / 2 + 2 * s_kq]) =
                make_float2(av[u].y, av[u].w);
        }
        if (!BN_LAYOUT) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = s_row + 32 * u;
                *reinterpret_cast<float2 *>(&Bs[r * GG_ROW + 2 * s_kq]) =
                    make_float2(bv[u].x, bv[u].z);
                *reinterpret_cast<float2 *>(&Bs[r * GG_ROW + GG_BK / 2 + 2 * s_kq]) =
                    make_float2(bv[u].y, bv[u].w);
            }
        } else {
            // n-contiguous weights keep their orientation in LDS: Bs[k][n], rows of
            // GG_BN + 4 floats (a parity-split image would need 4-byte writes 4 rows
            // apart: 16-way bank conflicts)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int kk = (tid >> 5) + 8 * u;
                *reinterpret_cast<float4 *>(&Bs[kk * GG_BNROW + 4 * (tid & 31)]) = bv[u];
            }
        }
        __syncthreads();
        if (k0 + GG_BK < g.K) load_tile(k0 + GG_BK);

        // ---- MFMA: lane l holds A[row = l & 31][k = kk + (l >> 5)], B likewise ----
        const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int q = 0; q < GG_BK / 8; q++) {   // 8 k's (4 per parity) per iteration
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; i++)
                af[i] = *reinterpret_cast<const float4 *>(
                    &As[(wm + 32 * i + frow) * GG_ROW + fh * (GG_BK / 2) + 4 * q]);
            float b0[4], b1[4];
            if (!BN_LAYOUT) {
#pragma unroll
                for (int j = 0; j < 2; j++)
                    bf[j] = *reinterpret_cast<const float4 *>(
                        &Bs[(wn + 32 * j + frow) * GG_ROW + fh * (GG_BK / 2) + 4 * q]);
                b0[0] = bf[0].x; b0[1] = bf[0].y; b0[2] = bf[0].z; b0[3] = bf[0].w;
                b1[0] = bf[1].x; b1[1] = bf[1].y; b1[2] = bf[1].z; b1[3] = bf[1].w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int kk = 8 * q + 2 * e + fh;
                    b0[e] = Bs[kk * GG_BNROW + wn + frow];
                    b1[e] = Bs[kk * GG_BNROW + wn + 32 + frow];
                }
            }
            const float a0[4] = {af[0].x, af[0].y, af[0].z, af[0].w};
            const float a1[4] = {af[1].x, af[1].y, af[1].z, af[1].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31] ----
    const int ccol = lane & 31, chalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int p = row_lo + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * chalf;
            if (p >= row_hi) continue;
            const float rs = g.rowscale ? g.rowscale[p] : 1.0f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int n = n0 + wn + 32 * j + ccol;
                if (n < g.N) {
                    float v = acc[i][j][r];
                    if (g.bias) v += g.bias[(size_t)bucket * g.N + n];
                    g.out[(size_t)p * g.N + n] = rs * v;
                }
            }
        }
    }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_grouped_gemm(const float *a, const int32_t *gather, const float *w,
                                const float *bias, const float *rowscale,
                                const int32_t *offsets, float *out, int n_rows, int k, int n,
                                int n_groups, int lda, long long w_group_stride, int w_ldn,
                                int w_ldk, void *stream) {
    if (!a || !w || !offsets || !out) return SPT_EINVAL;
    if (n_rows <= 0 || k <= 0 || n <= 0 || n_groups <= 0 || lda < k) return SPT_EINVAL;
    if (k % 4 != 0 || lda % 4 != 0) return SPT_ESHAPE;       // float4 rows of A
    if (w_ldk != 1 && w_ldn != 1) return SPT_EUNSUP;
    if (w_ldk == 1 && (w_ldn % 4 != 0 || w_group_stride % 4 != 0)) return SPT_ESHAPE;
    if (w_ldk != 1 && (w_ldk % 4 != 0 || n % 4 != 0 || w_group_stride % 4 != 0)) return SPT_ESHAPE;
    GroupedArgs g;
    g.a = a; g.gather = gather; g.w = w; g.bias = bias; g.rowscale = rowscale;
    g.offsets = offsets; g.out = out;
    g.P = n_rows; g.K = k; g.N = n; g.G = n_groups; g.lda = lda;
    g.gstride = w_group_stride; g.ldn = w_ldn; g.ldk = w_ldk;
    const unsigned row_tiles = (unsigned)((n_rows + GG_BM - 1) / GG_BM + n_groups);
    const unsigned col_tiles = (unsigned)((n + GG_BN - 1) / GG_BN);
    if (col_tiles > 65535) return SPT_EUNSUP;
    dim3 grid(row_tiles, col_tiles);
    hipStream_t s = (hipStream_t)stream;
    if (w_ldk == 1) hipLaunchKernelGGL((grouped_gemm_kernel<false>), grid, dim3(GG_THREADS), 0, s, g);
    else hipLaunchKernelGGL((grouped_gemm_kernel<true>), grid, dim3(GG_THREADS), 0, s, g);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
