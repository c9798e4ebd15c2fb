// routing.hip -- token -> expert-block buckets of the routed FFN in one launch.
//
// Reference: RoutedFFN.forward (naive_gpt/layers/sparse/feedforward.py:56-85) and its LoRA
// variants (layers/tuning/lora_ffn.py:87-111) take `topk(prob, k)` per token and then loop
// over the blocks with boolean masks (`x[mask]`, one device->host sync per block).  The
// grouped GEMMs want the (token, block) pairs sorted by block instead, stable in the token id
// (= the order of `x[mask]`).  With torch ops that is topk + argsort + bincount + cumsum +
// gathers, ~15 launches and ~130 us at T = 8192; here it is a counting sort in one workgroup:
//
//   pass 1  every thread selects the k largest of the G probabilities of its (contiguous)
//           tokens and counts its selections per block;
//   scan    exclusive prefix of the G counters over the 1024 threads (DPP wave scan + LDS);
//   pass 2  position of (token, block) = offsets[block] + prefix + running count.
//
// Selection order: larger probability first (NaN above every number, as torch.topk ranks it),
// ties to the lower block index (a total order; torch.topk leaves ties unspecified).  G <= 8, T <= 65536 (64 tokens per thread); bigger
// problems use the torch composition (layers/sparse/grouped.py).
#include "spt_common.h"

namespace spt {

constexpr int RT_THREADS = 1024;
constexpr int RT_MAXG = 8;

// Orderable key of a probability: ascending unsigned order == ascending float order, -0 == +0,
// and every NaN is the LARGEST key (torch.topk also ranks NaN above every number).  The rank
// below is then a total order for any input, so exactly k bits are set per token: with the
// float comparisons used before, a NaN compared false against everything, ranked 0 and let
// a token select more than k blocks -- writes past the T * k rows the caller allocated.
__device__ __forceinline__ unsigned order_key(float x) {
    if (x != x) return 0xFFFFFFFFu;
    const unsigned b = __float_as_uint(x + 0.0f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ unsigned select_topk(const float (&v)[RT_MAXG], int G, int k) {
    unsigned key[RT_MAXG];
#pragma unroll
    for (int j = 0; j < RT_MAXG; j++) key[j] = order_key(v[j]);
    unsigned mask = 0u;
#pragma unroll
    for (int j = 0; j < RT_MAXG; j++) {
        int rank = 0;
#pragma unroll
        for (int i = 0; i < RT_MAXG; i++)
            rank += (i < G) && (key[i] > key[j] || (key[i] == key[j] && i < j));
        if (j < G && rank < k) mask |= 1u << j;
    }
    return mask;
}

__global__ __launch_bounds__(RT_THREADS) void route_topk_kernel(
    const float *__restrict__ prob, int32_t *__restrict__ token, int32_t *__restrict__ block,
    int32_t *__restrict__ offsets, int32_t *__restrict__ pos, int T, int G, int k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t *sel = reinterpret_cast<uint8_t *>(smem);          // [T] selection masks
    __shared__ int wave_tot[RT_THREADS / 64][RT_MAXG];
    __shared__ int base[RT_MAXG + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tpt = (T + RT_THREADS - 1) / RT_THREADS;
    const int t0 = min(T, tid * tpt), t1 = min(T, t0 + tpt);

    // pass 1: selections (kept in LDS for pass 2) and per-thread counts.  The probabilities
    // of 8 tokens are requested before the first is used: a single workgroup has nothing
    // else to hide the load latency behind.
    int cnt[RT_MAXG];
#pragma unroll
    for (int g = 0; g < RT_MAXG; g++) cnt[g] = 0;
    for (int tb = t0; tb < t1; tb += 8) {
        float v[8][RT_MAXG];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = min(tb + u, t1 - 1);
            if (G == 4) {
                const float4 q = *reinterpret_cast<const float4 *>(prob + (size_t)t * 4);
                v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
#pragma unroll
                for (int j = 4; j < RT_MAXG; j++) v[u][j] = 0.0f;
            } else {
#pragma unroll
                for (int j = 0; j < RT_MAXG; j++) v[u][j] = j < G ? prob[(size_t)t * G + j] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (tb + u < t1) {
                const unsigned m = select_topk(v[u], G, k);
                sel[tb + u] = (uint8_t)m;
#pragma unroll
                for (int g = 0; g < RT_MAXG; g++) cnt[g] += (m >> g) & 1u;
            }
        }
    }
    // exclusive prefix over threads, per block
    int pre[RT_MAXG];
#pragma unroll
    for (int g = 0; g < RT_MAXG; g++) {
        int inc = cnt[g];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        pre[g] = inc - cnt[g];
        if (lane == 63) wave_tot[wave][g] = inc;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < G; g++) {
            base[g] = run;
            for (int w = 0; w < RT_THREADS / 64; w++) run += wave_tot[w][g];
        }
        base[G] = run;
        for (int g = 0; g <= G; g++) offsets[g] = base[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < RT_MAXG; g++) {
        int before = g < G ? base[g] : 0;
        for (int w = 0; w < wave; w++) before += wave_tot[w][g];
        pre[g] += before;
    }
    // pass 2: placement (a thread only reads the selections it wrote)
    for (int t = t0; t < t1; t++) {
        const unsigned m = sel[t];
        int j = 0;
#pragma unroll
        for (int g = 0; g < RT_MAXG; g++) {
            if ((m >> g) & 1u) {
                const int r = pre[g]++;
                token[r] = t;
                block[r] = g;
                pos[(size_t)t * k + j] = r;
                j++;
            }
        }
    }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_route_topk(const float *prob, int32_t *token, int32_t *block,
                              int32_t *offsets, int32_t *pos, int n_tokens, int n_blocks,
                              int k, void *stream) {
    if (!prob || !token || !block || !offsets || !pos) return SPT_EINVAL;
    if (n_tokens <= 0 || n_blocks <= 0 || k <= 0 || k > n_blocks) return SPT_EINVAL;
    if (n_blocks > RT_MAXG || n_tokens > 64 * RT_THREADS) return SPT_EUNSUP;
    const size_t lds = ((size_t)n_tokens + 15) & ~(size_t)15;
    if (lds > 64 * 1024)
        SPT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&route_topk_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(route_topk_kernel, dim3(1), dim3(RT_THREADS), lds, (hipStream_t)stream,
                       prob, token, block, offsets, pos, n_tokens, n_blocks, k);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
