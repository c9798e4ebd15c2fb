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
// Selection order: larger probability first, ties to the lower block index (a total order;
// torch.topk leaves ties unspecified).  G <= 8, T <= 65536 (64 tokens per thread); bigger
// problems use the torch composition (layers/sparse/grouped.py).
#include "spt_common.h"

namespace spt {

constexpr int RT_THREADS = 1024;
constexpr int RT_MAXG = 8;

__device__ __forceinline__ unsigned select_topk(const float *__restrict__ p, int G, int k) {
    float v[RT_MAXG];
#pragma unroll
    for (int j = 0; j < RT_MAXG; j++) v[j] = j < G ? p[j] : 0.0f;
    unsigned mask = 0u;
#pragma unroll
    for (int j = 0; j < RT_MAXG; j++) {
        int rank = 0;
#pragma unroll
        for (int i = 0; i < RT_MAXG; i++)
            rank += (i < G) && (v[i] > v[j] || (v[i] == v[j] && i < j));
        if (j < G && rank < k) mask |= 1u << j;
    }
    return mask;
}

__global__ __launch_bounds__(RT_THREADS) void route_topk_kernel(
    const float *__restrict__ prob, int32_t *__restrict__ token, int32_t *__restrict__ block,
    int32_t *__restrict__ offsets, int32_t *__restrict__ pos, int T, int G, int k) {
    __shared__ int wave_tot[RT_THREADS / 64][RT_MAXG];
    __shared__ int base[RT_MAXG + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tpt = (T + RT_THREADS - 1) / RT_THREADS;
    const int t0 = min(T, tid * tpt), t1 = min(T, t0 + tpt);

    int cnt[RT_MAXG];
#pragma unroll
    for (int g = 0; g < RT_MAXG; g++) cnt[g] = 0;
    for (int t = t0; t < t1; t++) {
        const unsigned m = select_topk(prob + (size_t)t * G, G, k);
#pragma unroll
        for (int g = 0; g < RT_MAXG; g++) cnt[g] += (m >> g) & 1u;
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
    for (int t = t0; t < t1; t++) {
        const unsigned m = select_topk(prob + (size_t)t * G, G, k);
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
    hipLaunchKernelGGL(route_topk_kernel, dim3(1), dim3(RT_THREADS), 0, (hipStream_t)stream, prob,
                       token, block, offsets, pos, n_tokens, n_blocks, k);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
