// routing.hip -- token -> expert-block buckets of the routed FFN in one launch.
//
// Reference: RoutedFFN.forward (naive_gpt/layers/sparse/feedforward.py:56-85) and its LoRA
// variants (layers/tuning/lora_ffn.py:87-111) take `topk(prob, k)` per token and then loop
// over the blocks with boolean masks (`x[mask]`, one device->host sync per block).  The
// grouped GEMMs want the (token, block) pairs sorted by block instead, stable in the token id
// (= the order of `x[mask]`).  With torch ops that is topk + argsort + bincount + cumsum +
// gathers, ~15 launches and ~130 us at T = 8192; here it is a counting sort in one launch:
//
//   select  a thread per token picks the k largest of its G probabilities (a bit mask);
//   count   ballots per block give the selections per wave; the sort position of a selection =
//           (selections of lower blocks) + (same block, earlier tokens) -- a function of the masks
//           of the earlier tokens alone, which every workgroup derives for itself (below);
//   place   token / block / pos written at that position.
//
// Selection order: larger probability first (NaN above every number, as torch.topk ranks it),
// ties to the lower block index (a total order; torch.topk leaves ties unspecified).  G <= 8, T <= 65536 (64 tokens per thread); bigger
// problems use the torch composition (layers/sparse/grouped.py).
#include "spt_common.h"

namespace spt {

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

template <int MG>
__device__ __forceinline__ unsigned select_topk(const float (&v)[MG], int G, int k) {
    unsigned key[MG];
#pragma unroll
    for (int j = 0; j < MG; j++) key[j] = order_key(v[j]);
    unsigned mask = 0u;
#pragma unroll
    for (int j = 0; j < MG; j++) {
        int rank = 0;
#pragma unroll
        for (int i = 0; i < MG; i++)
            rank += (i < G) && (key[i] > key[j] || (key[i] == key[j] && i < j));
        if (j < G && rank < k) mask |= 1u << j;
    }
    return mask;
}

// A workgroup places the selections of its own 256 tokens (a thread per token); what it needs
// of the other tokens -- how many selections per block precede its own -- it counts itself, from
// the probabilities: every workgroup ballots through ALL chunks (110 instructions per chunk and
// thread, 4 KiB of loads), which costs less than any exchange between workgroups would (a second
// launch, or a scratch buffer the C ABI would have to carry) and keeps the result a pure function
// of the input.  History: one workgroup of 1024 threads, 8 tokens per thread, selections parked in
// LDS: 39 us at T = 8192 (a single CU's latency chain).
// (MG = 4 or 8 block slots compiled in: the selection is a rank over MG x MG comparisons, and
// it is what the redundant counting repeats.)
constexpr int RT_CHUNK = 256;
template <int MG>
__global__ __launch_bounds__(RT_CHUNK) void route_topk_kernel(
    const float *__restrict__ prob, int32_t *__restrict__ token, int32_t *__restrict__ block,
    int32_t *__restrict__ offsets, int32_t *__restrict__ pos, int T, int G, int k,
    long long *__restrict__ token64, long long *__restrict__ block64, float *__restrict__ coeff,
    float scale, int ld, const float *__restrict__ bias, float *__restrict__ prob_out) {
    // (prob_out != null: `prob` holds the router's LOGITS without bias, rows `ld` floats apart --
    // the last block of spt_lora_down2.  The selection ranks logit + bias: the sigmoid is monotone, so
    // this is the ranking of the probabilities wherever two of them differ, and where distinct logits
    // round to ONE probability -- a saturated sigmoid; torch.topk leaves the pick among equal values
    // open -- the larger logit wins.  Only the chunk's owner forms sigmoid(logit + bias), for
    // prob_out and the coefficients: every workgroup doing so for every chunk it counts cost 10 us.)
    __shared__ int wave_before[RT_CHUNK / 64][MG];   // selections in chunks before mine, by wave
    __shared__ int wave_total[RT_CHUNK / 64][MG];    // ... in all chunks
    __shared__ int wave_mine[RT_CHUNK / 64][MG];     // ... in my chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mine = blockIdx.x, nchunks = (T + RT_CHUNK - 1) / RT_CHUNK;
    float bias_v[MG];
#pragma unroll
    for (int j = 0; j < MG; j++) bias_v[j] = (bias && j < G) ? bias[j] : 0.0f;
    auto load = [&](int c, float (&v)[MG]) {
        const int t = min(c * RT_CHUNK + tid, T - 1);           // (clamped: masked below)
        if (MG == 4 && G == 4 && (ld & 3) == 0) {
            const float4 q = *reinterpret_cast<const float4 *>(prob + (size_t)t * ld);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < MG; j++) v[j] = j < G ? prob[(size_t)t * ld + j] : 0.0f;
        }
        if (prob_out) {
#pragma unroll
            for (int j = 0; j < MG; j++) v[j] = j < G ? v[j] + bias_v[j] : 0.0f;
        }
    };
    int before[MG], total[MG], own[MG], rank[MG];   // wave-uniform counters
#pragma unroll
    for (int g = 0; g < MG; g++) before[g] = total[g] = own[g] = rank[g] = 0;
    unsigned my_mask = 0u;
    float my_v[MG];
#pragma unroll
    for (int g = 0; g < MG; g++) my_v[g] = 0.0f;
    const unsigned long long lower = (1ull << lane) - 1ull;
    constexpr int AHEAD = 4;                  // chunks whose loads are in flight together
    for (int c0 = 0; c0 < nchunks; c0 += AHEAD) {
        float v[AHEAD][MG];
#pragma unroll
        for (int u = 0; u < AHEAD; u++) load(min(c0 + u, nchunks - 1), v[u]);
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
            const int c = c0 + u;
            if (c >= nchunks) break;
            const unsigned m = (c * RT_CHUNK + tid < T) ? select_topk(v[u], G, k) : 0u;
            if (c == mine) {
                my_mask = m;
#pragma unroll                                                   // torch.sigmoid(linear(x, rw, rb))
                for (int g = 0; g < MG; g++)
                    my_v[g] = prob_out ? 1.0f / (1.0f + expf(-v[u][g])) : v[u][g];
            }
#pragma unroll
            for (int g = 0; g < MG; g++) {
                const unsigned long long b = __ballot((m >> g) & 1u);
                const int n = __popcll(b);
                total[g] += n;
                if (c < mine) before[g] += n;
                if (c == mine) {
                    own[g] = n;
                    rank[g] = __popcll(b & lower);
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < MG; g++) {
            wave_before[wave][g] = before[g];
            wave_total[wave][g] = total[g];
            wave_mine[wave][g] = own[g];
        }
    }
    __syncthreads();
    // position of this token's selection of block g: all selections of lower blocks, then those of
    // block g in earlier chunks, in lower waves of this chunk, in lower lanes of this wave
    int run = 0, place[MG];
#pragma unroll
    for (int g = 0; g < MG; g++) {
        int tot = 0, bef = 0;
#pragma unroll
        for (int w = 0; w < RT_CHUNK / 64; w++) {
            tot += wave_total[w][g];
            bef += wave_before[w][g] + (w < wave ? wave_mine[w][g] : 0);
        }
        place[g] = run + bef + rank[g];
        if (mine == 0 && tid == 0 && g < G) offsets[g] = run;
        run += g < G ? tot : 0;
    }
    if (mine == 0 && tid == 0) offsets[G] = run;
    const int t = mine * RT_CHUNK + tid;
    if (t < T && prob_out) {
#pragma unroll
        for (int g = 0; g < MG; g++)
            if (g < G) prob_out[(size_t)t * G + g] = my_v[g];
    }
    if (t < T) {
        int j = 0;
#pragma unroll
        for (int g = 0; g < MG; g++) {
            if ((my_mask >> g) & 1u) {
                const int r = place[g];
                token[r] = t;
                block[r] = g;
                pos[(size_t)t * k + j] = r;
                // (optional by-products: what the routed FFN otherwise derives with six more launches)
                if (token64) token64[r] = t;
                if (block64) block64[r] = g;
                if (coeff) coeff[r] = scale * my_v[g];
                j++;
            }
        }
    }
}


// The router-coefficient gradient of the routed LoRA FFN's backward (layers/sparse/grouped.py):
//   out[p] = (sum_j dot_main[p, j] + sum_j dot_act[p, j] - <du[p], u[token[p]]> - <dzt[token[p]], z[p]>)
//            / max(coeff[p], floor)
// dot_main / dot_act: the EPI_DACT epilogue's per-column-tile partial row dots [P, w]; du, z [P, r];
// u, dzt [T, r].  As torch operators: two reductions, two gathers' worth of products, seven
// elementwise kernels on 16384-element vectors -- ~40 us of launches for 2 MB of data.
// Adjoint of the `coeff` by-product of route_topk_kernel: d prob[t, g] = scale * d coeff[p] where row
// p is token t's selection of block g, zero for the blocks t did not select.  A thread per token.
// With `prob` (the router's sigmoid outputs [T, G]) the sigmoid's derivative is applied too:
// out = d logit[t, g] = d prob[t, g] * prob (1 - prob)   (torch's sigmoid backward: grad * (1 - y) * y).
__global__ __launch_bounds__(256) void route_coeff_backward_kernel(
    const float *__restrict__ dcoeff, const int32_t *__restrict__ pos, const int32_t *__restrict__ block,
    float scale, const float *__restrict__ prob, float *__restrict__ dprob, int T, int G, int k) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float out[RT_MAXG];
#pragma unroll
    for (int g = 0; g < RT_MAXG; g++) out[g] = 0.0f;
    for (int j = 0; j < k; j++) {
        const int p = pos[(size_t)t * k + j];
        const int b = block[p];
        const float v = scale * dcoeff[p];
#pragma unroll
        for (int g = 0; g < RT_MAXG; g++)
            if (g == b) out[g] = v;
    }
#pragma unroll
    for (int g = 0; g < RT_MAXG; g++)
        if (g < G) {
            float v = out[g];
            if (prob) { const float y = prob[(size_t)t * G + g]; v = (v * (1.0f - y)) * y; }
            dprob[(size_t)t * G + g] = v;
        }
}

__global__ __launch_bounds__(256) void ffn_coeff_grad_kernel(
    const float *__restrict__ dot_main, const float *__restrict__ dot_act, int w,
    const float *__restrict__ du, const float *__restrict__ u, const float *__restrict__ dzt,
    const float *__restrict__ z, const int32_t *__restrict__ token, const float *__restrict__ coeff,
    float floor_value, float *__restrict__ out, int P, int r) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float acc = 0.0f;
    for (int j = 0; j < w; j++) acc += dot_main[(size_t)p * w + j];
    float acc2 = 0.0f;
    for (int j = 0; j < w; j++) acc2 += dot_act[(size_t)p * w + j];
    const int t = token[p];
    float d1 = 0.0f, d2 = 0.0f;
    for (int j = 0; j < r; j += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(du + (size_t)p * r + j);
        const float4 b = *reinterpret_cast<const float4 *>(u + (size_t)t * r + j);
        const float4 c = *reinterpret_cast<const float4 *>(dzt + (size_t)t * r + j);
        const float4 e = *reinterpret_cast<const float4 *>(z + (size_t)p * r + j);
        d1 += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        d2 += (c.x * e.x + c.y * e.y) + (c.z * e.z + c.w * e.w);
    }
    out[p] = (((acc + acc2) - d1) - d2) / fmaxf(coeff[p], floor_value);
}

// ---- the gated FFN's elementwise middle (LLaMaFeedforward: h = silu(gate) * side,
// feedforward.py:120-131; routed + LoRA: lora_ffn.py:196-222) --------------------------------------
// forward: h = silu(g) * s.  backward, one pass over the [P, n] tensors instead of ~10 library
// launches: dg = dh * s * silu'(g), ds = dh * silu(g), and the three row dots the router-coefficient
// gradient needs -- <dh, h>, <dg, g>, <ds, s> (grouped.py: RoutedLoRALLaMAFFN.backward).
// A wave per row, float4 per lane; rows of n % 4 == 0 floats.
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    const float sg = 1.0f / (1.0f + expf(-x));
    return sg * (1.0f + x * (1.0f - sg));
}

__global__ __launch_bounds__(256) void swiglu_forward_kernel(const float *__restrict__ g,
                                                             const float *__restrict__ s,
                                                             float *__restrict__ h, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4 *>(g)[i], b = reinterpret_cast<const float4 *>(s)[i];
        reinterpret_cast<float4 *>(h)[i] =
            make_float4(silu_f(a.x) * b.x, silu_f(a.y) * b.y, silu_f(a.z) * b.z, silu_f(a.w) * b.w);
    }
}

__global__ __launch_bounds__(256) void swiglu_backward_kernel(
    const float *__restrict__ dh, const float *__restrict__ g, const float *__restrict__ s,
    float *__restrict__ dg, float *__restrict__ ds, float *__restrict__ dots, long long rows, int n) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                                     // (a wave leaves as a whole)
    const size_t base = (size_t)row * n;
    float d_h = 0.0f, d_g = 0.0f, d_s = 0.0f;
    for (int c = 4 * lane; c < n; c += 256) {
        const float4 vdh = *reinterpret_cast<const float4 *>(dh + base + c);
        const float4 vg = *reinterpret_cast<const float4 *>(g + base + c);
        const float4 vs = *reinterpret_cast<const float4 *>(s + base + c);
        const float xdh[4] = {vdh.x, vdh.y, vdh.z, vdh.w}, xg[4] = {vg.x, vg.y, vg.z, vg.w},
                    xs[4] = {vs.x, vs.y, vs.z, vs.w};
        float og[4], os[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float a = silu_f(xg[e]);
            os[e] = xdh[e] * a;
            og[e] = xdh[e] * xs[e] * silu_grad_f(xg[e]);
            d_h = fmaf(xdh[e], a * xs[e], d_h);
            d_g = fmaf(og[e], xg[e], d_g);
            d_s = fmaf(os[e], xs[e], d_s);
        }
        *reinterpret_cast<float4 *>(dg + base + c) = make_float4(og[0], og[1], og[2], og[3]);
        *reinterpret_cast<float4 *>(ds + base + c) = make_float4(os[0], os[1], os[2], os[3]);
    }
    d_h = group_sum<64>(d_h);
    d_g = group_sum<64>(d_g);
    d_s = group_sum<64>(d_s);
    if (lane == 0) {
        dots[row] = d_h;
        dots[rows + row] = d_g;
        dots[2 * rows + row] = d_s;
    }
}

}  // namespace spt

using namespace spt;

// ---- gradient of table[ids] (an embedding's rows): a segmented sum over SORTED ids ----
// lora.py:118-126: `left(x)` of a LoRA embedding is nn.Embedding; its backward adds grad[t] into row
// ids[t] of the table's gradient.  torch's own kernel sizes launches from a host-side count of the
// batch's distinct ids (not capturable); atomics make the sum order a matter of timing.  Here: the
// caller sorts the ids (stable) and passes the permutation; positions are cut into chunks at every
// run start and every multiple of 64; pass 1 sums each chunk's rows in order (fp64), pass 2 lets
// the first position of a run add its chunks' partial sums in order and write the row once.  Every
// launch's shape depends on T alone, the result on nothing but the data, and a non-finite gradient
// row stays inside its own id's row (a global running sum -- round 3's torch composition -- carried
// it into every later id).
namespace spt {
constexpr int ER_CHUNK = 64;
// thread = (position t, float4 column c4); r4 = columns / 4
__global__ __launch_bounds__(256) void embedding_rows_partial_kernel(
    const float *__restrict__ grad, long long ldg, const long long *__restrict__ sid,
    const long long *__restrict__ order, double *__restrict__ partial, long long T, int r4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = i / r4;
    const int c4 = (int)(i - t * r4);
    if (t >= T) return;
    const long long id = sid[t];
    if (!(t == 0 || (t % ER_CHUNK) == 0 || sid[t - 1] != id)) return;      // not a chunk leader
    const long long end = min(T, (t / ER_CHUNK + 1) * ER_CHUNK);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (long long u = t; u < end && sid[u] == id; u++) {
        const float4 g = *reinterpret_cast<const float4 *>(grad + order[u] * ldg + 4 * c4);
        a0 += g.x; a1 += g.y; a2 += g.z; a3 += g.w;
    }
    double *dst = partial + (t * r4 + c4) * 4;
    dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
}
__global__ __launch_bounds__(256) void embedding_rows_finish_kernel(
    const long long *__restrict__ sid, const double *__restrict__ partial, float *__restrict__ out,
    long long ldo, long long T, int r4, long long n_rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long t = i / r4;
    const int c4 = (int)(i - t * r4);
    if (t >= T) return;
    const long long id = sid[t];
    if (!(t == 0 || sid[t - 1] != id)) return;                              // not a run's first position
    if (id < 0 || id >= n_rows) return;
    const double *src = partial + (t * r4 + c4) * 4;
    double a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3];
    for (long long u = (t / ER_CHUNK + 1) * ER_CHUNK; u < T && sid[u] == id; u += ER_CHUNK) {
        const double *p = partial + (u * r4 + c4) * 4;
        a0 += p[0]; a1 += p[1]; a2 += p[2]; a3 += p[3];
    }
    *reinterpret_cast<float4 *>(out + id * ldo + 4 * c4) = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
}
}  // namespace spt

extern "C" long long spt_embedding_rows_backward_workspace_bytes(long long n_ids, int width) {
    return n_ids > 0 && width > 0 ? n_ids * width * (long long)sizeof(double) : 0;
}

extern "C" int spt_embedding_rows_backward(const float *grad, long long ldg, const long long *sorted_ids,
                                           const long long *order, float *out, long long ldo,
                                           void *workspace, long long n_ids, int width, long long n_rows,
                                           void *stream) {
    using namespace spt;
    if (!grad || !sorted_ids || !order || !out || !workspace) return SPT_EINVAL;
    if (n_ids <= 0 || n_rows <= 0 || width <= 0 || width % 4 || ldg % 4 || ldo % 4) return SPT_ESHAPE;
    if (n_rows * ldo > 0x7fffffffLL || n_ids * (width / 4) > 0x7fffffffLL * 256) return SPT_EUNSUP;
    hipStream_t s = (hipStream_t)stream;
    SPT_ZERO_WORDS(out, n_rows * ldo, s);            // rows no id names: zero gradient
    const int r4 = width / 4;
    const unsigned blocks = (unsigned)((n_ids * r4 + 255) / 256);
    hipLaunchKernelGGL(embedding_rows_partial_kernel, dim3(blocks), dim3(256), 0, s, grad, ldg, sorted_ids,
                       order, (double *)workspace, n_ids, r4);
    hipLaunchKernelGGL(embedding_rows_finish_kernel, dim3(blocks), dim3(256), 0, s, sorted_ids,
                       (const double *)workspace, out, ldo, n_ids, r4, n_rows);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_swiglu_forward(const float *gate, const float *side, float *h, long long n_elements,
                                  void *stream) {
    if (!gate || !side || !h || n_elements <= 0) return SPT_EINVAL;
    if (n_elements % 4 != 0 || ((reinterpret_cast<uintptr_t>(gate) | reinterpret_cast<uintptr_t>(side) |
                                 reinterpret_cast<uintptr_t>(h)) & 15) != 0)
        return SPT_ESHAPE;
    const long long n4 = n_elements / 4;
    const long long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(swiglu_forward_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, gate, side, h, n4);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_swiglu_backward(const float *grad_h, const float *gate, const float *side,
                                   float *grad_gate, float *grad_side, float *dots, long long rows, int n,
                                   void *stream) {
    if (!grad_h || !gate || !side || !grad_gate || !grad_side || !dots) return SPT_EINVAL;
    if (rows <= 0 || n <= 0 || rows > 0x7FFFFFFFll * 4) return SPT_EINVAL;
    if (n % 4 != 0 || ((reinterpret_cast<uintptr_t>(grad_h) | reinterpret_cast<uintptr_t>(gate) |
                        reinterpret_cast<uintptr_t>(side) | reinterpret_cast<uintptr_t>(grad_gate) |
                        reinterpret_cast<uintptr_t>(grad_side)) & 15) != 0)
        return SPT_ESHAPE;
    hipLaunchKernelGGL(swiglu_backward_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, grad_h, gate, side, grad_gate, grad_side, dots, rows, n);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

static int route_topk_any(const float *prob, int32_t *token, int32_t *block, int32_t *offsets,
                          int32_t *pos, long long *token64, long long *block64, float *coeff,
                          float scale, int n_tokens, int n_blocks, int k, void *stream, int ld = 0,
                          const float *bias = nullptr, float *prob_out = nullptr) {
    if (!prob || !token || !block || !offsets || !pos) return SPT_EINVAL;
    if (n_tokens <= 0 || n_blocks <= 0 || k <= 0 || k > n_blocks) return SPT_EINVAL;
    if (n_blocks > RT_MAXG || n_tokens > 65536) return SPT_EUNSUP;
    if (ld == 0) ld = n_blocks;
    if (ld < n_blocks || (reinterpret_cast<uintptr_t>(prob) & 15) != 0) return SPT_ESHAPE;
    const dim3 grid((n_tokens + RT_CHUNK - 1) / RT_CHUNK), threads(RT_CHUNK);
    if (n_blocks <= 4)
        hipLaunchKernelGGL(route_topk_kernel<4>, grid, threads, 0, (hipStream_t)stream, prob, token,
                           block, offsets, pos, n_tokens, n_blocks, k, token64, block64, coeff, scale,
                           ld, bias, prob_out);
    else
        hipLaunchKernelGGL(route_topk_kernel<RT_MAXG>, grid, threads, 0, (hipStream_t)stream, prob, token,
                           block, offsets, pos, n_tokens, n_blocks, k, token64, block64, coeff, scale,
                           ld, bias, prob_out);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_route_topk(const float *prob, int32_t *token, int32_t *block,
                              int32_t *offsets, int32_t *pos, int n_tokens, int n_blocks,
                              int k, void *stream) {
    return route_topk_any(prob, token, block, offsets, pos, nullptr, nullptr, nullptr, 1.0f,
                          n_tokens, n_blocks, k, stream);
}

extern "C" int spt_route_topk_coeff(const float *prob, int32_t *token, int32_t *block,
                                    int32_t *offsets, int32_t *pos, long long *token64,
                                    long long *block64, float *coeff, float scale, int n_tokens,
                                    int n_blocks, int k, void *stream) {
    if (!token64 || !block64 || !coeff) return SPT_EINVAL;
    return route_topk_any(prob, token, block, offsets, pos, token64, block64, coeff, scale,
                          n_tokens, n_blocks, k, stream);
}

extern "C" int spt_route_topk_logits(const float *logits, int ld, const float *bias, float *prob,
                                     int32_t *token, int32_t *block, int32_t *offsets, int32_t *pos,
                                     long long *token64, long long *block64, float *coeff, float scale,
                                     int n_tokens, int n_blocks, int k, void *stream) {
    if (!prob || !token64 || !block64 || !coeff) return SPT_EINVAL;
    return route_topk_any(logits, token, block, offsets, pos, token64, block64, coeff, scale, n_tokens,
                          n_blocks, k, stream, ld, bias, prob);
}

extern "C" int spt_route_coeff_backward(const float *dcoeff, const int32_t *pos, const int32_t *block,
                                        float scale, float *dprob, int n_tokens, int n_blocks,
                                        int k, void *stream) {
    if (!dcoeff || !pos || !block || !dprob) return SPT_EINVAL;
    if (n_tokens <= 0 || n_blocks <= 0 || k <= 0 || k > n_blocks) return SPT_EINVAL;
    if (n_blocks > RT_MAXG) return SPT_EUNSUP;
    hipLaunchKernelGGL(route_coeff_backward_kernel, dim3((n_tokens + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, dcoeff, pos, block, scale, nullptr, dprob, n_tokens, n_blocks, k);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_route_logit_backward(const float *dcoeff, const int32_t *pos, const int32_t *block,
                                        float scale, const float *prob, float *dlogit, int n_tokens,
                                        int n_blocks, int k, void *stream) {
    if (!dcoeff || !pos || !block || !prob || !dlogit) return SPT_EINVAL;
    if (n_tokens <= 0 || n_blocks <= 0 || k <= 0 || k > n_blocks) return SPT_EINVAL;
    if (n_blocks > RT_MAXG) return SPT_EUNSUP;
    hipLaunchKernelGGL(route_coeff_backward_kernel, dim3((n_tokens + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, dcoeff, pos, block, scale, prob, dlogit, n_tokens, n_blocks, k);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}

extern "C" int spt_ffn_coeff_grad(const float *dot_main, const float *dot_act, int width,
                                  const float *du, const float *u, const float *dzt, const float *z,
                                  const int32_t *token, const float *coeff, float floor_value,
                                  float *out, int n_rows, int rank, void *stream) {
    if (!dot_main || !dot_act || !du || !u || !dzt || !z || !token || !coeff || !out) return SPT_EINVAL;
    if (n_rows <= 0 || width <= 0 || rank <= 0) return SPT_EINVAL;
    if (rank % 4 != 0) return SPT_ESHAPE;
    hipLaunchKernelGGL(ffn_coeff_grad_kernel, dim3((n_rows + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, dot_main, dot_act, width, du, u, dzt, z, token, coeff,
                       floor_value, out, n_rows, rank);
    SPT_LAUNCH_CHECK();
    return SPT_OK;
}
