// ncx_knn.hip -- brute-force k nearest neighbours of image-feature rows.
//
// Replaces the reference's knn.py:41-58 (sklearn NearestNeighbors(n_neighbors=25).fit(features).kneighbors(batch),
// brute force, euclidean): the step that builds the 24-candidate lists the CX data set is made of.
//
//   1. halfneg[j] = -|x_j|^2 / 2                                         (k_knn_halfnorm, once per table)
//   2. V[i][j]    = q_i . x_j - |x_j|^2 / 2   for a block of queries      (the NT GEMM engine, bias = halfneg)
//                   |q_i - x_j|^2 = |q_i|^2 - 2 V[i][j]: per query, the k nearest rows are the k LARGEST V
//   3. per query row: linear-histogram select of the k+8 largest V (3 coalesced passes over the row), then the
//      candidates' distances are recomputed exactly as sum (q - x)^2 (no cancellation), sorted by (distance, index)
//      and the first k are written.                                       (k_knn_select, one workgroup per query)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ncx_internal.h"

namespace ncx {

constexpr int KNN_BINS = 1024;
constexpr int KNN_CAP = 1024;        // candidate buffer (elements at or above the threshold bin)
constexpr int KNN_MAXK = 128;        // k + margin
constexpr int KNN_LEVELS = 3;

__global__ __launch_bounds__(256) void k_knn_halfnorm(const float* __restrict__ x, int n, int dv, float* __restrict__ halfneg) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* p = x + (long long)row * dv;
    float s = 0.f;
    for (int c = lane; c < dv; c += 64) { const float v = p[c]; s = fmaf(v, v, s); }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) halfneg[row] = -0.5f * s;
}

struct KnnLevel { float lo, scale; int b; };

// membership of v in the candidate set defined by the refinement levels: bin above the chosen one at the first level
// where it differs, or inside the chosen bin at every level.  Same float expressions as the histogram pass.
__device__ inline int knn_bin(float v, float lo, float scale) {
    const float f = (v - lo) * scale;
    int b = (int)f;
    return b < 0 ? 0 : b > KNN_BINS - 1 ? KNN_BINS - 1 : b;
}
// returns +1 above, 0 inside the chosen bins of all `nl` levels, -1 below
__device__ inline int knn_classify(float v, const KnnLevel* lv, int nl) {
    for (int l = 0; l < nl; ++l) {
        const int b = knn_bin(v, lv[l].lo, lv[l].scale);
        if (b > lv[l].b) return 1;
        if (b < lv[l].b) return -1;
    }
    return 0;
}

__global__ __launch_bounds__(256) void k_knn_select(const float* __restrict__ V, long long ldv, const float* __restrict__ table,
                                                    const float* __restrict__ queries, int n, int dv, int k, int kc,
                                                    long long* __restrict__ out_idx, float* __restrict__ out_dist) {
    __shared__ int hist[KNN_BINS];
    __shared__ float cand_v[KNN_CAP];
    __shared__ int cand_j[KNN_CAP];
    __shared__ float red_a[4], red_b[4];
    __shared__ KnnLevel lv[KNN_LEVELS];
    __shared__ int s_nl, s_count, s_count_in, s_done, s_above_total, s_chosen;
    __shared__ float top_v[KNN_MAXK];
    __shared__ int top_j[KNN_MAXK];
    __shared__ double top_d[KNN_MAXK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long q = blockIdx.x;
    const float* row = V + q * ldv;

    // pass 1: range of the row
    float mx = -INFINITY, mn = INFINITY;
    for (int j = tid; j < n; j += 256) { const float v = row[j]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    if (lane == 0) { red_a[wave] = mx; red_b[wave] = mn; }
    if (tid == 0) { s_nl = 0; s_done = 0; s_above_total = 0; }
    __syncthreads();
    mx = fmaxf(fmaxf(red_a[0], red_a[1]), fmaxf(red_a[2], red_a[3]));
    mn = fminf(fminf(red_b[0], red_b[1]), fminf(red_b[2], red_b[3]));

    // refinement levels: histogram of the elements still undecided, pick the bin holding the kc-th largest
    float lo = mn, hi = mx;
    for (int level = 0; level < KNN_LEVELS; ++level) {
        for (int b = tid; b < KNN_BINS; b += 256) hist[b] = 0;
        const float range = hi - lo;
        const float scale = range > 0.f ? (float)KNN_BINS / range : 0.f;
        __syncthreads();
        const int nl = s_nl;
        for (int j = tid; j < n; j += 256) {
            const float v = row[j];
            if (knn_classify(v, lv, nl) == 0) atomicAdd(&hist[knn_bin(v, lo, scale)], 1);
        }
        __syncthreads();
        if (wave == 0) {
            // suffix counts from the top bin: lane L owns bins [16 L, 16 L + 16)
            int local = 0;
            for (int b = 0; b < 16; ++b) local += hist[lane * 16 + b];
            int above = 0;                                      // elements in bins owned by higher lanes
            for (int o = 1; o < 64; ++o) { const int other = __shfl(local, (lane + o) & 63); if (lane + o < 64) above += other; }
            const int need = kc - s_above_total;                // still to be found inside this level's range
            // the owning lane: above < need <= above + local
            if (above < need && need <= above + local) {
                int acc = above, chosen = lane * 16;
                for (int b = 15; b >= 0; --b) {
                    const int h = hist[lane * 16 + b];
                    if (acc + h >= need) { chosen = lane * 16 + b; break; }
                    acc += h;
                }
                s_chosen = chosen;
                lv[nl].lo = lo; lv[nl].scale = scale; lv[nl].b = chosen;
                s_nl = nl + 1;
                // stop when the chosen bin's elements fit the buffer's tie region (elements above it number < kc)
                if (hist[chosen] <= KNN_CAP - kc || scale == 0.f || level == KNN_LEVELS - 1) s_done = 1;
                else s_above_total += acc;
            }
        }
        __syncthreads();
        if (s_done) break;
        // refine inside the chosen bin: its value range, slightly widened (clamped bins absorb the rounding)
        const float w = range / (float)KNN_BINS;
        const float nlo = lo + w * (float)s_chosen - 1e-6f * fabsf(lo + w * (float)s_chosen);
        const float nhi = lo + w * (float)(s_chosen + 1) + 1e-6f * fabsf(lo + w * (float)(s_chosen + 1));
        lo = nlo; hi = nhi;
        __syncthreads();
    }

    // pass 3: collect everything above the chosen bins (from the front of the buffer) and inside them (from the back:
    // when more elements tie inside the last bin than the buffer holds, the overflow never displaces the ones above)
    if (tid == 0) { s_count = 0; s_count_in = 0; }
    __syncthreads();
    {
        const int nl = s_nl;
        for (int j = tid; j < n; j += 256) {
            const float v = row[j];
            const int cls = knn_classify(v, lv, nl);
            if (cls > 0) {
                const int slot = atomicAdd(&s_count, 1);
                if (slot < KNN_CAP) { cand_v[slot] = v; cand_j[slot] = j; }
            } else if (cls == 0) {
                const int slot = KNN_CAP - 1 - atomicAdd(&s_count_in, 1);
                if (slot >= kc) { cand_v[slot] = v; cand_j[slot] = j; }      // above-elements number < kc <= slot
            }
        }
    }
    __syncthreads();
    const int na = s_count;                                                   // < kc by construction
    const int nb = s_count_in < KNN_CAP - kc ? s_count_in : KNN_CAP - kc;
    const int nc = na + nb;
    const int kk = kc < nc ? kc : nc;
    auto phys = [&](int e) { return e < na ? e : KNN_CAP - 1 - (e - na); };
    // rank sort by (V descending, index ascending); the first kc go on
    for (int e = tid; e < nc; e += 256) {
        const float v = cand_v[phys(e)]; const int j = cand_j[phys(e)];
        int rank = 0;
        for (int f = 0; f < nc; ++f) {
            const float vf = cand_v[phys(f)];
            rank += (vf > v) || (vf == v && cand_j[phys(f)] < j);
        }
        if (rank < kk) { top_v[rank] = v; top_j[rank] = j; }
    }
    __syncthreads();
    // exact distances of the candidates in fp64 (fp32 inputs: the differences are exact, the ordering is the true one;
    // an fp32 sum swaps neighbours closer than ~1e-6 relative): one wave per candidate
    const float* qrow = queries + q * dv;
    for (int c = wave; c < kk; c += 4) {
        const float* xr = table + (long long)top_j[c] * dv;
        double s = 0.0;
        for (int t = lane; t < dv; t += 64) { const double df = (double)qrow[t] - (double)xr[t]; s = fma(df, df, s); }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) top_d[c] = s;
    }
    __syncthreads();
    if (tid < kk) {
        const double dd = top_d[tid]; const int j = top_j[tid];
        int rank = 0;
        for (int f = 0; f < kk; ++f) rank += (top_d[f] < dd) || (top_d[f] == dd && top_j[f] < j);
        if (rank < k) { out_idx[q * k + rank] = j; out_dist[q * k + rank] = (float)sqrt(dd); }
    }
    // fewer than k rows in the table: pad (callers validate k <= n, so this is unreachable in practice)
    for (int r = kk + tid; r < k; r += 256) { out_idx[q * k + r] = -1; out_dist[q * k + r] = INFINITY; }
}

}  // namespace ncx

using namespace ncx;

extern "C" size_t ncx_knn_workspace_bytes(int32_t n, int32_t block_rows) {
    if (n < 1 || block_rows < 1) return 0;
    return align_up((size_t)n * 4, 256) + (size_t)block_rows * (size_t)n * 4 + 256;
}

extern "C" int ncx_knn(const float* table, int32_t n, const float* queries, int32_t nq, int32_t dv, int32_t k,
                       int32_t norms_ready, void* workspace, size_t workspace_bytes, int64_t* out_idx, float* out_dist,
                       void* stream_) {
    if (!table || !queries || !workspace || !out_idx || !out_dist) return NCX_E_NULL;
    if (n < 1 || nq < 1 || dv < 4 || k < 1 || k > n || k + 8 > KNN_MAXK) return NCX_E_DIMS;
    if (workspace_bytes < ncx_knn_workspace_bytes(n, nq) || ((uintptr_t)workspace & 255)) return NCX_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    float* halfneg = (float*)workspace;
    float* V = (float*)((char*)workspace + align_up((size_t)n * 4, 256));
    if (!norms_ready) {
        hipLaunchKernelGGL(k_knn_halfnorm, dim3((n + 3) / 4), dim3(256), 0, s, table, n, dv, halfneg);
        NCX_HIP_TRY(hipGetLastError());
    }
    {
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = nq;
        a.a[0] = x_plain(queries, dv, nq, dv); a.b[0] = x_plain(table, dv, n, dv); a.klen[0] = dv;
        a.out[0] = V; a.ldo[0] = n; a.n_cols[0] = n; a.split[0] = 1;
        a.epi.bias = halfneg;
        GemmPlan pl = plan_gemm(FORM_NT, nq, n, (dv + GEMM_BK - 1) / GEMM_BK, true);
        pl.split = 1;
        const int rc = run_gemm_nt(a, pl.cfg, s);
        if (rc) return rc;
    }
    const int kc = k + 8 < n ? k + 8 : n;
    hipLaunchKernelGGL(k_knn_select, dim3(nq), dim3(256), 0, s, (const float*)V, (long long)n, table, queries, n, dv, k, kc,
                       (long long*)out_idx, out_dist);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}
