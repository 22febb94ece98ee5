// ncx_api.hip -- C ABI of libneuralcx_hip.so (include/neuralcx.h) and the bandwidth-bound kernels around
// the segmented MFMA GEMM engine (ncx_gemm.h).  gfx950 only.
//
// Forward of one batch (B triplets x K candidates, M = B*K rows), replacing vqa/models/cx.py:279-331:
//   k_prep           per row (b,k): feature-table row ids, ||v_o - v_k + 1e-6||_2 (cx.py:300), rank one-hot
//                    (cx.py:304-305) and the softmax statistics of a_knns[b,k,:] (cx.py:281)          [HBM]
//   Gt   = W1[:, a_emb_other] . E^T      [H, A]   re-association of K3: softmax(a).E.W^T = softmax(a).(E.W^T) [MFMA]
//   Sh   = b1 + [v_o | q | z_o | E[aid]] . W1[:, shared cols]^T      [B, H]  once per triplet        [MFMA]
//   h1   = drop(relu(Sh[b] + [v_k | v_o*v_k | dist,rank | z_k | softmax(a_k)] . [W1 slices | Gt]^T)) [MFMA]
//   h2, h3 (L >= 2), scores = h_L . w_out + b_out                                                     [MFMA/HBM]
// Backward mirrors it (see ncx_backward).  Nothing here allocates or synchronises.
#include "ncx_internal.h"
#include "ncx_dwred.h"
#include "ncx_bf16.h"
#include <stdlib.h>
#include <stdio.h>
#include <atomic>

namespace ncx {

// =================================================================================================
// small kernels
// =================================================================================================
// Wave-wide reductions on the DPP network (round 3).  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt: six dependent LDS round
// trips per reduction -- k_train_tail's 24 row dots were 187 of them, ~9 of the kernel's 16 us.  Here: four DPP steps inside each row
// of 16 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every lane then holds its row's result), then the four
// row results through v_readlane.  Fixed association ((quad pairs) half rows) rows: deterministic, the same in every kernel that
// calls it (the fused and unfused paths stay bit-identical to each other).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

// One wave per logical row r = b*K + k.  4 rows per 256-thread block.
__device__ __forceinline__ u16 f32_to_bf16(float x) { return __builtin_bit_cast(u16, (__bf16)x); }     // round to nearest even
// 4 consecutive bf16 at an 8-byte aligned position (one store); `valid` < 4 keeps the tail of a segment untouched
__device__ __forceinline__ void store_bf16x4(u16* p, const float (&v)[4], int valid) {
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    if (valid >= 4) { *(u16x4*)p = u16x4{f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])}; }
    else { for (int j = 0; j < valid; ++j) p[j] = f32_to_bf16(v[j]); }
}

// Padded weight copies (ncx_main.h reads every weight row up to the next multiple of 32 columns).  Slot i of the wpad region:
//   0 v_other  1 v_mult  2 dist | rank  3 z_other  4 a_other (a_emb lesion only)  5 linear_2  6 linear_3
// width 0: the slice is used in place (already a multiple of 32 wide, or the segment does not exist).
constexpr int WPAD_N = 7;
static inline int wpad_cols(const ncx_dims& d, int i) {
    switch (i) {
    case 0: return d.dv;
    case 1: return (d.flags & NCX_F_V_MULT) ? d.dv : 0;
    case 2: return d.K + 1;
    case 3: return d.dz;
    case 4: return (d.flags & NCX_F_A_EMB) ? 0 : d.da;
    case 5: return d.L >= 2 ? d.H : 0;
    default: return d.L >= 3 ? d.H : 0;
    }
}
static inline int wpad_width(const ncx_dims& d, int i) {
    const int c = wpad_cols(d, i);
    if (d.flags & NCX_F_BF16) return 0;
    return (c == 0 || c % 32 == 0) ? 0 : pad_to(c, 32);
}
struct PackArgs { const float* src[WPAD_N + 1]; float* dst[WPAD_N + 1]; long long lds[WPAD_N + 1]; int cols[WPAD_N + 1], ldd[WPAD_N + 1], zero_from[WPAD_N + 1]; int n, H; int nprep; };   // nprep: row blocks (4 candidate rows each) in front of the pack blocks
// dst[e][h][0 .. cols) = src[e][h][0 .. cols);  dst[e][h][zero_from .. ldd) = 0.   One block per (h, e): H * n blocks, which
// ride at the end of k_prep's grid (a launch of their own cost 5 us for 0.5 MB of copies).
__device__ __forceinline__ void pack_rows_block(const PackArgs& a, int idx) {
    const int h = idx % a.H, e = idx / a.H;
    float* drow = a.dst[e] + (long long)h * a.ldd[e];
    if (a.src[e]) {
        const float* srow = a.src[e] + (long long)h * a.lds[e];
        for (int c = threadIdx.x; c < a.cols[e]; c += 256) drow[c] = srow[c];
    }
    for (int c = a.zero_from[e] + threadIdx.x; c < a.ldd[e]; c += 256) drow[c] = 0.f;
}

// One wave per candidate row.  Rows of up to 2048 floats (the real widths: 2048-d features, 2000 answers) are read from
// memory ONCE into registers (8 x float4 per lane) and every pass -- distance, max, sum, bf16 pack -- runs on the
// registers; wider rows (RESIDENT = false) re-read them from memory per pass.  Same per-lane element order and the same
// wave reductions either way, so the results do not depend on the path.
template <bool RESIDENT>
__global__ __launch_bounds__(256) void k_prep(ncx_dims d, ncx_inputs in, int* __restrict__ idx_k,
                                              int* __restrict__ idx_o, int* __restrict__ idx_ob,
                                              float* __restrict__ mx, float* __restrict__ inv,
                                              float* __restrict__ misc, u16* __restrict__ xc, Bf16Cols cc, const PackArgs pk) {
    constexpr int NR = RESIDENT ? 8 : 1;
    const int lane = threadIdx.x & 63;
    const int M = d.B * d.K;
    const int nprep = pk.nprep;                // (M + 3) / 4, or 0 in a pack-only launch (ncx_forward_phase: the weights-only half)
    if ((int)blockIdx.x >= nprep) { pack_rows_block(pk, (int)blockIdx.x - nprep); return; }     // the weight-pack blocks
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int b = r / d.K, k = r - b * d.K;
    // RESIDENT: every row load is an UNCONDITIONAL 16-byte window (slid left at the row end, repaired below); the guarded form
    // put each load in its own basic block behind an s_waitcnt vmcnt(0): 24 serialised round trips per row, 2 TB/s.
    // The answer-logit row needs no index, so its loads go first; the two feature rows follow their indices.
    const bool aemb_ = d.flags & NCX_F_A_EMB;
    f32x4 ra[NR];
    if (RESIDENT && aemb_) {
        const float* arow = in.a_knns + (long long)r * d.A;
#pragma unroll
        for (int i = 0; i < NR; ++i) ra[i] = load_window(arow, lane * 4 + 256 * i, d.A);
    }
    const int io = in.img_idx[(long long)b * (d.K + 1)];
    const int ik = in.img_idx[(long long)b * (d.K + 1) + 1 + k];
    if (lane == 0) { idx_o[r] = io; idx_k[r] = ik; if (k == 0) idx_ob[b] = io; }
    const float* vo = in.feats + (long long)io * d.dv;
    const float* vk = in.feats + (long long)ik * d.dv;
    u16* xr = xc ? xc + (long long)r * cc.kc : nullptr;
    const bool own_dist = (d.flags & NCX_F_V_DIST) && !(d.flags & NCX_F_PRIV_DIST_IN_MAIN);
    const bool need_v = own_dist || xc;

    f32x4 ro[NR], rk[NR];
    if (RESIDENT && need_v) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { ro[i] = load_window(vo, lane * 4 + 256 * i, d.dv); rk[i] = load_window(vk, lane * 4 + 256 * i, d.dv); }
        if (d.dv % 256 != 0) {                     // (uniform: whole 256-column passes need no repair)
#pragma unroll
            for (int i = 0; i < NR; ++i) { ro[i] = fix_window(ro[i], lane * 4 + 256 * i, d.dv); rk[i] = fix_window(rk[i], lane * 4 + 256 * i, d.dv); }
        }
    }
    auto v_at = [&](int i, int c, f32x4& a, f32x4& e) __attribute__((always_inline)) {
        if (RESIDENT) { a = ro[i]; e = rk[i]; } else { a = load4(vo, c, d.dv); e = load4(vk, c, d.dv); }
    };
    float dist = 0.f;
    if (own_dist) {
        float s = 0.f;
        if (RESIDENT) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int c = lane * 4 + 256 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < d.dv) { const float t = ro[i][j] - rk[i][j] + 1e-6f; s += t * t; }
            }
        } else {
            for (int c = lane * 4; c < d.dv; c += 256) {
                f32x4 a, e; v_at(0, c, a, e);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < d.dv) { const float t = a[j] - e[j] + 1e-6f; s += t * t; }
            }
        }
        dist = sqrtf(wave_sum(s));
    }
    const int mw = pad_to(d.K + 1, 4);                        // (zero padded to whole 16-byte windows: ncx_main.h)
    float* mrow = misc + (long long)r * mw;
    if (lane == 0) mrow[0] = dist;
    for (int j = lane; j < mw - 1; j += 64)
        mrow[1 + j] = j >= d.K ? 0.f : (d.flags & NCX_F_V_RANK) ? (j == k ? 1.f : 0.f) : in.v_rank[((long long)r) * d.K + j];
    if (xc) {               // NCX_F_BF16: [ v_k | v_o * v_k | dist, rank | z_k | softmax (below) ], zero in the gaps
        if (RESIDENT) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < cc.c_vm) {                                     // (segments are padded to multiples of 8 columns)
                    float pk[4], pm[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pk[j] = rk[i][j]; pm[j] = ro[i][j] * rk[i][j]; }   // load4: zero beyond dv
                    store_bf16x4(xr + cc.c_vk + c, pk, 4);
                    store_bf16x4(xr + cc.c_vm + c, pm, 4);
                }
            }
        } else {
            for (int c = lane * 4; c < cc.c_vm; c += 256) {
                f32x4 a, e; v_at(0, c, a, e);
                float pk[4], pm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { pk[j] = e[j]; pm[j] = a[j] * e[j]; }
                store_bf16x4(xr + cc.c_vk + c, pk, 4);
                store_bf16x4(xr + cc.c_vm + c, pm, 4);
            }
        }
        for (int j = lane; j < cc.c_z - cc.c_misc; j += 64)
            xr[cc.c_misc + j] = f32_to_bf16(j == 0 ? dist : (j <= d.K && j - 1 == k ? 1.f : 0.f));
        const float* zk = in.z_knns + (long long)r * d.dz;
        for (int c = lane * 4; c < cc.c_p - cc.c_z; c += 256) {
            float pz[4];
            const f32x4 z4 = load4(zk, c, d.dz);
#pragma unroll
            for (int j = 0; j < 4; ++j) pz[j] = z4[j];
            store_bf16x4(xr + cc.c_z + c, pz, 4);
        }
        for (int c = cc.raw + lane; c < cc.kc; c += 64) xr[c] = 0;
    }

    if (d.flags & NCX_F_A_EMB) {
        const float* a = in.a_knns + (long long)r * d.A;
        if (RESIDENT && d.A % 256 != 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) ra[i] = fix_window(ra[i], lane * 4 + 256 * i, d.A);
        }
        float m = -INFINITY;
        if (RESIDENT) {
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) if (lane * 4 + 256 * i + j < d.A) m = fmaxf(m, ra[i][j]);
        } else {
            for (int c = lane * 4; c < d.A; c += 256) {
                const f32x4 v = load4(a, c, d.A);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < d.A) m = fmaxf(m, v[j]);
            }
        }
        m = wave_max(m);
        float s = 0.f;
        if (RESIDENT) {
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) if (lane * 4 + 256 * i + j < d.A) s += __expf(ra[i][j] - m);
        } else {
            for (int c = lane * 4; c < d.A; c += 256) {
                const f32x4 v = load4(a, c, d.A);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < d.A) s += __expf(v[j] - m);
            }
        }
        s = wave_sum(s);
        // base-2 log-sum-exp: softmax(a)[c] = exp2(a[c]*log2e - lse2)
        const float lse2 = m * 1.44269504088896341f + __log2f(s);
        if (lane == 0) { mx[r] = lse2; inv[r] = 0.f; }
        if (xc) {           // NCX_F_BF16: the softmax row itself, rounded to bf16, is the last segment of the packed row
            u16* xp = xr + cc.c_p;
            if (RESIDENT) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int c = lane * 4 + 256 * i;
                    if (c < d.A) {
                        float e[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(ra[i][j], 1.44269504088896341f, -lse2));
                        store_bf16x4(xp + c, e, d.A - c);
                    }
                }
            } else {
                for (int c = lane * 4; c < d.A; c += 256) {
                    const f32x4 v = load4(a, c, d.A);
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -lse2));
                    store_bf16x4(xp + c, e, d.A - c);
                }
            }
        }
    }
}

// (explicit fma order: k_scores and k_train_tail must round identically)
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& e) {
    return __builtin_fmaf(a[3], e[3], __builtin_fmaf(a[2], e[2], __builtin_fmaf(a[1], e[1], a[0] * e[0])));
}
// scores[r] = h[r,:] . w + b     (cx.py:327); one wave per row.
__global__ __launch_bounds__(256) void k_scores(const float* __restrict__ h, const float* __restrict__ w,
                                                const float* __restrict__ bias, float* __restrict__ scores,
                                                int M, int H) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const float* row = h + (long long)r * H;
    float s = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 a = load4(row, c, H), e = load4(w, c, H);
        s += dot4(a, e);
    }
    s = wave_sum(s);
    if (lane == 0) scores[r] = s + bias[0];
}

// Listwise softmax cross-entropy over the K candidates of a triplet + rank of the ground truth.
// One wave per triplet, lane k holds score k (K <= 64).
__global__ __launch_bounds__(256) void k_loss_rank(const float* __restrict__ scores, const int* __restrict__ gt,
                                                   int B, int K, float scale, float* __restrict__ loss_rows,
                                                   float* __restrict__ dscores, int* __restrict__ rank) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float s = lane < K ? scores[(long long)b * K + lane] : -INFINITY;
    const int g = gt[b];
    const float m = wave_max(s);
    const float e = lane < K ? __expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    const float sg = __shfl(s, g, 64);
    if (dscores && lane < K) dscores[(long long)b * K + lane] = (e / sum - (lane == g ? 1.f : 0.f)) * scale;
    const bool ahead = lane < K && (s > sg || (s == sg && lane < g));
    const unsigned long long bal = __ballot(ahead);
    if (lane == 0) {
        if (loss_rows) loss_rows[b] = (logf(sum) + m - sg) * scale;
        if (rank) rank[b] = __popcll(bal);
    }
}

// loss = sum(loss_rows); hits = {#rank<1, #rank<5}.  Single block: deterministic.
__global__ __launch_bounds__(256) void k_loss_finish(const float* __restrict__ loss_rows, const int* __restrict__ rank,
                                                     int B, float* __restrict__ loss, int* __restrict__ hits) {
    __shared__ float sl[4];
    __shared__ int s1[4], s5[4];
    float acc = 0.f; int h1 = 0, h5 = 0;
    for (int i = threadIdx.x; i < B; i += 256) {
        if (loss_rows) acc += loss_rows[i];
        if (rank) { const int rk = rank[i]; h1 += rk < 1; h5 += rk < 5; }
    }
    acc = wave_sum(acc);
    h1 = (int)wave_sum((float)h1); h5 = (int)wave_sum((float)h5);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sl[w] = acc; s1[w] = h1; s5[w] = h5; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (loss) loss[0] = sl[0] + sl[1] + sl[2] + sl[3];
        if (hits) { hits[0] = s1[0] + s1[1] + s1[2] + s1[3]; hits[1] = s5[0] + s5[1] + s5[2] + s5[3]; }
    }
}

// dpre[r,n] = gs[r] * w_out[n] * (h[r,n] > 0 ? scale : 0)          (backward of out + dropout + relu)
__global__ __launch_bounds__(256) void k_dpre_last(const float* __restrict__ gs, const float* __restrict__ w_out,
                                                   const float* __restrict__ h, float* __restrict__ dpre,
                                                   long long total, int H, float scale) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const int r = (int)(i / H), n = (int)(i - (long long)r * H);
    if (n + 3 < H && i + 3 < total) {
        const f32x4 hv = *(const f32x4u*)(h + i);
        const f32x4 wv = *(const f32x4u*)(w_out + n);
        const float g = gs[r];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = hv[j] > 0.f ? g * wv[j] * scale : 0.f;
        *(f32x4u*)(dpre + i) = o;
    } else {
        for (int j = 0; j < 4 && i + j < total; ++j) {
            const int rr = (int)((i + j) / H), nn = (int)((i + j) - (long long)rr * H);
            dpre[i + j] = h[i + j] > 0.f ? gs[rr] * w_out[nn] * scale : 0.f;
        }
    }
}

// partial[ch][n] = sum over rows of chunk ch of x[r][n] (* wgt[r]);  finish sums the chunks.
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, const float* __restrict__ wgt,
                                                        int M, int N, float* __restrict__ partial) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, nch = gridDim.y;
    if (n >= N) return;
    const int r0 = (int)((long long)M * ch / nch), r1 = (int)((long long)M * (ch + 1) / nch);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    if (wgt) {
        for (; r + 3 < r1; r += 4) {
            s0 += x[(long long)r * N + n] * wgt[r];           s1 += x[(long long)(r + 1) * N + n] * wgt[r + 1];
            s2 += x[(long long)(r + 2) * N + n] * wgt[r + 2]; s3 += x[(long long)(r + 3) * N + n] * wgt[r + 3];
        }
        for (; r < r1; ++r) s0 += x[(long long)r * N + n] * wgt[r];
    } else {
        for (; r + 3 < r1; r += 4) {
            s0 += x[(long long)r * N + n];       s1 += x[(long long)(r + 1) * N + n];
            s2 += x[(long long)(r + 2) * N + n]; s3 += x[(long long)(r + 3) * N + n];
        }
        for (; r < r1; ++r) s0 += x[(long long)r * N + n];
    }
    partial[(long long)ch * N + n] = (s0 + s1) + (s2 + s3);
}
// out[n] = sum_ch partial[ch][n]: 32 threads per column (8 columns per block), fixed-order tree -> deterministic
__global__ __launch_bounds__(256) void k_colsum_finish(const float* __restrict__ partial, int nch, int N,
                                                       float* __restrict__ out) {
    __shared__ float red[32][9];
    const int c = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int n = blockIdx.x * 8 + c;
    float s = 0.f;
    if (n < N) for (int ch = g; ch < nch; ch += 32) s += partial[(long long)ch * N + n];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[i][c];
        out[n] = t;
    }
}
// out[0] = sum(x[0..n)); single block.
__global__ __launch_bounds__(256) void k_sum_vec(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float sl[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sl[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sl[0] + sl[1] + sl[2] + sl[3];
}

// Backward prelude in one pass over h_L.  One WAVE per run of triplets, lane = 4 consecutive columns (+ 256 per column pass):
//   dpre[r][n] = gs[r] * w_out[n] * (h[r][n] > 0 ? scale : 0)                    (out + dropout + relu backward)
//   partial_w[wave][n] = sum_r gs[r] h[r][n]    partial_b[wave] = sum_r gs[r]     (-> d out.weight, d out.bias)
//   L == 1 only: dsh[b][n] = sum_k dpre[b*K+k][n],  partial_b1[wave][n] = sum_b dsh[b][n]   (-> d linear_1.bias)
// The K rows of a triplet are fetched 8 at a time with unconditional 16-byte loads (the column-per-thread form walked them one
// dependent load at a time on 4 waves per CU: 18.7 us for 28 MB at configs[1]); no LDS, no barrier: a lane owns its columns.
__global__ __launch_bounds__(256) void k_bwd_prelude(const float* __restrict__ gs, const float* __restrict__ w_out,
                                                     const float* __restrict__ h, float* __restrict__ dpre,
                                                     float* __restrict__ dsh, int B, int K, int H, float scale,
                                                     float* __restrict__ partial_w, float* __restrict__ partial_b1,
                                                     float* __restrict__ partial_b, float* __restrict__ zero_buf, long long zero_n) {
    const int blk = blockIdx.x, nblk = gridDim.x;
    if (zero_buf) {          // dGgt = one-hot(aid)^T dSh is scattered into zeros later in the backward: cleared here (saves a memset launch)
        const long long z0 = zero_n * blk / nblk / 4 * 4, z1 = blk + 1 == nblk ? zero_n : zero_n * (blk + 1) / nblk / 4 * 4;
        for (long long i = z0 + 4 * threadIdx.x; i < z1; i += 1024) {
            if (i + 3 < z1) *(f32x4u*)(zero_buf + i) = f32x4{0.f, 0.f, 0.f, 0.f};
            else for (long long j = i; j < z1; ++j) zero_buf[j] = 0.f;
        }
    }
    const int lane = threadIdx.x & 63;
    const int wv = blk * 4 + (threadIdx.x >> 6), nwv = nblk * 4;
    const int b0 = (int)((long long)B * wv / nwv), b1 = (int)((long long)B * (wv + 1) / nwv);
    auto put4 = [&](float* p, int c, const f32x4& v) __attribute__((always_inline)) {
        if (c + 3 < H) *(f32x4u*)(p + c) = v;
        else { p[c] = v[0]; if (c + 1 < H) p[c + 1] = v[1]; if (c + 2 < H) p[c + 2] = v[2]; }
    };
    for (int c = lane * 4; c < H; c += 256) {
        const bool edge = c + 4 > H;
        const f32x4 w = fix_window(load_window(w_out, c, H), c, H);
        f32x4 aw = {0.f, 0.f, 0.f, 0.f}, ab1 = {0.f, 0.f, 0.f, 0.f};
        for (int b = b0; b < b1; ++b) {
            const long long r0 = (long long)b * K;
            f32x4 ds = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < K; k0 += 8) {
                f32x4 hv[8]; float g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const long long r = r0 + min(k0 + j, K - 1);
                    hv[j] = load_window(h + r * H, c, H);
                    g[j] = k0 + j < K ? gs[r] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j < K) {
                        const f32x4 v = edge ? fix_window(hv[j], c, H) : hv[j];
                        f32x4 dp;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { dp[q] = v[q] > 0.f ? g[j] * w[q] * scale : 0.f; aw[q] = __builtin_fmaf(g[j], v[q], aw[q]); ds[q] += dp[q]; }
                        put4(dpre + (r0 + k0 + j) * H, c, dp);
                    }
                }
            }
            if (dsh) { put4(dsh + (long long)b * H, c, ds); ab1 += ds; }
        }
        put4(partial_w + (long long)wv * H, c, aw);
        if (dsh) put4(partial_b1 + (long long)wv * H, c, ab1);
    }
    float s = 0.f;
    for (long long r = (long long)b0 * K + lane; r < (long long)b1 * K; r += 64) s += gs[r];
    s = wave_sum(s);
    if (lane == 0) partial_b[wv] = s;
}
// out_w[n] = sum_blk partial_w[blk][n]; out_b1[n] likewise (nullable); out_b[0] = sum_blk partial_b[blk]
__global__ __launch_bounds__(256) void k_bwd_prelude_finish(const float* __restrict__ partial_w, const float* __restrict__ partial_b1,
                                                            const float* __restrict__ partial_b, int nblk, int H,
                                                            float* __restrict__ out_w, float* __restrict__ out_b1,
                                                            float* __restrict__ out_b, const float* __restrict__ loss_rows = nullptr,
                                                            const int* __restrict__ rank = nullptr, int B = 0,
                                                            float* __restrict__ loss = nullptr, int* __restrict__ hits = nullptr) {
    if (blockIdx.y == 2) {                                  // ncx_train_tail: k_loss_finish's sums ride in this launch (same order)
        if (blockIdx.x != 0) return;
        __shared__ float sl[4];
        __shared__ int s1[4], s5[4];
        float acc = 0.f; int h1 = 0, h5 = 0;
        for (int i = threadIdx.x; i < B; i += 256) {
            if (loss_rows) acc += loss_rows[i];
            if (rank) { const int rk = rank[i]; h1 += rk < 1; h5 += rk < 5; }
        }
        acc = wave_sum(acc);
        h1 = (int)wave_sum((float)h1); h5 = (int)wave_sum((float)h5);
        const int wq = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sl[wq] = acc; s1[wq] = h1; s5[wq] = h5; }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (loss) loss[0] = sl[0] + sl[1] + sl[2] + sl[3];
            if (hits) { hits[0] = s1[0] + s1[1] + s1[2] + s1[3]; hits[1] = s5[0] + s5[1] + s5[2] + s5[3]; }
        }
        return;
    }
    __shared__ float red[32][9];
    const int c = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int which = blockIdx.y;                                   // 0: out.weight, 1: linear_1.bias
    const float* part = which == 0 ? partial_w : partial_b1;
    float* out = which == 0 ? out_w : out_b1;
    if (!out) return;
    const int n = blockIdx.x * 8 + c;
    // the partial rows of this thread (ch = g, g + 32, ...) are requested 16 at a time before the first is added: a load per
    // iteration of the runtime-length loop was one dependent round trip per row (16 of them = the kernel's 7 us); same order of adds
    float s = 0.f;
    if (n < H) {
        for (int ch0 = g; ch0 < nblk; ch0 += 32 * 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int ch = ch0 + 32 * i; v[i] = part[(long long)min(ch, nblk - 1) * H + n]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) s += ch0 + 32 * i < nblk ? v[i] : 0.f;
        }
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && n < H) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[i][c];
        out[n] = t;
    }
    if (which == 0 && blockIdx.x == 0) {              // d out.bias: fixed-order tree over the partials
        __shared__ float sb[4];
        float v = 0.f;
        for (int i = threadIdx.x; i < nblk; i += 256) v += partial_b[i];
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0) sb[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) out_b[0] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    }
}

// Training-step fusion (ncx_train_tail): the out layer, the listwise loss / rank and the backward prelude in ONE pass over h_L.
// One wave per run of triplets, lane = 4 columns (H <= 256), the KB >= K rows of a triplet resident in registers:
//   scores[b,k] = h[(b,k),:] . w_out + b_out                       (k_scores)
//   loss_rows / dscores / rank of the triplet                      (k_loss_rank: lane k holds score k)
//   dpre, dSh, partial sums of d out.weight / d out.bias / d linear_1.bias   (k_bwd_prelude, same wave partition)
// Same per-lane arithmetic and the same wave reductions as the three kernels it replaces: scores, loss, ranks and every
// gradient except d out.bias (a sum of zeros-in-maths; other order) are bit-identical to the unfused path.
// FULL: H == 256 and K == KB exactly (every lane owns 4 in-range columns, every row slot a real row): the edge handling -- a branch per store, a repair per load -- is compiled
// out; the kernel is ONE wave's instruction stream per triplet and runs as long as that stream is (round 3: 5 100 -> see DESIGN 7)
template <int KB, bool FULL>
__global__ __launch_bounds__(64) void k_train_tail(const float* __restrict__ h, const float* __restrict__ w_out,
                                                    const float* __restrict__ b_out, const int* __restrict__ gt, int B, int K, int H,
                                                    float loss_scale, float gate_scale, float* __restrict__ scores,
                                                    float* __restrict__ loss_rows, float* __restrict__ dscores, int* __restrict__ rank,
                                                    float* __restrict__ dpre, float* __restrict__ dsh,
                                                    float* __restrict__ partial_w, float* __restrict__ partial_b1,
                                                    float* __restrict__ partial_b, float* __restrict__ zero_buf, long long zero_n, int zchunk) {
    if (FULL) { H = 256; K = KB; }                          // (compile-time extents: row addresses become base + constant, no clamps)
    const int blk = blockIdx.x, nblk = gridDim.x;
    if (zero_buf) {          // (as k_bwd_prelude: dGgt is scattered into zeros later in the backward; zchunk = ceil(zero_n / nblk) up to a
                             // multiple of 4, from the host: a 64-bit division here is ~300 scalar instructions of a 2 400-instruction kernel)
        const long long z0 = (long long)blk * zchunk, z1 = min(z0 + zchunk, zero_n);
        for (long long i = z0 + 4 * threadIdx.x; i < z1; i += 256) {
            if (i + 3 < z1) *(f32x4u*)(zero_buf + i) = f32x4{0.f, 0.f, 0.f, 0.f};
            else for (long long j = i; j < z1; ++j) zero_buf[j] = 0.f;
        }
    }
    const int lane = threadIdx.x;                           // one wave per block: 512 triplets spread over all CUs, not 128 of them
    const int wv = blk, nwv = nblk;
    const int b0 = (int)((unsigned)B * (unsigned)wv / (unsigned)nwv), b1 = (int)((unsigned)B * (unsigned)(wv + 1) / (unsigned)nwv);   // (B * nwv < 2^32: B <= 32768, nwv <= 4096)
    const int c = lane * 4;
    const bool live = FULL || c < H, edge = !FULL && c + 4 > H;
    auto put4 = [&](float* p, const f32x4& v) __attribute__((always_inline)) {
        if (FULL || c + 3 < H) *(f32x4u*)(p + c) = v;
        else if (live) { p[c] = v[0]; if (c + 1 < H) p[c + 1] = v[1]; if (c + 2 < H) p[c + 2] = v[2]; }
    };
    const f32x4 w = FULL ? *(const f32x4u*)(w_out + c) : fix_window(load_window(w_out, c, H), c, H);          // (lanes beyond H: all zeros)
    const float bias = b_out[0];
    f32x4 aw = {0.f, 0.f, 0.f, 0.f}, ab1 = {0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
    for (int b = b0; b < b1; ++b) {
        const long long r0 = (long long)b * K;
        f32x4 hv[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) hv[k] = FULL ? *(const f32x4u*)(h + (r0 + min(k, K - 1)) * H + c) : load_window(h + (r0 + min(k, K - 1)) * H, c, H);
        const int g = gt[b];
        if (edge) {
#pragma unroll
            for (int k = 0; k < KB; ++k) hv[k] = fix_window(hv[k], c, H);
        }
        float sc = 0.f;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const float t = wave_sum(0.f + dot4(hv[k], w));
            sc = lane == k ? t + bias : sc;
        }
        if (lane < K) scores[r0 + lane] = sc;
        // listwise softmax cross-entropy + rank (k_loss_rank)
        const float s = lane < K ? sc : -INFINITY;
        const float m = wave_max(s);
        const float e = lane < K ? __expf(s - m) : 0.f;
        const float sum = wave_sum(e);
        const float sg = __shfl(s, g, 64);
        const float dsc = lane < K ? (e / sum - (lane == g ? 1.f : 0.f)) * loss_scale : 0.f;
        if (dscores && lane < K) dscores[r0 + lane] = dsc;
        const bool ahead = lane < K && (s > sg || (s == sg && lane < g));
        const unsigned long long bal = __ballot(ahead);
        if (lane == 0) { loss_rows[b] = (logf(sum) + m - sg) * loss_scale; rank[b] = __popcll(bal); }
        sb += dsc;
        // backward of out + dropout + relu (k_bwd_prelude)
        f32x4 ds = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (k < K) {                                     // (uniform; a `break` would keep the loop rolled and hv[] in scratch)
                const float gk = __shfl(dsc, k, 64);
                f32x4 dp;
#pragma unroll
                for (int q = 0; q < 4; ++q) { dp[q] = hv[k][q] > 0.f ? gk * w[q] * gate_scale : 0.f; aw[q] = __builtin_fmaf(gk, hv[k][q], aw[q]); ds[q] += dp[q]; }
                put4(dpre + (r0 + k) * H, dp);
            }
        }
        if (dsh) { put4(dsh + (long long)b * H, ds); ab1 += ds; }
    }
    put4(partial_w + (long long)wv * H, aw);
    if (dsh) put4(partial_b1 + (long long)wv * H, ab1);
    sb = wave_sum(sb);
    if (lane == 0) partial_b[wv] = sb;
}

// dsh[b][n] = sum_k dpre[b*K + k][n]
__global__ __launch_bounds__(256) void k_rowgroup_sum(const float* __restrict__ dpre, int B, int K, int H,
                                                      float* __restrict__ dsh) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * H) return;
    const int b = (int)(i / H), n = (int)(i - (long long)b * H);
    const float* p = dpre + (long long)b * K * H + n;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += p[(long long)k * H];
    dsh[i] = s;
}

// dggt[n][a] = sum over {b : aid[b] == a} of dsh[b][n]   (dggt [H][A] pre-zeroed).  One block per triplet; the first
// occurrence of an answer id owns column a and adds its duplicates in batch order -> deterministic, no atomics on
// floats.  Backward of the a_emb_gt lookup (cx.py:280) in re-associated form:
//   dE += S^T (dSh . W1[:, a_emb_gt]) = (S^T dSh) . W1[:, a_emb_gt],  S = one-hot(aid)  -> a second pair of the dE GEMM.
constexpr int NCX_SCATTER_MAX_B = 32768;
__global__ __launch_bounds__(256) void k_scatter_dsh_by_answer(const float* __restrict__ dsh, const int* __restrict__ aid,
                                                               int B, int H, int A, float* __restrict__ dggt) {
    __shared__ unsigned bits[NCX_SCATTER_MAX_B / 32];
    const int b = blockIdx.x;
    const int id = aid[b];
    int earlier = 0;
    for (int j = threadIdx.x; j < b; j += 256) earlier |= aid[j] == id;
    if (__syncthreads_or(earlier)) return;                       // not the owner (uniform per block)
    const int nw = (B + 31) / 32;
    for (int w = threadIdx.x; w < nw; w += 256) bits[w] = 0u;
    __syncthreads();
    for (int j = b + threadIdx.x; j < B; j += 256)
        if (aid[j] == id) atomicOr(&bits[j >> 5], 1u << (j & 31));
    __syncthreads();
    for (int n = threadIdx.x; n < H; n += 256) {
        float s = 0.f;
        for (int w = b >> 5; w < nw; ++w) {
            unsigned m = bits[w];
            while (m) {
                const int j = (w << 5) + __ffs(m) - 1;
                m &= m - 1;
                s += dsh[(long long)j * H + n];
            }
        }
        dggt[(long long)n * A + id] = s;
    }
}

// fp32 path: the answer-embedding gradient runs on the fused forward kernel (ncx_main.h, NT form, 64 x 64 tiles at three
// workgroups per CU: 50 us against 80 us for the TN form of the generic engine -- tools/mb/mb_main.hip, "short chain"):
//   dE[a][j] = sum_n dGt^T[a][n] W1ak^T[j][n] + sum_n dGgt^T[a][n] W1agt^T[j][n]
// whose operands are rows with the reduction index contiguous.  ONE launch prepares them (roles by block range):
//   [0, B)            dGgt^T[id][n] = sum over {b : aid[b] == id} of dSh[b][n]   (owner-computes scatter as above, rows contiguous now;
//                     dGgt^T was cleared by k_bwd_prelude)
//   [B, B + T)        32 x 32 transposes through LDS: dGt [H][A] -> dGt^T [A][Hp4];  W1[:, a_other], W1[:, a_gt] -> [da][Hp32]
//                     (columns beyond H zero: the kernel's weight-side padding)
struct EmbPrepArgs {
    const float* dsh; const int* aid; float* dggtT;        // scatter
    const float* src[3]; float* dst[3]; long long lds_[3]; int cols[3], ldd[3], dcols[3], tile0[4];   // transposes: src [H][cols] -> dst [cols][ldd], dst cols < dcols written
    int B, H, A, Hp4;
};
// `fix` (valid: blocks beyond a.B + a.tile0[3]): the deferred split fix-up of the dW1[:, a_other] GEMM (64 x 64 tiles) rides along
__global__ __launch_bounds__(256) void k_emb_prep(const EmbPrepArgs a, const FixupArgs fix, const Tn8ReduceArgs red) {
    __shared__ unsigned bits[NCX_SCATTER_MAX_B / 32 > 32 * 33 ? NCX_SCATTER_MAX_B / 32 : 32 * 33];     // scatter: id bitmap; transposes: a [32][33] tile
    if ((int)blockIdx.x >= a.B + a.tile0[3]) {
        const int id = blockIdx.x - (a.B + a.tile0[3]);
        const int nfix = fix.valid ? fix.grid_x * 4 : 0;
        if (id < nfix) split_fixup_body<64, 64>(fix, id / 4, id % 4);
        else tn8_reduce_body(red, id - nfix);            // the partial tiles of the dW1[:, a_other] launch (ncx_dwtn.hip)
        return;
    }
    if ((int)blockIdx.x < a.B) {
        const int b = blockIdx.x, B = a.B;
        const int id = a.aid[b];
        int earlier = 0;
        for (int j = threadIdx.x; j < b; j += 256) earlier |= a.aid[j] == id;
        if (__syncthreads_or(earlier)) return;                       // not the owner (uniform per block)
        const int nw = (B + 31) / 32;
        for (int w = threadIdx.x; w < nw; w += 256) bits[w] = 0u;
        __syncthreads();
        for (int j = b + threadIdx.x; j < B; j += 256)
            if (a.aid[j] == id) atomicOr(&bits[j >> 5], 1u << (j & 31));
        __syncthreads();
        for (int n = threadIdx.x; n < a.H; n += 256) {
            float s = 0.f;
            for (int w = b >> 5; w < nw; ++w) {
                unsigned m = bits[w];
                while (m) { const int j = (w << 5) + __ffs(m) - 1; m &= m - 1; s += a.dsh[(long long)j * a.H + n]; }
            }
            a.dggtT[(long long)id * a.Hp4 + n] = s;
        }
        return;
    }
    float* tile = (float*)bits;                                        // [32][33]
    int t = blockIdx.x - a.B, e = 0;
    while (e < 2 && t >= a.tile0[e + 1]) ++e;
    t -= a.tile0[e];
    const int tiles_h = (a.dcols[e] + 31) / 32;                        // tiles along the source-row (h) direction
    const int th = t % tiles_h, tc = t / tiles_h;
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;              // 32 x 8 threads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = th * 32 + y + 8 * i, c = tc * 32 + x;
        tile[(y + 8 * i) * 33 + x] = (h < a.H && c < a.cols[e]) ? a.src[e][(long long)h * a.lds_[e] + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tc * 32 + y + 8 * i, h = th * 32 + x;
        if (c < a.cols[e] && h < a.dcols[e]) a.dst[e][(long long)c * a.ldd[e] + h] = tile[x * 33 + y + 8 * i];
    }
}

__global__ __launch_bounds__(256) void k_zero_cols(float* __restrict__ p, int rows, long long ld, int cols) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    p[(long long)r * ld + c] = 0.f;
}

// torch.optim.Adam (counterexamples.py:275-276): m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n, float step_size, float b1, float b2,
                                              float eps, float inv_bc2_sqrt, float gscale) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 pv = *(f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gj = gv[j] * gscale;
                mv[j] = mv[j] + (gj - mv[j]) * (1.f - b1);
                vv[j] = vv[j] * b2 + (1.f - b2) * gj * gj;
                const float den = sqrtf(vv[j]) * inv_bc2_sqrt + eps;
                pv[j] = pv[j] - step_size * (mv[j] / den);
            }
            *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
        } else {
            for (size_t j = i; j < n; ++j) {
                const float gj = g[j] * gscale;
                const float mj = m[j] + (gj - m[j]) * (1.f - b1);
                const float vj = v[j] * b2 + (1.f - b2) * gj * gj;
                m[j] = mj; v[j] = vj;
                p[j] = p[j] - step_size * (mj / (sqrtf(vj) * inv_bc2_sqrt + eps));
            }
        }
    }
}

// =================================================================================================
// planning
// =================================================================================================
static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Split planning.  A problem whose whole tiles cannot fill the chip is split along K into S aligned chunks and
// run as W = tiles*S equal workgroups through the stream-K path (W = tiles*S makes the unit ranges coincide with
// the chunks, so workgroups on the same chunk of different tiles share operand rows in L2; unaligned ranges lost
// that sharing and ran 1.3x slower).  Equal workgroups execute in rounds of `slots` = CUs x resident
// workgroups per CU, so S is chosen to fill whole rounds (1648 workgroups on 512 slots = 3.2 rounds cost 4).
int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else { (void)hipGetLastError(); n = 256; }
    }
    return n;
}

// Measured (DW1C on 512 slots, 412 tiles x 384 k-steps): S = 4/5/6/7/9/12/16/24/32 -> 0.541/0.538/0.521/0.513/
// 0.486/0.486/0.486/0.508/0.525 ms: aim for >= 6 rounds of workgroups but keep >= 24 k-steps in each.
static int choose_split(long long tiles, long long ksteps, long long slots) {
    if (tiles >= slots) return 1;                  // at least one full round of workgroups already (dE at H=1024: 217 us unsplit, 230 us x2)
    long long sp = cdiv(6 * slots, tiles);
    const long long smax = ksteps / 24 > 1 ? ksteps / 24 : 1;
    if (sp > smax) sp = smax;
    if (tiles * sp < slots) {                      // cannot even fill one round: trade pipeline depth for parallelism
        const long long smax2 = ksteps / 8 > 1 ? ksteps / 8 : 1;
        sp = cdiv(slots, tiles);
        if (sp > smax2) sp = smax2;
    }
    return (int)(sp < 1 ? 1 : sp);
}

// tiles_small: output tiles with 64x64 blocks; ksteps: 32-deep reduction steps per tile.
static GemmPlan plan_from_tiles(int form, long long tiles_big, long long tiles_small, long long ksteps, bool big_ok) {
    GemmPlan p; p.split = 1;
    // 128x128 tiles run one workgroup per CU: only when they fill whole rounds of CUs reasonably (dE at H=1024: 304 tiles =
    // 1.19 rounds took 362 us, the 64x64 plan 217 us)
    const long long cus = num_cus();
    const double eff_big = (double)tiles_big / (double)(cdiv(tiles_big, cus) * cus);
    if (big_ok && tiles_big >= 192 && ksteps >= 64 && eff_big >= 0.75) { p.cfg = CFG_128x128; return p; }
    // Measured on MI355X (DW1C, 412 small tiles x 384 k-steps): every TN tile shape (64x64, 128x64, 128x128) ends
    // at 0.45-0.51 ms; the 64x64 tile fits 3 workgroups per CU and is used for all split problems.
    p.cfg = CFG_64x64;
    const int occ = form == FORM_NT ? occupancy_nt(p.cfg) : form == FORM_TN ? occupancy_tn(p.cfg) : occupancy_nn(p.cfg);
    p.split = choose_split(tiles_small, ksteps, (long long)occ * num_cus());
    return p;
}

GemmPlan plan_gemm(int form, long long M, long long N, long long ksteps, bool allow_96) {
    // Epilogue GEMMs (forward layers) never split; both big NT tiles run one workgroup per CU, so pick the one
    // whose tile count fills whole rounds of CUs better (H=512: 384 tiles of 128x128 = 1.5 rounds, 512 of 96x128 = 2).
    // Short reductions (<= 32 k-steps, e.g. the 360-wide MUTAN / classifier products): the per-workgroup prologue and
    // epilogue of the big tiles dominate; measured at M=12288, N=2000, K=360: 645 / 335 / 245 us for 128x128 / 96x128 / 64x64.
    if (form == FORM_NT && allow_96 && ksteps <= 32 && M * N >= 64 * 64 * 512) {
        GemmPlan p; p.cfg = CFG_64x64; p.split = 1; return p;
    }
    if (form == FORM_NT && allow_96 && M >= 96 && N >= 128) {
        const double cus = (double)num_cus();
        const double t128 = (double)(cdiv(M, 128) * cdiv(N, 128)), t96 = (double)(cdiv(M, 96) * cdiv(N, 128));
        const double waste128 = (double)(cdiv(M, 128) * 128 * cdiv(N, 128) * 128) / (double)(M * N);
        const double waste96 = (double)(cdiv(M, 96) * 96 * cdiv(N, 128) * 128) / (double)(M * N);
        const double e128 = (t128 / cus) / (double)cdiv((long long)t128, (long long)cus) / waste128;
        const double e96 = (t96 / cus) / (double)cdiv((long long)t96, (long long)cus) / waste96;
        if (t96 >= 0.75 * cus || t128 >= 0.75 * cus) {
            GemmPlan p; p.split = 1; p.cfg = e96 > e128 * 1.02 ? CFG_96x128 : CFG_128x128; return p;
        }
    }
    const bool big_ok = M >= 96 && N >= 96;
    return plan_from_tiles(form, cdiv(M, 128) * cdiv(N, 128), cdiv(M, 64) * cdiv(N, 64), ksteps, big_ok);
}

// The GEMMs of one step, so that ws_layout and forward/backward agree on split-K slab sizes.
struct GemmUse { int form; long long M, N, ksteps; bool allow96; GemmPlan plan; long long slab_elems; long long tiles; long long wgs; };
enum { U_GT = 0, U_SH, U_MAIN, U_FWD_L, U_DW1C, U_DW1S, U_DE, U_DW1AK, U_DAGT, U_DWL, U_DXL, U_COUNT };

static inline long long ks(long long k) { return cdiv(k, GEMM_BK); }

// Split of one candidate-column problem of the grouped dW1 launch.  The big problems (2048 / 2000 columns) fill whole
// rounds of workgroup slots; the narrow ones (dist + rank: 25 columns, z_other: 360) are dispatched after them and would
// keep a few CUs busy for a full-length tail: they get twice as many, shorter k-chunks (still >= 8 k-steps each).
static int dw1c_seg_split(long long cols, int S, long long ksteps) {
    long long narrow = 512;
    if (const char* e = hook_env("NCX_SMALL_COLS")) narrow = atoll(e);
    if (S <= 1 || cols > narrow) return S;
    int mult = 2;                     // measured at C2: x1 0.445-0.450 ms, x2 0.427, x3 0.438, x4 0.443
    if (const char* e = hook_env("NCX_SMALL_MULT")) mult = atoi(e) > 0 ? atoi(e) : 1;
    long long sp = (long long)S * mult;
    const long long smax = ksteps / 8 > 1 ? ksteps / 8 : 1;
    if (sp > smax) sp = smax;
    return (int)(sp < S ? S : sp);
}

static void list_uses(const ncx_dims& d, GemmUse* u) {
    const long long M = (long long)d.B * d.K, H = d.H;
    const bool aemb = d.flags & NCX_F_A_EMB;
    const long long cand_k = ks(d.dv) * ((d.flags & NCX_F_V_MULT) ? 2 : 1) + ks(d.K + 1) + ks(d.dz) + (aemb ? ks(d.A) : ks(d.da));
    const long long cand_cols = (long long)d.dv * ((d.flags & NCX_F_V_MULT) ? 2 : 1) + d.K + 1 + d.dz + (aemb ? d.A : d.da);
    const long long sh_k = ks(d.dv) + ks(d.dq) + ks(d.dz) + ks(d.da);
    const long long sh_cols = (long long)d.dv + d.dq + d.dz + d.da;
    u[U_GT]    = {FORM_NT, H, d.A, ks(d.da), false};
    u[U_SH]    = {FORM_NT, d.B, H, sh_k, false};
    u[U_MAIN]  = {FORM_NT, M, H, cand_k, true};
    u[U_FWD_L] = {FORM_NT, M, H, ks(H), true};
    u[U_DW1C]  = {FORM_TN, H, cand_cols, ks(M), false};
    u[U_DW1S]  = {FORM_TN, H, sh_cols, ks(d.B), false};
    u[U_DE]    = {FORM_TN, d.A, d.da, 2 * ks(H), false};
    u[U_DW1AK] = {FORM_NN, H, d.da, ks(d.A), false};
    u[U_DAGT]  = {FORM_NN, 1, 1, 1, false};                 // (folded into U_DE)
    u[U_DWL]   = {FORM_TN, H, H, ks(M), false};
    u[U_DXL]   = {FORM_NN, M, H, ks(H), false};
    const bool hooks = experiment_hooks_on();
    for (int i = 0; i < U_COUNT; ++i) {
        // grouped launches (DW1C + DW1S): tiles are counted per column segment
        const bool km = dw_km_supported(d) && !(d.flags & NCX_F_BF16);     // v_other / v_mult columns: ncx_dwkm.hip
        const bool tn8 = dw_tn8_supported(d);                               // every other column block + dGt: ncx_dwtn.hip
        const long long segs_c[5] = {km ? 0 : d.dv, (!km && (d.flags & NCX_F_V_MULT)) ? d.dv : 0, tn8 ? 0 : d.K + 1, tn8 ? 0 : d.dz, tn8 ? 0 : (aemb ? d.A : d.da)};
        const bool tn8s = tn8 || dw_tn8_shapes_ok(d);                       // (the bf16 variant: its fp32 shared segments take the TN kernel too)
        const long long segs_s[5] = {tn8s ? 0 : d.dv, tn8s ? 0 : d.dq, tn8s ? 0 : d.dz, tn8s ? 0 : d.da, 0};
        const bool grouped = i == U_DW1C || i == U_DW1S;
        auto grouped_tiles = [&](int bm, int bn) {
            const long long* sg = i == U_DW1C ? segs_c : segs_s;
            long long t = 0;
            for (int q = 0; q < 5; ++q) t += cdiv(H, bm) * cdiv(sg[q], bn);
            return t;
        };
        if (grouped && grouped_tiles(64, 64) == 0) {      // nothing left for the grouped launch
            u[i].plan.cfg = CFG_128x64; u[i].plan.split = 1;
        } else if (grouped) {
            u[i].plan = plan_from_tiles(FORM_TN, grouped_tiles(128, 128), grouped_tiles(64, 64), u[i].ksteps, false);
            // Measured on MI355X (C2, H=256: 206 tiles of 128x64 x 384 k-steps): 128x64 with 6-8 k-chunks 0.448 ms vs
            // 0.476 ms for 64x64 x 8; three rounds of the 2-per-CU slots.
            if (i == U_DW1C && H >= 128 && u[i].plan.split > 1) {
                u[i].plan.cfg = CFG_128x64;
                const long long slots = (long long)occupancy_tn(CFG_128x64) * num_cus();
                long long sp = cdiv(3 * slots, grouped_tiles(128, 64));
                const long long smax = u[i].ksteps / 24 > 1 ? u[i].ksteps / 24 : 1;
                if (sp > 8) sp = 8;          // one k-chunk per XCD at most (measured with the fused v-column kernel: x8 0.346, x12 0.354, x16 0.363 ms)
                u[i].plan.split = (int)(sp > smax ? smax : sp < 1 ? 1 : sp);
            }
        } else {
            u[i].plan = plan_gemm(u[i].form, u[i].M, u[i].N, u[i].ksteps, u[i].allow96);
        }
        if (i == U_MAIN || i == U_FWD_L || i == U_DXL) u[i].plan.split = 1;       // epilogue GEMMs never split
        if (hooks) {   // experiment hooks (NCX_EXPERIMENT=1): NCX_SPLIT_<id>=S forces S aligned k-chunks, NCX_CFG_<id> the tile config
            char name[32];
            snprintf(name, sizeof name, "NCX_CFG_%d", i);
            const char* c = getenv(name);
            if (c) u[i].plan.cfg = atoi(c);
            snprintf(name, sizeof name, "NCX_SPLIT_%d", i);
            const char* e = getenv(name);
            if (e && i != U_MAIN && i != U_FWD_L && i != U_DXL) u[i].plan.split = atoi(e) > 1 ? atoi(e) : 1;
        }
        int bm, bn; cfg_tile(u[i].plan.cfg, bm, bn);
        long long tiles = cdiv(u[i].M, bm) * cdiv(u[i].N, bn);
        if (grouped) tiles = grouped_tiles(bm, bn);
        u[i].tiles = tiles;
        // slab slots = workgroup ids (the chunk-per-XCD layout of WgMap pads some problems)
        const int S = u[i].plan.split > 1 ? u[i].plan.split : 1;
        long long wgs = 0;
        if (grouped) {
            const long long* sg = i == U_DW1C ? segs_c : segs_s;
            for (int q = 0; q < 5; ++q)
                if (sg[q] > 0) wgs += WgMap{(int)cdiv(H, bm), (int)cdiv(sg[q], bn), i == U_DW1C ? dw1c_seg_split(sg[q], S, u[i].ksteps) : S}.count();
        } else {
            wgs = WgMap{(int)cdiv(u[i].M, bm), (int)cdiv(u[i].N, bn), S}.count();
        }
        u[i].wgs = wgs;
        u[i].slab_elems = S > 1 ? wgs * bm * bn : 0;
    }
    // DW1S rides in DW1C's launch (same tile config): its workgroups' slab slots follow DW1C's
    u[U_DW1S].plan.cfg = u[U_DW1C].plan.cfg;
    {
        int bm, bn; cfg_tile(u[U_DW1C].plan.cfg, bm, bn);
        const bool tn8 = dw_tn8_shapes_ok(d);
        const long long segs_s[4] = {tn8 ? 0 : d.dv, tn8 ? 0 : d.dq, tn8 ? 0 : d.dz, tn8 ? 0 : d.da};
        const int S = u[U_DW1S].plan.split > 1 ? u[U_DW1S].plan.split : 1;
        u[U_DW1S].tiles = 0; u[U_DW1S].wgs = 0;
        for (int q = 0; q < 4; ++q) {
            if (segs_s[q] == 0) continue;
            u[U_DW1S].tiles += cdiv(H, bm) * cdiv(segs_s[q], bn);
            u[U_DW1S].wgs += WgMap{(int)cdiv(H, bm), (int)cdiv(segs_s[q], bn), S}.count();
        }
        u[U_DW1C].slab_elems = (u[U_DW1C].wgs + u[U_DW1S].wgs) * bm * bn;
        u[U_DW1S].slab_elems = 0;
    }
}

WsLayout ws_layout(const ncx_dims& d) {
    WsLayout w{};
    const size_t M = (size_t)d.B * d.K, H = d.H;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.idx_k = take(M * 4); w.idx_o = take(M * 4); w.idx_ob = take((size_t)d.B * 4);
    w.mx = take(M * 4); w.inv = take(M * 4);
    w.ldm = pad_to(d.K + 1, 4);
    w.ldgt = (d.flags & NCX_F_BF16) ? d.A : pad_to(d.A, 32);
    w.misc = take(M * w.ldm * 4);
    w.gt = take(H * w.ldgt * 4);
    w.sh = take((size_t)d.B * H * 4);
    for (int l = 0; l < 3; ++l) w.h[l] = l < d.L ? take(M * H * 4) : 0;
    w.dpre[0] = take(M * H * 4);
    w.dpre[1] = d.L >= 2 ? take(M * H * 4) : 0;
    w.dsh = take((size_t)d.B * H * 4);
    w.dgt = take(2 * H * d.A * 4);                      // dGt[H][A], then (contiguous: one all-reduce bucket under DP)
    w.dagt = w.dgt + H * d.A * 4;                       // dGgt[H][A] = one-hot(aid)^T dSh, transposed
    w.dgtT = take((size_t)2 * d.A * pad_to(d.H, 4) * 4);
    w.w1aT = take((size_t)2 * d.da * pad_to(d.H, 32) * 4);
    w.dgtT2 = take(dw_tn8_supported(d) ? (size_t)pad_to(d.A, 32) * pad_to(d.H, 4) * 4 : 0);
    w.partial = take((size_t)NCX_PRELUDE_WAVES * H * 4 * 2 + (size_t)NCX_PRELUDE_WAVES * 4 + 256);     // (>= NCX_COLSUM_CHUNKS rows)
    GemmUse u[U_COUNT];
    list_uses(d, u);
    long long slab = 0;
    for (int i = 0; i < U_COUNT; ++i) slab = u[i].slab_elems > slab ? u[i].slab_elems : slab;
    w.slab_bytes = (size_t)slab * 4;
    if (dw_tn8_slab_bytes(d) > w.slab_bytes) w.slab_bytes = dw_tn8_slab_bytes(d);      // ncx_dwtn.hip: its partial tiles live here too
    w.slab = take(w.slab_bytes);
    {   // side-stream GEMMs (Gt, Sh forward; dW1ak, dE backward) get their own slab
        long long s2 = u[U_GT].slab_elems;
        if (u[U_SH].slab_elems > s2) s2 = u[U_SH].slab_elems;
        if (u[U_DW1AK].slab_elems > s2) s2 = u[U_DW1AK].slab_elems;
        if (u[U_DE].slab_elems > s2) s2 = u[U_DE].slab_elems;
        w.slab2_bytes = (size_t)s2 * 4;
        w.slab2 = take(w.slab2_bytes);
    }
    w.km_slab = take(dw_km_slab_bytes(d));
    {   // split-K slabs of the fused forward kernel: linear_1 / hidden layers at small batches
        const long long Tm = ks(d.dv) * ((d.flags & NCX_F_V_MULT) ? 2 : 1) + ks(pad_to(d.K + 1, 4)) + ks(d.dz) + ((d.flags & NCX_F_A_EMB) ? ks(d.A) : ks(d.da));
        auto need = [&](long long m, long long n, long long t) { const int sp = main_split(m, n, t); return sp > 1 ? (size_t)sp * m * n * 4 : (size_t)0; };
        size_t b = need((long long)M, (long long)H, Tm);
        if (d.L >= 2) b = b > need((long long)M, (long long)H, ks(H)) ? b : need((long long)M, (long long)H, ks(H));
        w.mslab_bytes = b;
        w.mslab = take(b);
    }
    {   // padded weight copies for the fused forward kernel: [H][pad32(width)] each, in the order pack_wpad fills them
        size_t e = 0;
        for (int i = 0; i < WPAD_N; ++i) e += (size_t)H * wpad_width(d, i);
        w.wpad = take(e * 4);
    }
    if (d.flags & NCX_F_BF16) {                          // packed bf16 operands of the two dominant GEMMs (ncx_bf16.h)
        w.xc = take(bf16_xc_bytes(d)); w.wc = take(bf16_wc_bytes(d));
        w.dpre_bf = take(bf16_dpre_bytes(d)); w.bf_slab = take(bf16_slab_bytes(d));
        w.bf_emb = take(bf16_emb_bytes(d));
    }
    w.total = off;
    return w;
}

// Internal side stream: independent small kernels (each under-fills 256 CUs, or is HBM-bound while the other is
// MFMA-bound) are forked from the caller's stream and joined back with events, so the work stays fully ordered
// with respect to `stream`.  One lazily created (stream, 2 events) triple per device.  Measured on MI355X at
// configs[1]: round 1 (Gt, Sh || k_prep; dW1ak || dE) 1.236-1.253 ms/step with it vs 1.221-1.228 without; round 2 (the
// whole answer-embedding chain || k_dw_km as well) 1.001-1.003 vs 0.985-0.992: the fork/join events cost what the
// overlap wins and a full round of long workgroups leaves the short ones no slots, so it is OFF unless NCX_SIDE_STREAM=1.
struct SideStream { hipStream_t s; hipEvent_t fork, join; int state, mode; };     // state: 0 new, 1 ready, -1 unavailable; mode = NCX_SIDE_STREAM (1 both passes, 2 forward only, 3 backward only)
static SideStream* side_stream() {
    static SideStream tab[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
    SideStream& t = tab[dev];
    if (t.state == 0) {
        const char* on = getenv("NCX_SIDE_STREAM");
        t.state = -1;
        t.mode = on ? atoi(on) : 0;
        if ((on && atoi(on)) &&
            hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess) t.state = 1;
        else (void)hipGetLastError();
    }
    return t.state == 1 ? &t : nullptr;
}
// fork: the side stream waits for everything enqueued on `main` so far
static int side_fork(SideStream* ss, hipStream_t main) {
    NCX_HIP_TRY(hipEventRecord(ss->fork, main));
    NCX_HIP_TRY(hipStreamWaitEvent(ss->s, ss->fork, 0));
    return 0;
}
// join: `main` waits for everything enqueued on the side stream so far
static int side_join(SideStream* ss, hipStream_t main) {
    NCX_HIP_TRY(hipEventRecord(ss->join, ss->s));
    NCX_HIP_TRY(hipStreamWaitEvent(main, ss->join, 0));
    return 0;
}

static int check_dims(const ncx_dims* d) {
    if (!d) return NCX_E_NULL;
    if (d->B < 1 || d->K < 1 || d->K > 64 || d->dv < 1 || d->dq < 1 || d->dz < 1 || d->da < 1 || d->A < 1 ||
        d->H < 1 || d->L < 1 || d->L > 3 || d->n_img < 1)
        return NCX_E_DIMS;
    if ((long long)d->B * d->K > (1ll << 30) / 4 || d->B > NCX_SCATTER_MAX_B) return NCX_E_DIMS;
    if (d->dv < 4 || d->dq < 4 || d->dz < 4 || d->da < 4 || d->A < 4 || d->H < 4 || d->K < 3) return NCX_E_DIMS;   // 16-byte windows
    if (d->drop_p < 0.f || d->drop_p >= 1.f) return NCX_E_DIMS;
    if (d->flags & ~(NCX_F_ALL | NCX_F_BF16 | NCX_F_REUSE_GT | NCX_F_FUSED_TAIL | NCX_F_X6)) return NCX_E_FLAGS;
    if ((d->flags & NCX_F_BF16) && (d->flags & NCX_F_ALL) != NCX_F_ALL) return NCX_E_FLAGS;    // bf16 variant: no lesions
    return NCX_OK;
}

// -------------------------------------------------------------------------------------------------
// Opt-in diagnostics (ncx_profile_begin/_end): HIP events around every launch of ONE chosen GEMM, on the
// stream it is launched on.  Off by default; the only process-global state of the library.
// -------------------------------------------------------------------------------------------------
struct ProfState { bool on; unsigned mask; int n, cap; hipEvent_t* ev; int* ids; };
static ProfState g_prof = {false, 0u, 0, 0, nullptr, nullptr};
// ncx_profile_stamps: in-kernel clock stamps of MAIN (diagnostic passes only).  One slot per DEVICE, like side_stream()'s table: the
// buffer is a device pointer, so a forward on another GPU of the same process must never see it; pointer and capacity are published
// together (the capacity is written first, the pointer last, and read in the opposite order) so a concurrent forward sees either
// the old pair or the new one.
struct StampSlot { std::atomic<unsigned long long*> ptr; std::atomic<long long> words; };
static StampSlot g_stamps[16];
static inline unsigned long long* stamps_for_current_device(long long need_words) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
    unsigned long long* p = g_stamps[dev].ptr.load(std::memory_order_acquire);
    if (!p || g_stamps[dev].words.load(std::memory_order_acquire) < need_words) return nullptr;
    return p;
}

static int run_gemm_impl(GemmArgs& a, int form, const GemmPlan& pl, float* slab, size_t slab_bytes,
                         const float* reduce_bias, hipStream_t s);

// HIP-event bracket of a launch sequence that is not a run_gemm call (the bf16 variant's GEMMs)
static int prof_open(int use_id, hipStream_t s) {
    if (g_prof.on && ((g_prof.mask >> use_id) & 1u) && g_prof.n < g_prof.cap) NCX_HIP_TRY(hipEventRecord(g_prof.ev[2 * g_prof.n], s));
    return NCX_OK;
}
static int prof_close(int use_id, hipStream_t s) {
    if (g_prof.on && ((g_prof.mask >> use_id) & 1u) && g_prof.n < g_prof.cap) {
        NCX_HIP_TRY(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s)); g_prof.ids[g_prof.n] = use_id; ++g_prof.n;
    }
    return NCX_OK;
}

// GEMM driver: runs `a` with plan `pl`; when split, redirects the outputs to slabs and reduces them.
static int run_gemm(int use_id, GemmArgs& a, int form, const GemmPlan& pl, float* slab, size_t slab_bytes,
                    const float* reduce_bias, hipStream_t s) {
    const bool rec = g_prof.on && ((g_prof.mask >> use_id) & 1u) && g_prof.n < g_prof.cap;
    if (rec) NCX_HIP_TRY(hipEventRecord(g_prof.ev[2 * g_prof.n], s));
    const int rc = run_gemm_impl(a, form, pl, slab, slab_bytes, reduce_bias, s);
    if (rec) { NCX_HIP_TRY(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s)); g_prof.ids[g_prof.n] = use_id; ++g_prof.n; }
    return rc;
}

static int run_gemm_impl(GemmArgs& a, int form, const GemmPlan& pl, float* slab, size_t slab_bytes,
                         const float* reduce_bias, hipStream_t s) {
    { const char* nf = hook_env("NCX_NO_FAST"); a.pad_ = nf && atoi(nf) ? 1 : 0; }
    const int np = a.mode == MODE_GROUP ? a.nseg : 1;
    bool any = false;
    for (int i = 0; i < np; ++i) {
        if (a.split[i] == 0) a.split[i] = pl.split;              // callers may preset per-problem splits
        any |= a.split[i] > 1;
    }
    if (any) {
        int bm, bn; cfg_tile(pl.cfg, bm, bn);
        const long long wgs = gemm_layout(a, bm, bn, nullptr);
        if ((size_t)wgs * bm * bn * 4 > slab_bytes) return NCX_E_WORKSPACE;
        a.slab = slab;
    }
    if (reduce_bias) a.epi.bias = reduce_bias;                   // applied by the epilogue or by the fix-up kernel
    if (form == FORM_NT) return run_gemm_nt(a, pl.cfg, s);
    if (form == FORM_TN) return run_gemm_tn(a, pl.cfg, s);
    return run_gemm_nn(a, pl.cfg, s);
}

static void set_dropout(EpiArgs& e, const ncx_dims& d, const ncx_inputs& in, int layer, long long M) {
    e.relu = 1;
    if (d.training && d.drop_p > 0.f) {
        e.drop_p = d.drop_p; e.drop_scale = 1.f / (1.f - d.drop_p);
        if (in.keep_mask) { e.dropout = 2; e.keep_mask = in.keep_mask + (long long)(layer - 1) * M * d.H; e.ld_mask = d.H; }
        else { e.dropout = 1; e.seed_lo = (unsigned)(d.seed & 0xFFFFFFFFull); e.seed_hi = (unsigned)(d.seed >> 32); e.layer = (unsigned)layer; }
    }
}

}  // namespace ncx

using namespace ncx;

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* ncx_version(void) { return "neuralcx-hip gfx950 fp32-mfma r4 (" __DATE__ ")"; }

int64_t ncx_input_size(const ncx_dims* d) { return d ? (int64_t)seg_offsets(*d).din : 0; }

size_t ncx_workspace_bytes(const ncx_dims* d) {
    if (check_dims(d) != NCX_OK) return 0;
    return ws_layout(*d).total;
}

static int forward_impl(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                        size_t workspace_bytes, float* scores, void* stream_, int phase);

int ncx_forward(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                size_t workspace_bytes, float* scores, void* stream_) {
    return forward_impl(dp, in, p, workspace, workspace_bytes, scores, stream_, 0);
}

int ncx_forward_phase(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                      size_t workspace_bytes, float* scores, int32_t phase, void* stream_) {
    if (phase < 0 || phase > 2) return NCX_E_FLAGS;
    return forward_impl(dp, in, p, workspace, workspace_bytes, scores, stream_, phase);
}

// phase 0: everything.  phase 1 (NCX_FWD_PRELUDE): the part that is a function of the DATA only -- k_prep's row pass (table row
// ids, pairwise distance, rank one-hot, softmax statistics of the answer logits; the bf16 variant's row pack) -- so a
// data-parallel job can run it for step n + 1 while step n's last gradient bucket is still on the wire and its weights are
// not final.  phase 2 (NCX_FWD_REST): everything that reads the weights (the padded weight copies, Gt, Sh, the Linear layers,
// the scores).  1 then 2 == 0 bit for bit: the same kernels on the same operands, k_prep's launch cut between its row blocks
// and its weight-pack blocks.
static int forward_impl(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                        size_t workspace_bytes, float* scores, void* stream_, int phase) {
    int rc = check_dims(dp);
    if (rc != NCX_OK) return rc;
    const bool do_pre = phase != 2, do_rest = phase != 1;
    if (!in || !p || !workspace || (do_rest && !scores && !(dp->flags & NCX_F_FUSED_TAIL))) return NCX_E_NULL;
    const ncx_dims& d = *dp;
    const bool aemb = d.flags & NCX_F_A_EMB;
    if (!in->feats || !in->img_idx || !in->q_emb || !in->z_orig || !in->z_knns || !in->a_knns) return NCX_E_NULL;
    if (aemb && (!in->answer_aids || !p->answer_embedding)) return NCX_E_NULL;
    if (!aemb && !in->a_emb_gt) return NCX_E_FLAGS;
    if (!(d.flags & NCX_F_V_RANK) && !in->v_rank) return NCX_E_FLAGS;
    if (!p->w1 || !p->b1 || !p->w_out || !p->b_out) return NCX_E_NULL;
    if (d.L >= 2 && (!p->w2 || !p->b2)) return NCX_E_NULL;
    if (d.L >= 3 && (!p->w3 || !p->b3)) return NCX_E_NULL;
    const WsLayout w = ws_layout(d);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return NCX_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const int M = d.B * d.K, H = d.H;
    const SegOffsets o = seg_offsets(d);
    const long long din = o.din;
    int* idx_k = (int*)(ws + w.idx_k); int* idx_o = (int*)(ws + w.idx_o); int* idx_ob = (int*)(ws + w.idx_ob);
    float* mx = (float*)(ws + w.mx); float* inv = (float*)(ws + w.inv); float* misc = (float*)(ws + w.misc);
    float* gt = (float*)(ws + w.gt); float* sh = (float*)(ws + w.sh); float* slab = (float*)(ws + w.slab);
    GemmUse u[U_COUNT];
    list_uses(d, u);

    // k_prep (HBM-bound) on the caller's stream  ||  Gt, Sh (small MFMA GEMMs; Sh only needs idx_ob, which k_prep
    // produces, so it gathers through img_idx directly) on the side stream
    SideStream* ss = side_stream();
    if (ss && ss->mode == 3) ss = nullptr;
    hipStream_t s2 = ss ? ss->s : s;
    float* slab_side = ss ? (float*)(ws + w.slab2) : slab;
    const size_t slab_side_bytes = ss ? w.slab2_bytes : w.slab_bytes;
    if (ss) { rc = side_fork(ss, s); if (rc) return rc; }
    const bool bf16 = d.flags & NCX_F_BF16;
    u16* xc = bf16 ? (u16*)(ws + w.xc) : nullptr;
    // zero-padded copies of the weight slices the fused forward kernel reads past their width (+ the pad columns of Gt):
    // functions of the weights only, like Gt -- evaluation passes reuse them (NCX_F_REUSE_GT)
    struct { const float* ptr[WPAD_N]; int width[WPAD_N]; } wp{};
    PackArgs pk{}; pk.H = H;
    {
        const float* srcs[WPAD_N] = {p->w1 + o.v_other, p->w1 + o.v_mult, p->w1 + o.v_dist, p->w1 + o.z_other, p->w1 + o.a_other, p->w2, p->w3};
        float* cur = (float*)(ws + w.wpad);
        for (int i = 0; i < WPAD_N; ++i) {
            wp.width[i] = wpad_width(d, i); wp.ptr[i] = cur;
            if (!wp.width[i]) continue;
            const int e = pk.n++;
            pk.src[e] = srcs[i]; pk.lds[e] = (i == 5 || i == 6) ? H : din; pk.cols[e] = wpad_cols(d, i); pk.dst[e] = cur; pk.ldd[e] = wp.width[i]; pk.zero_from[e] = pk.cols[e];
            cur += (size_t)H * wp.width[i];
        }
        if (aemb && w.ldgt > d.A) { const int e = pk.n++; pk.src[e] = nullptr; pk.dst[e] = gt; pk.ldd[e] = w.ldgt; pk.zero_from[e] = d.A; pk.cols[e] = 0; }
        if (d.flags & NCX_F_REUSE_GT) pk.n = 0;
    }
    // the pairwise distance rides in the fused forward kernel when that kernel sees whole v rows (no k-split, no slid windows)
    long long main_T = ks(d.dv) * ((d.flags & NCX_F_V_MULT) ? 2 : 1) + ks(w.ldm) + ks(d.dz) + (aemb ? ks(d.A) : ks(d.da));
    // K = 24, whole 32-column tiles, no k-split: the two v segments as one pass with the per-triplet fold (ncx_main.h, MK_VFOLD)
    // (K = 48: on 96-row tiles only -- two triplets per tile -- so only where those fill the chip)
    const bool vfold = main_fwd_dims_ok(d) && (d.flags & NCX_F_V_MULT) && (d.K == 24 || (d.K == 48 && (main_fold_rows(M, H) >= 96 || hook_env("NCX_FOLD4") || hook_env("NCX_FOLD8")))) && d.dv % 32 == 0 && d.dv >= 64 &&
                       main_split(M, H, main_T) == 1 && !hook_env("NCX_NO_VFOLD") &&
                       (long long)d.n_img * d.dv * 4 < (1ll << 32) - 65536 && (long long)H * din * 4 < (1ll << 32) - 65536;      // (the fold's buffer loads: 32-bit byte offsets)
    // Measured at configs[1]: inside the plain chain the distance costs the kernel 12 us and saves k_prep 30; inside the fold's
    // 48 x 64 tiles (a quarter of the MFMA work per vector instruction) it costs 31 us: there k_prep keeps computing it.
    // Round 4, measured again for the one-triplet-per-wave fold forms (96 / 192-row tiles, K = 24: a wave's loader holds v_o beside every v_k quad it loads,
    // the distance is ~14 vector operations per loaded quad and k_prep stops reading the feature rows): the fold loop goes from 5 085 to 5 790 cycles per k-step
    // (5 740 with the arithmetic spread over the sub-steps) -- vector instructions are not free under fp32 MFMAs, each costs the matrix pipe ~8-15 cycles --
    // so the kernel loses the 17 us k_prep gains (0.2938 against 0.2771 ms; step 0.8443 / 0.8483 against 0.8489).  Kept behind the hook NCX_DIST_IN_FOLD.
    const bool dist_in_fold = vfold && d.K == 24 && main_fold_rows_eff(M, H, d.K) >= 96 && !(d.flags & NCX_F_X6) && hook_env("NCX_DIST_IN_FOLD");
    const bool dist_in_main = main_fwd_dims_ok(d) && (d.flags & NCX_F_V_DIST) && (d.flags & NCX_F_V_MULT) && d.dv % 32 == 0 &&
                              main_split(M, H, main_T) == 1 && (!vfold || dist_in_fold) && !(hook_env("NCX_NO_DIST_IN_MAIN"));
    ncx_dims dprep = d;
    if (dist_in_main) dprep.flags |= NCX_F_PRIV_DIST_IN_MAIN;
    if (!do_rest) pk.n = 0;                                   // prelude: no weight is read
    pk.nprep = do_pre ? (int)cdiv(M, 4) : 0;                  // rest: the pack blocks alone
    const unsigned prep_grid = (unsigned)(pk.nprep + (long long)H * pk.n);
    if (prep_grid > 0) {
        if (d.dv <= 2048 && d.A <= 2048)
            hipLaunchKernelGGL(k_prep<true>, dim3(prep_grid), dim3(256), 0, s, dprep, *in, idx_k, idx_o, idx_ob, mx, inv, misc, xc, bf16_cols(d), pk);
        else
            hipLaunchKernelGGL(k_prep<false>, dim3(prep_grid), dim3(256), 0, s, dprep, *in, idx_k, idx_o, idx_ob, mx, inv, misc, xc, bf16_cols(d), pk);
        NCX_HIP_TRY(hipGetLastError());
    }
    if (!do_rest) {
        if (ss) { rc = side_join(ss, s); if (rc) return rc; }
        return NCX_OK;
    }

    // Gt and Sh are two back-to-back split GEMMs on 64 x 64 tiles: their fix-ups (6 us each, mostly launch and ramp) run as one
    FixupArgs fix_gt{}, fix_sh{};
    const bool merge_fix = !ss && !bf16 && aemb && !(d.flags & NCX_F_REUSE_GT) && u[U_GT].plan.cfg == u[U_SH].plan.cfg &&
                           !hook_env("NCX_NO_MERGE_FIX");
    // Gt[H, A] = W1[:, a_other] . E^T   (weights only: evaluation passes reuse it, NCX_F_REUSE_GT)
    if (aemb && bf16 && !(d.flags & NCX_F_REUSE_GT)) {       // bf16 copies of E / W1[:, a_*] (also the backward's operands)
        const Bf16Emb m = bf16_emb_layout(d, ws + w.bf_emb);
        rc = bf16_pack_embedding(d, p->answer_embedding, p->w1, m, s2); if (rc) return rc;
        rc = prof_open(U_GT, s2); if (rc) return rc;
        rc = bf16_gt(d, m, gt, s2); if (rc) return rc;
        rc = prof_close(U_GT, s2); if (rc) return rc;
    } else if (aemb && !(d.flags & NCX_F_REUSE_GT)) {
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = H;
        a.a[0] = x_plain(p->w1 + o.a_other, din, H, d.da);
        a.b[0] = x_plain(p->answer_embedding, d.da, d.A, d.da);
        a.klen[0] = d.da; a.out[0] = gt; a.ldo[0] = w.ldgt; a.n_cols[0] = d.A;
        if (merge_fix) a.defer_fix = &fix_gt;                  // (its partial tiles wait in the side slab for the merged fix-up below)
        rc = run_gemm(U_GT, a, FORM_NT, u[U_GT].plan, merge_fix ? (float*)(ws + w.slab2) : slab_side,
                      merge_fix ? w.slab2_bytes : slab_side_bytes, nullptr, s2);
        if (rc) return rc;
    }
    // Sh[B, H] = b1 + shared segments
    {
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 4; a.M = d.B;
        a.a[0] = ss ? x_gather_strided(in->feats, d.dv, in->img_idx, d.K + 1, d.B, d.dv)      // (idx_ob is written by k_prep, concurrently)
                    : x_gather(in->feats, d.dv, idx_ob, d.B, d.dv);
        a.b[0] = x_plain(p->w1 + o.v_orig, din, H, d.dv); a.klen[0] = d.dv;
        a.a[1] = x_plain(in->q_emb, d.dq, d.B, d.dq);            a.b[1] = x_plain(p->w1 + o.q_emb, din, H, d.dq);  a.klen[1] = d.dq;
        a.a[2] = x_plain(in->z_orig, d.dz, d.B, d.dz);           a.b[2] = x_plain(p->w1 + o.z_orig, din, H, d.dz); a.klen[2] = d.dz;
        a.a[3] = aemb ? x_gather(p->answer_embedding, d.da, in->answer_aids, d.B, d.da)
                      : x_plain(in->a_emb_gt, d.da, d.B, d.da);
        a.b[3] = x_plain(p->w1 + o.a_gt, din, H, d.da); a.klen[3] = d.da;
        a.out[0] = sh; a.ldo[0] = H; a.n_cols[0] = H;
        if (merge_fix) a.defer_fix = &fix_sh;
        rc = run_gemm(U_SH, a, FORM_NT, u[U_SH].plan, slab_side, slab_side_bytes, p->b1, s2);
        if (rc) return rc;
        if (merge_fix) { rc = run_fixup2(fix_gt, fix_sh, u[U_SH].plan.cfg, s); if (rc) return rc; }   // Gt's and Sh's reductions: one launch
        if (ss) { rc = side_join(ss, s); if (rc) return rc; }
    }
    // h1 = drop(relu(Sh[b] + candidate segments))
    if (bf16) {     // plain bf16 product of the packed rows with the packed weights (Wc repacked every step: weights move)
        rc = bf16_pack_wc(d, p->w1, gt, (u16*)(ws + w.wc), s); if (rc) return rc;
        EpiArgs e{};
        e.rowadd = sh; e.ld_rowadd = H; e.rowdiv = d.K;
        set_dropout(e, d, *in, 1, M);
        rc = prof_open(U_MAIN, s); if (rc) return rc;
        rc = bf16_main_forward(d, xc, (const u16*)(ws + w.wc), e, (float*)(ws + w.h[0]), s); if (rc) return rc;
        rc = prof_close(U_MAIN, s); if (rc) return rc;
    } else if (main_fwd_dims_ok(d)) {        // the fused forward kernel (ncx_main.h): weights zero-padded to 32 columns
        MainArgs a{}; a.M = M; a.N = H; a.x6 = (d.flags & NCX_F_X6) && !hook_env("NCX_NO_X6") && !hook_env("NCX_NO_MAIN_X6");
        int n = 0;
        auto seg = [&](int kind, const float* x, long long lda, int klen, const int* i1, const int* i2, const float* lse, int slot, const float* wgt, long long ldb) {
            MainSeg& g = a.seg[n++]; g.kind = kind; g.a = x; g.lda = lda; g.idx = i1; g.idx2 = i2; g.lse = lse; g.klen = klen;
            if (slot >= 0 && wp.width[slot]) { g.b = wp.ptr[slot]; g.ldb = wp.width[slot]; } else { g.b = wgt; g.ldb = ldb; } };
        if (vfold) {
            seg(MK_VFOLD, in->feats, d.dv, d.dv, idx_k, idx_o, nullptr, -1, p->w1 + o.v_other, din);
            a.seg[0].b2 = p->w1 + o.v_mult;
        } else {
            seg(MK_GATHER, in->feats, d.dv, d.dv, idx_k, nullptr, nullptr, 0, p->w1 + o.v_other, din);
            if (d.flags & NCX_F_V_MULT) seg(MK_GATHER_MUL, in->feats, d.dv, d.dv, idx_k, idx_o, nullptr, 1, p->w1 + o.v_mult, din);
        }
        seg(MK_PLAIN, misc, w.ldm, w.ldm, nullptr, nullptr, nullptr, 2, p->w1 + o.v_dist, din);       // (ldm - K - 1 zero columns on both sides)
        seg(MK_PLAIN, in->z_knns, d.dz, d.dz, nullptr, nullptr, nullptr, 3, p->w1 + o.z_other, din);
        if (aemb) seg(MK_SOFTMAX, in->a_knns, d.A, d.A, nullptr, nullptr, mx, -1, gt, w.ldgt);
        else      seg(MK_PLAIN, in->a_knns, d.da, d.da, nullptr, nullptr, nullptr, 4, p->w1 + o.a_other, din);
        a.nseg = n;
        a.out = (float*)(ws + w.h[0]); a.ldo = H;
        a.epi.rowadd = sh; a.epi.ld_rowadd = H; a.epi.rowdiv = d.K;
        set_dropout(a.epi, d, *in, 1, M);
        if (dist_in_main) { a.dist_out = misc; a.ld_dist = w.ldm; }
        {
            long long T = 0; for (int i = 0; i < n; ++i) T += ks(a.seg[i].klen);
            a.split = main_split(M, H, T); a.slab = (float*)(ws + w.mslab);
            if (a.split > 1 && (size_t)a.split * M * H * 4 > w.mslab_bytes) return NCX_E_WORKSPACE;
        }
        if (a.split <= 1) a.stamps = stamps_for_current_device((long long)(((M + 47) / 48 + 7) / 8 * 8) * ((H + 63) / 64) * 16);   // (bound: the smallest tile = the most workgroups; null unless armed on THIS device)
        rc = prof_open(U_MAIN, s); if (rc) return rc;
        rc = main_forward(a, s); if (rc) return rc;
        rc = prof_close(U_MAIN, s); if (rc) return rc;
    } else {                                  // widths that are not multiples of 4: the generic segmented engine
        GemmArgs a{}; a.mode = MODE_CHAIN; a.M = M;
        int n = 0;
        a.a[n] = x_gather(in->feats, d.dv, idx_k, M, d.dv); a.b[n] = x_plain(p->w1 + o.v_other, din, H, d.dv); a.klen[n] = d.dv; ++n;
        if (d.flags & NCX_F_V_MULT) {
            a.a[n] = x_gather_mul(in->feats, d.dv, idx_k, idx_o, M, d.dv); a.b[n] = x_plain(p->w1 + o.v_mult, din, H, d.dv); a.klen[n] = d.dv; ++n;
        }
        a.a[n] = x_plain(misc, w.ldm, M, d.K + 1); a.b[n] = x_plain(p->w1 + o.v_dist, din, H, d.K + 1); a.klen[n] = d.K + 1; ++n;
        a.a[n] = x_plain(in->z_knns, d.dz, M, d.dz); a.b[n] = x_plain(p->w1 + o.z_other, din, H, d.dz); a.klen[n] = d.dz; ++n;
        if (aemb) { a.a[n] = x_softmax(in->a_knns, d.A, mx, inv, M, d.A); a.b[n] = x_plain(gt, w.ldgt, H, d.A); a.klen[n] = d.A; ++n; }
        else      { a.a[n] = x_plain(in->a_knns, d.da, M, d.da); a.b[n] = x_plain(p->w1 + o.a_other, din, H, d.da); a.klen[n] = d.da; ++n; }
        a.nseg = n;
        a.out[0] = (float*)(ws + w.h[0]); a.ldo[0] = H; a.n_cols[0] = H;
        a.epi.rowadd = sh; a.epi.ld_rowadd = H; a.epi.rowdiv = d.K;
        set_dropout(a.epi, d, *in, 1, M);
        rc = run_gemm(U_MAIN, a, FORM_NT, u[U_MAIN].plan, slab, w.slab_bytes, nullptr, s);
        if (rc) return rc;
    }
    for (int l = 2; l <= d.L; ++l) {
        const float* wl = l == 2 ? p->w2 : p->w3;
        const float* bl = l == 2 ? p->b2 : p->b3;
        if (hidden_fwd_dims_ok(d) && !bf16) {
            MainArgs a{}; a.M = M; a.N = H; a.nseg = 1;
            MainSeg& g = a.seg[0]; g.kind = MK_PLAIN; g.a = (const float*)(ws + w.h[l - 2]); g.lda = H; g.klen = H;
            const int slot = 3 + l;                                       // 5: linear_2, 6: linear_3
            if (wp.width[slot]) { g.b = wp.ptr[slot]; g.ldb = wp.width[slot]; } else { g.b = wl; g.ldb = H; }
            a.out = (float*)(ws + w.h[l - 1]); a.ldo = H;
            a.epi.bias = bl;
            set_dropout(a.epi, d, *in, l, M);
            a.split = main_split(M, H, ks(H)); a.slab = (float*)(ws + w.mslab);
            if (a.split > 1 && (size_t)a.split * M * H * 4 > w.mslab_bytes) return NCX_E_WORKSPACE;
            rc = prof_open(U_FWD_L, s); if (rc) return rc;
            rc = main_forward(a, s); if (rc) return rc;
            rc = prof_close(U_FWD_L, s); if (rc) return rc;
        } else {
            GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = M;
            a.a[0] = x_plain((const float*)(ws + w.h[l - 2]), H, M, H); a.b[0] = x_plain(wl, H, H, H); a.klen[0] = H;
            a.out[0] = (float*)(ws + w.h[l - 1]); a.ldo[0] = H; a.n_cols[0] = H;
            a.epi.bias = bl;
            set_dropout(a.epi, d, *in, l, M);
            rc = run_gemm(U_FWD_L, a, FORM_NT, u[U_FWD_L].plan, slab, w.slab_bytes, nullptr, s);
            if (rc) return rc;
        }
    }
    if (d.flags & NCX_F_FUSED_TAIL) return NCX_OK;       // the out layer runs in ncx_train_tail
    hipLaunchKernelGGL(k_scores, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, s, (const float*)(ws + w.h[d.L - 1]),
                       p->w_out, p->b_out, scores, M, H);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

int ncx_loss_rank(const float* scores, const int32_t* gt, int32_t B, int32_t K, float scale, float* loss_rows,
                  float* loss, float* dscores, int32_t* rank, int32_t* hits, void* stream_) {
    if (!scores || !gt) return NCX_E_NULL;
    if (B < 1 || K < 1 || K > 64) return NCX_E_DIMS;
    if ((loss && !loss_rows) || (hits && !rank)) return NCX_E_NULL;
    hipStream_t s = (hipStream_t)stream_;
    if (scale <= 0.f) scale = 1.f / (float)B;
    hipLaunchKernelGGL(k_loss_rank, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, s, scores, gt, B, K, scale, loss_rows, dscores, rank);
    NCX_HIP_TRY(hipGetLastError());
    if (loss || hits) {
        hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, s, (const float*)loss_rows, (const int*)rank, B, loss, hits);
        NCX_HIP_TRY(hipGetLastError());
    }
    return NCX_OK;
}

}  // extern "C"
// The answer-embedding gradient runs in NT form (operands dGt^T | dGgt^T, csrc/ncx_main.h) for the fp32 path with the a_emb
// segment on.  ONE predicate: ncx_train_tail and backward_impl fill the block in that form, ncx_ws_region names it.
static inline bool emb_nt_form(const ncx_dims& d) {
    return (d.flags & NCX_F_A_EMB) && !(d.flags & NCX_F_BF16) && !ncx::hook_env("NCX_NO_EMB_NT");
}
extern "C" {
int ncx_train_tail(const ncx_dims* dp, const ncx_params* p, void* workspace, size_t workspace_bytes, const int32_t* gt,
                   float* scores, float* loss_rows, float* loss, float* dscores, int32_t* rank, int32_t* hits,
                   const ncx_grads* g, void* stream_) {
    int rc = check_dims(dp);
    if (rc != NCX_OK) return rc;
    if (!p || !workspace || !gt || !scores || !loss_rows || !rank || !g) return NCX_E_NULL;
    if (!p->w_out || !p->b_out || !g->w_out || !g->b_out || !g->b1) return NCX_E_NULL;
    const ncx_dims& d = *dp;
    if (!(d.flags & NCX_F_FUSED_TAIL)) return NCX_E_FLAGS;
    if (d.K > 32 || d.H > 256) return NCX_E_DIMS;            // the K rows of a triplet live in registers, one lane per 4 columns
    const WsLayout w = ws_layout(d);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return NCX_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const int H = d.H;
    const bool aemb = d.flags & NCX_F_A_EMB;
    const bool emb_nt = emb_nt_form(d);
    const int Hp4 = pad_to(H, 4);
    float* dagtT = (float*)(ws + w.dgtT) + (size_t)d.A * Hp4;
    float* dagt = (float*)(ws + w.dagt);
    float* partial = (float*)(ws + w.partial);
    float* part_w = partial;
    float* part_b1 = partial + (size_t)NCX_PRELUDE_WAVES * H;
    float* part_b = partial + (size_t)NCX_PRELUDE_WAVES * H * 2;
    const float dscale = (d.training && d.drop_p > 0.f) ? 1.f / (1.f - d.drop_p) : 1.f;
    const float scale = d.loss_scale > 0.f ? d.loss_scale : 1.f / (float)d.B;
    const int nblk = (int)cdiv(d.B < NCX_PRELUDE_WAVES ? d.B : NCX_PRELUDE_WAVES, 4);
    const bool fuse_l1 = d.L == 1;
    const float* hL = (const float*)(ws + w.h[d.L - 1]);
    float* dpre = (float*)(ws + w.dpre[0]);
    float* dsh = fuse_l1 ? (float*)(ws + w.dsh) : (float*)nullptr;
    float* zb = emb_nt ? dagtT : aemb ? dagt : (float*)nullptr;
    const long long zn = emb_nt ? (long long)d.A * Hp4 : (long long)H * d.A;
#define NCX_TAIL_LAUNCH(KB, FULL) hipLaunchKernelGGL((k_train_tail<KB, FULL>), dim3(nblk * 4), dim3(64), 0, s, hL, p->w_out, p->b_out, gt, d.B, d.K, H, scale, dscale, \
                                               scores, loss_rows, dscores, rank, dpre, dsh, part_w, part_b1, part_b, zb, zn, zchunk)
    const int zchunk = (int)((cdiv(zn, (long long)nblk * 4) + 3) / 4 * 4);
    if (H == 256 && d.K == 24) NCX_TAIL_LAUNCH(24, true);          // (the configuration of every options/cx/*.yaml with dim_h 256)
    else if (d.K <= 8) NCX_TAIL_LAUNCH(8, false); else if (d.K <= 16) NCX_TAIL_LAUNCH(16, false); else if (d.K <= 24) NCX_TAIL_LAUNCH(24, false); else NCX_TAIL_LAUNCH(32, false);
#undef NCX_TAIL_LAUNCH
    NCX_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_bwd_prelude_finish, dim3((unsigned)cdiv(H, 8), 3), dim3(256), 0, s, (const float*)part_w, (const float*)part_b1,
                       (const float*)part_b, nblk * 4, H, g->w_out, fuse_l1 ? g->b1 : (float*)nullptr, g->b_out,
                       (const float*)loss_rows, (const int*)rank, d.B, loss, hits);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

static int backward_impl(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                         size_t workspace_bytes, const float* dscores, const ncx_grads* g, void* stream_, int phase);

int ncx_backward(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                 size_t workspace_bytes, const float* dscores, const ncx_grads* g, void* stream_) {
    return backward_impl(dp, in, p, workspace, workspace_bytes, dscores, g, stream_, 0);
}

int ncx_backward_phase(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                       size_t workspace_bytes, const float* dscores, const ncx_grads* g, int32_t phase, void* stream_) {
    if (phase < 0 || phase > 5) return NCX_E_FLAGS;
    return backward_impl(dp, in, p, workspace, workspace_bytes, dscores, g, stream_, phase);
}

// phase 0: everything.  phase 1: out / hidden layers / b1 and the answer_embedding gradient (complete when it
// returns);  phase 2: linear_1.weight.  1 then 2 == 0 bit for bit (same kernels, the dGt problem launched alone).
// phase 3: everything except the answer_embedding GEMM (leaves dGt | dGgt in the workspace, ncx_ws_region);
// phase 5: phase 1 without the answer_embedding GEMM (leaves dGt | dGgt like phase 3): 5, 2, 4 == 0 bit for bit, and the
// region can be on the wire while phase 2 (the bulk of the backward) runs.
// phase 4: answer_embedding gradient from dGt | dGgt.  3 then 4 == 0 bit for bit; under data parallelism the
// 2 x [H, A] block is summed over ranks between the two, so the [A, da] embedding gradient never crosses the wire.
static int backward_impl(const ncx_dims* dp, const ncx_inputs* in, const ncx_params* p, void* workspace,
                         size_t workspace_bytes, const float* dscores, const ncx_grads* g, void* stream_, int phase) {
    int rc = check_dims(dp);
    if (rc != NCX_OK) return rc;
    if (!in || !p || !workspace || !g || (!dscores && !(dp->flags & NCX_F_FUSED_TAIL))) return NCX_E_NULL;
    const ncx_dims& d = *dp;
    const bool aemb = d.flags & NCX_F_A_EMB;
    if (!g->answer_embedding || !g->w1 || !g->b1 || !g->w_out || !g->b_out) return NCX_E_NULL;
    if (d.L >= 2 && (!g->w2 || !g->b2)) return NCX_E_NULL;
    if (d.L >= 3 && (!g->w3 || !g->b3)) return NCX_E_NULL;
    const WsLayout w = ws_layout(d);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return NCX_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const int M = d.B * d.K, H = d.H;
    const SegOffsets o = seg_offsets(d);
    const long long din = o.din;
    const int* idx_k = (const int*)(ws + w.idx_k); const int* idx_o = (const int*)(ws + w.idx_o);
    const int* idx_ob = (const int*)(ws + w.idx_ob);
    const float* mx = (const float*)(ws + w.mx); const float* inv = (const float*)(ws + w.inv);
    const float* misc = (const float*)(ws + w.misc); const float* gt = (const float*)(ws + w.gt);
    float* dsh = (float*)(ws + w.dsh); float* dgt = (float*)(ws + w.dgt); float* dagt = (float*)(ws + w.dagt);
    const bool emb_nt = emb_nt_form(d);   // answer-embedding gradient in NT form on the fused forward kernel
    const int Hp4 = pad_to(H, 4), Hp32 = pad_to(H, 32);
    float* dgtT = (float*)(ws + w.dgtT); float* dagtT = dgtT + (size_t)d.A * Hp4;
    float* w1akT = (float*)(ws + w.w1aT); float* w1agtT = w1akT + (size_t)d.da * Hp32;
    float* partial = (float*)(ws + w.partial); float* slab = (float*)(ws + w.slab);
    (void)gt;
    GemmUse u[U_COUNT];
    list_uses(d, u);
    const float dscale = (d.training && d.drop_p > 0.f) ? 1.f / (1.f - d.drop_p) : 1.f;
    const int nch = M < NCX_COLSUM_CHUNKS * 8 ? (int)cdiv(M, 8) : NCX_COLSUM_CHUNKS;

    auto colsum = [&](const float* x, const float* wgt, int rows, int cols, float* out) -> int {
        const int ch = rows < NCX_COLSUM_CHUNKS * 8 ? (int)cdiv(rows, 8) : NCX_COLSUM_CHUNKS;
        hipLaunchKernelGGL(k_colsum_partial, dim3((unsigned)cdiv(cols, 256), ch), dim3(256), 0, s, x, wgt, rows, cols, partial);
        hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)cdiv(cols, 8)), dim3(256), 0, s, (const float*)partial, ch, cols, out);
        return (int)hipGetLastError();
    };
    (void)nch;

    // ---- out layer + last hidden layer's activation ------------------------------------------------
    const float* hL = (const float*)(ws + w.h[d.L - 1]);
    float* dpre = (float*)(ws + w.dpre[0]);
    const bool only_de = phase == 4;                    // phase 4: just the dE GEMM
    const bool skip_de = phase == 3 || phase == 5;
    const bool do1 = phase != 2 && !only_de, do2 = phase != 1 && phase != 5 && !only_de;
    if (!do1) {                                       // phase 2: dpre_1 lives where phase 1 left it
        dpre = (float*)(ws + w.dpre[(d.L - 1) & 1]);
    } else if (d.flags & NCX_F_FUSED_TAIL) {
        // ncx_train_tail has produced dpre_L, dSh and the out-layer / linear_1.bias gradients
    } else {
        // one pass: dpre_L, d out.weight / d out.bias partials, and for L == 1 also dSh + d linear_1.bias partials
        const int nblk = (int)cdiv(d.B < NCX_PRELUDE_WAVES ? d.B : NCX_PRELUDE_WAVES, 4);      // one wave per run of triplets
        float* part_w = partial;
        float* part_b1 = partial + (size_t)NCX_PRELUDE_WAVES * H;
        float* part_b = partial + (size_t)NCX_PRELUDE_WAVES * H * 2;
        const bool fuse_l1 = d.L == 1;
        hipLaunchKernelGGL(k_bwd_prelude, dim3(nblk), dim3(256), 0, s, dscores, p->w_out, hL, dpre, fuse_l1 ? dsh : (float*)nullptr,
                           d.B, d.K, H, dscale, part_w, part_b1, part_b,
                           emb_nt ? dagtT : aemb ? dagt : (float*)nullptr, emb_nt ? (long long)d.A * Hp4 : (long long)H * d.A);
        NCX_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_bwd_prelude_finish, dim3((unsigned)cdiv(H, 8), fuse_l1 ? 2 : 1), dim3(256), 0, s, (const float*)part_w,
                           (const float*)part_b1, (const float*)part_b, nblk * 4, H, g->w_out, fuse_l1 ? g->b1 : (float*)nullptr, g->b_out);
        NCX_HIP_TRY(hipGetLastError());
    }
    // ---- hidden layers L..2 ---------------------------------------------------------------------------
    int cur = 0;
    for (int l = d.L; l >= 2 && do1; --l) {
        const float* wl = l == 2 ? p->w2 : p->w3;
        float* gw = l == 2 ? g->w2 : g->w3;
        float* gb = l == 2 ? g->b2 : g->b3;
        const float* hprev = (const float*)(ws + w.h[l - 2]);
        rc = colsum(dpre, nullptr, M, H, gb); if (rc) return rc;
        {   // dW_l[n][k] = sum_r dpre[r][n] h_{l-1}[r][k]
            GemmArgs a{}; a.mode = MODE_GROUP; a.nseg = 1; a.M = H;
            a.a[0] = x_plain(dpre, H, M, H); a.b[0] = x_plain(hprev, H, M, H); a.klen[0] = M;
            a.out[0] = gw; a.ldo[0] = H; a.n_cols[0] = H;
            rc = run_gemm(U_DWL, a, FORM_TN, u[U_DWL].plan, slab, w.slab_bytes, nullptr, s); if (rc) return rc;
        }
        {   // dpre_{l-1} = (dpre_l . W_l) * gate(h_{l-1})
            float* dnext = (float*)(ws + w.dpre[cur ^ 1]);
            GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = M;
            a.a[0] = x_plain(dpre, H, M, H); a.b[0] = x_plain(wl, H, H, H); a.klen[0] = H;
            a.out[0] = dnext; a.ldo[0] = H; a.n_cols[0] = H;
            a.epi.gate = hprev; a.epi.ld_gate = H; a.epi.gate_scale = dscale;
            rc = run_gemm(U_DXL, a, FORM_NN, u[U_DXL].plan, slab, w.slab_bytes, nullptr, s); if (rc) return rc;
            dpre = dnext; cur ^= 1;
        }
    }
    // ---- layer 1 ---------------------------------------------------------------------------------------
    if (do1 && d.L >= 2) {                              // (L == 1: already produced by k_bwd_prelude)
        hipLaunchKernelGGL(k_rowgroup_sum, dim3((unsigned)cdiv((long long)d.B * H, 256)), dim3(256), 0, s, (const float*)dpre, d.B, d.K, H, dsh);
        NCX_HIP_TRY(hipGetLastError());
        rc = colsum(dsh, nullptr, d.B, H, g->b1); if (rc) return rc;
    }
    bool km_deferred = false, km_reduced = false;
    auto run_km = [&](bool finish = true) -> int {
        int r = prof_open(U_DW1C, s); if (r) return r;
        r = dw_km(d, dpre, in->feats, idx_k, idx_o, (float*)(ws + w.km_slab), g->w1 + o.v_other, g->w1 + o.v_mult, din, s, finish);
        if (r) return r;
        return prof_close(U_DW1C, s);
    };
    {   // dW1 = [dpre^T . candidate segments (+ dGt) | dSh^T . shared segments]: ONE grouped launch, per-problem
        // reduction extent (M rows of dpre vs B rows of dSh) and k-split
        GemmArgs a{}; a.mode = MODE_GROUP; a.M = H;
        int n = 0;
        auto add_c = [&](const XDesc& x, float* out, long long ldo) {
            a.a[n] = x_plain(dpre, H, M, H); a.b[n] = x; a.klen[n] = M; a.out[n] = out; a.ldo[n] = ldo; a.n_cols[n] = x.cols;
            a.split[n] = dw1c_seg_split(x.cols, u[U_DW1C].plan.split, u[U_DW1C].ksteps); ++n; };
        auto add_s = [&](const XDesc& x, float* out) {
            a.a[n] = x_plain(dsh, H, d.B, H); a.b[n] = x; a.klen[n] = d.B; a.out[n] = out; a.ldo[n] = din; a.n_cols[n] = x.cols;
            a.split[n] = u[U_DW1S].plan.split; ++n; };
        // the dGt problem is what the answer_embedding gradient waits for: phase 1 launches it alone
        const bool bf16 = d.flags & NCX_F_BF16;
        const bool want_dgt = aemb && do1 && !bf16, want_rest = do2;
        if (bf16 && do1) {   // all candidate columns + dGt: dpre^T . Xc on the bf16 MFMA path (complete in phases 0, 1, 3)
            rc = prof_open(U_DW1C, s); if (rc) return rc;
            rc = bf16_dw1c(d, dpre, (u16*)(ws + w.dpre_bf), (const u16*)(ws + w.xc), (float*)(ws + w.bf_slab), g->w1, dgt, s);
            if (rc) return rc;
            rc = prof_close(U_DW1C, s); if (rc) return rc;
        }
        const bool km = dw_km_supported(d) && !bf16;
        const bool tn8 = dw_tn8_supported(d);               // dGt + every column block that is not the per-triplet fold's: ONE balanced launch (ncx_dwtn.hip)
        // With the side stream the per-triplet fold kernel (a full round of long workgroups) is launched AFTER the grouped
        // launch, next to the answer-embedding chain (dW1ak, dE: short latency-bound workgroups) that waits for dGt.
        km_deferred = want_rest && km && aemb && do1 && do2 && side_stream() != nullptr && side_stream()->mode != 2;
        // the sums over its k-chunk partials ride in one launch with the split fix-up of the grouped GEMM below
        const bool km_merge = want_rest && km && !km_deferred && !hook_env("NCX_NO_MERGE_FIX");
        if (want_rest && km && !km_deferred) {      // v_other and v_mult columns in one MFMA pass (per-triplet fold)
            rc = run_km(!km_merge); if (rc) return rc;
        }
        if (want_rest && !bf16 && !km) {
            add_c(x_gather(in->feats, d.dv, idx_k, M, d.dv), g->w1 + o.v_other, din);
            if (d.flags & NCX_F_V_MULT) add_c(x_gather_mul(in->feats, d.dv, idx_k, idx_o, M, d.dv), g->w1 + o.v_mult, din);
        }
        if (tn8 && (want_dgt || want_rest)) {
            // The problem list is the same in every phase (the chunking of the dGt part and of the rest do not depend on each other,
            // and the slab slots are numbered over the whole list): phases 5 | 2 launch its two parts separately, bit-identically.
            Tn8Prob tp[TN8_MAX_PROB]; int np = 0, n_al = 0;
            auto prob = [&](const float* A_, int rows, const float* X, long long ldx, const float* lse, int gsel, int N, float* out, long long ldo, int n_valid) {
                tp[np] = Tn8Prob{};
                Tn8Prob& q = tp[np++]; q.A = A_; q.rows = rows; q.X = X; q.ldx = ldx; q.lse = lse; q.gsel = gsel; q.N = N; q.out = out; q.ldo = ldo; q.n_valid = n_valid; };
            if (aemb) {
                prob(dpre, M, in->a_knns, d.A, mx, 0, d.A, dgt, d.A, d.A); n_al = 1;
                if (emb_nt) {       // the reduction also writes dGt^T (what the dE product and the DP exchange read) and a private copy with
                    // zero rows up to a multiple of 32 (the A operand of the dW1[:, a_other] = dGt . E launch below)
                    tp[0].outT = dgtT; tp[0].outT2 = (float*)(ws + w.dgtT2); tp[0].ldT = Hp4; tp[0].padT2 = pad_to(d.A, 32) - d.A;
                }
            }
            else prob(dpre, M, in->a_knns, d.da, nullptr, 0, d.da, g->w1 + o.a_other, din, d.da);
            prob(dpre, M, in->z_knns, d.dz, nullptr, 0, d.dz, g->w1 + o.z_other, din, d.dz);
            prob(dpre, M, misc, w.ldm, nullptr, 0, w.ldm, g->w1 + o.v_dist, din, d.K + 1);
            prob(dsh, d.B, in->feats, d.dv, nullptr, 1, d.dv, g->w1 + o.v_orig, din, d.dv);
            prob(dsh, d.B, in->q_emb, d.dq, nullptr, 0, d.dq, g->w1 + o.q_emb, din, d.dq);
            prob(dsh, d.B, in->z_orig, d.dz, nullptr, 0, d.dz, g->w1 + o.z_orig, din, d.dz);
            if (aemb) prob(dsh, d.B, p->answer_embedding, d.da, nullptr, 2, d.da, g->w1 + o.a_gt, din, d.da);
            else      prob(dsh, d.B, in->a_emb_gt, d.da, nullptr, 0, d.da, g->w1 + o.a_gt, din, d.da);
            // ... and ONE launch sums its partial tiles and the fold kernel's k-chunks (when that kernel ran just above)
            const bool with_km = want_rest && km && !km_deferred && km_merge;
            Tn8ReduceArgs red{};
            rc = prof_open(U_DW1C, s); if (rc) return rc;
            rc = dw_tn8_products(d, tp, np, n_al, want_dgt, want_rest, idx_ob, aemb ? in->answer_aids : nullptr, slab, w.slab_bytes, &red, s); if (rc) return rc;
            bool kvec = true;
            KmReduceArgs kr{};
            if (with_km) { kr = dw_km_reduce_args(d, (const float*)(ws + w.km_slab), g->w1 + o.v_other, g->w1 + o.v_mult, din, &kvec); km_reduced = true; }
            rc = dw_reduce_km_tn8(with_km ? &kr : nullptr, kvec, &red, s); if (rc) return rc;
            rc = prof_close(U_DW1C, s); if (rc) return rc;
        }
        if (want_rest && !bf16 && !tn8) {
            if (!aemb) add_c(x_plain(in->a_knns, d.da, M, d.da), g->w1 + o.a_other, din);
        }
        if (want_dgt && !tn8) add_c(x_softmax(in->a_knns, d.A, mx, inv, M, d.A), dgt, d.A);
        if (want_rest && !bf16 && !tn8) {      // the narrow problems last among the candidate ones (shorter k-chunks: see dw1c_seg_split)
            add_c(x_plain(in->z_knns, d.dz, M, d.dz), g->w1 + o.z_other, din);
            add_c(x_plain(misc, w.ldm, M, d.K + 1), g->w1 + o.v_dist, din);
        }
        if (want_rest && bf16 && dw_tn8_shapes_ok(d)) {      // bf16 variant: the fp32 shared segments' weight gradient on the balanced TN kernel
            Tn8Prob tp[4]; int np = 0;
            auto prob = [&](const float* X, long long ldx, int gsel, int N, float* out) {
                tp[np] = Tn8Prob{};
                Tn8Prob& q = tp[np++]; q.A = dsh; q.rows = d.B; q.X = X; q.ldx = ldx; q.gsel = gsel; q.N = N; q.out = out; q.ldo = din; q.n_valid = N; };
            prob(in->feats, d.dv, 1, d.dv, g->w1 + o.v_orig);
            prob(in->q_emb, d.dq, 0, d.dq, g->w1 + o.q_emb);
            prob(in->z_orig, d.dz, 0, d.dz, g->w1 + o.z_orig);
            if (aemb) prob(p->answer_embedding, d.da, 2, d.da, g->w1 + o.a_gt);
            else      prob(in->a_emb_gt, d.da, 0, d.da, g->w1 + o.a_gt);
            rc = dw_tn8(d, tp, np, 0, false, true, idx_ob, aemb ? in->answer_aids : nullptr, slab, w.slab_bytes, s); if (rc) return rc;
        } else
        if (want_rest && !tn8) {
            add_s(x_gather(in->feats, d.dv, idx_ob, d.B, d.dv), g->w1 + o.v_orig);
            add_s(x_plain(in->q_emb, d.dq, d.B, d.dq), g->w1 + o.q_emb);
            add_s(x_plain(in->z_orig, d.dz, d.B, d.dz), g->w1 + o.z_orig);
            add_s(aemb ? x_gather(p->answer_embedding, d.da, in->answer_aids, d.B, d.da) : x_plain(in->a_emb_gt, d.da, d.B, d.da),
                  g->w1 + o.a_gt);
        }
        a.nseg = n;
        FixupArgs fix_tn{};
        if (km_merge) a.defer_fix = &fix_tn;
        if (n > 0) { rc = run_gemm(U_DW1C, a, FORM_TN, u[U_DW1C].plan, slab, w.slab_bytes, nullptr, s); if (rc) return rc; }
        if (km_merge && !km_reduced) {
            rc = prof_open(U_DW1C, s); if (rc) return rc;
            rc = dw_km_finish(d, (const float*)(ws + w.km_slab), g->w1 + o.v_other, g->w1 + o.v_mult, din, &fix_tn, u[U_DW1C].plan.cfg, s);
            if (rc) return rc;
            rc = prof_close(U_DW1C, s); if (rc) return rc;
        }
        if (!(d.flags & NCX_F_V_MULT) && do2) {
            hipLaunchKernelGGL(k_zero_cols, dim3((unsigned)cdiv((long long)H * d.dv, 256)), dim3(256), 0, s, g->w1 + o.v_mult, H, din, d.dv);
            NCX_HIP_TRY(hipGetLastError());
        }
    }
    if (aemb) {
        // The consumers of dGt (dW1[:, a_other] = dGt . E, then dE) form a chain of short launches: with the side stream they
        // run there (own slab) while the caller's stream runs the per-triplet fold kernel
        SideStream* ss = km_deferred ? side_stream() : nullptr;
        hipStream_t se = ss ? ss->s : s;
        FixupArgs fix_ak{};
        Tn8ReduceArgs red_ak{};
        const bool tn8_ak = dw_tn8_supported(d) && emb_nt && !ss;
        if (ss) { rc = side_fork(ss, s); if (rc) return rc; }
        const bool bf16e = d.flags & NCX_F_BF16;
        const Bf16Emb bm = bf16e ? bf16_emb_layout(d, ws + w.bf_emb) : Bf16Emb{};
        if (do2 && bf16e) {
            rc = prof_open(U_DW1AK, se); if (rc) return rc;
            rc = bf16_dw1ak(d, bm, dgt, g->w1, se); if (rc) return rc;
            rc = prof_close(U_DW1AK, se); if (rc) return rc;
        } else if (do2 && tn8_ak) {   // dW1[:, a_other][n][j] = sum_a dGt^T[a][n] E[a][j]: a row-reduction over the A answers on the 8-wave TN kernel
            // (round 4; NN form on the generic engine: 39 us + fix-up for 2.46 GF).  Its A operand is the PRIVATE copy of dGt^T: under data
            // parallelism the other copy is being summed over ranks in place while this runs.
            Tn8Prob q{};
            q.A = (const float*)(ws + w.dgtT2); q.rows = pad_to(d.A, 32); q.rows_valid = d.A; q.X = p->answer_embedding; q.ldx = d.da; q.N = d.da;
            q.out = g->w1 + o.a_other; q.ldo = din; q.n_valid = d.da;
            rc = prof_open(U_DW1AK, se); if (rc) return rc;
            rc = dw_tn8_products(d, &q, 1, 0, false, true, nullptr, nullptr, slab, w.slab_bytes, &red_ak, se); if (rc) return rc;
            if (!(do1 && emb_nt)) { rc = dw_reduce_km_tn8(nullptr, true, &red_ak, se); if (rc) return rc; red_ak.n_tiles_total = 0; }   // (else: rides in k_emb_prep's launch)
            rc = prof_close(U_DW1AK, se); if (rc) return rc;
        } else if (do2) {   // dW1[:, a_other][n][j] = sum_a dGt[n][a] E[a][j]
            GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = H;
            a.a[0] = x_plain(dgt, d.A, H, d.A); a.b[0] = x_plain(p->answer_embedding, d.da, d.A, d.da); a.klen[0] = d.A;
            a.out[0] = g->w1 + o.a_other; a.ldo[0] = din; a.n_cols[0] = d.da;
            if (do1 && emb_nt && u[U_DW1AK].plan.cfg == CFG_64x64 && !hook_env("NCX_NO_MERGE_FIX")) a.defer_fix = &fix_ak;
            rc = run_gemm(U_DW1AK, a, FORM_NN, u[U_DW1AK].plan, ss ? (float*)(ws + w.slab2) : slab,
                          ss ? w.slab2_bytes : w.slab_bytes, nullptr, se); if (rc) return rc;
        }
        if (do1 && emb_nt) {   // dGgt^T, dGt^T and the transposed weight slices for the NT embedding gradient
            EmbPrepArgs ea{};
            ea.dsh = dsh; ea.aid = in->answer_aids; ea.dggtT = dagtT; ea.B = d.B; ea.H = H; ea.A = d.A; ea.Hp4 = Hp4;
            const float* srcs[3] = {dgt, p->w1 + o.a_other, p->w1 + o.a_gt};
            float* dsts[3] = {dgtT, w1akT, w1agtT};
            const long long ldss[3] = {d.A, din, din};
            const int colss[3] = {tn8_ak ? 0 : d.A, d.da, d.da}, ldds[3] = {Hp4, Hp32, Hp32};      // (tn8: dGt^T came out of the reduction)
            int tiles = 0;
            for (int e = 0; e < 3; ++e) {
                ea.src[e] = srcs[e]; ea.dst[e] = dsts[e]; ea.lds_[e] = ldss[e]; ea.cols[e] = colss[e]; ea.ldd[e] = ldds[e]; ea.dcols[e] = ldds[e];
                ea.tile0[e] = tiles; tiles += ((ldds[e] + 31) / 32) * ((colss[e] + 31) / 32);
            }
            ea.tile0[3] = tiles;
            // (the dW1[:, a_other] GEMM above left its split fix-up to this launch)
            hipLaunchKernelGGL(k_emb_prep, dim3(d.B + tiles + (fix_ak.valid ? fix_ak.grid_x * 4 : 0) + red_ak.n_tiles_total * 8), dim3(256), 0, se, ea, fix_ak, red_ak);
            fix_ak.valid = 0; red_ak.n_tiles_total = 0;
            NCX_HIP_TRY(hipGetLastError());
        } else if (do1) {   // dGgt = one-hot(aid)^T dSh   (dGgt was cleared by k_bwd_prelude)
            hipLaunchKernelGGL(k_scatter_dsh_by_answer, dim3(d.B), dim3(256), 0, se, (const float*)dsh, in->answer_aids, d.B, H, d.A, dagt);
            NCX_HIP_TRY(hipGetLastError());
        }
        if (fix_ak.valid) { rc = run_fixup2(fix_ak, FixupArgs{}, CFG_64x64, se); if (rc) return rc; }      // (not picked up above)
        if (((do1 && !skip_de) || only_de) && bf16e) {
            rc = prof_open(U_DE, se); if (rc) return rc;
            rc = bf16_de(d, bm, dgt, g->answer_embedding, se); if (rc) return rc;
            rc = prof_close(U_DE, se); if (rc) return rc;
        } else if (((do1 && !skip_de) || only_de) && emb_nt) {   // dE = dGt^T . (W1ak^T)^T + dGgt^T . (W1agt^T)^T on the fused forward kernel
            MainArgs a{}; a.M = d.A; a.N = d.da; a.nseg = 2;
            a.seg[0].kind = MK_PLAIN; a.seg[0].a = dgtT;  a.seg[0].lda = Hp4; a.seg[0].klen = Hp4; a.seg[0].b = w1akT;  a.seg[0].ldb = Hp32;
            a.seg[1].kind = MK_PLAIN; a.seg[1].a = dagtT; a.seg[1].lda = Hp4; a.seg[1].klen = Hp4; a.seg[1].b = w1agtT; a.seg[1].ldb = Hp32;
            a.out = g->answer_embedding; a.ldo = d.da;
            a.split = 1;
            rc = prof_open(U_DE, se); if (rc) return rc;
            rc = main_forward(a, se); if (rc) return rc;
            rc = prof_close(U_DE, se); if (rc) return rc;
        } else if ((do1 && !skip_de) || only_de) {   // dE[a][j] = sum_n dGt[n][a] W1ak[n][j] + sum_n dGgt[n][a] W1agt[n][j]
            GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 2; a.M = d.A;
            a.a[0] = x_plain(dgt, d.A, H, d.A);  a.b[0] = x_plain(p->w1 + o.a_other, din, H, d.da); a.klen[0] = H;
            a.a[1] = x_plain(dagt, d.A, H, d.A); a.b[1] = x_plain(p->w1 + o.a_gt, din, H, d.da);    a.klen[1] = H;
            a.out[0] = g->answer_embedding; a.ldo[0] = d.da; a.n_cols[0] = d.da;
            rc = run_gemm(U_DE, a, FORM_TN, u[U_DE].plan, ss ? (float*)(ws + w.slab2) : slab, ss ? w.slab2_bytes : w.slab_bytes, nullptr, se); if (rc) return rc;
        }
        if (km_deferred) { rc = run_km(); if (rc) return rc; }
        if (ss) { rc = side_join(ss, s); if (rc) return rc; }
    } else if (do1) {
        NCX_HIP_TRY(hipMemsetAsync(g->answer_embedding, 0, (size_t)d.A * d.da * 4, s));
    }
    return NCX_OK;
}

int ncx_ws_region(const ncx_dims* dp, int32_t which, size_t* offset, size_t* bytes) {
    if (check_dims(dp) != NCX_OK) return NCX_E_DIMS;
    if (!offset || !bytes) return NCX_E_NULL;
    if (which != NCX_WS_DGT && which != NCX_WS_H1 && which != NCX_WS_DPRE1) return NCX_E_FLAGS;
    const WsLayout w = ws_layout(*dp);
    if (which == NCX_WS_H1 || which == NCX_WS_DPRE1) {
        *offset = which == NCX_WS_H1 ? w.h[0] : w.dpre[(dp->L - 1) & 1];
        *bytes = (size_t)dp->B * dp->K * dp->H * 4;
        return NCX_OK;
    }
    // the block a DP job sums between phases 5 / 3 and 4 is the one backward_impl fills: the same predicate decides its form
    if (!emb_nt_form(*dp)) { *offset = w.dgt; *bytes = (dp->flags & NCX_F_A_EMB) ? (size_t)2 * dp->H * dp->A * 4 : 0; }               // dGt | dGgt, [H][A] x 2
    else { *offset = w.dgtT; *bytes = (size_t)2 * dp->A * pad_to(dp->H, 4) * 4; }                                                     // dGt^T | dGgt^T, [A][pad4(H)] x 2
    return NCX_OK;
}

int ncx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, float grad_scale, void* stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return NCX_E_NULL;
    if (step < 1 || n == 0) return NCX_E_DIMS;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return NCX_E_WORKSPACE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    size_t blocks = (n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq, n,
                       step_size, beta1, beta2, eps, inv_bc2_sqrt, grad_scale);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

// ---- fused MUTAN producer ---------------------------------------------------------------------------
struct VqaLayout { size_t xq, hq, xv, wcp, slab, slab_bytes, total; };
// dst[r][0 .. cols) = src[r][0 .. cols), dst[r][cols .. ldd) = 0: the classifier weights with their rows zero-padded to whole 32-column
// k-steps for the fused forward kernel (ncx_main.h reads the weight side of a segment up to the next multiple of 32 columns)
__global__ __launch_bounds__(256) void k_pad_rows(const float* __restrict__ src, long long lds_, int cols, float* __restrict__ dst, int ldd, int rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * ldd) return;
    const int r = (int)(i / ldd), c = (int)(i - (long long)r * ldd);
    dst[i] = c < cols ? src[(long long)r * lds_ + c] : 0.f;
}
static VqaLayout vqa_layout(const ncx_dims& d, const ncx_mutan_params& m, GemmPlan* plans /*[5]*/) {
    VqaLayout w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const long long Mv = (long long)d.B * (d.K + 1), RZ = (long long)m.R * d.dz;
    w.xq = take((size_t)d.B * m.dhq * 4);
    w.hq = take((size_t)d.B * RZ * 4);
    w.xv = take((size_t)Mv * m.dhv * 4);
    w.wcp = take(d.dz % 32 ? (size_t)d.A * pad_to(d.dz, 32) * 4 : 0);
    // 0: xq = act(q Wq^T)  1: hq = xq Whq^T  2: xv = act(gather(v) Wv^T)  3: z (fold, never split)  4: a = z Wc^T
    const long long shp[5][3] = {{d.B, m.dhq, ks(d.dq)}, {d.B, RZ, ks(m.dhq)}, {Mv, m.dhv, ks(d.dv)}, {Mv, d.dz, 0}, {(long long)d.B * d.K, d.A, ks(d.dz)}};
    long long slab = 0;
    for (int i = 0; i < 5; ++i) {
        plans[i] = plan_gemm(FORM_NT, shp[i][0], shp[i][1], shp[i][2], true);
        {   // experiment hook
            char name[32]; snprintf(name, sizeof name, "NCX_VQA_CFG_%d", i);
            const char* c = hook_env(name);
            if (c) { plans[i].cfg = atoi(c); plans[i].split = 1; }
        }
        if (i == 3) { plans[i].cfg = CFG_64x64; plans[i].split = 1; }
        int bm, bn; cfg_tile(plans[i].cfg, bm, bn);
        const long long e = plans[i].split > 1 ? (long long)WgMap{(int)cdiv(shp[i][0], bm), (int)cdiv(shp[i][1], bn), plans[i].split}.count() * bm * bn : 0;
        if (e > slab) slab = e;
    }
    w.slab_bytes = (size_t)slab * 4;
    w.slab = take(w.slab_bytes);
    w.total = off;
    return w;
}
static int check_mutan(const ncx_dims* d, const ncx_mutan_params* m) {
    if (!d || !m) return NCX_E_NULL;
    if (d->B < 1 || d->K < 1 || d->dv < 4 || d->dq < 4 || d->dz < 4 || d->A < 4 || d->n_img < 1) return NCX_E_DIMS;
    if (m->dhv < 4 || m->dhq < 4 || m->R < 1 || m->R > NCX_MAX_SEG) return NCX_E_DIMS;
    if ((m->act_v != 0 && m->act_v != 2) || (m->act_q != 0 && m->act_q != 2)) return NCX_E_FLAGS;
    if (!m->wv || !m->bv || !m->wq || !m->bq || !m->whv || !m->bhv || !m->whq || !m->bhq || !m->wc || !m->bc) return NCX_E_NULL;
    return NCX_OK;
}

size_t ncx_vqa_workspace_bytes(const ncx_dims* d, const ncx_mutan_params* m) {
    if (check_mutan(d, m) != NCX_OK) return 0;
    GemmPlan plans[5];
    return vqa_layout(*d, *m, plans).total;
}

int ncx_vqa_forward(const ncx_dims* dp, const float* feats, const int32_t* img_idx, const float* q_emb,
                    const ncx_mutan_params* mp, void* workspace, size_t workspace_bytes,
                    float* z_orig, float* z_knns, float* a_knns, float* a_orig, void* stream_) {
    int rc = check_mutan(dp, mp);
    if (rc != NCX_OK) return rc;
    if (!feats || !img_idx || !q_emb || !workspace || !z_orig || !z_knns || !a_knns) return NCX_E_NULL;
    const ncx_dims& d = *dp; const ncx_mutan_params& m = *mp;
    GemmPlan plans[5];
    const VqaLayout w = vqa_layout(d, m, plans);
    if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return NCX_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* xq = (float*)(ws + w.xq); float* hq = (float*)(ws + w.hq); float* xv = (float*)(ws + w.xv);
    float* slab = (float*)(ws + w.slab);
    const int Mv = d.B * (d.K + 1), RZ = m.R * d.dz;
    {   // x_q = act_q(q . Wq^T + bq)                                            fusion.py:88-93
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = d.B;
        a.a[0] = x_plain(q_emb, d.dq, d.B, d.dq); a.b[0] = x_plain(m.wq, d.dq, m.dhq, d.dq); a.klen[0] = d.dq;
        a.out[0] = xq; a.ldo[0] = m.dhq; a.n_cols[0] = m.dhq; a.epi.relu = m.act_q;
        rc = run_gemm_impl(a, FORM_NT, plans[0], slab, w.slab_bytes, m.bq, s); if (rc) return rc;
    }
    {   // hq[b][r*dz + j] = x_q . Whq_r^T + bhq_r   (all R at once)               fusion.py:103-107
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = d.B;
        a.a[0] = x_plain(xq, m.dhq, d.B, m.dhq); a.b[0] = x_plain(m.whq, m.dhq, RZ, m.dhq); a.klen[0] = m.dhq;
        a.out[0] = hq; a.ldo[0] = RZ; a.n_cols[0] = RZ;
        rc = run_gemm_impl(a, FORM_NT, plans[1], slab, w.slab_bytes, m.bhq, s); if (rc) return rc;
    }
    if (d.dv % 32 == 0 && d.dv >= 64 && m.dhv >= 4 && !hook_env("NCX_VQA_NO_MAIN")) {
        // x_v on the fused forward kernel (ncx_main.h): one gathered segment, bias + activation in the epilogue (round 3: 249 -> see DESIGN 5b)
        MainArgs a{}; a.M = Mv; a.N = m.dhv; a.nseg = 1;
        a.seg[0].kind = MK_GATHER; a.seg[0].a = feats; a.seg[0].lda = d.dv; a.seg[0].idx = img_idx; a.seg[0].klen = d.dv;
        a.seg[0].b = m.wv; a.seg[0].ldb = d.dv;
        a.out = xv; a.ldo = m.dhv; a.epi.bias = m.bv; a.epi.relu = m.act_v; a.split = 1;
        rc = main_forward(a, s); if (rc) return rc;
    } else
    {   // x_v = act_v(gather(feats, img_idx) . Wv^T + bv) for the B*(K+1) images      fusion.py:82-87 (+ the host gather)
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = Mv;
        a.a[0] = x_gather(feats, d.dv, img_idx, Mv, d.dv); a.b[0] = x_plain(m.wv, d.dv, m.dhv, d.dv); a.klen[0] = d.dv;
        a.out[0] = xv; a.ldo[0] = m.dhv; a.n_cols[0] = m.dhv; a.epi.relu = m.act_v;
        rc = run_gemm_impl(a, FORM_NT, plans[2], slab, w.slab_bytes, m.bv, s); if (rc) return rc;
    }
    if (mutan_fold_supported(d, m)) {
        // z = sum_r (x_v . Whv_r^T + bhv_r) * hq_r[question] as ONE product per question against Weff_q = sum_r diag(hq_r[q]) Whv_r,
        // built on the vector ALU on the way into LDS (ncx_mutan.hip): 4.2 + 0.7 GF instead of 33.2 at configs[2]        fusion.py:96-115
        rc = mutan_fold(d, m, xv, hq, z_orig, z_knns, s); if (rc) return rc;
    } else
    {   // z = sum_r (x_v . Whv_r^T + bhv_r) * hq_r[question]  -> z_orig / z_knns     fusion.py:96-115
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = m.R; a.M = Mv;
        for (int r = 0; r < m.R; ++r) {
            a.a[r] = x_plain(xv, m.dhv, Mv, m.dhv);
            a.b[r] = x_plain(m.whv + (long long)r * d.dz * m.dhv, m.dhv, d.dz, m.dhv);
            a.klen[r] = m.dhv;
        }
        a.out[0] = z_knns; a.ldo[0] = d.dz; a.n_cols[0] = d.dz; a.split[0] = 1;
        a.epi.fold_mul = hq; a.epi.ld_fold = RZ; a.epi.fold_div = d.K + 1; a.epi.fold_bias = m.bhv;
        a.epi.rowsplit_g = d.K + 1; a.epi.out0 = z_orig; a.epi.ldo0 = d.dz;
        rc = run_gemm_nt_fold(a, s); if (rc) return rc;
    }
    if (d.dz % 4 == 0 && d.dz >= 4 && d.A % 4 == 0 && !hook_env("NCX_VQA_NO_MAIN")) {      // (the kernel's epilogue wants whole 16-byte pieces per output row)
        // a_knns = z_knns . Wc^T + bc on the fused forward kernel (round 4): 11-12 k-steps per workgroup -> 64 x 64 tiles at three
        // workgroups per CU (the plan the answer-embedding gradient takes); generic engine: 241 us at configs[2]          noatt.py:24-29
        const float* wc = m.wc; long long ldw = d.dz;
        if (d.dz % 32) {
            float* wcp = (float*)(ws + w.wcp);
            const int ldd = pad_to(d.dz, 32);
            hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)cdiv((long long)d.A * ldd, 256)), dim3(256), 0, s, m.wc, (long long)d.dz, d.dz, wcp, ldd, d.A);
            NCX_HIP_TRY(hipGetLastError());
            wc = wcp; ldw = ldd;
        }
        MainArgs a{}; a.M = d.B * d.K; a.N = d.A; a.nseg = 1;
        a.seg[0].kind = MK_PLAIN; a.seg[0].a = z_knns; a.seg[0].lda = d.dz; a.seg[0].klen = d.dz; a.seg[0].b = wc; a.seg[0].ldb = ldw;
        a.out = a_knns; a.ldo = d.A; a.epi.bias = m.bc; a.split = 1;
        rc = main_forward(a, s); if (rc) return rc;
    } else
    {   // a_knns = z_knns . Wc^T + bc                                               noatt.py:24-29 (dropout off in eval)
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = d.B * d.K;
        a.a[0] = x_plain(z_knns, d.dz, d.B * d.K, d.dz); a.b[0] = x_plain(m.wc, d.dz, d.A, d.dz); a.klen[0] = d.dz;
        a.out[0] = a_knns; a.ldo[0] = d.A; a.n_cols[0] = d.A;
        rc = run_gemm_impl(a, FORM_NT, plans[4], slab, w.slab_bytes, m.bc, s); if (rc) return rc;
    }
    if (a_orig) {
        GemmArgs a{}; a.mode = MODE_CHAIN; a.nseg = 1; a.M = d.B;
        a.a[0] = x_plain(z_orig, d.dz, d.B, d.dz); a.b[0] = x_plain(m.wc, d.dz, d.A, d.dz); a.klen[0] = d.dz;
        a.out[0] = a_orig; a.ldo[0] = d.A; a.n_cols[0] = d.A; a.split[0] = 1;
        GemmPlan pl; pl.cfg = CFG_64x64; pl.split = 1;
        rc = run_gemm_impl(a, FORM_NT, pl, slab, w.slab_bytes, m.bc, s); if (rc) return rc;
    }
    return NCX_OK;
}

// ---- diagnostics ---------------------------------------------------------------------------------------
int ncx_profile_begin(uint32_t gemm_mask, int32_t max_launches) {
    if (g_prof.on || gemm_mask == 0 || (gemm_mask >> U_COUNT) != 0 || max_launches < 1 || max_launches > 65536) return NCX_E_DIMS;
    g_prof.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * (size_t)max_launches);
    g_prof.ids = (int*)malloc(sizeof(int) * (size_t)max_launches);
    if (!g_prof.ev || !g_prof.ids) return NCX_E_NULL;
    for (int i = 0; i < 2 * max_launches; ++i) NCX_HIP_TRY(hipEventCreate(&g_prof.ev[i]));
    g_prof.mask = gemm_mask; g_prof.n = 0; g_prof.cap = max_launches; g_prof.on = true;
    return NCX_OK;
}

int ncx_profile_end(float* ms, int32_t* ids, int32_t cap) {
    if (!g_prof.on) return NCX_E_FLAGS;
    int n = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) break;
        if (n < cap) { if (ms) ms[n] = t; if (ids) ids[n] = g_prof.ids[i]; }
        ++n;
    }
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    free(g_prof.ev); free(g_prof.ids);
    g_prof.ev = nullptr; g_prof.ids = nullptr; g_prof.on = false; g_prof.mask = 0; g_prof.n = 0; g_prof.cap = 0;
    return n < cap ? n : cap;
}

int ncx_profile_stamps(unsigned long long* stamps, int64_t words) {
    if ((stamps == nullptr) != (words == 0) || words < 0) return NCX_E_DIMS;
    int dev = 0;
    NCX_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return NCX_E_DIMS;
    // armed for the CURRENT device only (the buffer lives there); disarm = pointer first, so no reader pairs the old pointer with 0 words
    if (!stamps) { g_stamps[dev].ptr.store(nullptr, std::memory_order_release); g_stamps[dev].words.store(0, std::memory_order_release); }
    else { g_stamps[dev].ptr.store(nullptr, std::memory_order_release); g_stamps[dev].words.store(words, std::memory_order_release);
           g_stamps[dev].ptr.store(stamps, std::memory_order_release); }
    return NCX_OK;
}

int ncx_wgmap_check(int32_t tiles_m, int32_t tiles_n, int32_t S) {
    if (tiles_m < 1 || tiles_n < 1 || S < 1 || (long long)tiles_m * tiles_n * S > (1 << 24)) return NCX_E_DIMS;
    const WgMap w{tiles_m, tiles_n, S};
    const int total = tiles_m * tiles_n * S, count = w.count();
    if (count < total) return 1;
    unsigned char* seen = (unsigned char*)calloc((size_t)total, 1);
    if (!seen) return NCX_E_NULL;
    int bad = 0, nvalid = 0;
    for (int lw = 0; lw < count && !bad; ++lw) {
        int tm = -1, tn = -1, z = -1;
        if (!w.decode(lw, tm, tn, z)) continue;
        ++nvalid;
        if (tm < 0 || tm >= tiles_m || tn < 0 || tn >= tiles_n || z < 0 || z >= S) { bad = 2; break; }
        if (w.encode(tm, tn, z) != lw) { bad = 3; break; }
        unsigned char& f = seen[((size_t)z * tiles_m + tm) * tiles_n + tn];
        if (f) { bad = 4; break; }
        f = 1;
    }
    if (!bad && nvalid != total) bad = 5;
    free(seen);
    return bad;
}

int ncx_plan_query(const ncx_dims* d, int32_t gemm_id, int32_t* out6) {
    if (check_dims(d) != NCX_OK || !out6 || gemm_id < 0 || gemm_id >= U_COUNT) return NCX_E_DIMS;
    GemmUse u[U_COUNT];
    list_uses(*d, u);
    out6[0] = u[gemm_id].form; out6[1] = (int32_t)u[gemm_id].M; out6[2] = (int32_t)u[gemm_id].N;
    out6[3] = (int32_t)u[gemm_id].ksteps; out6[4] = u[gemm_id].plan.cfg; out6[5] = u[gemm_id].plan.split;
    // linear_1 / hidden-layer forward: the fused kernel of ncx_main.h (tile codes 5: 48x128, 6: 48x64 with the per-triplet fold)
    const bool fast = (gemm_id == U_MAIN && main_fwd_dims_ok(*d)) || (gemm_id == U_FWD_L && hidden_fwd_dims_ok(*d) && !(d->flags & NCX_F_BF16));
    if (fast) {
        const long long M = (long long)d->B * d->K, T = u[gemm_id].ksteps;
        const int sp = main_split(M, d->H, T);
        const bool vfold = gemm_id == U_MAIN && (d->flags & NCX_F_V_MULT) && (d->K == 24 || (d->K == 48 && main_fold_rows(M, d->H) >= 96)) && d->dv % 32 == 0 && d->dv >= 64 && sp == 1;
        const long long tiles96 = ((M + 95) / 96) * ((d->H + 127) / 128);
        out6[4] = vfold ? (main_fold_rows(M, d->H) == 192 ? 8 : (main_fold_rows(M, d->H) == 96 || d->K == 48) ? 7 : 6) : (sp == 1 && tiles96 * 10 >= (long long)num_cus() * 9) ? CFG_96x128 : 5;
        out6[5] = sp;
    }
    return NCX_OK;
}

}  // extern "C"
