// ncx_mutan.hip -- the rank-R fusion of the frozen MUTAN producer with per-question effective weights (round 4).
//
// Reference: MutanFusion.forward, vqa/models/fusion.py:96-115 (called from CXModelBase.vqa_forward, cx.py:83-92):
//   z[r][j] = sum_rr (x_v[r] . Whv_rr[j] + bhv_rr[j]) * hq_rr[question(r)][j]            r = one of the B (K + 1) images, rr < R
// As written that is R products of [B (K + 1), dhv] x [dhv, dz]: 33.2 GF at configs[2] (B = 512, K = 24, dhv = dz = 360, R = 10), which
// the generic engine ran as a chained GEMM with a per-pair fold in 443 us.  But the K + 1 images of a question share hq, so
//   z[r][j] = x_v[r] . Weff_q[j] + c_q[j],     Weff_q[j][k] = sum_rr hq_rr[q][j] Whv_rr[j][k],     c_q[j] = sum_rr hq_rr[q][j] bhv_rr[j]
// -- ONE product per question against an effective weight tile that is built on the vector ALU on the way into LDS and never touches
// memory (the forward twin of what ncx_main.h does with diag(v_o) W_m): 4.2 GF on the matrix cores (the 25 rows of a question padded to
// 32) + 0.66 G fused multiply-adds on the vector ALU instead of 33.2 GF.  Only the fp32 summation order changes (reference tolerance of
// the path: z within 1e-4 of its maximum; tests/test_dropin_gpu.py).
//
// One workgroup = 8 waves = 8 questions x 32 output columns: wave w multiplies question w's 32-row block (K + 1 <= 32 rows, the rest
// padding whose outputs are dropped) with its own effective tile; the Whv slices every question needs (R x 32 x 32 floats per k-step)
// are fetched once per workgroup, straight into registers (thread = one (column, 16-byte k-quad) pair, its R hq factors per question
// stay in registers for the whole workgroup).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "ncx_internal.h"

namespace ncx {

constexpr int MF_RMAX = 10, MF_Q = 8, MF_J = 32, MF_KS = 32, MF_PT = 36, MF_TH = 512;

typedef const __attribute__((address_space(1))) float* mf_gfptr;
typedef const __attribute__((address_space(1))) f32x4u* mf_gf4ptr;

struct MutanFoldArgs {
    const float* xv; const float* hq; const float* whv; const float* bhv;
    float* z_orig; float* z_knns;
    long long ld_hq;
    int B, K1, dz, dhv, R, groups, tiles_j;
};

__global__ __launch_bounds__(MF_TH, 1) void k_mutan_fold(const MutanFoldArgs a) {
    constexpr int Q = MF_Q, J = MF_J, KS = MF_KS, P = MF_PT, T = MF_TH, RMAX = MF_RMAX;
    extern __shared__ __attribute__((aligned(16))) float mfs[];
    float* const lds_a = mfs;                              // [2][Q][32 rows][P]
    float* const lds_w = mfs + 2 * Q * 32 * P;             // [2][Q][J][P]
    float* const lds_h = mfs + 2 * Q * (32 + J) * P;       // [RMAX][J][Q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    // Workgroup ids are dealt round-robin over the 8 XCDs (id & 7), each with its own 4 MB L2.  Column-tile major inside an XCD: the
    // workgroups resident together then share a few Whv slices (R x 32 x dhv floats each, the operand every question needs) instead of
    // cycling all of Whv (5.2 MB at configs[2]) through every L2
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    const int gpx = (a.groups + 7) >> 3;                    // question groups per XCD
    const int tj = local / gpx, g = (local - tj * gpx) * 8 + xcd;
    if (g >= a.groups || tj >= a.tiles_j) return;
    const int q0 = g * Q, j0 = tj * J;
    const int nst = (a.dhv + KS - 1) / KS;

    // ---- roles ---------------------------------------------------------------------------------------------------------------------
    // Waves 0-3 BUILD the effective weights: thread = (column jl, k-quad kq) of the 32 x 32 tile; it requests its R Whv quads once and forms
    // the tile of all 8 questions from them.  Waves 4-7 stage the x_v rows (8 questions x 32 rows x 8 quads, 8 per thread).  (First version:
    // every thread built 4 questions, so each Whv quad was requested by two threads: 112 KB per workgroup and k-step through a texture
    // path that delivers ~35 GB/s per CU -- 125 us; one SIMD = one builder wave + one stager wave, so the two roles stay balanced.)
    const bool builder = __builtin_amdgcn_readfirstlane(tid >> 6) < 4;
    const int lt = tid & 255, jl = lt >> 3, kq = lt & 7;
    const int jc = min(j0 + jl, a.dz - 1);
    // the workgroup's hq factors in LDS as [rank][column][question]: a build thread reads the eight questions of one (rank, column) with two
    // 16-byte loads (kept in registers they cost 40+ VGPRs and the kernel spilled)
    for (int i = tid; i < RMAX * J * Q; i += T) {
        const int q = i % Q, jx = (i / Q) % J, rr = i / (Q * J);
        const int b = min(q0 + q, a.B - 1), jj = min(j0 + jx, a.dz - 1);
        lds_h[i] = rr < a.R ? a.hq[(long long)b * a.ld_hq + (long long)rr * a.dz + jj] : 0.f;
    }
    // one pointer per load slot; builders: rank i of Whv (ranks beyond R re-read the last one, times hq = 0); stagers: x_v row (lt + 256 i) >> 3
    mf_gfptr pl[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int f = lt + 256 * min(i, 7), row = f >> 3, q = row >> 5, ri = row & 31;
        const long long r = (long long)min(q0 + q, a.B - 1) * a.K1 + min(ri, a.K1 - 1);
        pl[i] = builder ? (mf_gfptr)a.whv + ((long long)min(i, a.R - 1) * a.dz + jc) * a.dhv : (mf_gfptr)a.xv + r * a.dhv;
    }
    // TWO register sets of global loads in flight: tile t + 1 (requested during step t - 1) is built while tile t + 2 is on its way
    f32x4 vr[2][RMAX];
    auto issue = [&](auto set_c, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        const int c = min(min(t, nst - 1) * KS + 4 * kq, a.dhv - 4);         // (both roles: the thread's 16-byte k-quad is lt & 7)
#pragma unroll
        for (int i = 0; i < 8; ++i) vr[S][i] = *(mf_gf4ptr)(pl[i] + c);
        if (builder) {
#pragma unroll
            for (int i = 8; i < RMAX; ++i) vr[S][i] = *(mf_gf4ptr)(pl[i] + c);
        }
    };
    // the tile of k-step t from the registers -> LDS buffer buf (columns beyond dhv: zero effective weights; dhv % 4 == 0)
    auto build = [&](auto set_c, int t, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if (builder) {
            const float keep = t * KS + 4 * kq < a.dhv ? 1.f : 0.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {                 // questions 4 hf .. 4 hf + 3
                f32x4 w4[4];
#pragma unroll
                for (int rr = 0; rr < RMAX; ++rr) {
                    const f32x4 h4 = *(const f32x4*)(lds_h + (rr * J + jl) * Q + 4 * hf);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) w4[q][e] = rr == 0 ? h4[q] * vr[S][rr][e] : __builtin_fmaf(h4[q], vr[S][rr][e], w4[q][e]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(lds_w + ((buf * Q + 4 * hf + q) * J + jl) * P + 4 * kq) = w4[q] * keep;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(f32x4*)(lds_a + (buf * Q * 32 + ((lt + 256 * i) >> 3)) * P + 4 * kq) = vr[S][i];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma_step = [&](int buf) __attribute__((always_inline)) {
        const float* pa_ = lds_a + ((buf * Q + wave) * 32 + li) * P + 2 * lk;
        const float* pb_ = lds_w + ((buf * Q + wave) * J + li) * P + 2 * lk;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x2 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { af[i] = *(const f32x2*)(pa_ + i * 16 * P + 8 * s); bf[i] = *(const f32x2*)(pb_ + i * 16 * P + 8 * s); }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    issue(S0{}, 0);
    issue(S1{}, 1);
    __syncthreads();                                       // (the hq table)
    build(S0{}, 0, 0);
    issue(S0{}, 2);
    __syncthreads();
    // step t: the matrix cores multiply tile t (LDS buffer t & 1) while the vector ALU builds tile t + 1 from the register set that was
    // requested during step t - 1, then requests tile t + 3 into it
    auto step = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef std::integral_constant<int, PAR ^ 1> SS;
        mfma_step(PAR);
        __builtin_amdgcn_sched_barrier(0);
        build(SS{}, t + 1, PAR ^ 1);                        // (beyond the last tile: a copy of it into the buffer nobody reads again)
        __builtin_amdgcn_sched_barrier(0);                   // (the loads of tile t + 3 reuse the registers the build has just consumed)
        issue(SS{}, t + 3);
        __syncthreads();
    };
    int t = 0;
    for (; t + 1 < nst; t += 2) { step(S0{}, t); step(S1{}, t + 1); }
    if (t < nst) step(S0{}, t);
    // ---- epilogue: + c_q[j], rows -> z_orig (image 0) / z_knns (images 1 .. K) ------------------------------------------------------------
    const int b = q0 + wave;
    if (b >= a.B) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j0 + 16 * j + li;
        if (n >= a.dz) continue;
        // c_q[n] = sum_rr hq_rr[q][n] bhv_rr[n]: the hq factors are in LDS, the R bias values are requested together (a load per iteration
        // of a runtime-length loop is a dependent round trip per rank: 10 x ~1 us at the end of every workgroup)
        float bh[RMAX];
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) bh[rr] = ((mf_gfptr)a.bhv)[(long long)min(rr, a.R - 1) * a.dz + n];
        float c = 0.f;
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) c = __builtin_fmaf(lds_h[(rr * J + 16 * j + li) * Q + wave], bh[rr], c);     // (ranks beyond R: hq = 0)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ri = 16 * i + 4 * lk + rg;
                if (ri >= a.K1) continue;
                const float v = acc[i][j][rg] + c;
                if (ri == 0) a.z_orig[(long long)b * a.dz + n] = v;
                else a.z_knns[((long long)b * (a.K1 - 1) + ri - 1) * a.dz + n] = v;
            }
    }
}

bool mutan_fold_supported(const ncx_dims& d, const ncx_mutan_params& m) {
    if (hook_env("NCX_NO_MUTAN_FOLD")) return false;
    return d.K + 1 <= 32 && m.R >= 1 && m.R <= MF_RMAX && m.dhv >= 4 && m.dhv % 4 == 0 && d.dz >= 1;
}

int mutan_fold(const ncx_dims& d, const ncx_mutan_params& m, const float* xv, const float* hq, float* z_orig, float* z_knns, hipStream_t s) {
    MutanFoldArgs a{};
    a.xv = xv; a.hq = hq; a.whv = m.whv; a.bhv = m.bhv; a.z_orig = z_orig; a.z_knns = z_knns;
    a.ld_hq = (long long)m.R * d.dz; a.B = d.B; a.K1 = d.K + 1; a.dz = d.dz; a.dhv = m.dhv; a.R = m.R;
    a.groups = (d.B + MF_Q - 1) / MF_Q; a.tiles_j = (d.dz + MF_J - 1) / MF_J;
    const int lds = 2 * MF_Q * (32 + MF_J) * MF_PT * 4 + MF_RMAX * MF_J * MF_Q * 4;
    static DevMask attr{0};
    NCX_HIP_TRY(set_max_lds_once(attr, (const void*)k_mutan_fold, lds));
    const int grid = ((a.groups + 7) / 8) * 8 * a.tiles_j;
    hipLaunchKernelGGL(k_mutan_fold, dim3(grid), dim3(MF_TH), lds, s, a);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

}  // namespace ncx
