// ncx_dwkm.hip -- weight gradient of the v_other and v_orig*v_other segments of linear_1 in ONE MFMA pass.
//
//   dW1[:, v_other][h][c] = sum_r dpre[r][h] * v_k[r][c]
//   dW1[:, v_mult ][h][c] = sum_r dpre[r][h] * v_o[b(r)][c] * v_k[r][c]  =  sum_b v_o[b][c] * T_b[h][c],   T_b = sum_k dpre[(b,k)][h] v_k[(b,k)][c]
//
// The K candidate rows of a triplet share v_o, so a reduction step that is exactly one triplet (K rows = K/4
// v_mfma_f32_16x16x4_f32 steps) yields T_b in a scratch accumulator, which is then folded into BOTH outputs on the vector
// ALU: acc_k += T_b, acc_m += v_o[b] (.) T_b.  Half the MFMA work of treating the two segments as separate GEMM problems
// (2 x 12.9 GF at configs[1]).  Reference: the same gradients autograd produces for cx.py:296,309-322.
//
// 128 (h) x 64 (c) output tile pair per workgroup, 256 threads = 2 x 2 waves (64 x 32 each), split over the triplets in
// S chunks (chunk z on XCD z: its dpre rows stay in that L2), partial tiles to a slab, fixed-order reduction afterwards.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "ncx_internal.h"
#include "ncx_dwred.h"

namespace ncx {

// LDS pitches = 16 mod 32: ds_read_b32 at its 2-cycle floor

// K = rows per reduction step (a whole triplet, or a 24-row part of one: the rows of a step share v_o); Kc = candidates
// per triplet, a multiple of K.
// EDGE: some tile of the launch reaches beyond H or dv (16-byte windows slid left and repaired); interior launches compile
// the repair out -- the reduction step is then ONE basic block (a branch in it costs ~15 %: ncx_main.h).
// T threads: 256 = 2 x 2 waves on a 128 (or 64) x 64 tile pair, two (four) workgroups per CU; 512 = 2 x 4 waves on a 128 x 128 tile pair, ONE
// workgroup per CU (round 3: the dpre tile is shared by twice the columns, no old / young workgroup pair)
template <int K, int KM_BM, int OCC, bool EDGE, int ABL = 0, int T = 256>
__global__ __launch_bounds__(T, OCC) void k_dw_km(const float* __restrict__ dpre, int H, const float* __restrict__ feats, int dv,
                                                  const int* __restrict__ idx_k, const int* __restrict__ idx_o, int B, int Kc,
                                                  int chunk, int tiles_m, int S, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float km_smem[];
    constexpr int KM_BN = T == 512 ? 128 : 64, KM_PB = KM_BN + 16, WGN = KM_BN / 32;
    constexpr int KM_PA = KM_BM + 16, QA = KM_BM / 4, QB = KM_BN / 4;   // (QA, QB: 16-byte quads per A / B row)
    constexpr int NA = (K * QA + T - 1) / T, NB = (K * QB + T - 1) / T;
    constexpr int KA = NA * T / QA, KB = NB * T / QB;   // tile rows incl. the ones only the unconditional stores touch
    constexpr int a_elems = KA * KM_PA, b_elems = KB * KM_PB;
    float* const lds_a = km_smem;                       // [2][KA][KM_PA]
    float* const lds_b = km_smem + 2 * a_elems;         // [2][KB][KM_PB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave / WGN) * (KM_BM / 2), wn0 = (wave % WGN) * 32;
    const int z = blockIdx.x % S, t = blockIdx.x / S;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int m0 = tm * KM_BM, n0 = tn * KM_BN;
    const int spt = Kc / K;                              // reduction steps per triplet
    const int t0 = z * chunk, t1 = min(t0 + chunk, B);   // triplets of this chunk
    if (t0 >= B) return;
    const int b0 = t0 * spt, b1 = t1 * spt;              // reduction steps of this chunk (step s: rows s*K .. s*K+K-1)

    constexpr int WM = KM_BM / 32, WN = 2;
    f32x4 acc_k[WM][WN], acc_m[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) { acc_k[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // loader roles: A tile [K][128]: float4 f = tid + 256 i -> row f / 32, quad f % 32;
    //               B tile [K][64]:  float4 f = tid + 256 i -> row f / 16, quad f % 16.
    // Two register sets: the loads of triplet b+2 are issued before the MFMAs of triplet b, so every load has two
    // reduction steps to land; every load / LDS store is unconditional (triplet index clamped, a surplus step is folded
    // with weight 0): a branch around the loads makes the compiler drain vmcnt first.
    f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    // Gather indices travel one more step ahead than the rows they name (set `ix` for triplet b+4 is requested while the rows
    // of triplet b+2 are): an index fetched inside issue() put a full memory round trip in front of that step's row loads.
    // The B tile's spare rows (K .. KB-1) carry the triplet's v_o slice: row K is what the fold multiplies by, so v_o needs
    // no registers of its own (a register copy of it per step made the loop wait for loads it had just issued).
    static_assert(KB > K, "the B tile needs a spare row for v_o");
    struct Ix { int k[NB]; };
    auto issue_idx = [&](Ix& ix, int b) __attribute__((always_inline)) {
        b = min(b, b1 - 1);
        const long long r0 = (long long)b * K;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (tid + T * i) / QB;
            ix.k[i] = *(row < K ? idx_k + r0 + row : idx_o + r0);         // (idx_o is per row: any row of the step names its triplet's v_o)
        }
    };
    auto issue = [&](f32x4 (&ra)[NA], f32x4 (&rb)[NB], const Ix& ix, int b) __attribute__((always_inline)) {
        const float keep = b < b1 ? 1.f : 0.f;
        b = min(b, b1 - 1);
        const long long r0 = (long long)b * K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + T * i, row = min(f / QA, K - 1), c = m0 + 4 * (f % QA);
            ra[i] = EDGE ? load_window(dpre + (r0 + row) * H, c, H) : *(const f32x4u*)(dpre + (r0 + row) * H + c);   // (window repaired in stash)
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + T * i, c = n0 + 4 * (f % QB);
            rb[i] = EDGE ? load_window(feats + (long long)ix.k[i] * dv, c, dv) : *(const f32x4u*)(feats + (long long)ix.k[i] * dv + c);
        }
        return keep;
    };
    auto stash = [&](const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + T * i;
            const f32x4 v = EDGE ? fix_window(ra[i], m0 + 4 * (f % QA), H) : ra[i];
            *(f32x4*)(lds_a + buf * a_elems + (f / QA) * KM_PA + 4 * (f % QA)) = v;      // (rows >= K: copies of row K-1, never read)
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + T * i;
            const f32x4 v = EDGE ? fix_window(rb[i], n0 + 4 * (f % QB), dv) : rb[i];
            *(f32x4*)(lds_b + buf * b_elems + (f / QB) * KM_PB + 4 * (f % QB)) = v;
        }
    };
    constexpr int nk4 = K / 4;
    // One reduction step = one triplet, software-pipelined over THREE triplets (round 3): under the MFMAs of triplet b run
    //   * the fold of triplet b-1 (its T_b and v_o slice are carried in registers: tp / vp) -- the fold used to run after the last
    //     MFMA of its own step with the matrix pipe idle (timing ablation: 9 us of the kernel),
    //   * the LDS stores of triplet b+1's tile into the other buffer (they used to sit between the step and its barrier: 20 us),
    //   * the global loads of triplet b+2 and the gather indices of triplet b+4.
    // The operand fragments of sub-step s4+2 are read from LDS under the MFMAs of sub-step s4 (read just before use, each group
    // of 4 MFMAs waited ~100 cycles for its ds_read).  Same arithmetic in the same order as before: results bit-identical.
    auto read_frag = [&](const float* pa, const float* pb, int s4, float (&af)[WM], float (&bf)[WN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = pa[s4 * 4 * KM_PA + 16 * i];
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = pb[s4 * 4 * KM_PB + 16 * j];
    };
    f32x4 tp[WM][WN];                                    // T of the previous triplet (not folded yet)
    float vp[WN], keep_p = 0.f;                          // its v_o slice (times its weight) and its weight
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) tp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WN; ++j) vp[j] = 0.f;
    auto fold_block = [&](int blk) __attribute__((always_inline)) {          // block blk = (i, j) of the previous triplet's T
        const int i = blk / WN, j = blk % WN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ABL == 4) { acc_k[i][j][q] += tp[i][j][q]; continue; }
            acc_k[i][j][q] = __builtin_fmaf(tp[i][j][q], keep_p, acc_k[i][j][q]);
            acc_m[i][j][q] = __builtin_fmaf(tp[i][j][q], vp[j], acc_m[i][j][q]);
        }
        // pin the fold to the sub-step it was placed in (the optimiser otherwise sinks a whole step's fold down to its only user,
        // the next step's fold: one step with 128 vector instructions under its MFMAs, the other with none)
        asm volatile("" : "+v"(acc_k[i][j]), "+v"(acc_m[i][j]));
    };
    // stash item it (0 .. NA + NB - 1) of the NEXT triplet's register set into LDS buffer `buf`
    auto stash_item = [&](const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], int buf, int it) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i != it) continue;
            const int f = tid + T * i;
            const f32x4 v = EDGE ? fix_window(ra[i], m0 + 4 * (f % QA), H) : ra[i];
            *(f32x4*)(lds_a + buf * a_elems + (f / QA) * KM_PA + 4 * (f % QA)) = v;      // (rows >= K: copies of row K-1, never read)
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i + NA != it) continue;
            const int f = tid + T * i;
            const f32x4 v = EDGE ? fix_window(rb[i], n0 + 4 * (f % QB), dv) : rb[i];
            *(f32x4*)(lds_b + buf * b_elems + (f / QB) * KM_PB + 4 * (f % QB)) = v;
        }
    };
    constexpr int NBLK = WM * WN, NIT = NA + NB;
    // (rn, rbn): register set of triplet b+1, stashed into LDS[buf ^ 1] here; (ri, rbi): the set the loads of triplet b+2 go to
    // -- the set triplet b came from, stashed during the previous step.  Returns the weight of triplet b+2.
    auto step = [&](f32x4 (&ri)[NA], f32x4 (&rbi)[NB], const f32x4 (&rn)[NA], const f32x4 (&rbn)[NB], Ix& ix, int bnext, int buf, float keep) __attribute__((always_inline)) -> float {
        const float* pa = lds_a + buf * a_elems + lk * KM_PA + wm0 + li;
        const float* pb = lds_b + buf * b_elems + lk * KM_PB + wn0 + li;
        f32x4 tt[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) tt[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float af[nk4][WM], bf[nk4][WN], vm[WN];
        read_frag(pa, pb, 0, af[0], bf[0]);
        read_frag(pa, pb, 1, af[1], bf[1]);
#pragma unroll
        for (int j = 0; j < WN; ++j) vm[j] = lds_b[buf * b_elems + K * KM_PB + wn0 + 16 * j + li];
        __builtin_amdgcn_sched_barrier(0);
        float knext = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < nk4; ++s4) {               // (one scheduling region per sub-step: reads cannot sink to their use)
            if (s4 + 2 < nk4) read_frag(pa, pb, s4 + 2, af[s4 + 2], bf[s4 + 2]);
            if (s4 == 0 && ABL != 2) knext = issue(ri, rbi, ix, bnext);
            if (s4 == 1 && ABL != 2) issue_idx(ix, bnext + 2);
            if (ABL == 2) knext = 1.f;
            // the previous triplet's fold and the next triplet's LDS stores, spread over the sub-steps (the stores from sub-step 1
            // on: their loads were issued a whole step ago)
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) if (blk * nk4 / NBLK == s4) fold_block(blk);
            if (ABL != 3 && s4 >= 1) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) if (it * (nk4 - 1) / NIT == s4 - 1) stash_item(rn, rbn, buf ^ 1, it);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) tt[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4][i], bf[s4][j], tt[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) tp[i][j] = tt[i][j];
#pragma unroll
        for (int j = 0; j < WN; ++j) vp[j] = vm[j] * keep;
        keep_p = keep;
        return knext;
    };

    Ix ix0, ix1;
    issue_idx(ix0, b0); issue_idx(ix1, b0 + 1);
    float k0 = issue(ra0, rb0, ix0, b0);
    issue_idx(ix0, b0 + 2);
    float k1 = issue(ra1, rb1, ix1, b0 + 1);
    issue_idx(ix1, b0 + 3);
    stash(ra0, rb0, 0);
    float kc = k0;                                       // weight of the triplet whose tile sits in LDS[0]
    __syncthreads();
    for (int b = b0; b < b1; b += 2) {
        // LDS[0] = triplet b (kc); set 1 = triplet b+1 (landing; stored to LDS[1] inside the step); ix0 / ix1 = indices of triplets b+2 / b+3
        k0 = step(ra0, rb0, ra1, rb1, ix0, b + 2, 0, kc);
        const float kn = k1;
        if (ABL != 1) __syncthreads();
        // LDS[1] = triplet b+1 (kn); set 0 = triplet b+2 (landing; stored to LDS[0] inside the step)
        k1 = step(ra1, rb1, ra0, rb0, ix1, b + 3, 1, kn);
        kc = k0;
        if (ABL != 1) __syncthreads();
    }
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) fold_block(blk);      // the last triplet's fold
    float* dk = slab + ((long long)z * 2 + 0) * H * dv;
    float* dm = slab + ((long long)z * 2 + 1) * H * dv;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + wm0 + 16 * i + 4 * lk + q;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + wn0 + 16 * j + li;
                if (m < H && n < dv) { dk[(long long)m * dv + n] = acc_k[i][j][q]; dm[(long long)m * dv + n] = acc_m[i][j][q]; }
            }
        }
}


// ---- the same pass as ONE 8-wave workgroup per CU on a 256 (all of H) x 64 tile pair, in the style of k_dw_tn8 (round 4) -------------------------
// What the balanced TN kernel taught (ncx_dwtn.hip): all of H per workgroup (a dpre row is fetched and stored to LDS once per 64 columns, not twice),
// the interleaved block mapping (one ds_read_b128 / ds_read_b64 feeds four / two MFMA blocks: 4 LDS reads per 16 MFMAs where k_dw_km issues one
// ds_read_b32 per operand), a rotated loop (the last sub-step's MFMAs after the barrier).  A reduction step is a 24-row triplet part = THREE 8-deep
// sub-steps; T_b accumulates from an inline-zero C operand into one of two scratch accumulators (by the step's parity) and is folded into both outputs
// under the MFMAs of the next step.  Same k-chunks and slab layout as k_dw_km (its reduction follows unchanged); the summation order inside a
// triplet differs from k_dw_km's, so the two kernels agree to fp32 rounding, not bitwise.
// (Tried: two reduction steps per LDS buffer = 96 MFMAs per wave between barriers, one register set re-requested item by item: 118 spilled
// registers, 0.556 ms for DW1C -- dropped.)
constexpr int KM8_PA = 256, KM8_PB = 80;                            // LDS pitches (floats), as k_dw_tn8
constexpr int KM8_LDS = 2 * (24 * KM8_PA + 32 * KM8_PB) * 4;

__global__ __launch_bounds__(512, 1) void k_dw_km8(const float* __restrict__ dpre, int H, const float* __restrict__ feats, int dv,
                                                   const int* __restrict__ idx_k, const int* __restrict__ idx_o, int B, int Kc,
                                                   int chunk, int tiles_m, int S, float* __restrict__ slab) {
    constexpr int BM = 256, BN = 64, K = 24, PA = KM8_PA, PB = KM8_PB, NA = 3;
    constexpr int A_EL = K * PA, B_EL = 32 * PB;
    extern __shared__ __attribute__((aligned(16))) float km8_smem[];
    float* const lds_a = km8_smem;                     // [2][24][PA]
    float* const lds_b = km8_smem + 2 * A_EL;          // [2][32][PB]   rows 0 .. 23: v_k of the step's rows; rows 24 .. 31: the triplet's v_o (eight copies)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    const int z = blockIdx.x % S, t = blockIdx.x / S;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int spt = Kc / K;
    const int t0 = z * chunk, t1 = min(t0 + chunk, B);
    if (t0 >= B) return;
    const int b0 = t0 * spt, b1 = t1 * spt;

    f32x4 acc_k[4][2], acc_m[4][2], tt[2][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc_k[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            tt[0][i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; tt[1][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int arow = tid >> 6, aq = tid & 63;
    const int brow = tid >> 4, bq = tid & 15;
    typedef const __attribute__((address_space(1))) int* gip;
    f32x4 va[2][NA], vb[2];
    auto load_idx = [&](int b) __attribute__((always_inline)) -> int {
        const long long r0 = (long long)min(b, b1 - 1) * K;
        return brow < K ? ((gip)idx_k)[r0 + brow] : ((gip)idx_o)[r0];
    };
    int ixn = load_idx(b0);                                        // gather row of the NEXT issue (requested one issue ahead)
    // Addresses as UNIFORM base (scalar registers, scalar arithmetic) + a 32-bit per-lane byte offset that never changes (dpre rows) or costs one
    // v_mad_u32_u24 (the gathered feature row: n_img x dv x 4 < 2^32): vector instructions are not hidden under fp32 MFMAs (DESIGN S5d), and the 64-bit
    // multiply-adds of `base + (r0 + row) * H` per load were a fifth of this loop's.
    typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dpre), 0, 0xFFFFFFF0u, 0x00020000);      // (buffer loads: no vector instruction per load)
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats), 0, 0xFFFFFFF0u, 0x00020000);
    unsigned offA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) offA[i] = (unsigned)(((arow + 8 * i) * H + m0 + 4 * aq) * 4);
    const unsigned offX = (unsigned)((n0 + 4 * bq) * 4), dv4 = (unsigned)dv * 4u;
    auto issue = [&](auto set_c, int b) __attribute__((always_inline)) {
        constexpr int SS_ = decltype(set_c)::value;
        const long long r0 = (long long)min(b, b1 - 1) * K;
        const unsigned ao = (unsigned)(r0 * H * 4);                      // uniform
        // (the index request goes out FIRST: it is the load the next issue needs soonest, and vmcnt counts in order -- as the youngest load of the
        // issue it made the next issue wait for every quad of this one)
        const int ix = ixn;
        ixn = load_idx(b + 1);
#pragma unroll
        for (int i = 0; i < NA; ++i) va[SS_][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsD, offA[i], ao, 0));
        vb[SS_] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsF, __umul24((unsigned)ix, dv4) + offX, 0, 0));
    };
    auto stash = [&](auto set_c, int buf, int h0, int h1) __attribute__((always_inline)) {
        constexpr int SS_ = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            *(f32x4*)(lds_a + buf * A_EL + (arow + 8 * i) * PA + 4 * aq) = va[SS_][i];
        }
        if (NA >= h0 && NA < h1) *(f32x4*)(lds_b + buf * B_EL + brow * PB + 4 * bq) = vb[SS_];
    };
    // fragment sets alternate with the step's parity (three sub-steps per step): F[PAR ^ 1] comes in holding sub-step 2 of the previous step
    f32x4 fa[2][2]; f32x2 fb[2][2];                    // [set][e]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) { fa[q][e] = f32x4{0.f, 0.f, 0.f, 0.f}; fb[q][e] = f32x2{0.f, 0.f}; }
    f32x2 vo[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // [parity of the step]: v_o of the lane's two columns
    // interleaved block mapping: block (i, j) of a wave owns tile rows wm0 + 4 r + i, columns wn0 + 2 c + j; k order: MFMA (s, e) takes k = 8 s + 2 lk + e
    auto read_frags = [&](int buf, int s, f32x4 (&af)[2], f32x2 (&bf)[2]) __attribute__((always_inline)) {
        const int kk = 8 * s + 2 * lk;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            af[e] = *(const f32x4*)(lds_a + buf * A_EL + (kk + e) * PA + wm0 + 4 * li);
            bf[e] = *(const f32x2*)(lds_b + buf * B_EL + (kk + e) * PB + wn0 + 2 * li);
        }
    };
    auto mfma = [&](const f32x4 (&af)[2], const f32x2 (&bf)[2], f32x4 (&c)[4][2], auto first_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;           // the step's first sub-step: e = 0 starts from zero
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e][i], bf[e][j], (FIRST && e == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : c[i][j], 0, 0, 0);
    };
    auto fold = [&](int i0, int i1, const f32x4 (&c)[4][2], const f32x2& v) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < i0 || i >= i1) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {                             // (packed: 2 v_pk_add_f32 + 2 v_pk_fma_f32 per block instead of 8 scalar operations)
                acc_k[i][j] += c[i][j];
                acc_m[i][j] = __builtin_elementwise_fma(c[i][j], f32x4{v[j], v[j], v[j], v[j]}, acc_m[i][j]);
            }
        }
    };
    auto pin = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    typedef std::true_type Tt; typedef std::false_type Ff;
    issue(S0{}, b0);
    issue(S1{}, b0 + 1);
    stash(S0{}, 0, 0, NA + 1);
    issue(S0{}, b0 + 2);
    __syncthreads();
    auto step = [&](auto par_c, int b) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef std::integral_constant<int, PAR ^ 1> SS;
        // sub-step 2 of the previous step (zeros at the start) completes its T over the first reads of this buffer
        read_frags(PAR, 0, fa[PAR], fb[PAR]);
        vo[PAR] = *(const f32x2*)(lds_b + PAR * B_EL + K * PB + wn0 + 2 * li);
        mfma(fa[PAR ^ 1], fb[PAR ^ 1], tt[PAR ^ 1], Ff{});
#pragma unroll
        for (int q = 0; q < 16; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(PAR, 1, fa[PAR ^ 1], fb[PAR ^ 1]);
        stash(SS{}, PAR ^ 1, 0, 2);
        mfma(fa[PAR], fb[PAR], tt[PAR], Tt{});
        fold(0, 2, tt[PAR ^ 1], vo[PAR ^ 1]);
        pin();
        read_frags(PAR, 2, fa[PAR], fb[PAR]);
        stash(SS{}, PAR ^ 1, 2, NA + 1);
        issue(SS{}, b + 3);
        mfma(fa[PAR ^ 1], fb[PAR ^ 1], tt[PAR], Ff{});
        fold(2, 4, tt[PAR ^ 1], vo[PAR ^ 1]);
        pin();
        __syncthreads();
    };
    int b = b0;
    for (; b + 1 < b1; b += 2) { step(S0{}, b); step(S1{}, b + 1); }
    if (b < b1) { step(S0{}, b); mfma(fa[0], fb[0], tt[0], Ff{}); fold(0, 4, tt[0], vo[0]); }
    else { mfma(fa[1], fb[1], tt[1], Ff{}); fold(0, 4, tt[1], vo[1]); }
    float* dk = slab + ((long long)z * 2 + 0) * H * dv;
    float* dm = slab + ((long long)z * 2 + 1) * H * dv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const long long m = m0 + wm0 + 4 * (4 * lk + rg) + i;
            const int n = n0 + wn0 + 2 * li;
            *(f32x2*)(dk + m * dv + n) = f32x2{acc_k[i][0][rg], acc_k[i][1][rg]};
            *(f32x2*)(dm + m * dv + n) = f32x2{acc_m[i][0][rg], acc_m[i][1][rg]};
        }
}

// ---- the same pass on the bf16 matrix path with fp32-grade operands (NCX_F_X6; not the default) -------------------------------------------
// Operands as in k_dw_tn8_x6 (ncx_dwtn.hip): every fp32 element is cut into three bf16 values by truncation when it is stored to LDS
// (x = x1 + x2 + x3 exactly), six products per block on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  One 8-wave workgroup per CU on a
// 256 (all of H) x 64 tile pair; a reduction step is one triplet part of 24 rows in a 32-deep MFMA step (LDS rows 24 .. 31 of the dpre planes
// are zeroed once and never written: their products vanish whatever the feature planes hold there).  T_b of a 16-row block row comes out of
// its 12 MFMAs into a scratch accumulator that starts at zero, and is folded into both outputs (acc_k += T_b, acc_m += v_o (.) T_b) under the
// MFMAs of the NEXT block row; the last block row's MFMAs of a step run after the barrier, over the first fragment reads of the next step.
// v_o travels in fp32 (a 64-float row per LDS buffer).  Same k-chunks and slab layout as k_dw_km: the reduction that follows is unchanged.
typedef __bf16 km_bf16x8 __attribute__((ext_vector_type(8)));
typedef short km_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int km_u32x2 __attribute__((ext_vector_type(2)));
constexpr int KM6_PA = 544, KM6_PB = 160;                                    // bytes per row of one plane (512 + 32, 128 + 32)
constexpr int KM6_PL = 32 * (KM6_PA + KM6_PB), KM6_BUF = 3 * KM6_PL;
constexpr int KM6_LDS = 2 * KM6_BUF + 2 * 64 * 4 + 64 * 16;                  // planes, v_o rows, a dump row for the stores of threads without a v_o quad

__global__ __launch_bounds__(512, 1) void k_dw_km_x6(const float* __restrict__ dpre, int H, const float* __restrict__ feats, int dv,
                                                     const int* __restrict__ idx_k, const int* __restrict__ idx_o, int B, int Kc,
                                                     int chunk, int tiles_m, int S, float* __restrict__ slab) {
    constexpr int T = 512, BM = 256, BN = 64, K = 24, PA = KM6_PA, PB = KM6_PB, PL = KM6_PL, BUF = KM6_BUF, A_PL = 32 * PA;
    constexpr int NA = 3;                                          // dpre quads per thread and step: rows arow + 8 i
    extern __shared__ __attribute__((aligned(1024))) unsigned char km6_smem[];
    float* const lds_vo = (float*)(km6_smem + 2 * BUF);           // [2][64]
    float* const lds_dump = lds_vo + 2 * 64;                      // [64][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    const int z = blockIdx.x % S, t = blockIdx.x / S;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int spt = Kc / K;
    const int t0 = z * chunk, t1 = min(t0 + chunk, B);
    if (t0 >= B) return;
    const int b0 = t0 * spt, b1 = t1 * spt;
    for (int i = tid; i < 2 * BUF / 16; i += T) ((f32x4*)km6_smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    f32x4 acc_k[4][2], acc_m[4][2], tt[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc_k[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) tt[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int arow = tid >> 6, aq = tid & 63;
    const int brow = tid >> 4, bq = tid & 15;                      // rows 0 .. 23: v_k of the step's rows; rows 24 .. 31: the triplet's v_o (same quad, same value)
    typedef const __attribute__((address_space(1))) int* gip;
    f32x4 va[2][NA], vb[2];
    auto load_idx = [&](int b) __attribute__((always_inline)) -> int {
        const long long r0 = (long long)min(b, b1 - 1) * K;
        return brow < K ? ((gip)idx_k)[r0 + brow] : ((gip)idx_o)[r0];
    };
    int ixn = load_idx(b0);                                        // gather row of the NEXT issue (requested one issue ahead)
    // (buffer loads: descriptor + 32-bit lane offset + scalar step offset, as k_dw_km8)
    typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dpre), 0, 0xFFFFFFF0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats), 0, 0xFFFFFFF0u, 0x00020000);
    unsigned offA[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) offA[i] = (unsigned)(((arow + 8 * i) * H + m0 + 4 * aq) * 4);
    const unsigned offX = (unsigned)((n0 + 4 * bq) * 4), dv4 = (unsigned)dv * 4u;
    auto issue = [&](auto set_c, int b) __attribute__((always_inline)) {
        constexpr int SS_ = decltype(set_c)::value;
        const long long r0 = (long long)min(b, b1 - 1) * K;
        const unsigned ao = (unsigned)(r0 * H * 4);                      // uniform
        const int ix = ixn;                                              // (the index request first: vmcnt counts in order)
        ixn = load_idx(b + 1);
#pragma unroll
        for (int i = 0; i < NA; ++i) va[SS_][i] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsD, offA[i], ao, 0));
        vb[SS_] = __builtin_bit_cast(f32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(rsF, __umul24((unsigned)ix, dv4) + offX, 0, 0));
    };
    auto split_store = [&](f32x4 v, unsigned char* base) __attribute__((always_inline)) {
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xj = v[j];               // (a scalar copy first: __builtin_bit_cast applied to the vector element itself reads element 0 -- hipcc 7.2)
            p1[j] = __builtin_bit_cast(unsigned, xj) & 0xFFFF0000u;
            const float r1 = xj - __builtin_bit_cast(float, p1[j]);
            p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
            p3[j] = __builtin_bit_cast(unsigned, r2);
        }
        const km_u32x2 w1 = {__builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const km_u32x2 w2 = {__builtin_amdgcn_perm(p2[1], p2[0], 0x07060302u), __builtin_amdgcn_perm(p2[3], p2[2], 0x07060302u)};
        const km_u32x2 w3 = {__builtin_amdgcn_perm(p3[1], p3[0], 0x07060302u), __builtin_amdgcn_perm(p3[3], p3[2], 0x07060302u)};
        *(km_u32x2*)(base) = w1; *(km_u32x2*)(base + PL) = w2; *(km_u32x2*)(base + 2 * PL) = w3;
    };
    // the fp32 v_o quad goes to the buffer's v_o row from the threads that hold one (rows >= 24; eight copies of the same value), to the dump row from the others: no branch
    float* const vo_dst0 = brow >= K ? lds_vo + 4 * bq : lds_dump + 4 * lane;
    const int vo_step = brow >= K ? 64 : 0;
    auto stash = [&](auto set_c, int buf, int h0, int h1) __attribute__((always_inline)) {
        constexpr int SS_ = decltype(set_c)::value;
        unsigned char* const base = km6_smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            split_store(va[SS_][i], base + (arow + 8 * i) * PA + aq * 8);
        }
        if (NA >= h0 && NA < h1) {
            split_store(vb[SS_], base + A_PL + brow * PB + bq * 8);
            *(f32x4*)(vo_dst0 + buf * vo_step) = vb[SS_];
        }
    };
    typedef __attribute__((address_space(3))) km_s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;
    const int fragA = (4 * lk + tq) * PA + (wm0 + 4 * tp) * 2, fragB = A_PL + (4 * lk + tq) * PB + (wn0 + 4 * tp) * 2;
    auto read_a = [&](int buf, int i, km_bf16x8 (&af)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned char* ta = km6_smem + buf * BUF + p * PL + fragA + i * 32;
            const km_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta));
            const km_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta + 16 * PA));
            af[p] = __builtin_bit_cast(km_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto read_b = [&](int buf, km_bf16x8 (&bf)[3][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char* tb = km6_smem + buf * BUF + p * PL + fragB + j * 32;
                const km_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb));
                const km_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + 16 * PB));
                bf[p][j] = __builtin_bit_cast(km_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
    };
    // T of one block row: 12 MFMAs from zero, the two column blocks alternate, small terms first
    auto mfma12z = [&](const km_bf16x8 (&af)[3], const km_bf16x8 (&bf)[3][2], f32x4 (&c)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2][j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[1][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[0][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[1][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[0][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[0][j], c[j], 0, 0, 0);
    };
    auto fold = [&](int i, const f32x4 (&c)[2], const float (&vo)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc_k[i][j][q] += c[j][q];
                acc_m[i][j][q] = __builtin_fmaf(c[j][q], vo[j], acc_m[i][j][q]);
            }
    };
    auto pin = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1;
    km_bf16x8 afA[3], afB[3], bfr[2][3][2];
    float vo[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                      // [parity of the step][column block]: v_o of the lane's columns
    const km_bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        afB[p] = zero8;
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[1][p][j] = zero8;
    }
    issue(S0{}, b0);
    issue(S1{}, b0 + 1);
    stash(S0{}, 0, 0, NA + 1);
    issue(S0{}, b0 + 2);
    __syncthreads();
    auto step = [&](auto par_c, int b) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef std::integral_constant<int, PAR ^ 1> SS;
        // block row 3 of the previous step (zeros at the start) over the first reads of this buffer; the fold of its block row 2
        read_b(PAR, bfr[PAR]);
        read_a(PAR, 0, afA);
#pragma unroll
        for (int j = 0; j < 2; ++j) vo[PAR][j] = lds_vo[PAR * 64 + wn0 + 16 * j + li];
        mfma12z(afB, bfr[PAR ^ 1], tt[1]);
        fold(2, tt[0], vo[PAR ^ 1]);
#pragma unroll
        for (int q = 0; q < 12; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
        __builtin_amdgcn_sched_barrier(0);
        read_a(PAR, 1, afB); stash(SS{}, PAR ^ 1, 0, 2); mfma12z(afA, bfr[PAR], tt[0]); fold(3, tt[1], vo[PAR ^ 1]); pin();
        read_a(PAR, 2, afA); stash(SS{}, PAR ^ 1, 2, NA + 1); mfma12z(afB, bfr[PAR], tt[1]); fold(0, tt[0], vo[PAR]); pin();
        read_a(PAR, 3, afB); issue(SS{}, b + 3); mfma12z(afA, bfr[PAR], tt[0]); fold(1, tt[1], vo[PAR]); pin();
        __syncthreads();
    };
    int b = b0;
    for (; b + 1 < b1; b += 2) { step(S0{}, b); step(S1{}, b + 1); }
    if (b < b1) { step(S0{}, b); mfma12z(afB, bfr[0], tt[1]); fold(2, tt[0], vo[0]); fold(3, tt[1], vo[0]); }
    else { mfma12z(afB, bfr[1], tt[1]); fold(2, tt[0], vo[1]); fold(3, tt[1], vo[1]); }
    float* dk = slab + ((long long)z * 2 + 0) * H * dv;
    float* dm = slab + ((long long)z * 2 + 1) * H * dv;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long m = m0 + wm0 + 16 * i + 4 * lk + q;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn0 + 16 * j + li;
                dk[m * dv + n] = acc_k[i][j][q]; dm[m * dv + n] = acc_m[i][j][q];
            }
        }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_dw_km_reduce(const KmReduceArgs r) { km_reduce_body<VEC>(r, blockIdx.x); }
// ... and, in the same launch, the split fix-up of the grouped GEMM that computes the other columns of linear_1's gradient
// (two ~7-12 us reductions whose cost is mostly launch and ramp)
template <bool VEC, int BM, int BN>
__global__ __launch_bounds__(256) void k_dw_km_reduce_fixup(const KmReduceArgs r, const FixupArgs f) {
    constexpr int NS = BM * BN / 1024;
    if ((int)blockIdx.x < r.nblk) { km_reduce_body<VEC>(r, blockIdx.x); return; }
    const int id = blockIdx.x - r.nblk;
    split_fixup_body<BM, BN>(f, id / NS, id % NS);
}

bool dw_km_supported(const ncx_dims& d) {
    // candidates per triplet a multiple of 24 (the reference's knn_size 24, configs[4]'s 48: two 24-row steps per triplet);
    // rows need 4 columns for the 16-byte windows
    // ... and at least 256 triplets: below that the step is a latency chain and the grouped GEMM path measured faster
    // (B = 32 / 64 / 128: 0.489 / 0.493 / 0.553 ms against 0.52 / 0.53 / 0.58 with this kernel)
    return (d.flags & NCX_F_V_MULT) && d.K % 24 == 0 && d.H >= 4 && d.dv >= 4 && (d.B >= 256 || hook_env("NCX_KM_FORCE"));
}
size_t dw_km_slab_bytes(const ncx_dims& d) { return dw_km_supported(d) ? (size_t)DW_KM_SPLIT * 2 * d.H * d.dv * 4 : 0; }

static int km_chunks(const ncx_dims& d, int& chunk) {
    // up to 8 k-chunks (one per XCD), at least 16 triplets each: tiny batches are not worth 8 partial tiles
    const int S = d.B / 16 >= DW_KM_SPLIT ? DW_KM_SPLIT : (d.B / 16 >= 1 ? d.B / 16 : 1);
    chunk = (d.B + S - 1) / S;
    return S;
}

KmReduceArgs dw_km_reduce_args(const ncx_dims& d, const float* slab, float* g_vother, float* g_vmult, long long din, bool* vec) {
    int chunk; km_chunks(d, chunk);
    KmReduceArgs r{};
    r.slab = slab; r.nz = (d.B + chunk - 1) / chunk; r.H = d.H; r.dv = d.dv; r.din = din; r.g_vother = g_vother; r.g_vmult = g_vmult;
    const long long n = (long long)d.H * d.dv;
    *vec = d.dv % 4 == 0 && ((uintptr_t)slab & 15) == 0;
    r.nblk = (int)(((*vec ? n / 4 : n) + 255) / 256);
    return r;
}

// The fixed-order sum of the k-chunk partials; `fix` (may be null / invalid): a deferred split fix-up of tile shape `fix_cfg` that
// rides in the same launch.
int dw_km_finish(const ncx_dims& d, const float* slab, float* g_vother, float* g_vmult, long long din, const FixupArgs* fix, int fix_cfg,
                 hipStream_t s) {
    int chunk; km_chunks(d, chunk);
    KmReduceArgs r{};
    r.slab = slab; r.nz = (d.B + chunk - 1) / chunk; r.H = d.H; r.dv = d.dv; r.din = din; r.g_vother = g_vother; r.g_vmult = g_vmult;
    const long long n = (long long)d.H * d.dv;
    const bool vec = d.dv % 4 == 0 && ((uintptr_t)slab & 15) == 0;      // 16-byte loads of the partials
    r.nblk = (int)(((vec ? n / 4 : n) + 255) / 256);
    const bool with_fix = fix && fix->valid && (fix_cfg == CFG_128x64 || fix_cfg == CFG_64x64);
    if (!with_fix) {
        if (vec) hipLaunchKernelGGL(k_dw_km_reduce<true>, dim3(r.nblk), dim3(256), 0, s, r);
        else     hipLaunchKernelGGL(k_dw_km_reduce<false>, dim3(r.nblk), dim3(256), 0, s, r);
        NCX_HIP_TRY(hipGetLastError());
        if (fix && fix->valid) return run_fixup2(*fix, FixupArgs{}, fix_cfg, s);
        return NCX_OK;
    }
    const int nfix = fix->grid_x * (fix_cfg == CFG_128x64 ? 8 : 4);
    if (fix_cfg == CFG_128x64) {
        if (vec) hipLaunchKernelGGL((k_dw_km_reduce_fixup<true, 128, 64>), dim3(r.nblk + nfix), dim3(256), 0, s, r, *fix);
        else     hipLaunchKernelGGL((k_dw_km_reduce_fixup<false, 128, 64>), dim3(r.nblk + nfix), dim3(256), 0, s, r, *fix);
    } else {
        if (vec) hipLaunchKernelGGL((k_dw_km_reduce_fixup<true, 64, 64>), dim3(r.nblk + nfix), dim3(256), 0, s, r, *fix);
        else     hipLaunchKernelGGL((k_dw_km_reduce_fixup<false, 64, 64>), dim3(r.nblk + nfix), dim3(256), 0, s, r, *fix);
    }
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

// finish = false: the caller sums the partials later with dw_km_finish (merged with the grouped GEMM's fix-up)
int dw_km(const ncx_dims& d, const float* dpre, const float* feats, const int* idx_k, const int* idx_o, float* slab,
          float* g_vother, float* g_vmult, long long din, hipStream_t s, bool finish) {
    int chunk;
    const int S = km_chunks(d, chunk);
    constexpr int R = 24;
    // the 8-wave 256 x 64 form where the shape has whole tiles (configs[1] on one box: DW1C 0.2962-0.2970 ms against 0.3048, step 0.8465-0.8478 ms against 0.8578)
    const bool x6 = (d.flags & NCX_F_X6) && !hook_env("NCX_NO_X6") && !hook_env("NCX_NO_KM_X6");
    const bool fits32 = (long long)d.n_img * d.dv * 4 < (1ll << 32) - 65536 && (long long)d.B * d.K * d.H * 4 < (1ll << 32) - 65536;      // (k_dw_km8's buffer loads: 32-bit byte offsets)
    if (d.H % 256 == 0 && d.dv % 64 == 0 && fits32 && !x6 && !hook_env("NCX_NO_KM8") && !hook_env("NCX_KM_BM") && !hook_env("NCX_KM_T") && !hook_env("NCX_KM_ABL")) {
        static DevMask attr8{0};
        NCX_HIP_TRY(set_max_lds_once(attr8, (const void*)k_dw_km8, KM8_LDS));
        hipLaunchKernelGGL(k_dw_km8, dim3((d.H / 256) * (d.dv / 64) * S), dim3(512), KM8_LDS, s, dpre, d.H, feats, d.dv, idx_k, idx_o, d.B, d.K, chunk, d.H / 256, S, slab);
        NCX_HIP_TRY(hipGetLastError());
        return finish ? dw_km_finish(d, slab, g_vother, g_vmult, din, nullptr, 0, s) : NCX_OK;
    }
    if (x6 && d.H % 256 == 0 && d.dv % 64 == 0 && fits32) {      // (K % 24 == 0: dw_km_supported)
        static DevMask attr6{0};
        NCX_HIP_TRY(set_max_lds_once(attr6, (const void*)k_dw_km_x6, KM6_LDS));
        hipLaunchKernelGGL(k_dw_km_x6, dim3((d.H / 256) * (d.dv / 64) * S), dim3(512), KM6_LDS, s, dpre, d.H, feats, d.dv, idx_k, idx_o, d.B, d.K, chunk, d.H / 256, S, slab);
        NCX_HIP_TRY(hipGetLastError());
        return finish ? dw_km_finish(d, slab, g_vother, g_vmult, din, nullptr, 0, s) : NCX_OK;
    }
    // 128-row tiles at two workgroups per CU, or 64-row tiles at four (hook NCX_KM_BM=64)
    int bm = 128;
    if (const char* e = hook_env("NCX_KM_BM")) bm = atoi(e) == 64 ? 64 : 128;
    // 128 x 128 tile pairs as ONE 8-wave workgroup per CU (what gave the forward kernel 3 %): measured here 124.8 us against 123.7 for
    // two 4-wave workgroups per CU (DW1C 0.3231 / 0.3233 against 0.3215 / 0.3220 ms) -- kept behind the hook NCX_KM_T=512 only
    bool wide = false;
    if (const char* e = hook_env("NCX_KM_T")) wide = atoi(e) == 512 && bm == 128;
    const int bn = wide ? 128 : 64;
    const int tiles_m = (d.H + bm - 1) / bm, tiles_n = (d.dv + bn - 1) / bn;
    const bool edge = d.H % bm != 0 || d.dv % bn != 0;
    auto go = [&](auto bm_c, auto occ_c, auto edge_c) -> int {
        constexpr int BM = decltype(bm_c)::value, OCC = decltype(occ_c)::value;
        constexpr bool EDGE = decltype(edge_c)::value;
        constexpr int QA = BM / 4, NA = (R * QA + 255) / 256, KA = NA * 256 / QA, KB = (R * 16 + 255) / 256 * 16;
        const int lds = 2 * (KA * (BM + 16) + KB * (64 + 16)) * 4;
        static DevMask attr{0};
        NCX_HIP_TRY(set_max_lds_once(attr, (const void*)k_dw_km<R, BM, OCC, EDGE>, lds));
        hipLaunchKernelGGL((k_dw_km<R, BM, OCC, EDGE>), dim3(tiles_m * tiles_n * S), dim3(256), lds, s, dpre, d.H, feats, d.dv, idx_k, idx_o, d.B, d.K,
                           chunk, tiles_m, S, slab);
        return (int)hipGetLastError();
    };
    auto go8 = [&](auto edge_c) -> int {                   // 512 threads, 128 x 128 tile pair
        constexpr bool EDGE = decltype(edge_c)::value;
        constexpr int T = 512, QA = 32, QB = 32, NA = (R * QA + T - 1) / T, NB = (R * QB + T - 1) / T, KA = NA * T / QA, KB = NB * T / QB;
        const int lds = 2 * (KA * (128 + 16) + KB * (128 + 16)) * 4;
        static DevMask attr{0};
        NCX_HIP_TRY(set_max_lds_once(attr, (const void*)k_dw_km<R, 128, 2, EDGE, 0, T>, lds));
        hipLaunchKernelGGL((k_dw_km<R, 128, 2, EDGE, 0, T>), dim3(tiles_m * tiles_n * S), dim3(T), lds, s, dpre, d.H, feats, d.dv, idx_k, idx_o, d.B, d.K,
                           chunk, tiles_m, S, slab);
        return (int)hipGetLastError();
    };
    typedef std::integral_constant<int, 128> B128; typedef std::integral_constant<int, 64> B64;
    typedef std::integral_constant<int, 2> O2; typedef std::integral_constant<int, 4> O4;
    int rc;
    int abl = 0;
    if (const char* e = hook_env("NCX_KM_ABL")) abl = atoi(e);
    if (wide) rc = edge ? go8(std::true_type{}) : go8(std::false_type{});
    else if (abl >= 1 && abl <= 4 && !edge && bm == 128) {        // timing ablations (results are wrong): 1 no barrier, 2 no global loads, 3 no LDS stores, 4 no fold
        const int lds = 2 * (24 * (128 + 16) + 32 * (64 + 16)) * 4;
        auto run = [&](auto abl_c) -> int {
            constexpr int A = decltype(abl_c)::value;
            NCX_HIP_TRY(hipFuncSetAttribute((const void*)k_dw_km<R, 128, 2, false, A>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL((k_dw_km<R, 128, 2, false, A>), dim3(tiles_m * tiles_n * S), dim3(256), lds, s, dpre, d.H, feats, d.dv, idx_k, idx_o, d.B, d.K,
                               chunk, tiles_m, S, slab);
            return (int)hipGetLastError();
        };
        rc = abl == 1 ? run(std::integral_constant<int, 1>{}) : abl == 2 ? run(std::integral_constant<int, 2>{})
           : abl == 3 ? run(std::integral_constant<int, 3>{}) : run(std::integral_constant<int, 4>{});
    } else
    if (bm == 128) rc = edge ? go(B128{}, O2{}, std::true_type{}) : go(B128{}, O2{}, std::false_type{});
    else           rc = edge ? go(B64{}, O4{}, std::true_type{}) : go(B64{}, O4{}, std::false_type{});
    if (rc) return rc;
    return finish ? dw_km_finish(d, slab, g_vother, g_vmult, din, nullptr, 0, s) : NCX_OK;
}

}  // namespace ncx
