// mb_tn.hip -- structure microbenchmark for the row-reduction (TN) weight-gradient products of linear_1 that are NOT the per-triplet
// fold (ncx_dwkm.hip):   C[h][n] = sum_r D[r][h] * x(r, n),   D = dpre [M][256],  x = softmax(a_knns) [M][2000] | z_knns [M][360] | ...
// Not part of the library: explores the 8-wave 256-row workgroup (ONE per CU, all of H in one tile: every operand row is staged and
// transformed once) against the generic engine's 128 x 64 / 4-wave / two-per-CU instantiation (measured in the library: dGt alone
// 125 us = 0.64 of the fp32-MFMA peak) before it goes into csrc/.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mb/mb_tn.hip -o tools/mb/mb_tn
//   run:   tools/mb/mb_tn [M=12288] [N=2000] [reps=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef const __attribute__((address_space(1))) float* gfptr;
typedef const __attribute__((address_space(1))) f32x4u* gf4ptr;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int V> struct IntC { static constexpr int value = V; };

struct TnArgs {
    const float* D; int H;            // [M][H], H == 256 == the tile's rows
    const float* X; long long ldx;    // operand rows
    const float* lse;                 // SOFTMAX: base-2 log-sum-exp per row
    int M, N;
    int tiles_n, S, n_hi;             // the first n_hi column tiles are cut into S + 1 row chunks, the others into S
    float* slab;                      // [workgroup][256][BN]
    unsigned long long* stamps;
};

// 512 threads = 8 waves as 4 (rows) x 2 (columns); tile 256 x BN; k-step = 32 operand rows; double-buffered LDS, two register sets of
// global loads in flight (a load has 1.5 k-steps to land), LDS stores of the next tile and the loads of the one after spread under the
// MFMAs of the current one; ONE basic block per k-step; the last sub-step's MFMAs are issued after the barrier (rotated loop).
template <int BN, bool SOFTMAX, int ABL = 0>
__global__ __launch_bounds__(512, 1) void k_tn8(const TnArgs a) {
    constexpr int T = 512, BM = 256, BK = 32;
    constexpr int PA = 256, PB = BN == 128 ? 128 : 80;
    constexpr int WM = 4, WN = BN / 32;                       // 16 x 16 blocks per wave: 64 rows x BN / 2 columns
    constexpr int NA = BK * (BM / 4) / T, NB = BK * (BN / 4) / T;       // float4 per thread and k-step: 4, and 2 (BN 128) or 1 (BN 64)
    constexpr int QB = BN / 4;
    constexpr int NMF = 2 * WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const lds_a = smem;                                 // [2][32][PA]
    float* const lds_b = smem + 2 * BK * PA;                   // [2][32][PB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = (BN / 2) * (wave & 1);
    // work decode
    int id = blockIdx.x, tile, z, nz;
    if (id < a.n_hi * (a.S + 1)) { tile = id / (a.S + 1); z = id - tile * (a.S + 1); nz = a.S + 1; }
    else { id -= a.n_hi * (a.S + 1); tile = a.n_hi + id / a.S; z = id - (tile - a.n_hi) * a.S; nz = a.S; }
    const int n0 = tile * BN;
    const int total_steps = (a.M + BK - 1) / BK;
    const int g0 = (int)((long long)total_steps * z / nz), g1 = (int)((long long)total_steps * (z + 1) / nz);
    unsigned long long* const stamps = a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
    if (stamps && tid == 0) { stamps[0] = __builtin_readcyclecounter(); stamps[6] = __builtin_amdgcn_s_memrealtime(); }

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // loader roles (fixed columns per thread: the column mask is a constant of the thread)
    const int arow = tid >> 6, aq = tid & 63;                   // A item i: tile row arow + 8 i, quad aq
    const int brow = tid / QB, bq = tid % QB;                   // X item i: tile row brow + (T / QB) i, quad bq
    const int bc = n0 + 4 * bq;
    const float bkeep = bc < a.N ? 1.f : 0.f;                   // N % 4 == 0: a quad is inside or outside
    const int bcc = min(bc, a.N - 4);
    f32x4 va[2][NA], vb[2][NB];
    float vl[2][NB];
    auto issue = [&](auto set_c, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if (ABL == 2) return;
        const int r0 = min(t, g1 - 1) * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = min(r0 + arow + 8 * i, a.M - 1);
            va[S][i] = *(gf4ptr)((gfptr)a.D + (long long)r * a.H + 4 * aq);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = min(r0 + brow + (T / QB) * i, a.M - 1);
            vb[S][i] = *(gf4ptr)((gfptr)a.X + (long long)r * a.ldx + bcc);
            if (SOFTMAX) vl[S][i] = ((gfptr)a.lse)[r];
        }
    };
    auto stash = [&](auto set_c, int buf, int h0, int h1) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if (ABL == 3) return;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            *(f32x4*)(lds_a + buf * BK * PA + (arow + 8 * i) * PA + 4 * aq) = va[S][i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i + NA < h0 || i + NA >= h1) continue;
            f32x4 v = vb[S][i];
            if (SOFTMAX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -vl[S][i]));
            }
            v = v * bkeep;
            *(f32x4*)(lds_b + buf * BK * PB + (brow + (T / QB) * i) * PB + 4 * bq) = v;
        }
    };
    f32x4 afA[2], afB[2];                 // [e]: 4 interleaved A blocks per 16-byte read
    f32x4 bfA[2], bfB[2];                 // WN == 4: 16 bytes; WN == 2: the low 8 bytes
#pragma unroll
    for (int e = 0; e < 2; ++e) { afB[e] = f32x4{0.f, 0.f, 0.f, 0.f}; bfB[e] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto read_frags = [&](int buf, int s, f32x4 (&af)[2], f32x4 (&bf)[2]) __attribute__((always_inline)) {
        const int kk = 8 * s + 2 * lk;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            af[e] = *(const f32x4*)(lds_a + buf * BK * PA + (kk + e) * PA + wm0 + 4 * li);
            if (WN == 4) bf[e] = *(const f32x4*)(lds_b + buf * BK * PB + (kk + e) * PB + wn0 + 4 * li);
            else { const f32x2 v = *(const f32x2*)(lds_b + buf * BK * PB + (kk + e) * PB + wn0 + 2 * li); bf[e][0] = v[0]; bf[e][1] = v[1]; }
        }
    };
    auto mfma = [&](const f32x4 (&af)[2], const f32x4 (&bf)[2]) __attribute__((always_inline)) {
        if (ABL == 1) return;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e][i], bf[e][j], acc[i][j], 0, 0, 0);
    };
    typedef IntC<0> S0; typedef IntC<1> S1;
    issue(S0{}, g0);
    issue(S1{}, g0 + 1);
    stash(S0{}, 0, 0, NA + NB);
    issue(S0{}, g0 + 2);
    __syncthreads();
    auto step = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef IntC<PAR ^ 1> SS;
        read_frags(PAR, 0, afA, bfA);
        mfma(afB, bfB);                                  // (t - 1, last sub-step): covers the reads above
#pragma unroll
        for (int q = 0; q < NMF; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            auto& afc = (s & 1) ? afB : afA; auto& bfc = (s & 1) ? bfB : bfA;
            auto& afn = (s & 1) ? afA : afB; auto& bfn = (s & 1) ? bfA : bfB;
            read_frags(PAR, s + 1, afn, bfn);
            if (s == 0) stash(SS{}, PAR ^ 1, 0, (NA + NB) / 2);
            if (s == 1) stash(SS{}, PAR ^ 1, (NA + NB) / 2, NA + NB);
            if (s == 2) issue(SS{}, t + 3);
            mfma(afc, bfc);
#pragma unroll
            for (int q = 0; q < NMF; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    int t = g0;
    for (; t + 1 < g1; t += 2) { step(IntC<0>{}, t); step(IntC<1>{}, t + 1); }
    if (t < g1) step(IntC<0>{}, t);
    mfma(afB, bfB);                                      // the last sub-step
    if (stamps && tid == 0) stamps[1] = __builtin_readcyclecounter();
    // partial tile -> slab slot [256][BN]: lane holds, per (block q, reg), WN consecutive columns
    float* const slot = a.slab + (long long)blockIdx.x * (BM * BN);
#pragma unroll
    for (int q = 0; q < WM; ++q)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = wm0 + 4 * (4 * lk + rg) + q;
            if (WN == 4) { const f32x4 v = {acc[q][0][rg], acc[q][1][rg], acc[q][2][rg], acc[q][3][rg]}; *(f32x4*)(slot + row * BN + wn0 + 4 * li) = v; }
            else         { const f32x2 v = {acc[q][0][rg], acc[q][1][rg]}; *(f32x2*)(slot + row * BN + wn0 + 2 * li) = v; }
        }
    if (stamps && tid == 0) { stamps[2] = __builtin_readcyclecounter(); stamps[7] = __builtin_amdgcn_s_memrealtime(); }
}

// out[h][n] = sum over the chunks of tile n / BN (fixed order)
template <int BN>
__global__ void k_reduce(const float* slab, int H, int N, int S, int n_hi, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * N) return;
    const int h = (int)(i / N), n = (int)(i - (long long)h * N);
    const int tile = n / BN, c = n - tile * BN;
    const int nz = tile < n_hi ? S + 1 : S;
    const long long w0 = tile < n_hi ? (long long)tile * (S + 1) : (long long)n_hi * (S + 1) + (long long)(tile - n_hi) * S;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += slab[(w0 + z) * (256LL * BN) + (long long)h * BN + c];
    out[i] = s;
}


// ---- the same tile with LDS-DMA staging (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) ---------------------------
// Rings of THREE k-step slots per operand: tile t+2 is requested when step t starts, into the slot step t-1 read (everybody is past
// the barrier that closed step t-1), so ONE counted wait + raw barrier per step suffices and every load has two whole steps to land.
//   A (dpre rows, plain copy):   slot [32][256] floats, linear: one wave-instruction = one 1 KB row
//   X (operand rows):            slot [32][64] floats; 16-byte chunk c of row r sits at chunk c ^ 8 ((r >> 1) & 1) -- swizzled through the
//                                SOURCE address, so the two lane groups of a 32-lane half (rows two apart) read disjoint bank halves
//   lse (SOFTMAX):               every wave keeps its own copy of the step's 32 row statistics (one global_load_lds_dword per wave: no
//                                wave-dependent branch in the loop); the softmax is formed on the way from LDS to the matrix core:
//                                x = exp2(x log2e - lse[r]) -- 16 fma + 16 exp per lane and step under 64 MFMAs
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <bool SOFTMAX, int ABL = 0, bool NT = true>
__global__ __launch_bounds__(512, 1) void k_tn8d(const TnArgs a) {
    constexpr int BN = 64, BM = 256, BK = 32, WM = 4, WN = 2, NMF = 2 * WM * WN;
    constexpr unsigned A_SLOT = BK * BM * 4, X_SLOT = BK * BN * 4, L_SLOT = 8 * 256;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_d[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_d;
    const unsigned ldsA = lds0, ldsX = lds0 + 3 * A_SLOT, ldsL = ldsX + 3 * X_SLOT;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    int id = blockIdx.x, tile, z, nz;
    if (id < a.n_hi * (a.S + 1)) { tile = id / (a.S + 1); z = id - tile * (a.S + 1); nz = a.S + 1; }
    else { id -= a.n_hi * (a.S + 1); tile = a.n_hi + id / a.S; z = id - (tile - a.n_hi) * a.S; nz = a.S; }
    const int n0 = tile * BN;
    const int total_steps = a.M / BK;                           // (M % 32 == 0)
    const int g0 = (int)((long long)total_steps * z / nz), g1 = (int)((long long)total_steps * (z + 1) / nz);
    unsigned long long* const stamps = a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
    if (stamps && tid == 0) { stamps[0] = __builtin_readcyclecounter(); stamps[6] = __builtin_amdgcn_s_memrealtime(); }

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // DMA sources of this lane
    const float* srcA = a.D + (long long)(4 * wave) * a.H + 4 * lane;                 // + i rows, + 32 t rows
    const int xrow = 4 * wave + (lane >> 4), xlc = (lane & 15) ^ (8 * ((lane >> 5) & 1));
    const float* srcX = a.X + (long long)xrow * a.ldx + min(n0 + 4 * xlc, a.N - 4);
    const float* srcL = a.lse + (lane & 31);
    auto issue = [&](int t) __attribute__((always_inline)) {
        if (ABL == 2) return;
        const int k = min(t, g1 - 1);
        const unsigned slot = (unsigned)(t % 3);
        const long long r0 = (long long)k * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(srcA + (r0 + i) * a.H, ldsA + slot * A_SLOT + (unsigned)(4 * wave + i) * 1024u);
        if (NT) glds16_nt(srcX + r0 * a.ldx, ldsX + slot * X_SLOT + (unsigned)wave * 1024u);
        else    glds16(srcX + r0 * a.ldx, ldsX + slot * X_SLOT + (unsigned)wave * 1024u);
        if (SOFTMAX) glds4(srcL + r0, ldsL + slot * L_SLOT + (unsigned)wave * 256u);
    };
    constexpr int NDMA = SOFTMAX ? 6 : 5;                         // DMA instructions per wave and step
    f32x4 afA[2], afB[2];
    f32x2 bfA[2], bfB[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { afB[e] = f32x4{0.f, 0.f, 0.f, 0.f}; bfB[e] = f32x2{0.f, 0.f}; }
    const int xoff = 4 * ((wn0 / 4 + (li >> 1)) ^ (8 * (lk & 1))) + 2 * (li & 1);       // float offset inside an X row (swizzled chunk)
    // raw fragments of sub-step s (the softmax is applied later, by xform, once the reads have had a sub-step to land)
    f32x2 lA, lB;
    auto read_frags = [&](unsigned slot, int s, f32x4 (&af)[2], f32x2 (&bf)[2], f32x2& l2) __attribute__((always_inline)) {
        const int kk = 8 * s + 2 * lk;
        const float* pa = (const float*)(smem_d + slot * A_SLOT) + kk * BM + wm0 + 4 * li;
        const float* px = (const float*)(smem_d + 3 * A_SLOT + slot * X_SLOT) + kk * BN + xoff;
        if (SOFTMAX) l2 = *(const f32x2*)((const float*)(smem_d + 3 * A_SLOT + 3 * X_SLOT + slot * L_SLOT + wave * 256) + kk);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            af[e] = *(const f32x4*)(pa + e * BM);
            bf[e] = *(const f32x2*)(px + e * BN);
        }
    };
    auto xform = [&](f32x2 (&bf)[2], const f32x2& l2) __attribute__((always_inline)) {
        if (!SOFTMAX) return;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            bf[e][0] = __builtin_amdgcn_exp2f(__builtin_fmaf(bf[e][0], 1.44269504088896341f, -l2[e]));
            bf[e][1] = __builtin_amdgcn_exp2f(__builtin_fmaf(bf[e][1], 1.44269504088896341f, -l2[e]));
        }
    };
    auto mfma = [&](const f32x4 (&af)[2], const f32x2 (&bf)[2]) __attribute__((always_inline)) {
        if (ABL == 1) return;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e][i], bf[e][j], acc[i][j], 0, 0, 0);
    };
    issue(g0); issue(g0 + 1);
    if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    lB = f32x2{0.f, 0.f};
    // one sub-step: the raw fragments of the NEXT sub-step are requested under the first MFMAs, the softmax of those fragments is
    // formed under the last ones (their reads have landed by then: no exposed LDS latency)
    auto groups = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < 6) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
            else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int t = g0; t < g1; ++t) {
        const unsigned slot = (unsigned)(t % 3);
        read_frags(slot, 0, afA, bfA, lA);
        issue(t + 2);
        mfma(afB, bfB);                                  // (t - 1, last sub-step): covers the reads above
        xform(bfA, lA);
        groups();
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            auto& afc = (s & 1) ? afB : afA; auto& bfc = (s & 1) ? bfB : bfA;
            auto& afn = (s & 1) ? afA : afB; auto& bfn = (s & 1) ? bfA : bfB;
            auto& ln = (s & 1) ? lA : lB;
            read_frags(slot, s + 1, afn, bfn, ln);
            mfma(afc, bfc);
            xform(bfn, ln);
            groups();
        }
        if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // tile t + 1 landed (t + 2 may still fly)
        else           asm volatile("s_waitcnt vmcnt(5)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    mfma(afB, bfB);                                      // the last sub-step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamps && tid == 0) stamps[1] = __builtin_readcyclecounter();
    float* const slot_out = a.slab + (long long)blockIdx.x * (BM * BN);
#pragma unroll
    for (int q = 0; q < WM; ++q)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = wm0 + 4 * (4 * lk + rg) + q;
            const f32x2 v = {acc[q][0][rg], acc[q][1][rg]};
            *(f32x2*)(slot_out + row * BN + wn0 + 2 * li) = v;
        }
    if (stamps && tid == 0) { stamps[2] = __builtin_readcyclecounter(); stamps[7] = __builtin_amdgcn_s_memrealtime(); }
}

template <bool SOFTMAX, int ABL, bool NT>
static float run_d(const TnArgs& a0, int wgs_target, int reps, float* out, bool verbose, const char* name) {
    constexpr int BN = 64;
    TnArgs a = a0;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.S = wgs_target / a.tiles_n; if (a.S < 1) a.S = 1;
    a.n_hi = wgs_target - a.S * a.tiles_n; if (a.n_hi < 0 || a.n_hi > a.tiles_n) a.n_hi = 0;
    const int wgs = a.S * a.tiles_n + a.n_hi;
    const int lds = 3 * (32 * 256 * 4 + 32 * 64 * 4 + 8 * 256);
    CHECK(hipFuncSetAttribute((const void*)k_tn8d<SOFTMAX, ABL, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMalloc(&a.slab, (size_t)wgs * 256 * BN * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tn8d<SOFTMAX, ABL, NT>), dim3(wgs), dim3(512), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_tn8d<SOFTMAX, ABL, NT>), dim3(wgs), dim3(512), lds, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    if (out) {
        hipLaunchKernelGGL((k_reduce<BN>), dim3((unsigned)(((long long)256 * a.N + 255) / 256)), dim3(256), 0, 0, a.slab, 256, a.N, a.S, a.n_hi, out);
        CHECK(hipDeviceSynchronize());
    }
    const double gf = 2.0 * a.M * 256.0 * a.N / 1e9;
    if (verbose) printf("%-34s wgs %3d (S %d, %d tiles +1)  lds %6d  min %.1f med %.1f us  %.1f TFLOP/s (%.3f of 157.3)\n", name, wgs, a.S, a.n_hi, lds,
                        ts[0] * 1e3, med * 1e3, gf / med, gf / med / 157.3);
    CHECK(hipFree(a.slab));
    return med;
}


// ---- the same tile on the bf16 matrix path with fp32-grade operands ("bf16 x 6") ------------------------------------------------------------
// x = x1 + x2 + x3 EXACTLY (x1 = x & 0xFFFF0000, x2 = (x - x1) & 0xFFFF0000, x3 = x - x1 - x2: three bf16 values by truncation, every
// residual exact in fp32), products x1.w1 + (x1.w2 + x2.w1) + (x1.w3 + x2.w2 + x3.w1) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the
// dropped terms are 2^-24 relative, the fp32 rounding error itself (tools/exp_bf16x3.py measures it on the oracle: closer to fp64 than fp32 MFMA).
// 6 MFMAs of 16 cycles replace 8 of 32 per 32-deep block: 2.67 x the rate.  Split at LDS-store time (once per element); LDS holds three bf16
// planes per operand in the global [k][m] layout, fragments come from ds_read_b64_tr_b16 (hardware transpose), row pitches = 32 mod 256 bytes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool SOFTMAX, int ABL = 0, int MASK = 63>
__global__ __launch_bounds__(512, 1) void k_tn8x6(const TnArgs a) {
    constexpr int T = 512, BM = 256, BN = 64, BK = 32;
    constexpr int PA = 544, PB = 160;                          // bytes per k-row of a plane (512 + 32, 128 + 32)
    constexpr int A_PL = BK * PA, B_PL = BK * PB;              // one plane of one buffer
    constexpr int BUF = 3 * (A_PL + B_PL);
    constexpr int NA = 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    int id = blockIdx.x, tile, z, nz;
    if (id < a.n_hi * (a.S + 1)) { tile = id / (a.S + 1); z = id - tile * (a.S + 1); nz = a.S + 1; }
    else { id -= a.n_hi * (a.S + 1); tile = a.n_hi + id / a.S; z = id - (tile - a.n_hi) * a.S; nz = a.S; }
    const int n0 = tile * BN;
    const int total_steps = a.M / BK;
    const int g0 = (int)((long long)total_steps * z / nz), g1 = (int)((long long)total_steps * (z + 1) / nz);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int arow = tid >> 6, aq = tid & 63, brow = tid >> 4, bq = tid & 15;
    const int bcc = min(n0 + 4 * bq, a.N - 4);
    f32x4 va[2][NA], vb[2];
    float vl[2];
    auto issue = [&](auto set_c, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if (ABL == 2) return;
        const int r0 = min(t, g1 - 1) * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) va[S][i] = *(gf4ptr)((gfptr)a.D + (long long)(r0 + arow + 8 * i) * a.H + 4 * aq);
        vb[S] = *(gf4ptr)((gfptr)a.X + (long long)(r0 + brow) * a.ldx + bcc);
        if (SOFTMAX) vl[S] = ((gfptr)a.lse)[r0 + brow];
    };
    // x -> three bf16 planes (4 values: one 8-byte store per plane)
    auto split_store = [&](f32x4 v, unsigned char* base) __attribute__((always_inline)) {
        unsigned u[4], p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xj = v[j];                   // (a scalar copy: __builtin_bit_cast applied to the vector element v[j] itself reads element 0 -- hipcc 7.2)
            u[j] = __builtin_bit_cast(unsigned, xj);
            p1[j] = u[j] & 0xFFFF0000u;
            const float r1 = xj - __builtin_bit_cast(float, p1[j]);
            p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
            p3[j] = __builtin_bit_cast(unsigned, r2) & 0xFFFF0000u;
        }
        const u32x2 w1 = {(p1[0] >> 16) | p1[1], (p1[2] >> 16) | p1[3]};
        const u32x2 w2 = {(p2[0] >> 16) | p2[1], (p2[2] >> 16) | p2[3]};
        const u32x2 w3 = {(p3[0] >> 16) | p3[1], (p3[2] >> 16) | p3[3]};
        *(u32x2*)(base) = w1; *(u32x2*)(base + (A_PL + B_PL)) = w2; *(u32x2*)(base + 2 * (A_PL + B_PL)) = w3;
    };
    // buffer b: [plane p][A rows | B rows]: plane p at b * BUF + p * (A_PL + B_PL); A part first
    auto stash = [&](auto set_c, int buf, int h0, int h1) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if (ABL == 3) return;
        unsigned char* const base = sm6 + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            split_store(va[S][i], base + (arow + 8 * i) * PA + aq * 8);
        }
        if (NA >= h0 && NA < h1) {
            f32x4 v = vb[S];
            if (SOFTMAX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], 1.44269504088896341f, -vl[S]));
            }
            split_store(v, base + A_PL + brow * PB + bq * 8);
        }
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        if (ABL == 1) return;
        const unsigned char* const base = sm6 + buf * BUF;
        bf16x8 af[3][4], bf[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned char* ta = base + p * (A_PL + B_PL) + (4 * lk + tq) * PA + (wm0 + 4 * tp) * 2;
            const unsigned char* tb = base + p * (A_PL + B_PL) + A_PL + (4 * lk + tq) * PB + (wn0 + 4 * tp) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta + i * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta + i * 32 + 16 * PA));
                af[p][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + j * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + j * 32 + 16 * PB));
                bf[p][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
        // smallest terms first: (x1 w3 + x2 w2 + x3 w1), (x1 w2 + x2 w1), x1 w1
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 c = acc[i][j];
                if (MASK & 1)  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[2][j], c, 0, 0, 0);
                if (MASK & 2)  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bf[1][j], c, 0, 0, 0);
                if (MASK & 4)  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][i], bf[0][j], c, 0, 0, 0);
                if (MASK & 8)  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[1][j], c, 0, 0, 0);
                if (MASK & 16) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bf[0][j], c, 0, 0, 0);
                if (MASK & 32) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[0][j], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };
    typedef IntC<0> S0; typedef IntC<1> S1;
    issue(S0{}, g0);
    issue(S1{}, g0 + 1);
    stash(S0{}, 0, 0, NA + 1);
    issue(S0{}, g0 + 2);
    __syncthreads();
    auto step = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef IntC<PAR ^ 1> SS;
        stash(SS{}, PAR ^ 1, 0, NA + 1);
        issue(SS{}, t + 3);
        compute(PAR);
        __syncthreads();
    };
    int t = g0;
    for (; t + 1 < g1; t += 2) { step(IntC<0>{}, t); step(IntC<1>{}, t + 1); }
    if (t < g1) step(IntC<0>{}, t);
    float* const slot = a.slab + (long long)blockIdx.x * (BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = wm0 + 16 * i + 4 * lk + rg;
#pragma unroll
            for (int j = 0; j < 2; ++j) slot[row * BN + wn0 + 16 * j + li] = acc[i][j][rg];
        }
}


// The same arithmetic, pipelined like k_tn8: a k-step is four sub-steps (one per 16-row block column i of the wave's A side: 12 MFMAs), the
// fragment reads of sub-step i + 1, the split + LDS stores of the next tile (two quads per sub-step) and the global loads of the tile after it
// are pinned between the MFMAs of sub-step i; the last sub-step's MFMAs run after the barrier, over the first reads of the next buffer.
template <bool SOFTMAX, int VARIANT = 0>
__global__ __launch_bounds__(512, 1) void k_tn8x6p(const TnArgs a) {
    constexpr int T = 512, BM = 256, BN = 64, BK = 32;
    constexpr int PA = 544, PB = 160;
    constexpr int A_PL = BK * PA, B_PL = BK * PB, PL = A_PL + B_PL;
    constexpr int BUF = 3 * PL;
    constexpr int NA = 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 32 * (wave & 1);
    int id = blockIdx.x, tile, z, nz;
    if (id < a.n_hi * (a.S + 1)) { tile = id / (a.S + 1); z = id - tile * (a.S + 1); nz = a.S + 1; }
    else { id -= a.n_hi * (a.S + 1); tile = a.n_hi + id / a.S; z = id - (tile - a.n_hi) * a.S; nz = a.S; }
    unsigned long long* const stamps = a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
    if (stamps && tid == 0) { stamps[0] = __builtin_readcyclecounter(); stamps[6] = __builtin_amdgcn_s_memrealtime(); }
    const int n0 = tile * BN;
    const int total_steps = a.M / BK;
    const int g0 = (int)((long long)total_steps * z / nz), g1 = (int)((long long)total_steps * (z + 1) / nz);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int arow = tid >> 6, aq = tid & 63, brow = tid >> 4, bq = tid & 15;
    const int bcc = min(n0 + 4 * bq, a.N - 4);
    constexpr int NSET = (VARIANT >= 20 && VARIANT < 30) ? 3 : 2;       // register sets of global loads in flight
    f32x4 va[NSET][NA], vb[NSET];
    float vl[NSET];
    auto issue_part = [&](auto set_c, int t, int h0, int h1) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if (VARIANT == 12) return;
        const int r0 = VARIANT == 15 ? (t & 3) * BK : min(t, g1 - 1) * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) { if (i < h0 || i >= h1) continue; va[S][i] = *(gf4ptr)((gfptr)a.D + (long long)(r0 + arow + 8 * i) * a.H + 4 * aq); }
        if (NA >= h0 && NA < h1) {
            vb[S] = *(gf4ptr)((gfptr)a.X + (long long)(r0 + brow) * a.ldx + bcc);
            if (SOFTMAX) vl[S] = ((gfptr)a.lse)[r0 + brow];
        }
    };
    auto issue = [&](auto set_c, int t) __attribute__((always_inline)) { issue_part(set_c, t, 0, NA + 1); };
    float dummy = 0.f; const float cst = (float)a.M * 0.37f;
    auto split_store = [&](f32x4 v, unsigned char* base) __attribute__((always_inline)) {
        if (VARIANT == 13) return;
        if (VARIANT == 16) {
            dummy += (float)v[0] + (float)v[3];
            v = f32x4{cst, cst * 3.f, cst * 5.f, cst * 7.f};
        }
        if (VARIANT == 10) {
            const u32x2 w1 = {__builtin_bit_cast(unsigned, (float)v[0]), __builtin_bit_cast(unsigned, (float)v[1])}, w2 = {__builtin_bit_cast(unsigned, (float)v[2]), __builtin_bit_cast(unsigned, (float)v[3])};
            *(u32x2*)(base) = w1; *(u32x2*)(base + PL) = w2; *(u32x2*)(base + 2 * PL) = w1;
            return;
        }
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xj = v[j];
            p1[j] = __builtin_bit_cast(unsigned, xj) & 0xFFFF0000u;
            const float r1 = xj - __builtin_bit_cast(float, p1[j]);
            p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
            p3[j] = __builtin_bit_cast(unsigned, r2);            // (at most 8 significant bits are left: the high half is all of it)
        }
        // v_perm_b32: the high halves of two dwords -> one dword (element j in the low half)
        const u32x2 w1 = {__builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const u32x2 w2 = {__builtin_amdgcn_perm(p2[1], p2[0], 0x07060302u), __builtin_amdgcn_perm(p2[3], p2[2], 0x07060302u)};
        const u32x2 w3 = {__builtin_amdgcn_perm(p3[1], p3[0], 0x07060302u), __builtin_amdgcn_perm(p3[3], p3[2], 0x07060302u)};
        *(u32x2*)(base) = w1; *(u32x2*)(base + PL) = w2; *(u32x2*)(base + 2 * PL) = w3;
    };
    auto stash = [&](auto set_c, int buf, int h0, int h1) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        unsigned char* const base = sm6 + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            split_store(va[S][i], base + (arow + 8 * i) * PA + aq * 8);
        }
        if (NA >= h0 && NA < h1) {
            f32x4 v = vb[S];
            if (SOFTMAX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float xj = v[j]; v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(xj, 1.44269504088896341f, -vl[S])); }
            }
            split_store(v, base + A_PL + brow * PB + bq * 8);
        }
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;
    const int offA = (4 * lk + tq) * PA + (wm0 + 4 * tp) * 2, offB = A_PL + (4 * lk + tq) * PB + (wn0 + 4 * tp) * 2;
    auto read_a = [&](int buf, int i, bf16x8 (&af)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (VARIANT == 14 && p > 0) { af[p] = af[0]; continue; }
            const unsigned char* ta = sm6 + buf * BUF + p * PL + offA + i * 32;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta + 16 * PA));
            af[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto read_b = [&](int buf, bf16x8 (&bf)[3][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (VARIANT == 14 && p > 0) { bf[p][j] = bf[0][j]; continue; }
                const unsigned char* tb = sm6 + buf * BUF + p * PL + offB + j * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + 16 * PB));
                bf[p][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
    };
    // 12 MFMAs of block row i: the two column blocks alternate (a dependent MFMA is two issues away); small terms first
    auto mfma12 = [&](const bf16x8 (&af)[3], const bf16x8 (&bf)[3][2], f32x4 (&c)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2][j], c[j], 0, 0, 0);
        if (VARIANT == 11) { c[0][0] += (float)af[1][0] + (float)af[2][0] + (float)bf[1][0][0] + (float)bf[1][1][0] + (float)bf[0][0][0] + (float)bf[0][1][0]; return; }
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
    typedef IntC<0> S0; typedef IntC<1> S1;
    bf16x8 afA[3], afB[3], bfr[2][3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        afB[p] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[1][p][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    auto pin = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VARIANT == 1 ? 3 : 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NSET == 3) {
        // three sets: tile t lives in set (t - g0) % 3; step n stashes tile n + 1 and re-issues set n % 3 (stashed one step ago) FIRST: a load has two k-steps to land
        issue(IntC<0>{}, g0); issue(IntC<1>{}, g0 + 1); issue(IntC<2>{}, g0 + 2);
        stash(IntC<0>{}, 0, 0, NA + 1);
        __syncthreads();
        auto step3 = [&](auto par_c, auto ns_c, int t) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value, NS = decltype(ns_c)::value;
            typedef IntC<(NS + 1) % 3> SS;
            read_b(PAR, bfr[PAR]);
            read_a(PAR, 0, afA);
            issue(IntC<NS>{}, t + 3);
            mfma12(afB, bfr[PAR ^ 1], acc[3]);
#pragma unroll
            for (int q = 0; q < 12; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            read_a(PAR, 1, afB);
            stash(SS{}, PAR ^ 1, 0, 2);
            mfma12(afA, bfr[PAR], acc[0]);
            pin();
            read_a(PAR, 2, afA);
            stash(SS{}, PAR ^ 1, 2, 4);
            mfma12(afB, bfr[PAR], acc[1]);
            pin();
            read_a(PAR, 3, afB);
            stash(SS{}, PAR ^ 1, 4, 5);
            mfma12(afA, bfr[PAR], acc[2]);
            pin();
            __syncthreads();
        };
        int t = g0;
        for (; t + 5 < g1; t += 6) {
            step3(IntC<0>{}, IntC<0>{}, t); step3(IntC<1>{}, IntC<1>{}, t + 1); step3(IntC<0>{}, IntC<2>{}, t + 2);
            step3(IntC<1>{}, IntC<0>{}, t + 3); step3(IntC<0>{}, IntC<1>{}, t + 4); step3(IntC<1>{}, IntC<2>{}, t + 5);
        }
        // (the microbenchmark's chunks are multiples of 6 k-steps; anything else would need the tail here)
        mfma12(afB, bfr[1], acc[3]);
    } else {
    issue(S0{}, g0);
    issue(S1{}, g0 + 1);
    stash(S0{}, 0, 0, NA + 1);
    issue(S0{}, g0 + 2);
    __syncthreads();
    // VARIANT 0 / 1x: the loads of tile t + 3 are spread over the step (one every four MFMAs): a burst of six per wave from eight waves
    // at one program point queues on the CU's address unit (16 cycles per 1 KB load) and stalls the in-order waves behind it
    auto pin_spread = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (q % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stash_b = [&](auto set_c, int buf) __attribute__((always_inline)) { stash(set_c, buf, NA, NA + 1); };
    auto step = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef IntC<PAR ^ 1> SS;
        read_b(PAR, bfr[PAR]);
        read_a(PAR, 0, afA);
        mfma12(afB, bfr[PAR ^ 1], acc[3]);
#pragma unroll
        for (int q = 0; q < 12; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT == 1) {          // the burst form (all six loads in the last sub-step)
            read_a(PAR, 1, afB); stash(SS{}, PAR ^ 1, 0, 2); mfma12(afA, bfr[PAR], acc[0]); pin();
            read_a(PAR, 2, afA); stash(SS{}, PAR ^ 1, 2, 4); mfma12(afB, bfr[PAR], acc[1]); pin();
            read_a(PAR, 3, afB); stash(SS{}, PAR ^ 1, 4, 5); issue(SS{}, t + 3); mfma12(afA, bfr[PAR], acc[2]); pin();
        } else {
            read_a(PAR, 1, afB); stash_b(SS{}, PAR ^ 1); stash(SS{}, PAR ^ 1, 0, 1); mfma12(afA, bfr[PAR], acc[0]); pin_spread();
            read_a(PAR, 2, afA); stash(SS{}, PAR ^ 1, 1, 3); issue_part(SS{}, t + 3, NA, NA + 1); issue_part(SS{}, t + 3, 0, 1); mfma12(afB, bfr[PAR], acc[1]); pin_spread();
            read_a(PAR, 3, afB); stash(SS{}, PAR ^ 1, 3, 4); issue_part(SS{}, t + 3, 1, 4); mfma12(afA, bfr[PAR], acc[2]); pin_spread();
        }
        __syncthreads();
    };
    int t = g0;
    for (; t + 1 < g1; t += 2) { step(IntC<0>{}, t); step(IntC<1>{}, t + 1); }
    if (t < g1) { step(IntC<0>{}, t); mfma12(afB, bfr[0], acc[3]); }
    else mfma12(afB, bfr[1], acc[3]);
    }
    if (VARIANT == 16) acc[0][0][0] += dummy;
    if (stamps && tid == 0) { stamps[1] = __builtin_readcyclecounter(); stamps[2] = stamps[1]; stamps[7] = __builtin_amdgcn_s_memrealtime(); }
    float* const slot = a.slab + (long long)blockIdx.x * (BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = wm0 + 16 * i + 4 * lk + rg;
#pragma unroll
            for (int j = 0; j < 2; ++j) slot[row * BN + wn0 + 16 * j + li] = acc[i][j][rg];
        }
}


// 256 x 128 tiles on the bf16 x 6 path: a wave owns 64 x 64 (fragment bytes per flop -33 %, operand loads per flop -40 % against 256 x 64); the column
// fragments of a step are read once (no carry over the barrier: 48 registers), LDS = 2 x 3 x 32 x (544 + 288) = 159 744 bytes.
template <bool SOFTMAX>
__global__ __launch_bounds__(512, 1) void k_tn8x6w(const TnArgs a) {
    constexpr int T = 512, BM = 256, BN = 128, BK = 32;
    constexpr int PA = 544, PB = 288;
    constexpr int A_PL = BK * PA, B_PL = BK * PB, PL = A_PL + B_PL;
    constexpr int BUF = 3 * PL;
    constexpr int NA = 4, NX = 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm6[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = 64 * (wave >> 1), wn0 = 64 * (wave & 1);
    int id = blockIdx.x, tile, z, nz;
    if (id < a.n_hi * (a.S + 1)) { tile = id / (a.S + 1); z = id - tile * (a.S + 1); nz = a.S + 1; }
    else { id -= a.n_hi * (a.S + 1); tile = a.n_hi + id / a.S; z = id - (tile - a.n_hi) * a.S; nz = a.S; }
    const int n0 = tile * BN;
    const int total_steps = a.M / BK;
    const int g0 = (int)((long long)total_steps * z / nz), g1 = (int)((long long)total_steps * (z + 1) / nz);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int arow = tid >> 6, aq = tid & 63;                 // A: rows arow + 8 i, quad aq
    const int brow = tid >> 5, bq = tid & 31;                 // X: rows brow + 16 i, quad bq (32 quads per 128-column row)
    const int bcc = min(n0 + 4 * bq, a.N - 4);
    f32x4 va[2][NA], vb[2][NX];
    float vl[2][NX];
    auto issue = [&](auto set_c, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        const int r0 = min(t, g1 - 1) * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) va[S][i] = *(gf4ptr)((gfptr)a.D + (long long)(r0 + arow + 8 * i) * a.H + 4 * aq);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            vb[S][i] = *(gf4ptr)((gfptr)a.X + (long long)(r0 + brow + 16 * i) * a.ldx + bcc);
            if (SOFTMAX) vl[S][i] = ((gfptr)a.lse)[r0 + brow + 16 * i];
        }
    };
    auto split_store = [&](f32x4 v, unsigned char* base) __attribute__((always_inline)) {
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xj = v[j];
            p1[j] = __builtin_bit_cast(unsigned, xj) & 0xFFFF0000u;
            const float r1 = xj - __builtin_bit_cast(float, p1[j]);
            p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
            p3[j] = __builtin_bit_cast(unsigned, r2);
        }
        const u32x2 w1 = {__builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const u32x2 w2 = {__builtin_amdgcn_perm(p2[1], p2[0], 0x07060302u), __builtin_amdgcn_perm(p2[3], p2[2], 0x07060302u)};
        const u32x2 w3 = {__builtin_amdgcn_perm(p3[1], p3[0], 0x07060302u), __builtin_amdgcn_perm(p3[3], p3[2], 0x07060302u)};
        *(u32x2*)(base) = w1; *(u32x2*)(base + PL) = w2; *(u32x2*)(base + 2 * PL) = w3;
    };
    auto stash = [&](auto set_c, int buf, int h0, int h1) __attribute__((always_inline)) {        // items 0 .. 3: A quads, 4 .. 5: X quads
        constexpr int S = decltype(set_c)::value;
        unsigned char* const base = sm6 + buf * BUF;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < h0 || i >= h1) continue;
            split_store(va[S][i], base + (arow + 8 * i) * PA + aq * 8);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (NA + i < h0 || NA + i >= h1) continue;
            f32x4 v = vb[S][i];
            if (SOFTMAX) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float xj = v[j]; v[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(xj, 1.44269504088896341f, -vl[S][i])); }
            }
            split_store(v, base + A_PL + (brow + 16 * i) * PB + bq * 8);
        }
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;
    const int offA = (4 * lk + tq) * PA + (wm0 + 4 * tp) * 2, offB = A_PL + (4 * lk + tq) * PB + (wn0 + 4 * tp) * 2;
    auto read_a = [&](int buf, int i, bf16x8 (&af)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned char* ta = sm6 + buf * BUF + p * PL + offA + i * 32;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ta + 16 * PA));
            af[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto read_b = [&](int buf, bf16x8 (&bf)[3][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char* tb = sm6 + buf * BUF + p * PL + offB + j * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + 16 * PB));
                bf[p][j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
    };
    auto mfma24 = [&](const bf16x8 (&af)[3], const bf16x8 (&bf)[3][4], f32x4 (&c)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[1][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[0][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[1][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[0][j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[0][j], c[j], 0, 0, 0);
    };
    auto pin = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef IntC<0> S0; typedef IntC<1> S1;
    issue(S0{}, g0);
    issue(S1{}, g0 + 1);
    stash(S0{}, 0, 0, NA + NX);
    issue(S0{}, g0 + 2);
    __syncthreads();
    auto step = [&](auto par_c, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef IntC<PAR ^ 1> SS;
        bf16x8 bfr[3][4], afA[3], afB[3];
        read_b(PAR, bfr);
        read_a(PAR, 0, afA);
        read_a(PAR, 1, afB); stash(SS{}, PAR ^ 1, 0, 2); mfma24(afA, bfr, acc[0]); pin();
        read_a(PAR, 2, afA); stash(SS{}, PAR ^ 1, 2, 4); mfma24(afB, bfr, acc[1]); pin();
        read_a(PAR, 3, afB); stash(SS{}, PAR ^ 1, 4, 6); mfma24(afA, bfr, acc[2]); pin();
        issue(SS{}, t + 3); mfma24(afB, bfr, acc[3]); pin();
        __syncthreads();
    };
    int t = g0;
    for (; t + 1 < g1; t += 2) { step(IntC<0>{}, t); step(IntC<1>{}, t + 1); }
    if (t < g1) step(IntC<0>{}, t);
    float* const slot = a.slab + (long long)blockIdx.x * (BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = wm0 + 16 * i + 4 * lk + rg;
#pragma unroll
            for (int j = 0; j < 4; ++j) slot[row * BN + wn0 + 16 * j + li] = acc[i][j][rg];
        }
}

template <bool SOFTMAX>
static float run_x6w(const TnArgs& a0, int wgs_target, int reps, float* out, bool verbose, const char* name) {
    constexpr int BN = 128;
    TnArgs a = a0;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.S = wgs_target / a.tiles_n; if (a.S < 1) a.S = 1;
    a.n_hi = wgs_target - a.S * a.tiles_n; if (a.n_hi < 0 || a.n_hi > a.tiles_n) a.n_hi = 0;
    const int wgs = a.S * a.tiles_n + a.n_hi;
    const int lds = 2 * 3 * 32 * (544 + 288);
    CHECK(hipFuncSetAttribute((const void*)k_tn8x6w<SOFTMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMalloc(&a.slab, (size_t)wgs * 256 * BN * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tn8x6w<SOFTMAX>), dim3(wgs), dim3(512), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_tn8x6w<SOFTMAX>), dim3(wgs), dim3(512), lds, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    if (out) {
        hipLaunchKernelGGL((k_reduce<BN>), dim3((unsigned)(((long long)256 * a.N + 255) / 256)), dim3(256), 0, 0, a.slab, 256, a.N, a.S, a.n_hi, out);
        CHECK(hipDeviceSynchronize());
    }
    const double gf = 2.0 * a.M * 256.0 * a.N / 1e9;
    if (verbose) printf("%-34s wgs %3d (S %d)  lds %6d  min %.1f med %.1f us  %.1f TFLOP/s fp32-equivalent (%.3f of the 157.3 fp32-MFMA peak)  slab %.1f MB\n", name, wgs, a.S, lds,
                        ts[0] * 1e3, med * 1e3, gf / med, gf / med / 157.3, wgs * 256.0 * BN * 4 / 1e6);
    CHECK(hipFree(a.slab));
    return med;
}

template <bool SOFTMAX, int VARIANT>
static float run_x6p(const TnArgs& a0, int wgs_target, int reps, float* out, bool verbose, const char* name) {
    constexpr int BN = 64;
    TnArgs a = a0;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.S = wgs_target / a.tiles_n; if (a.S < 1) a.S = 1;
    a.n_hi = wgs_target - a.S * a.tiles_n; if (a.n_hi < 0 || a.n_hi > a.tiles_n) a.n_hi = 0;
    const int wgs = a.S * a.tiles_n + a.n_hi;
    const int lds = 2 * 3 * 32 * (544 + 160);
    CHECK(hipFuncSetAttribute((const void*)k_tn8x6p<SOFTMAX, VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMalloc(&a.slab, (size_t)wgs * 256 * BN * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tn8x6p<SOFTMAX, VARIANT>), dim3(wgs), dim3(512), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_tn8x6p<SOFTMAX, VARIANT>), dim3(wgs), dim3(512), lds, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    if (out) {
        hipLaunchKernelGGL((k_reduce<BN>), dim3((unsigned)(((long long)256 * a.N + 255) / 256)), dim3(256), 0, 0, a.slab, 256, a.N, a.S, a.n_hi, out);
        CHECK(hipDeviceSynchronize());
    }
    const double gf = 2.0 * a.M * 256.0 * a.N / 1e9;
    if (verbose) printf("%-34s wgs %3d (S %d)  lds %6d  min %.1f med %.1f us  %.1f TFLOP/s fp32-equivalent (%.3f of the 157.3 fp32-MFMA peak)\n", name, wgs, a.S, lds,
                        ts[0] * 1e3, med * 1e3, gf / med, gf / med / 157.3);
    CHECK(hipFree(a.slab));
    return med;
}

template <bool SOFTMAX, int ABL, int MASK = 63>
static float run_x6(const TnArgs& a0, int wgs_target, int reps, float* out, bool verbose, const char* name) {
    constexpr int BN = 64;
    TnArgs a = a0;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.S = wgs_target / a.tiles_n; if (a.S < 1) a.S = 1;
    a.n_hi = wgs_target - a.S * a.tiles_n; if (a.n_hi < 0 || a.n_hi > a.tiles_n) a.n_hi = 0;
    const int wgs = a.S * a.tiles_n + a.n_hi;
    const int lds = 2 * 3 * 32 * (544 + 160);
    CHECK(hipFuncSetAttribute((const void*)k_tn8x6<SOFTMAX, ABL, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMalloc(&a.slab, (size_t)wgs * 256 * BN * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tn8x6<SOFTMAX, ABL, MASK>), dim3(wgs), dim3(512), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_tn8x6<SOFTMAX, ABL, MASK>), dim3(wgs), dim3(512), lds, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    if (out) {
        hipLaunchKernelGGL((k_reduce<BN>), dim3((unsigned)(((long long)256 * a.N + 255) / 256)), dim3(256), 0, 0, a.slab, 256, a.N, a.S, a.n_hi, out);
        CHECK(hipDeviceSynchronize());
    }
    const double gf = 2.0 * a.M * 256.0 * a.N / 1e9;
    if (verbose) printf("%-34s wgs %3d (S %d)  lds %6d  min %.1f med %.1f us  %.1f TFLOP/s fp32-equivalent (%.3f of the 157.3 fp32-MFMA peak)\n", name, wgs, a.S, lds,
                        ts[0] * 1e3, med * 1e3, gf / med, gf / med / 157.3);
    CHECK(hipFree(a.slab));
    return med;
}

template <int BN, bool SOFTMAX, int ABL>
static float run(const TnArgs& a0, int wgs_target, int reps, float* out, bool verbose, const char* name) {
    TnArgs a = a0;
    a.tiles_n = (a.N + BN - 1) / BN;
    // fill wgs_target workgroups: S = floor(target / tiles), the first n_hi tiles take one chunk more
    a.S = wgs_target / a.tiles_n; if (a.S < 1) a.S = 1;
    a.n_hi = wgs_target - a.S * a.tiles_n; if (a.n_hi < 0 || a.n_hi > a.tiles_n) a.n_hi = 0;
    const int wgs = a.S * a.tiles_n + a.n_hi;
    constexpr int PB = BN == 128 ? 128 : 80;
    const int lds = 2 * 32 * (256 + PB) * 4;
    CHECK(hipFuncSetAttribute((const void*)k_tn8<BN, SOFTMAX, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMalloc(&a.slab, (size_t)wgs * 256 * BN * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tn8<BN, SOFTMAX, ABL>), dim3(wgs), dim3(512), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_tn8<BN, SOFTMAX, ABL>), dim3(wgs), dim3(512), lds, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    if (out) {
        hipLaunchKernelGGL((k_reduce<BN>), dim3((unsigned)(((long long)256 * a.N + 255) / 256)), dim3(256), 0, 0, a.slab, 256, a.N, a.S, a.n_hi, out);
        CHECK(hipDeviceSynchronize());
    }
    const double gf = 2.0 * a.M * 256.0 * a.N / 1e9;
    if (verbose) printf("%-34s wgs %3d (S %d, %d tiles +1)  lds %6d  min %.1f med %.1f us  %.1f TFLOP/s (%.3f of 157.3)  slab %.1f MB\n", name, wgs, a.S, a.n_hi, lds,
                        ts[0] * 1e3, med * 1e3, gf / med, gf / med / 157.3, wgs * 256.0 * BN * 4 / 1e6);
    CHECK(hipFree(a.slab));
    return med;
}


// ---- diagnostic: what the split stores and the transposed reads of k_tn8x6 really see ------------------------------------------------------
__global__ void k_dbg_split(const float* x /*[32][64]*/, unsigned short* planes /*[3][32][64]*/, unsigned short* frag /*[3][64 lanes][8]*/) {
    constexpr int PB = 160, B_PL = 32 * PB;
    __shared__ __attribute__((aligned(1024))) unsigned char sm[3 * B_PL];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int brow = tid >> 4, bq = tid & 15;                // 512 threads: 32 rows x 16 quads
    const f32x4 v = *(const f32x4*)(x + brow * 64 + 4 * bq);
    unsigned p1[4], p2[4], p3[4];
    for (int j = 0; j < 4; ++j) {
        const float xj = v[j];
        const unsigned u = __builtin_bit_cast(unsigned, xj);
        p1[j] = u & 0xFFFF0000u;
        const float r1 = xj - __builtin_bit_cast(float, p1[j]);
        p2[j] = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, p2[j]);
        p3[j] = __builtin_bit_cast(unsigned, r2) & 0xFFFF0000u;
    }
    unsigned char* base = sm + brow * PB + bq * 8;
    *(u32x2*)(base) = u32x2{(p1[0] >> 16) | p1[1], (p1[2] >> 16) | p1[3]};
    *(u32x2*)(base + B_PL) = u32x2{(p2[0] >> 16) | p2[1], (p2[2] >> 16) | p2[3]};
    *(u32x2*)(base + 2 * B_PL) = u32x2{(p3[0] >> 16) | p3[1], (p3[2] >> 16) | p3[3]};
    __syncthreads();
    for (int i = tid; i < 3 * 32 * 64; i += 512) {
        const int p = i / (32 * 64), r = (i / 64) % 32, c = i % 64;
        planes[i] = *(const unsigned short*)(sm + p * B_PL + r * PB + c * 2);
    }
    if (tid < 64) {
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        const int tq = li >> 2, tp = li & 3;
        for (int p = 0; p < 3; ++p) {
            const unsigned char* tb = sm + p * B_PL + (4 * lk + tq) * PB + (0 + 4 * tp) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tb + 16 * PB));
            for (int j = 0; j < 4; ++j) { frag[(p * 64 + lane) * 8 + j] = (unsigned short)lo[j]; frag[(p * 64 + lane) * 8 + 4 + j] = (unsigned short)hi[j]; }
        }
    }
}
static void debug_split() {
    std::vector<float> hx(32 * 64);
    for (int i = 0; i < 32 * 64; ++i) hx[i] = (float)(i / 64) + (float)(i % 64) / 256.f + 1.0f / 3.0f;       // row r, column c encoded in the value
    float* dx; unsigned short *dp, *df;
    CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dp, 3 * 32 * 64 * 2)); CHECK(hipMalloc(&df, 3 * 64 * 8 * 2));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dbg_split, dim3(1), dim3(512), 0, 0, dx, dp, df);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned short> hp(3 * 32 * 64), hf(3 * 64 * 8);
    CHECK(hipMemcpy(hp.data(), dp, hp.size() * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hf.data(), df, hf.size() * 2, hipMemcpyDeviceToHost));
    auto b2f = [](unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
    int bad = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 64; ++c) {
        const float x = hx[r * 64 + c];
        const double sum = (double)b2f(hp[r * 64 + c]) + b2f(hp[2048 + r * 64 + c]) + b2f(hp[4096 + r * 64 + c]);
        if (fabs(sum - x) > 1e-7 * fabs(x) || fabs(b2f(hp[2048 + r * 64 + c])) > 0.01 * fabs(x)) { if (bad < 4) printf("   plane mismatch r %d c %d: x %.8f planes %.8f %.3e %.3e\n", r, c, x, b2f(hp[r * 64 + c]), b2f(hp[2048 + r * 64 + c]), b2f(hp[4096 + r * 64 + c])); ++bad; }
    }
    printf("dbg split: %d of 2048 elements whose three planes do not sum to x (or whose middle plane is not small)\n", bad);
    // fragments of plane 0: which (row, column) does lane l, element e hold?  value = r + c / 256 + 1/3 (truncated to bf16: r exact for r < 128)
    for (int l = 0; l < 64; l += 9) {
        printf("   lane %2d (li %2d lk %d): ", l, l & 15, l >> 4);
        for (int e = 0; e < 8; ++e) printf("%.3f ", b2f(hf[l * 8 + e]));
        printf(" | mid plane: ");
        for (int e = 0; e < 8; ++e) printf("%.1e ", b2f(hf[(64 + l) * 8 + e]));
        printf("\n");
    }
}

int main(int argc, char** argv) {
    if (getenv("MB_TN_DEBUG_SPLIT")) debug_split();
    const int M = argc > 1 ? atoi(argv[1]) : 12288, N = argc > 2 ? atoi(argv[2]) : 2000, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int H = 256;
    std::vector<float> hD((size_t)M * H), hX((size_t)M * N), hl(M);
    srand(1);
    for (auto& v : hD) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
    for (int r = 0; r < M; ++r) {
        double mx = -1e30, s = 0; for (int c = 0; c < N; ++c) mx = std::max(mx, (double)hX[(size_t)r * N + c]);
        for (int c = 0; c < N; ++c) s += exp((double)hX[(size_t)r * N + c] - mx);
        hl[r] = (float)((log(s) + mx) * 1.4426950408889634);
    }
    TnArgs a{};
    float *dD, *dX, *dl, *dout;
    CHECK(hipMalloc(&dD, hD.size() * 4)); CHECK(hipMalloc(&dX, hX.size() * 4)); CHECK(hipMalloc(&dl, M * 4)); CHECK(hipMalloc(&dout, (size_t)H * N * 4));
    CHECK(hipMemcpy(dD, hD.data(), hD.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dl, hl.data(), M * 4, hipMemcpyHostToDevice));
    a.D = dD; a.H = H; a.X = dX; a.ldx = N; a.lse = dl; a.M = M; a.N = N;
    // correctness on a sample of outputs (fp64 reference)
    auto check = [&](const char* name) {
        std::vector<float> ho((size_t)H * N);
        CHECK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, mag = 0;
        for (int s = 0; s < 64; ++s) {
            const int h = (s * 37) % H, n = s < 8 ? N - 1 - s : (s * 997) % N;
            double ref = 0;
            for (int r = 0; r < M; ++r) ref += (double)hD[(size_t)r * H + h] * exp2((double)hX[(size_t)r * N + n] * 1.4426950408889634 - (double)hl[r]);
            worst = std::max(worst, fabs(ref - ho[(size_t)h * N + n])); mag = std::max(mag, fabs(ref));
        }
        double omax = 0; for (size_t q = 0; q < ho.size(); q += 97) omax = std::max(omax, (double)fabs(ho[q]));
        printf("   check %-24s max |err| %.3e (max |ref| %.3e; max |out| sampled %.3e)\n", name, worst, mag, omax);
    };
    // warm the chip
    for (int i = 0; i < 30; ++i) run<64, true, 0>(a, 256, 3, nullptr, false, "");
    run<64, true, 0>(a, 256, reps, dout, true, "256x64  8 waves, 256 wgs"); check("256x64");
    run<128, true, 0>(a, 256, reps, dout, true, "256x128 8 waves, 256 wgs"); check("256x128");
    run<64, true, 0>(a, 248, reps, nullptr, true, "256x64  8 waves, 248 wgs");
    run<64, true, 1>(a, 256, reps, nullptr, true, "256x64  ABL no mfma");
    run<64, true, 2>(a, 256, reps, nullptr, true, "256x64  ABL no global loads");
    run<64, true, 3>(a, 256, reps, nullptr, true, "256x64  ABL no LDS stores");
    run<128, true, 1>(a, 256, reps, nullptr, true, "256x128 ABL no mfma");
    run<128, true, 2>(a, 256, reps, nullptr, true, "256x128 ABL no global loads");
    run<128, true, 3>(a, 256, reps, nullptr, true, "256x128 ABL no LDS stores");
    run<64, false, 0>(a, 256, reps, nullptr, true, "256x64  plain operand");
    run_x6<true, 0>(a, 256, reps, dout, true, "256x64  bf16 x 6 (fp32-grade)"); check("256x64 bf16x6");
    run_x6p<true, 0>(a, 256, reps, dout, true, "256x64  bf16 x 6 pipelined"); check("256x64 bf16x6 pipelined");
    run_x6p<true, 1>(a, 256, reps, dout, true, "256x64  bf16 x 6 pipelined, load burst"); check("256x64 bf16x6 pipelined v1");
    run_x6p<false, 0>(a, 256, reps, nullptr, true, "256x64  bf16 x 6 pipelined plain");
    run_x6p<true, 20>(a, 256, reps, dout, true, "256x64  bf16 x 6, 3 load sets"); check("256x64 bf16x6 3 sets");
    run_x6w<true>(a, 256, reps, dout, true, "256x128 bf16 x 6"); check("256x128 bf16x6");
    run_x6p<true, 15>(a, 256, reps, nullptr, true, "   x6p ABL loads of the same 4 tiles");
    run_x6p<true, 16>(a, 256, reps, nullptr, true, "   x6p ABL stores do not depend on loads");
    run_x6p<true, 10>(a, 256, reps, nullptr, true, "   x6p ABL no split arithmetic");
    run_x6p<true, 11>(a, 256, reps, nullptr, true, "   x6p ABL 2 of 12 MFMAs");
    run_x6p<true, 12>(a, 256, reps, nullptr, true, "   x6p ABL no global loads");
    run_x6p<true, 13>(a, 256, reps, nullptr, true, "   x6p ABL no LDS stores");
    run_x6p<true, 14>(a, 256, reps, nullptr, true, "   x6p ABL 1/3 of the fragment reads");
    run_x6<true, 0, 32>(a, 256, 2, dout, false, ""); check("x6 terms: hi.hi");
    run_x6<true, 0, 32 + 8>(a, 256, 2, dout, false, ""); check("x6 terms: + hi.mid");
    run_x6<true, 0, 32 + 16>(a, 256, 2, dout, false, ""); check("x6 terms: + mid.hi");
    run_x6<true, 0, 32 + 8 + 16>(a, 256, 2, dout, false, ""); check("x6 terms: + hi.mid + mid.hi");
    run_x6<true, 0, 8>(a, 256, 2, dout, false, ""); check("x6 terms: hi.mid ALONE");
    run_x6<true, 0, 16>(a, 256, 2, dout, false, ""); check("x6 terms: mid.hi ALONE");
    run_x6<true, 1>(a, 256, reps, nullptr, true, "256x64  bf16 x 6 ABL no mfma/reads");
    run_x6<true, 2>(a, 256, reps, nullptr, true, "256x64  bf16 x 6 ABL no global loads");
    run_x6<true, 3>(a, 256, reps, nullptr, true, "256x64  bf16 x 6 ABL no LDS stores");
    run_d<true, 0, true>(a, 256, reps, dout, true, "256x64  LDS-DMA (nt X)"); check("256x64 dma");
    run_d<true, 0, false>(a, 256, reps, nullptr, true, "256x64  LDS-DMA (no nt hint)");
    run_d<true, 1, true>(a, 256, reps, nullptr, true, "256x64  LDS-DMA ABL no mfma");
    run_d<true, 2, true>(a, 256, reps, nullptr, true, "256x64  LDS-DMA ABL no loads");
    run_d<false, 0, true>(a, 256, reps, nullptr, true, "256x64  LDS-DMA plain operand");
    // stamps: per-workgroup loop / epilogue cycles and the held clock
    auto stamped = [&](const char* name, auto runner) {
        unsigned long long* st; CHECK(hipMalloc(&st, 256 * 8 * 8)); CHECK(hipMemset(st, 0, 256 * 8 * 8));
        TnArgs b = a; b.stamps = st;
        runner(b);
        std::vector<unsigned long long> hs(256 * 8);
        CHECK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> loop, epi, mhz, dur;
        for (int w = 0; w < 256; ++w) { const auto* s = &hs[w * 8]; if (s[7] > s[6]) { loop.push_back((double)(s[1] - s[0])); epi.push_back((double)(s[2] - s[1])); mhz.push_back((double)(s[2] - s[0]) / (s[7] - s[6]) * 100.0); dur.push_back((s[7] - s[6]) / 100.0); } }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, x); return m; };
        printf("stamps %-28s loop cycles med %.0f max %.0f | epilogue med %.0f | sclk med %.0f MHz | workgroup us med %.1f max %.1f\n", name, med(loop), mx(loop), med(epi), med(mhz), med(dur), mx(dur));
        CHECK(hipFree(st));
    };
    stamped("256x64 fp32", [&](const TnArgs& b) { run<64, true, 0>(b, 256, 3, nullptr, false, ""); });
    stamped("256x64 bf16x6 pipelined", [&](const TnArgs& b) { run_x6p<true, 0>(b, 256, 3, nullptr, false, ""); });
    stamped("256x64 bf16x6 no loads", [&](const TnArgs& b) { run_x6p<true, 12>(b, 256, 3, nullptr, false, ""); });
    stamped("256x64 bf16x6 no split", [&](const TnArgs& b) { run_x6p<true, 10>(b, 256, 3, nullptr, false, ""); });
    stamped("256x64 bf16x6 R + M only", [&](const TnArgs& b) { run_x6p<true, 13>(b, 256, 3, nullptr, false, ""); });
    return 0;
}
