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

int main(int argc, char** argv) {
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
        printf("   check %-24s max |err| %.3e (max |ref| %.3e)\n", name, worst, mag);
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
    run_d<true, 0, true>(a, 256, reps, dout, true, "256x64  LDS-DMA (nt X)"); check("256x64 dma");
    run_d<true, 0, false>(a, 256, reps, nullptr, true, "256x64  LDS-DMA (no nt hint)");
    run_d<true, 1, true>(a, 256, reps, nullptr, true, "256x64  LDS-DMA ABL no mfma");
    run_d<true, 2, true>(a, 256, reps, nullptr, true, "256x64  LDS-DMA ABL no loads");
    run_d<false, 0, true>(a, 256, reps, nullptr, true, "256x64  LDS-DMA plain operand");
    // stamps: per-workgroup loop / epilogue cycles and the held clock
    {
        unsigned long long* st; CHECK(hipMalloc(&st, 256 * 8 * 8)); CHECK(hipMemset(st, 0, 256 * 8 * 8));
        TnArgs b = a; b.stamps = st;
        run<64, true, 0>(b, 256, 3, nullptr, false, "");
        std::vector<unsigned long long> hs(256 * 8);
        CHECK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> loop, epi, mhz, dur;
        for (int w = 0; w < 256; ++w) { const auto* s = &hs[w * 8]; if (s[7] > s[6]) { loop.push_back((double)(s[1] - s[0])); epi.push_back((double)(s[2] - s[1])); mhz.push_back((double)(s[2] - s[0]) / (s[7] - s[6]) * 100.0); dur.push_back((s[7] - s[6]) / 100.0); } }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, x); return m; };
        printf("stamps 256x64: loop cycles med %.0f max %.0f | epilogue med %.0f | sclk med %.0f MHz | workgroup us med %.1f max %.1f\n", med(loop), mx(loop), med(epi), med(mhz), med(dur), mx(dur));
        CHECK(hipFree(st));
    }
    return 0;
}
