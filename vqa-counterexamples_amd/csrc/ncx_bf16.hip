// ncx_bf16.hip -- bf16-operand GEMMs of the NCX_F_BF16 variant (see ncx_bf16.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ncx_bf16.h"
#include "ncx_internal.h"

namespace ncx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 to_bf16(float x) { return __builtin_bit_cast(u16, (__bf16)x); }   // RNE (v_cvt_pk_bf16_f32)

// ---- weight pack: Wc[h][c] ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_wc(ncx_dims d, Bf16Cols cc, SegOffsets o, const float* __restrict__ w1,
                                                 const float* __restrict__ gt, u16* __restrict__ wc, int Hp) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Hp * cc.kc) return;
    const int h = (int)(i / cc.kc), c = (int)(i - (long long)h * cc.kc);
    float v = 0.f;
    if (h < d.H && c < cc.raw) {
        const float* row = w1 + (long long)h * o.din;
        if (c < cc.c_vm) { if (c < d.dv) v = row[o.v_other + c]; }
        else if (c < cc.c_misc) { if (c - cc.c_vm < d.dv) v = row[o.v_mult + (c - cc.c_vm)]; }
        else if (c < cc.c_z) { if (c - cc.c_misc < d.K + 1) v = row[o.v_dist + (c - cc.c_misc)]; }   // v_dist column, then the K v_rank columns
        else if (c < cc.c_p) { if (c - cc.c_z < d.dz) v = row[o.z_other + (c - cc.c_z)]; }
        else v = gt[(long long)h * d.A + (c - cc.c_p)];
    }
    wc[i] = to_bf16(v);
}

// 4 columns per thread (H % 4 == 0 and Hp % 4 == 0: one 16-byte load, one 8-byte store), else one
template <bool VEC>
__global__ __launch_bounds__(256) void k_dpre_to_bf16(const float* __restrict__ x, int M, int H, int Hp, u16* __restrict__ y) {
    constexpr int W = VEC ? 4 : 1;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * W;
    if (i >= (long long)M * Hp) return;
    const int r = (int)(i / Hp), c = (int)(i - (long long)r * Hp);
    if (VEC) {
        typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < H) v = *(const f32x4*)(x + (long long)r * H + c);
        *(u16x4*)(y + i) = u16x4{to_bf16(v[0]), to_bf16(v[1]), to_bf16(v[2]), to_bf16(v[3])};
    } else {
        y[i] = to_bf16(c < H ? x[(long long)r * H + c] : 0.f);
    }
}

// ---- NT: C[M, N] = A[M, Kc] . B[N, Kc]^T, both row-major bf16 (reduction along the contiguous dimension) ---------
// 256 threads = 2 x 2 waves, wave tile (BM/2) x (BN/2) in 16x16 blocks, K-step 64 (two v_mfma_f32_16x16x32_bf16 per
// block), register-staged double-buffered LDS with TWO K-steps of global loads in flight per workgroup (the product
// is bound by operand delivery, not by the MFMA pipe).  LDS rows are padded to 144 B: ds_read_b128 conflict-free.
constexpr int NT_PITCH = 144;                 // bytes per LDS row: 64 bf16 + 16 B pad
template <int BM, int BN>
__global__ __launch_bounds__(256, (BM * BN <= 64 * 64) ? 4 : (BM * BN <= 64 * 128) ? 3 : 2) void gemm_bf16_nt_kernel(const u16* __restrict__ A, int M, const u16* __restrict__ B, int N,
                                                              int Kc, float* __restrict__ out, long long ldo, const EpiArgs epi) {
    constexpr int NA = BM / 32, NB = BN / 32, WM = BM / 32, WN = BN / 32;
    constexpr int A_BYTES = BM * NT_PITCH, B_BYTES = BN * NT_PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf16[];
    unsigned char* const lds_a = smem_bf16;                    // [2][A_BYTES]
    unsigned char* const lds_b = smem_bf16 + 2 * A_BYTES;      // [2][B_BYTES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const WgMap wmap{(M + BM - 1) / BM, (N + BN - 1) / BN, 1};
    int tm, tn, z;
    wmap.decode(blockIdx.x, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int lr = tid >> 3, lc = tid & 7;                     // loader: row within a 32-row slab, 16-byte chunk of the 128-byte k-window
    const u16* pa[NA];
    const u16* pb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[i] = A + (long long)min(m0 + lr + 32 * i, M - 1) * Kc + 8 * lc;
#pragma unroll
    for (int i = 0; i < NB; ++i) pb[i] = B + (long long)min(n0 + lr + 32 * i, N - 1) * Kc + 8 * lc;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    auto issue = [&](u32x4 (&ra)[NA], u32x4 (&rb)[NB], int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *(const u32x4*)(pa[i] + (long long)kt * 64);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *(const u32x4*)(pb[i] + (long long)kt * 64);
    };
    auto stash = [&](const u32x4 (&ra)[NA], const u32x4 (&rb)[NB], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *(u32x4*)(lds_a + buf * A_BYTES + (lr + 32 * i) * NT_PITCH + lc * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *(u32x4*)(lds_b + buf * B_BYTES + (lr + 32 * i) * NT_PITCH + lc * 16) = rb[i];
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* ta = lds_a + buf * A_BYTES + (wm0 + li) * NT_PITCH + lk * 16;
        const unsigned char* tb = lds_b + buf * B_BYTES + (wn0 + li) * NT_PITCH + lk * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const bf16x8*)(ta + i * 16 * NT_PITCH + s * 64);
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = *(const bf16x8*)(tb + j * 16 * NT_PITCH + s * 64);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // Kc is a multiple of 128: an even number of K-steps.  Every load and LDS store of the loop is UNCONDITIONAL (the
    // K-step index is clamped, the surplus tiles of the last iteration are never read): a branch around the loads made
    // the compiler drain vmcnt to 0 before issuing them, i.e. one K-step in flight and the full latency exposed.
    const int nk = Kc / 64;
    issue(ra0, rb0, 0);
    issue(ra1, rb1, 1);
    stash(ra0, rb0, 0);
    __syncthreads();
    for (int t = 0; t < nk; t += 2) {
        // (order left to the compiler: pinned as issue | compute | stash -- loads two K-steps ahead, all LDS stores at the end of the
        // step -- this kernel ran 296 us instead of 263 at the configs[4] shape: the stores must spread under the MFMAs)
        issue(ra0, rb0, min(t + 2, nk - 1));
        compute(0);
        stash(ra1, rb1, 1);
        __syncthreads();
        issue(ra1, rb1, min(t + 3, nk - 1));
        compute(1);
        stash(ra0, rb0, 0);
        __syncthreads();
    }
    // row-block loop kept rolled (compile-time-indexed selects pick the block's accumulators): fully unrolled, the 64
    // inlined epilogues exceed the unroller's size limit and the accumulator array ends up in scratch
#pragma unroll 1
    for (int i = 0; i < WM; ++i) {
        f32x4 row[WN];
#pragma unroll
        for (int ii = 0; ii < WM; ++ii)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (ii == 0) row[j] = acc[0][j];
                else row[j] = (ii == i) ? acc[ii][j] : row[j];
            }
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = m0 + wm0 + 16 * i + 4 * lk + q, n = n0 + wn0 + 16 * j + li;
                if (r < M && n < N) out[(long long)r * ldo + n] = apply_epilogue(epi, row[j][q], r, n, N);
            }
    }
}

// ---- NT on 192 x 256 tiles, LDS-DMA staged (round 3; BASELINE configs[4]: M = 49 152 rows, N = H = 256) --------------------
// At configs[4] the forward product streams 642 MB of packed rows for 164 GF: at the bf16 MFMA rate that is an HBM-bound
// stream (107 us at 6 TB/s against 66 us of MFMA time).  The 128 x 128 kernel above moves every packed row through L2 -> LDS
// twice (two column tiles) and the weights 768 times: 2.6 GB of L2 -> LDS traffic, delivered at ~10 TB/s = 264 us.  Here one
// workgroup owns 192 rows x ALL 256 columns (256 workgroups = one per CU at configs[4]): every packed row is read once, 1.5 GB
// of L2 -> LDS traffic.  512 threads = 8 waves as 2 (rows) x 4 (columns), wave tile 96 x 64 = 6 x 4 blocks of
// v_mfma_f32_16x16x32_bf16, K-step 64.  Staging is LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass), 8-row x 128-B
// pieces; the LDS image is linear per piece and XOR-swizzled through the SOURCE address (16-byte chunk c of tile row r sits at
// chunk c ^ ((r >> 1) & 7): conflict-free ds_read_b128 for the hardware's 16-lane groups).  The HBM operand (A) runs a ring of
// three slots = two K-steps in flight, the L2-resident weights (B) a ring of two; the DMA of step t+2 / t+1 stays in flight
// across the barrier of step t (counted s_waitcnt vmcnt(3) + raw s_barrier: __syncthreads would drain it).
constexpr int N8_BM = 192, N8_BN = 256, N8_A = N8_BM * 128, N8_BH = N8_BN * 64, N8_LDS = 3 * N8_A + 4 * N8_BH;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// ... with the non-temporal hint: bytes that ONE workgroup reads ONCE (the packed rows) must not push the weights every CU
// re-reads out of the 4 MB L2
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Rings: A (the HBM operand) three K-step slots of 192 rows x 128 B: A(t+2) is issued when step t starts, two steps ahead;
// B (the weights, L2-resident but 32 KB per K-step for every CU) FOUR half-step slots of 256 rows x 64 B (k32 halves): half h+3
// is issued when half h starts (a ring of two whole K-step slots could only run one step ahead -- the slot of step t+1 is being
// read until step t-1 ends -- and every K-step then waited ~2 us for its weights: 214 us).  A step is two phases (k32 halves), each
// closed by a counted wait + raw barrier; DMA issue order  ... Bh(odd), A, Bh(even), Bh(odd), A ...  makes the counts 10 / 7.
// Measured at configs[4] (M = 49 152, Kc = 6 528; tools: NCX_NT8_VAR timing ablations of round 3, since removed): 128 x 128 kernel 265 us;
// this kernel 215-228 us, 203 with the non-temporal hint on the packed rows (they are read once by one CU and otherwise push the
// 3.3 MB of weights out of the 4 MB L2s).  What it waits for is data delivery, not the matrix pipe: without any MFMA it takes the
// same time; with the rows coming from L2 (every workgroup streaming the same block) 165-170 us = 1.6 us per 56 KB K-step = 35 GB/s
// per CU; the rows addressed as if the pack were stored tile by tile (1 KiB contiguous per DMA instruction) 189 us (not adopted: the
// pack and the TN kernel would have to follow).
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt8_kernel(const u16* __restrict__ A, int M, const u16* __restrict__ B, int Nrows,
                                                               int N, int Kc, float* __restrict__ out, long long ldo, const EpiArgs epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_bf16[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 2) * 96, wn0 = (wave & 3) * 64;
    const int m0 = blockIdx.x * N8_BM;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_bf16;   // LDS byte address of the A ring (the B ring follows)
    // loader roles.  A: this wave's pieces are tile rows 8 (wave + 8 i) .. + 7 (i < 3), lane -> row lane >> 3, physical chunk lane & 7 of
    // the 128-B row, holding logical chunk pc ^ ((row >> 1) & 7).  B halves: pieces of 16 rows x 64 B, rows 16 (2 wave + i) .. + 15
    // (i < 2), lane -> row lane >> 2, physical chunk lane & 3, holding logical chunk pc ^ g(row), g = (-(row >> 2)) & 3.
    const u16* srcA[3]; const u16* srcB[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int r = 8 * (wave + 8 * i) + (lane >> 3); srcA[i] = A + (long long)min(m0 + r, M - 1) * Kc + 8 * ((lane & 7) ^ ((r >> 1) & 7)); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int r = 16 * (2 * wave + i) + (lane >> 2); srcB[i] = B + (long long)min(r, Nrows - 1) * Kc + 8 * ((lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3)); }
    const int nk = Kc / 64;
    auto issueA = [&](int kt) __attribute__((always_inline)) {
        const int k = min(kt, nk - 1);
        const unsigned dst = lds0 + (unsigned)(kt % 3) * N8_A + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16_nt(srcA[i] + (long long)k * 64, dst + (unsigned)i * 8192u);
    };
    auto issueBh = [&](int h) __attribute__((always_inline)) {
        const int k = min(h, 2 * nk - 1);
        const unsigned dst = lds0 + 3u * N8_A + (unsigned)(h & 3) * N8_BH + (unsigned)wave * 2048u;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(srcB[i] + (long long)k * 32, dst + (unsigned)i * 1024u);
    };
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swa = li >> 1;                                                      // (row >> 1) & 7 of this lane's A fragment rows
    const int chb = (lk ^ ((4 - (li >> 2)) & 3)) * 16;                            // this lane's physical chunk in a B half row
    auto phase = [&](int kt, int s) __attribute__((always_inline)) {             // the k32 half s of K-step kt
        const unsigned char* ta = smem_bf16 + (kt % 3) * N8_A + (wm0 + li) * 128 + ((4 * s + lk) ^ swa) * 16;
        const unsigned char* tb = smem_bf16 + 3 * N8_A + ((2 * kt + s) & 3) * N8_BH + (wn0 + li) * 64 + chb;
        bf16x8 af[6], bf[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) af[i] = *(const bf16x8*)(ta + i * 16 * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *(const bf16x8*)(tb + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    issueBh(0); issueA(0); issueBh(1); issueA(1); issueBh(2);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                              // Bh(0), A(0) landed
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nk; ++t) {
        issueBh(2 * t + 3);
        issueA(t + 2);
        phase(t, 0);
        asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // Bh(2t+1) landed
        __builtin_amdgcn_s_barrier();
        issueBh(2 * t + 4);
        phase(t, 1);
        asm volatile("s_waitcnt vmcnt(7)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // Bh(2t+2), A(t+1) landed
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue (row-block loop kept rolled, as in the kernel above)
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        f32x4 row[4];
#pragma unroll
        for (int ii = 0; ii < 6; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ii == 0) row[j] = acc[0][j];
                else row[j] = (ii == i) ? acc[ii][j] : row[j];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = m0 + wm0 + 16 * i + 4 * lk + q, n = wn0 + 16 * j + li;
                if (r < M && n < N) out[(long long)r * ldo + n] = apply_epilogue(epi, row[j][q], r, n, N);
            }
    }
}

// ---- TN: C[Hm, Kc] (+)= sum over rows k of A[k][m] . B[k][n], both row-major bf16 with the reduction along ROWS --------
// The LDS image keeps the global layout ([64 k-rows][128 columns], rows padded to 288 B) and ds_read_b64_tr_b16 delivers
// the MFMA operands transposed: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block and
// receives column (lane & 15) of the 4 rows.  Group g takes rows 4g..4g+3 and 16+4g..16+4g+3 of each 32-row sub-step
// for BOTH operands (same k order on both sides; the two blocks of a 32-lane half fall on disjoint banks).
// Workgroup id -> k-chunk z = id % 8 (one chunk per XCD: its slice of A stays in that L2), tile = id / 8, m-tiles adjacent.
constexpr int TN_PITCH = 288;
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(const u16* __restrict__ A, int lda, const u16* __restrict__ B, int ldb,
                                                              int rows, int chunk_rows, int tiles_m, int S, float* __restrict__ slab,
                                                              int H, int Kc) {
    constexpr int WM = BM / 32, WN = BN / 32;
    constexpr int A_BYTES = 64 * TN_PITCH, B_BYTES = 64 * TN_PITCH;
    static_assert(BM == 128 && BN == 128, "loader mapping assumes 128-column tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf16[];
    unsigned char* const lds_a = smem_bf16;
    unsigned char* const lds_b = smem_bf16 + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int z = blockIdx.x % S, t = blockIdx.x / S;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int k0 = z * chunk_rows, k1 = min(k0 + chunk_rows, rows);
    if (k0 >= rows) return;                                      // empty chunk (uniform per workgroup)
    const int lr = tid >> 4, lc = tid & 15;                      // loader: row within a 16-row slab, 16-byte chunk of the 256-byte row
    const u16* pa = A + m0 + 8 * lc;
    const u16* pb = B + n0 + 8 * lc;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    unsigned km0[4], km1[4];
    auto issue_m = [&](u32x4 (&ra)[4], u32x4 (&rb)[4], unsigned (&km)[4], int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kt * 64 + lr + 16 * i, kc = min(k, rows - 1);
            const unsigned keep = k < k1 ? 0xFFFFFFFFu : 0u;             // (AND mask at store time: no branch around the load)
            ra[i] = *(const u32x4*)(pa + (long long)kc * lda);
            rb[i] = *(const u32x4*)(pb + (long long)kc * ldb);
            km[i] = keep;
        }
    };
    auto stash_m = [&](const u32x4 (&ra)[4], const u32x4 (&rb)[4], const unsigned (&km)[4], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(lds_a + buf * A_BYTES + (lr + 16 * i) * TN_PITCH + lc * 16) = ra[i] & km[i];
            *(u32x4*)(lds_b + buf * B_BYTES + (lr + 16 * i) * TN_PITCH + lc * 16) = rb[i] & km[i];
        }
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;                          // transposed-read role of this lane inside its 16-lane group
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* ta = lds_a + buf * A_BYTES + (4 * lk + tq) * TN_PITCH + (wm0 + 4 * tp) * 2;
        const unsigned char* tb = lds_b + buf * B_BYTES + (4 * lk + tq) * TN_PITCH + (wn0 + 4 * tp) * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const unsigned char* p = ta + s * 32 * TN_PITCH + i * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 16 * TN_PITCH));
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const unsigned char* p = tb + s * 32 * TN_PITCH + j * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 16 * TN_PITCH));
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    // K-steps rounded up to an even count: rows at or beyond k1 load as zeros, so a surplus step adds nothing; all loads
    // and LDS stores are unconditional (see the NT kernel)
    const int nk = ((k1 - k0 + 63) / 64 + 1) & ~1;
    issue_m(ra0, rb0, km0, 0);
    issue_m(ra1, rb1, km1, 1);
    stash_m(ra0, rb0, km0, 0);
    __syncthreads();
    for (int t2 = 0; t2 < nk; t2 += 2) {
        issue_m(ra0, rb0, km0, t2 + 2);
        __builtin_amdgcn_sched_barrier(0);        // keeps the loads ahead of the MFMA block (measured: 198 -> 186 us; the NT kernel lost with it)
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        stash_m(ra1, rb1, km1, 1);
        __syncthreads();
        issue_m(ra1, rb1, km1, t2 + 3);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        stash_m(ra0, rb0, km0, 0);
        __syncthreads();
    }
    float* dst = slab + (long long)z * H * Kc;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + wm0 + 16 * i + 4 * lk + q;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + wn0 + 16 * j + li;
                if (m < H && n < Kc) dst[(long long)m * Kc + n] = acc[i][j][q];
            }
        }
}

// ---- TN on 256 x 128 tiles (round 3; BASELINE configs[4]) -----------------------------------------------------------------------
// The 128 x 128 kernel above reads every packed row TWICE (two row tiles for H = 256) and runs 816 workgroups in 1.6 rounds of the
// two-per-CU slots.  Here one workgroup owns ALL 256 rows of the output x 128 columns: the packed rows are read once, 51 column
// tiles x 5 k-chunks = 255 workgroups = one round at one per CU, 5 slabs to reduce instead of 8.  512 threads = 8 waves as 4 (rows)
// x 2 (columns), the same 64 x 64 wave tile and transposed LDS reads as above; the dpre tile's 512-byte rows are padded to 544 B
// (= 32 mod 256, the same bank pattern as the 288-byte rows).
constexpr int TN8_PA = 544, TN8_PB = 288, TN8_LDS = 2 * 64 * (TN8_PA + TN8_PB);
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn8_kernel(const u16* __restrict__ A, int lda, const u16* __restrict__ B, int ldb,
                                                               int rows, int chunk_rows, int S, float* __restrict__ slab, int H, int Kc) {
    constexpr int A_BYTES = 64 * TN8_PA, B_BYTES = 64 * TN8_PB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_bf16[];
    unsigned char* const lds_a = smem_bf16;
    unsigned char* const lds_b = smem_bf16 + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int z = blockIdx.x % S, tn = blockIdx.x / S;
    const int n0 = tn * 128;
    const int k0 = z * chunk_rows, k1 = min(k0 + chunk_rows, rows);
    if (k0 >= rows) return;                                      // empty chunk (uniform per workgroup)
    // loader: A tile 64 rows x 32 chunks of 16 B (4 per thread: rows ar + 16 i), B tile 64 rows x 16 chunks (2 per thread: rows br + 32 i)
    const int ar = tid >> 5, ac = tid & 31, br = tid >> 4, bc = tid & 15;
    const u16* pa = A + 8 * ac;
    const u16* pb = B + n0 + 8 * bc;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ra0[4], rb0[2], ra1[4], rb1[2];
    unsigned ka0[4], kb0[2], ka1[4], kb1[2];
    auto issue_m = [&](u32x4 (&ra)[4], u32x4 (&rb)[2], unsigned (&ka)[4], unsigned (&kb)[2], int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + kt * 64 + ar + 16 * i;
            ra[i] = *(const u32x4*)(pa + (long long)min(k, rows - 1) * lda);
            ka[i] = k < k1 ? 0xFFFFFFFFu : 0u;                   // (AND mask at store time: no branch around the load)
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + kt * 64 + br + 32 * i;
            rb[i] = __builtin_nontemporal_load((const u32x4*)(pb + (long long)min(k, rows - 1) * ldb));     // (the packed rows: read once)
            kb[i] = k < k1 ? 0xFFFFFFFFu : 0u;
        }
    };
    auto stash_m = [&](const u32x4 (&ra)[4], const u32x4 (&rb)[2], const unsigned (&ka)[4], const unsigned (&kb)[2], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(lds_a + buf * A_BYTES + (ar + 16 * i) * TN8_PA + ac * 16) = ra[i] & ka[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(lds_b + buf * B_BYTES + (br + 32 * i) * TN8_PB + bc * 16) = rb[i] & kb[i];
    };
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int tq = li >> 2, tp = li & 3;                          // transposed-read role of this lane inside its 16-lane group
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* ta = lds_a + buf * A_BYTES + (4 * lk + tq) * TN8_PA + (wm0 + 4 * tp) * 2;
        const unsigned char* tb = lds_b + buf * B_BYTES + (4 * lk + tq) * TN8_PB + (wn0 + 4 * tp) * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned char* p = ta + s * 32 * TN8_PA + i * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 16 * TN8_PA));
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char* p = tb + s * 32 * TN8_PB + j * 32;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 16 * TN8_PB));
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = ((k1 - k0 + 63) / 64 + 1) & ~1;               // even; rows at or beyond k1 load as zeros (see the kernel above)
    issue_m(ra0, rb0, ka0, kb0, 0);
    issue_m(ra1, rb1, ka1, kb1, 1);
    stash_m(ra0, rb0, ka0, kb0, 0);
    __syncthreads();
    for (int t2 = 0; t2 < nk; t2 += 2) {
        issue_m(ra0, rb0, ka0, kb0, t2 + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        stash_m(ra1, rb1, ka1, kb1, 1);
        __syncthreads();
        issue_m(ra1, rb1, ka1, kb1, t2 + 3);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        stash_m(ra0, rb0, ka0, kb0, 0);
        __syncthreads();
    }
    float* dst = slab + (long long)z * H * Kc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = wm0 + 16 * i + 4 * lk + q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn0 + 16 * j + li;
                if (m < H && n < Kc) dst[(long long)m * Kc + n] = acc[i][j][q];
            }
        }
}

// dWc[h][c] = sum of the non-empty k-chunk slabs in fixed order, scattered to d linear_1.weight / dGt
__global__ __launch_bounds__(256) void k_bf16_reduce_dwc(ncx_dims d, Bf16Cols cc, SegOffsets o, const float* __restrict__ slab, int nz,
                                                         float* __restrict__ g_w1, float* __restrict__ dgt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)d.H * cc.raw) return;
    const int h = (int)(i / cc.raw), c = (int)(i - (long long)h * cc.raw);
    float v = 0.f;
    for (int z = 0; z < nz; ++z) v += slab[((long long)z * d.H + h) * cc.kc + c];
    float* row = g_w1 + (long long)h * o.din;
    if (c < cc.c_vm) { if (c < d.dv) row[o.v_other + c] = v; }
    else if (c < cc.c_misc) { if (c - cc.c_vm < d.dv) row[o.v_mult + (c - cc.c_vm)] = v; }
    else if (c < cc.c_z) { if (c - cc.c_misc < d.K + 1) row[o.v_dist + (c - cc.c_misc)] = v; }
    else if (c < cc.c_p) { if (c - cc.c_z < d.dz) row[o.z_other + (c - cc.c_z)] = v; }
    else dgt[(long long)h * d.A + (c - cc.c_p)] = v;
}

// ---- host side -------------------------------------------------------------------------------------------------
int bf16_pack_wc(const ncx_dims& d, const float* w1, const float* gt, u16* wc, hipStream_t s) {
    const Bf16Cols cc = bf16_cols(d);
    const int Hp = (d.H + 127) / 128 * 128;
    const long long n = (long long)Hp * cc.kc;
    hipLaunchKernelGGL(k_pack_wc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, cc, seg_offsets(d), w1, gt, wc, Hp);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

template <int BM, int BN>
static int launch_bf16_nt(const u16* xc, int M, const u16* wc, int N, int kc, const EpiArgs& epi, float* out, hipStream_t s,
                          long long ldo = 0) {
    const int lds = 2 * (BM + BN) * NT_PITCH;
    static DevMask attr{0};
    NCX_HIP_TRY(set_max_lds_once(attr, (const void*)gemm_bf16_nt_kernel<BM, BN>, lds));
    const int wgs = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN>), dim3(wgs), dim3(256), lds, s, xc, M, wc, N, kc, out, ldo ? ldo : (long long)N, epi);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

int bf16_main_forward(const ncx_dims& d, const u16* xc, const u16* wc, const EpiArgs& epi, float* h1, hipStream_t s) {
    const Bf16Cols cc = bf16_cols(d);
    const int M = d.B * d.K;
    // The product is bound by operand delivery (global-load latency x bytes in flight), not by the MFMA pipe: prefer the
    // tile that puts >= 2 workgroups on every CU.  NCX_BF16_NT_CFG (0..3) overrides for experiments.
    int cfg = -1;
    { const char* e = hook_env("NCX_BF16_NT_CFG"); if (e) cfg = atoi(e); }
    // one 192 x 256 workgroup per CU (LDS-DMA staged): all of H in one tile, >= 3/4 of the CUs busy
    if ((cfg < 0 || cfg == 8) && d.H <= N8_BN && (d.H + 127) / 128 * 128 == N8_BN && (cfg == 8 || (long long)(M + N8_BM - 1) / N8_BM * 4 >= 3LL * num_cus())) {
        static DevMask attr8{0};
        NCX_HIP_TRY(set_max_lds_once(attr8, (const void*)gemm_bf16_nt8_kernel, N8_LDS));
        hipLaunchKernelGGL(gemm_bf16_nt8_kernel, dim3((M + N8_BM - 1) / N8_BM), dim3(512), N8_LDS, s, xc, M, wc, N8_BN, d.H, cc.kc, h1, (long long)d.H, epi);
        NCX_HIP_TRY(hipGetLastError());
        return NCX_OK;
    }
    if (cfg < 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((d.H + 127) / 128);
        cfg = t128 >= 2LL * num_cus() ? 0 : 3;
    }
    switch (cfg) {
    case 0:  return launch_bf16_nt<128, 128>(xc, M, wc, d.H, cc.kc, epi, h1, s);
    case 1:  return launch_bf16_nt<64, 128>(xc, M, wc, d.H, cc.kc, epi, h1, s);
    case 2:  return launch_bf16_nt<128, 64>(xc, M, wc, d.H, cc.kc, epi, h1, s);
    default: return launch_bf16_nt<64, 64>(xc, M, wc, d.H, cc.kc, epi, h1, s);
    }
}

int bf16_dw1c(const ncx_dims& d, const float* dpre, u16* dpre_bf, const u16* xc, float* slab, float* g_w1, float* dgt,
              hipStream_t s) {
    const Bf16Cols cc = bf16_cols(d);
    const int M = d.B * d.K, Hp = (d.H + 127) / 128 * 128;
    {
        const long long n = (long long)M * Hp;
        if (d.H % 4 == 0 && (((uintptr_t)dpre | (uintptr_t)dpre_bf) & 15) == 0)
            hipLaunchKernelGGL(k_dpre_to_bf16<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, dpre, M, d.H, Hp, dpre_bf);
        else
            hipLaunchKernelGGL(k_dpre_to_bf16<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dpre, M, d.H, Hp, dpre_bf);
        NCX_HIP_TRY(hipGetLastError());
    }
    int nz;
    // all 256 output rows in one workgroup (the packed rows read once), S k-chunks so that tiles x S fills one round of the CUs
    const int tiles_n8 = cc.kc / 128;
    int S8 = num_cus() / tiles_n8; if (S8 > BF16_SPLIT) S8 = BF16_SPLIT;
    const bool tn8 = Hp == 256 && S8 >= 2 && ((long long)M >= 64LL * 16 * S8 || hook_env("NCX_BF16_TN8")) && !hook_env("NCX_BF16_NO_TN8");
    if (tn8) {
        static DevMask attr8{0};
        NCX_HIP_TRY(set_max_lds_once(attr8, (const void*)gemm_bf16_tn8_kernel, TN8_LDS));
        const int chunk = (int)(((long long)(M + S8 - 1) / S8 + 63) / 64 * 64);
        nz = (M + chunk - 1) / chunk;
        hipLaunchKernelGGL(gemm_bf16_tn8_kernel, dim3(tiles_n8 * S8), dim3(512), TN8_LDS, s, (const u16*)dpre_bf, Hp, xc, cc.kc, M, chunk, S8, slab, d.H, cc.kc);
        NCX_HIP_TRY(hipGetLastError());
    } else {
    constexpr int BM = 128, BN = 128;
    const int lds = 4 * 64 * TN_PITCH;
    static DevMask attr{0};
    NCX_HIP_TRY(set_max_lds_once(attr, (const void*)gemm_bf16_tn_kernel<BM, BN>, lds));
    const int chunk = (int)(((long long)(M + BF16_SPLIT - 1) / BF16_SPLIT + 63) / 64 * 64);
    nz = (M + chunk - 1) / chunk;
    const int tiles_m = Hp / BM, tiles_n = cc.kc / BN;
    hipLaunchKernelGGL((gemm_bf16_tn_kernel<BM, BN>), dim3(tiles_m * tiles_n * BF16_SPLIT), dim3(256), lds, s, (const u16*)dpre_bf, Hp,
                       xc, cc.kc, M, chunk, tiles_m, BF16_SPLIT, slab, d.H, cc.kc);
    NCX_HIP_TRY(hipGetLastError());
    }
    const long long n = (long long)d.H * cc.raw;
    hipLaunchKernelGGL(k_bf16_reduce_dwc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, cc, seg_offsets(d), (const float*)slab, nz,
                       g_w1, dgt);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

// ---- the answer-embedding products of the bf16 variant ------------------------------------------------------------
// dst[r][c] = bf16(src[r*ld + c]) for r < R, c < C; zero elsewhere in the [Rp][Cp] image
template <bool VEC>      // VEC: 4 columns per thread (C, Cp, ld multiples of 4, 16-byte aligned source rows)
__global__ __launch_bounds__(256) void k_pack2d(const float* __restrict__ src, long long ld, int R, int C, u16* __restrict__ dst, int Rp, int Cp) {
    constexpr int W = VEC ? 4 : 1;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * W;
    if (i >= (long long)Rp * Cp) return;
    const int r = (int)(i / Cp), c = (int)(i - (long long)r * Cp);
    if (VEC) {
        typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R && c < C) v = *(const f32x4*)(src + (long long)r * ld + c);
        *(u16x4*)(dst + i) = u16x4{to_bf16(v[0]), to_bf16(v[1]), to_bf16(v[2]), to_bf16(v[3])};
    } else {
        dst[i] = to_bf16(r < R && c < C ? src[(long long)r * ld + c] : 0.f);
    }
}
// dst[c][r] = bf16(src[r][c]) (transposed image [C][Rp], zero for r >= R); 32x32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void k_pack_transposed(const float* __restrict__ src, int R, int C, u16* __restrict__ dst, int Rp) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(long long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < Rp) dst[(long long)c * Rp + r] = to_bf16(tile[tx][j]);
    }
}

static int pack2d(const float* src, long long ld, int R, int C, u16* dst, int Rp, int Cp, hipStream_t s) {
    const long long n = (long long)Rp * Cp;
    if (C % 4 == 0 && Cp % 4 == 0 && ld % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
        hipLaunchKernelGGL(k_pack2d<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, src, ld, R, C, dst, Rp, Cp);
    else
        hipLaunchKernelGGL(k_pack2d<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld, R, C, dst, Rp, Cp);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

int bf16_pack_embedding(const ncx_dims& d, const float* E, const float* w1, const Bf16Emb& m, hipStream_t s) {
    const SegOffsets o = seg_offsets(d);
    int rc = pack2d(E, d.da, d.A, d.da, m.e_bf, d.A, m.dap, s); if (rc) return rc;                       // E       [A][dap]
    hipLaunchKernelGGL(k_pack_transposed, dim3((unsigned)((d.da + 31) / 32), (unsigned)((m.Ap + 31) / 32)), dim3(256), 0, s,
                       E, d.A, d.da, m.et_bf, m.Ap);                                                      // E^T     [da][Ap]
    NCX_HIP_TRY(hipGetLastError());
    rc = pack2d(w1 + o.a_other, o.din, d.H, d.da, m.w1a_bf, d.H, m.dap, s); if (rc) return rc;            // W1ak    [H][dap]
    return pack2d(w1 + o.a_gt, o.din, d.H, d.da, m.w1a_bf + (long long)d.H * m.dap, d.H, m.dap, s);       // W1agt   [H][dap] (stacked below)
}

// Gt[H, A] = bf16(W1ak) . bf16(E)^T
int bf16_gt(const ncx_dims& d, const Bf16Emb& m, float* gt, hipStream_t s) {
    EpiArgs e{};
    return launch_bf16_nt<64, 64>(m.w1a_bf, d.H, m.e_bf, d.A, m.dap, e, gt, s, d.A);
}

// dW1[:, a_other] = bf16(dGt) . bf16(E)
int bf16_dw1ak(const ncx_dims& d, const Bf16Emb& m, const float* dgt_dagt, float* g_w1, hipStream_t s) {
    const SegOffsets o = seg_offsets(d);
    int rc = pack2d(dgt_dagt, d.A, 2 * d.H, d.A, m.dg_bf, 2 * d.H, m.Ap, s); if (rc) return rc;             // [dGt; dGgt]  [2H][Ap]
    EpiArgs e{};
    return launch_bf16_nt<64, 64>(m.dg_bf, d.H, m.et_bf, d.da, m.Ap, e, g_w1 + o.a_other, s, o.din);
}

// dE = bf16([dGt; dGgt])^T . bf16([W1ak; W1agt])   (one 2H-deep reduction instead of two chained H-deep ones)
int bf16_de(const ncx_dims& d, const Bf16Emb& m, const float* dgt_dagt, float* g_E, hipStream_t s) {
    int rc = pack2d(dgt_dagt, d.A, 2 * d.H, d.A, m.dg_bf, 2 * d.H, m.Ap, s); if (rc) return rc;
    constexpr int BM = 128, BN = 128;
    const int lds = 4 * 64 * TN_PITCH;
    static DevMask attr{0};
    NCX_HIP_TRY(set_max_lds_once(attr, (const void*)gemm_bf16_tn_kernel<BM, BN>, lds));
    const int rows = 2 * d.H, tiles_m = m.Ap / BM, tiles_n = m.dap / BN;
    hipLaunchKernelGGL((gemm_bf16_tn_kernel<BM, BN>), dim3(tiles_m * tiles_n), dim3(256), lds, s, (const u16*)m.dg_bf, m.Ap,
                       (const u16*)m.w1a_bf, m.dap, rows, (rows + 63) / 64 * 64, tiles_m, 1, g_E, d.A, d.da);
    NCX_HIP_TRY(hipGetLastError());
    return NCX_OK;
}

}  // namespace ncx
