// mb_gemm.hip -- structure microbenchmark for the fp32-MFMA forward GEMM of linear_1 (NT form: C[M,N] = A[M,K] . B[N,K]^T).
// Not part of the library: explores tile / wave / occupancy / pipeline choices at the configs[1] shape before they go into
// csrc/ncx_main.hip.   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mb/mb_gemm.hip -o tools/mb/mb_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// BM x BN tile, WGM x WGN waves, BK-deep k-steps, OCC workgroups per CU, DEPTH register sets of global loads in flight,
// ROT: MFMAs of the last sub-step are issued after the barrier (cover the first fragment reads of the next tile)
template <int BM, int BN, int WGM, int WGN, int BK, int OCC, int DEPTH, bool ROT>
__global__ __launch_bounds__(64 * WGM * WGN, (OCC * WGM * WGN + 3) / 4) void mb_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                                                      float* __restrict__ C, int M, int N, int K) {
    constexpr int T = 64 * WGM * WGN;
    constexpr int P = BK + 4;
    constexpr int WTM = BM / WGM, WTN = BN / WGN, WM = WTM / 16, WN = WTN / 16;
    constexpr int QPR = BK / 4;                            // f32x4 per tile row
    constexpr int NA = (BM * QPR + T - 1) / T, NB = (BN * QPR + T - 1) / T;
    constexpr int NSUB = BK / 8;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const lds_a = smem;                             // [2][BM][P]
    float* const lds_b = smem + 2 * BM * P;                // [2][BN][P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wm0 = (wave / WGN) * WTM, wn0 = (wave % WGN) * WTN;
    // XCD-aware order: the column tiles of one row tile sit on one XCD (id % 8), next to each other in time
    const int tiles_n = N / BN;
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    const int tn = local % tiles_n, tm = (local / tiles_n) * 8 + xcd;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pa[NA]; const float* pb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) { const int f = min(tid + T * i, BM * QPR - 1); pa[i] = A + (long long)(m0 + f / QPR) * K + 4 * (f % QPR); }
#pragma unroll
    for (int i = 0; i < NB; ++i) { const int f = min(tid + T * i, BN * QPR - 1); pb[i] = B + (long long)(n0 + f / QPR) * K + 4 * (f % QPR); }
    f32x4 ra[DEPTH][NA], rb[DEPTH][NB];

    auto issue = [&](int set, int k) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[set][i] = *(const f32x4*)(pa[i] + k);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[set][i] = *(const f32x4*)(pb[i] + k);
    };
    auto stash = [&](int set, int buf, int i0a, int i1a, int i0b, int i1b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) if (i >= i0a && i < i1a) { const int f = tid + T * i; if ((BM * QPR) % T == 0 || f < BM * QPR) *(f32x4*)(lds_a + buf * BM * P + (f / QPR) * P + 4 * (f % QPR)) = ra[set][i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) if (i >= i0b && i < i1b) { const int f = tid + T * i; if ((BN * QPR) % T == 0 || f < BN * QPR) *(f32x4*)(lds_b + buf * BN * P + (f / QPR) * P + 4 * (f % QPR)) = rb[set][i]; }
    };
    auto read_frags = [&](int buf, int s, f32x2 (&af)[WM], f32x2 (&bf)[WN]) __attribute__((always_inline)) {
        const float* a = lds_a + buf * BM * P + (wm0 + li) * P + 8 * s + 2 * lk;
        const float* b = lds_b + buf * BN * P + (wn0 + li) * P + 8 * s + 2 * lk;
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *(const f32x2*)(a + i * 16 * P);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *(const f32x2*)(b + j * 16 * P);
    };
    auto mfma = [&](const f32x2 (&af)[WM], const f32x2 (&bf)[WN]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    };
    constexpr int NMF = 2 * WM * WN;

    const int nsteps = K / BK;
    // prologue: tile 0 -> LDS, tiles 1 .. DEPTH in flight
    issue(0, 0);
    stash(0, 0, 0, NA, 0, NB);
    if (DEPTH == 2) { issue(1, BK); issue(0, 2 * BK < K ? 2 * BK : 0); }
    else { issue(0, BK < K ? BK : 0); }
    __syncthreads();

    f32x2 afA[WM], bfA[WN], afB[WM], bfB[WN];
    // One k-step.  SET: register set holding tile t+1 (DEPTH 2: t+2 goes into the other one after t+1 was stored... see below)
    auto step = [&](auto first_c, auto par_c, int t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int PAR = decltype(par_c)::value;             // t & 1 (compile time: register sets / buffers statically indexed)
        constexpr int buf = PAR;
        // DEPTH 2: set (t+1)&1 holds tile t+1; set t&1 holds tile t+2 (issued during step t-1 ... or the prologue).
        // DEPTH 1: set 0 holds tile t+1, re-issued for tile t+2 after it has been stored.
        constexpr int sset = DEPTH == 2 ? (PAR ^ 1) : 0;
        const int knext = (t + 1 + DEPTH) * BK < K ? (t + 1 + DEPTH) * BK : 0;       // (clamped: surplus loads are never stored)
        if (ROT) {
            read_frags(buf, 0, afA, bfA);
            if (!FIRST) {
                mfma(afB, bfB);                                      // (t-1, last sub-step): covers the reads above
#pragma unroll
                for (int q = 0; q < NMF; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            read_frags(buf, 0, afA, bfA);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const bool last = s == NSUB - 1;
            if (ROT && last) break;
            auto& afc = (s & 1) ? afB : afA; auto& bfc = (s & 1) ? bfB : bfA;
            auto& afn = (s & 1) ? afA : afB; auto& bfn = (s & 1) ? bfA : bfB;
            if (!last) read_frags(buf, s + 1, afn, bfn);
            // stores of tile t+1 spread over the first two sub-steps; loads of tile t+1+DEPTH right after them
            if (s == 0) stash(sset, buf ^ 1, 0, (NA + 1) / 2, 0, (NB + 1) / 2);
            if (s == 1) stash(sset, buf ^ 1, (NA + 1) / 2, NA, (NB + 1) / 2, NB);
            if (s == 2 || (NSUB <= 2 && s == 1)) issue(sset, knext);
            mfma(afc, bfc);
#pragma unroll
            for (int q = 0; q < NMF; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    typedef std::integral_constant<int, 0> P0; typedef std::integral_constant<int, 1> P1;
    step(std::true_type{}, P0{}, 0);
    int t = 1;
    for (; t + 1 < nsteps; t += 2) { step(std::false_type{}, P1{}, t); step(std::false_type{}, P0{}, t + 1); }
    if (t < nsteps) step(std::false_type{}, P1{}, t);
    if (ROT) mfma((NSUB & 1) ? afA : afB, (NSUB & 1) ? bfA : bfB);

#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                C[(long long)(m0 + wm0 + 16 * i + 4 * lk + q) * N + n0 + wn0 + 16 * j + li] = acc[i][j][q];
}

template <int BM, int BN, int WGM, int WGN, int BK, int OCC, int DEPTH, bool ROT>
static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
    constexpr int T = 64 * WGM * WGN;
    const int lds = 2 * (BM + BN) * (BK + 4) * 4;
    auto kern = mb_gemm<BM, BN, WGM, WGN, BK, OCC, DEPTH, ROT>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, T, lds));
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    CHECK(hipMemset(C, 0, (size_t)M * N * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, A, B, C, M, N, K);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    float best = 1e9f, tot = 0.f;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, A, B, C, M, N, K);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; tot += ms;
    }
    CHECK(hipGetLastError());
    // spot check
    std::vector<float> hC((size_t)M * N);
    CHECK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int s = 0; s < 64; ++s) {
        const int r = (int)((1103515245u * (unsigned)(s + 1) + 12345u) % (unsigned)M), c = (int)((22695477u * (unsigned)(s + 7) + 1u) % (unsigned)N);
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)r * N + c]));
    }
    const double gf = 2.0 * M * N * (double)K * 1e-9;
    printf("%-34s occ %d grid %4d lds %6d  avg %7.1f us  best %7.1f us  %6.1f TF (best %6.1f)  %.3f of 157.3  err %.2e\n", name, occ, grid, lds,
           tot / reps * 1e3, best * 1e3, gf / (tot / reps), gf / best, gf / (tot / reps) / 157.3, maxerr);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12288, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 6528;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f)) * 2.f - 1.f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd() * 0.05f;
    float *A, *B, *C;
    CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&B, hB.size() * 4)); CHECK(hipMalloc(&C, (size_t)M * N * 4));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    printf("M %d N %d K %d  (%.2f GF)\n", M, N, K, 2.0 * M * N * (double)K * 1e-9);
#define RUN(...) run<__VA_ARGS__>(#__VA_ARGS__, A, B, C, M, N, K, hA, hB)
    RUN(96, 128, 2, 2, 32, 1, 2, true);
    RUN(96, 128, 2, 2, 32, 1, 2, false);
    RUN(96, 128, 2, 2, 32, 1, 1, true);
    RUN(96, 128, 2, 4, 32, 1, 2, true);      // 8 waves (2 per SIMD), one workgroup
    RUN(96, 128, 2, 4, 32, 1, 1, true);
    RUN(96, 64, 2, 2, 32, 2, 2, true);       // two independent workgroups per CU
    RUN(96, 64, 2, 2, 32, 2, 1, true);
    RUN(96, 64, 2, 2, 32, 2, 2, false);
    RUN(48, 128, 1, 4, 32, 2, 2, true);
    RUN(48, 128, 1, 4, 32, 2, 1, true);
    RUN(96, 128, 2, 2, 64, 1, 1, true);      // 64-deep k-steps, one workgroup per CU
    RUN(96, 64, 2, 2, 64, 2, 1, true);
    RUN(64, 64, 2, 2, 32, 3, 2, true);
    RUN(64, 64, 2, 2, 32, 4, 1, true);
    return 0;
}
