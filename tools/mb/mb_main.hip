// mb_main.hip -- times variants of the fused forward kernel (csrc/ncx_main.h) on synthetic operands of the configs[1] shape,
// whole and per segment kind.  Development tool, not part of the library.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ivqa-counterexamples_amd/csrc tools/mb/mb_main.hip -o tools/mb/mb_main
#include "../../vqa-counterexamples_amd/csrc/ncx_main.h"
#include <stdio.h>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>

using namespace ncx;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed, bool positive = false) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; float u = ((s >> 8) * (1.0f / 16777216.0f)); v = (positive ? u : 2.f * u - 1.f) * scale; }
    float* d; CHECK(hipMalloc(&d, n * 4)); CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}

struct Problem { MainArgs full; int B, K, H; float* out; };

template <class CFG>
static float time_cfg(const char* name, const Problem& p, unsigned segmask, const float* ref_out, bool print = true) {
    MainArgs a = p.full;
    int n = 0; double ksteps = 0, cols = 0;
    for (int i = 0; i < p.full.nseg; ++i) if ((segmask >> i) & 1u) { a.seg[n++] = p.full.seg[i]; ksteps += (p.full.seg[i].klen + 31) / 32; cols += p.full.seg[i].klen; }
    a.nseg = n;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_main_fwd<CFG>(a, 0) != 0) { printf("launch failed\n"); exit(1); }
    CHECK(hipDeviceSynchronize());
    const int reps = 20; float tot = 0.f, best = 1e9f;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0, 0)); launch_main_fwd<CFG>(a, 0); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = ms < best ? ms : best;
    }
    double maxdiff = -1;
    if (ref_out) {
        const size_t nn = (size_t)a.M * a.N;
        std::vector<float> h(nn), r(nn);
        CHECK(hipMemcpy(h.data(), a.out, nn * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r.data(), ref_out, nn * 4, hipMemcpyDeviceToHost));
        maxdiff = 0; for (size_t i = 0; i < nn; ++i) { double d = fabs((double)h[i] - r[i]); if (!(d <= maxdiff)) maxdiff = d; }
    }
    const double gf = 2.0 * a.M * a.N * cols * 1e-9;
    int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_main_fwd<CFG, false, MK_GATHER, MK_GATHER_MUL, MK_PLAIN, MK_PLAIN, MK_SOFTMAX>, MF_T, CFG::LDS));
    if (print) printf("%-22s segs %02x  ksteps %4.0f  avg %7.1f us  best %7.1f  %6.1f TF  (%.3f us/kstep)  occ %d  maxdiff %.2e\n", name, segmask, ksteps,
                      tot / reps * 1e3, best * 1e3, gf / (tot / reps), tot / reps * 1e3 / ksteps, occ, maxdiff);
    fflush(stdout);
    return tot / reps;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, K = 24, H = argc > 2 ? atoi(argv[2]) : 256, dv = 2048, dz = 360, A = 2000, n_img = 20000;
    const int M = B * K; const long long din = 3LL * dv + 2 * 2400 + 2 * dz + 2400 + K + 1;
    float* feats = dev_rand((size_t)n_img * dv, 0.45f, 1, true);
    float* misc = dev_rand((size_t)M * 28, 1.f, 2);
    float* z = dev_rand((size_t)M * dz, 1.f, 3);
    float* logits = dev_rand((size_t)M * A, 4.f, 4);
    float* w1 = dev_rand((size_t)H * din, 0.0084f, 5);
    float* gt = dev_rand((size_t)H * 2016, 0.4f, 6);
    float* sh = dev_rand((size_t)B * H, 0.3f, 7);
    std::vector<int> hk(M), ho(M); std::vector<float> hl(M);
    unsigned s = 99;
    for (int r = 0; r < M; ++r) { s = s * 1664525u + 1013904223u; hk[r] = (s >> 8) % n_img; }
    for (int b = 0; b < B; ++b) { s = s * 1664525u + 1013904223u; for (int k = 0; k < K; ++k) ho[b * K + k] = (s >> 8) % n_img; }
    for (int r = 0; r < M; ++r) hl[r] = 4.f * 1.4427f + log2f(2000.f * 0.35f);       // ~ lse2 of uniform(-4, 4) logits (keeps exp2 in range)
    int *idx_k, *idx_o; float* lse;
    CHECK(hipMalloc(&idx_k, M * 4)); CHECK(hipMalloc(&idx_o, M * 4)); CHECK(hipMalloc(&lse, M * 4));
    CHECK(hipMemcpy(idx_k, hk.data(), M * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx_o, ho.data(), M * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(lse, hl.data(), M * 4, hipMemcpyHostToDevice));
    float *out, *ref; CHECK(hipMalloc(&out, (size_t)M * H * 4)); CHECK(hipMalloc(&ref, (size_t)M * H * 4));

    Problem p{}; p.B = B; p.K = K; p.H = H;
    MainArgs& a = p.full; a.M = M; a.N = H; a.out = out; a.ldo = H;
    int n = 0;
    auto seg = [&](int kind, const float* x, long long lda, const int* i1, const int* i2, const float* l, const float* wgt, long long ldb, int klen) {
        MainSeg& g = a.seg[n++]; g.kind = kind; g.a = x; g.lda = lda; g.idx = i1; g.idx2 = i2; g.lse = l; g.b = wgt; g.ldb = ldb; g.klen = klen; };
    seg(MK_GATHER, feats, dv, idx_k, nullptr, nullptr, w1 + dv, din, dv);
    seg(MK_GATHER_MUL, feats, dv, idx_k, idx_o, nullptr, w1 + 2 * dv, din, dv);
    seg(MK_PLAIN, misc, 28, nullptr, nullptr, nullptr, w1 + 3 * dv, din, 28);
    seg(MK_PLAIN, z, dz, nullptr, nullptr, nullptr, w1 + 3 * dv + K + 1 + 2400 + dz, din, dz);
    seg(MK_SOFTMAX, logits, A, nullptr, nullptr, lse, gt, 2016, A);
    a.nseg = n;
    a.epi.rowadd = sh; a.epi.ld_rowadd = H; a.epi.rowdiv = K; a.epi.relu = 1;
    a.epi.dropout = 1; a.epi.drop_p = 0.25f; a.epi.drop_scale = 1.f / 0.75f; a.epi.seed_lo = 123; a.epi.seed_hi = 456; a.epi.layer = 1;
    printf("B %d K %d H %d  M %d\n", B, K, H, M);

    typedef MainCfg<48, 128, 1, 4, 2, 2> C_48_128_d2_o2;
    typedef MainCfg<48, 128, 1, 4, 1, 2> C_48_128_d1_o2;
    typedef MainCfg<48, 128, 1, 4, 2, 3> C_48_128_d2_o3;
    typedef MainCfg<48, 128, 1, 4, 1, 3> C_48_128_d1_o3;
    typedef MainCfg<96, 64, 2, 2, 2, 2> C_96_64_d2_o2;
    typedef MainCfg<96, 64, 2, 2, 1, 2> C_96_64_d1_o2;
    typedef MainCfg<96, 64, 2, 2, 1, 3> C_96_64_d1_o3;
    typedef MainCfg<64, 64, 2, 2, 2, 3> C_64_64_d2_o3;
    typedef MainCfg<64, 64, 2, 2, 1, 4> C_64_64_d1_o4;
    typedef MainCfg<96, 128, 2, 2, 2, 1> C_96_128_d2_o1;

    // reference output from the first config (all variants use the same k order per element -> bit-identical expected)
    time_cfg<C_48_128_d2_o2>("48x128 d2 o2", p, 0x1f, nullptr, false);
    CHECK(hipMemcpy(ref, out, (size_t)M * H * 4, hipMemcpyDeviceToDevice));
#define ALLSEG(C) time_cfg<C>(#C, p, 0x1f, ref)
    ALLSEG(C_48_128_d2_o2); ALLSEG(C_48_128_d1_o2); ALLSEG(C_48_128_d2_o3); ALLSEG(C_48_128_d1_o3);
    ALLSEG(C_96_64_d2_o2); ALLSEG(C_96_64_d1_o2); ALLSEG(C_96_64_d1_o3); ALLSEG(C_64_64_d2_o3); ALLSEG(C_64_64_d1_o4); ALLSEG(C_96_128_d2_o1);
    {   // which operand kind costs what: variations of the real problem, INTERLEAVED rounds (DVFS drifts run to run)
        typedef C_96_64_d1_o2 CA; typedef C_48_128_d2_o2 CB;
        std::vector<int> seq(M); for (int r = 0; r < M; ++r) seq[r] = r % n_img;
        int* idx_seq; CHECK(hipMalloc(&idx_seq, M * 4)); CHECK(hipMemcpy(idx_seq, seq.data(), M * 4, hipMemcpyHostToDevice));
        Problem ps = p; ps.full.seg[0].idx = idx_seq; ps.full.seg[1].idx = idx_seq;
        Problem q = p; q.full.seg[4].kind = MK_PLAIN;                  // softmax -> plain (no exp2)
        Problem r4 = p; r4.full.seg[1] = p.full.seg[2]; r4.full.seg[2] = p.full.seg[3]; r4.full.seg[3] = p.full.seg[4]; r4.full.nseg = 4;   // no v_mult
        const int KD = 6528;
        float* dA = dev_rand((size_t)M * KD, 1.f, 21); float* dB = dev_rand((size_t)H * KD, 0.05f, 22);
        Problem pd = p; MainArgs& d = pd.full; d.nseg = 1;
        d.seg[0] = MainSeg{}; d.seg[0].kind = MK_PLAIN; d.seg[0].a = dA; d.seg[0].lda = KD; d.seg[0].b = dB; d.seg[0].ldb = KD; d.seg[0].klen = KD;
        struct V { const char* name; std::function<void()> run; std::vector<float> ms; };
        std::vector<V> vs;
        auto add = [&](const char* name, std::function<void()> f) { vs.push_back(V{name, f, {}}); };
        add("real            96x64 d1", [&] { MainArgs a = p.full; launch_main_fwd<CA>(a, 0); });
        add("real            48x128 d2", [&] { MainArgs a = p.full; launch_main_fwd<CB>(a, 0); });
        add("sequential rows 96x64 d1", [&] { MainArgs a = ps.full; launch_main_fwd<CA>(a, 0); });
        add("sequential rows 48x128 d2", [&] { MainArgs a = ps.full; launch_main_fwd<CB>(a, 0); });
        add("softmax->plain  96x64 d1", [&] { MainArgs a = q.full; launch_main_fwd<CA>(a, 0); });
        add("softmax->plain  48x128 d2", [&] { MainArgs a = q.full; launch_main_fwd<CB>(a, 0); });
        add("no v_mult (140) 96x64 d1", [&] { MainArgs a = r4.full; launch_main_fwd<CA>(a, 0); });
        add("no v_mult (140) 48x128 d2", [&] { MainArgs a = r4.full; launch_main_fwd<CB>(a, 0); });
        add("dense 1 segment 96x64 d1", [&] { MainArgs a = pd.full; launch_main_fwd<CA>(a, 0); });
        add("dense 1 segment 96x64 d2", [&] { MainArgs a = pd.full; launch_main_fwd<C_96_64_d2_o2>(a, 0); });
        add("dense 1 segment 48x128 d2", [&] { MainArgs a = pd.full; launch_main_fwd<CB>(a, 0); });
        add("dense 1 segment 64x64 d2 o3", [&] { MainArgs a = pd.full; launch_main_fwd<C_64_64_d2_o3>(a, 0); });
        add("real            64x64 d2 o3", [&] { MainArgs a = p.full; launch_main_fwd<C_64_64_d2_o3>(a, 0); });
        add("real            96x128 d2 o1", [&] { MainArgs a = p.full; launch_main_fwd<C_96_128_d2_o1>(a, 0); });
        float* dist_buf; CHECK(hipMalloc(&dist_buf, (size_t)M * 28 * 4));
        add("real + in-kernel dist 96x128 d2 o1", [&] { MainArgs a = p.full; a.dist_out = dist_buf; a.ld_dist = 28; launch_main_fwd<C_96_128_d2_o1>(a, 0); });
        add("real + in-kernel dist 48x128 d2", [&] { MainArgs a = p.full; a.dist_out = dist_buf; a.ld_dist = 28; launch_main_fwd<CB>(a, 0); });
        typedef MainCfg<48, 64, 1, 4, 2, 2> CF;
        Problem pf = p; pf.full.seg[0].kind = MK_VFOLD; pf.full.seg[0].idx2 = idx_o; pf.full.seg[0].b2 = p.full.seg[1].b;
        pf.full.seg[1] = p.full.seg[2]; pf.full.seg[2] = p.full.seg[3]; pf.full.seg[3] = p.full.seg[4]; pf.full.nseg = 4;
        add("fold 48x64", [&] { MainArgs a = pf.full; launch_main_fwd<CF>(a, 0); });
        add("fold 48x64 + dist", [&] { MainArgs a = pf.full; a.dist_out = dist_buf; a.ld_dist = 28; launch_main_fwd<CF>(a, 0); });
        add("real 48x64 (no fold)", [&] { MainArgs a = p.full; launch_main_fwd<CF>(a, 0); });
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int round = 0; round < 7; ++round)
            for (auto& v : vs) {
                v.run();
                CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 4; ++i) v.run(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 4);
            }
        printf("-- variations, 7 interleaved rounds x 4 launches: median / min us\n");
        for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end()); printf("   %-30s %7.1f  %7.1f\n", v.name, v.ms[v.ms.size() / 2] * 1e3, v.ms[0] * 1e3); }
        CHECK(hipFree(dA)); CHECK(hipFree(dB));
    }
    {   // a SHORT chain (the dE shape: M 2000, N 2400, two plain segments of 256): where does a 16-step workgroup spend its time?
        const int Ms = 2000, Ns = 2400, Ks = 256;
        float* sA0 = dev_rand((size_t)Ms * Ks, 1.f, 31); float* sA1 = dev_rand((size_t)Ms * Ks, 1.f, 32);
        float* sB0 = dev_rand((size_t)Ns * Ks, 0.05f, 33); float* sB1 = dev_rand((size_t)Ns * Ks, 0.05f, 34);
        float* sout; CHECK(hipMalloc(&sout, (size_t)Ms * Ns * 4));
        MainArgs b{}; b.M = Ms; b.N = Ns; b.nseg = 2; b.out = sout; b.ldo = Ns;
        b.seg[0].kind = MK_PLAIN; b.seg[0].a = sA0; b.seg[0].lda = Ks; b.seg[0].klen = Ks; b.seg[0].b = sB0; b.seg[0].ldb = Ks;
        b.seg[1].kind = MK_PLAIN; b.seg[1].a = sA1; b.seg[1].lda = Ks; b.seg[1].klen = Ks; b.seg[1].b = sB1; b.seg[1].ldb = Ks;
        auto tm = [&](const char* nm, auto cfg_c) {
            typedef decltype(cfg_c) C;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) { MainArgs c = b; launch_main_fwd<C>(c, 0); }
            CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 10; ++i) { MainArgs c = b; launch_main_fwd<C>(c, 0); } CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const int tiles_m = (Ms + C::BM - 1) / C::BM, tiles_n = (Ns + C::BN - 1) / C::BN, grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
            unsigned long long* st; CHECK(hipMalloc(&st, (size_t)grid * 16 * 8)); CHECK(hipMemset(st, 0, (size_t)grid * 16 * 8));
            MainArgs c = b; c.stamps = st; launch_main_fwd<C>(c, 0); CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)grid * 16); CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            double s0 = 0, s1 = 0, s2 = 0, dur = 0; int cnt = 0;
            for (int g = 0; g < grid; ++g) { const unsigned long long* w = &h[g * 16]; if (!w[15]) continue; s0 += w[1] - w[0]; s1 += w[2] - w[1]; s2 += w[8] - w[2]; dur += (w[15] - w[14]) / 100.0; ++cnt; }
            printf("   short chain %-14s %6.1f us/launch  %d workgroups: seg0 %6.0f  seg1 %6.0f  epilogue %6.0f cycles, mean workgroup %.1f us\n", nm, ms * 100, cnt, s0 / cnt, s1 / cnt, s2 / cnt, dur / cnt);
            CHECK(hipFree(st));
        };
        printf("-- short chain (dE shape)\n");
        tm("48x128 d2 o2", C_48_128_d2_o2{}); tm("96x128 d2 o1", C_96_128_d2_o1{}); tm("96x64 d1 o2", C_96_64_d1_o2{}); tm("64x64 d2 o3", C_64_64_d2_o3{});
    }
    {   // the Gt shape (M 256, N 2000, K 2400, one plain segment) with a k-split: fused forward kernel + fix-up
        const int Mg = 256, Ng = 2000, Kg = 2400;
        float* gA = dev_rand((size_t)Mg * Kg, 1.f, 41); float* gB = dev_rand((size_t)Ng * Kg, 0.05f, 42);
        float* gout; CHECK(hipMalloc(&gout, (size_t)Mg * 2016 * 4));
        float* gslab; CHECK(hipMalloc(&gslab, (size_t)16 * Mg * Ng * 4));
        MainArgs b{}; b.M = Mg; b.N = Ng; b.nseg = 1; b.out = gout; b.ldo = 2016; b.slab = gslab;
        b.seg[0].kind = MK_PLAIN; b.seg[0].a = gA; b.seg[0].lda = Kg; b.seg[0].klen = Kg; b.seg[0].b = gB; b.seg[0].ldb = Kg;
        printf("-- Gt shape, fused forward kernel + fix-up (generic engine in the step: 41 us)\n");
        auto tm = [&](const char* nm, auto cfg_c, int split) {
            typedef decltype(cfg_c) C;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            MainArgs c = b; c.split = split;
            for (int i = 0; i < 3; ++i) launch_main_fwd<C>(c, 0);
            CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 10; ++i) launch_main_fwd<C>(c, 0); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("   %-14s split %2d  %6.1f us\n", nm, split, ms * 100);
        };
        for (int sp : {3, 4, 6}) { tm("64x64 d2 o3", C_64_64_d2_o3{}, sp); tm("48x128 d2 o2", C_48_128_d2_o2{}, sp); tm("96x64 d1 o2", C_96_64_d1_o2{}, sp); }
        // the Sh shape: M 512, N 256, K 7208 (one plain segment stands in for the four)
        const int Ms = 512, Ns = 256, Ksh = 7208;
        float* hA = dev_rand((size_t)Ms * Ksh, 1.f, 43); float* hB = dev_rand((size_t)Ns * 7232, 0.05f, 44);
        float* hslab; CHECK(hipMalloc(&hslab, (size_t)32 * Ms * Ns * 4));
        MainArgs h{}; h.M = Ms; h.N = Ns; h.nseg = 1; h.out = gout; h.ldo = Ns; h.slab = hslab;
        h.seg[0].kind = MK_PLAIN; h.seg[0].a = hA; h.seg[0].lda = Ksh; h.seg[0].klen = Ksh; h.seg[0].b = hB; h.seg[0].ldb = 7232;
        b = h;
        printf("-- Sh shape (generic engine in the step: 38 us)\n");
        for (int sp : {12, 16, 24}) { tm("64x64 d2 o3", C_64_64_d2_o3{}, sp); tm("48x128 d2 o2", C_48_128_d2_o2{}, sp); }
    }
    {   // in-kernel stamps of one launch of the library's configuration: where does a workgroup's time go?
        typedef C_48_128_d2_o2 C;
        const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (H + C::BN - 1) / C::BN, grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
        unsigned long long* st; CHECK(hipMalloc(&st, (size_t)grid * 16 * 8)); CHECK(hipMemset(st, 0, (size_t)grid * 16 * 8));
        MainArgs b = p.full; b.stamps = st;
        for (int i = 0; i < 3; ++i) launch_main_fwd<C>(b, 0);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)grid * 16);
        CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int g = 0; g < grid; ++g) { if (!h[g * 16 + 15]) continue; r0 = h[g * 16 + 14] < r0 ? h[g * 16 + 14] : r0; r1 = h[g * 16 + 15] > r1 ? h[g * 16 + 15] : r1; }
        printf("-- stamps (48x128 d2 o2): kernel span %.1f us (first start .. last end, 100 MHz counter)\n", (r1 - r0) / 100.0);
        double sum[10] = {0}, mx_start = 0, mn_dur = 1e30, mx_dur = 0, clk = 0; int cnt = 0; int per_xcd[8] = {0};
        for (int g = 0; g < grid; ++g) {
            const unsigned long long* w = &h[g * 16]; if (!w[15]) continue;
            const double start = (w[14] - r0) / 100.0, dur = (w[15] - w[14]) / 100.0;
            mx_start = start > mx_start ? start : mx_start; mn_dur = dur < mn_dur ? dur : mn_dur; mx_dur = dur > mx_dur ? dur : mx_dur;
            clk += (double)(w[8] - w[0]) / ((w[15] - w[14]) * 10.0);      // cycles per ns -> GHz
            for (int i = 0; i < 5; ++i) sum[i] += (double)(w[1 + i] - w[i]);
            sum[5] += (double)(w[8] - w[5]);
            per_xcd[w[13] & 7]++; ++cnt;
        }
        printf("   workgroups %d  latest start +%.1f us  duration min %.1f max %.1f us  mean clock %.2f GHz  per-XCD", cnt, mx_start, mn_dur, mx_dur, clk / cnt);
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcd[x]);
        printf("\n");
        const char* nm[6] = {"gather", "gather*mul", "misc", "z", "softmax", "epilogue"};
        const int ks[6] = {64, 64, 1, 12, 63, 1};
        for (int i = 0; i < 6; ++i) printf("   %-10s mean %9.0f cycles  (%7.0f per k-step; 48 MFMAs x 32 = 1536 issue cycles per wave, x2 workgroups per SIMD = 3072)\n", nm[i], sum[i] / cnt, sum[i] / cnt / ks[i]);
        // a few individual timelines
        for (int g : {0, 1, 8, 255, 256, 511}) if (g < grid && h[g * 16 + 15]) printf("   wg %3d xcd %llu start +%.1f us dur %.1f us\n", g, h[g * 16 + 13] & 7, (h[g * 16 + 14] - r0) / 100.0, (h[g * 16 + 15] - h[g * 16 + 14]) / 100.0);
    }
    {   // in-kernel stamps of one launch of the library's configuration: where does a workgroup's time go?
        typedef MainCfg<48, 64, 1, 4, 2, 2> C;
        const int tiles_m = (M + C::BM - 1) / C::BM, tiles_n = (H + C::BN - 1) / C::BN, grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
        unsigned long long* st; CHECK(hipMalloc(&st, (size_t)grid * 16 * 8)); CHECK(hipMemset(st, 0, (size_t)grid * 16 * 8));
        Problem pf = p; pf.full.seg[0].kind = MK_VFOLD; pf.full.seg[0].idx2 = idx_o; pf.full.seg[0].b2 = p.full.seg[1].b;
        pf.full.seg[1] = p.full.seg[2]; pf.full.seg[2] = p.full.seg[3]; pf.full.seg[3] = p.full.seg[4]; pf.full.nseg = 4;
        MainArgs b = pf.full; b.stamps = st;
        for (int i = 0; i < 3; ++i) launch_main_fwd<C>(b, 0);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)grid * 16);
        CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int g = 0; g < grid; ++g) { if (!h[g * 16 + 15]) continue; r0 = h[g * 16 + 14] < r0 ? h[g * 16 + 14] : r0; r1 = h[g * 16 + 15] > r1 ? h[g * 16 + 15] : r1; }
        printf("-- stamps (FOLD 48x64): kernel span %.1f us (first start .. last end, 100 MHz counter)\n", (r1 - r0) / 100.0);
        double sum[10] = {0}, mx_start = 0, mn_dur = 1e30, mx_dur = 0, clk = 0; int cnt = 0; int per_xcd[8] = {0};
        for (int g = 0; g < grid; ++g) {
            const unsigned long long* w = &h[g * 16]; if (!w[15]) continue;
            const double start = (w[14] - r0) / 100.0, dur = (w[15] - w[14]) / 100.0;
            mx_start = start > mx_start ? start : mx_start; mn_dur = dur < mn_dur ? dur : mn_dur; mx_dur = dur > mx_dur ? dur : mx_dur;
            clk += (double)(w[8] - w[0]) / ((w[15] - w[14]) * 10.0);      // cycles per ns -> GHz
            for (int i = 0; i < 4; ++i) sum[i] += (double)(w[1 + i] - w[i]);
            sum[5] += (double)(w[8] - w[4]);
            per_xcd[w[13] & 7]++; ++cnt;
        }
        printf("   workgroups %d  latest start +%.1f us  duration min %.1f max %.1f us  mean clock %.2f GHz  per-XCD", cnt, mx_start, mn_dur, mx_dur, clk / cnt);
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcd[x]);
        printf("\n");
        const char* nm[6] = {"fold(v)", "misc", "z", "softmax", "-", "epilogue"};
        const int ks[6] = {64, 1, 12, 63, 1, 1};
        for (int i = 0; i < 6; ++i) printf("   %-10s mean %9.0f cycles  (%7.0f per k-step; 48 MFMAs x 32 = 1536 issue cycles per wave, x2 workgroups per SIMD = 3072)\n", nm[i], sum[i] / cnt, sum[i] / cnt / ks[i]);
        // a few individual timelines
        for (int g : {0, 1, 8, 255, 256, 511}) if (g < grid && h[g * 16 + 15]) printf("   wg %3d xcd %llu start +%.1f us dur %.1f us\n", g, h[g * 16 + 13] & 7, (h[g * 16 + 14] - r0) / 100.0, (h[g * 16 + 15] - h[g * 16 + 14]) / 100.0);
    }
    return 0;
}
