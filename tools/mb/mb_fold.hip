// mb_fold.hip -- the per-triplet fold phase of the fused forward kernel (csrc/ncx_main.h, MK_VFOLD) alone and with the segments that
// follow it, on synthetic operands of the configs[1] shape.  Built once per -DNCX_ABL_FOLD=n to see what the phase waits for
// (1 no W_m loads, 2 no v_o loads, 3 no weight loads, 4 no v_k loads: results wrong).  Development tool, not part of the library.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DNCX_ABL_FOLD=n] tools/mb/mb_fold.hip -o tools/mb/mb_fold[_n]
#include "../../vqa-counterexamples_amd/csrc/ncx_main.h"
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <math.h>
using namespace ncx;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float* dev_rand(size_t n, float scale, unsigned seed, bool positive = false) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; float u = ((s >> 8) * (1.0f / 16777216.0f)); v = (positive ? u : 2.f * u - 1.f) * scale; }
    float* d; CHECK(hipMalloc(&d, n * 4)); CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}
int main() {
    const int B = 512, K = 24, H = 256, dv = 2048, dz = 360, A = 2000, n_img = 20000, M = B * K;
    const long long din = 3LL * dv + 2 * 2400 + 2 * dz + 2400 + K + 1;
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
    for (int r = 0; r < M; ++r) hl[r] = 4.f * 1.4427f + log2f(2000.f * 0.35f);
    int *idx_k, *idx_o; float* lse;
    CHECK(hipMalloc(&idx_k, M * 4)); CHECK(hipMalloc(&idx_o, M * 4)); CHECK(hipMalloc(&lse, M * 4));
    CHECK(hipMemcpy(idx_k, hk.data(), M * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(idx_o, ho.data(), M * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(lse, hl.data(), M * 4, hipMemcpyHostToDevice));
    float* out; CHECK(hipMalloc(&out, (size_t)M * H * 4));
    MainArgs a{}; a.M = M; a.N = H; a.out = out; a.ldo = H;
    int n = 0;
    auto seg = [&](int kind, const float* x, long long lda, const int* i1, const int* i2, const float* l, const float* wgt, long long ldb, int klen) {
        MainSeg& g = a.seg[n++]; g.kind = kind; g.a = x; g.lda = lda; g.idx = i1; g.idx2 = i2; g.lse = l; g.b = wgt; g.ldb = ldb; g.klen = klen; };
    seg(MK_VFOLD, feats, dv, idx_k, idx_o, nullptr, w1 + dv, din, dv); a.seg[0].b2 = w1 + 2 * dv;
    seg(MK_PLAIN, misc, 28, nullptr, nullptr, nullptr, w1 + 3 * dv, din, 28);
    seg(MK_PLAIN, z, dz, nullptr, nullptr, nullptr, w1 + 3 * dv + K + 1 + 2400 + dz, din, dz);
    seg(MK_SOFTMAX, logits, A, nullptr, nullptr, lse, gt, 2016, A);
    a.epi.rowadd = sh; a.epi.ld_rowadd = H; a.epi.rowdiv = K; a.epi.relu = 1;
    a.epi.dropout = 1; a.epi.drop_p = 0.25f; a.epi.drop_scale = 1.f / 0.75f; a.epi.seed_lo = 123; a.epi.seed_hi = 456; a.epi.layer = 1;
    typedef MainCfg<48, 64, 1, 4, 2, 2> CF;
    typedef MainCfg<96, 64, 2, 2, 2, 2> CF4;
    {   // the 96-row fold must reproduce the 48-row one bit for bit (same effective-weight expression, same k order)
        float* out4; CHECK(hipMalloc(&out4, (size_t)M * H * 4));
        MainArgs b = a; b.nseg = 4;
        if (launch_main_fwd<CF>(b, 0) != 0) { printf("launch CF failed\n"); return 1; }
        MainArgs c = a; c.nseg = 4; c.out = out4;
        const int rc4 = launch_main_fwd<CF4>(c, 0);
        if (rc4 != 0) { printf("launch CF4 failed: %d\n", rc4); return 1; }
        CHECK(hipDeviceSynchronize());
        std::vector<float> h0((size_t)M * H), h4((size_t)M * H);
        CHECK(hipMemcpy(h0.data(), out, h0.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h4.data(), out4, h4.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; size_t nz = 0;
        for (size_t i = 0; i < h0.size(); ++i) { md = std::max(md, (double)fabsf(h0[i] - h4[i])); nz += h0[i] != 0.f; }
        printf("96-row fold vs 48-row fold: max |diff| %.3e over %zu outputs (%zu non-zero)\n", md, h0.size(), nz);
    }
    struct V { const char* name; int nseg; std::vector<float> ms; };
    std::vector<V> vs = {{"fold + misc + z + softmax (140 k-steps)", 4, {}}, {"fold + misc + z + misc (78 k-steps)", -4, {}},
                         {"96-row: fold + misc + z + softmax", 104, {}}, {"96-row: fold + misc + z + misc", -104, {}}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int round = 0; round < 9; ++round)
        for (auto& v : vs) {
            MainArgs b = a; b.nseg = 4;
            if (v.nseg < 0) b.seg[3] = a.seg[1];                     // (V, P, P, P): the softmax segment replaced by a one-step plain one
            const bool four = v.nseg > 100 || v.nseg < -100;
            { const int rc = four ? launch_main_fwd<CF4>(b, 0) : launch_main_fwd<CF>(b, 0); if (rc != 0) { printf("launch failed: %d (%s)\n", rc, v.name); return 1; } }
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < 4; ++i) { if (four) launch_main_fwd<CF4>(b, 0); else launch_main_fwd<CF>(b, 0); }
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 4);
        }
    {   // in-kernel stamps of one 96-row launch: where does a workgroup's time go?
        const int grid = ((M + 95) / 96 + 7) / 8 * 8 * (H / 64);
        unsigned long long* st; CHECK(hipMalloc(&st, (size_t)grid * 16 * 8)); CHECK(hipMemset(st, 0, (size_t)grid * 16 * 8));
        MainArgs b = a; b.nseg = 4; b.stamps = st;
        launch_main_fwd<CF4>(b, 0); CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)grid * 16);
        CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int g = 0; g < grid; ++g) if (h[g * 16 + 15]) { r0 = std::min(r0, h[g * 16 + 14]); r1 = std::max(r1, h[g * 16 + 15]); }
        double sum[6] = {0, 0, 0, 0, 0, 0}, clk = 0, mn = 1e9, mx = 0; int cnt = 0;
        for (int g = 0; g < grid; ++g) {
            const unsigned long long* w = &h[g * 16];
            if (!w[15]) continue;
            const double dur = (w[15] - w[14]) / 100.0;
            mn = std::min(mn, dur); mx = std::max(mx, dur);
            clk += (double)(w[8] - w[0]) / ((w[15] - w[14]) * 10.0);
            for (int i = 0; i < 4; ++i) sum[i] += (double)(w[1 + i] - w[i]);
            sum[5] += (double)(w[8] - w[4]);
            ++cnt;
        }
        printf("-- stamps (96-row fold): kernel span %.1f us, %d workgroups, duration %.1f .. %.1f us, mean clock %.2f GHz\n", (r1 - r0) / 100.0, cnt, mn, mx, clk / cnt);
        {   // who is slow?  per XCD: mean / max duration and mean fold-phase cycles; per column tile (id >> 3) % 4 likewise
            double dx[8] = {0}, mxx[8] = {0}, fx[8] = {0}; int nx[8] = {0};
            double dt[4] = {0}, ft[4] = {0}; int nt[4] = {0};
            for (int g = 0; g < grid; ++g) {
                const unsigned long long* w = &h[g * 16];
                if (!w[15]) continue;
                const int x = (int)(w[13] & 7), tnn = (g >> 3) % 4;
                const double dur = (w[15] - w[14]) / 100.0;
                dx[x] += dur; mxx[x] = std::max(mxx[x], dur); fx[x] += (double)(w[1] - w[0]); ++nx[x];
                dt[tnn] += dur; ft[tnn] += (double)(w[1] - w[0]); ++nt[tnn];
            }
            for (int x = 0; x < 8; ++x) if (nx[x]) printf("   xcd %d: %3d workgroups  mean %.1f  max %.1f us  fold phase %.0f cycles\n", x, nx[x], dx[x] / nx[x], mxx[x], fx[x] / nx[x]);
            for (int t = 0; t < 4; ++t) if (nt[t]) printf("   column tile %d: mean %.1f us  fold phase %.0f cycles\n", t, dt[t] / nt[t], ft[t] / nt[t]);
            int hist[10] = {0};
            for (int g = 0; g < grid; ++g) if (h[g * 16 + 15]) { const double dur = (h[g * 16 + 15] - h[g * 16 + 14]) / 100.0; int b = (int)((dur - mn) / (mx - mn + 1e-9) * 10); hist[b > 9 ? 9 : b]++; }
            printf("   duration histogram (%.0f .. %.0f us in 10 bins):", mn, mx); for (int b = 0; b < 10; ++b) printf(" %d", hist[b]); printf("\n");
        }
        const char* nm[6] = {"fold(v)", "misc", "z", "softmax", "-", "epilogue"};
        const int ks[6] = {64, 1, 12, 63, 1, 1}, mf[6] = {64, 48, 48, 48, 1, 1};
        for (int i = 0; i < 6; ++i) if (i != 4)
            printf("   %-10s mean %9.0f cycles  %7.0f per k-step  (%d MFMAs per wave and step x 32 cycles x 2 workgroups per SIMD = %d)\n",
                   nm[i], sum[i] / cnt, sum[i] / cnt / ks[i], mf[i], mf[i] * 64);
    }
#ifdef NCX_ABL_FOLD
    printf("ablation %d:", NCX_ABL_FOLD);
#else
    printf("full:      ");
#endif
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end()); printf("  %s  median %.1f  min %.1f us;", v.name, v.ms[v.ms.size() / 2] * 1e3, v.ms[0] * 1e3); }
    printf("\n");
    return 0;
}
