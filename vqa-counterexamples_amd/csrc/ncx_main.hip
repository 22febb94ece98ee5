// ncx_main.hip -- instantiation of the fused forward kernel of the Linear layers (ncx_main.h) for libneuralcx_hip.so.
#include "ncx_main.h"

namespace ncx {

typedef MainCfg<48, 128, 1, 4, 2, 2> MainCfg0;      // two workgroups per CU, 48 x 128 tiles (2 triplets at K = 24), loads two k-steps ahead
typedef MainCfg<96, 64, 2, 2, 1, 2> MainCfg1;
typedef MainCfg<96, 128, 2, 2, 2, 1> MainCfg2;      // one workgroup per CU with the whole register file
typedef MainCfg<160, 128, 2, 2, 2, 1> MainCfgW;     // one workgroup per CU, 160 x 128: for single-segment products whose 96-row tiles fill the chip badly (x_v of the MUTAN producer)
typedef MainCfg<64, 64, 2, 2, 2, 3> MainCfg3;       // short chains (the answer-embedding gradient: 16 k-steps): three small workgroups per CU
typedef MainCfg<48, 64, 1, 4, 2, 2> MainCfgFold;    // MK_VFOLD sequences: 48 x 64 tiles (two triplets), two workgroups per CU
typedef MainCfg<96, 64, 2, 2, 2, 2> MainCfgFold4;   // MK_VFOLD sequences: 96 x 64 tiles (four triplets share the W_k | W_m tiles), two workgroups per CU
typedef MainCfg<192, 64, 4, 2, 2, 2, 512, true> MainCfgFold8X6;   // ... with the segments after the fold on the bf16 matrix path, three-plane operands (NCX_F_X6)
typedef MainCfg<192, 64, 4, 2, 2, 2, 512> MainCfgFold8;   // MK_VFOLD on 192 x 64 tiles: ONE 8-wave workgroup per CU, eight triplets share the W_k | W_m tiles (experiment: NCX_FOLD8)

// Measured inside the training step at configs[1] (B = 512: 256 tiles of 96 x 128): 338 us with one 96 x 128 workgroup per CU,
// 346 / 348 us with two 48 x 128 / 96 x 64 workgroups per CU (on back-to-back launches of the kernel alone the order is the
// other way round: the step leaves the caches in another state).  The big tile needs enough tiles to fill the chip; smaller
// batches (data-parallel shards: 64 triplets per GPU) take the 48-row tile: twice the workgroups.
int main_forward(MainArgs& a, hipStream_t s) {
    if (a.nseg > 0 && a.seg[0].kind == MK_VFOLD) {
        // One triplet per wave.  192 x 64 tiles = ONE 8-wave workgroup per CU (round 3: eight triplets share the W_k | W_m tiles, no
        // old / young workgroup pair: 0.279-0.282 ms against 0.286-0.293 for two 96 x 64 workgroups per CU at configs[1], bit-identical);
        // 96 x 64 = four triplets per workgroup, two workgroups per CU; 48 x 64 (two triplets) for small batches: twice the workgroups
        // (tools/mb/mb_fold.hip, bit-identical outputs: 293 us against 318).
        const int rows = main_fold_rows_eff(a.M, a.N, a.epi.rowdiv);
        if (rows == 192 && a.x6 && a.epi.rowdiv == 24) return launch_main_fwd<MainCfgFold8X6>(a, s);      // (K = 24: a triplet per wave)
        return rows == 192 ? launch_main_fwd<MainCfgFold8>(a, s) : rows == 96 ? launch_main_fwd<MainCfgFold4>(a, s) : launch_main_fwd<MainCfgFold>(a, s);
    }
    long long T = 0;
    for (int i = 0; i < a.nseg; ++i) T += (a.seg[i].klen + MF_BK - 1) / MF_BK;
    if (a.split <= 1 && T <= 32 && (long long)((a.M + 63) / 64) * ((a.N + 63) / 64) >= 2 * num_cus() && !hook_env("NCX_MAIN_CFG"))
        return launch_main_fwd<MainCfg3>(a, s);          // a workgroup that short spends as long starting and storing as multiplying: more of them per CU
    const long long tiles96 = (long long)((a.M + 95) / 96) * ((a.N + 127) / 128);
    if (a.split <= 1 && a.nseg == 1 && a.seg[0].kind == MK_GATHER && !hook_env("NCX_MAIN_CFG")) {
        // x_v of the MUTAN producer at configs[2]: 12 800 x 360 outputs = 402 tiles of 96 x 128 = 1.57 rounds of the one-per-CU slots (the second
        // round 57 % full); 160-row tiles: 240 workgroups, ONE round
        const long long cus = num_cus(), tiles160 = (long long)((a.M + 159) / 160) * ((a.N + 127) / 128);
        const double e96 = (double)tiles96 / (double)(((tiles96 + cus - 1) / cus) * cus), e160 = (double)tiles160 / (double)(((tiles160 + cus - 1) / cus) * cus);
        if (tiles160 * 10 >= cus * 7 && e160 > e96 + 0.1) return launch_main_fwd<MainCfgW>(a, s);
    }
    int cfg = (a.split <= 1 && tiles96 * 10 >= (long long)num_cus() * 9) ? 2 : 0;
    if (const char* e = hook_env("NCX_MAIN_CFG")) cfg = atoi(e);       // experiment hook (NCX_EXPERIMENT=1)
    if (cfg == 3) return launch_main_fwd<MainCfgW>(a, s);
    if (cfg == 1) return launch_main_fwd<MainCfg1>(a, s);
    if (cfg == 2) return launch_main_fwd<MainCfg2>(a, s);
    return launch_main_fwd<MainCfg0>(a, s);
}

// ... with the experiment hooks and the K = 48 rule applied: the form main_forward takes
int main_fold_rows_eff(long long M, long long N, int rowdiv) {
    int rows = main_fold_rows(M, N);
    if (rowdiv == 48 && rows == 48) rows = 96;                               // (K = 48 exists on the one-triplet-per-wave forms only)
    if (const char* f4 = hook_env("NCX_FOLD4")) rows = atoi(f4) != 0 ? 96 : (rowdiv == 48 ? 96 : 48);
    if (const char* f8 = hook_env("NCX_FOLD8")) rows = atoi(f8) != 0 ? 192 : (rows == 192 ? 96 : rows);
    return rows;
}

int main_fold_rows(long long M, long long N) {
    const long long cus = num_cus();
    const long long wg96 = ((M + 95) / 96) * ((N + 63) / 64), wg192 = ((M + 191) / 192) * ((N + 63) / 64);
    if (wg96 * 4 < cus * 2 * 3) return 48;                                // < 3/4 of the two-per-CU slots: the 48-row form, twice the workgroups
    // rounds of workgroup slots each form fills (192-row tiles: one 8-wave workgroup per CU; 96-row tiles: two 4-wave ones)
    const double e8 = (double)wg192 / (double)(((wg192 + cus - 1) / cus) * cus), e4 = (double)wg96 / (double)(((wg96 + 2 * cus - 1) / (2 * cus)) * 2 * cus);
    return e8 + 0.03 >= e4 ? 192 : 96;                                   // (the 8-wave form is ~3 % faster at equal fill)
}

int main_split(long long M, long long N, long long T) {
    const long long tiles = ((M + 47) / 48) * ((N + 127) / 128), cus = num_cus();
    if (tiles >= cus) return 1;
    long long S = 2 * cus / tiles;                  // all workgroups resident at once (two per CU): one round, no tail
    if (S > T / 8) S = T / 8;
    if (const char* e = hook_env("NCX_MAIN_SPLIT")) S = atoll(e);      // experiment hook (NCX_EXPERIMENT=1)
    return (int)(S < 1 ? 1 : S);
}

}  // namespace ncx
