"""Summarise rocprofv3 --pmc passes (SQ counters) per kernel: MFMA-pipe utilisation, wait / active shares, LDS conflicts.
usage: python tools/pmc_mfma.py <dir_pass1> <dir_pass2> <out.txt>
pass 1: SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass 2: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (x4 = cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(MI355X_MICROARCH.md, cycle constants)."""
import csv, glob, sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


a, b = load(sys.argv[1]), load(sys.argv[2])
lines = []
for k in sorted(a, key=lambda k: -sum(a[k].get("SQ_WAVE_CYCLES", [0]))):
    if "gemm" not in k and "k_main_fwd" not in k and "k_dw_km" not in k and "k_dw_tn8" not in k and "k_dw_reduce" not in k and "k_mutan" not in k:
        continue
    m = lambda d, n: (sum(d[k][n]) / len(d[k][n])) if k in d and d[k].get(n) else float("nan")
    wc, busy, wi, ai = m(a, "SQ_WAVE_CYCLES") * 4, m(a, "SQ_VALU_MFMA_BUSY_CYCLES"), m(a, "SQ_WAIT_INST_ANY") * 4, m(a, "SQ_ACTIVE_INST_ANY") * 4
    conf, idx, wany, waves = m(b, "SQ_LDS_BANK_CONFLICT"), m(b, "SQ_LDS_IDX_ACTIVE"), m(b, "SQ_WAIT_ANY") * 4, m(b, "SQ_WAVES")
    lines.append("%s\n    launches %d  waves/launch %.0f  wave-cycles %.3e  MFMA-busy cycles %.3e  -> MFMA pipe busy %.1f %% of resident wave time"
                 % (k, len(a[k].get("SQ_WAVE_CYCLES", [])), waves, wc, busy, 100.0 * busy / wc if wc else float("nan")))
    lines.append("    issue-stall (WAIT_INST_ANY) %.1f %%  wave-parked (WAIT_ANY: waitcnt/barrier) %.1f %%  issuing (ACTIVE_INST_ANY) %.1f %%  LDS bank conflicts %.1f %% of LDS-active cycles"
                 % (100 * wi / wc, 100 * wany / wc, 100 * ai / wc, 100 * conf / idx if idx else float("nan")))
open(sys.argv[3], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
