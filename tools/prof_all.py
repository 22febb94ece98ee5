"""Per-GEMM HIP-event timing of one training step (all gemm ids), honouring the NCX_CFG_<id>/NCX_SPLIT_<id> hooks."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx import _lib
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--H", type=int, default=256); ap.add_argument("--L", type=int, default=1)
ap.add_argument("--K", type=int, default=24); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
eng = NeuralCXEngine(K=a.K, H=a.H, L=a.L, device="cuda:0"); eng.init_parameters(seed=42)
data = SyntheticCX(n_triplets=4 * a.batch, K=a.K, n_img=20000, device="cuda:0")
pool = [data.batch(torch.arange(i * a.batch, (i + 1) * a.batch)) for i in range(4)]
for i in range(4): eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
names = [n for n in _lib.GEMM_IDS if n not in ("DW1S", "DAGT")]
_lib.profile_begin(names, max_launches=16 * a.steps + 64)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for i in range(a.steps): eng.train_step(*pool[i % 4])
ev1.record(); torch.cuda.synchronize()
prof = _lib.profile_end(cap=16 * a.steps + 64)
d0 = eng._dims(pool[0][0], True, 1.0 / a.batch)
tot = 0.0
for n in names:
    if n in prof:
        per_step = sum(prof[n]) / a.steps
        tot += per_step
        pl = _lib.plan_query(d0, n)
        tf = 2.0 * pl["M"] * pl["N"] * pl["ksteps"] * 32 / (per_step * 1e-3) / 1e12      # (padded k: upper bound of the useful rate)
        print("%-6s %7.1f us/step (%d launches/step)  %s %dx%dx%d ksteps  ~%.0fTF  tile %s split %d" % (
            n, per_step * 1e3, len(prof[n]) // a.steps, pl["form"], pl["M"], pl["N"], pl["ksteps"], tf, pl["tile"], pl["ksplit"]))
print("gemms %.1f us  step %.1f us" % (tot * 1e3, ev0.elapsed_time(ev1) / a.steps * 1e3))
