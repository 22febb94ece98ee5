"""Debug aid (GPU box): per-parameter / per-segment gradient error of the HIP path vs the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import ncx_oracle as orc
from helpers import load_golden
import test_hip_parity as T

name = sys.argv[1] if len(sys.argv) > 1 else "g1_small_H20_L2"
g, d, spec, params, batch = load_golden(name)
scores, lr, grads = T.run_hip(d, spec, params, batch)
s_ref, l_ref, g_ref = orc.loss_and_grads(params, d, batch, spec=spec)
print("scores err", np.abs(scores.numpy() - s_ref.numpy()).max(), "loss", float(lr["loss"]), float(l_ref))
for k, ref in g_ref.items():
    ref = ref.numpy(); got = grads[k].reshape(ref.shape)
    print("%-26s max|ref| %.3e  err %.3e" % (k, np.abs(ref).max(), np.abs(got - ref).max()))
off = d.offsets(); names = list(off.keys()); ref = g_ref["linear_1.weight"].numpy(); got = grads["linear_1.weight"]
for i, n in enumerate(names):
    lo = off[n]; hi = off[names[i + 1]] if i + 1 < len(names) else d.din
    e = np.abs(got[:, lo:hi] - ref[:, lo:hi])
    print("  seg %-12s [%5d,%5d) max|ref| %.3e err %.3e  (worst row %d col %d)" % (n, lo, hi, np.abs(ref[:, lo:hi]).max(), e.max(), *np.unravel_index(e.argmax(), e.shape)))
