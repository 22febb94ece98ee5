"""In-kernel stamps of the fused forward kernel at configs[1]: cycles per phase (fold, dist|rank, z, softmax segment, epilogue), median over workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx import _lib
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
B = 512
eng = NeuralCXEngine(device="cuda:0"); eng.init_parameters(seed=42)
data = SyntheticCX(n_triplets=4 * B, n_img=82783, device="cuda:0")
pool = [data.batch(torch.arange(i * B, (i + 1) * B)) for i in range(4)]
for i in range(30):
    eng.train_step(*pool[i % 4])
st = torch.zeros(16 * 8192, dtype=torch.int64, device="cuda:0")
_lib.profile_stamps(st)
try:
    for i in range(12):
        eng.train_step(*pool[i % 4])
    torch.cuda.synchronize()
finally:
    _lib.profile_stamps(None)
w = st.view(-1, 16).cpu()
w = w[(w[:, 15] > w[:, 14]) & (w[:, 8] > w[:, 0])].double()
names = ["fold (v_k + v_o*v_k, 64 k-steps)", "dist | rank (1 k-step)", "z_other (12 k-steps)", "softmax(a) . Gt (63 k-steps)"]
prev = w[:, 0]
for i, n in enumerate(names):
    cur = w[:, 1 + i]
    print("%-36s %8.0f cycles (median)  %6.1f per k-step" % (n, float((cur - prev).median()), float((cur - prev).median()) / [64, 1, 12, 63][i]))
    prev = cur
print("%-36s %8.0f cycles" % ("epilogue", float((w[:, 8] - prev).median())))
mhz = ((w[:, 8] - w[:, 0]) / (w[:, 15] - w[:, 14]) * 100.0).median()
print("workgroup %.0f cycles = %.1f us at %.0f MHz; %d workgroups" % (float((w[:, 8] - w[:, 0]).median()), float(((w[:, 15] - w[:, 14]) / 100.0).median()), float(mhz), w.shape[0]))
