"""Where the grouped weight-gradient launch spends its time (round 4): at configs[1] the phased backward launches the dGt problem
ALONE (phase 5: 512 workgroups = exactly one round of the 2-per-CU slots) and the rest of the group (phase 2: z_other, dist | rank,
the shared segments) in a launch of their own, so a kernel trace separates the long problem's in-loop efficiency from the tail of
short ones.  Run under rocprofv3 --kernel-trace; tools/exp_dw1c_split.sh parses the per-dispatch durations.
usage: python tools/exp_dw1c_split.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx import ops
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = 512
eng = NeuralCXEngine(device="cuda:0"); eng.init_parameters(seed=42)
data = SyntheticCX(n_triplets=2 * B, n_img=20000, device="cuda:0")
b, gt = data.batch(torch.arange(0, B))
for _ in range(3):
    eng.train_step(b, gt)
d = eng._dims(b, True, 1.0 / B)
f = eng.params.fields()
scores = ops.forward(d, b, f, eng._ws)
r = ops.ranking_loss(scores, gt, scale=1.0 / B)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tot = [0.0, 0.0, 0.0]
for i in range(reps + 3):
    ev[0].record()
    ops.backward(d, b, f, eng._ws, r["dscores"], eng.grads.fields(), phase=5)
    ev[1].record()
    ops.backward(d, b, f, eng._ws, r["dscores"], eng.grads.fields(), phase=2)
    ev[2].record()
    ops.backward(d, b, f, eng._ws, r["dscores"], eng.grads.fields(), phase=4)
    ev[3].record()
    torch.cuda.synchronize()
    if i >= 3:
        for j in range(3):
            tot[j] += ev[j].elapsed_time(ev[j + 1])
print("phase 5 (prelude + dGt alone + emb prep) %.1f us | phase 2 (k_dw_km + rest of the group + reductions + dW1ak) %.1f us | phase 4 (dE) %.1f us"
      % tuple(1e3 * t / reps for t in tot))
