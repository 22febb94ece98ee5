"""Experiment (round 4): can the NEXT step's data-only forward prelude (k_prep, HBM-bound, 47 us) run on a side stream under the
current step's backward (MFMA-bound) on one GPU?  Two workspaces alternate; the side stream starts the prelude of batch i + 1 when the
forward of batch i is done.  Prints ms/step of the plain sequence and of the overlapped one (same kernels, same results).
usage: python tools/exp_prelude_overlap.py [steps=60]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx import ops
from neuralcx._lib import NCX_F_FUSED_TAIL
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = 512
dev = "cuda:0"
eng = NeuralCXEngine(device=dev); eng.init_parameters(seed=42)
data = SyntheticCX(n_triplets=4 * B, n_img=82783, device=dev)
pool = [data.batch(torch.arange(i * B, (i + 1) * B)) for i in range(4)]
for i in range(6):
    eng.train_step(*pool[i % 4])
torch.cuda.synchronize()
f = eng.params.fields(); gr = eng.grads.fields()
d0 = eng._dims(pool[0][0], True, 1.0 / B)
d0.flags |= NCX_F_FUSED_TAIL
ws = [ops.alloc_workspace(d0, dev), ops.alloc_workspace(d0, dev)]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def run(overlap, n):
    pre_done = [torch.cuda.Event(), torch.cuda.Event()]
    fwd_done = torch.cuda.Event()
    ops.forward(d0, pool[0][0], f, ws[0], phase=ops.FWD_PRELUDE); pre_done[0].record(main)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        cur, nxt = i & 1, (i + 1) & 1
        b, gt = pool[i % 4]
        main.wait_event(pre_done[cur])
        scores = ops.forward(d0, b, f, ws[cur], phase=ops.FWD_REST)
        r = ops.train_tail(d0, f, ws[cur], scores, gt, gr)
        nb = pool[(i + 1) % 4][0]
        if overlap:
            fwd_done.record(main)
            side.wait_event(fwd_done)
            with torch.cuda.stream(side):
                ops.forward(d0, nb, f, ws[nxt], phase=ops.FWD_PRELUDE)
                pre_done[nxt].record(side)
        ops.backward(d0, b, f, ws[cur], r["dscores"], gr)
        ops.adam_step(eng.params.flat, eng.grads.flat, eng.exp_avg, eng.exp_avg_sq, eng.step_count + i + 1, lr=1e-4)
        if not overlap:
            ops.forward(d0, nb, f, ws[nxt], phase=ops.FWD_PRELUDE); pre_done[nxt].record(main)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    print("sequential %.4f ms/step   prelude under the backward %.4f ms/step" % (run(False, steps), run(True, steps)), flush=True)
