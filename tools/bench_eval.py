"""Forward-only (eval_model, counterexamples.py:450-490) throughput at full validation scale: 118 499 synthetic triplets,
forward + listwise loss + Recall@1/@5 on device, metrics accumulated without host syncs.  Inputs resident in HBM."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=118499); ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--H", type=int, default=256); ap.add_argument("--L", type=int, default=1)
a = ap.parse_args()
dev = "cuda:0"
eng = NeuralCXEngine(H=a.H, L=a.L, device=dev); eng.init_parameters(seed=42)
data = SyntheticCX(n_triplets=a.n, device=dev)
pool = [data.batch(torch.arange(i * a.batch, (i + 1) * a.batch)) for i in range(4)]        # resident batches, cycled
last = a.n % a.batch
tail = data.batch(torch.arange(a.n - last, a.n)) if last else None
for b, gt in pool[:2]:
    eng.eval_step(b, gt)
torch.cuda.synchronize()
tot = torch.zeros(4, dtype=torch.float64, device=dev)
t0 = time.perf_counter()
for i in range(a.n // a.batch):
    b, gt = pool[i % len(pool)]
    r = eng.eval_step(b, gt)
    tot[0] += r["loss_rows"].double().sum() * a.batch; tot[1] += r["hits"][0]; tot[2] += r["hits"][1]; tot[3] += a.batch
if tail is not None:
    r = eng.eval_step(*tail)
    tot[0] += r["loss_rows"].double().sum() * last; tot[1] += r["hits"][0]; tot[2] += r["hits"][1]; tot[3] += last
torch.cuda.synchronize()
dt = time.perf_counter() - t0
l, h1, h5, n = tot.tolist()
print(json.dumps({"mode": "eval (fwd + loss + recall)", "triplets": int(n), "seconds": round(dt, 4), "triplets_per_s": round(n / dt, 1),
                  "ms_per_batch": round(dt / (a.n / a.batch) * 1e3, 4), "loss": l / n, "recall_1": h1 / n, "recall_5": h5 / n,
                  "batch": a.batch, "H": a.H, "L": a.L}))
