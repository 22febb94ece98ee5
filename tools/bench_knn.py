"""Full-scale timing of the kNN step: 82 783 x 2048 synthetic rows, k = 25; scikit-learn (the reference's
knn.py path) timed on a sample of query rows on the host for comparison."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import numpy as np, torch
from neuralcx.knn import knn
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=82783); ap.add_argument("--dv", type=int, default=2048)
ap.add_argument("--k", type=int, default=25); ap.add_argument("--block_rows", type=int, default=4096); ap.add_argument("--cpu_rows", type=int, default=200)
a = ap.parse_args()
g = torch.Generator(device="cuda:0").manual_seed(0)
t = torch.randn(a.n, a.dv, generator=g, device="cuda:0").abs_() * 0.45
knn(t[:8192].contiguous(), k=a.k)                      # warm-up (module load, allocator)
torch.cuda.synchronize(); t0 = time.perf_counter()
idx, dist = knn(t, k=a.k, block_rows=a.block_rows)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
flops = 2.0 * a.n * a.n * a.dv
out = {"rows": a.n, "dv": a.dv, "k": a.k, "seconds": round(dt, 4), "rows_per_s": round(a.n / dt, 1), "gemm_tflops_equiv": round(flops / dt / 1e12, 2)}
if a.cpu_rows:
    from sklearn.neighbors import NearestNeighbors
    x = t.cpu().numpy()
    nb = NearestNeighbors(n_neighbors=a.k).fit(x)
    c0 = time.perf_counter(); rd, ri = nb.kneighbors(x[:a.cpu_rows]); cdt = time.perf_counter() - c0
    out["sklearn_rows_per_s"] = round(a.cpu_rows / cdt, 1); out["sklearn_sample_rows"] = a.cpu_rows
    out["sklearn_cores"] = os.cpu_count()
    out["indices_equal_on_sample"] = bool((idx[:a.cpu_rows].cpu().numpy() == ri).all())
print(json.dumps(out))
