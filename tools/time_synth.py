import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx.synth import SyntheticCX
d = SyntheticCX(n_triplets=8192, n_img=8192, device="cuda:0")
torch.cuda.synchronize()
for B in (512,):
    t0 = time.perf_counter()
    for i in range(10):
        b, gt = d.batch(torch.arange(i * B, (i + 1) * B))
    torch.cuda.synchronize()
    print("synthetic batch of %d: %.2f ms" % (B, (time.perf_counter() - t0) / 10 * 1e3))
g = torch.Generator(device="cuda:0")
t0 = time.perf_counter()
for i in range(10): g.manual_seed(i)
torch.cuda.synchronize(); print("manual_seed: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
t0 = time.perf_counter()
for i in range(10): a = torch.randn(512, 24, 2000, generator=g, device="cuda:0")
torch.cuda.synchronize(); print("randn 24.6M: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
