import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
dev = "cuda:0"
eng = NeuralCXEngine(device=dev); eng.init_parameters(seed=42)
data = SyntheticCX(n_triplets=16384, n_img=8192, device=dev)
def timed(label, fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); print("%-50s %.3f ms" % (label, (time.perf_counter() - t0) / n * 1e3))
b0, g0 = data.batch(torch.arange(0, 512))
timed("step, same resident batch", lambda i: eng.train_step(b0, g0))
bs = [data.batch(torch.arange(i * 512, (i + 1) * 512)) for i in range(10)]
timed("step, 10 resident batches", lambda i: eng.train_step(*bs[i]))
timed("batch() only", lambda i: data.batch(torch.arange(i * 512, (i + 1) * 512)))
def fresh(i):
    b, g = data.batch(torch.arange(i * 512, (i + 1) * 512)); eng.train_step(b, g)
timed("batch() + step", fresh)
timed("batch() + step (again)", fresh)
