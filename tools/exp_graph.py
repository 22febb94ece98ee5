"""Timing experiment: eager train_step vs replay of a captured hipGraph of the same step (seed / Adam step baked in:
timing only) at small batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
dev = "cuda:0"
for B in (32, 64, 128, 512):
    eng = NeuralCXEngine(device=dev); eng.init_parameters(seed=42)
    data = SyntheticCX(n_triplets=4 * B, n_img=8192, device=dev)
    b, gt = data.batch(torch.arange(0, B, device=dev), first_id=0)
    for _ in range(3): eng.train_step(b, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eng.train_step(b, gt)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.train_step(b, gt)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        r = eng.train_step(b, gt)
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 50
    print("B=%4d  eager %.3f ms  graph %.3f ms  (%.0f -> %.0f triplets/s)" % (B, eager * 1e3, graph * 1e3, B / eager, B / graph))
