"""Stage timing of the CLI training loop (counterexamples.py --synthetic): where does a step's wall time go?"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import torch
import counterexamples as cli
from neuralcx import dp

args = cli.build_parser().parse_args(["--synthetic", "--syn_train", "16384", "--syn_val", "1024", "--syn_images", "8192",
                                      "--epochs", "1", "--print_freq", "1000000"] + sys.argv[1:])
opt = cli.load_options(args)
r = cli.Runner(args, opt)
r.load_synthetic()
r.run_epoch(0)
torch.cuda.synchronize()
t0 = time.perf_counter(); tps = r.run_epoch(1); print("epoch: %.1f ms/step, %.0f triplets/s" % ((time.perf_counter() - t0) / (r.train.N // r.gb) * 1e3, tps))
stages = {"ids": 0.0, "batch": 0.0, "step": 0.0, "acc": 0.0}
acc = torch.zeros(3, dtype=torch.float64, device=r.dev)
n = 0
ids_dev, plan = dp.epoch_plan(r.train.N, r.gb, 2, 0, 1, r.dev, seed=42)
for lo, hi, ng, first, _active in plan:
    t = time.perf_counter(); it = ids_dev[lo:hi]; stages["ids"] += time.perf_counter() - t
    t = time.perf_counter(); b, gt = r.get_batch(r.train, it, first); torch.cuda.synchronize(); stages["batch"] += time.perf_counter() - t
    t = time.perf_counter(); res = r.engine.train_step(b, gt, global_batch=ng); torch.cuda.synchronize(); stages["step"] += time.perf_counter() - t
    t = time.perf_counter(); acc[0] += res["loss"][0].double() * ng; acc[1] += res["hits"][1].double(); acc[2] += hi - lo; torch.cuda.synchronize(); stages["acc"] += time.perf_counter() - t
    n += 1
print({k: "%.3f ms" % (v / n * 1e3) for k, v in stages.items()})
pr = cProfile.Profile(); pr.enable(); r.run_epoch(3); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
