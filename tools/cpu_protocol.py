"""BASELINE.md section 3, as written: the reference-faithful CPU path timed on the host cores of the GPU box.

    python tools/cpu_protocol.py [--out profiles/r4_cpu_protocol.json] [--threads N] [--heldout 2048]

What is timed: oracle.FaithfulCPUModel (the 24-iteration cat + Linear + ReLU + Dropout loop, softmax + bmm expected answer
embedding, CrossEntropyLoss(sum) / B, autograd backward, torch.optim.Adam: op for op what /root/reference's
vqa/models/cx.py:280-331 and counterexamples.py:325-339 execute; validated against the imported reference on the committed
fixtures, tests/test_oracle_golden.py) at BASELINE.json configs[0]: synthetic 2048-d features, K = 24, B = 32, H = 256, L = 1,
dropout 0.25, Adam lr 1e-4, ONE epoch over a 16 384-triplet synthetic set = 512 steps (data seed 1234, weights
torch.manual_seed(42), shuffle random.seed(42)).
Protocol: all usable cores (cgroup quota / affinity, as bench.py counts them), 10 warm-up steps, then 5 timed runs of 50 steps
each (the median is the figure), the rest of the epoch untimed; loss of the last step and Recall@1/@5 of the trained model on
held-out triplets of the same planted distribution; then a 1-thread run (10 warm-up + 30 timed steps on the trained weights).
bench.py keeps its bounded sample (50 steps); this is the long form, run once per round and committed under profiles/.
Test infrastructure: the oracle is the thing measured here as the CPU BASELINE, never the product."""
import argparse, json, os, platform, random, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
import numpy as np
import torch
from oracle import ncx_oracle as orc
from neuralcx.synth import SyntheticCX
from bench import usable_cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--triplets", type=int, default=16384)
    ap.add_argument("--heldout", type=int, default=2048)
    ap.add_argument("--n_img", type=int, default=82783)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--run-steps", type=int, default=50)
    a = ap.parse_args()
    threads = a.threads or usable_cores()
    torch.set_num_threads(threads)
    d = orc.Dims()
    B, p_drop, lr = 32, 0.25, 1e-4
    steps = a.triplets // B
    t_setup = time.time()
    data = SyntheticCX(n_triplets=a.triplets + a.heldout, K=d.K, n_img=a.n_img, seed=1234, device="cpu")
    random.seed(42)
    order = list(range(a.triplets)); random.shuffle(order)                 # batchify: one in-place shuffle (counterexamples.py:509-511)
    m = orc.FaithfulCPUModel(d, drop_p=p_drop, seed=42)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    print("setup %.0f s; %d threads; %d steps of batch %d" % (time.time() - t_setup, threads, steps, B), flush=True)

    def make(i):
        sel = torch.tensor(order[i * B:(i + 1) * B], dtype=torch.int64)
        b, gt = data.batch(sel, first_id=i * B)
        return dict(image_features=data.feats[b.img_idx.long()], q_emb=b.q_emb, z_orig=b.z_orig, z_knns=b.z_knns, a_knns=b.a_knns,
                    answer_aids=b.answer_aids.long(), gt=gt.long())

    def step(i, bt):
        m.keep_masks = [orc.dropout_keep_mask((42 << 32) ^ (i + 1), 1, B * d.K, d.H, p_drop)]
        scores = m(bt["image_features"], bt["q_emb"], bt["z_orig"], bt["z_knns"], bt["a_knns"], bt["answer_aids"])
        loss = torch.nn.functional.cross_entropy(scores, bt["gt"], reduction="sum") / B
        opt.zero_grad(); loss.backward(); opt.step()
        return float(loss)

    warm = 10
    rates, i, loss = [], 0, None
    for _ in range(warm):
        loss = step(i, make(i)); i += 1
    for r in range(a.runs):
        batches = [make(i + j) for j in range(a.run_steps)]                # (input synthesis is outside the timed region, as on the GPU)
        t0 = time.perf_counter()
        for j in range(a.run_steps):
            loss = step(i + j, batches[j])
        dt = time.perf_counter() - t0
        i += a.run_steps
        rates.append(B * a.run_steps / dt)
        print("run %d: %.1f triplets/s (loss %.4f)" % (r, rates[-1], loss), flush=True)
    while i < steps:
        loss = step(i, make(i)); i += 1
        if i % 64 == 0:
            print("step %d / %d  loss %.4f" % (i, steps, loss), flush=True)
    m.eval()
    ranks, hl = [], []
    with torch.no_grad():
        for lo in range(0, a.heldout, 256):
            sel = torch.arange(a.triplets + lo, a.triplets + min(lo + 256, a.heldout))
            b, gt = data.batch(sel, first_id=a.triplets + lo)
            sc = m(data.feats[b.img_idx.long()], b.q_emb, b.z_orig, b.z_knns, b.a_knns, b.answer_aids.long())
            ranks.append(orc.rank_of_gt(sc.numpy(), gt.numpy()))
            hl.append(float(torch.nn.functional.cross_entropy(sc, gt.long(), reduction="sum")))
    rank = np.concatenate(ranks)
    m.train()
    torch.set_num_threads(1)
    one = [make(j % steps) for j in range(40)]
    for j in range(10):
        step(steps + j, one[j])
    t0 = time.perf_counter()
    for j in range(10, 40):
        step(steps + j, one[j])
    rate1 = B * 30 / (time.perf_counter() - t0)
    cpu = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    out = dict(protocol="BASELINE.md section 3: configs[0] (B=32, K=24, 2048-d feats, H=256, L=1, dropout 0.25, Adam lr 1e-4), one %d-step epoch over %d synthetic "
                        "triplets; 10 warm-up steps, %d timed runs of %d steps, median" % (steps, a.triplets, a.runs, a.run_steps),
               value=round(float(np.median(rates)), 2), unit="triplets/s", cores=threads, runs=[round(r, 2) for r in rates],
               one_thread=dict(value=round(rate1, 2), unit="triplets/s", cores=1, steps=30),
               after_epoch=dict(final_train_loss=round(loss, 5), heldout_loss=round(sum(hl) / a.heldout, 5), heldout_triplets=a.heldout,
                                recall_at_1=round(float((rank < 1).mean()), 4), recall_at_5=round(float((rank < 5).mean()), 4),
                                chance="0.0417 / 0.2083"),
               host=dict(cpu=cpu, logical_cpus=os.cpu_count(), usable_cores=threads, torch=torch.__version__, python=platform.python_version(),
                         blas=("mkl" if torch.backends.mkl.is_available() else "other") + ("+mkldnn" if torch.backends.mkldnn.is_available() else "")),
               kind="port")
    print(json.dumps(out), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
