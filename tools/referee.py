"""fp64 referee for the training-equivalence claim.

Single run (the round-3 table; needs the GPU):   python tools/referee.py [steps] [lr] [held]
  (1) one batch: every gradient tensor of the HIP path and of the fp32 CPU oracle against the fp64 oracle;
  (2) a training run: per-step loss of HIP, fp32 oracle (1 thread / all threads) against the fp64 oracle;
  (3) held-out Recall of the three trained models.

Many seeds (round 4: north_star's "Recall@5 within +-0.1 of the reference" as a mean with a standard error, instead of one
seed that cannot separate noise from bias).  For every seed s: data SyntheticCX(seed 1000+s) generated on the CPU generator
(identical on every machine; both sides store checksums and --merge refuses to compare different data), weights
init_params(seed 42+s), 200 Adam steps of batch 32 at lr 1e-4 with dropout 0.25 from the counter-based generator, then
4 096 held-out triplets.  The CPU sides (the reference-faithful fp32 oracle = "ref32", the same oracle in fp64 = "f64") do
not need a GPU and run wherever there are cores; the HIP side needs the card and takes seconds:
    python tools/referee.py --seeds 8 --side cpu --out profiles/r4_referee_cpu.npz          (resumable per seed)
    python tools/referee.py --seeds 8 --side hip --out profiles/r4_referee_hip.npz
    python tools/referee.py --merge profiles/r4_referee_cpu.npz profiles/r4_referee_hip.npz --table profiles/r4_referee_table.json
The table holds, per seed and per k in {1, 5}: Recall of the three sides in percent, the pairwise differences, the number of
triplets classified differently, and over the seeds the mean and standard error of (HIP - ref32), (ref32 - f64), (HIP - f64).
tests/test_referee_table_cpu.py asserts |mean(HIP - ref32)| <= 0.1 pt on the committed table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import ncx_oracle as orc


def _multi_seed_main(argv):
    import argparse, json
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--side", choices=("cpu", "hip"))
    ap.add_argument("--out")
    ap.add_argument("--merge", nargs=2)
    ap.add_argument("--table")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--held", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args(argv)
    if a.merge:
        return _merge(a.merge[0], a.merge[1], a.table)
    B, p_drop, lr, steps, HELD = 32, 0.25, 1e-4, a.steps, a.held
    d = orc.Dims()
    if a.threads:
        torch.set_num_threads(a.threads)
    done = dict(np.load(a.out)) if (a.out and os.path.exists(a.out)) else {}
    hip = a.side == "hip"
    if hip:
        from neuralcx.engine import NeuralCXEngine
        from neuralcx import ops as _ops
    from neuralcx.synth import SyntheticCX
    for s in range(a.first_seed, a.first_seed + a.seeds):
        tag = "s%d_" % s
        if tag + "check" in done:
            print("seed %d: already in %s" % (s, a.out), flush=True)
            continue
        t0 = time.time()
        data = SyntheticCX(n_triplets=B * steps + HELD, n_img=1024, seed=1000 + s, device="cpu")
        params = orc.init_params(d, seed=42 + s)
        check = []

        def cpu_batch(b, gt, dt):
            return dict(image_features=data.feats[b.img_idx.long()].to(dt), q_emb=b.q_emb.to(dt), z_orig=b.z_orig.to(dt), z_knns=b.z_knns.to(dt),
                        a_knns=b.a_knns.to(dt), answer_aids=b.answer_aids.long(), gt=gt.long())

        def to_dev(b, feats_dev):
            mv = lambda t: t.to(DEV)
            return _ops.Batch(feats_dev, mv(b.img_idx), mv(b.q_emb), mv(b.z_orig), mv(b.z_knns), mv(b.a_knns), mv(b.answer_aids))
        if hip:
            eng = NeuralCXEngine(H=d.H, L=d.L, drop_p=p_drop, lr=lr, device=DEV)
            eng.seed = 42 + s
            eng.load_state(params)
            feats_dev = data.feats.to(DEV)
        else:
            sides = {"ref32": torch.float32, "f64": torch.float64}
            cur = {n: {k: v.to(dt) for k, v in params.items()} for n, dt in sides.items()}
            st = {n: orc.AdamState() for n in sides}
        losses = {n: [] for n in (("hip",) if hip else ("ref32", "f64"))}
        for it in range(steps):
            b, gt = data.batch(torch.arange(it * B, (it + 1) * B), first_id=it * B)
            if it in (0, steps - 1):
                check += [float(b.a_knns.double().sum()), float(b.q_emb.double().sum()), float(gt.sum())]
            if hip:
                r = eng.train_step(to_dev(b, feats_dev), gt.to(DEV))
                losses["hip"].append(float(r["loss"]))
            else:
                seed = ((42 + s) << 32) ^ (it + 1)                       # engine: (seed << 32) ^ (rank << 24) ^ step_count, rank 0
                mask = orc.dropout_keep_mask(seed, 1, B * d.K, d.H, p_drop)
                for n, dt in sides.items():
                    cur[n], _, l, _ = orc.train_step(cur[n], d, cpu_batch(b, gt, dt), st[n], lr=lr, drop_p=p_drop, keep_masks=[mask.to(dt)])
                    losses[n].append(float(l))
            if it % 50 == 49:
                print("seed %d step %d  %s  (%.0f s)" % (s, it + 1, {n: round(v[-1], 6) for n, v in losses.items()}, time.time() - t0), flush=True)
        ranks = {n: [] for n in losses}
        for lo in range(0, HELD, 512):
            hb, hgt = data.batch(torch.arange(B * steps + lo, B * steps + lo + 512), first_id=B * steps + lo)
            if lo == 0:
                check += [float(hb.a_knns.double().sum()), float(hgt.sum())]
            if hip:
                ranks["hip"].append(eng.eval_step(to_dev(hb, feats_dev), hgt.to(DEV))["rank"].cpu().numpy())
            else:
                for n, dt in sides.items():
                    hc = cpu_batch(hb, hgt, dt)
                    with torch.no_grad():
                        sc = orc.forward_faithful(cur[n], d, hc["image_features"], hc["q_emb"], hc["z_orig"], hc["z_knns"], hc["a_knns"], hc["answer_aids"])
                    ranks[n].append(orc.rank_of_gt(sc.numpy(), hc["gt"].numpy()))
        for n in losses:
            done[tag + n + "_rank"] = np.concatenate(ranks[n]).astype(np.int16)
            done[tag + n + "_loss"] = np.asarray(losses[n], np.float64)
        done[tag + "check"] = np.asarray(check, np.float64)
        if a.out:
            np.savez_compressed(a.out, **done)
        print("seed %d done in %.0f s: %s" % (s, time.time() - t0, {n: "R@1 %.3f R@5 %.3f" % (100.0 * (done[tag + n + "_rank"] < 1).mean(),
              100.0 * (done[tag + n + "_rank"] < 5).mean()) for n in losses}), flush=True)


def _merge(cpu_file, hip_file, table_file):
    import json
    c, h = np.load(cpu_file), np.load(hip_file)
    seeds = sorted(int(k[1:].split("_")[0]) for k in c.files if k.endswith("_check"))
    seeds = [s for s in seeds if ("s%d_check" % s) in h.files]
    rows, diffs = [], {}
    for s in seeds:
        t = "s%d_" % s
        # (sums of ~1.5 M doubles: the order of torch's parallel reduction follows the host's thread count, so the two machines agree to
        #  ~1e-16 relative, not bit for bit; different data would differ in the first digits)
        if not np.allclose(c[t + "check"], h[t + "check"], rtol=1e-12, atol=0.0):
            raise SystemExit("seed %d: the two sides trained on different data (checksums %s vs %s)" % (s, c[t + "check"], h[t + "check"]))
        R = {"hip": h[t + "hip_rank"], "ref32": c[t + "ref32_rank"], "f64": c[t + "f64_rank"]}
        row = {"seed": s, "held_out": int(len(R["hip"])), "max_abs_loss_diff_first40": {
            "hip_f64": float(np.abs(h[t + "hip_loss"][:40] - c[t + "f64_loss"][:40]).max()),
            "ref32_f64": float(np.abs(c[t + "ref32_loss"][:40] - c[t + "f64_loss"][:40]).max())}}
        for k in (1, 5):
            rec = {n: 100.0 * float((R[n] < k).mean()) for n in R}
            row["recall@%d" % k] = rec
            for a_, b_ in (("hip", "ref32"), ("ref32", "f64"), ("hip", "f64")):
                key = "%s-%s@%d" % (a_, b_, k)
                diffs.setdefault(key, []).append(rec[a_] - rec[b_])
                row.setdefault("disagree@%d" % k, {})["%s/%s" % (a_, b_)] = int(((R[a_] < k) != (R[b_] < k)).sum())
        rows.append(row)
    summary = {}
    for key, v in diffs.items():
        v = np.asarray(v)
        summary[key] = {"mean_pt": float(v.mean()), "stderr_pt": float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else None,
                        "min_pt": float(v.min()), "max_pt": float(v.max()), "n_seeds": int(len(v))}
    out = {"protocol": "200 Adam steps of batch 32 at lr 1e-4, dropout 0.25 (counter-based masks shared by the sides), full widths "
                       "(K=24, dv=2048, dq=da=2400, dz=360, A=2000, H=256, L=1); 4096 held-out triplets per seed; data seed 1000+s, weight seed 42+s; "
                       "ref32 = reference-faithful fp32 CPU oracle (torch CPU), f64 = the same in fp64, hip = the HIP engine",
           "summary": summary, "per_seed": rows}
    if table_file:
        with open(table_file, "w") as f:
            json.dump(out, f, indent=1)
    for key in sorted(summary):
        m = summary[key]
        print("%-16s mean %+.4f pt  stderr %s  range [%+.3f, %+.3f]  (n = %d)" % (key, m["mean_pt"], "%.4f" % m["stderr_pt"] if m["stderr_pt"] is not None else "-",
              m["min_pt"], m["max_pt"], m["n_seeds"]))
    return out


if any(x.startswith("--") for x in sys.argv[1:]):
    DEV = "cuda:0"
    _multi_seed_main(sys.argv[1:])
    sys.exit(0)

from neuralcx import ops
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
DEV = "cuda:0"
d = orc.Dims()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
B, p_drop = 32, 0.25
HELD = int(sys.argv[3]) if len(sys.argv) > 3 else 0
data = SyntheticCX(n_triplets=B * steps + max(128, HELD), n_img=1024, seed=77, device=DEV)
feats_cpu = data.feats.cpu()


def cpu_batch(b, gt, dtype=torch.float32):
    f = lambda t: t.cpu().to(dtype)
    return dict(image_features=feats_cpu[b.img_idx.cpu().long()].to(dtype), q_emb=f(b.q_emb), z_orig=f(b.z_orig), z_knns=f(b.z_knns),
                a_knns=f(b.a_knns), answer_aids=b.answer_aids.cpu().long(), gt=gt.cpu().long())


params = orc.init_params(d, seed=42)
# ---- (1) one batch ------------------------------------------------------------------------------------------------
b, gt = data.batch(torch.arange(0, B, device=DEV), first_id=0)
seed = (42 << 32) ^ 1
p = {ops.STATE_TO_FIELD[k]: v.to(DEV) for k, v in params.items()}
dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A, training=True, drop_p=p_drop, seed=seed)
ws = ops.alloc_workspace(dims, DEV)
scores = ops.forward(dims, b, p, ws)
lrk = ops.ranking_loss(scores, gt)
grads = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
ops.backward(dims, b, p, ws, lrk["dscores"], grads)
torch.cuda.synchronize()
mask = orc.dropout_keep_mask(seed, 1, B * d.K, d.H, p_drop)
s32, l32, g32 = orc.loss_and_grads(params, d, cpu_batch(b, gt), drop_p=p_drop, keep_masks=[mask])
p64 = {k: v.double() for k, v in params.items()}
s64, l64, g64 = orc.loss_and_grads(p64, d, cpu_batch(b, gt, torch.float64), drop_p=p_drop, keep_masks=[mask.double()])
print("scores: HIP-f64 %.2e  o32-f64 %.2e   loss: HIP-f64 %.2e o32-f64 %.2e" % (float((scores.cpu().double() - s64).abs().max()),
      float((s32.double() - s64).abs().max()), abs(float(lrk["loss"]) - float(l64)), abs(float(l32) - float(l64))))
for k, ref in g64.items():
    gh = grads[ops.STATE_TO_FIELD[k]].cpu().double(); go = g32[k].double()
    mx = float(ref.abs().max())
    eh, eo = (gh - ref).abs(), (go - ref).abs()
    small = ref.abs() < 1e-3 * mx
    print("%-24s max|g| %.2e | HIP: max %.2e (%.1e of max) rms %.2e | o32: max %.2e (%.1e) rms %.2e | sign flips on |g|>1e-10: HIP %d o32 %d of %d"
          % (k, mx, float(eh.max()), float(eh.max()) / max(mx, 1e-30), float(eh.pow(2).mean().sqrt()), float(eo.max()), float(eo.max()) / max(mx, 1e-30),
             float(eo.pow(2).mean().sqrt()), int(((gh * ref < 0) & (ref.abs() > 1e-10)).sum()), int(((go * ref < 0) & (ref.abs() > 1e-10)).sum()), ref.numel()))
# ---- (2) training -------------------------------------------------------------------------------------------------
eng = NeuralCXEngine(H=d.H, L=d.L, drop_p=p_drop, lr=lr, device=DEV)
eng.load_state(params)
runs = {"o32_t1": (torch.float32, 1), "o32_tN": (torch.float32, max(1, len(os.sched_getaffinity(0)) if len(os.sched_getaffinity(0)) <= 16 else 16)), "f64": (torch.float64, 16)}
if HELD:
    runs.pop("o32_t1")
state = {n: ({k: v.to(dt) for k, v in params.items()}, orc.AdamState()) for n, (dt, th) in runs.items()}
hist = {n: [] for n in list(runs) + ["hip"]}
for s in range(steps):
    b, gt = data.batch(torch.arange(s * B, (s + 1) * B, device=DEV), first_id=s * B)
    r = eng.train_step(b, gt)
    hist["hip"].append(float(r["loss"]))
    sd = (eng.seed << 32) ^ eng.step_count
    mask = orc.dropout_keep_mask(sd, 1, B * d.K, d.H, p_drop)
    for n, (dt, th) in runs.items():
        torch.set_num_threads(th)
        cur, st = state[n]
        cur, _, l, _ = orc.train_step(cur, d, cpu_batch(b, gt, dt), st, lr=lr, drop_p=p_drop, keep_masks=[mask.to(dt)])
        state[n] = (cur, st)
        hist[n].append(float(l))
    if "o32_t1" not in hist or not hist["o32_t1"]:
        hist["o32_t1"] = hist["o32_tN"]
    print("step %3d  loss f64 %.7f | HIP-f64 %.2e | o32_t1-f64 %.2e | o32_tN-f64 %.2e | t1-tN %.2e" % (
        s, hist["f64"][-1], abs(hist["hip"][-1] - hist["f64"][-1]), abs(hist["o32_t1"][-1] - hist["f64"][-1]),
        abs(hist["o32_tN"][-1] - hist["f64"][-1]), abs(hist["o32_t1"][-1] - hist["o32_tN"][-1])), flush=True)
# weights after training
sd_h = eng.state_dict()
for k in params:
    w64 = state["f64"][0][k]
    print("%-24s |w_hip - w64| max %.2e   |w_o32 - w64| max %.2e" % (k, float((sd_h[k].cpu().double() - w64).abs().max()),
          float((state["o32_tN"][0][k].double() - w64).abs().max())))

# ---- (3) held-out Recall of the three trained models --------------------------------------------------------------
if HELD:
    torch.set_num_threads(16)
    ranks = {"hip": [], "o32_tN": [], "f64": []}
    scs = {"hip": [], "o32_tN": [], "f64": []}
    gts = []
    for lo in range(0, HELD, 256):
        hb, hgt = data.batch(torch.arange(B * steps + lo, B * steps + min(lo + 256, HELD), device=DEV), first_id=B * steps + lo)
        ev = eng.eval_step(hb, hgt)
        ranks["hip"].append(ev["rank"].cpu().numpy()); scs["hip"].append(ev["scores"].cpu().double().numpy())
        gts.append(hgt.cpu().numpy())
        for n, dt in (("o32_tN", torch.float32), ("f64", torch.float64)):
            hc = cpu_batch(hb, hgt, dt)
            with torch.no_grad():
                sc = orc.forward_faithful(state[n][0], d, hc["image_features"], hc["q_emb"], hc["z_orig"], hc["z_knns"], hc["a_knns"], hc["answer_aids"])
            ranks[n].append(orc.rank_of_gt(sc.numpy(), hc["gt"].numpy())); scs[n].append(sc.double().numpy())
        print("held-out %d" % lo, flush=True)
    gt_all = np.concatenate(gts)
    R = {n: np.concatenate(v) for n, v in ranks.items()}
    S = {n: np.concatenate(v) for n, v in scs.items()}
    cen = lambda x: x - x.mean(1, keepdims=True)
    for n in R:
        print("%-7s Recall@1 %.4f (%d)  Recall@5 %.4f (%d)" % (n, (R[n] < 1).mean(), (R[n] < 1).sum(), (R[n] < 5).mean(), (R[n] < 5).sum()))
    for a_, b_ in (("hip", "f64"), ("o32_tN", "f64"), ("hip", "o32_tN")):
        print("%s vs %s: max |centred score diff| %.2e; disagreeing triplets @1 %d @5 %d; rank differs on %d" % (
            a_, b_, np.abs(cen(S[a_]) - cen(S[b_])).max(), ((R[a_] < 1) != (R[b_] < 1)).sum(), ((R[a_] < 5) != (R[b_] < 5)).sum(), (R[a_] != R[b_]).sum()))
        # margin of the gt score to the k-th / (k+1)-th best of the reference side on the disagreeing triplets
        for k in (1, 5):
            dis = np.nonzero((R[a_] < k) != (R[b_] < k))[0]
            srt = -np.sort(-S[b_], axis=1); sg = S[b_][np.arange(len(gt_all)), gt_all]
            if len(dis):
                m = np.minimum(np.abs(sg[dis] - srt[dis, k - 1]), np.abs(sg[dis] - srt[dis, k]))
                print("    @%d margins of the %d disagreeing triplets on the %s side: max %.2e" % (k, len(dis), b_, m.max()))
