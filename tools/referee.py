"""fp64 referee for the training-equivalence claim: on the full-width training test data (tests/test_dropin_gpu.py),
(1) one batch: every gradient tensor of the HIP path and of the fp32 CPU oracle against the fp64 oracle;
(2) a training run: per-step loss of HIP, fp32 oracle (1 thread / all threads) against the fp64 oracle.
usage: python tools/referee.py [steps] [lr]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import ncx_oracle as orc
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
