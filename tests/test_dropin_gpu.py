"""GPU: the reference-shaped surfaces on top of the HIP path -- NeuralModel (nn.Module + autograd),
the training engine over several steps (loss curve + Recall@5 vs the CPU oracle), and the CLI."""
import os

import numpy as np
import pytest
import torch

from conftest import PKG
from oracle import ncx_oracle as orc
from helpers import grad_tol

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tiny_opt(dv=64, dq=48, dz=16):
    return dict(arch="MutanNoAtt", seq2vec=dict(arch="gru", emb_size=8, dropout=0.0),
                fusion=dict(dim_v=dv, dim_q=dq, dim_hv=dz, dim_hq=dz, dim_mm=dz, R=3, dropout_v=0.5, dropout_q=0.5,
                            activation_v="tanh", activation_q="tanh", dropout_hv=0, dropout_hq=0),
                classif=dict(dropout=0.5))


@pytest.mark.parametrize("L", [1, 3])
def test_neuralmodel_module_forward_and_autograd(L):
    import vqa.models as M
    from vqa.models.cx import NeuralModel
    torch.manual_seed(0)
    A, B = 20, 6
    vqa = M.factory(_tiny_opt(), ["w%d" % i for i in range(30)], ["a%d" % i for i in range(A)], cuda=True, data_parallel=False)
    spec = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True, q_emb=True, a_emb=True, z_emb=True)
    m = NeuralModel(model_spec=spec, dim_h=16, n_layers=L, emb=None, drop_p=0.25, vqa_model=vqa, knn_size=24, trainable_vqa=False).cuda()
    d = orc.Dims(dv=64, dq=48, dz=16, A=A, H=16, L=L)
    params = orc.init_params(d, seed=3, gain=3.0)
    m.load_state_dict({**{k: v for k, v in m.state_dict().items() if k.startswith("vqa_model.")}, **params})
    m.eval()                                                     # dropout off (cx_model.eval(), counterexamples.py:451)
    feats = (torch.randn(B, 25, 64).abs() * 0.45).to(DEV)
    wids = torch.zeros(B, 26, dtype=torch.long)
    for b in range(B):
        n = 3 + b
        wids[b, :n] = torch.randint(1, 31, (n,))
    wids = wids.to(DEV)
    aids = torch.randint(0, A, (B,)).to(DEV)
    gt = torch.randint(0, 24, (B,)).to(DEV)
    scores = m(feats, wids, aids)                                # reference call signature (cx.py:261)
    assert scores.shape == (B, 24)
    loss = torch.nn.CrossEntropyLoss(reduction="sum")(scores, gt) / B          # counterexamples.py:310,334
    m.zero_grad()
    loss.backward()                                              # counterexamples.py:338 -> ncx_backward
    a_o, z_o, a_k, z_k, q = [t.cpu() for t in m.vqa_forward(feats, wids)]
    batch = dict(image_features=feats.cpu(), q_emb=q, z_orig=z_o, z_knns=z_k, a_knns=a_k, answer_aids=aids.cpu(), gt=gt.cpu())
    s_ref, l_ref, g_ref = orc.loss_and_grads(params, d, batch)
    assert float((scores.detach().cpu() - s_ref).abs().max()) <= 1e-4
    assert abs(float(loss.detach()) - float(l_ref)) <= 1e-5
    own = {n: p for n, p in m.named_parameters() if not n.startswith("vqa_model.")}
    for n, p in own.items():
        ref = g_ref[n]
        tol = grad_tol(n, ref.numpy(), 1e-4)
        assert float((p.grad.cpu() - ref).abs().max()) <= tol, n
    for n, p in m.named_parameters():
        if n.startswith("vqa_model."):
            assert p.grad is None                                # frozen VQA model (cx.py:79-80)
    # torch.optim.Adam on the module parameters works as in the reference (counterexamples.py:275-276)
    opt = torch.optim.Adam([p for p in own.values()], lr=1e-4)
    opt.step()
    st = orc.AdamState()
    new = orc.adam_update(params, {k: v for k, v in g_ref.items()}, st, lr=1e-4)
    for n, p in own.items():
        if n != "out.bias":      # zero gradient in maths; elsewhere ignore entries whose gradient is ~eps (Adam's sqrt(v)+1e-8)
            big = g_ref[n].abs() > 1e-5 * g_ref[n].abs().max()
            assert float((p.detach().cpu() - new[n]).abs()[big].max()) <= 2e-6, n


@pytest.mark.parametrize("off", [("v_emb",), ("q_emb",), ("z_emb",), ("v_rank",), ("a_emb",), ("v_mult", "v_dist"), ("q_emb", "z_emb", "a_emb")])
def test_neuralmodel_lesion_specs_run_like_the_reference(off):
    """model_spec lesions (cx.py:265-307): noise-substituted segments (v_emb, q_emb, z_emb, v_rank, a_emb) and zeroed ones
    (v_mult, v_dist) go through the module's forward / backward; noise lesions are reproducible under torch.manual_seed."""
    import vqa.models as M
    from vqa.models.cx import NeuralModel
    A, B = 20, 4
    torch.manual_seed(1)
    vqa = M.factory(_tiny_opt(), ["w%d" % i for i in range(30)], ["a%d" % i for i in range(A)], cuda=True, data_parallel=False)
    spec = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True, q_emb=True, a_emb=True, z_emb=True)
    for k in off:
        spec[k] = False
    m = NeuralModel(model_spec=spec, dim_h=16, n_layers=2, emb=None, drop_p=0.25, vqa_model=vqa, knn_size=24, trainable_vqa=False).cuda()
    m.eval()
    feats = (torch.randn(B, 25, 64).abs() * 0.45).to(DEV)
    wids = torch.randint(1, 31, (B, 26)).to(DEV)
    aids = torch.randint(0, A, (B,)).to(DEV)
    torch.manual_seed(7); s1 = m(feats, wids, aids)
    torch.manual_seed(7); s2 = m(feats, wids, aids)
    assert s1.shape == (B, 24) and torch.isfinite(s1).all() and torch.equal(s1, s2)
    loss = torch.nn.CrossEntropyLoss(reduction="sum")(s1, torch.randint(0, 24, (B,)).to(DEV)) / B
    loss.backward()
    assert torch.isfinite(m.linear_1.weight.grad).all() and float(m.linear_1.weight.grad.abs().sum()) > 0
    o = 3 * 64                                                       # v_orig | v_other | v_mult | v_dist | v_rank ...
    if "v_mult" in off:
        assert float(m.linear_1.weight.grad[:, 2 * 64:3 * 64].abs().max()) == 0.0 and float(m.linear_1.weight.grad[:, o].abs().max()) == 0.0


def test_engine_training_matches_oracle_training():
    """30 Adam steps with dropout (shared counter-based masks) on identical synthetic data: per-step loss within
    1e-4 of the CPU oracle's training run, and Recall@1/@5 on held-out triplets identical (+-0.1 pt allowed)."""
    from neuralcx import ops
    from neuralcx.engine import NeuralCXEngine
    d = orc.Dims(dv=64, dq=48, dz=16, A=20, H=32, L=2)
    B, steps, p_drop, lr = 16, 30, 0.25, 1e-3
    rng = np.random.default_rng(7)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))

    def make(n):
        feats = np.abs(rng.standard_normal((n, d.K + 1, d.dv))) * 0.45
        dist = np.linalg.norm(feats[:, :1] - feats[:, 1:], axis=2)
        gt = dist.argmin(1)                                         # learnable: the closest candidate
        return dict(image_features=t(feats), q_emb=t(rng.standard_normal((n, d.dq)) * 0.3), z_orig=t(rng.standard_normal((n, d.dz))),
                    z_knns=t(rng.standard_normal((n, d.K, d.dz))), a_knns=t(rng.standard_normal((n, d.K, d.A)) * 2),
                    answer_aids=torch.from_numpy(rng.integers(0, d.A, size=n)), gt=torch.from_numpy(gt))
    train = [make(B) for _ in range(5)]
    held = make(64)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=p_drop, lr=lr, device=DEV)
    params = orc.init_params(d, seed=21, gain=2.0)
    eng.load_state(params)
    st = orc.AdamState()
    cur = {k: v.clone() for k, v in params.items()}
    to_batch = lambda bt: ops.Batch.from_dense(*[bt[k].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    for s in range(steps):
        bt = train[s % len(train)]
        r = eng.train_step(to_batch(bt), bt["gt"].to(DEV).to(torch.int32))
        seed = (eng.seed << 32) ^ eng.step_count
        masks = [orc.dropout_keep_mask(seed, l, B * d.K, d.H, p_drop) for l in range(1, d.L + 1)]
        cur, _, l_ref, _ = orc.train_step(cur, d, bt, st, lr=lr, drop_p=p_drop, keep_masks=masks)
        assert abs(float(r["loss"]) - float(l_ref)) <= 1e-4, (s, float(r["loss"]), float(l_ref))
    ev = eng.eval_step(to_batch(held), held["gt"].to(DEV).to(torch.int32))
    s_ref = orc.forward_faithful(cur, d, held["image_features"], held["q_emb"], held["z_orig"], held["z_knns"], held["a_knns"], held["answer_aids"])
    # out.bias has a zero gradient in maths; Adam turns its round-off into +-lr per step, differently on every platform
    # (SURVEY 7): after training, logits are compared modulo a per-row constant (ranking and loss are unaffected).
    centred = lambda x: x - x.mean(1, keepdim=True)
    assert float((centred(ev["scores"].cpu()) - centred(s_ref)).abs().max()) <= 1e-3      # 30 Adam steps on both sides
    for k, i in ((1, 0), (5, 1)):
        r_hip = 100.0 * int(ev["hits"][i]) / 64
        r_ref = 100.0 * orc.recall_at_k(s_ref, held["gt"], k).sum() / 64
        assert abs(r_hip - r_ref) <= 0.1 + 1e-9, (k, r_hip, r_ref)
    l_held = float(orc.ranking_loss(s_ref, held["gt"]))
    assert abs(float(ev["loss"]) - l_held) <= 1e-3


def test_full_width_training_against_the_fp64_referee_and_heldout_recall():
    """north_star: "Recall@5 matching the reference to +-0.1 on identical synthetic data".  Full model widths (2048-d features,
    2400-d question / answer embeddings, 2000 answers, H=256, L=1, dropout 0.25), 200 Adam steps of batch 32 (BASELINE
    configs[0]'s batch) at the reference's lr 1e-4 (options/cx/*.yaml:61) on planted synthetic triplets, then 4 096 held-out
    triplets.  THREE trainings on identical batches with identical (counter-based) dropout masks: the HIP engine, the
    reference-faithful fp32 CPU oracle (the PyTorch CPU path: 24-iteration cat + Linear loop, autograd, Adam), and the
    same oracle in fp64 as REFEREE.

    What the referee shows (tools/referee.py prints the full table): two correct fp32 implementations of this
    training do not stay together -- a ReLU input within rounding of zero switches its unit on one side only, Adam's
    normalised update amplifies whatever differs, and after ~100 steps the trajectories are decorrelated at the 1e-2 level of
    the scores.  The fp32 CPU reference itself separates from fp64 by 1e-3 in the loss within 40 steps; the HIP path stays
    within 5e-5 of fp64 over the same steps (its gradients are 1e-3..1e-4 times closer to fp64 than torch's fp32 autograd).
    So the checks are:
      * steps 0 .. 39, against the referee: |loss_HIP - loss_f64| <= 2e-4 (round 1's bound; step 0: <= 1e-5);
      * after 40 steps, 4 096 held-out triplets HIP vs referee: Recall@1 / @5 totals within 0.1 pt, and every triplet
        whose ground truth is further than 1e-3 from its rank boundary classified identically (count of excluded
        near-ties reported: 17 / 36 of 4 096 measured, bound 2 %);
      * after 200 steps (Recall@5 > 0.40, chance 0.208) the three trajectories are decorrelated: the fp32 reference and the
        HIP run each classify ~60 of the 4 096 triplets differently from the referee at k = 5 (measured: 61 and 61; 25 and 18
        at k = 1), i.e. the REFERENCE's Recall@5 is only defined to ~+-0.2 pt at this horizon (0.07 .. 0.22 pt observed on two
        held-out sets).  Checks: the HIP run disagrees with the referee on no more triplets than the fp32 reference does
        (x 2 + 20: the reference's own count depends on the host's BLAS), and its total differs from the referee's by no more than a symmetric random walk over the disagreeing
        triplets allows (3 sigma = 3 sqrt(n)): no systematic bias.  Totals are printed in percent."""
    from neuralcx.engine import NeuralCXEngine
    from neuralcx.synth import SyntheticCX
    d = orc.Dims()                                           # K=24, dv=2048, dq=2400, dz=360, da=2400, A=2000, H=256, L=1
    B, steps, p_drop, lr, HELD = 32, 200, 0.25, 1e-4, 4096
    data = SyntheticCX(n_triplets=B * steps + HELD, n_img=1024, seed=77, device=DEV)
    eng = NeuralCXEngine(H=d.H, L=d.L, drop_p=p_drop, lr=lr, device=DEV)
    params = orc.init_params(d, seed=42)
    eng.load_state(params)
    feats_cpu = data.feats.cpu()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))

    def cpu_batch(b, gt, dt):
        f = lambda t: t.cpu().to(dt)
        return dict(image_features=feats_cpu[b.img_idx.cpu().long()].to(dt), q_emb=f(b.q_emb), z_orig=f(b.z_orig), z_knns=f(b.z_knns),
                    a_knns=f(b.a_knns), answer_aids=b.answer_aids.cpu().long(), gt=gt.cpu().long())
    sides = {"o32": torch.float32, "f64": torch.float64}
    cur = {n: {k: v.to(dt) for k, v in params.items()} for n, dt in sides.items()}
    st = {n: orc.AdamState() for n in sides}

    def heldout(names):
        """-> {side: rank-of-gt [HELD]}, f64 scores, gt"""
        R = {n: [] for n in names + ["hip"]}
        s64, gts = [], []
        for lo in range(0, HELD, 512):
            hb, hgt = data.batch(torch.arange(B * steps + lo, B * steps + lo + 512, device=DEV), first_id=B * steps + lo)
            R["hip"].append(eng.eval_step(hb, hgt)["rank"].cpu().numpy())
            gts.append(hgt.cpu().numpy())
            for n in names:
                hc = cpu_batch(hb, hgt, sides[n])
                with torch.no_grad():
                    sc = orc.forward_faithful(cur[n], d, hc["image_features"], hc["q_emb"], hc["z_orig"], hc["z_knns"], hc["a_knns"], hc["answer_aids"])
                R[n].append(orc.rank_of_gt(sc.numpy(), hc["gt"].numpy()))
                if n == "f64":
                    s64.append(sc.numpy())
        return {n: np.concatenate(v) for n, v in R.items()}, (np.concatenate(s64) if s64 else None), np.concatenate(gts)

    worst = {"hip": 0.0, "o32": 0.0}
    for s in range(steps):
        b, gt = data.batch(torch.arange(s * B, (s + 1) * B, device=DEV), first_id=s * B)
        r = eng.train_step(b, gt)
        seed = (eng.seed << 32) ^ eng.step_count
        mask = orc.dropout_keep_mask(seed, 1, B * d.K, d.H, p_drop)
        loss = {}
        for n, dt in sides.items():
            cur[n], _, l, _ = orc.train_step(cur[n], d, cpu_batch(b, gt, dt), st[n], lr=lr, drop_p=p_drop, keep_masks=[mask.to(dt)])
            loss[n] = float(l)
        if s < 40:
            worst["hip"] = max(worst["hip"], abs(float(r["loss"]) - loss["f64"]))
            worst["o32"] = max(worst["o32"], abs(loss["o32"] - loss["f64"]))
            assert worst["hip"] <= (1e-5 if s == 0 else 2e-4), (s, float(r["loss"]), loss["f64"])
        if s == 39:
            # (measured on the builder's boxes: HIP 4.5e-5, the fp32 CPU reference 1.6e-3 -- printed, not asserted: which side meets a
            # ReLU kink first depends on the host's BLAS summation order)
            print("first 40 steps: max |loss - loss_f64|  HIP %.2e   fp32 CPU reference %.2e" % (worst["hip"], worst["o32"]))
            R, s64, gtn = heldout(["f64"])
            sg = s64[np.arange(HELD), gtn]
            srt = -np.sort(-s64, axis=1)
            for k in (1, 5):
                margin = np.minimum(np.abs(sg - srt[:, k - 1]) + (R["f64"] < k) * 1e9, np.abs(sg - srt[:, k]) + (R["f64"] >= k) * 1e9)
                near = margin < 1e-3                                   # gt within 1e-3 of the side of the boundary it would cross
                assert ((R["hip"] < k) == (R["f64"] < k))[~near].all(), k
                assert near.mean() <= 0.02, (k, int(near.sum()))           # (measured: 17 and 36 of 4 096)
                n_hip, n_ref = int((R["hip"] < k).sum()), int((R["f64"] < k).sum())
                assert abs(n_hip - n_ref) <= 0.001 * HELD, (k, n_hip, n_ref)
                print("after 40 steps: Recall@%d HIP %d / f64 %d of %d (near-ties excluded from the per-triplet check: %d)" % (k, n_hip, n_ref, HELD, int(near.sum())))
    R, _, _ = heldout(["o32", "f64"])
    for k in (1, 5):
        rec = {n: 100.0 * float((R[n] < k).mean()) for n in R}
        print("after %d steps: Recall@%d  HIP %.3f  fp32 CPU reference %.3f  fp64 referee %.3f  (%%; disagreeing triplets HIP/f64 %d, o32/f64 %d)"
              % (steps, k, rec["hip"], rec["o32"], rec["f64"], int(((R["hip"] < k) != (R["f64"] < k)).sum()), int(((R["o32"] < k) != (R["f64"] < k)).sum())))
        dis_h, dis_o = int(((R["hip"] < k) != (R["f64"] < k)).sum()), int(((R["o32"] < k) != (R["f64"] < k)).sum())
        # Bounds that do not depend on the fp32 CPU run's luck (round 4; the 10-seed table profiles/r4_referee_table.json gives the
        # distribution: HIP / f64 disagreements at k = 5 range 0 .. 141 with mean 67, the reference's own 0 .. 124 with mean 65):
        # an absolute cap at ~mean + 3 sigma, and no bias beyond a symmetric random walk over the HIP run's own disagreements.
        # The north_star's +-0.1 pt itself is asserted per seed at the 40-step horizon above and, at 200 steps, on the MEAN over
        # the 10 seeds by tests/test_referee_table_cpu.py (-0.002 +- 0.067 pt at k = 5).
        assert dis_h <= 200, (k, dis_h, dis_o)
        assert abs(int((R["hip"] < k).sum()) - int((R["f64"] < k).sum())) <= 3.0 * np.sqrt(max(dis_h, 1)), (k, rec, dis_h, dis_o)
        if k == 5:
            assert rec["hip"] > 40.0, rec                                # learned (chance 20.8 %)


def test_cli_synthetic_smoke(tmp_path, capsys):
    import counterexamples as cli
    cli.main(["--synthetic", "--path_opt", os.path.join(PKG, "options", "cx", "neuralcx_256_1_all.yaml"), "-b", "64",
              "--epochs", "1", "--syn_train", "256", "--syn_val", "128", "--syn_images", "2048", "-p", "2", "-t",
              "--project_dir", str(tmp_path)])
    out = capsys.readouterr().out
    assert "Epoch 1 train: loss:" in out and "Epoch 1 val: loss:" in out and "Saved checkpoint" in out
    runs = os.listdir(os.path.join(str(tmp_path), "logs", "cx"))
    assert len(runs) == 1
    base = os.path.join(str(tmp_path), "logs", "cx", runs[0])
    for sub in ("ckpt", "best"):
        state = torch.load(os.path.join(base, sub, "model.ckpt"))
        assert state["linear_1.weight"].shape == (256, 14089) and state["answer_embedding.weight"].shape == (2000, 2400)
        info = torch.load(os.path.join(base, sub, "info.ckpt"))
        assert {"loss", "recall"} <= set(info[-1])
    assert os.path.exists(os.path.join(base, "final_results.txt"))
    runs = os.path.join(str(tmp_path), "runs", os.path.basename(base))
    assert os.path.exists(os.path.join(runs, "val.jsonl"))            # scalars the reference sends to tensorboard


def test_cli_configs3_scale_on_one_gpu(tmp_path, capsys):
    """BASELINE configs[3] at its stated size on the hardware a test can have: ~440 k synthetic triplets (2048-d features over
    a COCO-train-sized table of 82 783 rows, 24 candidates, global batch 512), one epoch through the CLI on ONE GPU -- the
    8-GPU partition of the same epoch is dp.epoch_plan's contiguous slices, covered by the gloo tests.  Every triplet is
    consumed once, the planted ranking signal is learned (chance Recall@5 = 0.208), the last partial batch (440 000 = 859 x 512
    + 192) is kept as the reference's batchify keeps it (counterexamples.py:509-516)."""
    import re
    import counterexamples as cli
    n_train = 440000
    cli.main(["--synthetic", "--path_opt", os.path.join(PKG, "options", "cx", "neuralcx_256_1_all.yaml"), "-b", "512",
              "--epochs", "1", "--syn_train", str(n_train), "--syn_val", "4096", "--syn_images", "82783", "-p", "860",
              "--project_dir", str(tmp_path)])
    out = capsys.readouterr().out
    m = re.search(r"Epoch 1 train: loss: ([0-9.]+), recall: ([0-9.]+), triplets/s: ([0-9.]+)", out)
    assert m, out[-2000:]                                     # printed at step 860 = the partial last batch: all 440 000 seen
    train_loss, train_r5, rate = float(m.group(1)), float(m.group(2)), float(m.group(3))
    v = re.search(r"Epoch 1 val: loss: ([0-9.]+), recall: ([0-9.]+), recall_1: ([0-9.]+), recall_5: ([0-9.]+)", out)
    assert v, out[-2000:]
    val_loss, val_r5, val_r1 = float(v.group(1)), float(v.group(4)), float(v.group(3))
    assert train_loss < np.log(24.0) - 0.2 and val_loss < np.log(24.0) - 0.2      # (chance loss = ln 24 = 3.178)
    assert val_r5 > 0.45 and val_r1 > 0.10, (val_r1, val_r5)                      # chance: 0.208 / 0.042
    assert rate > 1e5                                          # whole-epoch rate incl. batch synthesis (loose: boxes differ)
    base = os.path.join(str(tmp_path), "logs", "cx")
    info = torch.load(os.path.join(base, os.listdir(base)[0], "ckpt", "info.ckpt"))
    assert abs(info[-1]["recall"] - val_r5) < 1e-3


def test_cli_real_data_mode_on_reference_format_files(tmp_path, capsys):
    """the CLI's default (real-data) mode on files in the reference's on-disk formats -- pickles, feature
    tables, answer embedding -- kept resident on the device; the per-batch pipeline (index slices -> question encoder ->
    ncx_vqa_forward -> NeuralCX) must agree with the NeuralModel module fed the reference's dense batch."""
    import counterexamples as cli
    from neuralcx import formats
    from vqa.models.cx import NeuralModel
    paths = formats.write_synthetic_cx_files(os.path.join(str(tmp_path), "data"), n_train=192, n_val=96, n_img=300, seed=5)
    argv = ["--path_opt", os.path.join(PKG, "options", "cx", "neuralcx_256_1_all.yaml"), "-b", "64", "--epochs", "2", "-p", "2",
            "-t", "--untrained_vqa", "--project_dir", str(tmp_path), "--path_trainset", paths["path_trainset"],
            "--path_features", paths["path_features"]]
    cli.main(argv)
    out = capsys.readouterr().out
    assert "Epoch 2 train: loss:" in out and "Epoch 2 val: loss:" in out and "test:" in out
    base = os.path.join(str(tmp_path), "logs", "cx")
    base = os.path.join(base, os.listdir(base)[0])
    state = torch.load(os.path.join(base, "best", "model.ckpt"))
    assert state["linear_1.weight"].shape == (256, 14089) and any(k.startswith("vqa_model.") for k in state)
    assert os.path.exists(os.path.join(base, "final_results.txt"))
    # the answer embedding was initialised from answer_embedding.pickle (counterexamples.py:250-253): still close to it
    emb = formats.load_answer_embedding(os.path.join(paths["path_trainset"], "answer_embedding.pickle"))
    assert float((state["answer_embedding.weight"] - torch.from_numpy(emb)).abs().max()) < 1e-2

    # same checkpoint through the drop-in module with the reference's dense inputs (getDataFromBatch layout)
    args = cli.build_parser().parse_args(argv)
    r = cli.Runner(args, cli.load_options(args))
    r.load_real()
    r.engine.load_state({k: v for k, v in state.items() if not k.startswith("vqa_model.")})
    r.vqa.load_state_dict({k[len("vqa_model."):]: v for k, v in state.items() if k.startswith("vqa_model.")})
    r.mutan = type(r.mutan)(r.vqa)
    ids = list(range(40))
    assert r.val.vqa_cache is not None and "cached VQA outputs of the train split" in capsys.readouterr().out
    sel = torch.tensor(ids, device=DEV)
    b_cached, _ = r.get_batch(r.val, sel, ids[0])             # per-split cache of the frozen VQA model's outputs
    r.val.vqa_cache = None
    b, gt = r.get_batch(r.val, sel, ids[0])                    # produced per batch (--no_vqa_cache)
    for name in ("q_emb", "z_orig", "z_knns", "a_knns"):
        assert float((getattr(b, name) - getattr(b_cached, name)).abs().max()) <= 1e-5, name
    ev = r.engine.eval_step(b, gt)
    opt = cli.load_options(args)
    m = NeuralModel(model_spec=opt["cx_model"], dim_h=256, n_layers=1, emb=None, drop_p=0.25, vqa_model=r.vqa, knn_size=24,
                    trainable_vqa=False).cuda()
    m.load_state_dict(state)
    m.eval()
    img_idx, wids, aids, gt2 = r.val.batch_indices(torch.tensor(ids))
    scores = m(r.val.dense_features(img_idx), wids, aids.long())
    assert float((scores.detach() - ev["scores"]).abs().max()) <= 1e-4
    assert torch.equal(gt, gt2)


def test_cli_baseline_scorers(tmp_path, capsys):
    """RandomBaseline / DistanceBaseline / BlackBox (cx.py:20-44,114-136) evaluated with the on-device
    Recall@1/@5 kernel.  DistanceBaseline's Recall@k is exactly the fraction of triplets with knn_index < k."""
    import counterexamples as cli
    from neuralcx import formats
    from vqa.models.cx import BlackBox
    from oracle import ncx_oracle as orc
    paths = formats.write_synthetic_cx_files(os.path.join(str(tmp_path), "data"), n_train=64, n_val=160, n_img=200, seed=9)
    common = ["--path_opt", os.path.join(PKG, "options", "cx", "neuralcx_256_1_all.yaml"), "-b", "64", "-t", "--untrained_vqa",
              "--project_dir", str(tmp_path), "--path_trainset", paths["path_trainset"], "--path_features", paths["path_features"]]
    val = formats.load_cx_pickle(os.path.join(paths["path_trainset"], "pickle_old", "valset_augmented.pickle"))
    gt = np.array([e["comp"]["knn_index"] for e in val["examples_list"]])
    res = cli.main(common + ["--cx_model", "DistanceBaseline"])
    assert res["recall_5"] == pytest.approx((gt < 5).mean(), abs=1e-12) and res["recall_1"] == pytest.approx((gt < 1).mean(), abs=1e-12)
    res = cli.main(common + ["--cx_model", "RandomBaseline"])
    assert 0.05 < res["recall_5"] < 0.45                                       # 5/24 = 0.208 on 160 triplets
    res = cli.main(common + ["--cx_model", "BlackBox"])
    # the same scorer as an nn.Module on the reference's dense batch, ranked by the CPU oracle's recallAtK restatement
    args = cli.build_parser().parse_args(common + ["--cx_model", "BlackBox"])
    r = cli.Runner(args, cli.load_options(args))
    r.load_real()
    bb = BlackBox(r.vqa, 24)
    hits1 = hits5 = 0
    for i in range(0, r.test.N, 64):
        img_idx, wids, aids, g = r.test.batch_indices(torch.arange(i, min(i + 64, r.test.N)))
        s = bb(r.test.dense_features(img_idx), wids, aids).cpu()
        hits1 += int(orc.recall_at_k(s, g.cpu().long(), 1).sum()); hits5 += int(orc.recall_at_k(s, g.cpu().long(), 5).sum())
    assert res["recall_1"] == pytest.approx(hits1 / r.test.N, abs=1e-12) and res["recall_5"] == pytest.approx(hits5 / r.test.N, abs=1e-12)
    assert "BlackBox: 160 triplets" in capsys.readouterr().out


def _dp_gpu_worker(rank, world, port, q, bf16=False):
    import os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from conftest import PKG, ROOT
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from neuralcx import dp, ops
    from neuralcx.engine import NeuralCXEngine
    dp.init_distributed(backend="gloo")
    d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=64, L=2)
    Bg = 12
    batch = _dp_batch(d, Bg)
    params = orc.init_params(d, seed=4, gain=2.0)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.0, lr=1e-3, device=DEV,
                         world_size=world, bf16=bf16)
    eng.rank = rank
    eng.load_state(params)
    ids = dp.shard(list(range(Bg)), rank, world)
    sub = {k: v[ids] for k, v in batch.items()}
    b = ops.Batch.from_dense(*[sub[k].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    r = eng.train_step(b, sub["gt"].to(DEV).to(torch.int32), global_batch=Bg)
    eng.flush()                                                     # (the last gradient bucket's wait + Adam slice are deferred: engine.pipeline)
    torch.cuda.synchronize()
    loss = torch.tensor([float(r["loss"])], dtype=torch.float64)
    dist.all_reduce(loss)                                           # sum of the 1/B_global-scaled local losses
    if rank == 0:
        out = {k: v.cpu().numpy() for k, v in eng.grads.views.items()}     # all-reduced gradient of the global batch
        out["__loss__"] = float(loss)
        q.put(out)
    dist.barrier(); dist.destroy_process_group()


def _dp_batch(d, B):
    rng = np.random.default_rng(3)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    return dict(image_features=t(np.abs(rng.standard_normal((B, d.K + 1, d.dv))) * 0.45), q_emb=t(rng.standard_normal((B, d.dq)) * 0.3),
                z_orig=t(rng.standard_normal((B, d.dz))), z_knns=t(rng.standard_normal((B, d.K, d.dz))),
                a_knns=t(rng.standard_normal((B, d.K, d.A)) * 2), answer_aids=torch.from_numpy(rng.integers(0, d.A, size=B)),
                gt=torch.from_numpy(rng.integers(0, d.K, size=B)))


def test_eval_passes_reuse_gt_and_notice_weight_changes():
    """NCX_F_REUSE_GT: from the second batch of an evaluation pass on the engine skips Gt = W1[:, a_other] . E^T; the
    scores must stay bit-identical, and a weight update / a different batch size must invalidate the cached block."""
    from neuralcx import ops
    from neuralcx.engine import NeuralCXEngine
    d = orc.Dims(dv=64, dq=48, dz=16, A=20, H=32, L=1)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.0, lr=1e-2, device=DEV)
    eng.load_state(orc.init_params(d, seed=5, gain=2.0))
    mk = lambda B, seed: _dp_batch(d, B) if seed == 3 else None
    bt = _dp_batch(d, 12)
    to_batch = lambda t, sl: ops.Batch.from_dense(*[t[k][sl].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    gt = lambda t, sl: t["gt"][sl].to(DEV).to(torch.int32)
    a, b = slice(0, 6), slice(6, 12)
    s1 = eng.eval_step(to_batch(bt, a), gt(bt, a))["scores"].clone()
    s2 = eng.eval_step(to_batch(bt, b), gt(bt, b))["scores"].clone()          # reuses Gt
    s1b = eng.eval_step(to_batch(bt, a), gt(bt, a))["scores"].clone()         # reuses Gt
    assert torch.equal(s1, s1b)
    ref = orc.forward_faithful(orc.init_params(d, seed=5, gain=2.0), d, *[bt[k][b] for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    assert float((s2.cpu() - ref).abs().max()) <= 1e-4
    eng.train_step(to_batch(bt, a), gt(bt, a))                                 # weights move
    s3 = eng.eval_step(to_batch(bt, a), gt(bt, a))["scores"].clone()
    fresh = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.0, lr=1e-2, device=DEV)
    fresh.load_state(eng.state_dict())
    assert torch.equal(s3, fresh.eval_step(to_batch(bt, a), gt(bt, a))["scores"]) and not torch.equal(s3, s1)
    s4 = eng.eval_step(to_batch(bt, slice(0, 12)), gt(bt, slice(0, 12)))["scores"]   # other batch size: other layout
    assert torch.equal(s4, fresh.eval_step(to_batch(bt, slice(0, 12)), gt(bt, slice(0, 12)))["scores"])


@pytest.mark.parametrize("bf16", [False, True])
def test_dp2_hip_engine_equals_dp1(bf16):
    """DP-R == DP-1 at the same global batch.  Two gloo ranks share the card (RCCL needs one GPU per rank);
    each runs the HIP engine on its shard with loss_scale = 1/B_global, phased backward + async all-reduce, Adam.
    bf16: the configs[4] variant (its answer_embedding gradient comes from the bf16 image of the SUMMED dGt | dGgt block,
    which is also what a single process computes for the global batch)."""
    import torch.multiprocessing as mp
    from neuralcx import ops
    from neuralcx.engine import NeuralCXEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port + int(bf16), q, bf16)) for r in range(2)]
    [p.start() for p in procs]
    dp2 = q.get(timeout=240)
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=64, L=2)
    batch = _dp_batch(d, 12)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.0, lr=1e-3, device=DEV, bf16=bf16)
    eng.load_state(orc.init_params(d, seed=4, gain=2.0))
    b = ops.Batch.from_dense(*[batch[k].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    r = eng.train_step(b, batch["gt"].to(DEV).to(torch.int32))
    assert abs(dp2.pop("__loss__") - float(r["loss"])) <= 1e-5
    rel = 2e-3 if bf16 else 1e-5      # bf16: each rank rounds ITS rows' dpre to bf16 -- same values; dE sees bf16(sum) on both sides
    for k, v in eng.grads.views.items():
        ref = v.cpu().numpy()
        assert np.abs(dp2[k] - ref).max() <= grad_tol(k, ref, rel), k      # summation order only


@pytest.mark.parametrize("name", ["g1_small_L1", "g1_small_H20_L2"])
def test_hip_vqa_forward_vs_golden(name):
    """ncx_vqa_forward (gather + linear_v + tanh, folded R-term fusion, classifier) against the outputs the
    reference's own vqa_forward produced (tests/golden, G6)."""
    from helpers import load_golden
    from neuralcx import ops
    g, d, spec, params, batch = load_golden(name)
    R = int(g["vqa_R"])

    class _Lin:                      # minimal stand-ins so MutanWeights can stack the golden weights
        def __init__(self, w, b): self.weight, self.bias = w, b
    vp = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("vqa/")}

    class _F: pass
    f = _F()
    f.linear_v = _Lin(vp["fusion.linear_v.weight"], vp["fusion.linear_v.bias"])
    f.linear_q = _Lin(vp["fusion.linear_q.weight"], vp["fusion.linear_q.bias"])
    f.list_linear_hv = [_Lin(vp["fusion.list_linear_hv.%d.weight" % i], vp["fusion.list_linear_hv.%d.bias" % i]) for i in range(R)]
    f.list_linear_hq = [_Lin(vp["fusion.list_linear_hq.%d.weight" % i], vp["fusion.list_linear_hq.%d.bias" % i]) for i in range(R)]

    class _V: pass
    vqa = _V()
    vqa.fusion = f
    vqa.linear_classif = _Lin(vp["linear_classif.weight"], vp["linear_classif.bias"])
    vqa.opt = dict(fusion=dict(dim_hv=d.dz, dim_hq=d.dz, dim_mm=d.dz, R=R, activation_v="tanh", activation_q="tanh"), classif={})
    mw = ops.MutanWeights(vqa)
    feats = batch["image_features"].to(DEV)
    B, K1, dv = feats.shape
    idx = torch.arange(B * K1, device=DEV, dtype=torch.int32).view(B, K1)
    a_o, z_o, a_k, z_k = ops.vqa_forward(feats.reshape(B * K1, dv).contiguous(), idx, batch["q_emb"].to(DEV), mw, want_a_orig=True)
    for got, key in ((a_o, "a_orig"), (z_o, "z_orig"), (a_k, "a_knns"), (z_k, "z_knns")):
        assert np.abs(got.cpu().numpy() - g[key]).max() <= 1e-4 * max(1.0, np.abs(g[key]).max()), key


def test_hip_vqa_forward_full_dims_vs_torch_module():
    """Full widths (dv 2048, dq 2400, dim_mm 360, R 10, 2000 answers, B=64): the HIP producer against the plain PyTorch
    path of the same module on the GPU (hipBLASLt fp32), and NeuralModel scores through both."""
    import vqa.models as M
    from vqa.models.cx import NeuralModel
    torch.manual_seed(1)
    opt = dict(arch="MutanNoAtt", seq2vec=dict(arch="gru", emb_size=32, dropout=0.0),
               fusion=dict(dim_v=2048, dim_q=2400, dim_hv=360, dim_hq=360, dim_mm=360, R=10, dropout_v=0.5, dropout_q=0.5,
                           activation_v="tanh", activation_q="tanh", dropout_hv=0, dropout_hq=0), classif=dict(dropout=0.5))
    vqa = M.factory(opt, ["w%d" % i for i in range(50)], ["a%d" % i for i in range(2000)], cuda=True, data_parallel=False)
    spec = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True, q_emb=True, a_emb=True, z_emb=True)
    m = NeuralModel(model_spec=spec, dim_h=256, n_layers=1, emb=None, drop_p=0.25, vqa_model=vqa, knn_size=24, trainable_vqa=False).cuda().eval()
    B = 64
    feats = (torch.randn(B, 25, 2048, device=DEV).abs() * 0.45)
    wids = torch.randint(1, 51, (B, 26), device=DEV)
    aids = torch.randint(0, 2000, (B,), device=DEV)
    hip = m.vqa_forward(feats, wids)
    m.use_hip_vqa = False
    ref = m.vqa_forward(feats, wids)
    for h, r, nm in zip(hip, ref, ("a_orig", "z_orig", "a_knns", "z_knns", "q_emb")):
        assert h.shape == r.shape, nm
        assert float((h - r).abs().max()) <= 2e-4 * max(1.0, float(r.abs().max())), (nm, float((h - r).abs().max()))
    # ... and against the ORACLE at the same full widths: oracle.mutan_vqa_forward is the CPU restatement of vqa_forward below the
    # question encoder (cx.py:64-104, fusion.py:78-121, noatt.py:24-29), fed the module's own weights and the q_emb the
    # encoder produced (the encoder is an input producer, not part of the row): z and a within 1e-4 of their max
    vp = {k: v.detach().cpu() for k, v in vqa.state_dict().items()}
    o_ref = orc.mutan_vqa_forward(vp, feats.cpu(), hip[4].detach().cpu(), R=10)
    for h, r, nm in zip(hip[:4], o_ref, ("a_orig", "z_orig", "a_knns", "z_knns")):
        assert h.shape == r.shape, nm
        err, mx = float((h.detach().cpu() - r).abs().max()), float(r.abs().max())
        assert err <= 1e-4 * mx, (nm, err, mx)
    s_ref = m(feats, wids, aids)
    m.use_hip_vqa = True
    s_hip = m(feats, wids, aids)
    assert float((s_hip - s_ref).detach().abs().max()) <= 1e-4


@pytest.mark.parametrize("B,dz,dhv,R,A", [(13, 44, 28, 4, 52), (8, 360, 360, 10, 100), (70, 32, 36, 1, 40)])
def test_hip_vqa_forward_ragged_shapes_vs_oracle(B, dz, dhv, R, A):
    """The effective-weight fusion kernel (csrc/ncx_mutan.hip: one product per question against sum_r diag(hq_r[q]) Whv_r) at shapes that
    fill none of its tiles: batches that are not multiples of the 8 questions of a workgroup, dim_mm / dim_hv that are not multiples of the
    32-column / 32-deep tiles (dim_hv = 28, 36: a partial last k-step), R = 1 and R < 10; the classifier then runs on the fused forward
    kernel (A a multiple of 4) with zero-padded weight rows.  Reference: oracle.mutan_vqa_forward (fusion.py:78-121, noatt.py:24-29)."""
    import vqa.models as M
    from neuralcx import ops
    torch.manual_seed(7 + B)
    opt = dict(arch="MutanNoAtt", seq2vec=dict(arch="gru", emb_size=8, dropout=0.0),
               fusion=dict(dim_v=96, dim_q=48, dim_hv=dhv, dim_hq=20, dim_mm=dz, R=R, dropout_v=0.5, dropout_q=0.5,
                           activation_v="tanh", activation_q="tanh", dropout_hv=0, dropout_hq=0), classif=dict(dropout=0.5))
    vqa = M.factory(opt, ["w%d" % i for i in range(10)], ["a%d" % i for i in range(A)], cuda=True, data_parallel=False).eval()
    mw = ops.MutanWeights(vqa)
    K1 = 25
    feats = (torch.randn(B, K1, 96, device=DEV).abs() * 0.45)
    q_emb = torch.randn(B, 48, device=DEV) * 0.5
    idx = torch.arange(B * K1, device=DEV, dtype=torch.int32).view(B, K1)
    hip = ops.vqa_forward(feats.reshape(B * K1, 96).contiguous(), idx, q_emb, mw, want_a_orig=True)
    vp = {k: v.detach().cpu() for k, v in vqa.state_dict().items()}
    ref = orc.mutan_vqa_forward(vp, feats.cpu(), q_emb.cpu(), R=R)
    for h, r, nm in zip(hip, ref, ("a_orig", "z_orig", "a_knns", "z_knns")):
        assert h.shape == r.shape, nm
        err, mx = float((h.detach().cpu() - r).abs().max()), float(r.abs().max())
        assert err <= 1e-4 * max(mx, 1e-3), (nm, err, mx)


def test_module_is_reentrant_two_forwards_before_backward():
    """a second model(...) before loss.backward() -- an evaluation pass inside the train loop, exactly
    what counterexamples.py:357-361 does with eval_freq, or gradient accumulation -- must not disturb the first graph:
    backward(A) after {forward A, eval forward B, forward C} == backward(A) alone, and C's own backward is C's."""
    import vqa.models as M
    from vqa.models.cx import NeuralModel
    torch.manual_seed(1)
    A, B = 20, 6
    vqa = M.factory(_tiny_opt(), ["w%d" % i for i in range(30)], ["a%d" % i for i in range(A)], cuda=True, data_parallel=False)
    spec = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True, q_emb=True, a_emb=True, z_emb=True)
    m = NeuralModel(model_spec=spec, dim_h=16, n_layers=2, emb=None, drop_p=0.0, vqa_model=vqa, knn_size=24, trainable_vqa=False).cuda()
    own = {n: p for n, p in m.named_parameters() if not n.startswith("vqa_model.")}

    def inputs(seed, B):
        g = torch.Generator().manual_seed(seed)
        wids = torch.zeros(B, 26, dtype=torch.long)
        for b in range(B):
            wids[b, :3 + b] = torch.randint(1, 31, (3 + b,), generator=g)
        return ((torch.randn(B, 25, 64, generator=g).abs() * 0.45).to(DEV), wids.to(DEV), torch.randint(0, A, (B,), generator=g).to(DEV),
                torch.randint(0, 24, (B,), generator=g).to(DEV))

    def grads_of(*batches, interleave=None):
        m.zero_grad()
        losses = []
        for i, (f, w, a, gt) in enumerate(batches):
            losses.append(torch.nn.CrossEntropyLoss(reduction="sum")(m(f, w, a), gt) / f.shape[0])
            if interleave is not None and i == 0:
                with torch.no_grad():
                    m.eval(); interleave(); m.train()
        return losses

    ia, ib, ic = inputs(10, 6), inputs(11, 4), inputs(12, 5)
    m.train()
    (la,) = grads_of(ia); la.backward()
    ga = {n: p.grad.clone() for n, p in own.items()}
    (lc,) = grads_of(ic); lc.backward()
    gc = {n: p.grad.clone() for n, p in own.items()}
    la2, lc2 = grads_of(ia, ic, interleave=lambda: m(*ib[:3]))
    la2.backward()
    for n, p in own.items():
        assert torch.equal(p.grad, ga[n]), n
    m.zero_grad()
    lc2.backward()
    for n, p in own.items():
        assert torch.equal(p.grad, gc[n]), n
    # nn.Embedding's index error (cx.py:280) is kept, without a host sync per forward: the ids are clamped for the kernels and
    # the verdict is read at the next forward / by check_answer_ids()
    m.check_answer_ids()                                           # nothing pending from the valid calls above
    s_bad = m(ia[0], ia[1], torch.full_like(ia[2], A))
    assert torch.isfinite(s_bad).all()                             # (clamped ids: no out-of-bounds gather)
    with pytest.raises(IndexError):
        m.check_answer_ids()
    m(ia[0], ia[1], torch.full_like(ia[2], -1))
    # ... or surfaces at a later forward: at the very next one if the flag's copy has landed by then, else -- the host runs ahead of
    # the GPU -- at the one after; VALID forwards in between do not hide the verdict (the device flag is sticky until reported)
    raised = 0
    for i in range(3):
        try:
            m(ia[0], ia[1], ia[2])
        except IndexError:
            raised += 1
        torch.cuda.synchronize()
    assert raised == 1
    m(ia[0], ia[1], ia[2]); m.check_answer_ids()                   # and the flag clears once reported
    # a bad id in the LAST forward of an epoch / evaluation pass cannot be scored silently and then checkpointed: the deferred
    # verdict is collected by every mode switch and by state_dict() (the reference's loop does both: counterexamples.py:320,451,555)
    m(ia[0], ia[1], torch.full_like(ia[2], A))
    with pytest.raises(IndexError):
        m.eval()
    m(ia[0], ia[1], torch.full_like(ia[2], A))
    with pytest.raises(IndexError):
        m.state_dict()
    m.train(); m.state_dict()                                      # reported once, then clear
    # strict_ids: nn.Embedding's behaviour exactly -- the offending call raises, nothing is launched
    m.strict_ids = True
    with pytest.raises(IndexError):
        m(ia[0], ia[1], torch.full_like(ia[2], A))
    m.check_answer_ids(); m(ia[0], ia[1], ia[2])
    m.strict_ids = False


def test_cli_resume_continues_bit_for_bit(tmp_path, capsys):
    """--resume (counterexamples.py:563-580, App. B-2: info entries with `recall` only or with `recall_5`): one epoch,
    then `--resume <run>` for the second == an uninterrupted 2-epoch run (weights bit-identical: the checkpoint also
    carries Adam's moments and step, net-new), and a checkpoint written the reference's way (no optim.ckpt, info
    without recall_5) still resumes."""
    import counterexamples as cli
    common = ["--synthetic", "--path_opt", os.path.join(PKG, "options", "cx", "neuralcx_256_1_all.yaml"), "-b", "64",
              "--syn_train", "192", "--syn_val", "64", "--syn_images", "1024", "-p", "100"]
    d_full, d_res = os.path.join(str(tmp_path), "full"), os.path.join(str(tmp_path), "res")
    cli.main(common + ["--epochs", "2", "--project_dir", d_full])
    cli.main(common + ["--epochs", "1", "--project_dir", d_res])
    run = os.listdir(os.path.join(d_res, "logs", "cx"))[0]
    capsys.readouterr()
    cli.main(common + ["--epochs", "2", "--project_dir", d_res, "--resume", run])
    out = capsys.readouterr().out
    assert "Epoch 2 val: loss:" in out and "Epoch 1 val" not in out          # started at epoch 2
    run_full = os.listdir(os.path.join(d_full, "logs", "cx"))[0]
    s_full = torch.load(os.path.join(d_full, "logs", "cx", run_full, "ckpt", "model.ckpt"))
    s_res = torch.load(os.path.join(d_res, "logs", "cx", run, "ckpt", "model.ckpt"))
    for k in s_full:
        assert torch.equal(s_full[k], s_res[k]), k
    i_full = torch.load(os.path.join(d_full, "logs", "cx", run_full, "ckpt", "info.ckpt"))
    i_res = torch.load(os.path.join(d_res, "logs", "cx", run, "ckpt", "info.ckpt"))
    assert len(i_res) == 2 and i_res == i_full
    # a reference-style checkpoint: model + info only, info entries with `recall` but no `recall_5`
    base = os.path.join(d_res, "logs", "cx", run, "ckpt")
    os.remove(os.path.join(base, "optim.ckpt"))
    torch.save([{"loss": e["loss"], "recall": e["recall"]} for e in i_res], os.path.join(base, "info.ckpt"))
    cli.main(common + ["--epochs", "3", "--project_dir", d_res, "--resume", run])
    assert "Epoch 3 val: loss:" in capsys.readouterr().out
    assert len(torch.load(os.path.join(base, "info.ckpt"))) == 3


def _dp_pipeline_worker(rank, world, port, q, pipeline, H=64, Bg=12):
    import os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from conftest import PKG, ROOT
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from neuralcx import dp, ops
    from neuralcx.engine import NeuralCXEngine
    dp.init_distributed(backend="gloo")
    d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=H, L=1)
    steps = 5
    batch = _dp_batch(d, Bg * steps)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.25, lr=1e-3, device=DEV, world_size=world)
    eng.rank = rank
    eng.pipeline = pipeline
    eng.load_state(orc.init_params(d, seed=4, gain=2.0))
    losses, deferred = [], 0
    for s_ in range(steps):
        ids = [s_ * Bg + i for i in dp.shard(list(range(Bg)), rank, world)]
        sub = {k: v[ids] for k, v in batch.items()}
        b = ops.Batch.from_dense(*[sub[k].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
        r = eng.train_step(b, sub["gt"].to(DEV).to(torch.int32), global_batch=Bg)
        deferred += eng._pending is not None
        losses.append(float(r["loss"]))
        if s_ == 2:                                                 # an evaluation between two steps reads the FINAL weights of step 3
            losses.append(float(eng.eval_step(b, sub["gt"].to(DEV).to(torch.int32))["loss"]))
    out = {k: v.cpu().numpy() for k, v in eng.state_dict().items()}
    torch.cuda.synchronize()
    st = eng.optimizer_state()
    out["__exp_avg__"] = st["exp_avg"].numpy(); out["__exp_avg_sq__"] = st["exp_avg_sq"].numpy()
    out["__losses__"] = np.asarray(losses); out["__rank__"] = rank; out["__deferred__"] = deferred
    q.put(out)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("H,Bg", [(64, 12), (256, 256)])
def test_dp2_pipelined_bucket2_is_bit_identical_to_the_unpipelined_engine(H, Bg):
    """engine.pipeline (round 4): under data parallelism the wait for the LAST gradient bucket and the Adam slice it feeds are
    deferred to the next train_step, behind that step's data-only forward prelude (ncx_forward_phase PRELUDE: k_prep), so the
    exchange has the prelude to hide behind.  Same kernels, same operands, every dependency respected: after 5 steps with
    dropout on two ranks (gloo on one card) the weights, both Adam moments, every per-step loss and an evaluation taken between
    two steps are BIT-identical to the unpipelined engine's, and the replicas are identical.  (H = 256 with 128 triplets per rank: the
    phased backward then runs the balanced 8-wave TN launch in its two parts, csrc/ncx_dwtn.hip.)"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for pipeline in (True, False):
        q = ctx.Queue()
        port = 29400 + os.getpid() % 200 + int(pipeline) + (2 if H == 256 else 0)
        procs = [ctx.Process(target=_dp_pipeline_worker, args=(r, 2, port, q, pipeline, H, Bg)) for r in range(2)]
        [p.start() for p in procs]
        outs = [q.get(timeout=240), q.get(timeout=240)]
        [p.join(120) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        outs.sort(key=lambda o: o["__rank__"])
        for k in outs[0]:
            if not k.startswith("__") or k in ("__exp_avg__", "__exp_avg_sq__"):
                assert np.array_equal(outs[0][k], outs[1][k]), (pipeline, k)          # replicas identical
        assert outs[0]["__deferred__"] == (5 if pipeline else 0)                        # (the deferral really happened / really did not)
        res[pipeline] = outs
    for rank in (0, 1):
        a, b = res[True][rank], res[False][rank]
        for k in a:
            if k not in ("__rank__", "__deferred__"):
                assert np.array_equal(a[k], b[k]), (rank, k)


def _dp_idle_rank_worker(rank, world, port, q):
    import os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from conftest import PKG, ROOT
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from neuralcx import dp, ops
    from neuralcx.engine import NeuralCXEngine
    dp.init_distributed(backend="gloo")
    d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=64, L=1)
    batch = _dp_batch(d, 5)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.0, lr=1e-3, device=DEV, world_size=world)
    eng.rank = rank
    eng.load_state(orc.init_params(d, seed=4, gain=2.0))
    ids, plan = dp.epoch_plan(5, 4, 1, rank, world, "cpu", shuffle=False)      # batches of 4 and 1: rank 0 idles in the second
    for lo, hi, n_global, first, active in plan:
        sel = ids[lo:hi]
        sub = {k: v[sel] for k, v in batch.items()}
        b = ops.Batch.from_dense(*[sub[k].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
        r = eng.train_step(b, sub["gt"].to(DEV).to(torch.int32), global_batch=n_global, active=active)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in eng.state_dict().items()}
    out["__step__"] = eng.step_count; out["__rank__"] = rank; out["__loss__"] = float(r["loss"])
    q.put(out)
    dist.barrier(); dist.destroy_process_group()


def test_dp2_rank_without_triplets_still_steps_in_lockstep():
    """A global batch of ONE triplet on two ranks (the short last batch): the idle rank runs a zero-weight padding step,
    enters both all-reduces and Adam -- afterwards both replicas hold the SAME weights, equal to a single-process run."""
    import multiprocessing as mp
    from neuralcx import ops
    from neuralcx.engine import NeuralCXEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_dp_idle_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    outs = [q.get(timeout=240), q.get(timeout=240)]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    outs.sort(key=lambda o: o["__rank__"])
    assert outs[0]["__step__"] == outs[1]["__step__"] == 2
    assert outs[0]["__loss__"] == 0.0 and outs[1]["__loss__"] > 0.0     # rank 0's slice of the 1-triplet batch is empty: it reports no loss
    d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=64, L=1)
    batch = _dp_batch(d, 5)
    eng = NeuralCXEngine(K=d.K, dv=d.dv, dq=d.dq, dz=d.dz, da=d.da, A=d.A, H=d.H, L=d.L, drop_p=0.0, lr=1e-3, device=DEV)
    eng.load_state(orc.init_params(d, seed=4, gain=2.0))
    for sl in (slice(0, 4), slice(4, 5)):
        b = ops.Batch.from_dense(*[batch[k][sl].to(DEV) for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
        eng.train_step(b, batch["gt"][sl].to(DEV).to(torch.int32))
    ref = {k: v.cpu().numpy() for k, v in eng.state_dict().items()}
    for k, v in ref.items():
        assert np.array_equal(outs[0][k], outs[1][k]), k           # replicas identical
        # against the single-process run: Adam normalises by sqrt(v), so entries whose gradient is round-off noise (the
        # columns of linear_1 that meet all-zero inputs, out.bias) move by up to +-lr per step either way; everything
        # else agrees to summation order
        diff = np.abs(outs[0][k] - v)
        assert diff.max() <= 2 * 1e-3 + 1e-6, k                    # never more than two steps of lr apart
        if k != "out.bias":
            assert (diff > 2e-6).mean() <= 0.02, (k, float((diff > 2e-6).mean()))


def test_bench_two_ranks_on_one_card_gloo_rehearsal():
    """`python bench.py --gpus 2` end to end on the GPU box: the launcher starts two ranks (torch.distributed.run), each builds its
    engine on the card, the data-parallel step runs its phased backward with both gradient exchanges (5 | 2 | 4: the dGt block, then
    the flat buffer without answer_embedding), rank 0 prints ONE JSON line with the multi-rank fields.  One card cannot host two
    RCCL ranks, so the collective backend is gloo here (NCX_DIST_BACKEND); with RCCL the same command fails with a clear message
    (tests/test_bench_cpu.py).  The RCCL run itself needs the driver's multi-GPU node."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NCX_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64",
                        "--n_img", "4096", "--preheat-ms", "0", "--heldout", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4 and d["warmup"] == 2
    assert d["config"]["global_batch"] == 128 and d["config"]["per_rank_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["rccl"]["nranks"] == 2 and "gloo" in d["rccl"]["note"]
    assert d["rccl"]["exposed_wait"]["bucket1_ms"] >= 0 and d["rccl"]["exposed_wait"]["bucket2_ms"] >= 0      # (stall of the compute stream per bucket)
    assert d["value"] > 0 and abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert np.isfinite(d["config"]["final_loss"]) and 2.0 < d["config"]["final_loss"] < 4.0
