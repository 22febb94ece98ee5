"""CPU: the oracle (our restatement) against every golden vector produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import ncx_oracle as orc
from helpers import GOLDEN, check_grads_against_golden, golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_forward_loss_recall_grads(name):
    g, d, spec, params, batch = load_golden(name)
    scores, loss, grads = orc.loss_and_grads(params, d, batch, spec=spec)
    # same operator sequence on the same torch build -> essentially exact
    assert np.abs(scores.numpy() - g["scores"]).max() <= 2e-6
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    for k in (1, 5):
        assert (orc.recall_at_k(scores, batch["gt"], k) == g["recall%d" % k]).all()
        assert ((orc.rank_of_gt(scores.numpy(), g["gt"]) < k).astype(np.int32) == g["recall%d" % k]).all()
    check_grads_against_golden(g, {k: v.numpy() for k, v in grads.items()}, rel=1e-5)


def test_adam_step_matches_reference_optimizer():
    g, d, spec, params, batch = load_golden("g1_small_L1")
    st = orc.AdamState()
    new, _, _, _ = orc.train_step(params, d, batch, st, lr=1e-4, spec=spec)
    for k, v in new.items():
        ref = g["adam1/" + k]
        assert np.abs(v.numpy() - ref).max() <= 1e-7 + 1e-6 * np.abs(ref).max(), k


def test_adam_restatement_vs_torch_optim_multi_step():
    torch.manual_seed(0)
    p0 = {"w": torch.randn(37, 11), "b": torch.randn(11)}
    tp = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = torch.optim.Adam(list(tp.values()), lr=1e-3)
    st = orc.AdamState()
    cur = {k: v.clone() for k, v in p0.items()}
    for step in range(5):
        grads = {k: torch.randn_like(v) * (0.1 + step) for k, v in p0.items()}
        for k in tp:
            tp[k].grad = grads[k].clone()
        opt.step()
        cur = orc.adam_update(cur, grads, st, lr=1e-3)
        for k in tp:
            assert torch.allclose(cur[k], tp[k].detach(), rtol=1e-6, atol=1e-7)


def test_recall_and_loss_known_answers():
    g = dict(np.load(GOLDEN + "/g4_recall_loss.npz"))
    for pre in ("dist", "rand"):
        s, gt = torch.from_numpy(g[pre + "_scores"]), torch.from_numpy(g[pre + "_gt"])
        for k in (1, 5):
            assert (orc.recall_at_k(s, gt, k) == g["%s_recall%d" % (pre, k)]).all()
        if pre == "rand":   # no ties -> deterministic rank rule == topk membership
            for k in (1, 5):
                assert ((orc.rank_of_gt(s.numpy(), gt.numpy()) < k) == g["rand_recall%d" % k]).all()
    # DistanceBaseline known answer: recall@k == (knn_index < k)   (vqa/models/cx.py:33-44)
    assert (g["dist_recall5"] == (g["dist_gt"] < 5)).all()
    s, gt = torch.from_numpy(g["rand_scores"]), torch.from_numpy(g["rand_gt"])
    assert abs(float(orc.ranking_loss(s, gt)) - float(g["rand_loss"])) < 1e-6


def test_dropout_mask_generator_statistics_and_determinism():
    m1 = orc.dropout_keep_mask(1234, 1, 24 * 64, 256, 0.25)
    m2 = orc.dropout_keep_mask(1234, 1, 24 * 64, 256, 0.25)
    m3 = orc.dropout_keep_mask(1234, 2, 24 * 64, 256, 0.25)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    assert abs(float(m1.mean()) - 0.75) < 5e-3
    assert float(orc.dropout_keep_mask(7, 1, 100, 16, 0.0).min()) == 1.0


def test_train_mode_masks_change_output_consistently():
    g, d, spec, params, batch = load_golden("g1_small_L2")
    B = batch["gt"].shape[0]
    masks = [orc.dropout_keep_mask(99, l, B * d.K, d.H, 0.25) for l in (1, 2)]
    s1, l1, _ = orc.loss_and_grads(params, d, batch, spec=spec, drop_p=0.25, keep_masks=masks)
    ones = [torch.ones(B * d.K, d.H)] * 2
    s0, l0, _ = orc.loss_and_grads(params, d, batch, spec=spec, drop_p=0.0, keep_masks=ones)
    assert np.abs(s0.numpy() - g["scores"]).max() <= 2e-6
    assert np.abs(s1.numpy() - g["scores"]).max() > 1e-4


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("g1_") or n.startswith("g3_")])
def test_mutan_vqa_forward_restatement(name):
    """G6: the oracle's MUTAN producer against what the reference's vqa_forward handed to the MLP."""
    g, d, spec, params, batch = load_golden(name)
    vp = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("vqa/")}
    a_o, z_o, a_k, z_k = orc.mutan_vqa_forward(vp, batch["image_features"], batch["q_emb"], int(g["vqa_R"]))
    for got, key in ((a_o, "a_orig"), (z_o, "z_orig"), (a_k, "a_knns"), (z_k, "z_knns")):
        assert np.abs(got.numpy() - g[key]).max() <= 2e-6 * max(1.0, np.abs(g[key]).max()), key


def test_faithful_cpu_model_with_shared_dropout_masks_equals_the_functional_oracle():
    """bench.py's CPU baseline trains oracle.FaithfulCPUModel with the counter-based dropout masks the HIP kernels use
    (keep_masks) so that both sides see identical training; with the same masks the module must reproduce the functional
    restatement (forward_faithful) exactly, and without masks it falls back to torch's own Dropout."""
    d = orc.Dims(dv=32, dq=24, dz=8, A=20, H=16, L=2)
    B = 5
    rng = np.random.default_rng(0)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    x = dict(image_features=t(np.abs(rng.standard_normal((B, d.K + 1, d.dv)))), q_emb=t(rng.standard_normal((B, d.dq))),
             z_orig=t(rng.standard_normal((B, d.dz))), z_knns=t(rng.standard_normal((B, d.K, d.dz))),
             a_knns=t(rng.standard_normal((B, d.K, d.A))), answer_aids=torch.from_numpy(rng.integers(0, d.A, size=B)))
    m = orc.FaithfulCPUModel(d, drop_p=0.25, seed=3)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    masks = [orc.dropout_keep_mask(0x1234567, l, B * d.K, d.H, 0.25) for l in (1, 2)]
    m.train(); m.keep_masks = masks
    s_mod = m(x["image_features"], x["q_emb"], x["z_orig"], x["z_knns"], x["a_knns"], x["answer_aids"])
    s_fun = orc.forward_faithful(params, d, x["image_features"], x["q_emb"], x["z_orig"], x["z_knns"], x["a_knns"], x["answer_aids"],
                                 drop_p=0.25, keep_masks=masks)
    assert torch.equal(s_mod.detach(), s_fun)
    m.eval()                                                     # evaluation ignores the masks (dropout is the identity)
    s_eval = m(x["image_features"], x["q_emb"], x["z_orig"], x["z_knns"], x["a_knns"], x["answer_aids"])
    s_ref = orc.forward_faithful(params, d, x["image_features"], x["q_emb"], x["z_orig"], x["z_knns"], x["a_knns"], x["answer_aids"])
    assert torch.equal(s_eval.detach(), s_ref)
