"""GPU: the HIP path (through the C ABI) against the golden vectors of the reference and the CPU oracle.

Tolerances (BASELINE.json north_star / SURVEY 8c): logits <= 1e-4 abs (fp32), loss <= 1e-5,
gradients <= 1e-4 of the tensor's max |grad|, Recall@1/@5 vectors exact.
"""
import numpy as np
import pytest
import torch

from oracle import ncx_oracle as orc
from helpers import GOLDEN, check_grads_against_golden, grad_tol, golden_names, load_golden

pytestmark = pytest.mark.gpu

FIELD = {"answer_embedding.weight": "answer_embedding", "linear_1.weight": "w1", "linear_1.bias": "b1",
         "linear_2.weight": "w2", "linear_2.bias": "b2", "linear_3.weight": "w3", "linear_3.bias": "b3",
         "out.weight": "w_out", "out.bias": "b_out"}


def dev():
    return torch.device("cuda:0")


def to_dev_params(params):
    return {FIELD[k]: v.to(dev()).contiguous() for k, v in params.items()}


def to_dev_batch(batch, spec=None, keep_mask=None, extra=None):
    from neuralcx.ops import Batch
    extra = extra or {}
    g = lambda k: batch[k].to(dev())
    return Batch.from_dense(g("image_features"), g("q_emb"), g("z_orig"), g("z_knns"), g("a_knns"),
                            g("answer_aids"), keep_mask=None if keep_mask is None else keep_mask.to(dev()),
                            **{k: v.to(dev()) for k, v in extra.items()})


def run_hip(d, spec, params, batch, training=False, drop_p=0.0, keep_mask=None, seed=0, extra=None):
    from neuralcx import ops
    b = to_dev_batch(batch, spec, keep_mask, extra)
    p = to_dev_params(params)
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A, flags=ops.flags_from_spec(spec), training=training,
                         drop_p=drop_p, seed=seed)
    ws = ops.alloc_workspace(dims, dev())
    scores = ops.forward(dims, b, p, ws)
    gt = batch["gt"].to(dev()).to(torch.int32)
    lr = ops.ranking_loss(scores, gt)
    grads = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    ops.backward(dims, b, p, ws, lr["dscores"], grads)
    torch.cuda.synchronize()
    inv = {v: k for k, v in FIELD.items()}
    return (scores.cpu(), lr, {inv[k]: v.cpu().numpy() for k, v in grads.items()})


@pytest.mark.parametrize("name", golden_names())
def test_golden_forward_loss_recall_backward(name):
    g, d, spec, params, batch = load_golden(name)
    scores, lr, grads = run_hip(d, spec, params, batch)
    assert np.abs(scores.numpy() - g["scores"]).max() <= 1e-4
    assert abs(float(lr["loss"].cpu()) - float(g["loss"])) <= 1e-5
    rank = lr["rank"].cpu().numpy()
    for k in (1, 5):
        assert ((rank < k).astype(np.int32) == g["recall%d" % k]).all()
    hits = lr["hits"].cpu().numpy()
    assert hits[0] == g["recall1"].sum() and hits[1] == g["recall5"].sum()
    for n, v in grads.items():
        assert np.isfinite(v).all(), n
    check_grads_against_golden(g, grads, rel=1e-4)


def random_case(seed, B, d, scale=0.45):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    batch = dict(image_features=t(np.abs(rng.standard_normal((B, d.K + 1, d.dv))) * scale),
                 q_emb=t(rng.standard_normal((B, d.dq)) * 0.3), z_orig=t(rng.standard_normal((B, d.dz))),
                 z_knns=t(rng.standard_normal((B, d.K, d.dz))), a_knns=t(rng.standard_normal((B, d.K, d.A)) * 2),
                 answer_aids=torch.from_numpy(rng.integers(0, d.A, size=B)), gt=torch.from_numpy(rng.integers(0, d.K, size=B)))
    return batch


def compare_with_oracle(d, spec, params, batch, training=False, drop_p=0.0, masks=None, seed=0, extra=None, use_rng=False):
    keep = None if masks is None or use_rng else torch.stack(masks)
    scores, lr, grads = run_hip(d, spec, params, batch, training=training, drop_p=drop_p, keep_mask=keep, seed=seed, extra=extra)
    ob = dict(batch)
    if extra:
        ob.update(extra)
    s_ref, l_ref, g_ref = orc.loss_and_grads(params, d, ob, spec=spec, drop_p=drop_p, keep_masks=masks)
    assert np.abs(scores.numpy() - s_ref.numpy()).max() <= 1e-4
    assert abs(float(lr["loss"].cpu()) - float(l_ref)) <= 1e-5
    sr, gtn = s_ref.numpy(), batch["gt"].numpy()
    gap = np.abs(sr - sr[np.arange(len(gtn)), gtn][:, None]); gap[np.arange(len(gtn)), gtn] = np.inf
    safe = gap.min(1) > 2e-4                                  # rows without a near-tie around the ground truth
    assert (lr["rank"].cpu().numpy()[safe] == orc.rank_of_gt(sr, gtn)[safe]).all()
    for k, ref in g_ref.items():
        ref = ref.numpy()
        tol = grad_tol(k, ref, 1e-4)
        err = np.abs(grads[k].reshape(ref.shape) - ref).max()
        assert err <= tol, (k, err, tol)
    return scores, lr, grads


@pytest.mark.parametrize("B,K,H,L", [(1, 24, 16, 1), (7, 24, 20, 2), (13, 24, 48, 3), (5, 48, 32, 1), (33, 24, 96, 2)])
def test_ragged_shapes_vs_oracle(B, K, H, L):
    """Odd batch sizes / widths (nothing a multiple of the 16/32/64 tile sizes), and K = 48 (config 5)."""
    d = orc.Dims(K=K, dv=70, dq=50, dz=18, A=45, H=H, L=L)
    params = orc.init_params(d, seed=11 + B, gain=3.0)
    batch = random_case(100 + B, B, d)
    batch["answer_aids"][0] = batch["answer_aids"][B - 1]
    compare_with_oracle(d, None, params, batch)


@pytest.mark.parametrize("S", [2, 3, 4, 8, 16])
def test_forced_split_k_layouts_vs_oracle(S, monkeypatch):
    """Aligned split-K with every workgroup layout of WgMap (chunk-per-XCD for S | 8 and 8 | S, incl. padding ids;
    the interleaved layout otherwise) on the weight-gradient GEMMs, forced through the planner's experiment hooks."""
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    for gid in (0, 1, 4, 6, 7, 9):                       # Gt, Sh, grouped dW1, dE, dW1ak, dW_l
        monkeypatch.setenv("NCX_SPLIT_%d" % gid, str(S))
    d = orc.Dims(dv=70, dq=50, dz=18, A=45, H=128, L=2)
    params = orc.init_params(d, seed=3, gain=3.0)
    batch = random_case(900 + S, 44, d)                   # M = 1056 rows: 33 k-steps of the row-reduction GEMMs
    compare_with_oracle(d, None, params, batch)


def run_hip_bf16(d, params, batch, training=False, drop_p=0.0, keep_mask=None, seed=0):
    from neuralcx import ops, _lib
    b = to_dev_batch(batch, None, keep_mask)
    p = to_dev_params(params)
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A, flags=_lib.NCX_F_ALL | _lib.NCX_F_BF16, training=training,
                         drop_p=drop_p, seed=seed)
    ws = ops.alloc_workspace(dims, dev())
    scores = ops.forward(dims, b, p, ws)
    lr = ops.ranking_loss(scores, batch["gt"].to(dev()).to(torch.int32))
    grads = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    ops.backward(dims, b, p, ws, lr["dscores"], grads)
    torch.cuda.synchronize()
    inv = {v: k for k, v in FIELD.items()}
    return scores.cpu(), lr, {inv[k]: v.cpu().numpy() for k, v in grads.items()}


@pytest.mark.parametrize("B,K,H,L,dv,strict", [(3, 24, 16, 1, 70, True), (9, 24, 128, 2, 64, True), (40, 48, 200, 1, 96, True),
                                                (16, 24, 256, 3, 128, True), (24, 24, 256, 3, 128, False), (64, 24, 256, 3, 128, False)])
def test_bf16_variant_vs_bf16_oracle(B, K, H, L, dv, strict):
    """BASELINE configs[4] (bf16 operands on the two dominant GEMMs, fp32 accumulate; K = 48 included): against the
    oracle's restatement with the same operands rounded to bf16.  Products of bf16 values are exact in fp32, so what
    remains is summation order -- logits <= 2e-3, loss <= 2e-4, gradients <= 1e-3 of the tensor's max on the strict
    cases -- plus two discrete events the larger cases do hit: an operand computed on the device (the distance
    column: ~8 +- 1 fp32 ulp) landing on the other side of a bf16 rounding boundary (a 0.06 step), and a ReLU input
    within that noise of zero switching its unit.  Those cases bound the logits by 1e-2 and every gradient tensor's
    Frobenius error by 5 %.  The variant itself stays within bf16 distance of the fp32 net."""
    d = orc.Dims(K=K, dv=dv, dq=50, dz=18, A=45, H=H, L=L)
    params = orc.init_params(d, seed=17 + B, gain=3.0)
    batch = random_case(300 + B, B, d)
    seed = 0xABCDEF12345
    masks = [orc.dropout_keep_mask(seed, l, B * d.K, d.H, 0.25) for l in range(1, L + 1)]
    for training in (False, True):
        scores, lr, grads = run_hip_bf16(d, params, batch, training=training, drop_p=0.25 if training else 0.0, seed=seed)
        s_ref, l_ref, g_ref = orc.loss_and_grads_bf16(params, d, batch, drop_p=0.25 if training else 0.0,
                                                      keep_masks=masks if training else None)
        assert np.abs(scores.numpy() - s_ref.numpy()).max() <= (2e-3 if strict else 1e-2)
        assert abs(float(lr["loss"].cpu()) - float(l_ref)) <= (2e-4 if strict else 1e-3)
        for k, ref in g_ref.items():
            ref = ref.numpy()
            tol = grad_tol(k, ref, 1e-3)
            err = np.abs(grads[k].reshape(ref.shape) - ref)
            assert np.isfinite(grads[k]).all(), k
            if strict:
                assert err.max() <= tol, (k, err.max(), tol)
            elif k != "out.bias":
                assert np.linalg.norm(err) <= 5e-2 * np.linalg.norm(ref), (k, np.linalg.norm(err), np.linalg.norm(ref))
    s32, _, _ = orc.loss_and_grads(params, d, batch)
    s16, _, _ = orc.loss_and_grads_bf16(params, d, batch)
    assert 1e-4 < float((s16 - s32).abs().max()) <= 0.3          # eval mode: bf16 operands vs the fp32 network


@pytest.mark.parametrize("B,K,H", [(9, 24, 256), (37, 24, 200), (13, 48, 256)])
def test_bf16_forward_product_on_192x256_tiles_ragged(B, K, H, monkeypatch):
    """The LDS-DMA staged 192 x 256 forward kernel of the bf16 variant (csrc/ncx_bf16.hip, gemm_bf16_nt8_kernel; the planner takes it
    where its tiles fill the chip, i.e. at configs[4]'s size -- covered by test_configs4_shape...) forced onto small RAGGED shapes:
    M = B K is not a multiple of 192 (clamped DMA rows, guarded stores), H below the 256-column tile, one and several workgroups.
    Logits against the bf16 restatement, and bit-identical to the 128 x 128 kernel's (same products, same k order per element)."""
    d = orc.Dims(K=K, dv=64, dq=50, dz=18, A=45, H=H, L=1)
    params = orc.init_params(d, seed=5 + B, gain=3.0)
    batch = random_case(900 + B, B, d)
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_BF16_NT_CFG", "8")
    monkeypatch.setenv("NCX_BF16_TN8", "1")                        # ... and the 256 x 128 weight-gradient kernel (gemm_bf16_tn8_kernel)
    s8, lr8, g8 = run_hip_bf16(d, params, batch)
    monkeypatch.setenv("NCX_BF16_NT_CFG", "0")
    monkeypatch.delenv("NCX_BF16_TN8")
    monkeypatch.setenv("NCX_BF16_NO_TN8", "1")
    s0, lr0, g0 = run_hip_bf16(d, params, batch)
    s_ref, l_ref, g_ref = orc.loss_and_grads_bf16(params, d, batch)
    assert np.abs(s8.numpy() - s_ref.numpy()).max() <= 2e-3
    assert np.isfinite(s8.numpy()).all()
    assert np.abs(s8.numpy() - s0.numpy()).max() <= 1e-5          # (another k-step grouping of the fp32 accumulation)
    for k, ref in g_ref.items():
        assert np.abs(g8[k].reshape(ref.shape) - ref.numpy()).max() <= grad_tol(k, ref.numpy(), 1e-3), k
        assert np.abs(g8[k] - g0[k]).max() <= grad_tol(k, ref.numpy(), 1e-5), k           # (the two TN kernels: other k-chunk boundaries only)


def test_rows_wider_than_the_register_resident_prep_path():
    """k_prep keeps rows of up to 2048 floats in registers (single pass); wider feature / answer rows take the re-reading
    path.  Both must agree with the oracle (fp32 and the bf16 variant's row pack)."""
    d = orc.Dims(dv=2100, dq=40, dz=12, A=2070, H=16, L=1)
    params = orc.init_params(d, seed=2, gain=3.0)
    batch = random_case(41, 3, d)
    compare_with_oracle(d, None, params, batch)
    scores, lr, grads = run_hip_bf16(d, params, batch)
    s_ref, l_ref, g_ref = orc.loss_and_grads_bf16(params, d, batch)
    assert np.abs(scores.numpy() - s_ref.numpy()).max() <= 5e-3
    for k, ref in g_ref.items():
        ref = ref.numpy()
        assert np.abs(grads[k].reshape(ref.shape) - ref).max() <= grad_tol(k, ref, 2e-3), k


@pytest.mark.parametrize("B,K,H,L,dv", [(1, 24, 16, 1, 70), (7, 24, 20, 2, 64), (13, 48, 48, 1, 130), (40, 24, 200, 1, 96), (9, 48, 132, 3, 2048)])
def test_fused_v_gradient_kernel_vs_oracle(B, K, H, L, dv, monkeypatch):
    """k_dw_km (d linear_1.weight[:, v_other] and [:, v_mult] in one MFMA pass with the per-triplet fold) is used from 256
    triplets on; forced here on small / ragged shapes (edge tiles in both directions, K = 24 and 48, chunks with a single
    triplet, empty chunks) and compared with the oracle like every other path."""
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_KM_FORCE", "1")
    d = orc.Dims(K=K, dv=dv, dq=50, dz=18, A=45, H=H, L=L)
    params = orc.init_params(d, seed=23 + B, gain=3.0)
    batch = random_case(700 + B, B, d)
    compare_with_oracle(d, None, params, batch)


def test_train_mode_explicit_masks_and_generator():
    d = orc.Dims(dv=64, dq=48, dz=16, A=20, H=32, L=3)
    params = orc.init_params(d, seed=5, gain=3.0)
    batch = random_case(77, 9, d)
    seed = 0x1234567890ABCDEF
    masks = [orc.dropout_keep_mask(seed, l, 9 * d.K, d.H, 0.25) for l in (1, 2, 3)]
    # (a) explicit masks handed to both sides
    compare_with_oracle(d, None, params, batch, training=True, drop_p=0.25, masks=masks)
    # (b) the kernel's own counter-based generator must reproduce the oracle's restatement of it
    compare_with_oracle(d, None, params, batch, training=True, drop_p=0.25, masks=masks, seed=seed, use_rng=True)


@pytest.mark.parametrize("spec", [dict(v_mult=False), dict(v_dist=False), dict(v_rank=False), dict(a_emb=False),
                                  dict(v_mult=False, v_dist=False, v_rank=False, a_emb=False)])
def test_lesion_flags_vs_oracle(spec):
    d = orc.Dims(dv=64, dq=48, dz=16, A=20, H=32, L=2)
    B = 6
    params = orc.init_params(d, seed=8, gain=3.0)
    batch = random_case(55, B, d)
    extra = {}
    torch.manual_seed(3)
    if not spec.get("v_rank", True):
        extra["v_rank"] = torch.rand(B, d.K, d.K)
    if not spec.get("a_emb", True):
        batch["a_knns"] = torch.rand(B, d.K, d.da)        # cx.py:285 noise block
        extra["a_emb_gt"] = torch.rand(B, d.da)            # cx.py:284
    full = dict(orc.DEFAULT_SPEC, **spec)
    compare_with_oracle(d, full, params, batch, extra=extra)


@pytest.mark.parametrize("case", ["all", "no_a_emb", "no_v_mult", "H512_L2", "K48", "dropout"])
def test_balanced_tn_launch_paths_vs_oracle(case):
    """The 8-wave balanced TN launch (csrc/ncx_dwtn.hip) takes dGt and every weight-gradient column block outside the per-triplet
    fold whenever H is a multiple of 256 and the batch a multiple of 32 (>= 128): reduced widths keep the oracle fast, H and B
    are the real ones.  Cases: the full model (aligned dGt part + rest sequence + the dW1[:, a_other] launch); the a_emb lesion (no
    aligned part, the a_other block joins the rest sequence as a plain operand); the v_mult lesion (no fold kernel: the v columns
    stay on the generic grouped launch next to the TN launch); H = 512 (two row tiles per column tile) with a hidden layer;
    K = 48; train mode with the counter-based dropout.  B = 160 is not a multiple of the 8 row chunks' k-steps: ragged chunking."""
    spec, extra, kw = None, {}, {}
    d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=256, L=1)
    B = 160
    if case == "H512_L2":
        d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=512, L=2); B = 128
    if case == "K48":
        d = orc.Dims(K=48, dv=96, dq=64, dz=24, A=40, H=256, L=1); B = 128
    params = orc.init_params(d, seed=21, gain=3.0)
    batch = random_case(2100 + len(case), B, d)
    batch["answer_aids"][0] = batch["answer_aids"][B - 1]           # a duplicated answer id (owner-computes scatter)
    if case == "no_a_emb":
        torch.manual_seed(4)
        batch["a_knns"] = torch.rand(B, d.K, d.da); extra["a_emb_gt"] = torch.rand(B, d.da)
        spec = dict(orc.DEFAULT_SPEC, a_emb=False)
    if case == "no_v_mult":
        spec = dict(orc.DEFAULT_SPEC, v_mult=False)
    if case == "dropout":
        seed = 0x0BADC0FFEE
        masks = [orc.dropout_keep_mask(seed, 1, B * d.K, d.H, 0.25)]
        kw = dict(training=True, drop_p=0.25, masks=masks, seed=seed, use_rng=True)
    from neuralcx import _lib
    from neuralcx import ops
    compare_with_oracle(d, spec, params, batch, extra=extra, **kw)


def test_full_dims_property_checks():
    """BASELINE size (B=512, K=24, full widths, H=256): size-independent properties instead of the oracle.
    (1) row permutation equivariance of scores; (2) shifting all K logits of a row by a constant (out.bias)
    leaves loss/rank/dscores unchanged; (3) dscores rows sum to 0; (4) loss == mean of per-row losses."""
    from neuralcx import ops
    d = orc.Dims()
    B = 512
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n_img = 4096
    feats = (torch.randn(n_img, d.dv).abs() * 0.45).to(dev())
    idx = torch.from_numpy(rng.integers(0, n_img, size=(B, d.K + 1)).astype(np.int32)).to(dev())
    mk = lambda *s: torch.randn(*s, device=dev())
    b = ops.Batch(feats, idx, mk(B, d.dq) * 0.3, mk(B, d.dz), mk(B, d.K, d.dz), mk(B, d.K, d.A) * 2,
                  torch.from_numpy(rng.integers(0, d.A, size=B).astype(np.int32)).to(dev()))
    p = to_dev_params(orc.init_params(d, seed=42))
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    ws = ops.alloc_workspace(dims, dev())
    s1 = ops.forward(dims, b, p, ws).clone()
    perm = torch.randperm(B, device=dev())
    b2 = ops.Batch(feats, idx[perm].contiguous(), b.q_emb[perm].contiguous(), b.z_orig[perm].contiguous(),
                   b.z_knns[perm].contiguous(), b.a_knns[perm].contiguous(), b.answer_aids[perm].contiguous())
    s2 = ops.forward(dims, b2, p, ws)
    assert torch.equal(s1[perm], s2)                       # same tiles, same order of operations per row
    gt = torch.from_numpy(rng.integers(0, d.K, size=B).astype(np.int32)).to(dev())
    r1 = ops.ranking_loss(s1, gt)
    r2 = ops.ranking_loss(s1 + 3.25, gt)
    assert torch.equal(r1["rank"], r2["rank"])
    assert float((r1["loss"] - r2["loss"]).abs()) < 1e-5
    assert float(r1["dscores"].sum(1).abs().max()) < 1e-6
    assert abs(float(r1["loss"]) - float(r1["loss_rows"].sum())) < 1e-5
    assert int(r1["hits"][1]) == int((r1["rank"] < 5).sum()) and int(r1["hits"][0]) == int((r1["rank"] < 1).sum())
    # determinism: a second identical forward is bit-identical
    assert torch.equal(ops.forward(dims, b, p, ws), s1)


def _full_size_case(d, B, seed, bf16=False):
    from helpers import condition_away_from_kinks, random_case_f32
    params = orc.init_params(d, seed=42)
    batch = random_case_f32(seed, B, d)
    batch["answer_aids"][1] = batch["answer_aids"][0]            # a duplicated answer id (owner-computes scatter)
    redrawn = condition_away_from_kinks(params, d, batch, seed, bf16=bf16)
    assert redrawn < 4 * B
    return params, batch


def test_configs1_full_size_every_logit_and_gradient_vs_oracle():
    """BASELINE configs[1] at its stated size (B = 512, K = 24, dv = 2048, H = 256, L = 1: the real launch plan with every
    tile shape, k-split and the fused v-column kernel): ALL 12 288 logits, the loss, the ranks and EVERY gradient element
    against the reference-faithful CPU oracle (vqa/models/cx.py:261-333 + counterexamples.py:334-338 restated op for op),
    logits <= 1e-4, loss <= 1e-5, gradients <= 1e-4 of the tensor's max with no floor (out.bias excepted: zero in
    maths).  Inputs are conditioned away from the ReLU kinks (helpers.condition_away_from_kinks)."""
    d = orc.Dims()
    params, batch = _full_size_case(d, 512, 2024)
    compare_with_oracle(d, None, params, batch)


def test_configs1_unconditioned_logits_vs_oracle():
    """The logits are continuous in the pre-activations, so they need no conditioning: B = 512 at configs[1]'s widths drawn
    as they come (NO redraw of triplets near a ReLU kink), all 12 288 logits <= 1e-4 and the loss <= 1e-5 against the
    reference-faithful fp32 CPU oracle.  (Gradients do have a discontinuity at the kink: the test above conditions.)"""
    from helpers import random_case_f32
    d = orc.Dims()
    params = orc.init_params(d, seed=42)
    batch = random_case_f32(31337, 512, d)
    b = to_dev_batch(batch, None, None, None)
    from neuralcx import ops
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    ws = ops.alloc_workspace(dims, dev())
    scores = ops.forward(dims, b, to_dev_params(params), ws)
    lr = ops.ranking_loss(scores, batch["gt"].to(dev()).to(torch.int32))
    with torch.no_grad():
        s_ref = orc.forward_faithful(params, d, batch["image_features"], batch["q_emb"], batch["z_orig"], batch["z_knns"], batch["a_knns"], batch["answer_aids"])
    assert float((scores.cpu() - s_ref).abs().max()) <= 1e-4
    assert abs(float(lr["loss"]) - float(orc.ranking_loss(s_ref, batch["gt"]))) <= 1e-5


def test_in_kernel_clock_stamps_of_the_forward_kernel():
    """ncx_profile_stamps (include/neuralcx.h): while armed, every workgroup of the fp32 forward kernel leaves its entry / exit
    stamps (shader-cycle counter and the 100 MHz constant-rate counter); the held clock comes out in the chip's range, the scores
    are unchanged bit for bit, and a disarmed launch writes nothing."""
    from neuralcx import ops, _lib
    d = orc.Dims()
    B = 512
    rng = np.random.default_rng(3)
    feats = (torch.randn(4096, d.dv).abs() * 0.45).to(dev())
    idx = torch.from_numpy(rng.integers(0, 4096, size=(B, d.K + 1)).astype(np.int32)).to(dev())
    mk = lambda *s: torch.randn(*s, device=dev())
    b = ops.Batch(feats, idx, mk(B, d.dq) * 0.3, mk(B, d.dz), mk(B, d.K, d.dz), mk(B, d.K, d.A) * 2,
                  torch.from_numpy(rng.integers(0, d.A, size=B).astype(np.int32)).to(dev()))
    p = to_dev_params(orc.init_params(d, seed=42))
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    ws = ops.alloc_workspace(dims, dev())
    s0 = ops.forward(dims, b, p, ws).clone()
    st = torch.zeros(16 * 4096, dtype=torch.int64, device=dev())
    _lib.profile_stamps(st)
    try:
        s1 = ops.forward(dims, b, p, ws).clone()
    finally:
        _lib.profile_stamps(None)
    torch.cuda.synchronize()
    assert torch.equal(s0, s1)
    w = st.view(-1, 16).cpu()
    w = w[w[:, 15] > w[:, 14]]
    tile = _lib.plan_query(dims, "MAIN")["tile"]
    assert w.shape[0] == (256 if tile.startswith("192x64") else 512), (w.shape[0], tile)      # one record per workgroup of the plan
    mhz = (w[:, 8] - w[:, 0]).double() / (w[:, 15] - w[:, 14]).double() * 100.0
    assert 800.0 < float(mhz.median()) < 2600.0, float(mhz.median())
    assert set(w[:, 13].tolist()) <= set(range(8))                       # XCC ids
    st.zero_()
    ops.forward(dims, b, p, ws)
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0                                      # disarmed: nothing written


def test_ragged_full_width_batch_vs_oracle():
    """B = 389 at configs[1]'s widths: 9 336 candidate rows = 97 whole 96-row tiles + one with a single triplet (the four-triplet
    forward fold's ragged last tile), an odd number of triplets for the weight-gradient chunks, a batch the wave-per-triplet tail
    kernels do not divide evenly.  Every logit and every gradient against the oracle, same bounds as the configs[1] test."""
    d = orc.Dims()
    params, batch = _full_size_case(d, 389, 777)
    compare_with_oracle(d, None, params, batch)


def test_configs4_shape_fp32_and_bf16_vs_oracles():
    """BASELINE configs[4] at its stated shape (K = 48 candidates, B = 1024, full widths): the fp32 path against the
    faithful oracle with the fp32 tolerances, and the bf16-operand variant against the oracle's bf16 restatement
    (identical operands rounded to bf16, so only the summation order differs: logits <= 2e-3, loss <= 2e-4, gradients
    <= 1e-3 of the tensor's max, no floor)."""
    d = orc.Dims(K=48)
    B = 1024
    params, batch = _full_size_case(d, B, 4048)
    compare_with_oracle(d, None, params, batch)
    # bf16 variant: its own conditioning (bf16 pre-activations / the distance column near a bf16 rounding boundary)
    from helpers import condition_away_from_kinks
    condition_away_from_kinks(params, d, batch, 4049, bf16=True)
    scores, lr, grads = run_hip_bf16(d, params, batch)
    s_ref, l_ref, g_ref = orc.loss_and_grads_bf16(params, d, batch)
    assert np.abs(scores.numpy() - s_ref.numpy()).max() <= 2e-3
    assert abs(float(lr["loss"].cpu()) - float(l_ref)) <= 2e-4
    for k, ref in g_ref.items():
        ref = ref.numpy()
        assert np.isfinite(grads[k]).all(), k
        err = np.abs(grads[k].reshape(ref.shape) - ref).max()
        assert err <= grad_tol(k, ref, 1e-3), (k, err, grad_tol(k, ref, 1e-3))


def test_loss_rank_kernel_known_answers():
    from neuralcx import ops
    g = dict(np.load(GOLDEN + "/g4_recall_loss.npz"))
    for pre in ("dist", "rand"):
        s = torch.from_numpy(g[pre + "_scores"]).to(dev())
        gt = torch.from_numpy(g[pre + "_gt"].astype(np.int32)).to(dev())
        r = ops.ranking_loss(s, gt)
        rank = r["rank"].cpu().numpy()
        for k in (1, 5):
            assert ((rank < k).astype(np.int32) == g["%s_recall%d" % (pre, k)]).all()
        if pre == "rand":
            assert abs(float(r["loss"]) - float(g["rand_loss"])) < 1e-5
            assert np.abs(r["dscores"].cpu().numpy() - g["rand_dscores"]).max() < 1e-7
    # ties: all-equal scores -> rank == gt index (deterministic tie rule), never an undefined top-k order
    s = torch.zeros(8, 24, device=dev())
    gt = torch.arange(8, dtype=torch.int32, device=dev())
    assert (ops.ranking_loss(s, gt)["rank"].cpu().numpy() == np.arange(8)).all()


def test_fused_adam_vs_oracle_and_torch():
    from neuralcx import ops
    torch.manual_seed(1)
    n = 100003
    p0 = torch.randn(n); st = orc.AdamState(); cur = {"p": p0.clone()}
    tp = p0.clone().requires_grad_(True); opt = torch.optim.Adam([tp], lr=1e-3)
    P = p0.clone().to(dev()); M = torch.zeros(n, device=dev()); V = torch.zeros(n, device=dev())
    for step in range(1, 6):
        g = torch.randn(n) * (0.01 * step)
        cur = orc.adam_update(cur, {"p": g}, st, lr=1e-3)
        tp.grad = g.clone(); opt.step()
        ops.adam_step(P, g.to(dev()), M, V, step, lr=1e-3)
        assert torch.allclose(P.cpu(), cur["p"], rtol=2e-6, atol=1e-7)
        assert torch.allclose(P.cpu(), tp.detach(), rtol=2e-6, atol=1e-7)



def test_one_train_step_matches_golden_adam():
    """G5: forward + loss + backward + Adam on the HIP path == the reference's optimizer.step()."""
    from neuralcx import ops
    g, d, spec, params, batch = load_golden("g1_small_L1")
    _, lr, grads = run_hip(d, spec, params, batch)
    for k, p0 in params.items():
        if k == "out.bias":      # zero gradient in maths: Adam turns its round-off noise into +-lr, on every platform differently
            continue
        P = p0.clone().to(dev()).view(-1)
        G = torch.from_numpy(grads[k]).to(dev()).view(-1)
        M = torch.zeros_like(P); V = torch.zeros_like(P)
        ops.adam_step(P, G, M, V, 1, lr=1e-4)
        ref = g["adam1/" + k].reshape(-1)
        # Adam's first step is p - lr * f(g) with f(g) = g / (|g| + 1e-8): it normalises by sqrt(v), so what a gradient error does
        # to the parameter depends on the entry's own size.  The gradients agree to tol = 1e-4 of the tensor's max (checked by
        # the parity tests), so EVERY entry must lie within the image of [g - tol, g + tol] under the step, element by element
        # (+ 2e-6 of fp32 rounding in p): entries well above tol are pinned to ~1e-6, only entries below tol (whose sign the
        # tolerance does not determine) may move by up to 2 lr.
        gr = g["grad/" + k].reshape(-1).astype(np.float64)
        tol = grad_tol(k, gr)
        f = lambda x: x / (np.abs(x) + 1e-8)
        bound = 1e-4 * np.maximum(np.abs(f(gr + tol) - f(gr)), np.abs(f(gr - tol) - f(gr))) + 2e-6
        err = np.abs(P.cpu().numpy().astype(np.float64) - ref)
        assert (err <= bound).all(), (k, float((err - bound).max()))
        # (where |g| >> tol and |g| >> Adam's eps the bound is ~2e-6: e.g. every linear_1.weight entry above 3 tol; tensors whose
        # gradients are of the order of eps = 1e-8 themselves sit in Adam's linear regime and the bound follows f's slope there)


@pytest.mark.parametrize("shape", ["ragged_small", "configs1", "bf16"])
def test_phased_forward_is_bit_identical(shape):
    """ncx_forward_phase PRELUDE (the data-only part: row ids, distance, rank one-hot, softmax statistics) then REST (everything
    that reads the weights) == ncx_forward, bit for bit: scores AND every gradient of the backward that follows -- and the
    prelude really reads no weight: it runs BEFORE the weights are replaced (what the pipelined data-parallel step does: the
    next step's prelude is enqueued while the last Adam slice of the current one is still pending)."""
    from neuralcx import ops
    from neuralcx._lib import NCX_F_ALL, NCX_F_BF16
    if shape == "configs1":
        d, B = orc.Dims(), 512
    elif shape == "bf16":
        d, B = orc.Dims(dv=96, dq=64, dz=24, A=40, H=128, L=2), 9
    else:
        d, B = orc.Dims(dv=96, dq=64, dz=24, A=40, H=64, L=2), 31
    flags = NCX_F_ALL | (NCX_F_BF16 if shape == "bf16" else 0)
    params = orc.init_params(d, seed=9, gain=3.0)
    batch = random_case(33, B, d)
    b = to_dev_batch(batch)
    p = to_dev_params(params)
    gt = batch["gt"].to(dev()).to(torch.int32)
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A, flags=flags, training=True, drop_p=0.25, seed=77)
    ws = ops.alloc_workspace(dims, dev())
    s0 = ops.forward(dims, b, p, ws)
    lr0 = ops.ranking_loss(s0, gt)
    g0 = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    ops.backward(dims, b, p, ws, lr0["dscores"], g0)
    # a fresh workspace (nothing left over from the whole call), weights poisoned during the prelude
    ws2 = ops.alloc_workspace(dims, dev()); ws2.zero_()
    poisoned = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    assert ops.forward(dims, b, poisoned, ws2, phase=ops.FWD_PRELUDE) is None
    s1 = ops.forward(dims, b, p, ws2, phase=ops.FWD_REST)
    assert torch.equal(s0, s1)
    lr1 = ops.ranking_loss(s1, gt)
    g1 = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    ops.backward(dims, b, p, ws2, lr1["dscores"], g1)
    for k in g0:
        assert torch.isfinite(g1[k]).all() and torch.equal(g0[k], g1[k]), k


def test_phased_backward_is_bit_identical():
    """ncx_backward_phase 1 then 2, and 3 then 4, == ncx_backward (what the data-parallel engine relies on)."""
    from neuralcx import ops
    # (H = 64, B = 20: the generic engine's grouped launch; H = 256, B = 128 / 160: the balanced 8-wave TN launch, whose aligned dGt part and
    #  rest sequence are launched separately by phases 5 | 2 and together by phase 0 -- same chunking, same slab slots, same sums)
    for L, H, B in ((1, 64, 20), (2, 64, 20), (1, 256, 128), (2, 256, 160)):
        d = orc.Dims(dv=96, dq=64, dz=24, A=40, H=H, L=L)
        params = orc.init_params(d, seed=9, gain=3.0)
        batch = random_case(31, B, d)
        b = to_dev_batch(batch)
        p = to_dev_params(params)
        dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
        ws = ops.alloc_workspace(dims, dev())
        scores = ops.forward(dims, b, p, ws)
        lr = ops.ranking_loss(scores, batch["gt"].to(dev()).to(torch.int32))
        g0 = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
        ops.backward(dims, b, p, ws, lr["dscores"], g0)
        g12 = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
        ops.backward(dims, b, p, ws, lr["dscores"], g12, phase=1)
        assert torch.isfinite(g12["answer_embedding"]).all() and torch.equal(g12["answer_embedding"], g0["answer_embedding"])
        ops.backward(dims, b, p, ws, lr["dscores"], g12, phase=2)
        for k in g0:
            assert torch.equal(g0[k], g12[k]), k
        # the other cut (3 | 4): everything but the embedding gradient, then the embedding gradient from dGt | dGgt;
        # scaling that workspace block by 2 in between doubles the embedding gradient exactly (linearity: what DP sums)
        g34 = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
        ops.backward(dims, b, p, ws, lr["dscores"], g34, phase=3)
        for k in g0:
            if k != "answer_embedding":
                assert torch.equal(g0[k], g34[k]), k
        blk = ops.ws_dgt_view(dims, ws)
        assert blk.numel() == 2 * d.H * d.A
        ops.backward(dims, b, p, ws, lr["dscores"], g34, phase=4)
        assert torch.equal(g0["answer_embedding"], g34["answer_embedding"])
        blk.mul_(2.0)
        ops.backward(dims, b, p, ws, lr["dscores"], g34, phase=4)
        assert torch.equal(2.0 * g0["answer_embedding"], g34["answer_embedding"])
        # the three-way cut the DP engine uses (5 | 2 | 4): the block first, then linear_1.weight, then the embedding gradient
        g524 = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
        ops.backward(dims, b, p, ws, lr["dscores"], g524, phase=5)
        assert torch.isnan(g524["answer_embedding"]).all() and torch.isnan(g524["w1"]).all()
        blk5 = ops.ws_dgt_view(dims, ws).clone()
        ops.backward(dims, b, p, ws, lr["dscores"], g524, phase=2)
        assert torch.equal(blk5, ops.ws_dgt_view(dims, ws))          # phase 2 leaves the exchanged block alone
        ops.backward(dims, b, p, ws, lr["dscores"], g524, phase=4)
        for k in g0:
            assert torch.equal(g0[k], g524[k]), k


def test_c_abi_rccl_handle_allreduce_on_one_rank():
    """ncx_comm_* / ncx_allreduce (include/neuralcx.h): RCCL is loaded, a communicator is created on this GPU and the fp32 sum
    all-reduce runs on the caller's stream.  One rank is what a one-GPU box allows (RCCL refuses two ranks on one device): the
    sum over one rank is the identity, and it must be ordered after the kernel that produced the buffer and before its reader."""
    from neuralcx import _lib
    torch.cuda.set_device(0)
    uid = _lib.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = _lib.comm_create(uid, 1, 0)
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            x = torch.randn(1 << 22, device=dev())                    # 16 MB: the size of the engine's second bucket
            ref = x.clone()
            y = x * 2.0                                              # producer on the same stream
            _lib.allreduce(comm, y)
            z = y + 1.0                                              # consumer on the same stream
        side.synchronize()
        assert torch.equal(y, ref * 2.0) and torch.equal(z, ref * 2.0 + 1.0)
        _lib.allreduce(comm, torch.empty(0, device=dev()))            # n == 0: nothing enqueued
        with pytest.raises(TypeError):
            _lib.allreduce(comm, torch.zeros(4, dtype=torch.float64, device=dev()))
    finally:
        _lib.comm_destroy(comm)


@pytest.mark.parametrize("shape", [dict(B=20, K=24, H=64, L=1), dict(B=7, K=5, H=50, L=2), dict(B=33, K=32, H=256, L=1)])
def test_fused_training_tail_equals_the_three_calls(shape):
    """NCX_F_FUSED_TAIL (ncx_forward, ncx_train_tail, ncx_backward) against ncx_forward, ncx_loss_rank, ncx_backward: scores, loss,
    ranks, hit counts and every gradient bit-identical (same per-lane arithmetic, same reductions), except d out.bias -- a sum
    that is zero in mathematics, taken in another order."""
    from neuralcx import ops, _lib
    d = orc.Dims(K=shape["K"], dv=96, dq=64, dz=24, A=40, H=shape["H"], L=shape["L"])
    params = orc.init_params(d, seed=12, gain=3.0)
    batch = random_case(41, shape["B"], d)
    b = to_dev_batch(batch)
    p = to_dev_params(params)
    gt = batch["gt"].to(dev()).to(torch.int32)

    def run(fused):
        dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A, training=True, drop_p=0.25, loss_scale=1.0 / shape["B"], seed=77)
        if fused:
            assert ops.fused_tail_ok(dims)
            dims.flags |= _lib.NCX_F_FUSED_TAIL
        ws = ops.alloc_workspace(dims, dev())
        g = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
        scores = ops.forward(dims, b, p, ws)
        r = ops.train_tail(dims, p, ws, scores, gt, g, want_dscores=True) if fused else ops.ranking_loss(scores, gt, scale=1.0 / shape["B"])
        ops.backward(dims, b, p, ws, None if fused else r["dscores"], g)
        return scores, r, g

    s0, r0, g0 = run(False)
    s1, r1, g1 = run(True)
    assert torch.equal(s0, s1)
    for k in ("loss", "loss_rows", "dscores", "rank", "hits"):
        assert torch.equal(r0[k], r1[k]), k
    for k in g0:
        if d.L < 3 and k in ("w3", "b3") or d.L < 2 and k in ("w2", "b2"):
            continue
        if k == "b_out":
            assert abs(float(g0[k]) - float(g1[k])) <= 1e-6, k
        else:
            assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("H", [64, 300])
@pytest.mark.parametrize("K,fold4", [(24, "1"), (24, "0"), (48, "1")])
def test_forward_fold_forms_vs_oracle(K, fold4, H, monkeypatch):
    """The per-triplet forward fold (ncx_main.h, MK_VFOLD) in both tile forms -- 48-row (two triplets per workgroup, effective weight
    tiles in LDS) and 96-row (four triplets at K = 24 / two at K = 48, v_o in the A block's spare row, effective weight = one fma per
    MFMA operand) -- forced at a small size through the experiment hook, against the oracle; and the two forms against each other:
    bit-identical (same expression, same k order).  B = 9 leaves the last 96-row tile ragged, H = 300 the last column tile (44 of 64).
    The 192-row form (eight waves, one workgroup per CU: what the planner takes at configs[1]) is forced the same way."""
    from neuralcx import ops
    monkeypatch.setattr(ops, "EXTRA_FLAGS", 0)          # (bit-identity is a property of the three fp32 forms: under NCX_X6=1 the 192-row form is another kernel, test_x6_forward_192_row_form_vs_oracle)
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_FOLD4", fold4)
    d = orc.Dims(K=K, dv=96 if H == 64 else 160, dq=64, dz=24, A=40, H=H, L=1)
    params = orc.init_params(d, seed=5, gain=3.0)
    batch = random_case(77, 9, d)
    b, p = to_dev_batch(batch), to_dev_params(params)
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    ws = ops.alloc_workspace(dims, dev())
    scores = ops.forward(dims, b, p, ws).cpu()
    ref = orc.forward_faithful(params, d, *[batch[k] for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    assert (scores - ref.reshape(scores.shape)).abs().max() <= 1e-4
    if K == 24:
        monkeypatch.setenv("NCX_FOLD4", "1" if fold4 == "0" else "0")
        other = ops.forward(dims, b, p, ws).cpu()
        assert torch.equal(scores, other)
    # ... and the 192-row form (round 3: one 8-wave workgroup, a triplet per wave -- eight at K = 24, four at K = 48): bit-identical too
    monkeypatch.setenv("NCX_FOLD8", "1")
    eight = ops.forward(dims, b, p, ws).cpu()
    assert torch.equal(scores, eight)



# ---- NCX_F_X6: the balanced TN weight-gradient launch on the bf16 matrix path with three-plane operands (not the default) --------------------
# The SAME tests at the SAME tolerances ("no tolerance may move for it"); `NCX_X6=1 python -m pytest tests -m gpu` runs the
# whole suite that way, these wrappers keep the cases that exercise the launch in the default run.
@pytest.fixture
def x6(monkeypatch):
    from neuralcx import _lib, ops
    monkeypatch.setattr(ops, "EXTRA_FLAGS", _lib.NCX_F_X6)


@pytest.mark.parametrize("case", ["all", "no_a_emb", "no_v_mult", "H512_L2", "K48", "dropout"])
def test_x6_balanced_tn_launch_paths_vs_oracle(case, x6):
    test_balanced_tn_launch_paths_vs_oracle(case)


def test_x6_configs1_full_size_every_logit_and_gradient_vs_oracle(x6):
    test_configs1_full_size_every_logit_and_gradient_vs_oracle()


def test_x6_phased_backward_is_bit_identical(x6):
    test_phased_backward_is_bit_identical()


def test_x6_runs_and_agrees_with_the_fp32_kernels_to_rounding(monkeypatch):
    """The flag really switches kernels (another summation order: not bitwise equal), and what it computes is the fp32 result to fp32
    rounding: at configs[1]'s full size every gradient element within 1e-5 of its tensor's max of the fp32 kernels' (the suite's bound
    against the oracle is 1e-4 of the max)."""
    from neuralcx import _lib, ops
    d = orc.Dims()
    params, batch = _full_size_case(d, 512, 77)
    monkeypatch.setattr(ops, "EXTRA_FLAGS", 0)                       # (the suite may itself be running under NCX_X6=1)
    _, _, g32 = run_hip(d, None, params, batch)
    monkeypatch.setattr(ops, "EXTRA_FLAGS", _lib.NCX_F_X6)
    _, _, g6 = run_hip(d, None, params, batch)
    differs = False
    for k in g32:                                    # (the forward's first layer runs on the split operands too: every gradient moves, by rounding)
        if k == "out.bias":                          # zero in maths (the listwise loss is shift-invariant): rounding noise of 1e-9 on both sides
            continue
        a, b = g32[k], g6[k]
        differs |= not np.array_equal(a, b)
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(a).max(), 1e-30), (k, np.abs(a - b).max(), np.abs(a).max())
    assert differs


@pytest.mark.parametrize("B,K,H,L,dv", [(1, 24, 256, 1, 64), (7, 24, 256, 2, 128), (13, 48, 256, 1, 192), (37, 24, 512, 1, 64)])
def test_x6_fused_v_gradient_kernel_vs_oracle(B, K, H, L, dv, monkeypatch, x6):
    """k_dw_km_x6 (the per-triplet fold pass on the bf16 matrix path with three-plane operands) takes H = multiples of 256 and dv = multiples
    of 64 under NCX_F_X6: forced on small batches (chunks of one triplet, empty chunks, an odd number of reduction steps per chunk, K = 48 =
    two 24-row steps per triplet, two row tiles) and compared with the oracle at the suite's tolerances."""
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_KM_FORCE", "1")
    d = orc.Dims(K=K, dv=dv, dq=50, dz=18, A=45, H=H, L=L)
    params = orc.init_params(d, seed=23 + B, gain=3.0)
    batch = random_case(700 + B, B, d)
    compare_with_oracle(d, None, params, batch)


@pytest.mark.parametrize("H", [64, 300])
@pytest.mark.parametrize("K", [24])
def test_x6_forward_192_row_form_vs_oracle(K, H, monkeypatch, x6):
    """Under NCX_F_X6 the 192-row forward form at K = 24 -- the per-triplet fold and the segments after it (dist | rank, z_k, softmax(a_k) . Gt) --
    runs on the bf16 matrix path with three-plane operands: forced at a small size (B = 9: a ragged row tile; H = 300: a ragged column tile), logits against the
    oracle at the suite's 1e-4; not bit-identical to the fp32 form (another summation order), but equal to it to fp32 rounding."""
    from neuralcx import ops
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_FOLD8", "1")
    d = orc.Dims(K=K, dv=96 if H == 64 else 160, dq=64, dz=24, A=40, H=H, L=1)
    params = orc.init_params(d, seed=5, gain=3.0)
    batch = random_case(77, 9, d)
    b, p = to_dev_batch(batch), to_dev_params(params)
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    ws = ops.alloc_workspace(dims, dev())
    scores = ops.forward(dims, b, p, ws).cpu()
    ref = orc.forward_faithful(params, d, *[batch[k] for k in ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids")])
    assert (scores - ref.reshape(scores.shape)).abs().max() <= 1e-4
    monkeypatch.setattr(ops, "EXTRA_FLAGS", 0)
    dims32 = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    s32 = ops.forward(dims32, b, p, ws).cpu()
    assert not torch.equal(scores, s32)
    assert (scores - s32).abs().max() <= 2e-6 * max(1.0, float(s32.abs().max()))


def test_x6_configs1_unconditioned_logits_vs_oracle(x6):
    test_configs1_unconditioned_logits_vs_oracle()


@pytest.mark.parametrize("B,K,H,L,dv", [(1, 24, 256, 1, 64), (7, 24, 256, 2, 128), (13, 48, 256, 1, 192), (37, 24, 512, 1, 64)])
def test_fused_v_gradient_8_wave_form_vs_oracle(B, K, H, L, dv, monkeypatch):
    """k_dw_km8 (the per-triplet fold pass as one 8-wave workgroup per CU on 256 x 64 tile pairs, fp32 MFMA: what shapes with H a multiple of
    256 and dv a multiple of 64 take) on forced small shapes: chunks of one triplet, empty chunks, an odd number of reduction steps per chunk,
    K = 48 (two 24-row steps per triplet), two row tiles.  (k_dw_km keeps every other shape: test_fused_v_gradient_kernel_vs_oracle.)"""
    from neuralcx import ops
    monkeypatch.setattr(ops, "EXTRA_FLAGS", 0)
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_KM_FORCE", "1")
    d = orc.Dims(K=K, dv=dv, dq=50, dz=18, A=45, H=H, L=L)
    params = orc.init_params(d, seed=23 + B, gain=3.0)
    batch = random_case(700 + B, B, d)
    compare_with_oracle(d, None, params, batch)


@pytest.mark.parametrize("H", [64, 300])
@pytest.mark.parametrize("form", ["96", "192"])
def test_distance_inside_the_fold_forms_vs_oracle(form, H, monkeypatch):
    """The one-triplet-per-wave fold forms (96 / 192-row tiles, K = 24) can compute the pairwise distance ||v_o - v_k + 1e-6|| (cx.py:300) while they
    stream the rows (k_prep then skips the feature rows; experiment hook NCX_DIST_IN_FOLD -- measured a wash at configs[1], so not the default): forced
    at a small ragged size, logits, loss and every gradient -- d linear_1.weight[:, dist] reads the distance the forward kernel stored -- against the
    oracle; and against the same form with k_prep's distance (equal to rounding)."""
    from neuralcx import ops
    monkeypatch.setattr(ops, "EXTRA_FLAGS", 0)
    monkeypatch.setenv("NCX_EXPERIMENT", "1")
    monkeypatch.setenv("NCX_FOLD4", "1")
    monkeypatch.setenv("NCX_FOLD8", "1" if form == "192" else "0")
    monkeypatch.setenv("NCX_DIST_IN_FOLD", "1")
    d = orc.Dims(K=24, dv=96 if H == 64 else 160, dq=64, dz=24, A=40, H=H, L=1)
    params = orc.init_params(d, seed=5, gain=3.0)
    batch = random_case(78, 9, d)
    s_in, _, g_in = compare_with_oracle(d, None, params, batch)
    monkeypatch.delenv("NCX_DIST_IN_FOLD")
    s_prep, _, g_prep = compare_with_oracle(d, None, params, batch)
    assert not torch.equal(s_in, s_prep)                # (the hook really switched the source of the distance)
    assert (s_in - s_prep).abs().max() <= 2e-6 * max(1.0, float(s_prep.abs().max()))
    a, b = g_in["linear_1.weight"], g_prep["linear_1.weight"]
    assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


def test_x6_logits_are_as_close_to_fp64_as_the_fp32_kernels(monkeypatch):
    """What "fp32-grade" means, measured: configs[1] at its full size (B = 512, drawn as it comes), all 12 288 logits of the fp32-MFMA forward and of
    the NCX_F_X6 forward against the oracle's forward evaluated in FLOAT64 on the same fp32 inputs and weights.  The split-bf16 path may not be
    further from fp64 than the fp32 kernels are (a factor 1.5 of slack for summation-order luck), and both sit far inside the suite's 1e-4."""
    from helpers import random_case_f32
    from neuralcx import _lib, ops
    d = orc.Dims()
    params = orc.init_params(d, seed=42)
    batch = random_case_f32(4242, 512, d)
    with torch.no_grad():
        p64 = {k: v.double() for k, v in params.items()}
        ref64 = orc.forward_faithful(p64, d, batch["image_features"].double(), batch["q_emb"].double(), batch["z_orig"].double(),
                                     batch["z_knns"].double(), batch["a_knns"].double(), batch["answer_aids"])
    b, p = to_dev_batch(batch), to_dev_params(params)
    err = {}
    for name, flag in (("fp32", 0), ("x6", _lib.NCX_F_X6)):
        monkeypatch.setattr(ops, "EXTRA_FLAGS", flag)
        dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
        ws = ops.alloc_workspace(dims, dev())
        s = ops.forward(dims, b, p, ws).cpu().double()
        err[name] = float((s - ref64.reshape(s.shape)).abs().max())
    print("max |logit - fp64|: fp32 kernels %.3e, X6 kernels %.3e" % (err["fp32"], err["x6"]))
    assert err["fp32"] <= 1e-4 and err["x6"] <= 1e-4
    assert err["x6"] <= 1.5 * err["fp32"] + 1e-7


def test_a_feature_table_beyond_4_gib_keeps_the_generic_engine():
    """The forward kernel, the 8-wave fold weight-gradient kernel and the balanced TN launch address their operands with 32-bit byte offsets
    (buffer loads: DESIGN 4f); a feature table of 4 GiB or more must therefore stay on the generic engine.  Asked of the planner (no such table
    is allocated): configs[1]'s shape takes the 192-row fold form with the reference's 82 783 images and a plan of the generic engine with 600 000."""
    from neuralcx import _lib, ops
    d = orc.Dims()
    B = 512
    feats = torch.zeros(8, d.dv, device=dev())
    idx = torch.zeros(B, d.K + 1, dtype=torch.int32, device=dev())
    mk = lambda *s: torch.zeros(*s, device=dev())
    b = ops.Batch(feats, idx, mk(B, d.dq), mk(B, d.dz), mk(B, d.K, d.dz), mk(B, d.K, d.A), torch.zeros(B, dtype=torch.int32, device=dev()))
    dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A)
    dims.n_img = 82783
    assert _lib.plan_query(dims, "MAIN")["tile"].startswith("192x64")
    dims.n_img = 600000                                    # x 2048 x 4 bytes = 4.9 GB
    assert "fold" not in _lib.plan_query(dims, "MAIN")["tile"]
