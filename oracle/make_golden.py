"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

Imports gabegrand/VQA-Counterexamples from /root/reference (read-only mount) with the
harness-side shims of SURVEY.md appendix A (the reference files are never edited or copied):

  * empty stub modules for absent third-party imports (``skipthoughts`` -- un-vendored
    submodule, vqa/models/seq2vec.py:7-8; ``torchvision`` -- vqa/models/utils.py:5);
  * ``.cuda()`` -> identity (hard-coded in vqa/models/cx.py:243,266-268,...; no GPU here);
  * ``F.pairwise_distance`` forced to keepdim=True (torch>=0.4 returns [N], torch 0.3
    returned [N,1]; torch.cat at cx.py:309 needs the column).

Then drives ``vqa.models.cx.NeuralModel`` (cx.py:218-333), ``nn.CrossEntropyLoss(size_average=
False)/B`` (counterexamples.py:310,334), ``recallAtK`` semantics (counterexamples.py:501-506,
restated inline because importing counterexamples.py needs h5py/tensorboard), autograd and
``torch.optim.Adam`` (counterexamples.py:275) on seeded inputs and stores inputs + outputs.

The fixtures are DATA (inputs and expected outputs).  This script hard-fails when
/root/reference is absent (e.g. on the GPU box); nothing under tests/ or the product imports it.

Usage:  python oracle/make_golden.py            # rewrites tests/golden/*.npz
"""
import copy
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

if not os.path.isdir(os.path.join(REF, "vqa", "models")):
    raise SystemExit("make_golden.py: /root/reference is not mounted; fixtures can only be "
                     "generated in the build container")

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, REF)


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


class _StandInSeq2Vec(nn.Module):
    """Stand-in for skipthoughts.BayesianUniSkip (un-vendored; GRU 620 -> dim_q).
    The question encoder is an INPUT producer of the hot path, not part of it; its output
    q_emb is stored in the fixture."""
    hidden = 2400

    def __init__(self, dir_st, vocab, dropout=0.25, fixed_emb=False):
        super().__init__()
        self.emb = nn.Embedding(len(vocab) + 1, 32, padding_idx=0)
        self.gru = nn.GRU(32, _StandInSeq2Vec.hidden, batch_first=True)

    def forward(self, wids):
        x = self.emb(wids)
        lens = (wids > 0).sum(1).clamp(min=1)
        out, _ = self.gru(x)
        return out[torch.arange(wids.shape[0]), lens - 1]


_st = _stub("skipthoughts")
_st.BayesianUniSkip = _StandInSeq2Vec
_tv = _stub("torchvision")
_tv.models = _stub("torchvision.models")
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self
_pd = F.pairwise_distance
F.pairwise_distance = lambda x1, x2, p=2.0, eps=1e-6, keepdim=True: _pd(x1, x2, p, eps, True)

import vqa.models as ref_models            # noqa: E402  (the reference package)
from vqa.models.cx import NeuralModel      # noqa: E402


FULL_SPEC = dict(name="golden", pretrained_vqa=False, trainable_vqa=False, v_emb=True, v_mult=True,
                 v_dist=True, v_rank=True, q_emb=True, pretrained_emb=False, a_emb=True, z_emb=True)


def build(dims, H, L, seed, spec, drop_p=0.25, gain=1.0):
    dv, dq, dz, A = dims
    _StandInSeq2Vec.hidden = dq
    vocab_words = ["w%d" % i for i in range(40)]
    vocab_answers = ["a%d" % i for i in range(A)]
    opt = dict(arch="MutanNoAtt",
               seq2vec=dict(arch="skipthoughts", dir_st="", type="BayesianUniSkip", dropout=0.25,
                            fixed_emb=False),
               fusion=dict(dim_v=dv, dim_q=dq, dim_hv=dz, dim_hq=dz, dim_mm=dz, R=3 if dz < 100 else 10,
                           dropout_v=0.5, dropout_q=0.5, activation_v="tanh", activation_q="tanh",
                           dropout_hv=0, dropout_hq=0),
               classif=dict(dropout=0.5))
    torch.manual_seed(seed)
    vqa = ref_models.factory(copy.deepcopy(opt), vocab_words, vocab_answers, cuda=False, data_parallel=False)
    vqa.eval()
    m = NeuralModel(model_spec=spec, dim_h=H, n_layers=L, emb=None, drop_p=drop_p, vqa_model=vqa,
                    knn_size=24, trainable_vqa=False)
    # numpy-seeded trainable weights (stable across torch builds); same distributions as torch init
    rng = np.random.default_rng(seed + 1000)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith("vqa_model."):
                continue
            if name == "answer_embedding.weight":
                a = rng.standard_normal(tuple(p.shape), dtype=np.float32)
            else:
                fan_in = p.shape[1] if p.dim() == 2 else dict(m.named_parameters())[name.replace("bias", "weight")].shape[1]
                b = 1.0 / np.sqrt(fan_in)
                a = (rng.uniform(-b, b, size=tuple(p.shape)) * gain).astype(np.float32)
            p.copy_(torch.from_numpy(a))
    return m, vqa, len(vocab_words)


def recall_vec(scores, gt, k):
    # counterexamples.py:501-506 restated (topk indices == gt)
    _, top = scores.topk(k)
    return (top.numpy() == gt.numpy().reshape(-1, 1)).sum(axis=1).astype(np.int32)


def gaps_ok(scores, gt, min_gap):
    s = scores.detach().numpy()
    g = gt.numpy()
    sg = s[np.arange(len(g)), g][:, None]
    d = np.abs(s - sg)
    d[np.arange(len(g)), g] = np.inf
    return d.min() > min_gap


def run_case(name, dims, H, L, B, seed, spec_over=None, scale_feat=0.45, full_grads=True,
             adam=False, min_gap=1e-4, gain=1.0):
    spec = dict(FULL_SPEC, **(spec_over or {}))
    dv, dq, dz, A = dims
    for attempt in range(50):
        s = seed + 7919 * attempt
        m, vqa, V = build(dims, H, L, s, spec, gain=gain)
        m.eval()          # dropout off: parity in eval mode (torch's Philox mask is not reproducible)
        rng = np.random.default_rng(s)
        feats = (np.abs(rng.standard_normal((B, 25, dv))) * scale_feat).astype(np.float32)
        wids = np.zeros((B, 26), np.int64)
        for b in range(B):
            n = rng.integers(3, 27)
            wids[b, :n] = rng.integers(1, V + 1, size=n)
        aids = rng.integers(0, A, size=B).astype(np.int64)
        if B >= 4:
            aids[1] = aids[0]            # duplicate answer id: exercises the embedding scatter-add
        gt = rng.integers(0, 24, size=B).astype(np.int64)
        feats_t, wids_t, aids_t, gt_t = map(torch.from_numpy, (feats, wids, aids, gt))

        # capture what vqa_forward hands to the MLP (cx.py:270-271)
        cap = {}
        orig_vf = m.vqa_forward

        def vf(image_features, question_wids):
            r = orig_vf(image_features, question_wids)
            cap["a_orig"], cap["z_orig"], cap["a_knns"], cap["z_knns"], cap["q_emb"] = [
                t.detach().clone() for t in r]
            return r
        m.vqa_forward = vf

        scores = m(feats_t, wids_t, aids_t)                                   # cx.py:261
        loss = nn.CrossEntropyLoss(size_average=False)(scores, gt_t) / B      # counterexamples.py:310,334
        if not gaps_ok(scores, gt_t, min_gap):
            continue
        m.zero_grad()
        loss.backward()
        break
    else:
        raise RuntimeError("no seed with safe rank gaps for " + name)

    out = dict(seed=np.int64(s), dims=np.array([24, dv, dq, dz, 2400, A, H, L], np.int64), B=np.int64(B),
               spec=np.array([int(spec[k]) for k in ("v_emb", "v_mult", "v_dist", "v_rank", "q_emb", "a_emb", "z_emb")],
                             np.int64),
               image_features=feats, question_wids=wids, answer_aids=aids, gt=gt,
               q_emb=cap["q_emb"].numpy(), z_orig=cap["z_orig"].numpy(), z_knns=cap["z_knns"].numpy(),
               a_orig=cap["a_orig"].numpy(), a_knns=cap["a_knns"].numpy(),
               scores=scores.detach().numpy(), loss=np.float32(loss.item()),
               recall1=recall_vec(scores.detach(), gt_t, 1), recall5=recall_vec(scores.detach(), gt_t, 5))
    named = {n: p for n, p in m.named_parameters() if not n.startswith("vqa_model.")}
    for n, p in m.named_parameters():
        if n.startswith("vqa_model."):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n   # frozen (cx.py:79-80)
    idx_rng = np.random.default_rng(12345)
    for n, p in named.items():
        g = p.grad.detach().numpy()
        out["gradnorm/" + n] = np.float64(np.linalg.norm(g.astype(np.float64)))
        if full_grads:
            out["grad/" + n] = g.copy()
        else:
            flat = g.reshape(-1)
            ii = idx_rng.integers(0, flat.size, size=min(2048, flat.size))
            out["gradidx/" + n] = ii.astype(np.int64)
            out["gradval/" + n] = flat[ii].copy()
            if flat.size <= 4096:
                out["grad/" + n] = g.copy()
    if full_grads:      # G6 (small cases): the frozen MUTAN weights, so vqa_forward itself can be checked (cx.py:64-104)
        for n, p in m.named_parameters():
            if n.startswith("vqa_model.fusion.") or n.startswith("vqa_model.linear_classif."):
                out["vqa/" + n[len("vqa_model."):]] = p.detach().numpy().copy()
        out["vqa_R"] = np.int64(len(vqa.fusion.list_linear_hv))
    out["weight_seed"] = np.int64(s + 1000)
    out["weight_gain"] = np.float64(gain)
    if adam:
        opt = torch.optim.Adam([p for n, p in named.items()], lr=1e-4)       # counterexamples.py:275-276
        opt.step()
        for n, p in named.items():
            out["adam1/" + n] = p.detach().numpy().copy()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s seed=%d loss=%.6f R@1=%d/%d R@5=%d/%d  %.1f KB" % (
        name, s, loss.item(), out["recall1"].sum(), B, out["recall5"].sum(), B, os.path.getsize(path) / 1024))


def recall_known_answers():
    """G4: recallAtK known answers.  DistanceBaseline scores (cx.py:33-44) are reversed(range(K)),
    so recall@k == (knn_index < k); plus random score matrices with safe gaps."""
    rng = np.random.default_rng(99)
    K = 24
    cases = {}
    B = 64
    dist_scores = np.tile(np.arange(K - 1, -1, -1, dtype=np.float32), (B, 1))
    gt = rng.integers(0, K, size=B).astype(np.int64)
    cases["dist_scores"], cases["dist_gt"] = dist_scores, gt
    for k in (1, 5):
        cases["dist_recall%d" % k] = recall_vec(torch.from_numpy(dist_scores), torch.from_numpy(gt), k)
        assert (cases["dist_recall%d" % k] == (gt < k)).all()
    s = rng.standard_normal((256, K)).astype(np.float32) * 3
    g = rng.integers(0, K, size=256).astype(np.int64)
    cases["rand_scores"], cases["rand_gt"] = s, g
    for k in (1, 5):
        cases["rand_recall%d" % k] = recall_vec(torch.from_numpy(s), torch.from_numpy(g), k)
    st, gt_t = torch.from_numpy(s), torch.from_numpy(g)
    cases["rand_loss"] = np.float32((nn.CrossEntropyLoss(size_average=False)(st, gt_t) / 256).item())
    st.requires_grad_(True)
    (nn.CrossEntropyLoss(size_average=False)(st, gt_t) / 256).backward()
    cases["rand_dscores"] = st.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g4_recall_loss.npz"), **cases)
    print("g4_recall_loss written")


if __name__ == "__main__":
    small = (64, 48, 16, 20)          # dim_v, dim_q, dim_mm, |answers|   (dim_a = 2400 is fixed: cx.py:235)
    for L in (1, 2, 3):
        run_case("g1_small_L%d" % L, small, H=16, L=L, B=8, seed=100 + L, adam=(L == 1), gain=3.0)
    run_case("g1_small_H20_L2", (68, 52, 20, 37), H=20, L=2, B=5, seed=140, gain=3.0)     # ragged: nothing a multiple of 16
    run_case("g3_lesion_nomult_nodist", small, H=16, L=1, B=8, seed=200, spec_over=dict(v_mult=False, v_dist=False), gain=3.0)
    run_case("g2_full_B4_H256_L1", (2048, 2400, 360, 2000), H=256, L=1, B=4, seed=300, full_grads=False)
    run_case("g2_full_B3_H300_L2", (2048, 2400, 360, 2000), H=300, L=2, B=3, seed=320, full_grads=False)
    recall_known_answers()
