"""CPU oracle for the NeuralCX hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``vqa-counterexamples_amd/``) never does: it
calls the HIP library through the C-ABI in ``include/neuralcx.h`` and fails loudly when
that library is missing.

What this restates (reference = gabegrand/VQA-Counterexamples, paths relative to its root):

* ``NeuralModel.forward``            vqa/models/cx.py:261-333  -> :func:`forward_faithful`
* answer-embedding lookups (K3/K4)    vqa/models/cx.py:280-282  -> inside :func:`forward_faithful`
* listwise ranking loss               counterexamples.py:310,334 -> :func:`ranking_loss`
* ``recallAtK``                       counterexamples.py:501-506 -> :func:`recall_at_k`
* one training step (Adam)            counterexamples.py:325-339 -> :func:`train_step`
* ``CXModelBase.vqa_forward`` + MUTAN  vqa/models/cx.py:64-104, fusion.py:78-121, noatt.py:24-29 -> :func:`mutan_vqa_forward`

The arithmetic of the path is PyTorch's own operators (nn.Linear / softmax / bmm /
pairwise_distance / CrossEntropyLoss / topk / optim.Adam); the reference pins no torch
version (requirements.txt:2) and ships no tests.  Parity is therefore pinned by golden
vectors generated from the reference itself, imported in the build container under
torch 2.10.0 CPU by ``oracle/make_golden.py`` and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against every one of them.

Two semantic choices follow torch 2.x (what the reference executes today):
``pairwise_distance`` = ||x1 - x2 + 1e-6||_2 kept as an ``[B,1]`` column, and
``CrossEntropyLoss(size_average=False)`` = ``reduction='sum'``.

Dropout: torch's Philox stream cannot be reproduced by another implementation, so the
train-mode oracle takes explicit 0/1 keep-masks (one ``[B*K, H]`` matrix per hidden layer,
row ``b*K + k``) and applies ``relu(x) * mask / (1 - p)``.  :func:`dropout_keep_mask`
restates the counter-based generator the HIP kernels use so both sides can build the
same masks from a seed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

DIM_A = 2400  # hard-coded in the reference: vqa/models/cx.py:235

# Lesion switches of model_spec (vqa/models/cx.py:265-307).  True = feature present.
DEFAULT_SPEC = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True,
                    q_emb=True, a_emb=True, z_emb=True)


@dataclass
class Dims:
    """Sizes of one NeuralCX problem (reference names in vqa/models/cx.py:230-251)."""
    K: int = 24          # knn_size
    dv: int = 2048       # dim_v
    dq: int = 2400       # dim_q
    dz: int = 360        # dim_mm
    da: int = DIM_A      # dim_a
    A: int = 2000        # ans_size
    H: int = 256         # dim_h
    L: int = 1           # n_layers

    @property
    def din(self) -> int:
        # cx.py:245-251
        return 3 * self.dv + 2 * self.da + 2 * self.dz + self.dq + self.K + 1

    def offsets(self) -> Dict[str, int]:
        """Column offsets of the concat (cx.py:309-320) == layout of linear_1.weight."""
        o, out = 0, {}
        for name, w in (("v_orig", self.dv), ("v_other", self.dv), ("v_mult", self.dv),
                        ("v_dist", 1), ("v_rank", self.K), ("q_emb", self.dq),
                        ("z_orig", self.dz), ("z_other", self.dz),
                        ("a_emb_gt", self.da), ("a_emb_other", self.da)):
            out[name] = o
            o += w
        assert o == self.din
        return out


def param_shapes(d: Dims) -> Dict[str, tuple]:
    """state_dict keys/shapes of the trainable part (cx.py:240-257)."""
    s = {"answer_embedding.weight": (d.A, d.da),
         "linear_1.weight": (d.H, d.din), "linear_1.bias": (d.H,)}
    if d.L >= 2:
        s["linear_2.weight"] = (d.H, d.H); s["linear_2.bias"] = (d.H,)
    if d.L >= 3:
        s["linear_3.weight"] = (d.H, d.H); s["linear_3.bias"] = (d.H,)
    s["out.weight"] = (1, d.H); s["out.bias"] = (1,)
    return s


def init_params(d: Dims, seed: int = 42, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """numpy-seeded parameters with torch's default init *distributions*
    (Embedding N(0,1); Linear U(+-1/sqrt(fan_in)) times ``gain``).  Same draw order as
    oracle/make_golden.py, so fixtures store only the seed."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in param_shapes(d).items():
        if name == "answer_embedding.weight":
            a = rng.standard_normal(shp, dtype=np.float32)
        else:
            fan_in = shp[1] if len(shp) == 2 else param_shapes(d)[name.replace("bias", "weight")][1]
            bound = 1.0 / math.sqrt(fan_in)
            a = (rng.uniform(-bound, bound, size=shp) * gain).astype(np.float32)
        p[name] = torch.from_numpy(a)
    return p


# --------------------------------------------------------------------------------------
# forward, op-for-op as the reference executes it
# --------------------------------------------------------------------------------------
def forward_faithful(params: Dict[str, torch.Tensor], d: Dims,
                     image_features: torch.Tensor,     # [B, K+1, dv]
                     q_emb: torch.Tensor,              # [B, dq]
                     z_orig: torch.Tensor,             # [B, dz]
                     z_knns: torch.Tensor,             # [B, K, dz]
                     a_knns: torch.Tensor,             # [B, K, A] logits (or [B,K,da] if not a_emb)
                     answer_aids: torch.Tensor,        # [B] int64
                     spec: Optional[dict] = None,
                     drop_p: float = 0.0,
                     keep_masks: Optional[Sequence[torch.Tensor]] = None,
                     a_emb_gt_override: Optional[torch.Tensor] = None,
                     v_rank_override: Optional[torch.Tensor] = None,
                     taps: Optional[dict] = None) -> torch.Tensor:
    """scores[B,K].  Follows vqa/models/cx.py:261-333 line by line: a Python loop over the
    K candidates, ``torch.cat`` of the ten segments, ``F.linear`` + relu (+ dropout) per
    hidden layer, ``out``.  ``keep_masks`` = None means eval mode (dropout is identity).

    Lesions that the reference fills with ``torch.rand`` (cx.py:266,274-277,284-285,307) are
    supplied by the caller as ordinary inputs (``a_knns`` then holds the [B,K,da] noise block,
    ``a_emb_gt_override`` the [B,da] one, ``v_rank_override`` [B,K,K] the per-candidate rank
    noise); zero-lesions (v_mult, v_dist) are handled here.
    ``taps`` (tests only): receives ``pre1`` [B,K,H], the pre-activations of linear_1, and ``dist`` [B,K].
    """
    spec = dict(DEFAULT_SPEC, **(spec or {}))
    B = image_features.shape[0]
    assert image_features.shape[1] == d.K + 1                       # cx.py:263
    v_orig = image_features[:, 0]                                   # cx.py:267
    v_knns = image_features[:, 1:]                                  # cx.py:268
    E = params["answer_embedding.weight"]
    if spec["a_emb"]:
        a_emb_gt = F.embedding(answer_aids, E)                      # cx.py:280
        p = F.softmax(a_knns, dim=-1)                               # cx.py:281
        a_emb_knns = torch.bmm(p, E.view(1, d.A, d.da).expand(B, -1, -1))   # cx.py:282
    else:
        a_emb_gt = a_emb_gt_override
        a_emb_knns = a_knns
    scores = []
    for i in range(d.K):                                            # cx.py:289
        v_other = v_knns[:, i]
        z_other = z_knns[:, i]
        a_emb_other = a_emb_knns[:, i]
        if spec["v_mult"]:
            v_mult = v_orig * v_other                               # cx.py:296
        else:
            v_mult = torch.zeros(B, d.dv)
        if spec["v_dist"]:
            v_dist = F.pairwise_distance(v_orig, v_other, keepdim=True)   # cx.py:300 (+1e-6 inside the norm)
        else:
            v_dist = torch.zeros(B, 1)
        if spec["v_rank"]:
            v_rank = torch.zeros(B, d.K)
            v_rank[:, i] = 1                                        # cx.py:304-305
        else:
            v_rank = v_rank_override[:, i]
        x = torch.cat((v_orig, v_other, v_mult, v_dist, v_rank, q_emb, z_orig, z_other,
                       a_emb_gt, a_emb_other), dim=1)               # cx.py:309-320
        h = x
        for l in range(1, d.L + 1):                                 # cx.py:322-326
            pre = F.linear(h, params[f"linear_{l}.weight"], params[f"linear_{l}.bias"])
            if taps is not None and l == 1:
                taps.setdefault("pre1", []).append(pre.detach())
                taps.setdefault("dist", []).append(v_dist.detach()[:, 0])
            h = F.relu(pre)
            if keep_masks is not None:
                m = keep_masks[l - 1].view(B, d.K, d.H)[:, i]
                h = h * m / (1.0 - drop_p)
        scores.append(F.linear(h, params["out.weight"], params["out.bias"]))   # cx.py:327
    if taps is not None:
        taps["pre1"] = torch.stack(taps["pre1"], dim=1)
        taps["dist"] = torch.stack(taps["dist"], dim=1)
    return torch.cat(scores, dim=1)                                 # cx.py:331


# --------------------------------------------------------------------------------------
# BASELINE configs[4]: "bf16 weights" variant (net-new: the reference computes in fp32 only)
# --------------------------------------------------------------------------------------
def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.bfloat16().float()          # round to nearest even, as v_cvt_pk_bf16_f32


class _Bf16Product(torch.autograd.Function):
    """y = bf16(x) . bf16(w)^T with fp32 accumulation; grad_w = bf16(g)^T . bf16(x) (what the bf16 MFMA path
    computes: products of bf16 values are exact in fp32, so only the summation order differs); x needs no gradient."""

    @staticmethod
    def forward(ctx, x, w):
        xb = _bf(x)
        ctx.save_for_backward(xb)
        return xb @ _bf(w).t()

    @staticmethod
    def backward(ctx, g):
        (xb,) = ctx.saved_tensors
        return None, _bf(g).t() @ xb


class _Bf16GtProduct(torch.autograd.Function):
    """Gt = bf16(W1ak) . bf16(E)^T;  grad_W1ak = bf16(g) . bf16(E),  grad_E = bf16(g)^T . bf16(W1ak)."""

    @staticmethod
    def forward(ctx, w, e):
        wb, eb = _bf(w), _bf(e)
        ctx.save_for_backward(wb, eb)
        return wb @ eb.t()

    @staticmethod
    def backward(ctx, g):
        wb, eb = ctx.saved_tensors
        gb = _bf(g)
        return gb @ eb, gb.t() @ wb


class _SharedAnswerEmbedding(torch.autograd.Function):
    """E[aid] . W1agt^T of the per-triplet shared segments: fp32 forward and fp32 grad_W1agt (they ride in the fp32 Sh /
    shared-column GEMMs); grad_E = bf16(dGgt)^T . bf16(W1agt) with dGgt[h][a] = sum over triplets with answer a of g[b][h]
    (the second half of the stacked bf16 dE product)."""

    @staticmethod
    def forward(ctx, e, w, aids):
        ctx.save_for_backward(e, w, aids)
        return F.embedding(aids, e) @ w.t()

    @staticmethod
    def backward(ctx, g):
        e, w, aids = ctx.saved_tensors
        dggt = torch.zeros(w.shape[0], e.shape[0]).index_add_(1, aids, g.t().contiguous())
        return _bf(dggt).t() @ _bf(w), g.t() @ F.embedding(aids, e), None


def forward_bf16(params: Dict[str, torch.Tensor], d: Dims, image_features, q_emb, z_orig, z_knns, a_knns, answer_aids,
                 drop_p: float = 0.0, keep_masks: Optional[Sequence[torch.Tensor]] = None,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """scores[B,K] of the NCX_F_BF16 variant: the same network as forward_faithful (vqa/models/cx.py:261-333) in its
    segmented form -- the four per-triplet segments in fp32; Gt = W1[:, a_emb_other] . E^T and the five per-candidate
    segments [v_k | v_o*v_k | dist, rank | z_k | softmax(a_k)] against [W1 slices | Gt] with BOTH operands rounded to
    bf16 and fp32 accumulation; the answer-embedding gradients from bf16 images of dGt | dGgt, E and W1[:, a_*]; hidden
    layers >= 2, `out`, loss and Adam in fp32."""
    B, K = image_features.shape[0], d.K
    W1, b1, E = params["linear_1.weight"], params["linear_1.bias"], params["answer_embedding.weight"]
    o, c = {}, 0
    for name, n in (("v_orig", d.dv), ("v_other", d.dv), ("v_mult", d.dv), ("v_dist", 1), ("v_rank", K), ("q_emb", d.dq),
                    ("z_orig", d.dz), ("z_other", d.dz), ("a_gt", d.da), ("a_other", d.da)):      # cx.py:309-320
        o[name] = (c, c + n); c += n
    cols = lambda n: W1[:, o[n][0]:o[n][1]]
    v_o, v_k = image_features[:, 0], image_features[:, 1:]
    shared = torch.cat((v_o, q_emb, z_orig), 1) @ torch.cat((cols("v_orig"), cols("q_emb"), cols("z_orig")), 1).t() + b1 \
        + _SharedAnswerEmbedding.apply(E, cols("a_gt"), answer_aids)
    gt_mat = _Bf16GtProduct.apply(cols("a_other"), E)                                       # [H, A]
    dist = (v_o[:, None, :] - v_k + 1e-6).norm(dim=2, keepdim=True)                         # cx.py:300
    rank = torch.eye(K).view(1, K, K).expand(B, K, K)                                       # cx.py:304-305
    xc = torch.cat((v_k, v_o[:, None, :] * v_k, dist, rank, z_knns, F.softmax(a_knns, dim=-1)), 2).reshape(B * K, -1)
    wc = torch.cat((cols("v_other"), cols("v_mult"), cols("v_dist"), cols("v_rank"), cols("z_other"), gt_mat), 1)
    pre = shared.repeat_interleave(K, 0) + _Bf16Product.apply(xc, wc)
    if taps is not None:
        taps["pre1"] = pre.detach().view(B, K, d.H)
        taps["dist"] = dist.detach()[:, :, 0]
    h = F.relu(pre)
    if keep_masks is not None:
        h = h * keep_masks[0].view(B * K, d.H) / (1.0 - drop_p)
    for l in range(2, d.L + 1):
        h = F.relu(F.linear(h, params[f"linear_{l}.weight"], params[f"linear_{l}.bias"]))
        if keep_masks is not None:
            h = h * keep_masks[l - 1].view(B * K, d.H) / (1.0 - drop_p)
    return F.linear(h, params["out.weight"], params["out.bias"]).view(B, K)


def loss_and_grads_bf16(params, d: Dims, batch: dict, drop_p=0.0, keep_masks=None):
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    scores = forward_bf16(leaf, d, batch["image_features"], batch["q_emb"], batch["z_orig"], batch["z_knns"], batch["a_knns"],
                          batch["answer_aids"], drop_p=drop_p, keep_masks=keep_masks)
    loss = ranking_loss(scores, batch["gt"])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return scores.detach(), loss.detach(), grads


# --------------------------------------------------------------------------------------
# the frozen MUTAN producer upstream of the MLP
# --------------------------------------------------------------------------------------
def mutan_vqa_forward(vp: Dict[str, torch.Tensor], image_features: torch.Tensor, q_emb: torch.Tensor, R: int):
    """CXModelBase.vqa_forward below the question encoder, op for op (vqa/models/cx.py:69-96 with
    MutanFusion.forward, vqa/models/fusion.py:78-121, and AbstractNoAtt._classif, vqa/models/noatt.py:24-29;
    eval mode: every dropout is the identity; options/cx/*.yaml: tanh on v and q, no other activation).
    vp: state_dict entries of the VQA model ("fusion.linear_v.weight", ..., "linear_classif.bias").
    -> a_orig [B,A], z_orig [B,dz], a_knns [B,K,A], z_knns [B,K,dz]."""
    B, K1 = image_features.shape[0], image_features.shape[1]
    v = image_features.reshape(B * K1, -1)                                             # cx.py:69-70
    q = q_emb.view(B, 1, -1).expand(B, K1, q_emb.shape[-1]).contiguous().view(B * K1, -1)   # cx.py:83-84
    x_v = torch.tanh(F.linear(v, vp["fusion.linear_v.weight"], vp["fusion.linear_v.bias"]))   # fusion.py:82-87
    x_q = torch.tanh(F.linear(q, vp["fusion.linear_q.weight"], vp["fusion.linear_q.bias"]))   # fusion.py:88-93
    x_mm = []
    for i in range(R):                                                                 # fusion.py:96-109
        hv = F.linear(x_v, vp["fusion.list_linear_hv.%d.weight" % i], vp["fusion.list_linear_hv.%d.bias" % i])
        hq = F.linear(x_q, vp["fusion.list_linear_hq.%d.weight" % i], vp["fusion.list_linear_hq.%d.bias" % i])
        x_mm.append(torch.mul(hq, hv))
    z = torch.stack(x_mm, dim=1).sum(1)                                                # fusion.py:111-112
    a = F.linear(z, vp["linear_classif.weight"], vp["linear_classif.bias"])            # noatt.py:28
    a, z = a.view(B, K1, -1), z.view(B, K1, -1)                                         # cx.py:90-96
    return a[:, 0].contiguous(), z[:, 0].contiguous(), a[:, 1:].contiguous(), z[:, 1:].contiguous()


# --------------------------------------------------------------------------------------
# loss / metric
# --------------------------------------------------------------------------------------
def ranking_loss(scores: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """counterexamples.py:310,334: CrossEntropyLoss(size_average=False)(scores, gt) / len(batch)."""
    return F.cross_entropy(scores, gt, reduction="sum") / scores.shape[0]


def recall_at_k(scores: torch.Tensor, gt: torch.Tensor, k: int = 5) -> np.ndarray:
    """counterexamples.py:501-506: 1 where gt is among the top-k scores, else 0 (int array [B])."""
    assert scores.shape[0] == gt.shape[0]
    _, top = scores.topk(k)
    return (top.numpy() == gt.numpy().reshape(-1, 1)).sum(axis=1)


def rank_of_gt(scores: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """Deterministic rank used by the HIP loss kernel: #{k: s_k > s_gt} + #{k < gt: s_k == s_gt}.
    recall@k == (rank < k) whenever there are no ties at the boundary (fixtures guarantee it)."""
    s = np.asarray(scores); g = np.asarray(gt).astype(np.int64)
    sg = s[np.arange(s.shape[0]), g][:, None]
    k = np.arange(s.shape[1])[None, :]
    return ((s > sg).sum(1) + ((s == sg) & (k < g[:, None])).sum(1)).astype(np.int32)


# --------------------------------------------------------------------------------------
# counter-based dropout generator shared with the HIP kernels (csrc/ncx_common.h: ncx_hash_u32)
# --------------------------------------------------------------------------------------
def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x85EBCA6B)
    x ^= x >> np.uint32(13); x *= np.uint32(0xC2B2AE35)
    x ^= x >> np.uint32(16)
    return x


def dropout_keep_mask(seed: int, layer: int, rows: int, H: int, p: float) -> torch.Tensor:
    """keep[r, n] = u(r, n) >= p with u = top 24 bits of a murmur3-finalised counter.
    layer is 1-based; seed is the 64-bit per-step seed the host passes to ncx_forward."""
    with np.errstate(over="ignore"):
        idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(H)
               + np.arange(H, dtype=np.uint64)[None, :])
        lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (idx >> np.uint64(32)).astype(np.uint32)
        s_lo = np.uint32(seed & 0xFFFFFFFF); s_hi = np.uint32((seed >> 32) & 0xFFFFFFFF)
        x = _mix32(lo ^ s_lo)
        x = _mix32(x ^ hi ^ s_hi ^ (np.uint32(layer) * np.uint32(0x9E3779B9)))
    u = (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u >= np.float32(p)).astype(np.float32))


# --------------------------------------------------------------------------------------
# one optimisation step, as counterexamples.py:325-339 does it
# --------------------------------------------------------------------------------------
@dataclass
class AdamState:
    step: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def loss_and_grads(params, d: Dims, batch: dict, spec=None, drop_p=0.0, keep_masks=None):
    """forward + CE/B + autograd backward.  Returns (scores, loss, grads dict)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    scores = forward_faithful(leaf, d, batch["image_features"], batch["q_emb"], batch["z_orig"],
                              batch["z_knns"], batch["a_knns"], batch["answer_aids"], spec=spec,
                              drop_p=drop_p, keep_masks=keep_masks,
                              a_emb_gt_override=batch.get("a_emb_gt"),
                              v_rank_override=batch.get("v_rank"))
    loss = ranking_loss(scores, batch["gt"])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return scores.detach(), loss.detach(), grads


def adam_update(params, grads, state: AdamState, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (counterexamples.py:275-276): no weight decay, no amsgrad.
    Written out so the fused HIP Adam can be checked element-wise; validated against
    torch.optim.Adam itself in tests/test_oracle_golden.py."""
    state.step += 1
    t = state.step
    b1, b2 = betas
    bc1 = 1.0 - b1 ** t
    bc2_sqrt = math.sqrt(1.0 - b2 ** t)
    step_size = lr / bc1
    out = {}
    for k, p in params.items():
        g = grads[k]
        m = state.m.setdefault(k, torch.zeros_like(p))
        v = state.v.setdefault(k, torch.zeros_like(p))
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        out[k] = p - step_size * (m / denom)
    return out


def train_step(params, d: Dims, batch, state: AdamState, lr=1e-4, spec=None, drop_p=0.0,
               keep_masks=None):
    scores, loss, grads = loss_and_grads(params, d, batch, spec, drop_p, keep_masks)
    return adam_update(params, grads, state, lr=lr), scores, loss, grads


# --------------------------------------------------------------------------------------
# "reference-faithful" trainable module for the CPU baseline timing (bench.py cpu_baseline)
# --------------------------------------------------------------------------------------
class FaithfulCPUModel(torch.nn.Module):
    """nn.Module wrapper around :func:`forward_faithful` with torch's own Dropout, so the CPU
    baseline executes the same operator sequence as cx.py:280-331 + counterexamples.py:325-339
    (24-iteration cat+Linear loop, softmax+bmm expected embedding, autograd, optim.Adam).
    ``keep_masks`` (a list of [B*K, H] keep masks, one per hidden layer, e.g. from :func:`dropout_keep_mask`) replaces
    torch's Dropout stream by the counter-based one the HIP kernels use, so that a HIP run and this model can be trained
    on identical batches with identical dropout (bench.py's Recall comparison); None = nn.Dropout."""

    def __init__(self, d: Dims, drop_p: float = 0.25, seed: int = 42):
        super().__init__()
        self.d, self.drop_p = d, drop_p
        self.keep_masks = None
        p = init_params(d, seed)
        self.answer_embedding = torch.nn.Embedding(d.A, d.da)
        self.linear_1 = torch.nn.Linear(d.din, d.H)
        if d.L >= 2: self.linear_2 = torch.nn.Linear(d.H, d.H)
        if d.L >= 3: self.linear_3 = torch.nn.Linear(d.H, d.H)
        self.out = torch.nn.Linear(d.H, 1)
        self.drop = torch.nn.Dropout(p=drop_p)
        with torch.no_grad():
            for k, v in self.state_dict().items():
                v.copy_(p[k])

    def forward(self, image_features, q_emb, z_orig, z_knns, a_knns, answer_aids):
        d = self.d
        B = image_features.shape[0]
        v_orig = image_features[:, 0].clone().requires_grad_(True)   # cx.py:267 (leaf w/ grad: wasteful, kept)
        v_knns = image_features[:, 1:].clone().requires_grad_(True)
        a_emb_gt = self.answer_embedding(answer_aids)
        a_emb_knns = torch.bmm(F.softmax(a_knns, dim=-1),
                               self.answer_embedding.weight.view(1, d.A, d.da).expand(B, -1, -1))
        scores = []
        for i in range(d.K):
            v_other = v_knns[:, i]
            v_rank = torch.zeros(B, d.K); v_rank[:, i] = 1
            x = torch.cat((v_orig, v_other, v_orig * v_other,
                           F.pairwise_distance(v_orig, v_other, keepdim=True), v_rank, q_emb,
                           z_orig, z_knns[:, i], a_emb_gt, a_emb_knns[:, i]), dim=1)
            km = self.keep_masks if self.training else None
            drop = (lambda h, l: h * km[l].view(B, d.K, d.H)[:, i] / (1.0 - self.drop_p)) if km is not None else (lambda h, l: self.drop(h))
            h = drop(F.relu(self.linear_1(x)), 0)
            if d.L >= 2: h = drop(F.relu(self.linear_2(h)), 1)
            if d.L >= 3: h = drop(F.relu(self.linear_3(h)), 2)
            scores.append(self.out(h))
        return torch.cat(scores, dim=1)
