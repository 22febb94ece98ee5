"""Host-side operators over the C ABI: torch tensors in, HIP kernels enqueued on torch's current stream.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every arithmetic step of the
hot path runs in libneuralcx_hip.so.  Reference surface these mirror:

  neuralcx_forward / NeuralCXFunction   vqa/models/cx.py:279-333 (NeuralModel.forward below vqa_forward)
  ranking_loss                           counterexamples.py:310,334 + recallAtK (counterexamples.py:501-506)
  adam_step                              torch.optim.Adam as used at counterexamples.py:275-276,339
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional

import torch

import os

from . import _lib
from ._lib import (NCX_F_A_EMB, NCX_F_ALL, NCX_F_V_DIST, NCX_F_V_MULT, NCX_F_V_RANK, NcxDims, NcxGrads,
                   NcxInputs, NcxMutanParams, NcxParams)

PARAM_FIELDS = ("answer_embedding", "w1", "b1", "w2", "b2", "w3", "b3", "w_out", "b_out")
# state_dict names of the reference (vqa/models/cx.py:240-257) -> C ABI field
STATE_TO_FIELD = {"answer_embedding.weight": "answer_embedding", "linear_1.weight": "w1", "linear_1.bias": "b1",
                  "linear_2.weight": "w2", "linear_2.bias": "b2", "linear_3.weight": "w3", "linear_3.bias": "b3",
                  "out.weight": "w_out", "out.bias": "b_out"}


def flags_from_spec(spec: Optional[dict]) -> int:
    """model_spec lesion switches (cx.py:265-307) -> NCX_F_* bits handled inside the kernels."""
    if spec is None:
        return NCX_F_ALL
    f = 0
    if spec.get("v_mult", True): f |= NCX_F_V_MULT
    if spec.get("v_dist", True): f |= NCX_F_V_DIST
    if spec.get("v_rank", True): f |= NCX_F_V_RANK
    if spec.get("a_emb", True): f |= NCX_F_A_EMB
    return f


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor], dtype, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.NcxError("%s must be a device tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.NcxError("%s must be contiguous %s, got %s%s" % (name, dtype, t.dtype, "" if t.is_contiguous() else " (strided)"))
    return C.c_void_p(t.data_ptr())


@dataclass
class Batch:
    """Device-resident inputs of one forward (ncx_inputs).  int32 indices, fp32 everything else."""
    feats: torch.Tensor            # [n_img, dv]
    img_idx: torch.Tensor          # [B, K+1] int32
    q_emb: torch.Tensor            # [B, dq]
    z_orig: torch.Tensor           # [B, dz]
    z_knns: torch.Tensor           # [B, K, dz]
    a_knns: torch.Tensor           # [B, K, A] logits (or [B, K, da] noise without a_emb)
    answer_aids: Optional[torch.Tensor] = None   # [B] int32
    a_emb_gt: Optional[torch.Tensor] = None      # lesion
    v_rank: Optional[torch.Tensor] = None        # lesion
    keep_mask: Optional[torch.Tensor] = None     # [L, B*K, H] explicit dropout masks (tests)

    @staticmethod
    def from_dense(image_features, q_emb, z_orig, z_knns, a_knns, answer_aids, **kw):
        """The reference hands NeuralModel a gathered [B, K+1, dv] block (counterexamples.py:540-541)."""
        B, K1, dv = image_features.shape
        feats = image_features.reshape(B * K1, dv).contiguous()
        idx = torch.arange(B * K1, device=feats.device, dtype=torch.int32).view(B, K1)
        aids = None if answer_aids is None else answer_aids.to(torch.int32).contiguous()
        return Batch(feats, idx, q_emb.contiguous(), z_orig.contiguous(), z_knns.contiguous(),
                     a_knns.contiguous(), aids, **kw)

    def c_struct(self) -> NcxInputs:
        s = NcxInputs()
        s.feats = _ptr(self.feats, torch.float32, "feats")
        s.img_idx = _ptr(self.img_idx, torch.int32, "img_idx")
        s.q_emb = _ptr(self.q_emb, torch.float32, "q_emb")
        s.z_orig = _ptr(self.z_orig, torch.float32, "z_orig")
        s.z_knns = _ptr(self.z_knns, torch.float32, "z_knns")
        s.a_knns = _ptr(self.a_knns, torch.float32, "a_knns")
        s.answer_aids = _ptr(self.answer_aids, torch.int32, "answer_aids")
        s.a_emb_gt = _ptr(self.a_emb_gt, torch.float32, "a_emb_gt")
        s.v_rank = _ptr(self.v_rank, torch.float32, "v_rank")
        s.keep_mask = _ptr(self.keep_mask, torch.float32, "keep_mask")
        return s


# Flag bits OR-ed into every ncx_dims built here.  NCX_X6=1 in the environment puts the whole process on the split-bf16 ("bf16 x 6") variant
# of the balanced TN weight-gradient launch (include/neuralcx.h: NCX_F_X6; not the default) -- how the unchanged GPU suite is run against it.
EXTRA_FLAGS = _lib.NCX_F_X6 if os.environ.get("NCX_X6", "0") not in ("", "0") else 0


def make_dims(batch: Batch, H: int, L: int, da: int, A: int, flags: int = NCX_F_ALL, training: bool = False,
              drop_p: float = 0.0, loss_scale: float = 0.0, seed: int = 0) -> NcxDims:
    B, K1 = batch.img_idx.shape
    d = NcxDims()
    d.B, d.K = B, K1 - 1
    d.dv, d.dq, d.dz = batch.feats.shape[1], batch.q_emb.shape[1], batch.z_orig.shape[1]
    d.da, d.A, d.H, d.L = da, A, H, L
    d.n_img = batch.feats.shape[0]
    d.flags, d.training, d.drop_p, d.loss_scale, d.seed = flags | EXTRA_FLAGS, int(training), float(drop_p), float(loss_scale), int(seed) & (2 ** 64 - 1)
    # shape validation before any launch (the reference's asserts: cx.py:65,263)
    K = d.K
    assert batch.z_knns.shape == (B, K, d.dz), batch.z_knns.shape
    assert batch.q_emb.shape[0] == B and batch.z_orig.shape == (B, d.dz)
    if flags & NCX_F_A_EMB:
        assert batch.a_knns.shape == (B, K, A), (batch.a_knns.shape, (B, K, A))
        assert batch.answer_aids is not None and batch.answer_aids.shape == (B,)
    else:
        assert batch.a_knns.shape == (B, K, da) and batch.a_emb_gt is not None and batch.a_emb_gt.shape == (B, da)
    if not (flags & NCX_F_V_RANK):
        assert batch.v_rank is not None and batch.v_rank.shape == (B, K, K)
    if batch.keep_mask is not None:
        assert batch.keep_mask.shape == (L, B * K, H)
    return d


def _params_struct(params: Dict[str, torch.Tensor], cls):
    s = cls()
    for f in PARAM_FIELDS:
        setattr(s, f, _ptr(params.get(f), torch.float32, f))
    return s


def workspace_bytes(d: NcxDims) -> int:
    n = _lib.lib().ncx_workspace_bytes(C.byref(d))
    if n == 0:
        raise _lib.NcxError("ncx_workspace_bytes: invalid dims")
    return n


def alloc_workspace(d: NcxDims, device) -> torch.Tensor:
    return torch.empty(workspace_bytes(d) + 256, dtype=torch.uint8, device=device)


def _ws_ptr(ws: torch.Tensor):
    base = ws.data_ptr()
    aligned = (base + 255) // 256 * 256
    return C.c_void_p(aligned), ws.numel() - (aligned - base)


def fused_tail_ok(d: NcxDims) -> bool:
    """Shapes ncx_train_tail takes (the K rows of a triplet live in registers, one lane per 4 columns)."""
    return d.K <= 32 and d.H <= 256


def train_tail(d: NcxDims, params: Dict[str, torch.Tensor], ws: torch.Tensor, scores: torch.Tensor, gt: torch.Tensor,
               grads: Dict[str, torch.Tensor], want_dscores: bool = False):
    """NCX_F_FUSED_TAIL: `out` + listwise loss / rank / Recall hits + the head of the backward in one pass (ncx_train_tail).
    Fills `scores` (the tensor ops.forward returned untouched) and d out.weight / d out.bias (/ d linear_1.bias)."""
    dev = scores.device
    loss_rows = torch.empty(d.B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dscores = torch.empty_like(scores) if want_dscores else None
    rank = torch.empty(d.B, dtype=torch.int32, device=dev)
    hits = torch.empty(2, dtype=torch.int32, device=dev)
    p, n = _ws_ptr(ws)
    ps, gs = _params_struct(params, NcxParams), _params_struct(grads, NcxGrads)
    _lib.check(_lib.lib().ncx_train_tail(C.byref(d), C.byref(ps), p, n, _ptr(gt, torch.int32, "gt"), C.c_void_p(scores.data_ptr()),
                                         C.c_void_p(loss_rows.data_ptr()), C.c_void_p(loss.data_ptr()),
                                         _ptr(dscores, torch.float32, "dscores"), C.c_void_p(rank.data_ptr()),
                                         C.c_void_p(hits.data_ptr()), C.byref(gs), _stream()), "ncx_train_tail")
    return dict(loss=loss, loss_rows=loss_rows, dscores=dscores, rank=rank, hits=hits)


FWD_ALL, FWD_PRELUDE, FWD_REST = 0, 1, 2


def forward(d: NcxDims, batch: Batch, params: Dict[str, torch.Tensor], ws: torch.Tensor, phase: int = FWD_ALL) -> Optional[torch.Tensor]:
    """-> scores [B, K] (with NCX_F_FUSED_TAIL in d.flags: allocated here, written by train_tail).
    phase (ncx_forward_phase): FWD_PRELUDE = the data-only part (returns None), FWD_REST = everything that reads the weights;
    PRELUDE then REST on the same workspace == the whole forward, bit for bit."""
    p, n = _ws_ptr(ws)
    ins, ps = batch.c_struct(), _params_struct(params, NcxParams)
    if phase == FWD_PRELUDE:
        _lib.check(_lib.lib().ncx_forward_phase(C.byref(d), C.byref(ins), C.byref(ps), p, n, None, FWD_PRELUDE, _stream()), "ncx_forward_phase")
        return None
    scores = torch.empty(d.B, d.K, dtype=torch.float32, device=batch.feats.device)
    if phase == FWD_ALL:
        _lib.check(_lib.lib().ncx_forward(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                          C.c_void_p(scores.data_ptr()), _stream()), "ncx_forward")
    else:
        _lib.check(_lib.lib().ncx_forward_phase(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                                C.c_void_p(scores.data_ptr()), int(phase), _stream()), "ncx_forward_phase")
    return scores


def backward(d: NcxDims, batch: Batch, params: Dict[str, torch.Tensor], ws: torch.Tensor, dscores: torch.Tensor,
             grads: Dict[str, torch.Tensor], phase: int = 0) -> None:
    """phase 0: whole backward.  phases 1 | 2, 3 | 4 and 5 | 2 | 4: the ways to cut it for comm overlap (ncx_backward_phase)."""
    p, n = _ws_ptr(ws)
    ins, ps, gs = batch.c_struct(), _params_struct(params, NcxParams), _params_struct(grads, NcxGrads)
    if phase == 0:
        _lib.check(_lib.lib().ncx_backward(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                           _ptr(dscores, torch.float32, "dscores"), C.byref(gs), _stream()), "ncx_backward")
    else:
        _lib.check(_lib.lib().ncx_backward_phase(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                                 _ptr(dscores, torch.float32, "dscores"), C.byref(gs), int(phase), _stream()),
                   "ncx_backward_phase")


def ws_dgt_view(d: NcxDims, ws: torch.Tensor) -> torch.Tensor:
    """fp32 view of the workspace block dGt | dGgt (2 x [H, A]) that phase 3 of backward leaves and phase 4 consumes:
    the bucket a data-parallel job sums over ranks instead of the [A, da] embedding gradient."""
    off, nbytes = C.c_size_t(0), C.c_size_t(0)
    _lib.check(_lib.lib().ncx_ws_region(C.byref(d), 1, C.byref(off), C.byref(nbytes)), "ncx_ws_region")
    base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
    return ws[base + off.value: base + off.value + nbytes.value].view(torch.float32)


def ranking_loss(scores: torch.Tensor, gt: torch.Tensor, scale: float = 0.0, want_grad: bool = True):
    """Listwise softmax-CE / B + rank of the ground truth + Recall@1/@5 hit counts in one pass."""
    B, K = scores.shape
    dev = scores.device
    loss_rows = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dscores = torch.empty_like(scores) if want_grad else None
    rank = torch.empty(B, dtype=torch.int32, device=dev)
    hits = torch.empty(2, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().ncx_loss_rank(_ptr(scores, torch.float32, "scores"), _ptr(gt, torch.int32, "gt"), B, K,
                                        float(scale), C.c_void_p(loss_rows.data_ptr()), C.c_void_p(loss.data_ptr()),
                                        _ptr(dscores, torch.float32, "dscores"), C.c_void_p(rank.data_ptr()),
                                        C.c_void_p(hits.data_ptr()), _stream()), "ncx_loss_rank")
    return dict(loss=loss, loss_rows=loss_rows, dscores=dscores, rank=rank, hits=hits)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    n = param.numel()
    assert grad.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n
    _lib.check(_lib.lib().ncx_adam_step(_ptr(param, torch.float32, "param"), _ptr(grad, torch.float32, "grad"),
                                        _ptr(exp_avg, torch.float32, "exp_avg"), _ptr(exp_avg_sq, torch.float32, "exp_avg_sq"),
                                        n, lr, betas[0], betas[1], eps, int(step), float(grad_scale), _stream()),
               "ncx_adam_step")


class MutanWeights:
    """Frozen MutanNoAtt parameters in the layout ncx_vqa_forward wants: the R rank-1 projections stacked
    (fusion.list_linear_hv.{i} -> [R*dim_mm, dim_hv]).  Built once per model; re-stack after loading a checkpoint."""

    def __init__(self, vqa_model):
        f, opt = vqa_model.fusion, vqa_model.opt["fusion"]
        for k in ("activation_hv", "activation_hq", "activation_mm"):
            if k in opt:
                raise _lib.NcxError("ncx_vqa_forward supports the options/cx/*.yaml MUTAN (no %s)" % k)
        if "activation" in vqa_model.opt.get("classif", {}):
            raise _lib.NcxError("ncx_vqa_forward: classif.activation is not supported")
        act = {None: 0, "tanh": 2}
        if opt.get("activation_v") not in act or opt.get("activation_q") not in act:
            raise _lib.NcxError("ncx_vqa_forward supports activation_v/q in {none, tanh}")
        c = lambda t: t.detach().float().contiguous()
        self.t = dict(wv=c(f.linear_v.weight), bv=c(f.linear_v.bias), wq=c(f.linear_q.weight), bq=c(f.linear_q.bias),
                      whv=c(torch.cat([l.weight for l in f.list_linear_hv])), bhv=c(torch.cat([l.bias for l in f.list_linear_hv])),
                      whq=c(torch.cat([l.weight for l in f.list_linear_hq])), bhq=c(torch.cat([l.bias for l in f.list_linear_hq])),
                      wc=c(vqa_model.linear_classif.weight), bc=c(vqa_model.linear_classif.bias))
        self.dhv, self.dhq, self.R, self.dz = opt["dim_hv"], opt["dim_hq"], opt["R"], opt["dim_mm"]
        self.A = self.t["wc"].shape[0]
        self.act_v, self.act_q = act[opt.get("activation_v")], act[opt.get("activation_q")]

    def c_struct(self):
        m = NcxMutanParams()
        for k, v in self.t.items():
            setattr(m, k, _ptr(v, torch.float32, k))
        m.dhv, m.dhq, m.R, m.act_v, m.act_q = self.dhv, self.dhq, self.R, self.act_v, self.act_q
        return m


def vqa_forward(feats: torch.Tensor, img_idx: torch.Tensor, q_emb: torch.Tensor, mw: MutanWeights, want_a_orig=False, ws=None):
    """HIP replacement of CXModelBase.vqa_forward below the question encoder (cx.py:64-104).
    -> (a_orig or None, z_orig [B,dz], a_knns [B,K,A], z_knns [B,K,dz])."""
    B, K1 = img_idx.shape
    d = NcxDims()
    d.B, d.K, d.dv, d.dq, d.dz, d.da, d.A, d.H, d.L = B, K1 - 1, feats.shape[1], q_emb.shape[1], mw.dz, 4, mw.A, 4, 1
    d.n_img = feats.shape[0]
    m = mw.c_struct()
    need = _lib.lib().ncx_vqa_workspace_bytes(C.byref(d), C.byref(m))
    if need == 0:
        raise _lib.NcxError("ncx_vqa_workspace_bytes: invalid dims")
    if ws is None or ws.numel() < need + 256:
        ws = torch.empty(need + 256, dtype=torch.uint8, device=feats.device)
    dev = feats.device
    z_o = torch.empty(B, mw.dz, device=dev); z_k = torch.empty(B, K1 - 1, mw.dz, device=dev)
    a_k = torch.empty(B, K1 - 1, mw.A, device=dev)
    a_o = torch.empty(B, mw.A, device=dev) if want_a_orig else None
    p, n = _ws_ptr(ws)
    _lib.check(_lib.lib().ncx_vqa_forward(C.byref(d), _ptr(feats, torch.float32, "feats"), _ptr(img_idx, torch.int32, "img_idx"),
                                          _ptr(q_emb, torch.float32, "q_emb"), C.byref(m), p, n, C.c_void_p(z_o.data_ptr()),
                                          C.c_void_p(z_k.data_ptr()), C.c_void_p(a_k.data_ptr()),
                                          C.c_void_p(a_o.data_ptr()) if a_o is not None else None, _stream()), "ncx_vqa_forward")
    return a_o, z_o, a_k, z_k


class WorkspacePool:
    """Workspaces (saved activations + scratch of ncx_forward / ncx_backward) of one module.  A workspace is OWNED by the
    call that took it until that call's backward has been enqueued (or, for a call that needs no gradient, until its
    forward has been enqueued -- the stream orders the reuse); only then does it return to the pool.  Two forwards
    before a backward therefore never share saved activations (the reference module has no such limit either:
    counterexamples.py:357-361 evaluates inside the train loop)."""

    def __init__(self):
        self.free = []

    def take(self, nbytes: int, device) -> torch.Tensor:
        for i, w in enumerate(self.free):
            if w.numel() >= nbytes and w.device == device:
                return self.free.pop(i)
        self.free.clear()                       # (sizes changed: drop the stale buffers)
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def give(self, ws: torch.Tensor) -> None:
        if len(self.free) < 2:
            self.free.append(ws)


class NeuralCXFunction(torch.autograd.Function):
    """scores = NeuralCX(batch; params) with the hand-written backward (ncx_backward).
    `call` = {dims, batch, names, pool, record}: a per-call snapshot -- backward reads the dims / inputs / workspace of ITS
    forward from ctx, never from shared module state."""

    @staticmethod
    def forward(ctx, call, *param_tensors):
        d, batch, names, pool = call["dims"], call["batch"], call["names"], call["pool"]
        params = dict(zip(names, param_tensors))
        ws = pool.take(workspace_bytes(d) + 256, batch.feats.device)
        scores = forward(d, batch, params, ws)
        if call["record"]:                      # (ctx.needs_input_grad stays True under torch.no_grad: the caller tells)
            ctx.call, ctx.ws = dict(call), ws
            ctx.save_for_backward(*param_tensors)
        else:
            pool.give(ws)                       # no graph is recorded (torch.no_grad / frozen parameters)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        call = ctx.call
        if ctx.ws is None:
            raise RuntimeError("NeuralCXFunction: backward ran twice on one forward (retain_graph is not supported: "
                               "the saved activations live in a workspace that has been released)")
        d, batch, names = call["dims"], call["batch"], call["names"]
        params = dict(zip(names, ctx.saved_tensors))
        gbuf = {n: torch.empty_like(t) for n, t in params.items()}
        backward(d, batch, params, ctx.ws, dscores.contiguous(), gbuf)
        call["pool"].give(ctx.ws)
        ctx.ws = None
        return (None,) + tuple(gbuf[n] for n in names)
