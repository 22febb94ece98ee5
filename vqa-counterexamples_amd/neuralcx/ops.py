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

from . import _lib
from ._lib import (NCX_F_A_EMB, NCX_F_ALL, NCX_F_V_DIST, NCX_F_V_MULT, NCX_F_V_RANK, NcxDims, NcxGrads,
                   NcxInputs, NcxParams)

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


def make_dims(batch: Batch, H: int, L: int, da: int, A: int, flags: int = NCX_F_ALL, training: bool = False,
              drop_p: float = 0.0, loss_scale: float = 0.0, seed: int = 0) -> NcxDims:
    B, K1 = batch.img_idx.shape
    d = NcxDims()
    d.B, d.K = B, K1 - 1
    d.dv, d.dq, d.dz = batch.feats.shape[1], batch.q_emb.shape[1], batch.z_orig.shape[1]
    d.da, d.A, d.H, d.L = da, A, H, L
    d.n_img = batch.feats.shape[0]
    d.flags, d.training, d.drop_p, d.loss_scale, d.seed = flags, int(training), float(drop_p), float(loss_scale), int(seed) & (2 ** 64 - 1)
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


def forward(d: NcxDims, batch: Batch, params: Dict[str, torch.Tensor], ws: torch.Tensor) -> torch.Tensor:
    scores = torch.empty(d.B, d.K, dtype=torch.float32, device=batch.feats.device)
    p, n = _ws_ptr(ws)
    ins, ps = batch.c_struct(), _params_struct(params, NcxParams)
    _lib.check(_lib.lib().ncx_forward(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                      C.c_void_p(scores.data_ptr()), _stream()), "ncx_forward")
    return scores


def backward(d: NcxDims, batch: Batch, params: Dict[str, torch.Tensor], ws: torch.Tensor, dscores: torch.Tensor,
             grads: Dict[str, torch.Tensor], phase: int = 0) -> None:
    """phase 0: whole backward.  phase 1 / 2: first / second half (see ncx_backward_phase) for comm overlap."""
    p, n = _ws_ptr(ws)
    ins, ps, gs = batch.c_struct(), _params_struct(params, NcxParams), _params_struct(grads, NcxGrads)
    if phase == 0:
        _lib.check(_lib.lib().ncx_backward(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                           _ptr(dscores, torch.float32, "dscores"), C.byref(gs), _stream()), "ncx_backward")
    else:
        _lib.check(_lib.lib().ncx_backward_phase(C.byref(d), C.byref(ins), C.byref(ps), p, n,
                                                 _ptr(dscores, torch.float32, "dscores"), C.byref(gs), int(phase), _stream()),
                   "ncx_backward_phase")


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


class NeuralCXFunction(torch.autograd.Function):
    """scores = NeuralCX(batch; params) with the hand-written backward (ncx_backward)."""

    @staticmethod
    def forward(ctx, holder, *param_tensors):
        d, batch, names = holder["dims"], holder["batch"], holder["names"]
        params = dict(zip(names, param_tensors))
        ws = holder.get("workspace")
        need = workspace_bytes(d) + 256
        if ws is None or ws.numel() < need or ws.device != batch.feats.device:
            ws = torch.empty(need, dtype=torch.uint8, device=batch.feats.device)
            holder["workspace"] = ws
        scores = forward(d, batch, params, ws)
        ctx.holder, ctx.ws = holder, ws
        ctx.save_for_backward(*param_tensors)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        holder = ctx.holder
        d, batch, names = holder["dims"], holder["batch"], holder["names"]
        params = dict(zip(names, ctx.saved_tensors))
        gbuf = holder.get("grad_buffers")
        if gbuf is None:
            gbuf = {n: torch.empty_like(t) for n, t in params.items()}
        backward(d, batch, params, ctx.ws, dscores.contiguous(), gbuf)
        return (None,) + tuple(gbuf[n] for n in names)
