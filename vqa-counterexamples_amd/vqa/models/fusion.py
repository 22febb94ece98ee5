"""MUTAN fusion (Ben-younes et al.), the producer of the multimodal vector z consumed by NeuralCX.
Same arithmetic and state_dict keys as the reference's MutanFusion (vqa/models/fusion.py:53-121):
tanh(linear_v(v)), tanh(linear_q(q)), R rank-1 terms linear_hv_i(x_v) * linear_hq_i(x_q), summed.
Plain PyTorch-ROCm (hipBLASLt under torch): SURVEY 8 marks it "next" (f1), not part of the HIP hot path."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MutanFusion(nn.Module):
    def __init__(self, opt, visual_embedding=True, question_embedding=True):
        super().__init__()
        self.opt = opt
        self.visual_embedding, self.question_embedding = visual_embedding, question_embedding
        if visual_embedding:
            self.linear_v = nn.Linear(opt["dim_v"], opt["dim_hv"])
        if question_embedding:
            self.linear_q = nn.Linear(opt["dim_q"], opt["dim_hq"])
        self.list_linear_hv = nn.ModuleList([nn.Linear(opt["dim_hv"], opt["dim_mm"]) for _ in range(opt["R"])])
        self.list_linear_hq = nn.ModuleList([nn.Linear(opt["dim_hq"], opt["dim_mm"]) for _ in range(opt["R"])])

    def _act(self, x, key):
        return getattr(torch, self.opt[key])(x) if key in self.opt else x

    def embed_v(self, v):
        if not self.visual_embedding:
            return v
        v = F.dropout(v, p=self.opt["dropout_v"], training=self.training)
        return self._act(self.linear_v(v), "activation_v")

    def embed_q(self, q):
        if not self.question_embedding:
            return q
        q = F.dropout(q, p=self.opt["dropout_q"], training=self.training)
        return self._act(self.linear_q(q), "activation_q")

    def forward(self, input_v, input_q):
        if input_v.dim() != 2 or input_q.dim() != 2:
            raise ValueError("MutanFusion expects 2-D inputs")
        x_v, x_q = self.embed_v(input_v), self.embed_q(input_q)
        x_mm = None
        for lin_v, lin_q in zip(self.list_linear_hv, self.list_linear_hq):
            hv = self._act(lin_v(F.dropout(x_v, p=self.opt["dropout_hv"], training=self.training)), "activation_hv")
            hq = self._act(lin_q(F.dropout(x_q, p=self.opt["dropout_hq"], training=self.training)), "activation_hq")
            x_mm = hq * hv if x_mm is None else x_mm + hq * hv
        return self._act(x_mm, "activation_mm")
