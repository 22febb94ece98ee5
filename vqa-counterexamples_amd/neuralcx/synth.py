"""Synthetic VQA-CX data of the shapes BASELINE.json names (there are no datasets or checkpoints offline).

Mirrors what the reference's data path hands to the model (counterexamples.py:519-547 + vqa_forward,
cx.py:64-104): a resident image-feature table [n_img, 2048] (`|N(0,1)| * 0.45`, ResNet post-ReLU pooled
scale; COCO-train size 82 783 rows), and per triplet: the original image + 24 candidate rows, a question
embedding, MUTAN fusion vectors z, answer logits a for each candidate, the ground-truth answer id and the
index of the true counterexample among the candidates.

The ground truth is PLANTED so that ranking is learnable: the probability of candidate k being the
counterexample falls with its feature distance to the original image and with its neighbour rank (as in
the real data, where the distance baseline reaches R@5 = 44 %: README.md:39 of the reference).
"""
from typing import Optional

import numpy as np
import torch

from .ops import Batch


class SyntheticCX:
    def __init__(self, n_triplets=16384, K=24, dv=2048, dq=2400, dz=360, A=2000, n_img=82783, seed=1234,
                 device="cuda:0", feats: Optional[torch.Tensor] = None):
        self.N, self.K, self.dv, self.dq, self.dz, self.A, self.n_img = n_triplets, K, dv, dq, dz, A, n_img
        self.device = torch.device(device)
        g = torch.Generator(device="cpu").manual_seed(seed)
        rng = np.random.default_rng(seed)
        if feats is None:
            feats = torch.empty(n_img, dv)
            chunk = 8192
            for i in range(0, n_img, chunk):
                feats[i:i + chunk] = torch.randn(min(chunk, n_img - i), dv, generator=g).abs_() * 0.45
            feats = feats.to(self.device)
        self.feats = feats
        # original + K distinct other rows per triplet
        idx = rng.integers(0, n_img, size=(n_triplets, K + 1), dtype=np.int64)
        self.img_idx = torch.from_numpy(idx.astype(np.int32))
        self.answer_aids = torch.from_numpy(rng.integers(0, A, size=n_triplets).astype(np.int32))
        self.seed = seed
        # planted ground truth: p(k) ~ exp(-1.5 * standardised distance - 0.08 * k)
        gt = np.empty(n_triplets, np.int64)
        f_cpu = None
        step = 2048
        for i in range(0, n_triplets, step):
            ii = self.img_idx[i:i + step].to(self.device).long()
            vo = self.feats[ii[:, 0]]
            vk = self.feats[ii[:, 1:].reshape(-1)].view(ii.shape[0], K, dv)
            dist = (vo[:, None, :] - vk).norm(dim=2)
            zs = (dist - dist.mean(1, keepdim=True)) / (dist.std(1, keepdim=True) + 1e-6)
            logit = (-1.5 * zs - 0.08 * torch.arange(K, device=self.device)[None, :]).cpu().numpy()
            p = np.exp(logit - logit.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
            u = rng.random(p.shape[0])[:, None]
            gt[i:i + step] = (p.cumsum(1) < u).sum(1).clip(0, K - 1)
        self.gt = torch.from_numpy(gt.astype(np.int32)).to(self.device)
        self.img_idx = self.img_idx.to(self.device)                    # index arrays live on the device: a batch is
        self.answer_aids = self.answer_aids.to(self.device)            # three index_selects, no host-to-device copy
        self._gen = torch.Generator(device=self.device)

    def batch(self, sel: torch.Tensor, first_id: Optional[int] = None) -> (Batch, torch.Tensor):
        """sel: int64 triplet ids, on the device (no host-to-device copy per step) or on the CPU.  q/z/a blocks are
        generated on device from a per-batch seed (deterministic for a given first id and size) so a 440 k-triplet set
        never has to be resident (98 MB of logits per 512).  first_id: the first id as a Python int (avoids reading it
        back from a device tensor)."""
        B, K = sel.numel(), self.K
        if first_id is None:
            first_id = int(sel[0])
        sel = sel.to(self.device)
        g = self._gen
        g.manual_seed(self.seed * 1000003 + int(first_id) * 7919 + B)
        q = torch.randn(B, self.dq, generator=g, device=self.device) * 0.3
        z_o = torch.randn(B, self.dz, generator=g, device=self.device)
        z_k = torch.randn(B, K, self.dz, generator=g, device=self.device)
        a_k = torch.randn(B, K, self.A, generator=g, device=self.device) * 2.0
        b = Batch(self.feats, self.img_idx.index_select(0, sel), q, z_o, z_k, a_k, self.answer_aids.index_select(0, sel))
        return b, self.gt.index_select(0, sel)
