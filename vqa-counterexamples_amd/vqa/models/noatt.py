"""No-attention VQA model used as the frozen feature producer of NeuralCX (reference: vqa/models/noatt.py:9-58).
Exposes what vqa.models.cx needs: .seq2vec, ._fusion(v, q), ._classif(z), .opt['fusion'], .vocab_answers."""
import torch.nn as nn
import torch.nn.functional as F

from . import fusion, seq2vec


class MutanNoAtt(nn.Module):
    def __init__(self, opt=None, vocab_words=(), vocab_answers=()):
        super().__init__()
        opt = opt or {}
        opt["fusion"]["dim_h"] = opt["fusion"]["dim_mm"]
        self.opt, self.vocab_words, self.vocab_answers = opt, vocab_words, vocab_answers
        self.num_classes = len(vocab_answers)
        self.seq2vec = seq2vec.factory(vocab_words, opt["seq2vec"], dim_q=opt["fusion"]["dim_q"])
        self.linear_classif = nn.Linear(opt["fusion"]["dim_h"], self.num_classes)
        self.fusion = fusion.MutanFusion(opt["fusion"])

    def _fusion(self, input_v, input_q):
        return self.fusion(input_v, input_q)

    def _classif(self, x):
        if "activation" in self.opt["classif"]:
            x = getattr(F, self.opt["classif"]["activation"])(x)
        x = F.dropout(x, p=self.opt["classif"]["dropout"], training=self.training)
        return self.linear_classif(x)

    def forward(self, input_v, input_q):
        return self._classif(self._fusion(input_v, self.seq2vec(input_q)))
