"""Question encoders.  The reference uses skipthoughts.BayesianUniSkip (GRU 620 -> 2400) from an un-vendored
submodule that needs downloaded tables (vqa/models/seq2vec.py:79-85); offline we provide a GRU encoder with the
same interface (wids[B, T] right-padded with 0 -> [B, dim_q]).  It is an INPUT producer of the hot path."""
import torch
import torch.nn as nn


class GRUEncoder(nn.Module):
    def __init__(self, vocab_words, dim_q=2400, dim_emb=620, dropout=0.25):
        super().__init__()
        self.embedding = nn.Embedding(len(vocab_words) + 1, dim_emb, padding_idx=0)
        self.gru = nn.GRU(dim_emb, dim_q, batch_first=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, wids):
        x = self.embedding(wids)
        out, _ = self.gru(x)
        last = (wids > 0).sum(1).clamp(min=1) - 1            # last valid step (right padding)
        return self.dropout(out[torch.arange(wids.shape[0], device=wids.device), last])


def factory(vocab_words, opt, dim_q=2400):
    arch = opt.get("arch", "skipthoughts")
    if arch == "skipthoughts":
        try:
            import skipthoughts                      # optional: the real encoder when the submodule is present
            return getattr(skipthoughts, opt["type"])(opt["dir_st"], vocab_words, dropout=opt["dropout"],
                                                      fixed_emb=opt["fixed_emb"])
        except ImportError:
            return GRUEncoder(vocab_words, dim_q=dim_q, dropout=opt.get("dropout", 0.25))
    if arch in ("gru", "lstm", "2-lstm"):
        return GRUEncoder(vocab_words, dim_q=dim_q, dim_emb=opt.get("emb_size", 620), dropout=opt.get("dropout", 0.0))
    raise NotImplementedError(arch)
