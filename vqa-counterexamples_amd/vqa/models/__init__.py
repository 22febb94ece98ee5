"""Drop-in `vqa.models` package of the MI355X NeuralCX path: the VQA-model plugin surface the counterexample scorer needs
(`factory(opt, vocab_words, vocab_answers, cuda, data_parallel)`, `model_names`, the MUTAN no-attention model) plus, in
`vqa.models.cx`, the NeuralModel / baseline scorers whose hot path runs in libneuralcx_hip.so."""
from . import cx, fusion, noatt, seq2vec, utils

MutanNoAtt = noatt.MutanNoAtt
factory = utils.factory
model_names = utils.model_names

__all__ = ["MutanNoAtt", "factory", "model_names", "cx", "fusion", "noatt", "seq2vec", "utils"]
