"""Model factory with the reference's signature (vqa/models/utils.py:14-30).  One process per GPU: the
reference's nn.DataParallel wrapper (immediately unwrapped by counterexamples.py:221-225) is not used;
data_parallel=True is accepted and ignored."""
import copy

from .noatt import MutanNoAtt

_REGISTRY = {"MutanNoAtt": MutanNoAtt}
model_names = sorted(_REGISTRY)


def factory(opt, vocab_words, vocab_answers, cuda=True, data_parallel=True):
    opt = copy.deepcopy(opt)
    if opt["arch"] not in _REGISTRY:
        raise ValueError("unknown VQA arch %r (available: %s)" % (opt["arch"], model_names))
    model = _REGISTRY[opt["arch"]](opt, vocab_words, vocab_answers)
    if cuda:
        model.cuda()
    return model
