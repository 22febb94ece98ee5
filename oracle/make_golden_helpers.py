"""Generate tests/golden/g7_data_helpers.npz by RUNNING the reference's counterexamples.py helpers (build container only).

`recallAtK` (counterexamples.py:501-506), `batchify` (:509-516) and `getDataFromBatch` (:519-547) are imported from
/root/reference and run unmodified on a seeded synthetic example list; inputs and outputs are stored as arrays.  The
module imports only with stub modules for packages that are absent here (h5py, click, tqdm if missing, tensorboard,
torchvision.*, skipthoughts) and for reference-internal modules that cannot be imported on Python 3.10 / offline
(vqa.lib.engine: `async=True` SyntaxError; vqa.datasets: nltk/h5py + download side effects; train; cx_visu), and with
`.cuda()` -> identity (no GPU in the container).  The reference files are never edited or copied; this script hard-fails
when /root/reference is absent.  Usage: python oracle/make_golden_helpers.py
"""
import importlib
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
if not os.path.isfile(os.path.join(REF, "counterexamples.py")):
    raise SystemExit("make_golden_helpers.py: /root/reference is not mounted; fixtures can only be generated in the build container")

import torch
import torch.nn as nn

sys.path.insert(0, REF)


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


for mod in ("h5py", "click", "tqdm"):
    try:
        importlib.import_module(mod)
    except ImportError:
        _stub(mod, tqdm=lambda x, **k: x)
_stub("tensorboard", SummaryWriter=object)
_tv = _stub("torchvision")
_tv.models = _stub("torchvision.models"); _tv.transforms = _stub("torchvision.transforms"); _tv.utils = _stub("torchvision.utils")
_stub("skipthoughts", BayesianUniSkip=nn.Module)
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self
import vqa                                             # noqa: E402  (reference package)
import vqa.lib                                         # noqa: E402
sys.modules["vqa.lib.engine"] = _stub("vqa.lib.engine"); vqa.lib.engine = sys.modules["vqa.lib.engine"]
sys.modules["vqa.datasets"] = _stub("vqa.datasets"); vqa.datasets = sys.modules["vqa.datasets"]
_stub("train", load_checkpoint=lambda *a, **k: None)
_stub("cx_visu", viz_knns=None, viz_qa=None)
import counterexamples as ref_cx                       # noqa: E402  (the reference script, imported as a module)


def main():
    rng = np.random.default_rng(77)
    K, T, n_img, dv, n_ex, B = 24, 26, 90, 40, 37, 8
    names = ["img_%04d" % i for i in range(n_img)]
    perm = rng.permutation(n_img)
    name_to_index = {names[i]: int(perm[i]) for i in range(n_img)}
    features = rng.standard_normal((n_img, dv)).astype(np.float32)
    rows = np.stack([rng.choice(n_img, size=K + 1, replace=False) for _ in range(n_ex)])          # name ids
    wids = rng.integers(0, 50, size=(n_ex, T)).astype(np.int64)
    aids = rng.integers(0, 2000, size=n_ex).astype(np.int64)
    comps = rng.integers(0, K, size=n_ex).astype(np.int64)
    examples = [{"image_name": names[rows[i, 0]], "knns": [names[r] for r in rows[i, 1:]], "comp": {"knn_index": int(comps[i])},
                 "question_wids": [int(w) for w in wids[i]], "answer_aid": int(aids[i]), "uid": i} for i in range(n_ex)]
    out = dict(name_ids=rows.astype(np.int32), name_to_index=np.array([name_to_index[n] for n in names], np.int32),
               features=features, question_wids=wids, answer_aids=aids, comp_idxs=comps)

    # batchify: in-place shuffle under random.seed(42) (the reference's seed, counterexamples.py:119), slices of B
    random.seed(42)
    lst = list(examples)
    batches = ref_cx.batchify(lst, B)
    out["batchify_order"] = np.array([ex["uid"] for b in batches for ex in b], np.int32)
    out["batchify_sizes"] = np.array([len(b) for b in batches], np.int32)
    out["batchify_inplace_order"] = np.array([ex["uid"] for ex in lst], np.int32)
    noshuf = ref_cx.batchify(list(examples), B, shuffle=False)
    out["batchify_noshuffle_sizes"] = np.array([len(b) for b in noshuf], np.int32)

    # getDataFromBatch on the first and the last (partial) batch of the unshuffled list
    for tag, b in (("first", noshuf[0]), ("last", noshuf[-1])):
        f, w, a, c = ref_cx.getDataFromBatch(b, features, name_to_index)
        out["gdfb_%s_uids" % tag] = np.array([ex["uid"] for ex in b], np.int32)
        out["gdfb_%s_features" % tag] = f.numpy().copy()
        out["gdfb_%s_wids" % tag] = w.numpy().copy()
        out["gdfb_%s_aids" % tag] = a.numpy().copy()
        out["gdfb_%s_comp" % tag] = c.data.numpy().copy()
    # pairwise variant: [comp, random other] under a fixed seed
    random.seed(7)
    f, w, a, c = ref_cx.getDataFromBatch(noshuf[0], features, name_to_index, pairwise=True)
    out["gdfb_pairwise_features"] = f.numpy().copy()

    # recallAtK: the real function, ties-free random scores + DistanceBaseline rows
    s = (rng.standard_normal((96, K)) * 3).astype(np.float32)
    g = rng.integers(0, K, size=96).astype(np.int64)
    out["recall_scores"], out["recall_gt"] = s, g
    for k in (1, 5, 24):
        out["recall_at_%d" % k] = np.asarray(ref_cx.recallAtK(torch.from_numpy(s), torch.from_numpy(g), k=k)).astype(np.int32)
    d = np.tile(np.arange(K - 1, -1, -1, dtype=np.float32), (32, 1))
    gd = rng.integers(0, K, size=32).astype(np.int64)
    out["dist_scores"], out["dist_gt"] = d, gd
    out["dist_recall_at_5"] = np.asarray(ref_cx.recallAtK(torch.from_numpy(d), torch.from_numpy(gd), k=5)).astype(np.int32)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "g7_data_helpers.npz")
    np.savez_compressed(path, **out)
    print("g7_data_helpers written: %.1f KB, recall@5 %d/96" % (os.path.getsize(path) / 1024, out["recall_at_5"].sum()))


if __name__ == "__main__":
    main()
