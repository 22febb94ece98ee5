"""On-disk formats of the reference's CX data path -> device-resident tables.

What the reference reads every run (counterexamples.py:181-207, 250-253) and how this package holds it:

  pickle_old/{trainset_augmented[_small],valset_augmented[_small]}.pickle
        dict: 'examples_list' (list of {'image_name', 'knns': [24 names], 'comp': {'knn_index', ...},
        'question_wids': [26 ints], 'answer_aid': int}), 'name_to_index' {image name -> feature row},
        'vocab_words', 'vocab_answers'                      -> int32/int64 index arrays on the GPU (CXDeviceDataset)
  {trainset,valset}.hdf5['noatt']  [N_img, 2048] f32 (+ .txt with the image-name order, extract.py:90-150)
        -> one resident fp32 table per split; `.npy` files of the same arrays are accepted directly, hdf5 needs
        h5py (absent in the offline image: `convert_hdf5_features` is the one-time conversion to run where it is)
  answer_embedding.pickle  [2000, 2400] float           -> initial value of answer_embedding.weight

The reference re-does a Python loop over the batch, a numpy gather of [B, 25, 2048] and a 105 MB host-to-device
copy every step (getDataFromBatch, counterexamples.py:519-547); here the name->row lookups are done once, the
tables live in HBM and a batch is three index_selects of a few KB: the feature gather itself happens inside the
kernels (ncx_forward / ncx_vqa_forward take the table + row indices).
"""
import os
import pickle
from typing import Dict, Optional, Sequence

import numpy as np
import torch

DATASET_KEYS = ("examples_list", "name_to_index", "vocab_words", "vocab_answers")
EXAMPLE_KEYS = ("image_name", "knns", "comp", "question_wids", "answer_aid")


def load_cx_pickle(path: str) -> dict:
    """One of the *_augmented*.pickle files; checks the keys the CX script uses (counterexamples.py:222-223,312,526-537)."""
    with open(path, "rb") as f:
        data = pickle.load(f)
    missing = [k for k in DATASET_KEYS if k not in data]
    if missing:
        raise KeyError("%s: missing dataset keys %s" % (path, missing))
    if len(data["examples_list"]):
        ex = data["examples_list"][0]
        missing = [k for k in EXAMPLE_KEYS if k not in ex]
        if missing:
            raise KeyError("%s: examples lack keys %s" % (path, missing))
    return data


def convert_hdf5_features(hdf5_path: str, npy_path: str, dataset: str = "noatt", chunk_rows: int = 8192) -> tuple:
    """One-time hdf5 -> npy conversion of a feature table, streamed in row chunks (the train table is 678 MB)."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("convert_hdf5_features needs h5py (not installed here); run the conversion where the "
                          "reference's extract.py ran, or provide <split>set.npy") from e
    with h5py.File(hdf5_path, "r") as f:
        src = f[dataset]
        out = np.lib.format.open_memmap(npy_path, mode="w+", dtype=np.float32, shape=tuple(src.shape))
        for i in range(0, src.shape[0], chunk_rows):
            out[i:i + chunk_rows] = src[i:i + chunk_rows]
        out.flush()
        return tuple(src.shape)


def load_feature_table(features_dir: str, split: str) -> np.ndarray:
    """`<dir>/<split>set.npy` (memory-mapped) or, with h5py, `<dir>/<split>set.hdf5['noatt']` (counterexamples.py:199-207)."""
    npy = os.path.join(features_dir, "%sset.npy" % split)
    if os.path.isfile(npy):
        t = np.load(npy, mmap_mode="r")
    else:
        h5 = os.path.join(features_dir, "%sset.hdf5" % split)
        if not os.path.isfile(h5):
            raise FileNotFoundError("no feature table %s or %s" % (npy, h5))
        try:
            import h5py
        except ImportError as e:
            raise ImportError("%s needs h5py; convert it once with neuralcx.formats.convert_hdf5_features" % h5) from e
        t = np.asarray(h5py.File(h5, "r").get("noatt"))
    if t.ndim != 2 or t.dtype != np.float32:
        raise ValueError("feature table must be a [N_img, dim_v] float32 array, got %s %s" % (t.shape, t.dtype))
    return t


def read_name_order(txt_path: str) -> list:
    """The .txt next to each hdf5: one image name per line, row order of the table (extract.py:139-150)."""
    with open(txt_path) as f:
        return [line.strip() for line in f if line.strip()]


def load_answer_embedding(path: str, n_answers: Optional[int] = None, dim_a: int = 2400) -> np.ndarray:
    with open(path, "rb") as f:
        emb = np.asarray(pickle.load(f), dtype=np.float32)
    if emb.ndim != 2 or emb.shape[1] != dim_a or (n_answers is not None and emb.shape[0] != n_answers):
        raise ValueError("answer embedding must be [%s, %d], got %s" % (n_answers or "A", dim_a, emb.shape))
    return emb


def examples_to_index_arrays(examples: Sequence[dict], name_to_index: Dict[str, int], knn_size: int = 24,
                             n_answers: Optional[int] = None):
    """Whole example list -> (img_idx int32 [N, K+1], question_wids int64 [N, T], answer_aids int32 [N],
    comp_idxs int32 [N]): the per-batch lookups of getDataFromBatch (counterexamples.py:525-537) done once.
    n_answers: when given, answer_aid is range-checked here (the kernels gather / scatter answer_embedding rows by it;
    the reference's nn.Embedding raises an index error for a bad id, cx.py:280)."""
    n = len(examples)
    img_idx = np.empty((n, knn_size + 1), np.int32)
    aids = np.empty(n, np.int32)
    comps = np.empty(n, np.int32)
    T = len(examples[0]["question_wids"]) if n else 0
    wids = np.zeros((n, T), np.int64)
    for i, ex in enumerate(examples):
        knns = ex["knns"]
        if len(knns) != knn_size:
            raise ValueError("example %d has %d neighbours, expected %d" % (i, len(knns), knn_size))
        img_idx[i, 0] = name_to_index[ex["image_name"]]
        img_idx[i, 1:] = [name_to_index[k] for k in knns]
        w = ex["question_wids"]
        if len(w) != T:
            raise ValueError("example %d: question_wids has length %d, expected %d" % (i, len(w), T))
        wids[i] = w
        aids[i] = ex["answer_aid"]
        comps[i] = ex["comp"]["knn_index"]
    if n and (comps.min() < 0 or comps.max() >= knn_size):
        raise ValueError("comp.knn_index outside [0, %d)" % knn_size)
    if n and n_answers is not None and (aids.min() < 0 or aids.max() >= n_answers):
        raise IndexError("answer_aid outside [0, %d) (example %d)" % (n_answers, int(np.argmax((aids < 0) | (aids >= n_answers)))))
    return img_idx, wids, aids, comps


class CXDeviceDataset:
    """A CX split resident on one device: feature table + index arrays; `batch_indices(sel)` is three index_selects."""

    def __init__(self, data: dict, features: np.ndarray, device="cuda:0", knn_size: int = 24,
                 feats: Optional[torch.Tensor] = None, upload_rows: int = 16384):
        self.device = torch.device(device)
        self.K = knn_size
        self.vocab_words, self.vocab_answers = data["vocab_words"], data["vocab_answers"]
        img_idx, wids, aids, comps = examples_to_index_arrays(data["examples_list"], data["name_to_index"], knn_size,
                                                              n_answers=len(self.vocab_answers) or None)
        self.N = img_idx.shape[0]
        n_rows = features.shape[0] if feats is None else feats.shape[0]
        if self.N and (img_idx.min() < 0 or img_idx.max() >= n_rows):
            raise IndexError("name_to_index points outside the feature table (%d rows)" % n_rows)
        if feats is None:
            feats = torch.empty(features.shape, dtype=torch.float32, device=self.device)
            for i in range(0, n_rows, upload_rows):          # streamed: never a second whole host copy of a memmap
                feats[i:i + upload_rows] = torch.from_numpy(np.array(features[i:i + upload_rows], dtype=np.float32)).to(self.device)
        self.feats = feats
        self.img_idx = torch.from_numpy(img_idx).to(self.device)
        self.question_wids = torch.from_numpy(wids).to(self.device)
        self.answer_aids = torch.from_numpy(aids).to(self.device)
        self.gt = torch.from_numpy(comps).to(self.device)
        self.vqa_cache = None

    def batch_indices(self, sel: torch.Tensor):
        """sel: int64 ids (CPU or device) -> img_idx [B, K+1] i32, question_wids [B, T] i64, answer_aids [B] i32, gt [B] i32."""
        sel = sel.to(self.device, non_blocking=True)
        return (self.img_idx.index_select(0, sel), self.question_wids.index_select(0, sel),
                self.answer_aids.index_select(0, sel), self.gt.index_select(0, sel))

    def cache_vqa_outputs(self, producer, block: int = 2048) -> int:
        """The VQA model is frozen (cx.py:73-80), so q_emb / z / answer logits are functions of the example alone:
        compute them ONCE for the whole split (`producer(img_idx, wids) -> q, z_orig, z_knns, a_knns`, device tensors)
        and keep them resident -- 233 KB per example in fp32, 49 GB for the 211 k-example train split, which is what
        288 GB of HBM are for.  Later batches are index_selects; returns the bytes held."""
        N, K = self.N, self.K
        q = z_o = z_k = a_k = None
        for i in range(0, N, block):
            sel = torch.arange(i, min(i + block, N), device=self.device)
            bq, bzo, bzk, bak = producer(self.img_idx.index_select(0, sel), self.question_wids.index_select(0, sel))
            if q is None:
                q = torch.empty(N, bq.shape[1], dtype=torch.float32, device=self.device)
                z_o = torch.empty(N, bzo.shape[1], dtype=torch.float32, device=self.device)
                z_k = torch.empty(N, K, bzk.shape[2], dtype=torch.float32, device=self.device)
                a_k = torch.empty(N, K, bak.shape[2], dtype=torch.float32, device=self.device)
            q[i:i + block], z_o[i:i + block], z_k[i:i + block], a_k[i:i + block] = bq, bzo, bzk, bak
        self.vqa_cache = (q, z_o, z_k, a_k)
        return sum(t.numel() * 4 for t in self.vqa_cache)

    def cached_vqa(self, sel: torch.Tensor):
        q, z_o, z_k, a_k = self.vqa_cache
        return q.index_select(0, sel), z_o.index_select(0, sel), z_k.index_select(0, sel), a_k.index_select(0, sel)

    def dense_features(self, img_idx: torch.Tensor) -> torch.Tensor:
        """[B, K+1, dim_v] block as the reference materialises it (only for non-HIP VQA producers / tests)."""
        return self.feats.index_select(0, img_idx.reshape(-1).long()).view(img_idx.shape[0], img_idx.shape[1], -1)


def write_synthetic_cx_files(root: str, n_train=256, n_val=128, n_img=400, dim_v=2048, n_words=60, n_answers=2000,
                             knn_size=24, maxlength=26, seed=0, with_embedding=True, dim_a=2400) -> dict:
    """Writes a tiny dataset in the reference's on-disk layout (for tests and offline rehearsal of real-data mode):
    <root>/vqa/pickle_old/{trainset_augmented,trainset_augmented_small,valset_augmented_small,valset_augmented}.pickle,
    <root>/vqa/answer_embedding.pickle, <root>/features/{train,val}set.npy + .txt.  Returns the paths."""
    rng = np.random.default_rng(seed)
    vqa_dir, feat_dir = os.path.join(root, "vqa"), os.path.join(root, "features")
    os.makedirs(os.path.join(vqa_dir, "pickle_old"), exist_ok=True)
    os.makedirs(feat_dir, exist_ok=True)
    vocab_words = ["w%d" % i for i in range(n_words)]
    vocab_answers = ["a%d" % i for i in range(n_answers)]

    def split(name, n_ex, tag):
        names = ["COCO_%s2014_%012d.jpg" % (tag, i) for i in range(n_img)]
        feats = (np.abs(rng.standard_normal((n_img, dim_v))) * 0.45).astype(np.float32)
        np.save(os.path.join(feat_dir, "%sset.npy" % name), feats)
        with open(os.path.join(feat_dir, "%sset.txt" % name), "w") as f:
            f.write("\n".join(names) + "\n")
        order = rng.permutation(n_img)                       # name_to_index need not be the identity
        name_to_index = {names[j]: int(j) for j in order}
        examples = []
        for _ in range(n_ex):
            rows = rng.choice(n_img, size=knn_size + 1, replace=False)
            length = int(rng.integers(3, maxlength + 1))
            wids = [int(w) for w in rng.integers(1, n_words + 1, size=length)] + [0] * (maxlength - length)
            k = int(rng.integers(0, knn_size))
            examples.append({"image_name": names[rows[0]], "knns": [names[r] for r in rows[1:]],
                             "comp": {"knn_index": k, "image_name": names[rows[1 + k]]},
                             "question_wids": wids, "answer_aid": int(rng.integers(0, n_answers)),
                             "question_id": int(rng.integers(0, 1 << 30))})
        return {"examples_list": examples, "name_to_index": name_to_index, "vocab_words": vocab_words,
                "vocab_answers": vocab_answers}

    train, val = split("train", n_train, "train"), split("val", n_val, "val")
    files = {"trainset_augmented.pickle": train, "trainset_augmented_small.pickle": dict(train, examples_list=train["examples_list"][:64]),
             "valset_augmented_small.pickle": dict(val, examples_list=val["examples_list"][:max(32, n_val // 2)]),
             "valset_augmented.pickle": val}
    for fn, d in files.items():
        with open(os.path.join(vqa_dir, "pickle_old", fn), "wb") as f:
            pickle.dump(d, f)
    if with_embedding:
        with open(os.path.join(vqa_dir, "answer_embedding.pickle"), "wb") as f:
            pickle.dump((rng.standard_normal((n_answers, dim_a)) * 0.5).astype(np.float32), f)
    return {"path_trainset": vqa_dir, "path_features": feat_dir}
