"""Host-side batching helpers with the semantics of the reference's data path for the CX script.

`examples_to_arrays` mirrors getDataFromBatch (counterexamples.py:519-547) minus its host gather: it returns the
ROW INDICES of the original image + its K nearest neighbours (the kernels gather from the resident feature table),
question word ids, answer ids and the index of the complementary image among the neighbours.
Example schema (vqacx_*_builder notebooks): {'image_name', 'knns': [K names], 'comp': {'knn_index', ...},
'question_wids': [26 ints], 'answer_aid': int}.
"""
import random
from typing import Dict, List, Sequence

import numpy as np


def batchify(example_list: List[dict], batch_size: int, shuffle: bool = True, rng: random.Random = None):
    """counterexamples.py:509-516: in-place shuffle of the caller's list, slices of batch_size, last partial kept."""
    if shuffle:
        (rng or random).shuffle(example_list)
    return [example_list[i:i + batch_size] for i in range(0, len(example_list), batch_size)]


def examples_to_arrays(batch: Sequence[dict], name_to_index: Dict[str, int], pairwise: bool = False, rng: random.Random = None):
    """-> img_idx int32 [B, K+1], question_wids int64 [B, T], answer_aids int32 [B], comp_idxs int32 [B].
    pairwise=True keeps [comp, one random other] as the reference does (counterexamples.py:528-533)."""
    img_idx, wids, aids, comps = [], [], [], []
    for ex in batch:
        row = [name_to_index[ex["image_name"]]]
        knn = [name_to_index[n] for n in ex["knns"]]
        if pairwise:
            comp = knn[ex["comp"]["knn_index"]]
            others = list(knn)
            others.remove(comp)
            knn = [comp, (rng or random).choice(others)]
        img_idx.append(row + knn)
        wids.append(ex["question_wids"])
        aids.append(ex["answer_aid"])
        comps.append(ex["comp"]["knn_index"])
    return (np.asarray(img_idx, np.int32), np.asarray(wids, np.int64), np.asarray(aids, np.int32), np.asarray(comps, np.int32))
