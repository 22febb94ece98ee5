"""Generate tests/golden/g8_knn.npz: the procedure of the reference's knn.py:41-58 on seeded feature tables.

knn.py's arithmetic lives in scikit-learn (unpinned in requirements.txt; 1.7.2 here): NearestNeighbors(n_neighbors=k)
.fit(features) [auto -> brute force, euclidean] then kneighbors(features[i:i+batch]) over the table, concatenated.
This script repeats exactly those calls (the reference script itself cannot run: it needs h5py and an hdf5 file) and
stores inputs' seeds + the resulting indices / distances.  Usage: python oracle/make_golden_knn.py
"""
import os

import numpy as np
from sklearn.neighbors import NearestNeighbors

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def table(seed, n, dv, dup=0):
    rng = np.random.default_rng(seed)
    x = (np.abs(rng.standard_normal((n, dv))) * 0.45).astype(np.float32)     # ResNet post-ReLU pooled scale
    for i in range(dup):                                                      # exact duplicate rows (COCO has some)
        x[n - 1 - i] = x[i]
    return x


def knn_py(features, k, batch_size=10):                                      # knn.py:41-58
    nbrs = NearestNeighbors(n_neighbors=k)
    nbrs.fit(features)
    idx, dist = [], []
    for i in range(0, features.shape[0], batch_size):
        d, ind = nbrs.kneighbors(features[i:i + batch_size, :])
        idx.append(ind); dist.append(d)
    return np.concatenate(idx), np.concatenate(dist)


def main():
    out = {}
    for name, (seed, n, dv, k, dup) in {"a": (11, 700, 256, 25, 0), "b": (12, 333, 100, 25, 3), "c": (13, 64, 2048, 10, 0)}.items():
        x = table(seed, n, dv, dup)
        ind, dist = knn_py(x, k)
        out["%s_spec" % name] = np.array([seed, n, dv, k, dup], np.int64)
        out["%s_indices" % name] = ind.astype(np.int32)
        out["%s_distances" % name] = dist.astype(np.float64)
    np.savez_compressed(os.path.join(OUT, "g8_knn.npz"), **out)
    print("g8_knn written", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
