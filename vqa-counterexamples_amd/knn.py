#!/usr/bin/env python3
"""k nearest neighbours of COCO image features -- CLI drop-in for the reference's knn.py (same arguments and output
file), computed on one MI355X by neuralcx.knn (ncx_knn) instead of scikit-learn on the host.

    python knn.py data/coco/extract/arch,fbresnet152_size,448 --hdf5_file trainset.hdf5 \\
           --save_dir data/coco/extract --save_file knn_results_trainset.npy -k 25

Reads `<base_dir>/<hdf5_file>` (dataset 'noatt'; needs h5py) or the `.npy` conversion of it
(neuralcx.formats.convert_hdf5_features), writes np.save({"indices": int64 [N, k], "distances": float64 [N, k]}) as
knn.py:56-58 does.  `--batch_size` is accepted for compatibility; queries are processed in blocks of --block_rows.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from neuralcx.knn import knn                      # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("base_dir", type=str, help="path to dir containing extracted hdf5 (or converted npy) features")
    p.add_argument("--hdf5_file", type=str, default="trainset.hdf5")
    p.add_argument("--save_dir", type=str, default="data/coco/extract")
    p.add_argument("--save_file", type=str, default="knn_results_trainset.npy")
    p.add_argument("-k", "--k", default=25, type=int)
    p.add_argument("-b", "--batch_size", default=10, type=int, help="(ignored: kept for command-line compatibility)")
    p.add_argument("--block_rows", default=4096, type=int, help="query rows per GEMM block")
    return p


def load_features(base_dir, fname):
    path = os.path.join(base_dir, fname)
    npy = path if path.endswith(".npy") else os.path.splitext(path)[0] + ".npy"
    if os.path.exists(npy):
        return np.load(npy, mmap_mode="r")
    assert os.path.exists(path), path
    try:
        import h5py
    except ImportError as e:
        raise SystemExit("%s needs h5py; convert it once with neuralcx.formats.convert_hdf5_features" % path) from e
    return np.array(h5py.File(path, "r").get("noatt"))


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert os.path.isdir(args.save_dir)
    save_path = os.path.join(args.save_dir, args.save_file)
    if os.path.exists(save_path):
        print("Warning: {} already exists and will be overwritten.".format(save_path))
    print("Saving results to {}".format(save_path))
    features = load_features(args.base_dir, args.hdf5_file)
    print("Loaded features as array of size {}".format(features.shape))
    if not torch.cuda.is_available():
        raise SystemExit("knn.py: an MI355X is required (no CPU fallback)")
    table = torch.empty(features.shape, dtype=torch.float32, device="cuda:0")
    for i in range(0, features.shape[0], 16384):
        table[i:i + 16384] = torch.from_numpy(np.array(features[i:i + 16384], dtype=np.float32)).cuda()
    print("Starting KNN computations for k={}...".format(args.k))
    torch.cuda.synchronize(); t0 = time.time()
    idx, dist = knn(table, k=args.k, block_rows=args.block_rows)
    torch.cuda.synchronize()
    print("{} rows in {:.2f} s".format(features.shape[0], time.time() - t0))
    np.save(save_path, {"indices": idx.cpu().numpy(), "distances": dist.cpu().numpy().astype(np.float64)})
    return save_path


if __name__ == "__main__":
    main()
