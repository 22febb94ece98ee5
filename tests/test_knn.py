"""GPU brute-force kNN (ncx_knn) vs the reference's knn.py procedure (scikit-learn NearestNeighbors,
fixture tests/golden/g8_knn.npz made by oracle/make_golden_knn.py; live scikit-learn for the larger cases)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")]
G8 = dict(np.load(os.path.join(ROOT, "tests", "golden", "g8_knn.npz")))


def _table(seed, n, dv, dup=0):                       # same generator as oracle/make_golden_knn.py
    rng = np.random.default_rng(seed)
    x = (np.abs(rng.standard_normal((n, dv))) * 0.45).astype(np.float32)
    for i in range(dup):
        x[n - 1 - i] = x[i]
    return x


def _check(idx, dist, ref_idx, ref_dist, rtol=2e-6, table=None):
    """Distances must agree everywhere; indices must agree wherever the reference's neighbouring distances are not
    tied within rounding (scikit-learn computes |x|^2 + |y|^2 - 2xy: its self-distance is ~1e-7..1e-3, not 0)."""
    atol = 2e-3
    assert np.all(np.abs(dist - ref_dist) <= atol + rtol * ref_dist), float(np.abs(dist - ref_dist).max())
    differ = idx != ref_idx
    if differ.any():
        rows, cols = np.nonzero(differ)
        for r, c in zip(rows, cols):
            near = np.abs(ref_dist[r] - ref_dist[r, c]) <= atol + 1e-5 * ref_dist[r, c]
            ok = idx[r, c] in set(ref_idx[r][near])
            if not ok and table is not None:            # an exact duplicate of a tied reference row (cut at the k boundary)
                ok = any(np.array_equal(table[idx[r, c]], table[j]) for j in ref_idx[r][near])
            assert ok, (r, c, idx[r], ref_idx[r], ref_dist[r])
    return float(differ.mean())


def test_knn_abi_validation_cpu():
    from neuralcx import _lib
    L = _lib.lib()
    assert L.ncx_knn_workspace_bytes(82783, 4096) >= 82783 * 4096 * 4
    assert L.ncx_knn_workspace_bytes(0, 16) == 0
    assert L.ncx_knn(None, 10, None, 10, 64, 5, 0, None, 0, None, None, None) == -1          # NULL pointers: before any launch


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_knn_matches_reference_procedure_fixture(case):
    from neuralcx.knn import knn
    seed, n, dv, k, dup = [int(v) for v in G8[case + "_spec"]]
    x = _table(seed, n, dv, dup)
    idx, dist = knn(torch.from_numpy(x).cuda(), k=k, block_rows=256)         # several query blocks, ragged last block
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy().astype(np.float64)
    assert idx.dtype == np.int64 and idx.shape == (n, k)
    frac = _check(idx, dist, G8[case + "_indices"].astype(np.int64), G8[case + "_distances"], table=x)
    assert frac <= (0.02 if dup else 0.0)                                     # only exact duplicates may swap
    if not dup:
        assert np.array_equal(idx[:, 0], np.arange(n)) and np.all(dist[:, 0] == 0.0)        # self first, exactly 0


@pytest.mark.gpu
def test_knn_full_width_rows_vs_sklearn_and_properties():
    """2048-d rows at a table size that needs the edge tiles; queries != table; ascending order; k = 1 and 25."""
    from sklearn.neighbors import NearestNeighbors
    from neuralcx.knn import knn
    x = _table(21, 5003, 2048)
    q = _table(22, 301, 2048)
    t = torch.from_numpy(x).cuda()
    idx, dist = knn(t, k=25, queries=torch.from_numpy(q).cuda(), block_rows=128)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy().astype(np.float64)
    ref_d, ref_i = NearestNeighbors(n_neighbors=25).fit(x).kneighbors(q)
    assert _check(idx, dist, ref_i, ref_d) == 0.0
    assert np.all(np.diff(dist, axis=1) >= 0)
    exact = np.sqrt(((q[:, None, :].astype(np.float64) - x[idx].astype(np.float64)) ** 2).sum(-1))
    assert np.abs(exact - dist).max() <= 1e-5 * exact.max()
    i1, d1 = knn(t, k=1)
    assert np.array_equal(i1.cpu().numpy()[:, 0], np.arange(5003)) and float(d1.abs().max()) == 0.0
    with pytest.raises(ValueError):
        knn(t, k=0)
    with pytest.raises(ValueError):
        knn(t, k=121)


@pytest.mark.gpu
def test_knn_mass_ties():
    """All-identical rows (zero vectors) and k close to the table size: every distance is 0, indices are distinct."""
    from neuralcx.knn import knn
    x = np.zeros((1500, 64), np.float32)
    x[:40] = _table(5, 40, 64)
    idx, dist = knn(torch.from_numpy(x).cuda(), k=30)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert all(len(set(r)) == 30 for r in idx) and idx.min() >= 0 and idx.max() < 1500
    assert np.all(dist[40:] == 0.0) and np.all(idx[40:] >= 40)
    ref = np.sqrt(((x[:40, None, :] - x[None, :, :]) ** 2).sum(-1))
    assert np.allclose(np.sort(ref, axis=1)[:, :30], dist[:40], atol=1e-5)


@pytest.mark.gpu
def test_knn_cli_writes_reference_file_format(tmp_path):
    """knn.py drop-in: same arguments, np.save of {"indices", "distances"} (knn.py:56-58)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ncx_knn_cli", os.path.join(ROOT, "vqa-counterexamples_amd", "knn.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    seed, n, dv, k, dup = [int(v) for v in G8["a_spec"]]
    np.save(os.path.join(tmp_path, "trainset.npy"), _table(seed, n, dv, dup))
    path = cli.main([str(tmp_path), "--hdf5_file", "trainset.hdf5", "--save_dir", str(tmp_path), "-k", str(k)])
    res = np.load(path, allow_pickle=True).item()
    assert res["indices"].dtype == np.int64 and res["distances"].dtype == np.float64
    assert np.array_equal(res["indices"], G8["a_indices"].astype(np.int64))
