"""Shared helpers for the test-suite: load a golden fixture into oracle-shaped inputs."""
import os

import numpy as np
import torch

from oracle import ncx_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# out.bias has a mathematically zero gradient under a listwise softmax (sum_k (p_k - y_k) = 0; the reference
# itself holds 1e-8 round-off there), so "relative to max|grad|" needs an absolute floor.
GRAD_FLOOR = 1e-2
SPEC_KEYS = ("v_emb", "v_mult", "v_dist", "v_rank", "q_emb", "a_emb", "z_emb")


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix)
                  and f.startswith(("g1_", "g2_", "g3_")))          # model cases (g4 / g7 / g8: loss, data helpers, kNN)


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    K, dv, dq, dz, da, A, H, L = [int(x) for x in g["dims"]]
    d = orc.Dims(K=K, dv=dv, dq=dq, dz=dz, da=da, A=A, H=H, L=L)
    spec = {k: bool(v) for k, v in zip(SPEC_KEYS, g["spec"])}
    params = orc.init_params(d, int(g["weight_seed"]), float(g["weight_gain"]))
    batch = dict(image_features=torch.from_numpy(g["image_features"]),
                 q_emb=torch.from_numpy(g["q_emb"]), z_orig=torch.from_numpy(g["z_orig"]),
                 z_knns=torch.from_numpy(g["z_knns"]), a_knns=torch.from_numpy(g["a_knns"]),
                 answer_aids=torch.from_numpy(g["answer_aids"]), gt=torch.from_numpy(g["gt"]))
    return g, d, spec, params, batch


def check_grads_against_golden(g, grads, rel=1e-4):
    """grads: dict name -> np.ndarray.  Tolerance: rel * max|golden grad| per tensor."""
    for key in g:
        if key.startswith("grad/"):
            n = key[5:]
            ref = g[key]
            tol = rel * max(np.abs(ref).max(), GRAD_FLOOR)
            err = np.abs(np.asarray(grads[n]).reshape(ref.shape) - ref).max()
            assert err <= tol, (n, err, tol)
        elif key.startswith("gradval/"):
            n = key[8:]
            ref = g[key]
            got = np.asarray(grads[n]).reshape(-1)[g["gradidx/" + n]]
            scale = float(g["gradnorm/" + n]) / np.sqrt(np.asarray(grads[n]).size)   # rms of the tensor
            tol = rel * max(np.abs(ref).max(), scale, GRAD_FLOOR)
            assert np.abs(got - ref).max() <= tol, (n, np.abs(got - ref).max(), tol)
        elif key.startswith("gradnorm/"):
            n = key[9:]
            nrm = np.linalg.norm(np.asarray(grads[n]).astype(np.float64))
            assert abs(nrm - float(g[key])) <= 1e-4 * max(float(g[key]), GRAD_FLOOR), (n, nrm, float(g[key]))
