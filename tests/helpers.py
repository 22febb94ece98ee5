"""Shared helpers for the test-suite: load a golden fixture into oracle-shaped inputs."""
import os

import numpy as np
import torch

from oracle import ncx_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# Gradient tolerance of SURVEY 8c: rel * max|reference grad| per tensor, NO absolute floor -- except for out.bias, whose
# gradient is mathematically zero under a listwise softmax (sum_k (p_k - y_k) = 0; the reference itself holds ~4e-8 of
# round-off there), so "relative to its own max" is meaningless for that one tensor.
ZERO_GRAD_TENSORS = ("out.bias",)
ZERO_GRAD_FLOOR = 1e-2


def grad_tol(name, ref, rel=1e-4):
    """Absolute element tolerance for the gradient tensor `name` whose reference value is `ref` (array / tensor)."""
    m = float(np.abs(np.asarray(ref)).max()) if np.asarray(ref).size else 0.0
    if name in ZERO_GRAD_TENSORS:
        m = max(m, ZERO_GRAD_FLOOR)
    return rel * m
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
            tol = grad_tol(n, ref, rel)
            err = np.abs(np.asarray(grads[n]).reshape(ref.shape) - ref).max()
            assert err <= tol, (n, err, tol)
        elif key.startswith("gradval/"):
            n = key[8:]
            ref = g[key]
            got = np.asarray(grads[n]).reshape(-1)[g["gradidx/" + n]]
            tol = grad_tol(n, ref, rel)            # (max over the stored sample <= max over the tensor: the stricter bound)
            assert np.abs(got - ref).max() <= tol, (n, np.abs(got - ref).max(), tol)
        elif key.startswith("gradnorm/"):
            n = key[9:]
            nrm = np.linalg.norm(np.asarray(grads[n]).astype(np.float64))
            assert abs(nrm - float(g[key])) <= 1e-4 * max(float(g[key]), ZERO_GRAD_FLOOR if n in ZERO_GRAD_TENSORS else 0.0), (n, nrm, float(g[key]))


def random_case_f32(seed, B, d, scale=0.45):
    """Seeded synthetic inputs of the shapes NeuralModel.forward sees (|N(0,1)|*0.45 features etc.)."""
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    return dict(image_features=t(np.abs(rng.standard_normal((B, d.K + 1, d.dv), dtype=np.float32)) * scale),
                q_emb=t(rng.standard_normal((B, d.dq), dtype=np.float32) * 0.3), z_orig=t(rng.standard_normal((B, d.dz), dtype=np.float32)),
                z_knns=t(rng.standard_normal((B, d.K, d.dz), dtype=np.float32)),
                a_knns=t(rng.standard_normal((B, d.K, d.A), dtype=np.float32) * 2),
                answer_aids=torch.from_numpy(rng.integers(0, d.A, size=B)), gt=torch.from_numpy(rng.integers(0, d.K, size=B)))


def condition_away_from_kinks(params, d, batch, seed, tau=2e-5, bf16=False, max_rounds=40):
    """Full-size parity inputs must not sit ON a discontinuity of the network, where the fp32 summation order alone
    decides the outcome (in the reference too: two BLAS builds disagree there).  Triplets with a linear_1 pre-activation
    within `tau` of the ReLU kink -- and, for the bf16 variant, a distance feature within a few fp32 ulps of a bf16
    rounding boundary -- are redrawn until none is left (the same idea as the rank-gap guard of the Recall fixtures,
    SURVEY 7).  Returns the number of redrawn triplets; `batch` is modified in place."""
    B = batch["gt"].shape[0]
    keys = ("image_features", "q_emb", "z_orig", "z_knns", "a_knns", "answer_aids", "gt")
    todo = torch.arange(B)
    redrawn = 0
    for rnd in range(max_rounds):
        sub = {k: batch[k][todo] for k in keys}
        taps = {}
        with torch.no_grad():
            fwd = orc.forward_bf16 if bf16 else orc.forward_faithful
            fwd(params, d, sub["image_features"], sub["q_emb"], sub["z_orig"], sub["z_knns"], sub["a_knns"], sub["answer_aids"], taps=taps)
        bad = taps["pre1"].abs().flatten(1).min(1).values < tau
        if bf16:
            x = taps["dist"]
            bf = lambda t: t.bfloat16().float()
            bad |= ((bf(x * (1 + 2e-6)) != bf(x)) | (bf(x * (1 - 2e-6)) != bf(x))).any(1)
        todo = todo[bad]
        if todo.numel() == 0:
            return redrawn
        redrawn += todo.numel()
        fresh = random_case_f32(seed * 1000 + rnd + 1, todo.numel(), d)
        for k in keys:
            batch[k][todo] = fresh[k]
    raise AssertionError("could not condition the batch away from the ReLU kinks")
