"""Gradient of ONE batch of the full-width training test data (tests/test_dropin_gpu.py) HIP vs CPU oracle, per tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd"), os.path.join(ROOT, "tests")]
import torch, numpy as np
from oracle import ncx_oracle as orc
from neuralcx import ops
from neuralcx.synth import SyntheticCX
DEV = "cuda:0"
d = orc.Dims()
B, steps, p_drop = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 40, 0.25
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
data = SyntheticCX(n_triplets=B * steps + 128, n_img=1024, seed=77, device=DEV)
params = orc.init_params(d, seed=42)
b, gt = data.batch(torch.arange(which * B, (which + 1) * B, device=DEV), first_id=which * B)
seed = (42 << 32) ^ 1
p = {ops.STATE_TO_FIELD[k]: v.to(DEV) for k, v in params.items()}
TRAIN = os.environ.get("TRAIN", "1") == "1"
dims = ops.make_dims(b, H=d.H, L=d.L, da=d.da, A=d.A, training=TRAIN, drop_p=p_drop if TRAIN else 0.0, seed=seed)
ws = ops.alloc_workspace(dims, DEV)
scores = ops.forward(dims, b, p, ws)
lr = ops.ranking_loss(scores, gt)
grads = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
ops.backward(dims, b, p, ws, lr["dscores"], grads)
torch.cuda.synchronize()
masks = [orc.dropout_keep_mask(seed, 1, B * d.K, d.H, p_drop)]
cb = dict(image_features=data.feats.cpu()[b.img_idx.cpu().long()], q_emb=b.q_emb.cpu(), z_orig=b.z_orig.cpu(), z_knns=b.z_knns.cpu(),
          a_knns=b.a_knns.cpu(), answer_aids=b.answer_aids.cpu().long(), gt=gt.cpu().long())
s_ref, l_ref, g_ref = orc.loss_and_grads(params, d, cb, drop_p=p_drop if TRAIN else 0.0, keep_masks=masks if TRAIN else None)
print("scores err %.2e loss %.7f vs %.7f  dup aids %d" % (float((scores.cpu() - s_ref).abs().max()), float(lr["loss"]), float(l_ref),
      B - len(set(cb["answer_aids"].tolist()))))
import ctypes as C
from neuralcx import _lib
def region(which):
    o, nb = C.c_size_t(0), C.c_size_t(0)
    _lib.check(_lib.lib().ncx_ws_region(C.byref(dims), which, C.byref(o), C.byref(nb)), "ws_region")
    base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
    return ws[base + o.value: base + o.value + nb.value].view(torch.float32).view(B * d.K, d.H).cpu()
h1, dpre = region(2), region(3)
taps = {}
leaf = {k: v.clone() for k, v in params.items()}
orc.forward_faithful(leaf, d, cb["image_features"], cb["q_emb"], cb["z_orig"], cb["z_knns"], cb["a_knns"], cb["answer_aids"], drop_p=p_drop if TRAIN else 0.0,
                     keep_masks=masks if TRAIN else None, taps=taps)
pre = taps["pre1"].reshape(B * d.K, d.H)
h_ref = torch.relu(pre) * (masks[0] / (1 - p_drop) if TRAIN else 1.0)
eh = (h1 - h_ref).abs()
print("h1 max err %.3e  #(h1>0 != ref>0) %d   rows with mismatch: %s" % (float(eh.max()), int(((h1 > 0) != (h_ref > 0)).sum()),
      sorted(set(torch.nonzero((h1 > 0) != (h_ref > 0))[:, 0].tolist()))[:20]))
bad = torch.nonzero(eh > 1e-4)
print("entries with |err| > 1e-4:", bad.shape[0], bad[:10].tolist(), [float(h1[i, j]) for i, j in bad[:5].tolist()], [float(h_ref[i, j]) for i, j in bad[:5].tolist()])
off = orc.Dims().offsets()
for k, ref in g_ref.items():
    g = grads[ops.STATE_TO_FIELD[k]].cpu()
    err = (g - ref).abs()
    print("%-26s max|ref| %.3e  max err %.3e  (%.1e of max)  nan %d" % (k, float(ref.abs().max()), float(err.max()), float(err.max() / max(float(ref.abs().max()), 1e-30)), int(torch.isnan(g).sum())))
    if k == "linear_1.weight":
        names = list(off.keys()); cols = [off[n] for n in names] + [ref.shape[1]]
        for i, n in enumerate(names):
            e = err[:, cols[i]:cols[i + 1]]
            print("    %-12s err %.3e ref max %.3e" % (n, float(e.max()), float(ref[:, cols[i]:cols[i + 1]].abs().max())))
