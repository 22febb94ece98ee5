"""Per-step |loss(HIP) - loss(CPU oracle training)| of the 40-step full-width run (tests/test_dropin_gpu.py), for diagnosing
how fast two fp32 summation orders separate under Adam.  usage: python tools/train_curve.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import ncx_oracle as orc
from neuralcx.engine import NeuralCXEngine
from neuralcx.synth import SyntheticCX
DEV = "cuda:0"
d = orc.Dims()
B, steps, p_drop, lr = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 0.25, float(os.environ.get("LR", "1e-3"))
data = SyntheticCX(n_triplets=B * steps + 128, n_img=1024, seed=77, device=DEV)
eng = NeuralCXEngine(H=d.H, L=d.L, drop_p=p_drop, lr=lr, device=DEV)
params = orc.init_params(d, seed=42)
eng.load_state(params)
st = orc.AdamState(); cur = {k: v.clone() for k, v in params.items()}
feats_cpu = data.feats.cpu()
out = []
for s in range(steps):
    b, gt = data.batch(torch.arange(s * B, (s + 1) * B, device=DEV), first_id=s * B)
    r = eng.train_step(b, gt)
    seed = (eng.seed << 32) ^ eng.step_count
    masks = [orc.dropout_keep_mask(seed, 1, B * d.K, d.H, p_drop)]
    cb = dict(image_features=feats_cpu[b.img_idx.cpu().long()], q_emb=b.q_emb.cpu(), z_orig=b.z_orig.cpu(), z_knns=b.z_knns.cpu(),
              a_knns=b.a_knns.cpu(), answer_aids=b.answer_aids.cpu().long(), gt=gt.cpu().long())
    cur, _, l_ref, _ = orc.train_step(cur, d, cb, st, lr=lr, drop_p=p_drop, keep_masks=masks)
    out.append(abs(float(r["loss"]) - float(l_ref)))
    # weight distance (excluding out.bias)
wd = max(float((eng.state_dict()[k].cpu() - cur[k]).abs().max()) for k in cur if k != "out.bias")
print("lr %g  |dloss| per step: %s  max|dW| after %d steps: %.2e" % (lr, " ".join("%.1e" % x for x in out), steps, wd))
