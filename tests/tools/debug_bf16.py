import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vqa-counterexamples_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import ncx_oracle as orc
import test_hip_parity as T
B, K, H, L, dv = [int(x) for x in sys.argv[1:6]]
d = orc.Dims(K=K, dv=dv, dq=50, dz=18, A=45, H=H, L=L)
params = orc.init_params(d, seed=17 + B, gain=3.0)
batch = T.random_case(300 + B, B, d)
for training in (False, True):
    seed = 0xABCDEF12345
    masks = [orc.dropout_keep_mask(seed, l, B * d.K, d.H, 0.25) for l in range(1, L + 1)]
    scores, lr, grads = T.run_hip_bf16(d, params, batch, training=training, drop_p=0.25 if training else 0.0, seed=seed)
    s_ref, l_ref, g_ref = orc.loss_and_grads_bf16(params, d, batch, drop_p=0.25 if training else 0.0, keep_masks=masks if training else None)
    print("training", training, "scores err", float(np.abs(scores.numpy() - s_ref.numpy()).max()))
    for k, ref in g_ref.items():
        ref = ref.numpy(); g = grads[k].reshape(ref.shape)
        e = np.abs(g - ref)
        print("  %-24s max|ref| %.3e  err %.3e  at %s" % (k, np.abs(ref).max(), e.max(), np.unravel_index(e.argmax(), e.shape)))
    if "linear_1.weight" in g_ref:
        ref = g_ref["linear_1.weight"].numpy(); g = grads["linear_1.weight"]; e = np.abs(g - ref)
        cols = e.max(0); rows = e.max(1)
        print("   w1 err by column block:", [float(cols[a:b].max()) for a, b in ((0, dv), (dv, 2*dv), (2*dv, 3*dv), (3*dv, 3*dv+1+K), (3*dv+1+K, 3*dv+1+K+50), (3*dv+1+K+50, 3*dv+1+K+50+18), (3*dv+1+K+68, 3*dv+1+K+86))])
        print("   w1 err by row block of 32:", [float(rows[i:i+32].max()) for i in range(0, H, 32)])
