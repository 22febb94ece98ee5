"""VERDICT r3 item 6, the bounded experiment: would split-bf16 ("bf16x3") operands for linear_1 pass the UNCHANGED fp32 parity suite?

x . w  ~  x_hi . w_hi + x_hi . w_lo + x_lo . w_hi      (x_hi = bf16(x), x_lo = bf16(x - x_hi); products of bf16 values are exact in fp32; fp32 accumulation)
runs at 16 / 3 the fp32-MFMA rate on v_mfma_f32_16x16x32_bf16.  Before any kernel is written this script measures, on the CPU with the oracle's own
segmented restatement, what the rounding does to the quantities the parity suite bounds, at BASELINE configs[1]'s widths:
  * all logits against the fp32 oracle and against fp64 (suite: <= 1e-4 absolute, tests/test_hip_parity.py);
  * every gradient tensor against fp64, relative to the tensor's max (suite: <= 1e-4, no floor), incl. the number of linear_1 units whose ReLU
    switches side (the suite's full-size gradient test conditions inputs 2e-5 away from the kinks: tests/helpers.py);
  * for comparison the same numbers of plain fp32 (what the HIP path does today).
usage: python tools/exp_bf16x3.py [B=96] [terms=3 | 6]   (6: x = hi + mid + lo, six products -- fp32-grade, 16 / 6 the fp32 rate)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import ncx_oracle as orc
from helpers import random_case_f32

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
TERMS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.set_num_threads(min(8, os.cpu_count() or 1))
d = orc.Dims()
bf = lambda t: t.bfloat16().float()


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = bf(r); parts.append(p); r = r - p
    return parts


def mm_split(x, w):
    """x [m, k] . w [n, k]^T with split-bf16 operands, fp32 accumulation"""
    if TERMS == 3:
        (xh, xl), (wh, wl) = split(x, 2), split(w, 2)
        return xh @ wh.t() + (xh @ wl.t() + xl @ wh.t())
    (x1, x2, x3), (w1, w2, w3) = split(x, 3), split(w, 3)
    return x1 @ w1.t() + ((x1 @ w2.t() + x2 @ w1.t()) + (x1 @ w3.t() + x2 @ w2.t() + x3 @ w1.t()))


class SplitLinear(torch.autograd.Function):
    """y = x W^T with split operands in the forward, in dX = g W and in dW = g^T x (the three products the kernels would run this way)"""
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return mm_split(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return mm_split(g, w.t().contiguous()), mm_split(g.t().contiguous(), x.t().contiguous())


def forward(params, batch, dtype, split_mm):
    """linear_1 over the full concat row (the reference's formulation, cx.py:309-322), row = (b, k)"""
    p = {k: v.to(dtype) for k, v in params.items()}
    f = lambda k: batch[k].to(dtype)
    v, q, zo, zk, ak = f("image_features"), f("q_emb"), f("z_orig"), f("z_knns"), f("a_knns")
    Bn, K = v.shape[0], d.K
    E = p["answer_embedding.weight"]
    vo = v[:, :1].expand(-1, K, -1); vk = v[:, 1:]
    dist = (vo - vk + 1e-6).norm(dim=2, keepdim=True)
    rank = torch.eye(K, dtype=dtype).unsqueeze(0).expand(Bn, -1, -1)
    a_gt = E[batch["answer_aids"]].unsqueeze(1).expand(-1, K, -1)
    a_k = torch.softmax(ak, dim=-1) @ E
    x = torch.cat([vo, vk, vo * vk, dist, rank, q.unsqueeze(1).expand(-1, K, -1), zo.unsqueeze(1).expand(-1, K, -1), zk, a_gt, a_k], dim=2).reshape(Bn * K, -1)
    W1 = p["linear_1.weight"]
    pre = (SplitLinear.apply(x, W1) if split_mm else x @ W1.t()) + p["linear_1.bias"]
    h = torch.relu(pre)
    s = (h @ p["out.weight"].t() + p["out.bias"]).view(Bn, K)
    return s, pre


def run(params, batch, dtype, split_mm):
    leaf = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
    s, pre = forward(leaf, batch, dtype, split_mm)
    loss = torch.nn.functional.cross_entropy(s, batch["gt"], reduction="sum") / s.shape[0]
    loss.backward()
    return s.detach(), pre.detach(), {k: v.grad.detach() for k, v in leaf.items()}


params = orc.init_params(d, seed=42, gain=3.0)          # (the parity suite's weights: gain 3)
batch = random_case_f32(31337, B, d)
s64, pre64, g64 = run(params, batch, torch.float64, False)
s32, pre32, g32 = run(params, batch, torch.float32, False)
sx, prex, gx = run(params, batch, torch.float32, True)
out = {"B": B, "terms": TERMS, "max_abs_logit": float(s64.abs().max()), "max_abs_pre1": float(pre64.abs().max())}
for name, s, pre, g in (("fp32", s32, pre32, g32), ("bf16x%d" % TERMS, sx, prex, gx)):
    r = {"logits_vs_f64": float((s.double() - s64).abs().max()), "logits_vs_fp32_oracle": float((s - s32).abs().max()),
         "pre1_vs_f64": float((pre.double() - pre64).abs().max()), "relu_side_switches": int(((pre > 0) != (pre64 > 0)).sum()),
         "units": int(pre.numel()), "units_within_2e-5_of_kink": int((pre64.abs() < 2e-5).sum())}
    for k in g64:
        mx = float(g64[k].abs().max())
        r["grad:" + k] = float((g[k].double() - g64[k]).abs().max()) / max(mx, 1e-30)
    out[name] = r
print(json.dumps(out, indent=1))
