#!/usr/bin/env python
"""bench.py -- NeuralCX training throughput on MI355X (BASELINE.json: "VQA-CX triplets/sec").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 512] [--H 256] [--L 1] [--K 24]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training pass of the hot path over one batch of synthetic triplets PER RANK
(forward, listwise loss + recall, backward, [RCCL all-reduce], Adam) with every input resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- NeuralCX MLP, synthetic 2048-d features, 24 candidates,
batch 512, H=256, L=1, dropout 0.25, Adam lr 1e-4.  --scaling weak (default): each rank keeps batch 512
(global batch 512 N); --scaling strong: the GLOBAL batch stays 512 (SURVEY 8e's partition: 512 / N triplets per
rank, the optimisation problem of options/cx/*.yaml).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vqa-counterexamples_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, "Peak BF16/FP16 MFMA" (dense)


def usable_cores():
    """Cores this process may actually use: cgroup quota, else affinity mask (the GPU box exposes 256 logical
    CPUs but grants a 16-core share per GPU; oversubscribing OpenMP threads there is 100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    env = os.environ.get("NCX_CPU_THREADS")
    return int(env) if env else min(n, 16)


def cpu_baseline(steps=50, warmup=10, one_thread_steps=12, heldout=None):
    """Reference-faithful CPU path (oracle.FaithfulCPUModel: 24-iteration cat+Linear loop, softmax+bmm, autograd,
    torch.optim.Adam) at BASELINE configs[0] (B=32, H=256, L=1, full widths), the protocol of BASELINE.md 3 on a bounded
    sample: all usable cores, `warmup` untimed + `steps` timed steps (>= 50), then a short 1-thread run, loss and
    Recall@1/@5 of the CPU-trained model on `heldout` (dict of CPU tensors) after the pass."""
    from oracle import ncx_oracle as orc
    import numpy as np
    d = orc.Dims()
    B = 32
    torch.manual_seed(42)
    threads = usable_cores()
    m = orc.FaithfulCPUModel(d, drop_p=0.25, seed=42)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    rng = np.random.default_rng(1234)
    pool = []
    for _ in range(4):                        # a few distinct batches, cycled
        pool.append(((torch.randn(B, d.K + 1, d.dv).abs() * 0.45), torch.randn(B, d.dq) * 0.3, torch.randn(B, d.dz), torch.randn(B, d.K, d.dz),
                     torch.randn(B, d.K, d.A) * 2, torch.from_numpy(rng.integers(0, d.A, size=B)), torch.from_numpy(rng.integers(0, d.K, size=B))))

    def step(i):
        feats, q, zo, zk, ak, aid, gt = pool[i % len(pool)]
        scores = m(feats, q, zo, zk, ak, aid)
        loss = torch.nn.functional.cross_entropy(scores, gt, reduction="sum") / B
        opt.zero_grad(); loss.backward(); opt.step()
        return float(loss)

    def timed(nthreads, nwarm, nsteps):
        torch.set_num_threads(nthreads)
        for i in range(nwarm):
            step(i)
        t0 = time.perf_counter()
        for i in range(nsteps):
            loss = step(nwarm + i)
        return B * nsteps / (time.perf_counter() - t0), loss
    rate, loss = timed(threads, warmup, steps)
    rate1, _ = timed(1, 1, one_thread_steps)
    out = dict(value=round(rate, 2), unit="triplets/s", cores=threads, kind="port",
               one_thread=dict(value=round(rate1, 2), unit="triplets/s", cores=1, steps=one_thread_steps),
               final_loss=round(loss, 5),
               sample="%d timed train steps of batch 32 after %d warm-up (configs[0] shapes: K=24, 2048-d feats, H=256, L=1, dropout 0.25, "
                      "Adam lr 1e-4), torch %s CPU, %d threads; then %d steps on 1 thread" % (steps, warmup, torch.__version__, threads, one_thread_steps))
    if heldout is not None:
        torch.set_num_threads(threads)
        m.eval()
        with torch.no_grad():
            s = m(heldout["image_features"], heldout["q_emb"], heldout["z_orig"], heldout["z_knns"], heldout["a_knns"], heldout["answer_aids"])
        rank = orc.rank_of_gt(s.numpy(), heldout["gt"].numpy())
        out["recall_at_1"], out["recall_at_5"] = round(float((rank < 1).mean()), 4), round(float((rank < 5).mean()), 4)
        out["heldout_triplets"] = int(rank.shape[0])
    return out


def workload_label(args, c):
    """Which BASELINE.json config the flags select (the default flags are configs[1], the headline)."""
    if args.c3:
        return "configs[2]: fused HIP MUTAN producer (vqa_forward) + "
    if (args.batch, c["K"], c["H"], c["L"]) == (512, 24, 256, 1):
        return "configs[1] shape, bf16-operand variant (NOT the headline): " if args.bf16 else "configs[1]: "
    if (args.batch, c["K"]) == (1024, 48):
        return ("configs[4] shape (48 candidates, batch 1024, bf16 operands), one GPU: " if args.bf16
                else "configs[4] shape (48 candidates, batch 1024) in fp32, one GPU: ")
    return "non-BASELINE shape (--batch/--K/--H/--L given): "


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="triplets per rank per step")
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--L", type=int, default=1)
    ap.add_argument("--K", type=int, default=24)
    ap.add_argument("--n_img", type=int, default=82783)
    ap.add_argument("--pool", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--c3", action="store_true", help="configs[2]: z / answer logits produced per step by the fused HIP MUTAN (ncx_vqa_forward)")
    ap.add_argument("--bf16", action="store_true", help="configs[4] variant: bf16 MFMA operands for the two dominant GEMMs (NOT the headline: fp32 is)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=50, help="timed all-core steps of the CPU baseline (after 10 warm-up)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch triplets per rank; strong: --batch triplets globally (batch / N per rank)")
    ap.add_argument("--heldout", type=int, default=1024, help="held-out planted triplets for Recall@1/@5 after the timed steps")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    backend = os.environ.get("NCX_DIST_BACKEND", "nccl")               # "gloo": rehearsal of N ranks on one card
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend)

    from neuralcx import _lib
    from neuralcx.engine import NeuralCXEngine
    from neuralcx.synth import SyntheticCX

    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("--scaling strong: --batch (%d) must be a multiple of the number of ranks (%d)" % (args.batch, world))
        args.batch //= world                          # per-rank share of the fixed global batch
    eng = NeuralCXEngine(K=args.K, H=args.H, L=args.L, drop_p=0.25, lr=1e-4, device=dev, world_size=world, bf16=args.bf16)
    eng.rank = rank                                   # (per-rank dropout streams)
    eng.init_parameters(seed=42)                      # identical replicas on every rank
    data = SyntheticCX(n_triplets=args.batch * args.pool * world, K=args.K, n_img=args.n_img, seed=1234, device=dev)
    pool = []
    for i in range(args.pool):                        # rank r owns slice r of every global batch
        lo = (i * world + rank) * args.batch
        pool.append(data.batch(torch.arange(lo, lo + args.batch)))
    gb = args.batch * world

    mutan = None
    if args.c3:
        # random-init frozen MutanNoAtt of the options/cx/*.yaml shape (dim_hv = dim_hq = dim_mm = 360, R = 10, tanh)
        import vqa.models as M
        from neuralcx import ops
        opt = dict(arch="MutanNoAtt", seq2vec=dict(arch="gru", emb_size=8, dropout=0.0),
                   fusion=dict(dim_v=2048, dim_q=2400, dim_hv=360, dim_hq=360, dim_mm=360, R=10, dropout_v=0.5, dropout_q=0.5,
                               activation_v="tanh", activation_q="tanh", dropout_hv=0, dropout_hq=0), classif=dict(dropout=0.5))
        torch.manual_seed(42)
        vqa_model = M.factory(opt, ["w"], ["a%d" % i for i in range(2000)], cuda=True, data_parallel=False).eval()
        mutan = ops.MutanWeights(vqa_model)

    def step(i):
        b, gt = pool[i % args.pool]
        if mutan is not None:                       # configs[2]: the producer runs inside the timed step
            b = eng.make_batch_from_vqa(b.feats, b.img_idx, b.q_emb, b.answer_aids, mutan)
        return eng.train_step(b, gt, global_batch=gb)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    fence()                                         # (builds the RCCL communicator before any step)
    for i in range(args.warmup):
        r = step(i)
    fence()
    if rank == 0:
        _lib.profile_begin(["MAIN", "DW1C"], max_launches=4 * args.steps + 8)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(r["loss"].item())
    if world > 1:                                    # (every rank holds loss * B_local / B_global of its slice)
        lt = torch.tensor([loss], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(lt)
        loss = float(lt.item())
    # Recall@1/@5 (the other half of BASELINE.json's metric) of the weights after warmup + steps, on held-out triplets of the
    # same planted synthetic distribution (rank 0's replica; replicas are identical)
    heldout_cpu, rec = None, None
    if rank == 0 and args.heldout > 0 and mutan is None:
        from neuralcx import ops
        hd = SyntheticCX(n_triplets=args.heldout, K=args.K, n_img=args.n_img, seed=4321, device=dev, feats=data.feats)
        hits = torch.zeros(2, dtype=torch.int64, device=dev)
        hb = 256
        for lo in range(0, args.heldout, hb):
            b, gt = hd.batch(torch.arange(lo, min(lo + hb, args.heldout)))
            ev = eng.eval_step(b, gt)
            hits += ev["hits"].long()
            if lo == 0 and not args.no_cpu_baseline and world == 1:       # the same first 256 triplets go to the CPU-trained model
                heldout_cpu = dict(image_features=b.feats[b.img_idx.long()].cpu(), q_emb=b.q_emb.cpu(), z_orig=b.z_orig.cpu(),
                                   z_knns=b.z_knns.cpu(), a_knns=b.a_knns.cpu(), answer_aids=b.answer_aids.long().cpu(), gt=gt.long().cpu())
        rec = (float(hits[0]) / args.heldout, float(hits[1]) / args.heldout)

    if rank == 0:
        prof = _lib.profile_end()
        c = eng.cfg
        M = args.batch * c["K"]
        # Algorithmic flops per launch (SURVEY 8d conventions: one-hot rank columns = 0 flops; the a_emb_other
        # segment is consumed in its re-associated form softmax(a) . (E . W^T): A columns instead of da).
        p_cols = 2 * c["dv"] + 1 + c["dz"] + c["A"]                     # candidate segments
        s_cols = c["dv"] + c["dq"] + c["dz"] + c["da"]                  # per-triplet shared segments
        flops = {"MAIN": 2.0 * M * c["H"] * p_cols,                     # h1 = candidates . W1 slices
                 "DW1C": 2.0 * M * c["H"] * p_cols + 2.0 * args.batch * c["H"] * s_cols}   # all dW1 columns + dGt
        d0 = eng._dims(pool[0][0], True, 1.0 / gb)
        plans = {k: _lib.plan_query(d0, k) for k in ("MAIN", "DW1C")}
        names = {"MAIN": "k_main_fwd (csrc/ncx_main.h): linear_1 forward, the candidate segments chained into one fp32-MFMA accumulator "
                         "(v_other and v_orig*v_other as one per-triplet fold where the plan says so), Sh / ReLU / Dropout epilogue; tile: %s"
                         % plans["MAIN"]["tile"],
                 "DW1C": "linear_1 weight gradient (all columns + dGt): k_dw_km (v_other + v_mult columns in one MFMA pass, per-triplet "
                         "fold, 8 k-chunks) + seg_gemm TN %s grouped, %d-way aligned split-K (remaining columns), incl. reductions"
                         % (plans["DW1C"]["tile"], plans["DW1C"]["ksplit"])}
        peak = PEAK_F32_MFMA_TFLOPS
        if args.bf16:
            peak = PEAK_BF16_MFMA_TFLOPS
            names = {"MAIN": "gemm_bf16_nt (packed candidate rows . packed weights^T, fwd; 64x64 or 128x128 tiles by workgroup count; weight pack excluded)",
                     "DW1C": "gemm_bf16_tn 128x128, 8 k-chunks (one per XCD) + dpre cast + reduce/scatter (all candidate weight-grad columns + dGt)"}
            for k in plans:
                plans[k] = dict(plans[k], tile="bf16", ksplit=1 if k == "MAIN" else 8)
        # per STEP: the weight gradient of linear_1 is two launches under one id (fused v_other / v_mult kernel + grouped rest)
        per = {k: sum(v) / args.steps for k, v in prof.items() if v}
        # the dominant KERNEL is the one with the longest single launch (DW1C is two launches per step: the per-step sum of a
        # pair must not outrank the one launch that is longer than either of them)
        per_launch = {k: sum(v) / len(v) for k, v in prof.items() if v}
        dom = max(per_launch, key=per_launch.get) if per_launch else None
        roof = None
        if dom:
            ach = flops[dom] / (per[dom] * 1e-3) / 1e12
            # HBM bytes per launch of that kernel from the committed PMC passes (collected offline as the guide
            # prescribes: separate --pmc runs, FETCH_SIZE x2 on gfx950); only valid for the default workload
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "r2_traffic.json")
            if os.path.exists(tpath) and (args.batch, c["K"], c["H"], c["L"]) == (512, 24, 256, 1) and not args.bf16:
                tj = json.load(open(tpath))
                traffic = tj.get(dom, {}).get("bytes_per_launch")
                traffic_source = "profiles/r2_traffic.json: %s" % tj.get("_source", "rocprofv3 --pmc passes of this workload")
            roof = dict(bound="mfma", kernel=names[dom], achieved=round(ach, 2), peak=peak,
                        unit="TFLOP/s", frac=round(ach / peak, 4), traffic=traffic, traffic_source=traffic_source,
                        launch_ms=round(per[dom], 4), algorithmic_gflop_per_launch=round(flops[dom] / 1e9, 3),
                        other={k: dict(launch_ms=round(v, 4), launches_per_step=len(prof[k]) // args.steps, tflops=round(flops[k] / (v * 1e-3) / 1e12, 2),
                                       plan=plans[k]) for k, v in per.items()})
        out = dict(metric="VQA-CX triplets/sec (24 candidates each), NeuralCX training step",
                   value=round(gb * args.steps / dt, 1), unit="triplets/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True,
                   scaling=args.scaling, vs_baseline=None, dtype="bf16 operands / f32 accumulate (first-layer GEMMs), f32 elsewhere" if args.bf16 else "f32",
                   data="synthetic",
                   config=dict(workload=workload_label(args, c) +
                                        "NeuralCX MLP train step (fwd+listwise loss/recall+bwd+Adam), synthetic "
                                        "2048-d feats, %d candidates, batch %d per GPU (%s scaling: global batch %d), H=%d, L=%d, dropout 0.25, %s"
                                        % (c["K"], args.batch, args.scaling, gb, c["H"], c["L"],
                                           "bf16 operands / fp32 accumulate for the linear_1 and answer-embedding products, fp32 elsewhere" if args.bf16 else "fp32"),
                               global_batch=gb, candidates=c["K"], dim_h=c["H"], n_layers=c["L"],
                               parallelism="dp%d" % world, feature_table_rows=args.n_img, final_loss=round(loss, 5)),
                   roofline=roof)
        if rec is not None:
            out["recall_at_1"], out["recall_at_5"] = round(rec[0], 4), round(rec[1], 4)
            out["recall_note"] = ("HIP path, %d held-out planted synthetic triplets, after %d training steps (chance: 0.0417 / 0.2083)"
                                  % (args.heldout, args.warmup + args.steps))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(steps=args.cpu_steps, heldout=heldout_cpu)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
