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

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, "Peak BF16/FP16 MFMA" (dense)


def launch_ranks(n, argv, env=None, timeout=None):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves, exactly as the contract's
    launch line does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same flags>), as a CHILD process.  This parent makes no GPU call of any kind (torch is not even imported
    here) and never re-execs; it relays the children's output (rank 0 prints the one JSON line) and returns the child's
    exit code."""
    import socket
    import subprocess
    with socket.socket() as s:                     # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: WORLD_SIZE unset and --gpus %d: launching %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


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


def cpu_baseline(dev, feats, steps=50, warmup=10, one_thread_steps=12, heldout=1024):
    """Reference-faithful CPU path (oracle.FaithfulCPUModel: 24-iteration cat+Linear loop, softmax+bmm, autograd,
    torch.optim.Adam) at BASELINE configs[0] (B=32, H=256, L=1, full widths), the protocol of BASELINE.md 3 on a bounded
    sample: all usable cores, `warmup` untimed + `steps` timed steps (>= 50), then a short 1-thread run.

    The other half of BASELINE.json's metric, Recall@1/@5, on the SAME footing for both sides: the CPU model and a fresh
    HIP engine start from the same weights and train on the same `warmup + steps` planted synthetic batches with the
    same dropout masks (the counter-based generator, restated in the oracle), then both score the same `heldout`
    held-out triplets.  (tests/test_dropin_gpu.py holds the long version of this comparison with its assertions.)"""
    import numpy as np
    import torch
    from oracle import ncx_oracle as orc
    from neuralcx.engine import NeuralCXEngine
    from neuralcx.synth import SyntheticCX
    d = orc.Dims()
    B, p_drop, lr = 32, 0.25, 1e-4
    torch.manual_seed(42)
    threads = usable_cores()
    n_train = warmup + steps
    data = SyntheticCX(n_triplets=B * n_train + heldout, K=d.K, n_img=feats.shape[0], seed=977, device=dev, feats=feats)
    feats_cpu = feats.cpu()

    def to_cpu(b, gt):
        return dict(image_features=feats_cpu[b.img_idx.cpu().long()], q_emb=b.q_emb.cpu(), z_orig=b.z_orig.cpu(), z_knns=b.z_knns.cpu(),
                    a_knns=b.a_knns.cpu(), answer_aids=b.answer_aids.cpu().long(), gt=gt.cpu().long())
    # ---- HIP side: the same weights, batches and dropout masks ------------------------------------------------------------
    params0 = orc.init_params(d, seed=42)
    eng = NeuralCXEngine(K=d.K, H=d.H, L=d.L, drop_p=p_drop, lr=lr, device=dev)
    eng.load_state(params0)
    cpu_batches = []
    for s in range(n_train):
        b, gt = data.batch(torch.arange(s * B, (s + 1) * B, device=dev), first_id=s * B)
        eng.train_step(b, gt)
        cpu_batches.append(to_cpu(b, gt))
    hits = torch.zeros(2, dtype=torch.int64, device=dev)
    held_cpu = []
    for lo in range(0, heldout, 256):
        hi = min(lo + 256, heldout)
        b, gt = data.batch(torch.arange(B * n_train + lo, B * n_train + hi, device=dev), first_id=B * n_train + lo)
        ev = eng.eval_step(b, gt)
        hits += ev["hits"].long()
        held_cpu.append(to_cpu(b, gt))
    hip_rec = (float(hits[0]) / heldout, float(hits[1]) / heldout)
    hip_loss = float(ev["loss"])
    del eng
    # ---- CPU side ------------------------------------------------------------------------------------------------------------
    m = orc.FaithfulCPUModel(d, drop_p=p_drop, seed=42)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)

    def step(i):
        bt = cpu_batches[i]
        m.keep_masks = [orc.dropout_keep_mask((42 << 32) ^ (i + 1), 1, B * d.K, d.H, p_drop)]       # the engine's per-step seed (rank 0)
        scores = m(bt["image_features"], bt["q_emb"], bt["z_orig"], bt["z_knns"], bt["a_knns"], bt["answer_aids"])
        loss = torch.nn.functional.cross_entropy(scores, bt["gt"], reduction="sum") / B
        opt.zero_grad(); loss.backward(); opt.step()
        return float(loss)
    torch.set_num_threads(threads)
    for i in range(warmup):
        step(i)
    t0 = time.perf_counter()
    for i in range(warmup, n_train):
        loss = step(i)
    rate = B * steps / (time.perf_counter() - t0)
    m.eval()
    ranks = []
    with torch.no_grad():
        for h in held_cpu:
            sc = m(h["image_features"], h["q_emb"], h["z_orig"], h["z_knns"], h["a_knns"], h["answer_aids"])
            ranks.append(orc.rank_of_gt(sc.numpy(), h["gt"].numpy()))
    rank = np.concatenate(ranks)
    cpu_rec = (float((rank < 1).mean()), float((rank < 5).mean()))
    m.train()
    torch.set_num_threads(1)                         # 1-thread figure (the weights no longer matter)
    step(0)
    t0 = time.perf_counter()
    for i in range(one_thread_steps):
        step(1 + i % (n_train - 1))
    rate1 = B * one_thread_steps / (time.perf_counter() - t0)
    torch.set_num_threads(threads)
    return dict(value=round(rate, 2), unit="triplets/s", cores=threads, kind="port",
                one_thread=dict(value=round(rate1, 2), unit="triplets/s", cores=1, steps=one_thread_steps),
                final_loss=round(loss, 5),
                sample="%d timed train steps of batch 32 after %d warm-up (configs[0] shapes: K=24, 2048-d feats, H=256, L=1, dropout 0.25, "
                       "Adam lr 1e-4), torch %s CPU, %d threads; then %d steps on 1 thread" % (steps, warmup, torch.__version__, threads, one_thread_steps),
                recall=dict(note="both sides: the same initial weights, the same %d batches of 32 planted synthetic triplets, the same dropout masks; "
                                 "then the same %d held-out triplets (chance 0.0417 / 0.2083)" % (n_train, heldout),
                            cpu_recall_at_1=round(cpu_rec[0], 4), cpu_recall_at_5=round(cpu_rec[1], 4),
                            hip_recall_at_1=round(hip_rec[0], 4), hip_recall_at_5=round(hip_rec[1], 4),
                            train_steps=n_train, heldout_triplets=heldout))


def workload_label(args, c):
    """Which BASELINE.json config the flags select (the default flags are configs[1], the headline)."""
    if args.c3:
        return "configs[2]: fused HIP MUTAN producer (vqa_forward) + "
    if (args.batch, c["K"], c["H"], c["L"]) == (512, 24, 256, 1):
        return ("configs[1] shape, bf16-operand variant (NOT the headline): " if args.bf16 else
                "configs[1] with --x6 (NOT the headline): " if args.x6 else "configs[1]: ")
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
    ap.add_argument("--x6", action="store_true", help="NCX_F_X6: the balanced TN weight-gradient launch on the bf16 matrix path with three-plane fp32-grade operands (NOT the headline: fp32 MFMA is)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=50, help="timed all-core steps of the CPU baseline (after 10 warm-up)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch triplets per rank; strong: --batch triplets globally (batch / N per rank)")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="milliseconds of unrelated fp32 matmuls (torch) before the W warm-up steps: an idle MI355X needs ~30 ms of continuous load "
                         "to reach its operating point (measured: the first of 20 timed steps after 5 warm-up steps runs 8 %% slower than the "
                         "25th); 0 = none.  Reported in the JSON line")
    ap.add_argument("--heldout", type=int, default=1024, help="held-out planted triplets for Recall@1/@5 (after the timed steps; and of the CPU/HIP same-training comparison)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: this process becomes the launcher of the N ranks (no GPU call here)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    backend = os.environ.get("NCX_DIST_BACKEND", "nccl")               # "gloo": rehearsal of N ranks on one card
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py --gpus %d: RCCL needs one GPU per rank and this box exposes %d (two ranks on one device are "
                         "rejected by RCCL); set NCX_DIST_BACKEND=gloo to rehearse %d ranks on one card"
                         % (world, torch.cuda.device_count(), world))
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
    eng = NeuralCXEngine(K=args.K, H=args.H, L=args.L, drop_p=0.25, lr=1e-4, device=dev, world_size=world, bf16=args.bf16, x6=args.x6)
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
    # What the conditioning buys, on the same line: a short timed window from the cold state -- W warm-up steps, then
    # min(K, 10) timed steps, fenced like the headline window -- BEFORE the preheat.  Reported as `ms_per_step_no_preheat`
    # (never `value`); these steps train the same weights on the same batches, they are not part of the K timed steps.
    ms_cold = None
    if args.preheat_ms > 0 and args.steps > 0:
        nc = min(args.steps, 10)
        for i in range(args.warmup):
            step(i)
        fence()
        tc = time.perf_counter()
        for i in range(nc):
            step(args.warmup + i)
        eng.flush()
        fence()
        tcold = torch.tensor([time.perf_counter() - tc], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(tcold, op=torch.distributed.ReduceOp.MAX)
        ms_cold = dict(ms_per_step=round(float(tcold.item()) / nc * 1e3, 4), steps=nc, warmup=args.warmup,
                       note="the same fenced measurement from the cold state, taken before the preheat and before the headline window")
        eng.comm_events = []
    # The HIP events of the per-kernel timing are created BEFORE the warm-up (creating 2 x (4 K + 8) events takes the host
    # long enough for an idle chip to drop its clocks between warm-up and the timed region); the warm-up launches are
    # recorded too and dropped below.
    if rank == 0:
        _lib.profile_begin(["MAIN", "DW1C"], max_launches=4 * (args.steps + args.warmup) + 8)
    if args.preheat_ms > 0:             # device conditioning, not workload steps: reported as `preheat_ms`
        xa = torch.randn(4096, 4096, device=dev); xb = torch.randn(4096, 4096, device=dev)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + args.preheat_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(4):
                xc = xa @ xb
            torch.cuda.synchronize()
        del xa, xb, xc
    for i in range(args.warmup):
        r = step(i)
    eng.flush()                                     # (data parallelism: the deferred tail of the last warm-up step stays outside the timed region)
    fence()
    eng.comm_profile = world > 1                    # (events around the two waits for the gradient exchange: the exposed part of it)
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = step(args.warmup + i)
    eng.flush()                                     # ... and the K-th timed step's tail (wait for its last bucket + its Adam slice) inside it
    fence()
    dt = time.perf_counter() - t0
    eng.comm_profile = False
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(r["loss"].item())
    if world > 1:                                    # (every rank holds loss * B_local / B_global of its slice)
        lt = torch.tensor([loss], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(lt)
        loss = float(lt.item())
    # HIP events of the timed launches are read BEFORE anything else launches the same kernels (round 2's line summed the
    # MAIN launches of the 4 evaluation batches below into the timed steps' total: 0.344 ms reported for a 0.30 ms kernel)
    prof = _lib.profile_end() if rank == 0 else None
    if prof:                                        # drop the warm-up steps' launches (MAIN: 1 per step, DW1C: 1 or 2 per step)
        for k, v in prof.items():
            per_step = len(v) // (args.steps + args.warmup) if (args.steps + args.warmup) else 0
            prof[k] = v[per_step * args.warmup:]
    # Recall@1/@5 (the other half of BASELINE.json's metric) of the weights after warmup + steps, on held-out triplets of the
    # same planted synthetic distribution (rank 0's replica; replicas are identical)
    rec = None
    if rank == 0 and args.heldout > 0 and mutan is None:
        hd = SyntheticCX(n_triplets=args.heldout, K=args.K, n_img=args.n_img, seed=4321, device=dev, feats=data.feats)
        hits = torch.zeros(2, dtype=torch.int64, device=dev)
        hb = 256
        for lo in range(0, args.heldout, hb):
            b, gt = hd.batch(torch.arange(lo, min(lo + hb, args.heldout)))
            ev = eng.eval_step(b, gt)
            hits += ev["hits"].long()
        rec = (float(hits[0]) / args.heldout, float(hits[1]) / args.heldout)

    if rank == 0:
        c = eng.cfg
        M = args.batch * c["K"]
        # Diagnostic pass AFTER the timed region (never inside it): the shader clock the chip holds while the step runs
        # back to back, from in-kernel stamps of MAIN (ncx_profile_stamps: s_memtime / s_memrealtime at entry and exit of
        # every workgroup of the last of `nd` further steps).  Ranks > 0 of a multi-GPU run idle at the final barrier.
        clock = None
        if not args.bf16 and world == 1:
            nd = 12
            st = torch.zeros(16 * 8192, dtype=torch.int64, device=dev)
            _lib.profile_stamps(st)
            try:                                    # (the stamp buffer must never stay armed past its tensor's life)
                for i in range(nd):
                    step(args.warmup + args.steps + i)
                torch.cuda.synchronize()
            finally:
                _lib.profile_stamps(None)
            w = st.view(-1, 16).cpu()
            w = w[(w[:, 15] > w[:, 14]) & (w[:, 8] > w[:, 0])]
            if w.shape[0]:
                mhz = (w[:, 8] - w[:, 0]).double() / (w[:, 15] - w[:, 14]).double() * 100.0
                dur = (w[:, 15] - w[:, 14]).double() / 100.0
                clock = dict(sclk_mhz=round(float(mhz.median()), 1), sclk_mhz_min=round(float(mhz.min()), 1), sclk_mhz_max=round(float(mhz.max()), 1),
                             workgroups=int(w.shape[0]), workgroup_us_median=round(float(dur.median()), 1),
                             kernel_span_us=round(float(w[:, 15].max() - w[:, 14].min()) / 100.0, 1),
                             how="in-kernel stamps of the last of %d back-to-back diagnostic steps after the timed region: "
                                 "d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup, median" % nd)
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
                 "DW1C": "linear_1 weight gradient (all columns + dGt): k_dw_km8 (v_other + v_mult columns in one MFMA pass, per-triplet "
                         "fold, 8 k-chunks, one 8-wave workgroup per CU on 256 x 64 tile pairs; other shapes: k_dw_km) + k_dw_tn8 (dGt and every other column block: one balanced launch of 8-wave workgroups on 256 x 64 "
                         "tiles; shapes it does not cover: seg_gemm TN %s grouped) + their merged fixed-order reduction"
                         % plans["DW1C"]["tile"]}
        peak = PEAK_F32_MFMA_TFLOPS
        if args.x6:
            x6note = (" [--x6 (NOT the headline): %s -- three bf16 planes per fp32 operand, six v_mfma_f32_16x16x32_bf16 per 16x16x32 block, fp32 accumulate; "
                      "achieved / frac stay fp32-EQUIVALENT flops against the fp32-MFMA peak (the kernels are bound by LDS + load-return traffic, not by the matrix cores: DESIGN 5d)]")
            names["MAIN"] += x6note % "k_main_fwd<..., X6>"
            names["DW1C"] += x6note % "k_dw_km_x6 + k_dw_tn8_x6 instead of k_dw_km + k_dw_tn8"
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
        # flops the matrix cores EXECUTE for `flops` (padding of reduction extents to 32; the per-triplet folds run 32
        # (K = 24) or 64 (K = 48) row blocks per triplet for the two 2048-deep segments at once: 4 MFMA row blocks instead of 2 x 3)
        pad32 = lambda n: (n + 31) // 32 * 32
        rest_cols = pad32(c["K"] + 1) + pad32(c["dz"]) + pad32(c["A"])
        fold_fwd = plans["MAIN"]["tile"].startswith(("48x64", "96x64", "192x64"))
        blk = 32 if c["K"] == 24 else 64
        executed = {"MAIN": 2.0 * c["H"] * ((args.batch * blk if fold_fwd else 2 * M) * c["dv"] + M * rest_cols),
                    "DW1C": (2.0 * c["H"] * M * (c["dv"] + c["K"] + 1 + c["dz"] + c["A"]) + 2.0 * args.batch * c["H"] * s_cols) if c["K"] % 24 == 0 and args.batch >= 256
                            else flops["DW1C"]}
        roof = None
        if dom:
            ach = flops[dom] / (per[dom] * 1e-3) / 1e12
            # HBM bytes per launch of that kernel from the committed PMC passes (collected offline as the guide
            # prescribes: separate --pmc runs, FETCH_SIZE x2 on gfx950), per workload
            traffic, traffic_source = None, None
            wl = "c5" if (args.bf16 and (args.batch, c["K"]) == (1024, 48)) else "c3" if args.c3 else \
                 ("x6" if args.x6 else "c2") if (args.batch, c["K"], c["H"], c["L"], args.bf16) == (512, 24, 256, 1, False) else None
            for tname in ("r4_traffic.json", "r3_traffic.json", "r2_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", tname)
                if wl and os.path.exists(tpath):
                    tj = json.load(open(tpath))
                    tj = tj.get(wl, tj if wl == "c2" else {})
                    if tj.get(dom, {}).get("bytes_per_launch"):
                        traffic = tj[dom]["bytes_per_launch"]
                        traffic_source = "profiles/%s: %s" % (tname, tj.get("_source", "rocprofv3 --pmc passes of this workload"))
                        break
            # spread over the timed launches (per step: DW1C sums its two launches)
            nl = max(1, len(prof[dom]) // args.steps)
            series = [sum(prof[dom][i * nl:(i + 1) * nl]) for i in range(args.steps)]
            srt = sorted(series)
            k5 = min(5, len(series))
            roof = dict(bound="mfma", kernel=names[dom], achieved=round(ach, 2), peak=peak,
                        unit="TFLOP/s", frac=round(ach / peak, 4),
                        frac_basis="ALGORITHMIC flops (SURVEY 8d numerator) / launch time / peak; `frac_executed` beside it counts only the flops the matrix cores execute",
                        traffic=traffic, traffic_source=traffic_source,
                        launch_ms=round(per[dom], 4), launch_ms_min=round(srt[0], 4), launch_ms_median=round(srt[len(srt) // 2], 4),
                        launch_ms_max=round(srt[-1], 4), launch_ms_first5=round(sum(series[:k5]) / k5, 4),
                        launch_ms_last5=round(sum(series[-k5:]) / k5, 4), launch_ms_head=[round(x, 4) for x in series[:12]],
                        algorithmic_gflop_per_launch=round(flops[dom] / 1e9, 3),
                        other={k: dict(launch_ms=round(v, 4), launches_per_step=len(prof[k]) // args.steps, tflops=round(flops[k] / (v * 1e-3) / 1e12, 2),
                                       mfma_executed_gflop=round(executed[k] / 1e9, 3), plan=plans[k]) for k, v in per.items()})
            if args.x6:
                # the matrix cores execute SIX bf16 products per fp32-equivalent one: price the line against the dense bf16 peak by executed bf16 flops
                ex6 = 6.0 * executed[dom] / (per[dom] * 1e-3) / 1e12
                roof.update(achieved=round(ex6, 2), peak=PEAK_BF16_MFMA_TFLOPS, frac=round(ex6 / PEAK_BF16_MFMA_TFLOPS, 4),
                            frac_basis="EXECUTED bf16 MFMA flops (6 plane products per fp32-equivalent product, reduction extents padded to 32) / launch time / dense bf16 peak; "
                                       "the kernel is bound by LDS + load-return traffic, not by the matrix cores (DESIGN 5d)",
                            fp32_equivalent_algorithmic_tflops=round(ach, 2), bf16_mfma_executed_gflop=round(6.0 * executed[dom] / 1e9, 3))
            elif not args.bf16:
                ex = executed[dom] / (per[dom] * 1e-3) / 1e12
                roof.update(mfma_executed_gflop=round(executed[dom] / 1e9, 3), executed_tflops=round(ex, 2), frac_executed=round(ex / peak, 4),
                            peak_assumes_mhz=2400)
                if clock:
                    held = peak * clock["sclk_mhz"] / 2400.0
                    roof.update(clock, peak_at_held_clock=round(held, 1), frac_of_peak_at_held_clock=round(ach / held, 4),
                                frac_executed_at_held_clock=round(ex / held, 4))
        out = dict(metric="VQA-CX triplets/sec (24 candidates each), NeuralCX training step",
                   value=round(gb * args.steps / dt, 1), unit="triplets/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, preheat_ms=args.preheat_ms, ms_per_step=round(dt / args.steps * 1e3, 4),
                   ms_per_step_no_preheat=ms_cold, higher_is_better=True,
                   scaling=args.scaling, vs_baseline=None, dtype="bf16 operands / f32 accumulate (first-layer GEMMs), f32 elsewhere" if args.bf16 else
                         "f32 via bf16x6 in the three big kernels (linear_1 forward, its weight gradient: three bf16 planes per fp32 operand, six plane products, f32 accumulate; error vs fp64 = the fp32-MFMA kernels'), f32 elsewhere -- NOT the headline" if args.x6 else "f32",
                   data="synthetic",
                   config=dict(workload=workload_label(args, c) +
                                        "NeuralCX MLP train step (fwd+listwise loss/recall+bwd+Adam), synthetic "
                                        "2048-d feats, %d candidates, batch %d per GPU (%s scaling: global batch %d), H=%d, L=%d, dropout 0.25, %s"
                                        % (c["K"], args.batch, args.scaling, gb, c["H"], c["L"],
                                           "bf16 operands / fp32 accumulate for the linear_1 and answer-embedding products, fp32 elsewhere" if args.bf16 else "fp32"),
                               global_batch=gb, per_rank_batch=args.batch, candidates=c["K"], dim_h=c["H"], n_layers=c["L"],
                               parallelism="dp%d" % world, feature_table_rows=args.n_img, final_loss=round(loss, 5)),
                   roofline=roof)
        if rec is not None:
            out["recall_at_1"], out["recall_at_5"] = round(rec[0], 4), round(rec[1], 4)
            out["recall_note"] = ("HIP path, %d held-out planted synthetic triplets, after %d training steps (chance: 0.0417 / 0.2083)"
                                  % (args.heldout, args.warmup + args.steps))
        if world > 1:
            stall = None
            if eng.comm_events:                      # rank 0's compute stream: time spent waiting for bucket 1 / bucket 2 per step
                w1 = [e[0].elapsed_time(e[1]) for e in eng.comm_events]; w2 = [e[2].elapsed_time(e[3]) for e in eng.comm_events]
                stall = dict(bucket1_ms=round(sum(w1) / len(w1), 4), bucket2_ms=round(sum(w2) / len(w2), 4),
                             note="mean time per step rank 0's compute stream waits for each gradient bucket (what the backward did not hide)")
            if backend == "nccl":
                v = torch.cuda.nccl.version()
                out["rccl"] = dict(nranks=world, version=".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v),
                                   collective="sum all-reduce per step: dGt|dGgt block (%.1f MB) + flat gradient without answer_embedding (%.1f MB)"
                                              % (2 * c["H"] * ((c["A"] + 3) // 4 * 4) * 4 / 1e6, (eng.params.numel - eng.params.offsets["linear_1.weight"]) * 4 / 1e6))
            else:
                out["rccl"] = dict(nranks=world, version=None, note="NCX_DIST_BACKEND=%s rehearsal: NOT RCCL" % backend)
            if stall:
                out["rccl"]["exposed_wait"] = stall
        if not args.no_cpu_baseline and world == 1 and (c["dv"], c["K"]) == (2048, 24):
            out["cpu_baseline"] = cpu_baseline(dev, data.feats, steps=args.cpu_steps, heldout=args.heldout)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
