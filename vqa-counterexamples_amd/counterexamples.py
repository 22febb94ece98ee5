#!/usr/bin/env python3
"""NeuralCX training / evaluation driver -- CLI drop-in for the reference's counterexamples.py.

Keeps the reference's flags (counterexamples.py:39-87), YAML schema (options/cx/*.yaml), seeds (:119-121),
run-directory layout and checkpoint files (`logs/cx/<run>/{ckpt,best}/{model,info}.ckpt`, :550-580) and the
printed metrics (`Epoch e mode: loss: x, recall: y`, :493-498).  The per-batch body of the train loop
(:322-339) and of eval_model (:460-468) runs on the HIP engine (neuralcx.engine.NeuralCXEngine).

Net-new: `--synthetic` (no datasets offline: synthetic feature table + triplets of the real shapes) and data
parallelism: launch with `python -m torch.distributed.run --nproc-per-node N counterexamples.py ...`;
every rank holds the feature table and a replica, gradients are summed over RCCL.

Real-data mode (default, as in the reference) reads what the reference's notebooks produce -- `<vqa.path_trainset>/
pickle_old/{trainset_augmented[_small],valset_augmented_small,valset_augmented}.pickle`, `<coco.path_features>/
{train,val}set.{npy|hdf5}` (hdf5 needs h5py; neuralcx.formats.convert_hdf5_features converts once), optional
`answer_embedding.pickle` and the VQA checkpoint `<logs.dir_logs>/best_model.pth.tar` -- ONCE, keeps the feature tables
and index arrays resident in HBM (neuralcx.formats.CXDeviceDataset) and produces q / z / a per batch with the frozen VQA
model (question encoder in PyTorch, MUTAN fusion + classifier in ncx_vqa_forward).  No real datasets exist offline:
tests drive this mode with files written by neuralcx.formats.write_synthetic_cx_files.
"""
import argparse
import json
import os
import pickle
import random
import shutil
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import vqa.lib.utils as utils                     # noqa: E402
import vqa.models as models                       # noqa: E402
from vqa.models.cx import CXModelBase, blackbox_scores        # noqa: E402
from neuralcx import dp, ops                      # noqa: E402
from neuralcx.engine import NeuralCXEngine        # noqa: E402
from neuralcx.synth import SyntheticCX            # noqa: E402
from neuralcx import formats                      # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="Train/Evaluate NeuralCX counterexample models (MI355X HIP path)")
    p.add_argument("--path_opt", default=os.path.join(HERE, "options", "cx", "neuralcx_256_1_all.yaml"), type=str)
    p.add_argument("--vqa_model", default="mutan_noatt_train", type=str)
    # (the reference makes -cx required, counterexamples.py:44; here NeuralModel is the default and the flag spellings are kept)
    p.add_argument("-cx", "--cx_model", default="NeuralModel", type=str, help="NeuralModel | RandomBaseline | DistanceBaseline | BlackBox")
    p.add_argument("-lb", "--sb_lambda", type=float, help="semantic baseline lambda (counterexamples.py:49; no NeuralModel code path "
                                                          "reads it, in the reference either: accepted and ignored with a warning)")
    p.add_argument("--pairwise", action="store_true")
    p.add_argument("-dev", "--dev_mode", action="store_true", help="small train/val subsets")
    p.add_argument("--pretrained_vqa", dest="pretrained_vqa", action="store_true")
    p.add_argument("--untrained_vqa", dest="pretrained_vqa", action="store_false")
    p.set_defaults(pretrained_vqa=None)
    p.add_argument("--trainable_vqa", action="store_true")
    p.add_argument("-lr", "--learning_rate", type=float, help="initial learning rate")
    p.add_argument("-b", "--batch_size", type=int, help="mini-batch size (global, across all ranks)")
    p.add_argument("--epochs", type=int, help="number of total epochs to run")
    p.add_argument("--resume", type=str, default=None, help="run name to resume")
    p.add_argument("--best", action="store_true", help="resume the best checkpoint")
    p.add_argument("-c", "--comment", type=str, default="")
    p.add_argument("-p", "--print_freq", default=100, type=int)          # counterexamples.py:65
    p.add_argument("-v", "--eval_freq", default=-1, type=int)
    p.add_argument("-t", "--test", action="store_true", help="evaluate the best model on the full validation set")
    p.add_argument("--viz", action="store_true")
    p.add_argument("--project_dir", default=os.getcwd(), type=str)
    # net-new
    p.add_argument("--synthetic", action="store_true", help="synthetic data of the real shapes (no datasets offline)")
    p.add_argument("--syn_train", type=int, default=16384)
    p.add_argument("--syn_val", type=int, default=4096)
    p.add_argument("--syn_images", type=int, default=82783)
    p.add_argument("--max_steps", type=int, default=-1, help="stop an epoch early (smoke runs)")
    p.add_argument("--no_vqa_cache", action="store_true", help="produce q / z / answer logits per batch instead of once per split")
    p.add_argument("--bf16", action="store_true", help="bf16 operands on the two dominant GEMMs (fp32 accumulate / master weights)")
    p.add_argument("--x6", action="store_true", help="NCX_F_X6: the three big fp32 products on the bf16 matrix cores with three-plane fp32-grade operands (same results to fp32 rounding; not the default)")
    p.add_argument("--path_trainset", type=str, default=None, help="overrides vqa.path_trainset of the YAML")
    p.add_argument("--path_features", type=str, default=None, help="overrides coco.path_features / path_raw of the YAML")
    return p


def load_options(args):
    options = {"optim": {"lr": args.learning_rate, "batch_size": args.batch_size, "epochs": args.epochs},
               "cx_model": {"pretrained_vqa": args.pretrained_vqa, "trainable_vqa": args.trainable_vqa or None}}
    with open(args.path_opt) as f:
        from_yaml = yaml.safe_load(f)          # the reference's yaml.load(handle) breaks on PyYAML >= 6
    return utils.update_values(options, from_yaml)


def recall_from_rank(rank, k):
    return (rank < k)


class Runner:
    def __init__(self, args, options):
        self.args, self.opt = args, options
        self.rank, self.world, self.local = dp.init_distributed()
        if not torch.cuda.is_available():
            raise SystemExit("counterexamples.py: an MI355X is required (the HIP path has no CPU fallback)")
        self.local %= max(1, torch.cuda.device_count())          # (rehearsals of several ranks on one card)
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        random.seed(42); torch.manual_seed(42); torch.cuda.manual_seed(42)          # counterexamples.py:119-121
        cx = options["cx_model"]
        fus = options["model"]["fusion"]
        self.K = 24
        self.engine = NeuralCXEngine(K=self.K, dv=fus["dim_v"], dq=fus["dim_q"], dz=fus["dim_mm"], da=2400,
                                     A=options["vqa"]["nans"], H=cx["dim_h"], L=cx["n_layers"], drop_p=cx["drop_p"],
                                     lr=options["optim"]["lr"], device=self.dev,
                                     spec={k: cx.get(k, True) for k in ("v_mult", "v_dist", "v_rank", "a_emb")},
                                     world_size=self.world, bf16=args.bf16, x6=getattr(args, "x6", False))
        self.engine.rank = self.rank
        self.engine.init_parameters(seed=42)
        self.gb = options["optim"]["batch_size"]
        self.baseline = None if args.cx_model == "NeuralModel" else args.cx_model
        self.runs_dir = None

    def log(self, *a):
        if self.rank == 0:
            print(*a, flush=True)

    # ---- data -----------------------------------------------------------------------------------------------
    def load_synthetic(self):
        a = self.args
        fus = self.opt["model"]["fusion"]
        kw = dict(K=self.K, dv=fus["dim_v"], dq=fus["dim_q"], dz=fus["dim_mm"], A=self.opt["vqa"]["nans"], device=self.dev)
        n_tr = 1024 if a.dev_mode else a.syn_train
        self.train = SyntheticCX(n_triplets=n_tr, n_img=a.syn_images, seed=1234, **kw)
        self.val = SyntheticCX(n_triplets=a.syn_val, n_img=a.syn_images, seed=4321, feats=self.train.feats, **kw)
        self.test = self.val
        self.vqa = None

    def load_real(self):
        """counterexamples.py:181-262: pickles, feature tables, VQA model (+ checkpoint), answer embedding -- loaded once
        and kept on the device."""
        a, opt = self.args, self.opt
        vqa_dir = a.path_trainset or opt["vqa"].get("path_trainset")
        feat_dir = a.path_features or opt["coco"].get("path_features") or opt["coco"].get("path_raw")
        if not vqa_dir or not feat_dir:
            raise SystemExit("real-data mode needs vqa.path_trainset and coco.path_features (YAML or --path_trainset/--path_features)")
        pk = lambda fn: formats.load_cx_pickle(os.path.join(vqa_dir, "pickle_old", fn))
        self.log("=> Loading VQA dataset...")
        trainset = pk("trainset_augmented_small.pickle" if a.dev_mode else "trainset_augmented.pickle")
        valset = pk("valset_augmented_small.pickle")
        self.log("=> Loading COCO image features...")
        self.train = formats.CXDeviceDataset(trainset, formats.load_feature_table(feat_dir, "train"), self.dev, self.K)
        self.val = formats.CXDeviceDataset(valset, formats.load_feature_table(feat_dir, "val"), self.dev, self.K)
        self.test = (formats.CXDeviceDataset(pk("valset_augmented.pickle"), None, self.dev, self.K, feats=self.val.feats)
                     if a.test else self.val)
        self.log("=> Building model...")
        self.vqa = models.factory(opt["model"], trainset["vocab_words"], trainset["vocab_answers"], cuda=True, data_parallel=False)
        if opt["cx_model"].get("pretrained_vqa"):
            ck = os.path.join(opt["logs"]["dir_logs"], "best") + "_model.pth.tar"          # train.py:332-357
            if os.path.isfile(ck):
                self.vqa.load_state_dict(torch.load(ck, map_location=self.dev))
            else:
                self.log("Warning: no VQA checkpoint at '{}' (continuing with the untrained VQA model)".format(ck))
        self.vqa.eval()                                                                      # cx.py:73-80 (frozen)
        for p_ in self.vqa.parameters():
            p_.requires_grad_(False)
        self.mutan = ops.MutanWeights(self.vqa) if isinstance(self.vqa, models.MutanNoAtt) else None
        self._torch_vqa = None if self.mutan is not None else CXModelBase(self.vqa, self.K)
        emb = None
        if opt["cx_model"].get("pretrained_emb"):
            pe = os.path.join(vqa_dir, "answer_embedding.pickle")
            if os.path.isfile(pe):
                emb = formats.load_answer_embedding(pe, n_answers=len(trainset["vocab_answers"]))
            else:
                self.log("Warning: no answer embedding at '{}' (random initialisation)".format(pe))
        self.engine.init_parameters(seed=42, emb=emb)
        if not a.no_vqa_cache:                      # frozen VQA model: its outputs are per-example constants
            seen = set()
            for name, ds in (("train", self.train), ("val", self.val), ("test", self.test)):
                if id(ds) in seen:
                    continue
                seen.add(id(ds))
                nbytes = ds.cache_vqa_outputs(lambda img_idx, wids, ds=ds: self._vqa_outputs(ds, img_idx, wids))
                self.log("=> cached VQA outputs of the {} split: {} examples, {:.2f} GB".format(name, ds.N, nbytes / 1e9))

    def _vqa_outputs(self, data, img_idx, wids):
        """q_emb, z_orig, z_knns, a_knns of the frozen VQA model for a block of examples (vqa_forward, cx.py:64-104)."""
        with torch.no_grad():
            if self.mutan is not None:
                q = self.vqa.seq2vec(wids).float().contiguous()
                _, z_o, a_k, z_k = ops.vqa_forward(data.feats, img_idx, q, self.mutan, want_a_orig=False)
                return q, z_o, z_k, a_k
            _, z_o, a_k, z_k, q = self._torch_vqa.vqa_forward(data.dense_features(img_idx), wids)
            return q.float().contiguous(), z_o.float().contiguous(), z_k.float().contiguous(), a_k.float().contiguous()

    def get_batch(self, data, sel, first_id):
        """-> (ops.Batch, gt) for the device tensor `sel` of triplet ids.  Synthetic: generated on device.  Real: index
        slices of the resident tables; q / z / a come from the per-split cache of the frozen VQA model's outputs, or
        (--no_vqa_cache) are produced per batch."""
        if self.vqa is None:
            return data.batch(sel, first_id=first_id)
        img_idx, wids, aids, gt = data.batch_indices(sel)
        q, z_o, z_k, a_k = data.cached_vqa(sel) if data.vqa_cache is not None else self._vqa_outputs(data, img_idx, wids)
        return ops.Batch(data.feats, img_idx, q, z_o, z_k, a_k, aids), gt

    # ---- loops ----------------------------------------------------------------------------------------------
    def run_epoch(self, epoch):
        eng, tr = self.engine, self.train
        ids, plan = dp.epoch_plan(tr.N, self.gb, epoch, self.rank, self.world, self.dev, seed=42)   # one H2D copy per epoch
        t0 = time.time(); seen = 0
        acc = torch.zeros(3, dtype=torch.float64, device=self.dev)        # loss*B_local, hits5, count (no host sync)
        for bi, (lo, hi, n_global, first_id, active) in enumerate(plan):
            if 0 <= self.args.max_steps <= bi:
                break
            # (a rank with an empty slice of a short last batch runs a zero-weight padding triplet: every rank enters
            #  every collective below -- dp.epoch_plan)
            b, gt = self.get_batch(tr, ids[lo:hi], first_id)
            r = eng.train_step(b, gt, global_batch=n_global, active=active)
            acc[0] += r["loss"][0].double() * n_global; acc[1] += r["hits"][1].double(); acc[2] += (hi - lo) if active else 0
            seen += n_global
            if (bi + 1) % self.args.print_freq == 0:
                l, _, h5, n = dp.reduce_metrics(float(acc[0]), 0, int(acc[1]), int(acc[2]), self.dev)
                self.log("Epoch {} train: loss: {:.4f}, recall: {:.4f}, triplets/s: {:.0f}".format(
                    epoch, l / max(n, 1), h5 / max(n, 1), seen / (time.time() - t0)))
                self.scalars("train", epoch, (epoch - 1) * len(plan) + bi + 1,
                             dict(loss=l / max(n, 1), recall=h5 / max(n, 1), triplets_per_s=seen / (time.time() - t0)))
                acc.zero_()
            if self.args.eval_freq > 0 and (bi + 1) % self.args.eval_freq == 0:
                self.report("val", epoch, self.evaluate(self.val))
        eng.flush()                                  # (data parallelism: the last step's deferred gradient bucket + Adam slice)
        torch.cuda.synchronize()
        return seen / (time.time() - t0)

    def evaluate(self, data):
        eng = self.engine
        tot = torch.zeros(4, dtype=torch.float64, device=self.dev)
        ids, plan = dp.epoch_plan(data.N, self.gb, 0, self.rank, self.world, self.dev, shuffle=False)
        for lo, hi, n_global, first_id, active in plan:
            if not active:                              # (no collective inside this loop: skipping is safe)
                continue
            b, gt = self.get_batch(data, ids[lo:hi], first_id)
            r = eng.eval_step(b, gt) if self.baseline is None else self.baseline_step(b, gt)
            tot[0] += r["loss_rows"].double().sum() * (hi - lo); tot[1] += r["hits"][0]; tot[2] += r["hits"][1]; tot[3] += hi - lo
        l, h1, h5, n = dp.reduce_metrics(float(tot[0]), int(tot[1]), int(tot[2]), int(tot[3]), self.dev)
        return {"loss": l / n, "recall": h5 / n, "recall_1": h1 / n, "recall_5": h5 / n}

    def baseline_step(self, b, gt):
        """The reference's non-neural scorers (cx.py:20-44,114-136) through the same on-device loss / Recall kernel."""
        B, K = b.img_idx.shape[0], self.K
        if self.baseline == "RandomBaseline":
            scores = torch.rand(B, K, device=self.dev)
        elif self.baseline == "DistanceBaseline":
            scores = torch.arange(K - 1, -1, -1, dtype=torch.float32, device=self.dev).repeat(B, 1)
        else:
            scores = blackbox_scores(b.a_knns, b.answer_aids).contiguous()
        r = ops.ranking_loss(scores, gt, want_grad=False)
        r["scores"] = scores
        return r

    def report(self, mode, epoch, metrics, step=None):
        self.log("Epoch {} {}: {}".format(epoch, mode, "".join("{}: {:.4f}, ".format(k, v) for k, v in metrics.items())))
        self.scalars(mode, epoch, step, metrics)

    def scalars(self, mode, epoch, step, metrics):
        """The reference's log_results also feeds tensorboard writers under runs/<run>/{train,val} (counterexamples.py:
        168-169,493-498); the package is not available offline, so the same scalars go to runs/<run>/<mode>.jsonl."""
        if self.rank != 0 or self.runs_dir is None:
            return
        os.makedirs(self.runs_dir, exist_ok=True)
        with open(os.path.join(self.runs_dir, "%s.jsonl" % ("val" if mode == "test" else mode)), "a") as f:
            f.write(json.dumps(dict(mode=mode, epoch=epoch, step=step, **{k: float(v) for k, v in metrics.items()})) + "\n")

    # ---- checkpoints (counterexamples.py:550-580) -------------------------------------------------------------
    def save(self, save_dir, info, is_best):
        """Rank 0 writes; every rank waits for the files (a following --test / --resume load on another rank must not
        race the write)."""
        if self.rank == 0:
            self._save(save_dir, info, is_best)
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    def _save(self, save_dir, info, is_best):
        os.makedirs(os.path.join(save_dir, "ckpt"), exist_ok=True); os.makedirs(os.path.join(save_dir, "best"), exist_ok=True)
        pm, pi = os.path.join(save_dir, "ckpt", "model.ckpt"), os.path.join(save_dir, "ckpt", "info.ckpt")
        state = {k: v.cpu() for k, v in self.engine.state_dict().items()}
        if self.vqa is not None:                  # the reference's state_dict embeds the VQA model (cx.py:56)
            state.update({"vqa_model." + k: v.cpu() for k, v in self.vqa.state_dict().items()})
        po = os.path.join(save_dir, "ckpt", "optim.ckpt")          # net-new: Adam moments + step (the reference restarts Adam on --resume)
        torch.save(state, pm)
        torch.save(info, pi)
        torch.save(self.engine.optimizer_state(), po)
        if is_best:
            for src in (pm, pi, po):
                shutil.copyfile(src, os.path.join(save_dir, "best", os.path.basename(src)))
        self.log("{}Saved checkpoint to {}".format("* " if is_best else "", save_dir))

    def load(self, save_dir, best):
        sub = "best" if best else "ckpt"
        state = torch.load(os.path.join(save_dir, sub, "model.ckpt"), map_location="cpu")
        self.engine.load_state({k: v for k, v in state.items() if not k.startswith("vqa_model.")})
        info = torch.load(os.path.join(save_dir, sub, "info.ckpt"))
        assert len(info) > 0
        po = os.path.join(save_dir, sub, "optim.ckpt")
        if os.path.isfile(po):                      # (absent in checkpoints written by the reference: Adam restarts, as there)
            self.engine.load_optimizer_state(torch.load(po, map_location="cpu"))
        last = info[-1]
        return info, len(info) + 1, last.get("recall_5", last.get("recall"))     # the reference KeyErrors here (:580)


def main(argv=None):
    args = build_parser().parse_args(argv)
    options = load_options(args)
    if args.cx_model not in ("NeuralModel", "RandomBaseline", "DistanceBaseline", "BlackBox"):
        raise SystemExit("--cx_model {}: only NeuralModel and the RandomBaseline / DistanceBaseline / BlackBox scorers are "
                         "provided (the reference's other models are outside the accelerated path)".format(args.cx_model))
    if args.sb_lambda is not None:
        print("warning: -lb/--sb_lambda {} is accepted for command-line compatibility and ignored (no {} code path reads it)".format(
            args.sb_lambda, args.cx_model), file=sys.stderr)
    if args.pairwise or args.viz:
        raise SystemExit("--pairwise / --viz are outside the accelerated path (SURVEY 8: out of scope)")
    r = Runner(args, options)
    run = args.resume or "{}_{}".format(options["cx_model"].get("name", "neuralcx"), time.strftime("%m%d_%H%M%S"))
    save_dir = os.path.join(args.project_dir, "logs", "cx", run)
    r.runs_dir = os.path.join(args.project_dir, "runs", run)
    if args.synthetic:
        r.load_synthetic()
    else:
        r.load_real()
    info, start_epoch, best_recall = [], 1, 0.0
    if args.resume:
        info, start_epoch, best_recall = r.load(save_dir, args.best)
    if r.baseline is not None:            # nothing to train (optimizer is None in the reference, :336): evaluate and report
        t0 = time.time()
        res = r.evaluate(r.test if args.test else r.val)
        torch.cuda.synchronize()
        n = (r.test if args.test else r.val).N
        r.report("test" if args.test else "val", 1, res)
        r.log("{}: {} triplets in {:.2f} s ({:.0f} triplets/s)".format(r.baseline, n, time.time() - t0, n / (time.time() - t0)))
        if r.rank == 0 and args.test:
            os.makedirs(save_dir, exist_ok=True)
            with open(os.path.join(save_dir, "final_results.txt"), "w") as f:
                json.dump(res, f)
        if torch.distributed.is_initialized():
            torch.distributed.barrier(); torch.distributed.destroy_process_group()
        return res
    r.log("=> Starting training... ({} GPU(s), global batch {}, {} train / {} val triplets)".format(
        r.world, r.gb, r.train.N, r.val.N))
    for epoch in range(start_epoch, options["optim"]["epochs"] + 1):
        tps = r.run_epoch(epoch)
        res = r.evaluate(r.val)
        r.report("val", epoch, res)
        r.log("Epoch {} throughput: {:.0f} triplets/s".format(epoch, tps))
        info.append(res)
        is_best = res["recall"] > best_recall
        best_recall = max(best_recall, res["recall"])
        r.save(save_dir, info, is_best)
    if args.test:
        r.load(save_dir, best=True)
        res = r.evaluate(r.test)
        r.report("test", len(info), res)
        if r.rank == 0:
            with open(os.path.join(save_dir, "final_results.txt"), "w") as f:
                json.dump(res, f)
    if torch.distributed.is_initialized():
        torch.distributed.barrier(); torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
