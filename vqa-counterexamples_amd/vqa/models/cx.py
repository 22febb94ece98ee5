"""Counterexample models -- drop-in for the reference's vqa/models/cx.py plugin surface.

`NeuralModel` keeps the reference's constructor (cx.py:220-224), `forward(image_features, question_wids,
answer_aids) -> scores[B, K]` (cx.py:261), the mutable `knn_size` attribute, `vqa_forward` (cx.py:64-104) and
the state_dict keys (`answer_embedding.weight`, `linear_1.*`, `linear_2.*`, `linear_3.*`, `out.*`,
`vqa_model.*`; linear_1.weight keeps the concat column order of cx.py:309-320), so checkpoints and calling
code are interchangeable.  Everything below `vqa_forward` -- the answer-embedding products, the 24-way feature
synthesis/concat, the Linear+ReLU+Dropout stack, `out`, and the whole backward -- runs in the HIP library
(libneuralcx_hip.so) through `neuralcx.ops.NeuralCXFunction`.  There is no PyTorch fallback for that part:
on a machine without the library or without a GPU `forward` raises.

Deliberate differences from the reference (all supersets):
  * knn_size may be 1..64 (the reference asserts == 24, cx.py:226); config 5 of BASELINE.json uses 48;
  * `trainable_vqa=True` is rejected (input gradients are not produced; the reference's default and every
    options/cx/*.yaml use a frozen VQA model, cx.py:73-80);
  * lesion combination q_emb=z_emb=False with a_emb=True raises a clear error (NameError in the reference);
  * an answer id outside [0, ans_size) raises IndexError like nn.Embedding (cx.py:280), but by default NOT at the offending
    call: the verdict is read without a host sync and surfaces at the next point that syncs anyway -- the next forward
    whose flag copy has landed, `train()` / `eval()` (the reference switches modes around every evaluation and epoch),
    `state_dict()` (every checkpoint save), or `check_answer_ids()`.  `strict_ids=True` (constructor keyword or attribute)
    restores the immediate raise at the price of one host sync per forward; on a CPU device the check is always immediate.
"""
import torch
import torch.nn as nn

from neuralcx import ops

DIM_A = 2400      # cx.py:235


class RandomBaseline(nn.Module):
    """Uniform random scores: Recall@k ~ k / knn_size (reference cx.py:20-30; README 4.20 / 20.85)."""

    def __init__(self, knn_size):
        super().__init__()
        self.knn_size = knn_size

    def forward(self, image_features, question_wids, answer_aids):
        return torch.rand(image_features.size(0), self.knn_size, device=image_features.device)


class DistanceBaseline(nn.Module):
    """Scores knn_size-1 .. 0 for every row: Recall@k == fraction with knn_index < k (reference cx.py:33-44)."""

    def __init__(self, knn_size):
        super().__init__()
        self.knn_size = knn_size

    def forward(self, image_features, question_wids, answer_aids):
        s = torch.arange(self.knn_size - 1, -1, -1, dtype=torch.float32, device=image_features.device)
        return s.view(1, -1).expand(image_features.size(0), self.knn_size).contiguous()


class CXModelBase(nn.Module):
    def __init__(self, vqa_model, knn_size, trainable_vqa=False):
        super().__init__()
        self.vqa_model = vqa_model
        self.trainable_vqa = trainable_vqa
        if vqa_model is not None and not trainable_vqa:
            self.vqa_model.eval()
        self.knn_size = knn_size

    def _hip_vqa_ok(self, image_features):
        from .noatt import MutanNoAtt
        return (isinstance(self.vqa_model, MutanNoAtt) and image_features.is_cuda and not self.trainable_vqa
                and getattr(self, "use_hip_vqa", True))

    @torch.no_grad()
    def vqa_forward(self, image_features, question_wids):
        """Frozen VQA model on the original + K candidate images -> a_orig, z_orig, a_knns, z_knns, q_emb
        (same outputs as cx.py:64-104).  For MutanNoAtt on the GPU everything below the question encoder runs in
        the HIP library (ncx_vqa_forward: gather + linear_v + tanh, R-term fusion folded in one chained GEMM,
        classifier); otherwise plain PyTorch.  Either way the question branch is computed once per question
        (the reference duplicates q K+1 times first, cx.py:83-87)."""
        assert image_features.size(1) == self.knn_size + 1
        B, K1 = image_features.size(0), self.knn_size + 1
        vqa = self.vqa_model
        vqa.eval()
        q_emb = vqa.seq2vec(question_wids)
        if self._hip_vqa_ok(image_features):
            mw = self.__dict__.get("_mutan_weights")
            if mw is None or mw.t["wv"].device != image_features.device:
                mw = ops.MutanWeights(vqa)
                self.__dict__["_mutan_weights"] = mw
            feats = image_features.reshape(B * K1, -1).float().contiguous()
            idx = torch.arange(B * K1, device=feats.device, dtype=torch.int32).view(B, K1)
            a_o, z_o, a_k, z_k = ops.vqa_forward(feats, idx, q_emb.float().contiguous(), mw, want_a_orig=True)
            return a_o, z_o, a_k, z_k, q_emb
        v = image_features.reshape(B * K1, -1)
        q_dup = q_emb.view(B, 1, -1).expand(B, K1, q_emb.size(-1)).reshape(B * K1, -1)
        z = vqa._fusion(v, q_dup)
        a = vqa._classif(z)
        a, z = a.view(B, K1, -1), z.view(B, K1, -1)
        return (a[:, 0].contiguous(), z[:, 0].contiguous(), a[:, 1:].contiguous(), z[:, 1:].contiguous(), q_emb)

    def refresh_vqa_weights(self):
        """Call after loading a VQA checkpoint: the stacked MUTAN weights of the HIP path are cached."""
        self.__dict__.pop("_mutan_weights", None)

    def forward(self, image_features, question_wids, answer_aids):
        raise NotImplementedError


def blackbox_scores(a_knns, answer_aids):
    """-softmax(a_knns)[b, k, answer_aids[b]]: the VQA model's own probability of the original answer on each
    candidate, sign-flipped because the likeliest images are the worst counterexamples (cx.py:122-136)."""
    aid = answer_aids.long().view(-1, 1, 1).expand(-1, a_knns.size(1), 1)
    return -(a_knns.gather(2, aid).squeeze(2) - torch.logsumexp(a_knns, dim=2)).exp()


class BlackBox(CXModelBase):
    """Scores candidates with the frozen VQA model alone (reference cx.py:114-136; README row 'Hard negative mining')."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.vqa_model.eval()

    def forward(self, image_features, question_wids, answer_aids):
        _, _, a_knns, _, _ = self.vqa_forward(image_features, question_wids)
        return blackbox_scores(a_knns, answer_aids)


class NeuralModel(CXModelBase):
    def __init__(self, model_spec, dim_h, n_layers, emb, drop_p, **kwargs):
        super().__init__(vqa_model=kwargs["vqa_model"], knn_size=kwargs["knn_size"],
                         trainable_vqa=kwargs.get("trainable_vqa", False))
        if self.trainable_vqa:
            raise NotImplementedError("trainable_vqa=True is not supported by the HIP path (frozen VQA model only)")
        if not 3 <= self.knn_size <= 64:         # (the reference asserts knn_size == 24, cx.py:226; the kernels' 16-byte
            raise ValueError("knn_size must be in 3..64")     #  row windows need K + 1 >= 4 columns in the dist | rank block)
        if n_layers not in (1, 2, 3):
            raise ValueError("n_layers must be 1, 2 or 3")
        self.model_spec = dict(model_spec)
        fus = self.vqa_model.opt["fusion"]
        self.dim_v, self.dim_q, self.dim_z = fus["dim_v"], fus["dim_q"], fus["dim_mm"]
        self.ans_size = len(self.vqa_model.vocab_answers)
        self.dim_a = DIM_A
        self.dim_h, self.n_layers, self.drop_p = dim_h, n_layers, drop_p
        self._built_knn = self.knn_size

        self.answer_embedding = nn.Embedding(self.ans_size, self.dim_a)
        if emb is not None:
            assert emb.shape[1] == self.dim_a
            self.answer_embedding.weight.data = torch.as_tensor(emb, dtype=torch.float32)
        input_size = self.dim_v * 3 + self.dim_a * 2 + self.dim_z * 2 + self.dim_q + self.knn_size + 1
        self.linear_1 = nn.Linear(input_size, dim_h)
        if n_layers >= 2:
            self.linear_2 = nn.Linear(dim_h, dim_h)
        if n_layers >= 3:
            self.linear_3 = nn.Linear(dim_h, dim_h)
        self.out = nn.Linear(dim_h, 1)
        self.relu = nn.ReLU()
        self.drop = nn.Dropout(p=drop_p)
        self._pool = ops.WorkspacePool()
        self._step = 0
        self.dropout_seed = 42
        self.strict_ids = bool(kwargs.get("strict_ids", False))     # True: bad answer ids raise at the offending forward (host sync)

    # ---- the HIP hot path ---------------------------------------------------------------------------------
    def _param_fields(self):
        p = {"answer_embedding": self.answer_embedding.weight, "w1": self.linear_1.weight, "b1": self.linear_1.bias,
             "w_out": self.out.weight, "b_out": self.out.bias}
        if self.n_layers >= 2:
            p["w2"], p["b2"] = self.linear_2.weight, self.linear_2.bias
        if self.n_layers >= 3:
            p["w3"], p["b3"] = self.linear_3.weight, self.linear_3.bias
        return p

    def score_batch(self, batch: "ops.Batch"):
        """scores[B, K] for a device-resident ops.Batch (feature table + row indices: the gather of
        counterexamples.py:540-541 happens inside the kernels)."""
        if batch.img_idx.shape[1] != self._built_knn + 1:
            raise ValueError("linear_1 was built for knn_size=%d" % self._built_knn)
        fields = self._param_fields()
        names = tuple(fields)
        tensors = [fields[n] for n in names]
        for t in tensors:
            if not t.is_cuda:
                raise ops._lib.NcxError("NeuralModel parameters must live on the GPU: call .cuda() (no CPU fallback)")
        self._step += 1
        d = ops.make_dims(batch, H=self.dim_h, L=self.n_layers, da=self.dim_a, A=self.ans_size,
                          flags=ops.flags_from_spec(self.model_spec), training=self.training,
                          drop_p=self.drop_p if self.training else 0.0, seed=(self.dropout_seed << 32) ^ self._step)
        record = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
        call = dict(dims=d, batch=batch, names=names, pool=self._pool, record=record)      # per-call snapshot (re-entrant: see ops.WorkspacePool)
        return ops.NeuralCXFunction.apply(call, *tensors)

    _aid_host, _aid_event, _aid_pending, _aid_flag = None, None, False, None

    def check_answer_ids(self, wait=True):
        """Raises IndexError if the previous forward saw an answer id outside [0, ans_size) (what nn.Embedding raises at
        cx.py:280).  wait=False: only if the flag's copy has already landed (never blocks the host)."""
        if not self._aid_pending:
            return
        if self._aid_event is not None:
            if not wait and not self._aid_event.query():
                return
            self._aid_event.synchronize()
        self._aid_pending = False
        if bool(self._aid_host[0]):
            self._aid_flag.zero_()                                  # (sticky on the device until reported)
            raise IndexError("answer_aids outside [0, %d) in a previous forward" % self.ans_size)

    # The deferred verdict is collected wherever the caller synchronises anyway: a mode switch (the reference calls
    # cx_model.eval() / .train() around eval_model and at every epoch, counterexamples.py:320,451) and state_dict() (checkpoint
    # save, counterexamples.py:555) -- so a bad id in the last forward of an epoch or of an evaluation pass cannot be scored
    # silently and then checkpointed.
    def train(self, mode=True):
        self.check_answer_ids(wait=True)
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        self.check_answer_ids(wait=True)
        return super().state_dict(*args, **kwargs)

    def forward(self, image_features, question_wids, answer_aids):
        spec = self.model_spec
        B = image_features.size(0)
        assert image_features.size(1) == self.knn_size + 1                     # cx.py:263
        K, dev = self.knn_size, image_features.device
        if not spec.get("v_emb", True):                                         # cx.py:265-266
            image_features = torch.rand(B, K + 1, self.dim_v, device=dev)
        if spec.get("q_emb", True) or spec.get("z_emb", True):                  # cx.py:270-271
            a_orig, z_orig, a_knns, z_knns, q_emb = self.vqa_forward(image_features, question_wids)
        elif spec.get("a_emb", True):
            raise ValueError("model_spec with q_emb=z_emb=False needs a_emb=False (a_knns would be undefined)")
        if not spec.get("q_emb", True):                                         # cx.py:272-277
            q_emb = torch.rand(B, self.dim_q, device=dev)
        if not spec.get("z_emb", True):
            z_orig = torch.rand(B, self.dim_z, device=dev)
            z_knns = torch.rand(B, K, self.dim_z, device=dev)
        extra = {}
        if not spec.get("a_emb", True):                                         # cx.py:283-285
            a_knns = torch.rand(B, K, self.dim_a, device=dev)
            extra["a_emb_gt"] = torch.rand(B, self.dim_a, device=dev)
        if not spec.get("v_rank", True):                                        # cx.py:306-307
            extra["v_rank"] = torch.rand(B, K, K, device=dev)
        if spec.get("a_emb", True) and answer_aids.numel():
            # nn.Embedding raises on a bad index (cx.py:280); the kernels gather / scatter embedding rows by it.  No host
            # sync per forward: the ids are clamped for the kernels, the verdict goes to a device flag whose copy to pinned
            # host memory is read at the NEXT forward (long complete by then) or by check_answer_ids() -- the error of
            # step n surfaces as soon as its copy has landed (checked at every later forward) or when the caller asks.  The flag is
            # STICKY on the device (or-ed across forwards, cleared only when reported): the host runs ahead of the GPU, so the
            # copy of step n has usually not landed when step n + 1 is enqueued, and a later copy must still carry step n's verdict.
            self.check_answer_ids(wait=False)
            bad = ((answer_aids < 0) | (answer_aids >= self.ans_size)).any().view(1)
            if (self.strict_ids or dev.type != "cuda") and bool(bad):      # immediate raise, before anything is launched
                raise IndexError("answer_aids outside [0, %d)" % self.ans_size)
            answer_aids = answer_aids.clamp(0, self.ans_size - 1)
            if self._aid_host is None or self._aid_flag is None or self._aid_flag.device != dev:
                self._aid_host = torch.zeros(1, dtype=torch.bool).pin_memory() if dev.type == "cuda" else torch.zeros(1, dtype=torch.bool)
                self._aid_flag = torch.zeros(1, dtype=torch.bool, device=dev)
            self._aid_flag |= bad
            self._aid_host.copy_(self._aid_flag, non_blocking=True)
            if dev.type == "cuda":
                self._aid_event = torch.cuda.Event()
                self._aid_event.record(torch.cuda.current_stream(dev))
            self._aid_pending = True
        batch = ops.Batch.from_dense(image_features.float(), q_emb.float(), z_orig.float(), z_knns.float(),
                                     a_knns.float(), answer_aids, **extra)
        return self.score_batch(batch)
