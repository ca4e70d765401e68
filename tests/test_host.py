"""CPU tests of the host-side logic around the accelerated path: collation, command-line flags,
checkpoint key handling, batch sharding and the data-parallel gradient reducer (gloo, 2 ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import transformer_np as O  # noqa: E402
from specialk_amd.core import constants  # noqa: E402
from specialk_amd.core.dataset import TranslationDataset, collate_fn, paired_collate_fn  # noqa: E402


def test_constants_match_reference_ids():
    # core/constants.py:1-10 of the reference
    assert (constants.PAD, constants.UNK, constants.SOS, constants.EOS) == (0, 1, 2, 3)
    assert (constants.PAD_WORD, constants.UNK_WORD, constants.SOS_WORD, constants.EOS_WORD) == \
        ("<blank>", "<unk>", "<s>", "</s>")


def test_collate_matches_oracle_on_ragged_batches():
    rng = np.random.default_rng(3)
    for lens in ([1], [5, 2, 9], [7, 7, 7], list(rng.integers(1, 40, size=17))):
        insts = [list(rng.integers(4, 50, size=int(n))) for n in lens]
        seq, pos = collate_fn(insts)
        oseq, opos = O.collate(insts)
        assert seq.dtype == torch.int64 and pos.dtype == torch.int64
        np.testing.assert_array_equal(seq.numpy(), oseq)
        np.testing.assert_array_equal(pos.numpy(), opos)
    s = paired_collate_fn([([2, 5, 3], [2, 6, 7, 3]), ([2, 3], [2, 3])])
    assert [tuple(t.shape) for t in s] == [(2, 3), (2, 3), (2, 4), (2, 4)]
    assert s[1].tolist() == [[1, 2, 3], [1, 2, 0]]


def test_dataset_protocol():
    w2i = {"<blank>": 0, "<unk>": 1, "<s>": 2, "</s>": 3, "a": 4}
    ds = TranslationDataset(w2i, w2i, src_insts=[[2, 4, 3]], tgt_insts=[[2, 4, 4, 3]])
    assert len(ds) == 1 and ds.src_vocab_size == 5 and ds[0] == ([2, 4, 3], [2, 4, 4, 3])
    assert ds.tgt_idx2word[4] == "a"
    with pytest.raises(AssertionError):
        TranslationDataset(w2i, w2i, src_insts=[[2, 3]], tgt_insts=[[2, 3], [2, 3]])


def test_train_flags_and_defaults():
    from specialk_amd.train import load_args
    o = load_args().parse_args(["-data", "d.pt", "-model", "transformer", "-epochs", "3"])
    expect = dict(batch_size=64, layers=6, d_word_vec=300, dropout=0.1, d_model=512, d_inner_hid=2048, d_k=64,
                  d_v=64, n_head=8, n_warmup_steps=4000, embs_share_weight=False, proj_share_weight=False,
                  label_smoothing=False, save_mode="all", learning_rate=0.001, optim="adam", max_grad_norm=5,
                  freeze_encoder=False, freeze_decoder=False, checkpoint_encoder="", checkpoint_decoder="",
                  multi_gpu=False, cuda_device=None, dtype="fp32", rnn_size=500, brnn_merge="concat",
                  start_decay_at=8, learning_rate_decay=0.5, max_generator_batches=32, input_feed=0)
    for k, v in expect.items():
        assert getattr(o, k) == v, k
    with pytest.raises(SystemExit):
        load_args().parse_args(["-model", "transformer", "-epochs", "3"])          # -data is required
    with pytest.raises(SystemExit):
        load_args().parse_args(["-data", "d", "-model", "gru", "-epochs", "3"])    # unknown model


def test_translate_flags_and_defaults():
    from specialk_amd.translate import load_args
    o = load_args(["-model", "transformer", "-src", "s", "-vocab", "v", "-checkpoint_encoder", "e",
                   "-checkpoint_decoder", "d"])
    assert (o.batch_size, o.beam_size, o.n_best, o.output) == (128, 5, 1, "predictions.txt")
    assert o.save_model is False and o.new_directory is False and o.override_max_token_seq_len is None


def test_module_prefix_is_stripped():
    from specialk_amd.lib.TransformerModel import _strip_module_prefix
    sd = {"module.a.weight": 1, "module.b": 2}
    assert list(_strip_module_prefix(sd)) == ["a.weight", "b"]
    assert _strip_module_prefix({"a": 1}) == {"a": 1}


def test_decode_dict_drops_eos_and_accepts_either_direction():
    from specialk_amd.lib.nmtModel import NMTModel
    w2i = {"<blank>": 0, "<unk>": 1, "<s>": 2, "</s>": 3, "x": 4, "y": 5}
    hyp = [[[4, 5, 3]], [[5, 3], [4, 4]]]
    assert NMTModel.translate_decode_dict(hyp, w2i) == ["x y", "y", "x x"]
    assert NMTModel.translate_decode_dict(hyp, {i: w for w, i in w2i.items()}) == ["x y", "y", "x x"]


def test_noam_schedule_matches_oracle():
    from specialk_amd.lib.transformer.Optim import ScheduledOptim

    class Opt:
        param_groups = [{"lr": 0.0}]
        def step(self): pass
        def zero_grad(self): pass
    s = ScheduledOptim(Opt(), 512, 4000)
    for step in (1, 2, 100, 3999, 4000, 4001, 50000):
        s.n_current_steps = step - 1
        s._update_learning_rate()
        assert s._optimizer.param_groups[0]["lr"] == pytest.approx(O.noam_lr(step, 512, 4000), rel=1e-12)


# ---------------------------------------------------------------- data parallel, 2 ranks over gloo
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from specialk_amd.parallel import GradBucketReducer, all_reduce_scalars
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                              torch.nn.Linear(16, 3))
    unused = torch.nn.Parameter(torch.ones(5))
    params = list(net.parameters()) + [unused]
    red = GradBucketReducer(params, bucket_mb=1e-3)          # ~262 floats per bucket -> several buckets
    assert len(red.buckets) > 1
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    per = 8 // world
    res = []
    direct = []

    class DirectGradLinear(torch.autograd.Function):
        """A layer that, like ops.batched_weight_grads, writes its weight gradient into the reducer's bucket
        slice itself (claim: the first write of a step overwrites) and hands autograd None for it: the reducer
        learns about it through mark_ready()."""

        @staticmethod
        def forward(ctx, inp, W, b):
            ctx.save_for_backward(inp, W)
            return inp @ W.t() + b

        @staticmethod
        def backward(ctx, dy):
            inp, W = ctx.saved_tensors
            red.direct(W)
            view, acc = red.claim(W)
            assert not acc and W.grad is None
            view.copy_(dy.t() @ inp)
            W.grad = view
            direct.append(W)
            return dy @ W, None, dy.sum(0)

    def run(xs):
        h = torch.tanh(net[0](xs))
        h = torch.tanh(DirectGradLinear.apply(h, net[2].weight, net[2].bias))
        return net[4](h)

    for it in range(2):                                       # second pass checks zero_grad / hook re-arming
        red.zero_grad()
        xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
        loss = ((run(xs) - ys) ** 2).sum() * (it + 1)
        loss.backward()
        red.mark_ready(direct)
        del direct[:]
        red.finish()
        res.append([p.grad.clone() for p in params])
    tot = all_reduce_scalars([float(rank + 1), 2.0, 0.5], "cpu")
    if rank == 0:
        torch.save({"grads": res, "tot": tot}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_reducer_sums_over_two_ranks(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                              torch.nn.Linear(16, 3))
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    for it in range(2):
        net.zero_grad()
        (((net(x) - y) ** 2).sum() * (it + 1)).backward()
        for g, p in zip(got["grads"][it], net.parameters()):
            torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-6)
        assert torch.count_nonzero(got["grads"][it][-1]) == 0     # the unused parameter stays zero
    assert got["tot"].tolist() == [3.0, 4.0, 1.0]


def _dp_shared_worker(rank, world, port, out):
    """A matrix used by two layers, each use writing its part of the gradient into the bucket and reporting it with
    a mark_ready of its own (two grouped launches of ops.batched_weight_grads)."""
    import torch.distributed as dist
    from specialk_amd.parallel import GradBucketReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    W = torch.nn.Parameter(torch.randn(16, 16) * 0.3)
    first, last = torch.nn.Linear(6, 16), torch.nn.Linear(16, 3)
    params = list(first.parameters()) + [W] + list(last.parameters())
    red = GradBucketReducer(params, bucket_mb=1e-3)   # W (256 floats) fills a bucket of its own
    assert len(red.buckets) >= 3
    events = []

    class SharedUse(torch.autograd.Function):
        @staticmethod
        def forward(ctx, inp, W):
            ctx.save_for_backward(inp, W)
            return inp @ W.t()

        @staticmethod
        def backward(ctx, dy):
            inp, W = ctx.saved_tensors
            g = dy.t() @ inp
            red.direct(W)
            late = red.late_target(W)
            if late is not None:
                late.copy_(g)
                events.append("late")
            else:
                view, acc = red.claim(W)
                if acc:
                    view.add_(g)
                else:
                    view.copy_(g)
                    W.grad = view
                events.append("acc" if acc else "first")
            red.mark_ready([W])
            return dy @ W, None

    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    per = 8 // world
    res = []
    for it in range(3):
        red.zero_grad()
        xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
        h = torch.tanh(first(xs))
        h = torch.tanh(SharedUse.apply(h, W))
        h = torch.tanh(SharedUse.apply(h, W))
        (((last(h) - ys) ** 2).sum() * (it + 1)).backward()
        red.finish()
        res.append([p.grad.clone() for p in params])
    if rank == 0:
        torch.save({"grads": res, "events": events}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_waits_for_every_use_of_a_shared_matrix(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_shared_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # step 1 learns about the second use after the bucket has left (late tensor, reduced by itself);
    # from step 2 on the bucket waits for both reports
    assert got["events"] == ["first", "late", "first", "acc", "first", "acc"]
    torch.manual_seed(3)
    W = torch.nn.Parameter(torch.randn(16, 16) * 0.3)
    first, last = torch.nn.Linear(6, 16), torch.nn.Linear(16, 3)
    params = list(first.parameters()) + [W] + list(last.parameters())
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 3, generator=torch.Generator().manual_seed(2))
    for it in range(3):
        for p in params:
            p.grad = None
        h = torch.tanh(first(x))
        h = torch.tanh(torch.tanh(h @ W.t()) @ W.t())
        (((last(h) - y) ** 2).sum() * (it + 1)).backward()
        for g, p in zip(got["grads"][it], params):
            torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-6)


def test_single_process_reducer_is_a_no_op_wrapper():
    from specialk_amd.parallel import GradBucketReducer
    lin = torch.nn.Linear(4, 4)
    red = GradBucketReducer(lin.parameters())
    lin(torch.ones(2, 4)).sum().backward()
    red.finish()
    assert lin.weight.grad.data_ptr() == red.buckets[0][0].data_ptr() + 4 * lin.bias.numel()
    lin.zero_grad(set_to_none=True)
    red.zero_grad()
    assert lin.weight.grad is not None and float(lin.weight.grad.abs().sum()) == 0.0


def test_batch_shard_covers_batch_without_overlap(monkeypatch):
    from specialk_amd.lib import TransformerModel as TM
    batch = tuple(torch.arange(10)[:, None] + torch.zeros(10, 3, dtype=torch.long) for _ in range(4))
    for world in (1, 2, 3, 4, 8, 16):
        seen = []
        for r in range(world):
            monkeypatch.setattr(TM, "_world", lambda w=world: w)
            monkeypatch.setattr(TM, "_rank", lambda r=r: r)
            sh = TM.TransformerModel._shard(None, batch)
            seen += sh[0][:, 0].tolist()
        assert seen == list(range(10))


# ---------------------------------------------------------------- token-budget batching
def test_token_budget_sampler_covers_every_sentence_once_within_budget():
    from specialk_amd.core.dataset import TokenBudgetBatchSampler, TranslationDataset, paired_collate_fn
    rng = np.random.default_rng(0)
    n = 3000
    src = [[2] + rng.integers(5, 99, rng.integers(3, 50)).tolist() + [3] for _ in range(n)]
    tgt = [[2] + rng.integers(5, 99, rng.integers(3, 50)).tolist() + [3] for _ in range(n)]
    w2i = {str(i): i for i in range(100)}
    ds = TranslationDataset(w2i, w2i, src, tgt)
    sm = TokenBudgetBatchSampler.for_dataset(ds, 2000, max_sentences=64, pool_batches=8, seed=5)
    cost = np.array([max(len(a), len(b)) for a, b in zip(src, tgt)])
    e0 = list(sm)            # epoch 0; iterating advances the epoch
    e1 = list(sm)
    for batches in (e0, e1):
        flat = sorted(i for b in batches for i in b)
        assert flat == list(range(n))
        assert all(cost[b].sum() <= 2000 and len(b) <= 64 for b in batches)
    assert e0 != e1
    sm2 = TokenBudgetBatchSampler.for_dataset(ds, 2000, max_sentences=64, pool_batches=8, seed=5)
    assert list(sm2) == e0 and len(sm2) == len(e1)   # a function of (seed, epoch) only
    # bucketing: padded size close to the real token count (unsorted batches of 64 waste ~2x)
    padded = sum(len(b) * cost[b].max() for b in e0)
    assert padded <= 1.1 * cost.sum()
    # a sentence longer than the budget still gets a batch of its own
    lone = TokenBudgetBatchSampler([5, 3000, 7], 100, shuffle=False)
    assert sorted(map(sorted, lone)) == [[0, 2], [1]]
    # the DataLoader contract: batch_sampler + the reference's collate function
    loader = torch.utils.data.DataLoader(ds, batch_sampler=sm2, collate_fn=paired_collate_fn)
    s_seq, s_pos, t_seq, t_pos = next(iter(loader))
    assert s_seq.shape == s_pos.shape and t_seq.shape[0] == s_seq.shape[0] <= 64


def test_product_sinusoid_table_matches_reference_fixture():
    """get_sinusoid_encoding_table of the drop-in package (lib/transformer/Models.py:26-46) against the tables the
    reference's function produced (tests/golden/sinusoid.npz), bit for bit."""
    from conftest import load_golden
    from specialk_amd.lib.transformer.Models import get_sinusoid_encoding_table
    g = load_golden("sinusoid")
    np.testing.assert_array_equal(get_sinusoid_encoding_table(20, 32, padding_idx=0).numpy(), g["t_20_32"])
    np.testing.assert_array_equal(get_sinusoid_encoding_table(7, 6, padding_idx=0).numpy(), g["t_7_6"])
    np.testing.assert_array_equal(get_sinusoid_encoding_table(5, 8).numpy(), g["t_5_8_nopad"])
