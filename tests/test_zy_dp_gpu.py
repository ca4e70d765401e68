"""Data-parallel training step on the GPU path with two ranks (both on cuda:0, gloo transport so
that one card suffices; the reducer hands gloo host copies of its device buckets): every rank runs the padding-free forward / backward on its shard of the
golden batch with the bucketed gradient reducer hooked into backward; the summed gradients must
equal the reference's full-batch gradients (the loss is a token SUM).

The file name sorts late on purpose: a multi-process test is the likeliest to be disturbed by the
machine, and `pytest -x` should not lose the single-process tests behind it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, split_prefixed

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out, packed, bf16=False, backend="gloo", tied=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import faulthandler
    faulthandler.dump_traceback_later(40, exit=True)  # a stuck rank reports where, then fails the test
    import torch.distributed as dist
    from specialk_amd import ops
    from specialk_amd.parallel import GradBucketReducer, all_reduce_scalars
    from test_model_gpu import build_tiny, load_sd, t
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    torch.cuda.set_device(0)
    extra = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90), **extra)
    g = load_golden("transformer_tied" if tied else "transformer_untied")
    m = load_sd(build_tiny(g, tied), split_prefixed(g, "w:")).eval()
    if bf16:
        m = m.set_compute_dtype(torch.bfloat16)
    params = [p for p in m.parameters() if p.requires_grad]
    red = GradBucketReducer(params, bucket_mb=0.02, always_reduce=True)
    assert red.active and red._staged == (backend != "nccl")
    assert len(red.buckets) > 2
    batch = [t(g[k]) for k in ("src_seq", "src_pos", "tgt_seq", "tgt_pos")]
    per = batch[0].shape[0] // world
    shard = [x[rank * per:(rank + 1) * per].contiguous() for x in batch]
    for _ in range(2):  # second pass: buckets re-zeroed, hooks re-armed
        red.zero_grad()
        if packed:
            logits, gold = m.forward_packed(*shard)
        else:
            logits = m(*shard)
            logits, gold = logits.view(-1, logits.size(2)), shard[2][:, 1:]
        loss, stats = ops.xent(logits, gold, True, unit_grad=True)
        if bf16:  # the trainer's form: grouped weight gradients accumulate straight into the bucket views
            with ops.batched_weight_grads(on_ready=red.mark_ready, flush_tiles=8):
                loss.backward()
        else:
            loss.backward()
        red.finish()
    tot = all_reduce_scalars(stats, "cuda")
    if world == 1:  # all_reduce_scalars has nothing to do for one rank: reduce a device tensor over the backend anyway
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        res = {"grads": {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}, "stats": tot.cpu(),
               "gathered_matrix_bytes": red.gathered_matrix_bytes, "late_parts": red.late_parts}
        if bf16:  # reference for the bf16 run: the same arithmetic on the whole batch in one process, through autograd
            m2 = load_sd(build_tiny(g, tied), split_prefixed(g, "w:")).eval().set_compute_dtype(torch.bfloat16)
            logits, gold = m2.forward_packed(*batch) if packed else (None, None)
            if not packed:
                logits = m2(*batch)
                logits, gold = logits.view(-1, logits.size(2)), batch[2][:, 1:]
            loss, _ = ops.xent(logits, gold, True, unit_grad=True)
            loss.backward()
            res["ref"] = {k: p.grad.cpu() for k, p in m2.named_parameters() if p.grad is not None}
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tied", [False, True])
@pytest.mark.parametrize("packed", [True, False])
def test_two_rank_step_matches_full_batch_gradients(tmp_path, packed, tied):
    """tied: one matrix is source embedding, target embedding and generator (the Transformer-big setting); its
    gradient has three producers per rank and lives once in the reducer's buckets."""
    out = str(tmp_path / "dp.pt")
    # ranks are forked from the clean fork server started in conftest.py (see there)
    mp.start_processes(_worker, args=(2, _free_port(), out, packed, False, "gloo", tied), nprocs=2, join=True,
                       start_method="forkserver")
    got = torch.load(out)
    g = load_golden("transformer_tied" if tied else "transformer_untied")
    assert abs(float(got["stats"][0]) - float(g["loss_ls"])) <= 1e-5 * float(g["loss_ls"])
    assert int(got["stats"][1]) == int(g["ncorrect_ls"])
    for k, ref in split_prefixed(g, "g_ls:").items():
        a, b = got["grads"][k].double().numpy(), ref.astype(np.float64)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max() + 2e-6, k


@pytest.mark.parametrize("tied", [False, True])
@pytest.mark.parametrize("packed", [True, False])
def test_two_rank_step_bf16_grouped_weight_grads(tmp_path, packed, tied):
    """bf16 with the grouped weight-gradient launches accumulating into the reducer's bucket views:
    the two ranks' summed gradients against a one-process whole-batch run of the same arithmetic."""
    out = str(tmp_path / "dp.pt")
    mp.start_processes(_worker, args=(2, _free_port(), out, packed, True, "gloo", tied), nprocs=2, join=True,
                       start_method="forkserver")
    got = torch.load(out)
    assert got["grads"].keys() == got["ref"].keys()
    for k, ref in got["ref"].items():
        a, b = got["grads"][k].double(), ref.double()
        assert float((a - b).norm()) <= 3e-2 * float(b.norm()) + 1e-6, k


@pytest.mark.parametrize("tied", [False, True])
@pytest.mark.parametrize("bf16", [False, True])
def test_one_rank_rccl_rehearsal(tmp_path, bf16, tied):
    """The RCCL form of the exchange on the one card there is: a group of one rank over the nccl backend with
    the reducer forced to reduce — hooks and grouped launches start asynchronous all-reduces of the device
    buckets on RCCL's stream during backward, finish() waits for them, scalars are reduced on the device.
    A sum over one rank changes nothing, so the gradients must equal the whole-batch gradients.  In the trainer's
    form (bf16, grouped launches reporting to the reducer) every matrix gradient — the tied embedding / generator
    matrix with its three producers included — is written straight into its bucket slice: none takes the copy."""
    out = str(tmp_path / "dp.pt")
    mp.start_processes(_worker, args=(1, _free_port(), out, True, bf16, "nccl", tied), nprocs=1, join=True,
                       start_method="forkserver")
    got = torch.load(out)
    if bf16:
        assert got["gathered_matrix_bytes"] == 0
        assert got["late_parts"] == 0   # no bucket left before the last of its gradients was written
        for k, ref in got["ref"].items():
            a, b = got["grads"][k].double(), ref.double()
            assert float((a - b).norm()) <= 3e-2 * float(b.norm()) + 1e-6, k
    else:
        g = load_golden("transformer_tied" if tied else "transformer_untied")
        assert abs(float(got["stats"][0]) - float(g["loss_ls"])) <= 1e-5 * float(g["loss_ls"])
        for k, ref in split_prefixed(g, "g_ls:").items():
            a, b = got["grads"][k].double().numpy(), ref.astype(np.float64)
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max() + 2e-6, k


def _translate_worker(rank, world, port, argv):
    sys.path.insert(0, ROOT)
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from specialk_amd import translate as translate_cli
    translate_cli.main(argv)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_translate_shards_sentences_over_ranks(tmp_path, monkeypatch):
    """Three ranks (all on cuda:0) translate contiguous shards of seven sentences; the joined output equals the
    one-process output line for line (no communication: beam search of a sentence does not depend on its batch)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cli_gpu import make_corpus
    from specialk_amd import translate as translate_cli
    from specialk_amd.train import load_args, train_model
    monkeypatch.chdir(tmp_path)
    data_path = str(tmp_path / "toy.pt")
    make_corpus(data_path, n_train=48, n_valid=8)
    train_model(load_args().parse_args([
        "-data", data_path, "-model", "transformer", "-epochs", "1", "-batch_size", "8", "-layers", "1",
        "-d_model", "32", "-d_word_vec", "32", "-d_inner_hid", "64", "-d_k", "8", "-d_v", "8", "-n_head", "4",
        "-n_warmup_steps", "10", "-save_model", "-directory_name", "run", "-seed", "5"]))
    run = tmp_path / "models" / "run"
    rng = np.random.default_rng(1)
    sents = [" ".join(f"w{int(i)}" for i in rng.integers(4, 39, size=int(rng.integers(1, 9)))) for _ in range(7)]
    (tmp_path / "in.txt").write_text("\n".join(sents) + "\n")
    common = ["-model", "transformer", "-src", str(tmp_path / "in.txt"), "-vocab", data_path,
              "-checkpoint_encoder", str(run / "encoder_epoch_1.chkpt"), "-checkpoint_decoder", str(run / "decoder_epoch_1.chkpt"),
              "-batch_size", "2", "-beam_size", "3", "-n_best", "1"]
    translate_cli.main(common + ["-output", str(tmp_path / "one.txt")])
    mp.start_processes(_translate_worker, args=(3, _free_port(), common + ["-output", str(tmp_path / "many.txt")]),
                       nprocs=3, join=True, start_method="forkserver")
    one, many = (tmp_path / "one.txt").read_text(), (tmp_path / "many.txt").read_text()
    assert len(one.split("\n")) == 8 and one == many
    assert not list(tmp_path.glob("many.txt.rank*"))


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` started as a plain process must come back with a line measured by two ranks (it starts them
    itself).  Rehearsed on one card: both ranks on device 0, gloo transport."""
    import json
    import subprocess
    env = dict(os.environ, SK_BENCH_DEVICE="0", SK_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-decode", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and len(res["ms_per_step_by_rank"]) == 2
    assert res["steps"] == 3 and res["value"] > 0
