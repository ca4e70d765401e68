"""End-to-end test of the command-line drop-ins on the GPU: train two epochs on a toy corpus through
specialk_amd.train (checkpoints + logs in the reference's formats), reload the split checkpoints
through specialk_amd.translate, and check the written translations against the CPU oracle's beam
search on the same weights (token-exact in fp32)."""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle import transformer_np as O  # noqa: E402

pytestmark = pytest.mark.gpu

WORDS = ["<blank>", "<unk>", "<s>", "</s>"] + [f"w{i}" for i in range(4, 40)]
W2I = {w: i for i, w in enumerate(WORDS)}


def make_corpus(path, n_train=96, n_valid=24, seed=0):
    """A dataset file shaped like the reference's preprocess.py output: the target is the reversed
    source shifted by one id, so two epochs already move the loss."""
    rng = np.random.default_rng(seed)

    def pairs(n):
        src, tgt = [], []
        for _ in range(n):
            body = rng.integers(4, 39, size=int(rng.integers(2, 9))).tolist()
            src.append([2] + body + [3])
            tgt.append([2] + [b + 1 for b in reversed(body)] + [3])
        return {"src": src, "tgt": tgt}
    data = {"settings": Namespace(max_token_seq_len=12, max_word_seq_len=10, format="word", case_sensitive=True),
            "dict": {"src": W2I, "tgt": W2I}, "train": pairs(n_train), "valid": pairs(n_valid)}
    torch.save(data, path)
    return data


def test_train_save_load_translate(tmp_path, monkeypatch):
    from specialk_amd.train import load_args, train_model
    from specialk_amd import translate as translate_cli
    monkeypatch.chdir(tmp_path)
    data_path = str(tmp_path / "toy.pt")
    make_corpus(data_path)
    opt = load_args().parse_args([
        "-data", data_path, "-model", "transformer", "-epochs", "2", "-batch_size", "8", "-layers", "2",
        "-d_model", "32", "-d_word_vec", "32", "-d_inner_hid", "64", "-d_k", "8", "-d_v", "8", "-n_head", "4",
        "-n_warmup_steps", "10", "-label_smoothing", "-save_model", "-directory_name", "run", "-seed", "5"])
    model = train_model(opt)
    assert len(model.train_losses) == 2 and len(model.valid_accs) == 2
    assert all(np.isfinite(model.train_losses)) and model.train_losses[1] < model.train_losses[0]

    run = tmp_path / "models" / "run"
    for name in ("encoder_epoch_1.chkpt", "decoder_epoch_2.chkpt", "train.log", "valid.log"):
        assert (run / name).is_file(), name
    rows = (run / "train.log").read_text().strip().split("\n")
    assert rows[0] == "epoch,loss,ppl,accuracy" and len(rows) == 3 and rows[2].startswith("2,")
    enc = torch.load(run / "encoder_epoch_2.chkpt", weights_only=False)
    dec = torch.load(run / "decoder_epoch_2.chkpt", weights_only=False)
    assert set(enc) == {"type", "model", "epoch", "settings"} and enc["type"] == "transformer" and enc["epoch"] == 2
    assert set(dec) == {"type", "model", "generator", "epoch", "settings"}
    assert "src_word_emb.weight" in enc["model"] and "layer_stack.1.pos_ffn.w_2.bias" in enc["model"]
    assert "layer_stack.0.enc_attn.w_qs.weight" in dec["model"] and list(dec["generator"]) == ["weight"]
    assert enc["settings"].d_model == 32 and enc["settings"].src_vocab_size == len(WORDS)

    # a DataParallel-style checkpoint ("module." prefix) must load as well
    enc_dp = dict(enc, model={"module." + k: v for k, v in enc["model"].items()})
    torch.save(enc_dp, run / "encoder_dp.chkpt")

    src_txt = tmp_path / "test.txt"
    sents = [["w5", "w9", "w30"], ["w7"], ["w11", "w12", "w13", "w14", "w15", "w16"], ["nope", "w4"]]
    src_txt.write_text("\n".join(" ".join(s) for s in sents) + "\n")
    out_txt = tmp_path / "pred.txt"
    hyps = translate_cli.main([
        "-model", "transformer", "-src", str(src_txt), "-vocab", data_path, "-output", str(out_txt),
        "-checkpoint_encoder", str(run / "encoder_dp.chkpt"), "-checkpoint_decoder", str(run / "decoder_epoch_2.chkpt"),
        "-batch_size", "3", "-beam_size", "3", "-n_best", "1"])
    assert len(hyps) == len(sents)

    # oracle on the same weights
    w = {"encoder." + k: v.cpu().numpy() for k, v in enc["model"].items()}
    w.update({"decoder." + k: v.cpu().numpy() for k, v in dec["model"].items()})
    w["generator.weight"] = dec["generator"]["weight"].cpu().numpy()
    cfg = O.Config(2, 4, 8, 8, 32, 64)
    lines = out_txt.read_text().split("\n")[:-1]
    assert len(lines) == len(sents)
    k = 0
    for lo in range(0, len(sents), 3):
        insts = [[2] + [W2I.get(x, 1) for x in s] + [3] for s in sents[lo:lo + 3]]
        seq, pos = O.collate(insts)
        ohyp, _ = O.translate_batch(w, cfg, seq, pos, 3, 1, 12)
        for i in range(len(insts)):
            assert hyps[k][0] == list(ohyp[i][0]), (k, hyps[k][0], ohyp[i][0])
            assert lines[k] == " ".join(WORDS[t] for t in ohyp[i][0] if t != 3)
            k += 1


def test_frozen_encoder_keeps_its_weights(tmp_path, monkeypatch):
    from specialk_amd.train import load_args, train_model
    monkeypatch.chdir(tmp_path)
    data_path = str(tmp_path / "toy.pt")
    make_corpus(data_path, n_train=32, n_valid=8)
    args = ["-data", data_path, "-model", "transformer", "-epochs", "1", "-batch_size", "16", "-layers", "1",
            "-d_model", "32", "-d_word_vec", "32", "-d_inner_hid", "64", "-d_k", "8", "-d_v", "8", "-n_head", "4",
            "-n_warmup_steps", "5", "-seed", "7", "-dropout", "0.0"]
    ref = train_model(load_args().parse_args(args + ["-epochs", "1"]))
    torch.manual_seed(7)
    from specialk_amd.lib.TransformerModel import TransformerModel
    opt = load_args().parse_args(args + ["-freeze_encoder"])
    frozen = train_model(opt)
    # same seed -> same initial weights; encoder untouched, decoder trained
    init = TransformerModel(load_args().parse_args(args))
    torch.manual_seed(7)
    init.load_dataset()
    init.initiate()
    for (k, a), b in zip(init.model.encoder.state_dict().items(), frozen.model.encoder.state_dict().values()):
        assert torch.equal(a, b), k
    moved = [not torch.equal(a, b) for a, b in zip(init.model.decoder.state_dict().values(),
                                                   frozen.model.decoder.state_dict().values())]
    assert any(moved)
    assert np.isfinite(ref.train_losses[0]) and np.isfinite(frozen.train_losses[0])


def test_token_budget_batches_and_bf16_training_reduce_the_loss(tmp_path, monkeypatch):
    """-batch_tokens (length-bucketed batches under a token budget) with -dtype bf16 — the trainer's fast
    path: packed batches, grouped weight gradients — over a few epochs of the toy corpus: every
    sentence is seen once per epoch and the training loss goes down."""
    from specialk_amd.train import load_args, train_model
    monkeypatch.chdir(tmp_path)
    data_path = str(tmp_path / "toy.pt")
    make_corpus(data_path, n_train=96, n_valid=16)
    args = ["-data", data_path, "-model", "transformer", "-epochs", "6", "-batch_size", "16", "-batch_tokens", "120",
            "-layers", "1", "-d_model", "32", "-d_word_vec", "32", "-d_inner_hid", "64", "-d_k", "8", "-d_v", "8",
            "-n_head", "4", "-n_warmup_steps", "20", "-seed", "3", "-dropout", "0.0", "-dtype", "bf16"]
    tm = train_model(load_args().parse_args(args))
    sampler = tm.training_data.batch_sampler
    seen = sorted(i for batch in sampler for i in batch)
    assert seen == list(range(96))
    assert len(tm.train_losses) == 6 and all(np.isfinite(tm.train_losses))
    assert tm.train_losses[-1] < tm.train_losses[0]


def test_train_decoder_cli_fine_tunes_decoder_only(tmp_path, monkeypatch):
    """train -> train_decoder on the saved checkpoints with a classifier checkpoint: the decoder
    checkpoints appear in the reference's format and every decoder weight has moved."""
    from specialk_amd import train_decoder as TD
    from specialk_amd.classifier.CNNModels import ConvNet
    from specialk_amd.train import load_args, train_model
    monkeypatch.chdir(tmp_path)
    data_path = str(tmp_path / "toy.pt")
    make_corpus(data_path, n_train=48, n_valid=16)
    train_model(load_args().parse_args([
        "-data", data_path, "-model", "transformer", "-epochs", "1", "-batch_size", "8", "-layers", "2",
        "-d_model", "32", "-d_word_vec", "32", "-d_inner_hid", "64", "-d_k", "8", "-d_v", "8", "-n_head", "4",
        "-n_warmup_steps", "10", "-save_model", "-directory_name", "run", "-seed", "5"]))
    run = tmp_path / "models" / "run"
    torch.manual_seed(3)
    class_opt = Namespace(num_filters=6, filter_size=3, sequence_length=12, word_vec_size=8, dropout=0.2, num_classes=2)
    cnn = ConvNet(class_opt, len(WORDS))
    torch.save({"model": cnn.state_dict(), "opt": class_opt}, tmp_path / "cnn.pt")
    dec0 = torch.load(run / "decoder_epoch_1.chkpt", weights_only=False)["model"]
    TD.main(["-data", data_path, "-checkpoint_encoder", str(run / "encoder_epoch_1.chkpt"),
             "-checkpoint_decoder", str(run / "decoder_epoch_1.chkpt"), "-epochs", "2", "-batch_size", "8",
             "-classifier_model", str(tmp_path / "cnn.pt"), "-label0", "a", "-label1", "b", "-label_target", "b",
             "-label_smoothing", "-seed", "9"])
    for ep in (1, 2):
        assert (run / f"decoder_b_epoch_{ep}.chkpt").is_file()
    out = torch.load(run / "decoder_b_epoch_2.chkpt", weights_only=False)
    assert set(out) == {"type", "model", "generator", "epoch", "settings", "vocab"} and out["epoch"] == 2
    moved = [not torch.equal(v.cpu(), dec0[k].cpu()) for k, v in out["model"].items() if "position_enc" not in k]
    assert all(moved)
    assert all(torch.isfinite(v).all() for v in out["model"].values())
    # a classifier file that pickles anything but tensors and the options is refused by the loader
    torch.save({"model": cnn.state_dict(), "opt": class_opt, "optim": ConvNet}, tmp_path / "bad.pt")
    with pytest.raises(RuntimeError, match="classifier checkpoint"):
        TD.load_classifier(Namespace(classifier_model=str(tmp_path / "bad.pt"), label0="a", label1="b", label_target="b",
                                     batch_size=8), Namespace(opt=Namespace(max_token_seq_len=12, tgt_vocab_size=len(WORDS))))


def test_bpe_dataset_trains_and_translates_text(tmp_path, monkeypatch):
    """A dataset in the reference's BPE format (dict = Encoder.vocabs_to_dict()): training picks up the
    vocabularies, translate.py encodes raw text with the source vocabulary and writes decoded words."""
    from specialk_amd import translate as translate_cli
    from specialk_amd.core.bpe import Encoder
    from specialk_amd.lib.nmtModel import NMTModel
    from specialk_amd.train import load_args, train_model
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(4)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]
    sents = [" ".join(rng.choice(words, size=int(rng.integers(2, 6)))) for _ in range(80)]
    enc = Encoder(vocab_size=48, pct_bpe=0.75, ngram_min=1, ngram_max=2)
    enc.fit(sents)
    ids = [[2] + s + [3] for s in enc.transform(sents)]
    assert max(len(s) for s in ids) <= 40
    data = {"settings": Namespace(max_token_seq_len=40, max_word_seq_len=38, format="bpe", case_sensitive=True),
            "dict": {"src": enc.vocabs_to_dict(), "tgt": enc.vocabs_to_dict()},
            "train": {"src": ids[:64], "tgt": ids[:64]}, "valid": {"src": ids[64:], "tgt": ids[64:]}}
    data_path = str(tmp_path / "bpe.pt")
    torch.save(data, data_path)
    model = train_model(load_args().parse_args([
        "-data", data_path, "-model", "transformer", "-epochs", "1", "-batch_size", "16", "-layers", "1",
        "-d_model", "32", "-d_word_vec", "32", "-d_inner_hid", "64", "-d_k", "8", "-d_v", "8", "-n_head", "4",
        "-n_warmup_steps", "10", "-save_model", "-directory_name", "bpe", "-seed", "3", "-device_collate"]))
    assert model.tgt_bpe is not None and model.src_bpe is not None
    run = tmp_path / "models" / "bpe"
    src_txt = tmp_path / "in.txt"
    src_txt.write_text("alpha beta gamma\nkappa omega mu\n" + " ".join(["epsilon"] * 30) + "\n")
    out_txt = tmp_path / "out.txt"
    hyps = translate_cli.main([
        "-model", "transformer", "-src", str(src_txt), "-vocab", data_path, "-output", str(out_txt),
        "-checkpoint_encoder", str(run / "encoder_epoch_1.chkpt"), "-checkpoint_decoder", str(run / "decoder_epoch_1.chkpt"),
        "-batch_size", "2", "-beam_size", "2", "-n_best", "1"])
    lines = out_txt.read_text().split("\n")[:-1]
    assert len(hyps) == 3 and len(lines) == 3
    assert lines == NMTModel.translate_decode_bpe(hyps, enc)
