"""Style-transfer fine-tuning of the Transformer decoder, with the reference's interface
(base/train_decoder.py): the encoder is frozen and its output detached, the decoder and generator train
on  net_loss = l_reconstruction * l_classifier  (:471), where l_classifier is the binary cross entropy of
a frozen CNN style classifier run on softmax(class_input(decoder output)).

What runs where: encoder, decoder, generator, both losses, backward and Adam are HIP kernels
(specialk_amd.ops, specialk_amd.classifier.CNNModels, specialk_amd.optim); this file is host-side
orchestration.  Facts of the reference this file relies on:
  * classifier_loss feeds the classifier `pred.data` (:399) and class_input is created after
    setup_optimiser (:536, :549), so l_classifier reaches the model's parameters only as the scalar
    that multiplies the reconstruction gradient; that scalar stays on the device and scales the
    gradient inside the loss kernel;
  * the encoder output is detached (:368): encoder parameters receive no gradient (an embedding shared
    with the decoder still receives the decoder's and generator's share) and Adam leaves them alone.
Differences: batches run without their padding rows when they are right-padded (same losses and
gradients); per-batch statistics stay on the device until the epoch ends instead of four `.item()` reads
per step (:486-489); `load_classifier` also loads the checkpoint's weights (the reference builds the
classifier from the saved options and leaves it randomly initialised, :345-350) and reads the checkpoint
with torch's weights-only loader, which accepts tensors and the options namespace and nothing else.
"""
import argparse
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .classifier import CNNModels
from .lib.transformer.Models import plan_packing

description = "train_decoder.py — trains the transformer decoder for style-transfer specific purposes."


def load_args():
    """The reference's flags (train_decoder.py:24-329) plus -dtype / -seed of this package."""
    p = argparse.ArgumentParser(description="train_decoder.py")
    p.add_argument("-f")
    p.add_argument("-data", required=True)
    p.add_argument("-checkpoint_encoder", default="", type=str)
    p.add_argument("-checkpoint_decoder", default="", type=str)
    p.add_argument("-new_directory", action="store_true")
    p.add_argument("-log", action="store_true")
    p.add_argument("-directory_name", type=str, default="")
    p.add_argument("-save_model", action="store_true")
    p.add_argument("-save_mode", default="all", choices=["all", "best"])
    p.add_argument("-verbose", action="store_true")
    p.add_argument("-cuda", action="store_true")
    p.add_argument("-multi_gpu", action="store_true")
    p.add_argument("-cuda_device", type=int)
    p.add_argument("-batch_size", type=int, default=64)
    p.add_argument("-layers", type=int, default=6)
    p.add_argument("-d_word_vec", type=int, default=300)
    p.add_argument("-epochs", type=int, required=True, default=10)
    p.add_argument("-dropout", type=float, default=0.1)
    p.add_argument("-telegram", type=str, default="")
    p.add_argument("-d_model", type=int, default=512)
    p.add_argument("-d_inner_hid", type=int, default=2048)
    p.add_argument("-d_k", type=int, default=64)
    p.add_argument("-d_v", type=int, default=64)
    p.add_argument("-n_head", type=int, default=8)
    p.add_argument("-n_warmup_steps", type=int, default=4000)
    p.add_argument("-embs_share_weight", action="store_true")
    p.add_argument("-proj_share_weight", action="store_true")
    p.add_argument("-label_smoothing", action="store_true")
    p.add_argument("-learning_rate", type=float, default=0.001)
    p.add_argument("-learning_rate_decay", type=float, default=0.5)
    p.add_argument("-start_decay_at", type=int, default=8)
    p.add_argument("-optim", default="adam", choices=["sgd", "adagrad", "adadelta", "adam"])
    p.add_argument("-classifier_model", required=True, type=str)
    p.add_argument("-label0", required=True, type=str)
    p.add_argument("-label1", required=True, type=str)
    p.add_argument("-label_target", required=True, type=str)
    p.add_argument("-dtype", default="fp32", choices=["fp32", "bf16"])
    p.add_argument("-seed", type=int, default=None)
    return p


def load_classifier(opt, model):
    """Build the CNN classifier from a classifier checkpoint and load its weights
    (train_decoder.py:332-353).  The checkpoint is {"model": state_dict, "opt": argparse.Namespace}
    — the two entries of what classifier/cnn_train.py:364-375 writes that are used here.  It is read
    with the weights-only loader: a file that also pickles the classifier trainer's dictionary and
    optimiser objects has to be re-saved with those two entries first."""
    cnn_opt = argparse.Namespace()
    cnn_opt.model = opt.classifier_model
    cnn_opt.max_sent_length = model.opt.max_token_seq_len
    cnn_opt.label0, cnn_opt.label1, cnn_opt.tgt = opt.label0, opt.label1, opt.label_target
    cnn_opt.cuda = 0
    cnn_opt.batch_size = opt.batch_size
    cnn_opt.tgt_label = 1 if opt.label_target == opt.label1 else 0
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            data = torch.load(opt.classifier_model, map_location="cpu", weights_only=True)
    except Exception as e:  # the weights-only loader names what it refused
        raise RuntimeError(
            f"{opt.classifier_model}: a classifier checkpoint must hold only 'model' (tensors) and 'opt' "
            f"(argparse.Namespace); re-save those two entries of the original file.  Loader said: {e}") from e
    class_opt = data["opt"]
    class_model = CNNModels.ConvNet(class_opt, model.opt.tgt_vocab_size)
    state = data.get("model")
    if state:
        class_model.load_state_dict({k[7:] if k.startswith("module.") else k: v for k, v in state.items()})
    class_model.max_sent_length = cnn_opt.max_sent_length
    class_model.to(model.device)
    class_model.compute_dtype = model.model.compute_dtype
    class_model.eval()
    class_model.refs = cnn_opt
    return class_model


def make_class_input(model, class_model):
    """The layer that feeds decoder outputs to the classifier (train_decoder.py:546-551)."""
    model_dim = model.model.decoder.layer_stack[0].slf_attn.w_qs.out_features
    return nn.Sequential(nn.Linear(model_dim, class_model.word_lut.weight.shape[0])).to(model.device)


def tf_forward(self, src_seq, src_pos, tgt_seq, tgt_pos):
    """Decoder outputs (B, Lt-1, d) on the detached encoder output (train_decoder.py:360-370);
    `self` is the Transformer."""
    tgt_seq, tgt_pos = tgt_seq[:, :-1], tgt_pos[:, :-1]
    with torch.no_grad():
        enc_output, _ = self.encoder(src_seq, src_pos, True)
    dec_output, _, _ = self.decoder(tgt_seq, tgt_pos, src_seq, enc_output.detach(), True)
    return dec_output


def n_tokens_correct(pred, gold, pad_token):
    """train_decoder.py:373-382."""
    pred = pred.max(1)[1]
    gold = gold.contiguous().view(-1)
    return pred.eq(gold).masked_select(gold.ne(pad_token)).sum().item()


def classifier_loss(self, pred, class_model, class_input):
    """Mean binary cross entropy of the classifier's verdict on softmax(class_input(pred)) against the
    target label (train_decoder.py:385-424); pred (B, L, d) decoder outputs.  A one-element fp32 tensor."""
    B, L, d = pred.shape
    loss, _ = class_model.loss_from_decoder(pred.reshape(B * L, d), B, L, class_input, class_model.refs.tgt_label)
    return loss


def performance(self, pred_before, gold, class_input, class_model, smoothing=False):
    """(l_reconstruction, l_classifier, stats) of decoder outputs pred_before (B, L, d)
    (train_decoder.py:427-444); stats = [loss sum, n_correct, n_words] on the device.  The
    reconstruction loss carries the gradient of the product l_reconstruction * l_classifier:
    differentiate it directly (`l_reconstruction.backward()`)."""
    l_classifier = classifier_loss(self, pred_before, class_model, class_input)
    B, L, d = pred_before.shape
    pred = self.model.project(pred_before.reshape(B * L, d))
    l_reconstruction, stats = ops.xent(pred, gold.reshape(-1), smoothing, unit_grad=True, grad_scale_dev=l_classifier)
    return l_reconstruction, l_classifier, stats


def _style_batch(self, class_input, class_model, batch, smooth, want_logits=False):
    """Forward pass and both losses of one batch -> (l_recon, l_class (1,), stats, logits or None)."""
    m = self.model
    src_seq, src_pos, tgt_seq, tgt_pos = (x.to(self.device, non_blocking=True) for x in batch)
    gold = tgt_seq[:, 1:]
    if getattr(self, "packed", True) and not want_logits:
        try:
            tgt_in, tgt_pos_in = tgt_seq[:, :-1].contiguous(), tgt_pos[:, :-1].contiguous()
            check = None
            if m.encoder.check_ids:  # ids / positions outside their tables must raise, not reach a kernel
                enc, dec = m.encoder, m.decoder
                check = [(src_pos, enc.src_word_emb.num_embeddings, enc.position_enc.num_embeddings),
                         (tgt_pos_in, dec.tgt_word_emb.num_embeddings, dec.position_enc.num_embeddings)]
            src_plan, tgt_plan = plan_packing([src_seq, tgt_in], [0, 0], check)
            with torch.no_grad():
                enc_packed = m.encoder.forward_packed(src_seq, src_pos, src_plan)
            dec_packed = m.decoder.forward_packed(tgt_in, tgt_pos_in, tgt_plan, enc_packed, src_plan)
            l_class, _ = class_model.loss_from_decoder(dec_packed, tgt_plan.B, tgt_plan.L, class_input,
                                                       class_model.refs.tgt_label, off=tgt_plan.off)
            l_recon, stats = ops.xent(m.project(dec_packed), gold.reshape(-1)[tgt_plan.idx], smooth, unit_grad=True,
                                      grad_scale_dev=l_class)
            return l_recon, l_class, stats, None
        except ValueError:  # not a right-padded batch
            self.packed = False
    pred = tf_forward(m, src_seq, src_pos, tgt_seq, tgt_pos)
    l_recon, l_class, stats = performance(self, pred, gold, class_input, class_model, smooth)
    logits = None
    if want_logits:
        B, L, d = pred.shape
        logits = m.project(pred.reshape(B * L, d)).view(B, L, -1)
    return l_recon, l_class, stats, logits


def compute_epoch(self, class_input, class_model, dataset, validation=False):
    """One pass over `dataset` (train_decoder.py:447-497) -> per-batch lists
    (net losses, reconstruction losses, classifier losses, net loss per target word)."""
    m = self.model
    m.decoder.train(not validation)
    m.generator.train(not validation)
    smooth = self.opt.label_smoothing
    rows = []
    for batch in dataset:
        m.zero_grad()
        log_outputs = validation and getattr(self, "tgt_bpe", None) is not None
        l_recon, l_class, stats, logits = _style_batch(self, class_input, class_model, batch, smooth, log_outputs)
        if not validation:
            with ops.batched_weight_grads():
                l_recon.backward()  # the loss kernel already scaled the gradient by l_class
            self.optimiser.step_and_update_lr()
        elif log_outputs:
            self.save_eval_outputs(logits, output_dir="eval_outputs_st_" + str(getattr(self, "label_target", "")))
        rows.append(torch.stack([l_recon.detach().double(), l_class[0].double(), stats[2]]))
    if not rows:
        return [], [], [], []
    t = torch.stack(rows).cpu().numpy()
    net = t[:, 0] * t[:, 1]
    return net.tolist(), t[:, 0].tolist(), t[:, 1].tolist(), (net / np.maximum(t[:, 2], 1.0)).tolist()


def means(l):
    return [round(float(np.mean(x)), 3) if len(x) else float("nan") for x in l]


def save(decoder, generator, epoch, settings, vocab, filepath):
    """decoder_<label>_epoch_<n>.chkpt = {type, model, generator, epoch, settings, vocab}
    (train_decoder.py:504-517)."""
    if getattr(settings, "telegram", None):
        del settings.telegram
    torch.save({"type": "transformer", "model": decoder.state_dict(), "generator": generator.state_dict(),
                "epoch": epoch, "settings": settings, "vocab": vocab}, filepath)


def main(argv=None):
    """train_decoder.py:520-580."""
    from .lib.TransformerModel import TransformerModel as transformer
    opt = load_args().parse_args(argv)
    opt.model = "transformer"
    assert opt.epochs > 0
    if opt.seed is not None:
        torch.manual_seed(opt.seed)
        ops.manual_seed(opt.seed)
    model = transformer(opt)
    model.label_target = opt.label_target
    print("LOADING DATASET:", opt.data)
    model.load_dataset()
    vocab = model.tgt_bpe.vocabs_to_dict(False) if getattr(model, "tgt_bpe", None) is not None else None
    model.load(opt.checkpoint_encoder, opt.checkpoint_decoder)
    model.setup_optimiser()
    model.model.encoder.eval()
    classifier = load_classifier(opt, model)
    class_input = make_class_input(model, classifier)
    print("model.opt.directory:", model.opt.directory)
    for ep in range(1, opt.epochs + 1):
        model.opt.current_epoch = ep
        losses, recon, classes, accs = means(compute_epoch(model, class_input, classifier, model.training_data))
        print("Training Loss:", losses, "(", recon, classes, ")", accs, "%")
        os.makedirs(model.opt.directory, exist_ok=True)
        filepath = os.path.join(model.opt.directory, "decoder_" + opt.label_target + "_epoch_" + str(ep) + ".chkpt")
        save(model.model.decoder, model.model.generator, ep, opt, vocab, filepath)
        with torch.no_grad():
            losses, recon, classes, accs = means(compute_epoch(model, class_input, classifier, model.validation_data, True))
        print("Validation Loss:", losses, "(", recon, classes, ")", accs, "%")


if __name__ == "__main__":
    main()
