"""The style classifier of the reference (classifier/onmt/CNNModels.py:12-65) on the HIP kernels.

`ConvNet` keeps the reference's constructor, attribute names and state_dict layout (word_lut, conv1,
linear), so a classifier checkpoint loads unchanged; its forward pass — probabilities x embedding table,
a convolution over `filter_size` time steps of all embedding channels, ReLU, max over time, Linear,
sigmoid — runs as two matrix products and two small kernels of libspecialk_hip.so.  The network is used
frozen and in eval mode (train_decoder.py:349-350): dropout is the identity and no gradient is produced.

`loss_from_decoder` is the fused form of train_decoder.py:385-424 (class-input Linear, softmax, time
padding / trimming, ConvNet, BCELoss) on decoder rows stored with or without their padding.
"""
import torch
import torch.nn as nn

from .. import _lib as _L

PAD = 0  # classifier/onmt/Constants.py:1


def _ceil8(n):
    return (n + 7) // 8 * 8


class ConvNet(nn.Module):
    def __init__(self, opt, dicts):
        super().__init__()
        self.opt = opt
        self.num_filters = opt.num_filters
        self.filter_size = opt.filter_size
        self.strides = (1, 1)
        self.vocab_size = dicts if type(dicts) == int else dicts.size()
        self.word_vec_size = opt.word_vec_size
        self.word_lut = nn.Embedding(self.vocab_size, opt.word_vec_size, padding_idx=PAD)
        self.conv1 = nn.Conv2d(in_channels=self.word_vec_size, out_channels=self.num_filters,
                               kernel_size=(opt.filter_size, 1), stride=self.strides, bias=True)
        self.linear = nn.Linear(opt.num_filters, opt.num_classes - 1)
        self.compute_dtype = torch.float32
        self._prep = None

    def load_pretrained_vectors(self, opt):
        if opt.pre_word_vecs_enc is not None:
            pretrained = torch.load(opt.pre_word_vecs_enc, map_location="cpu", weights_only=True)
            self.word_lut.weight.data.copy_(pretrained)

    # ------------------------------------------------------------------ derived operand copies
    def _prepared(self):
        """Operands in the layout the kernels read, rebuilt when a weight or the compute type changes:
        the embedding table with its width padded to a multiple of 8 (16-byte rows), the convolution
        weight as (filters, filter_size * width) with the time step outermost, filters padded likewise."""
        ws = (self.word_lut.weight, self.conv1.weight, self.conv1.bias, self.linear.weight, self.linear.bias)
        key = (self.compute_dtype,) + tuple((w.data_ptr(), w._version) for w in ws)
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        if self.linear.weight.shape[0] != 1:
            raise ValueError("the style objective needs a two-class classifier (num_classes = 2)")
        dt, dev = self.compute_dtype, self.word_lut.weight.device
        E, F, fs = self.word_vec_size, self.num_filters, self.filter_size
        Ep, Fp = _ceil8(E), _ceil8(F)
        with torch.no_grad():
            lut = torch.zeros((self.vocab_size, Ep), dtype=dt, device=dev)
            lut[:, :E] = self.word_lut.weight
            w2 = torch.zeros((Fp, fs, Ep), dtype=dt, device=dev)
            w2[:F, :, :E] = self.conv1.weight[..., 0].permute(0, 2, 1)
            cb = torch.zeros(Fp, dtype=torch.float32, device=dev)
            cb[:F] = self.conv1.bias
            self._prep = {"key": key, "lut": lut, "w2": w2.view(Fp, fs * Ep), "conv_bias": cb, "Ep": Ep, "Fp": Fp,
                          "lin_w": self.linear.weight.detach().reshape(-1).float().contiguous(),
                          "lin_b": self.linear.bias.detach().float().contiguous(), "pad_rows": {}}
        return self._prep

    def _windows_to_outputs(self, emb, B, n_steps, target_label, off=None, pad_row=0, stride_b=None, stride_t=1):
        """emb (rows, Ep) embedded probabilities -> (sigmoid outputs (B,), mean BCE (1,))."""
        p = self._prepared()
        S, fs, F = self.opt.sequence_length, self.filter_size, self.num_filters
        P = S - fs + 1
        cols = _L.conv_windows(emb, B, int(n_steps), S, fs, p["Ep"], off=off, pad_row=pad_row, stride_b=stride_b,
                              stride_t=stride_t)
        K = fs * p["Ep"]
        h = _L.gemm(_L.GEMM_NT, cols, p["w2"], B * P, p["Fp"], K, K, K, bias=p["conv_bias"], epilogue=_L.EPI_RELU,
                   split_k=False)
        return _L.cnn_head(h, p["lin_w"], p["lin_b"], B, P, F, target_label)

    # ------------------------------------------------------------------ reference interface
    @torch.no_grad()
    def forward(self, input):
        """input (sequence_length, batch, vocab) class probabilities -> (batch, 1) sigmoid outputs
        (CNNModels.py:48-65).  The pooling window is fixed at construction to cover an input of exactly
        opt.sequence_length steps (:18,36-38), which is what classifier_loss supplies."""
        S, B, V = input.shape
        if S != self.opt.sequence_length or V != self.vocab_size:
            raise ValueError(f"ConvNet expects ({self.opt.sequence_length}, batch, {self.vocab_size}) inputs, got {tuple(input.shape)}")
        p = self._prepared()
        x = input.contiguous().to(self.compute_dtype).view(S * B, V)
        emb = _L.gemm(_L.GEMM_NN, x, p["lut"], S * B, p["Ep"], V, V, p["Ep"])
        y, _ = self._windows_to_outputs(emb, B, S, 1, stride_b=1, stride_t=B)
        return y.view(B, 1)

    # ------------------------------------------------------------------ fused classifier loss
    @torch.no_grad()
    def loss_from_decoder(self, dec2d, B, L_tgt, class_input, target_label, off=None):
        """train_decoder.py:385-424 on decoder outputs dec2d: (B * L_tgt, d) rows of a padded batch
        (pad rows zero, as the decoder leaves them), or — with off (B+1,) int32 — the rows of the
        non-pad positions only, sentence after sentence.  Returns (loss (1,) fp32, class_outputs (B,) fp32).
        No gradient: the reference cuts the classifier off the decoder (`pred.data`, :399) and never
        optimises class_input."""
        p = self._prepared()
        lin = class_input[0] if isinstance(class_input, nn.Sequential) else class_input
        dt = self.compute_dtype
        V, d = lin.weight.shape
        if V != self.vocab_size:
            raise ValueError("class_input must project onto the classifier's vocabulary")
        wkey = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version, dt)
        cw = p["pad_rows"].get("w")
        if cw is None or cw[0] != wkey:
            w = lin.weight.detach().to(dt).contiguous()
            b = lin.bias.detach().float().contiguous()
            # what a zeroed decoder row (a PAD position) embeds to: softmax(bias) x table
            row = _L.softmax_rows(b.to(dt).view(1, V))
            pad_emb = _L.gemm(_L.GEMM_NN, row, p["lut"], 1, p["Ep"], V, V, p["Ep"])
            cw = (wkey, w, b, pad_emb)
            p["pad_rows"]["w"] = cw
        _, w, b, pad_emb = cw
        x = dec2d.detach()
        if x.dtype != dt:
            x = x.to(dt)
        n = x.shape[0]
        probs = _L.gemm(_L.GEMM_NT, x, w, n, V, d, x.stride(0), d, bias=b)
        _L.softmax_rows(probs, out=probs)
        emb = torch.empty((n + 1, p["Ep"]), dtype=dt, device=x.device)
        _L.gemm(_L.GEMM_NN, probs, p["lut"], n, p["Ep"], V, V, p["Ep"], out=emb, ldc=p["Ep"])
        emb[n].copy_(pad_emb[0])
        if off is None:
            if n != B * L_tgt:
                raise ValueError("padded decoder rows must be (B * L_tgt, d)")
            y, loss = self._windows_to_outputs(emb, B, L_tgt, target_label)
        else:
            y, loss = self._windows_to_outputs(emb, B, L_tgt, target_label, off=off, pad_row=n)
        return loss, y

