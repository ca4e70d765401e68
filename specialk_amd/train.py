"""Command-line training entry with the reference's flags (train.py:22-361) and epoch loop
(train.py:364-399).

    python -m specialk_amd.train -data corpus.pt -model transformer -epochs 10 -save_model ...
    python -m torch.distributed.run --nproc-per-node 8 -m specialk_amd.train -data ... -multi_gpu

Flags keep the reference's names, types and defaults.  Flags of the recurrent model are accepted
so existing command lines parse, but `-model recurrent` is refused: only the Transformer path is
implemented here.  Extensions: `-dtype` (fp32 | bf16 compute) and `-seed`.

Multi-GPU: the reference wraps encoder and decoder in nn.DataParallel inside one process
(train.py:385-388).  Here `-multi_gpu` means one process per GPU, started by torch.distributed.run;
the batch of `-batch_size` sentences is split across ranks and gradients are summed over RCCL.
"""
import argparse
import os

import torch


def load_args():
    p = argparse.ArgumentParser(description="train.py")
    # data options
    p.add_argument("-data", required=True, help="path to the *.pt dataset written by preprocess.py")
    p.add_argument("-checkpoint_encoder", default="", help="encoder checkpoint to continue from")
    p.add_argument("-checkpoint_decoder", default="", help="decoder checkpoint to continue from")
    p.add_argument("-new_directory", action="store_true",
                   help="store checkpoints in a new directory instead of next to the loaded encoder")
    p.add_argument("-log", action="store_true", help="keep the validation outputs of every epoch")
    p.add_argument("-directory_name", type=str, default="", help="name of the run directory")
    p.add_argument("-save_model", action="store_true", help="save checkpoints and logs")
    p.add_argument("-save_mode", default="all", choices=["all", "best"])
    p.add_argument("-verbose", action="store_true")
    # model options
    p.add_argument("-model", choices=["transformer", "recurrent"], required=True)
    p.add_argument("-cuda", action="store_true", help="accepted for compatibility: a GPU is always used")
    p.add_argument("-multi_gpu", action="store_true", help="data parallel over the ranks of torch.distributed.run")
    p.add_argument("-cuda_device", type=int, help="GPU index (single-process runs)")
    p.add_argument("-batch_size", type=int, default=64)
    p.add_argument("-layers", type=int, default=6)
    p.add_argument("-d_word_vec", type=int, default=300)
    # training options
    p.add_argument("-epochs", type=int, required=True, default=10)
    p.add_argument("-dropout", type=float, default=0.1)
    p.add_argument("-telegram", type=str, default="", help="json file with a Telegram key and chat id")
    # transformer options
    p.add_argument("-d_model", type=int, default=512)
    p.add_argument("-d_inner_hid", type=int, default=2048)
    p.add_argument("-d_k", type=int, default=64)
    p.add_argument("-d_v", type=int, default=64)
    p.add_argument("-n_head", type=int, default=8)
    p.add_argument("-n_warmup_steps", type=int, default=4000)
    p.add_argument("-embs_share_weight", action="store_true")
    p.add_argument("-proj_share_weight", action="store_true")
    p.add_argument("-label_smoothing", action="store_true")
    # recurrent-model options (parsed, unused)
    p.add_argument("-max_generator_batches", type=int, default=32)
    p.add_argument("-input_feed", type=int, default=0)
    p.add_argument("-max_grad_norm", type=float, default=5)
    p.add_argument("-curriculum", action="store_true")
    p.add_argument("-brnn", action="store_true")
    p.add_argument("-brnn_merge", default="concat")
    p.add_argument("-rnn_size", type=int, default=500)
    # learning rate
    p.add_argument("-learning_rate", type=float, default=0.001)
    p.add_argument("-learning_rate_decay", type=float, default=0.5)
    p.add_argument("-start_decay_at", type=int, default=8)
    p.add_argument("-optim", default="adam", choices=["sgd", "adagrad", "adadelta", "adam"])
    # style-transfer options
    p.add_argument("-freeze_encoder", action="store_true")
    p.add_argument("-freeze_decoder", action="store_true")
    # extensions
    p.add_argument("-dtype", default="fp32", choices=["fp32", "bf16"],
                   help="arithmetic of the matrix products (fp32 master weights either way)")
    p.add_argument("-seed", type=int, default=0, help="seed of initialisation, shuffling and dropout")
    p.add_argument("-batch_tokens", type=int, default=0,
                   help="> 0: length-bucketed batches of at most this many tokens (and -batch_size sentences) "
                        "instead of the reference's fixed sentence count")
    p.add_argument("-device_collate", action="store_true",
                   help="pad the batches on the GPU (flat ids + offsets are copied, a kernel writes seq / pos)")
    return p


def init_distributed(opt):
    """Join the process group torch.distributed.run prepared (one rank per GPU)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        opt.cuda_device = local
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif opt.cuda_device is not None:
        torch.cuda.set_device(opt.cuda_device)
    return world


def train_model(opt):
    from . import ops
    from .lib.TransformerModel import TransformerModel
    if opt.model != "transformer":
        raise SystemExit("-model recurrent is not implemented in this package (Transformer path only)")
    init_distributed(opt)
    torch.manual_seed(opt.seed)      # same seed on every rank: same initial weights and batch order
    ops.manual_seed(opt.seed)
    model = TransformerModel(opt)
    print("Setup model wrapper.")
    model.load_dataset()
    print("Loaded data.")
    if opt.checkpoint_encoder:
        model.load(opt.checkpoint_encoder, opt.checkpoint_decoder)
        print("Loaded model encoder and decoder." if opt.checkpoint_decoder else "Loaded model encoder.")
    else:
        model.initiate()
        print("Initiated model and weights.")
    for part, frozen in ((model.model.encoder, opt.freeze_encoder), (model.model.decoder, opt.freeze_decoder),
                         (model.model.generator, opt.freeze_decoder)):
        if frozen:
            for prm in part.parameters():
                prm.requires_grad_(False)
    model.setup_optimiser()
    print("Training model.")
    if model.opt.save_model:
        model.init_logs()
    for epoch in range(1, model.opt.epochs + 1):
        model.opt.current_epoch = epoch
        model.train(epoch)
        if model.opt.save_model:
            model.save(epoch=epoch, note="epoch_" + str(epoch))
            model.update_logs(epoch)
    print("Done.")
    return model


if __name__ == "__main__":
    options = load_args().parse_args()
    assert options.epochs > 0
    train_model(options)
