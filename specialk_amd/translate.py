"""Command-line translation entry with the reference's flags (translate.py:17-152) and flow
(translate.py:157-168): load the split checkpoints, beam-search every line of -src, write -output.

    python -m specialk_amd.translate -model transformer -checkpoint_encoder enc.chkpt \
        -checkpoint_decoder dec.chkpt -vocab corpus.pt -src test.txt -output predictions.txt

Extension: `-dtype` (fp32 | bf16).  fp32 reproduces the reference's token sequences bit for bit.
Several GPUs: `python -m torch.distributed.run --nproc-per-node N -m specialk_amd.translate ...` — each rank
decodes a contiguous shard of the sentences, rank 0 writes the joined -output.
"""
import argparse
import os

import torch


def load_args(argv=None):
    p = argparse.ArgumentParser(description="translate.py")
    p.add_argument("-model", choices=["transformer", "recurrent"], required=True)
    p.add_argument("-src", required=True, help="source text, one sentence per line")
    p.add_argument("-vocab", required=True, help="dataset file (*.pt) holding the vocabularies")
    p.add_argument("-checkpoint_encoder", required=True, default="")
    p.add_argument("-checkpoint_decoder", required=True, default="")
    p.add_argument("-output", default="predictions.txt")
    p.add_argument("-copysrc", action="store_true", help="copy source words for UNK (unused by the transformer)")
    p.add_argument("-cuda", action="store_true", help="accepted for compatibility: a GPU is always used")
    p.add_argument("-cuda_device", type=int)
    p.add_argument("-telegram_key")
    p.add_argument("-batch_size", type=int, default=128)
    p.add_argument("-beam_size", type=int, default=5)
    p.add_argument("-n_best", type=int, default=1)
    p.add_argument("-verbose", action="store_true")
    p.add_argument("-telegram", type=str, default="")
    p.add_argument("-override_max_token_seq_len", type=int)
    p.add_argument("-dtype", default="fp32", choices=["fp32", "bf16"])
    opt = p.parse_args(argv)
    opt.save_model = False
    opt.new_directory = False
    return opt


def main(argv=None):
    from .lib.TransformerModel import TransformerModel
    opt = load_args(argv)
    if opt.model != "transformer":
        raise SystemExit("-model recurrent is not implemented in this package (Transformer path only)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # started by torch.distributed.run: one rank per GPU, each decodes its shard of the sentences
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        opt.cuda_device = local
        if not dist.is_initialized():
            dist.init_process_group("gloo")  # barriers only: decoding exchanges no tensors
    elif opt.cuda_device:
        torch.cuda.set_device(opt.cuda_device)
    model = TransformerModel(opt)
    print("Setup model wrapper.")
    model.load(opt.checkpoint_encoder, opt.checkpoint_decoder)
    print("Initiated model and weights.")
    return model.translate()


if __name__ == "__main__":
    main()
