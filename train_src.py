"""Source-only training entry point: same flags as the reference's train_src.py:37-62
(`-cfg FILE [--local_rank N] KEY VAL ...`) plus `--model` (the reference hard-codes main("gald", ...) at :62
although the DeepLab YAMLs need "aspp").  Under torchrun (WORLD_SIZE > 1) it runs data-parallel over RCCL."""
import argparse
import os

import torch
import torch.distributed as dist

from core.configs import cfg
from core.datasets.build import build_collate_fn, build_dataset
from core.trainers.aspp_trainer import ASPPTrainer


def main(name, cfg, local_rank):
    world = dist.get_world_size() if dist.is_initialized() else 1
    data = build_dataset(cfg, mode="train", is_source=True)
    sampler = torch.utils.data.distributed.DistributedSampler(data, shuffle=True, drop_last=True) if world > 1 else None
    loader = torch.utils.data.DataLoader(
        data, batch_size=max(1, cfg.SOLVER.BATCH_SIZE // world), shuffle=sampler is None, num_workers=4, pin_memory=True,
        collate_fn=build_collate_fn(cfg), sampler=sampler, drop_last=True)
    if name == "aspp":
        ASPPTrainer(name, cfg, loader, local_rank).train()
    elif name == "pranet":                                   # reference train_src.py:29-30
        from core.trainers.pranet_trainer import PraNetTrainer
        PraNetTrainer(name, cfg, loader, local_rank).train()
    elif name == "gald":                                     # reference train_src.py:33-34 (what run.sh launches)
        from core.trainers.gald_trainer import GALDTrainer
        GALDTrainer(name, cfg, loader, local_rank).train()
    else:
        raise NotImplementedError("model %r: 'aspp' (DeepLabV2-ResNet + ASPP), 'pranet' and 'gald' are on the MI355X engine" % name)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="MI355X Semantic Segmentation Training")
    parser.add_argument("-cfg", "--config-file", default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    parser.add_argument("--model", default="aspp", help="trainer to run (the reference edits a literal instead)")
    parser.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(args.local_rank)
        dist.init_process_group(backend="nccl", init_method="env://")     # "nccl" is RCCL on ROCm
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    main(args.model, cfg, args.local_rank)
