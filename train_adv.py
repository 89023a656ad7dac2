"""Adversarial (FADA) training entry point: same flags as the reference's train_adv.py:62-105
(`-cfg FILE [--local_rank N] KEY VAL ...`) plus `--model` (the reference hard-codes main("gald_fada", ...) at :105 although
the DeepLab YAMLs need "aspp_fada").  Source and target loaders each carry BATCH_SIZE // 2 images (train_adv.py:29,39),
the target set is repeated 9x (:17).  Under torchrun (WORLD_SIZE > 1) it runs data-parallel over RCCL."""
import argparse
import os

import torch
import torch.distributed as dist
from torch.utils.data import ConcatDataset

from core.combos.aspp_fada import AsppFada
from core.configs import cfg
from core.datasets.build import build_collate_fn, build_dataset


def main(name, cfg, local_rank):
    world = dist.get_world_size() if dist.is_initialized() else 1
    src = build_dataset(cfg, mode="train", is_source=True)
    tgt = ConcatDataset([build_dataset(cfg, mode="train", is_source=False)] * 9)
    per_rank = max(1, cfg.SOLVER.BATCH_SIZE // 2 // world)

    def loader(data, collate):
        sampler = torch.utils.data.distributed.DistributedSampler(data, shuffle=True, drop_last=True) if world > 1 else None
        return torch.utils.data.DataLoader(data, batch_size=per_rank, shuffle=sampler is None, num_workers=4, pin_memory=True,
                                           collate_fn=collate, sampler=sampler, drop_last=True)

    if name != "aspp_fada":
        raise NotImplementedError("combo %r: only 'aspp_fada' (DeepLabV2-ResNet + ASPP + FADA) is on the MI355X hot path" % name)
    AsppFada(name, cfg, loader(src, build_collate_fn(cfg)), loader(tgt, None), local_rank).train()


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="MI355X Semantic Segmentation Adversarial Training")
    parser.add_argument("-cfg", "--config-file", default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    parser.add_argument("--model", default="aspp_fada", help="combo to run (the reference edits a literal instead)")
    parser.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(args.local_rank)
        dist.init_process_group(backend="nccl", init_method="env://")     # "nccl" is RCCL on ROCm
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    main(args.model, cfg, args.local_rank)
