"""PraNet (BASELINE config[3]: configs/pranet_src_polyp.yaml, 352 x 352, batch 16) training-step throughput on one MI355X:
forward, four structure losses, backward, clamped Adam.  `python tools/pranet_bench.py [--batch 16] [--size 352] [--steps 20] [--graph]`."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd.host import pranet, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=352)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    net = pranet.PraNet().cuda().train()
    net.ensure_flat()
    opt = pranet.FlatAdam(net, 1e-4 / 8, grad_clamp=0.5)
    img, mask = synth.synth_polyp(a.batch, a.size, a.size, seed=3)
    x, gt = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()

    def step():
        opt.zero_grad()
        ls = [pranet.structure_loss(o, gt) for o in net(x)]
        (ls[3] + ls[2] + ls[1] + ls[0]).backward()
        opt.step()
        return ls[3]

    runner = step
    if a.graph:
        gs = pranet.GraphedStep(net, opt, x, gt)
        runner = lambda: gs()[3]
    for _ in range(a.warmup):
        loss = runner()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = runner()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "PraNet train images/s", "value": round(a.batch / dt, 1), "ms_per_step": round(dt * 1e3, 2), "batch": a.batch, "size": a.size,
                      "hip_graph": bool(a.graph), "loss_lateral2": round(float(loss), 4), "max_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
