"""Data-parallel path on CPU: world_size 2, gloo backend (what runs over RCCL on the 8-GPU node).
Checks (1) bucket construction, (2) GradAllReducer gives the cross-rank average of the per-rank gradients,
(3) an ASPPTrainer step in a 2-rank job equals a single-process step on the averaged gradient, on every rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _cases
from oracle import ref_model
from rnd_semantic_segmentation_amd.host import config as hc
from rnd_semantic_segmentation_amd.host import ddp, engine, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models():
    fe, cls = ref_model.RefFeatureExtractor((1, 1, 1, 1)), ref_model.RefASPP()
    synth.load_formula_weights(fe)
    synth.load_formula_weights(cls)
    return fe, cls


def _batch(rank):
    x, lab = _cases.net_inputs(1, 33, 70 + rank)
    return torch.from_numpy(x), torch.from_numpy(lab)


def test_bucket_ranges_cover_the_store_from_the_end():
    params = [("p%d" % i, torch.nn.Parameter(torch.zeros(n))) for i, n in enumerate((1000, 64, 5000, 300, 7000, 10))]
    st = engine.FlatStore(params, torch.device("cpu"))
    red = ddp.GradAllReducer([st], bucket_bytes=4 * 6000)
    bks = red.buckets[id(st)]
    assert bks[0].hi == st.total and bks[-1].lo == 0
    for a, b in zip(bks, bks[1:]):
        assert a.lo == b.hi                                   # contiguous, descending
    assert all(b.lo in st.offsets for b in bks)               # cut at parameter boundaries
    assert all(p.grad.data_ptr() == st.grad.data_ptr() + 4 * p._mi_off for _, p in params)


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        # (2) raw reducer
        fe, cls = _models()
        stores = [engine.FlatStore(list(cls.named_parameters()), torch.device("cpu")),
                  engine.FlatStore(list(fe.named_parameters()), torch.device("cpu"))]
        red = ddp.GradAllReducer(stores, bucket_bytes=1 << 20)
        x, lab = _batch(rank)
        for m in (fe, cls):
            for p in m.parameters():
                p.grad = None                                  # as a foreign optimizer's zero_grad would leave it
        loss = torch.nn.functional.cross_entropy(cls(fe(x), lab.shape[-2:]), lab.long(), ignore_index=255)
        loss.backward()
        red.finish()
        np.save(os.path.join(tmpdir, "grad_cls_%d.npy" % rank), stores[0].grad.numpy())
        np.save(os.path.join(tmpdir, "grad_fe_%d.npy" % rank), stores[1].grad.numpy())
        # (3) trainer step in a distributed job
        from rnd_semantic_segmentation_amd.host import trainer as tr
        cfg = hc.CfgNode(hc.default_tree())
        cfg.merge_from_list(["MODEL.NUM_CLASSES", 19, "MODEL.FREEZE_BN", True, "SOLVER.BASE_LR", 5e-4, "OUTPUT_DIR", tmpdir])
        cfg.freeze()

        class T(tr.ASPPTrainer):
            build_feature_extractor = staticmethod(lambda cfg: _models()[0])
            build_classifier = staticmethod(lambda cfg: _models()[1])

        import logging
        t = T("aspp", cfg, [None] * 4, rank, logger=logging.getLogger("ddp%d" % rank))
        assert t.distributed and t.world_size == 2 and t.reducer is not None
        t.train_step(x, lab, 100)
        flat = torch.cat([p.detach().reshape(-1) for p in list(t.feature_extractor.parameters()) + list(t.classifier.parameters())])
        np.save(os.path.join(tmpdir, "params_%d.npy" % rank), flat.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_average_and_trainer_step(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single-process expectation: average of the two per-rank gradients (mean of per-rank mean losses)
    grads = []
    for r in range(2):
        fe, cls = _models()
        x, lab = _batch(r)
        torch.nn.functional.cross_entropy(cls(fe(x), lab.shape[-2:]), lab.long(), ignore_index=255).backward()
        grads.append({k: p.grad.clone() for m in (fe, cls) for k, p in m.named_parameters()})
    avg = {k: (grads[0][k] + grads[1][k]) / 2 for k in grads[0]}
    fe, cls = _models()
    st_cls = engine.FlatStore(list(cls.named_parameters()), torch.device("cpu"))
    st_fe = engine.FlatStore(list(fe.named_parameters()), torch.device("cpu"))
    for st, tag in ((st_cls, "cls"), (st_fe, "fe")):
        want = torch.zeros(st.total)
        for name, p in zip(st.names, st.params):
            want[p._mi_off:p._mi_off + p.numel()] = avg[name].reshape(-1)
        for r in range(2):
            got = np.load(tmp_path / ("grad_%s_%d.npy" % (tag, r)))
            assert np.abs(got - want.numpy()).max() <= 1e-6 * max(np.abs(want.numpy()).max(), 1e-12) + 1e-9
    # trainer: both ranks hold identical parameters == one SGD step on the averaged gradient
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.array_equal(p0, p1)
    fe, cls = _models()
    of, oc = ref_model.make_optimizers(fe, cls, 5e-4)
    for m in (fe, cls):
        for k, p in m.named_parameters():
            p.grad = avg[k].clone()
    of.step()
    oc.step()
    want = torch.cat([p.detach().reshape(-1) for p in list(fe.parameters()) + list(cls.parameters())]).numpy()
    assert np.abs(p0 - want).max() <= 2e-6 * np.abs(want).max()
