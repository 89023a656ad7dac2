"""ASPPTrainer: the reference's source-only training loop (core/trainers/aspp_trainer.py:15-145) on the
MI355X engine.  Same constructor, attributes (feature_extractor, classifier, optimizer_fea, optimizer_cls,
iteration, checkpoint ...), checkpoint dict / file names (Aspp-{epoch}.pth), log line format and
aspp_chart_params.json.

Differences, all on the host side:
 * the step is `classifier.loss(feat, label)` (fused ASPP + upsample + CE) when the classifier offers it,
   else criterion(classifier(feat, size), label) exactly like aspp_trainer.py:89-91;
 * loss values stay on the device and are fetched every 20 iterations (when the reference logs), not with a
   `.item()` sync per step (aspp_trainer.py:96,106);
 * WORLD_SIZE > 1: gradients are averaged over ranks by host/ddp.py (RCCL) before the optimizer step, rank 0
   logs and checkpoints; images/s is added to the log line.
"""
import datetime
import os
import time

import torch
import torch.distributed as dist

from . import ddp
from .metrics import MetricLogger, adjust_learning_rate, dump_json, strip_prefix_if_present
from .modules import build_classifier, build_feature_extractor
from .plugin import BaseTrainer
from .sgd import FusedSGD


class ASPPTrainer(BaseTrainer):
    def __init__(self, name, cfg, train_loader, local_rank, logger=None):
        super(ASPPTrainer, self).__init__(name, cfg, train_loader, local_rank, logger)

    # factories are attributes so that tests / other plugins can substitute modules
    build_feature_extractor = staticmethod(build_feature_extractor)
    build_classifier = staticmethod(build_classifier)

    def make_optimizer(self, params, lr):
        cls = FusedSGD if self.device.type == "cuda" else torch.optim.SGD
        return cls(params, lr=lr, momentum=self.cfg.SOLVER.MOMENTUM, weight_decay=self.cfg.SOLVER.WEIGHT_DECAY)

    def init_params(self):
        self.feature_extractor = self.build_feature_extractor(self.cfg)
        self.feature_extractor.to(self.device)
        self.classifier = self.build_classifier(self.cfg)
        self.classifier.to(self.device)
        for m in (self.feature_extractor, self.classifier):
            if hasattr(m, "ensure_flat") and self.device.type == "cuda":
                m.ensure_flat()
        self.optimizer_fea = self.make_optimizer(self._ordered_params(self.feature_extractor), self.cfg.SOLVER.BASE_LR)
        self.optimizer_cls = self.make_optimizer(self._ordered_params(self.classifier), self.cfg.SOLVER.BASE_LR * 10)
        self.iteration = 0
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world_size > 1 else 0
        self.distributed = self.world_size > 1
        self.reducer = None
        if self.distributed or (dist.is_available() and dist.is_initialized() and os.environ.get("MI_DDP_FORCE") == "1"):
            stores = [m.ensure_flat() if hasattr(m, "ensure_flat") else _cpu_store(m) for m in (self.classifier, self.feature_extractor)]
            self.reducer = ddp.GradAllReducer(stores)
            self.reducer.broadcast_parameters(0, modules=(self.feature_extractor, self.classifier))
            if hasattr(self.feature_extractor, "sync_batchnorm") and not self.cfg.MODEL.FREEZE_BN:
                self.feature_extractor.sync_batchnorm(True)          # train_distill.py:53 convert_sync_batchnorm

    @staticmethod
    def _ordered_params(module):
        """module.parameters() order, exactly what the reference hands torch.optim.SGD (aspp_trainer.py:25-26): the optimizer
        state_dict is positional, so `optimizer_fea` / `optimizer_cls` of a checkpoint interchange only in this order
        (the flat store may lay the tensors out differently; FusedSGD matches a store by membership, not by order)."""
        return list(module.parameters())

    def _load_checkpoint(self):
        self.checkpoint = torch.load(self.cfg.resume, map_location=self.device)
        self.feature_extractor.load_state_dict(strip_prefix_if_present(self.checkpoint["feature_extractor"], "module."))
        self.classifier.load_state_dict(strip_prefix_if_present(self.checkpoint["classifier"], "module."))
        if "optimizer_fea" in self.checkpoint:
            self.logger.info("Loading optimizer_fea from {}".format(self.cfg.resume))
            self.optimizer_fea.load_state_dict(self.checkpoint["optimizer_fea"])
        if "optimizer_cls" in self.checkpoint:
            self.logger.info("Loading optimizer_cls from {}".format(self.cfg.resume))
            self.optimizer_cls.load_state_dict(self.checkpoint["optimizer_cls"])
        if "iteration" in self.checkpoint:
            self.iteration = self.checkpoint["iteration"]
        if "epoch" in self.checkpoint:
            self.start_epoch = self.checkpoint["epoch"] + 1

    def _save_checkpoint(self, epoch, save_path):
        clone = lambda sd: {k: v.detach().clone() for k, v in sd.items()}    # un-share the flat storage
        checkpoint = {
            "epoch": epoch,
            "iteration": self.iteration,
            "feature_extractor": clone(self.feature_extractor.state_dict()),
            "classifier": clone(self.classifier.state_dict()),
            "optimizer_fea": self.optimizer_fea.state_dict(),
            "optimizer_cls": self.optimizer_cls.state_dict(),
        }
        torch.save(checkpoint, save_path)

    # ---- HIP-graph mode (MI_GRAPH=1, one GPU): the whole step - zero_grad, forward, loss, hand-written backward on its two
    # streams, both fused SGD launches - is captured once after a few eager steps and replayed; inputs go through static
    # buffers, the learning rate through device memory.  Same kernels, same order per stream: losses are bit-equal to eager.
    GRAPH_WARMUP = 3

    def _graph_enabled(self):
        return (os.environ.get("MI_GRAPH") == "1" and self.device.type == "cuda" and self.reducer is None
                and isinstance(self.optimizer_fea, FusedSGD) and hasattr(self.classifier, "loss"))
        # (one GPU: BatchNorm2d statistics are not exchanged, and their finalize / running-statistics update run on the device - capturable)

    def _graph_signature(self):
        """What a captured step depends on besides the bytes it reads: the flat stores (pointers), the normalisation buffers' versions (a FrozenBN
        refold happens in prepare(), outside the replay) and the train / eval state of the packs."""
        from . import engine
        fe, cls = self.feature_extractor, self.classifier
        sig = [getattr(m, "_store", None) is not None and m._store.data.data_ptr() for m in (fe, cls)]
        eng = getattr(fe, "_engine", None)
        if eng is not None:
            sig.append(sum(engine.bn_versions(rt.bn) for rt in eng.convs) if getattr(fe, "freeze_bn", True) else -1)
            sig.append(bool(eng._have_dgrad))
        return tuple(sig)

    def _graph_step(self, src_input, src_label, current_lr):
        st = getattr(self, "_graph", None)
        if st is None:
            st = self._graph = {"eager": 0, "graph": None}
        if st["graph"] is None:
            if st["eager"] < self.GRAPH_WARMUP or src_input.shape != st.get("shape", src_input.shape):
                st["eager"] += 1
                st["shape"] = src_input.shape
                return None
            st["x"] = torch.empty_like(src_input, device=self.device)
            st["y"] = torch.empty(src_label.shape, dtype=torch.int64, device=self.device)
            for opt in (self.optimizer_fea, self.optimizer_cls):
                opt.set_device_hyper(True)
            st["x"].copy_(src_input, non_blocking=True)
            st["y"].copy_(src_label, non_blocking=True)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["loss"] = self._step_core(st["x"], st["y"])
            st["graph"] = g
            st["sig"] = self._graph_signature()
            # the capture did not execute anything: fall through to the first replay
        if src_input.shape != st["x"].shape or self._graph_signature() != st["sig"]:
            # another batch shape (a last partial batch), or something the capture froze has changed out of band (load_state_dict into the
            # BatchNorm buffers, a repack by an eval in between, parameters moved): this step runs eager, and the graph is rebuilt after
            # the usual warm-up
            self._graph = {"eager": 0, "graph": None}
            return None
        st["x"].copy_(src_input, non_blocking=True)
        st["y"].copy_(src_label, non_blocking=True)
        self.optimizer_fea.push_hyper()
        self.optimizer_cls.push_hyper()
        st["graph"].replay()
        return st["loss"].clone()

    def _step_core(self, src_input, src_label):
        self.optimizer_fea.zero_grad()
        self.optimizer_cls.zero_grad()
        feat = self.feature_extractor(src_input)
        loss = self.classifier.loss(feat, src_label, self.cfg.INPUT.IGNORE_LABEL)
        loss.backward()
        self.optimizer_fea.step()
        self.optimizer_cls.step()
        return loss.detach()

    # The host enqueues a step in ~10 ms and the GPU runs it in ~30 ms: unthrottled, the host runs the whole logging period ahead,
    # and every block the caching allocator sees freed while still in use on a second stream (record_stream) is replaced by a
    # fresh hipMalloc - reserved memory grows by a step's activations per queued step and throughput halves.  Two steps in
    # flight keep the GPU fed and the allocator in steady state.
    RUN_AHEAD = 2

    def _throttle(self):
        if self.device.type != "cuda":
            return
        q = self.__dict__.setdefault("_inflight", [])
        if len(q) >= self.RUN_AHEAD:
            q.pop(0).synchronize()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        q.append(ev)

    def train_step(self, src_input, src_label, max_iter):
        """aspp_trainer.py:77-95 for one minibatch; returns the loss as a device tensor (no sync of the step itself)."""
        self._throttle()
        current_lr = adjust_learning_rate(self.cfg.SOLVER.LR_METHOD, self.cfg.SOLVER.BASE_LR, self.iteration, max_iter,
                                          power=self.cfg.SOLVER.LR_POWER)
        for group in self.optimizer_fea.param_groups:
            group["lr"] = current_lr
        for group in self.optimizer_cls.param_groups:
            group["lr"] = current_lr * 10
        if self._graph_enabled():
            loss = self._graph_step(src_input.to(self.device, non_blocking=True), src_label.to(self.device, non_blocking=True).long(), current_lr)
            if loss is not None:
                return loss, current_lr
        self.optimizer_fea.zero_grad()
        self.optimizer_cls.zero_grad()
        src_input = src_input.to(self.device, non_blocking=True)
        src_label = src_label.to(self.device, non_blocking=True).long()
        feat = self.feature_extractor(src_input)
        if hasattr(self.classifier, "loss"):
            loss = self.classifier.loss(feat, src_label, self.cfg.INPUT.IGNORE_LABEL)
        else:
            output = self.classifier(feat, src_label.shape[-2:])
            loss = torch.nn.functional.cross_entropy(output, src_label, ignore_index=self.cfg.INPUT.IGNORE_LABEL)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer_fea.step()
        self.optimizer_cls.step()
        return loss.detach(), current_lr

    def train(self):
        output_dir = self.cfg.OUTPUT_DIR
        save_to_disk = self.local_rank == 0 and self.rank == 0
        self.iteration = (self.start_epoch - 1) * len(self.train_loader)
        max_iter = self.cfg.SOLVER.EPOCHS * len(self.train_loader)
        if self.rank == 0:
            self.logger.info("#" * 20 + " Start Training " + "#" * 20)
        meters = MetricLogger(delimiter="  ")
        self.feature_extractor.train()
        self.classifier.train()
        start_training_time = time.time()
        end = time.time()
        pending = []          # (device loss, lr) not yet fetched
        images = 0

        def flush():
            bad = getattr(getattr(self.classifier, "_engine", None), "bad_labels", None)
            if bad is not None and pending:                 # the losses are fetched here anyway: one more float, summed over every step since the last flush
                if self.world_size > 1 and torch.distributed.is_initialized():      # every rank must raise (or not) together: a lone ValueError would leave
                    torch.distributed.all_reduce(bad)                               # the other ranks blocked in the next collective
                n = int(bad.item())
                bad.zero_()
                if n:
                    raise ValueError("train labels: %d label values in the last %d steps lie outside [0, %d) and are not ignore_index - "
                                     "torch.nn.CrossEntropyLoss would raise a device assert; map the label ids to train ids first"
                                     % (n, len(pending), self.cfg.MODEL.NUM_CLASSES))
            for l, lr in pending:
                v = float(l)
                meters.update(loss_seg=v)
                self.loss_data.append(v)
                self.lr_data.append(lr)
            pending.clear()

        for epoch in range(self.start_epoch, self.cfg.SOLVER.EPOCHS + 1):
            sampler = getattr(self.train_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):                 # DistributedSampler: reshuffle per epoch like a single-process loader
                sampler.set_epoch(epoch)
            for i, (src_input, src_label, _) in enumerate(self.train_loader):
                data_time = time.time() - end
                loss, lr = self.train_step(src_input, src_label, max_iter)
                pending.append((loss, lr))
                self.iteration += 1
                images += src_input.shape[0] * self.world_size
                log_now = self.iteration % 20 == 0 or self.iteration == max_iter
                if log_now:
                    flush()                                   # the only device sync of the loop
                batch_time = time.time() - end
                end = time.time()
                meters.update(time=batch_time, data=data_time)
                if log_now and self.rank == 0:
                    eta_seconds = meters.time.global_avg * (max_iter - self.iteration)
                    mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if self.device.type == "cuda" else 0.0
                    self.logger.info(meters.delimiter.join([
                        "Epoch: {epoch}", "eta: {eta}", "iter: {iter}", "{meters}", "lr: {lr:.6f}", "max mem: {memory:.0f}",
                        "img/s: {ips:.2f}"]).format(
                        epoch=epoch, eta=str(datetime.timedelta(seconds=int(eta_seconds))), iter=self.iteration, meters=str(meters),
                        lr=self.optimizer_fea.param_groups[0]["lr"], memory=mem, ips=images / max(time.time() - start_training_time, 1e-9)))
            flush()
            if epoch % self.cfg.SOLVER.CHECKPOINT_PERIOD == 0 and save_to_disk:
                os.makedirs(output_dir, exist_ok=True)
                self._save_checkpoint(epoch, os.path.join(output_dir, "Aspp-{}.pth".format(epoch)))
        total_training_time = time.time() - start_training_time
        if self.rank == 0:
            self.logger.info("Total training time: {} ({:.4f} s / epoch)".format(
                str(datetime.timedelta(seconds=total_training_time)), total_training_time / max(self.cfg.SOLVER.EPOCHS, 1)))
        if save_to_disk:
            os.makedirs(output_dir, exist_ok=True)
            dump_json(os.path.join(output_dir, "aspp_chart_params.json"), {"learning rate": self.lr_data, "loss": self.loss_data})


def _cpu_store(module):
    """Flat storage for a module that does not bring its own (stand-in models in multi-process CPU tests)."""
    from .engine import FlatStore
    named = list(module.named_parameters())
    return FlatStore(named, named[0][1].device)
