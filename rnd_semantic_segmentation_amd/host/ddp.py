"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce of the modules' flat fp32
gradient buffers over RCCL (torch.distributed backend "nccl" on ROCm) on a side HIP stream, launched while
the rest of backward is still being enqueued.

The reference never wraps its models ('Parallelism'): base/base_trainer.py:23-24 only sets a flag,
and dead train_distill.py:48-64 shows the intended recipe (DDP per module, per-rank batch = BATCH_SIZE/world,
rank-0 checkpointing).  Semantics implemented: gradients are averaged over ranks (mean of per-rank mean
losses, as DistributedDataParallel does); FrozenBN needs no cross-rank statistics.

Backward produces weight gradients in reverse layer order and the flat buffer is laid out in forward order, so
finished gradients form a growing suffix of the buffer: buckets are contiguous ranges cut from the end.
"""
import os

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("lo", "hi", "launched", "work", "bufs")

    def __init__(self, lo, hi):
        self.lo, self.hi, self.launched, self.work, self.bufs = lo, hi, False, None, None


class GradAllReducer:
    def __init__(self, stores, bucket_bytes=32 << 20, process_group=None, overlap=True, payload=None):
        """stores: FlatStore objects in the order their gradients complete during backward.
        payload "fp32" (default) all-reduces the fp32 buckets (ring: each rank sends and receives 2 (N-1)/N x 175.2 MB per step);
        "bf16" (MI_DDP_PAYLOAD=bf16) halves the bytes on the wire at ANY N: a reduce-scatter built from all_to_all_single (rank j receives
        shard j of every rank's bucket as bf16 and adds the N contributions in FP32 in rank order: deterministic), the average rounded to
        bf16 once, then an all-gather of the averaged shards; each rank sends and receives 2 (N-1)/N x 87.6 MB.
        `diag`: what bench.py reports per rank - exposed exchange time (how long the compute stream waited at the join), payload bytes,
        bucket count, the number of ranks the process group really has."""
        self.stores = list(stores)
        if os.environ.get("MI_DDP_BUCKET_MB"):          # tuning knob for the 8-GPU node: bucket size without a code change
            bucket_bytes = int(float(os.environ["MI_DDP_BUCKET_MB"]) * (1 << 20))
        self.payload = payload or os.environ.get("MI_DDP_PAYLOAD", "fp32")
        if self.payload not in ("fp32", "bf16"):
            raise ValueError("payload %r (fp32 | bf16)" % (self.payload,))
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # MI_DDP_FORCE=1: issue the collectives even in a single-rank group (exercises the side-stream / RCCL path on one GPU)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("MI_DDP_FORCE") == "1")
        self.overlap = overlap
        self.buckets = {}
        self.lowwater = {}
        per = max(1, bucket_bytes // 4)
        for st in self.stores:
            bks, hi = [], st.total
            for start in reversed(st.offsets):          # cut at parameter boundaries, from the end of the buffer
                if hi - start >= per:
                    bks.append(_Bucket(start, hi))
                    hi = start
            if hi > 0:
                bks.append(_Bucket(0, hi))
            self.buckets[id(st)] = bks
            self.lowwater[id(st)] = st.total
            st.grad_hooks.append(self._on_ready)
        dev = self.stores[0].grad.device
        self.cuda = dev.type == "cuda"
        self.side = torch.cuda.Stream(device=dev) if self.cuda else None
        self.measure = False                    # bench.py: time the join on the compute stream with a HIP event pair per step
        self._joins = []
        n_elem = sum(st.total for st in self.stores)
        self.diag = {"ranks": self.world, "active": bool(self.active), "payload": self.payload, "buckets": sum(len(b) for b in self.buckets.values()),
                     "payload_bytes_per_step": n_elem * (2 if self.payload == "bf16" else 4),
                     "wire_bytes_per_rank_per_step": int(2 * (self.world - 1) / max(self.world, 1) * n_elem * (2 if self.payload == "bf16" else 4)),
                     "overlap": bool(overlap), "bucket_bytes": int(bucket_bytes),
                     # what bounds RCCL's footprint beside the compute kernels: its channel count = workgroups per collective (section 6)
                     "rccl_env": {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO") if k in os.environ}}

    def _exchange(self, chunk, bucket=None):
        """Enqueue the average of `chunk` over ranks (in place); returns an async work handle or None (already complete)."""
        if self.payload == "fp32":
            chunk.div_(self.world)                           # pre-divide: sum of 1/N-scaled == average
            return dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return self._exchange_bf16(chunk, bucket)

    def _exchange_bf16(self, chunk, b):
        """Reduce-scatter (all_to_all_single of bf16 shards + fp32 sum in rank order) and all-gather of the averaged bf16 shards."""
        n, W = chunk.numel(), self.world
        L = -(-n // W)
        if b is None or b.bufs is None or b.bufs[0].numel() != W * L:
            bufs = (torch.zeros(W * L, dtype=torch.bfloat16, device=chunk.device), torch.empty(W * L, dtype=torch.bfloat16, device=chunk.device),
                    torch.empty(L, dtype=torch.bfloat16, device=chunk.device), torch.empty(W * L, dtype=torch.bfloat16, device=chunk.device))
            if b is not None:
                b.bufs = bufs
        else:
            bufs = b.bufs
        send, recv, shard, full = bufs
        send[:n].copy_(chunk)                                   # fp32 -> bf16 (the padding past n stays zero)
        dist.all_to_all_single(recv, send, group=self.pg)       # recv[r*L:(r+1)*L] = rank r's copy of MY shard
        parts = recv.view(W, L)
        acc = parts[0].float()
        for r in range(1, W):
            acc.add_(parts[r].float())                          # rank order: the same bits on every rank
        shard.copy_(acc.div_(W))                                # one rounding of the average
        dist.all_gather_into_tensor(full, shard, group=self.pg)
        chunk.copy_(full[:n])
        return None

    def _launch(self, st, b):
        if b.launched or not self.active:
            b.launched = True
            return
        b.launched = True
        chunk = st.grad[b.lo:b.hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                b.work = self._exchange(chunk, b)
        else:
            b.work = self._exchange(chunk, b)

    def _on_ready(self, st, lo, hi):
        """Engine callback: gradients of [lo, hi) are enqueued; everything above the low-water mark is final."""
        if not self.overlap:
            return
        key = id(st)
        self.lowwater[key] = min(self.lowwater[key], lo)
        for b in self.buckets[key]:
            if not b.launched and b.lo >= self.lowwater[key]:
                self._launch(st, b)

    def _adopt_foreign_grads(self, st):
        """Parameters whose .grad is not the flat view (plain autograd + foreign optimizer): copy in, re-point."""
        for p in st.params:
            view = st.grad_view(p)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view

    def finish(self):
        """Call after loss.backward(), before optimizer.step(): reduce what is left, then make the compute
        stream wait for the side stream."""
        for st in self.stores:
            self._adopt_foreign_grads(st)
            for b in self.buckets[id(st)]:
                self._launch(st, b)
        for st in self.stores:
            for b in self.buckets[id(st)]:
                if b.work is not None:
                    b.work.wait()
                    b.work = None
                b.launched = False
            self.lowwater[id(st)] = st.total
        if self.cuda and self.active:
            cur = torch.cuda.current_stream()
            if self.measure:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.side)
                e1.record(cur)
                self._joins.append((e0, e1))
            else:
                cur.wait_stream(self.side)

    def exposed_ms(self):
        """Mean time per step the compute stream spent waiting for the exchange at the join (after a synchronize); None if not measured."""
        if not self._joins:
            return None
        v = [a.elapsed_time(b) for a, b in self._joins]
        self._joins = []
        return sum(v) / len(v)

    def broadcast_parameters(self, src=0, modules=()):
        """Rank-0 weights everywhere before the first step (what DDP's constructor does) - and, like DDP, the BUFFERS of `modules`
        (BatchNorm running statistics / num_batches_tracked): the synchronised BatchNorm path uses the running mean as the pilot of its
        one-pass variance, which is only correct if every rank holds the same bits."""
        if not self.active:
            return
        for st in self.stores:
            dist.broadcast(st.data, src=src, group=self.pg)
            st.generation += 1
        for m in modules:
            for b in m.buffers():
                dist.broadcast(b.data, src=src, group=self.pg)
