"""torch.optim.SGD(momentum, weight_decay) with the update done by one HIP kernel over a module's flat
parameter buffer (reference core/trainers/aspp_trainer.py:25-26, 94-95).

Subclasses torch.optim.SGD so param_groups / state_dict() / load_state_dict() keep torch's format and the
reference's checkpoints (keys optimizer_fea / optimizer_cls) interchange.  The arithmetic is
g' = g + wd*p; buf = mu*buf + g'; p -= lr*buf, with buf starting at zero (== torch's first-step buf = g').
"""
import torch

from .. import kernels as K


class FusedSGD(torch.optim.SGD):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, dampening=0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("FusedSGD implements the reference's configuration: dampening=0, nesterov=False")
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self._flat_mom = {}          # id(store) -> flat momentum buffer
        self.device_hyper = None     # 3-float device tensor (lr, momentum, weight_decay): set by set_device_hyper() for HIP-graph replay

    def set_device_hyper(self, enable=True):
        """Graph mode: step() reads lr / momentum / weight_decay of group 0 from a device tensor that push_hyper() refreshes from
        param_groups before every replay (kernel arguments passed by value would be frozen into the captured graph)."""
        if not enable:
            self.device_hyper = None
            return
        dev = self.param_groups[0]["params"][0].device
        g = self.param_groups[0]
        self.device_hyper = torch.tensor([g["lr"], g["momentum"], g["weight_decay"]], dtype=torch.float32, device=dev)

    def push_hyper(self):
        """Stream-ordered update of the learning rate (a fill kernel carrying the value as its argument: no host staging buffer
        that a later step could overwrite before the copy has run)."""
        self.device_hyper[0:1].fill_(float(self.param_groups[0]["lr"]))

    def zero_grad(self, set_to_none=True):
        """Engine-written gradients are OVERWRITTEN by the next backward (no memset of 170 MB); gradients that
        autograd accumulates (the stem conv) are zeroed."""
        stores = set()
        for group in self.param_groups:
            for p in group["params"]:
                st = getattr(p, "_mi_store", None)
                if st is not None and st.owns(p):
                    stores.add(st)
                    if id(p) in st.written:
                        continue
                    if p.grad is not None:
                        p.grad.zero_()
                elif p.grad is not None:
                    p.grad = None if set_to_none else p.grad.zero_()
        for st in stores:
            st.written.clear()

    def _momentum_view(self, p, st):
        flat = self._flat_mom.get(id(st))
        if flat is None:
            flat = self._flat_mom[id(st)] = torch.zeros_like(st.data)
        view = flat[p._mi_off:p._mi_off + p.numel()].view_as(p)
        state = self.state[p]
        cur = state.get("momentum_buffer")
        if cur is None:
            state["momentum_buffer"] = view
        elif cur.data_ptr() != view.data_ptr():      # restored by load_state_dict: adopt the values
            view.copy_(cur)
            state["momentum_buffer"] = view
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            lr, mu, wd = group["lr"], group["momentum"], group["weight_decay"]
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if not params[0].is_cuda:
                raise RuntimeError("FusedSGD updates parameters on the MI355X; got %s" % params[0].device)
            st = getattr(params[0], "_mi_store", None)
            # the group covers a whole flat store (in any order: the optimizer keeps module.parameters() order for
            # state_dict interchange with the reference, the store keeps the engine's layout)
            whole = (st is not None and len(params) == len(st.params) and {id(p) for p in params} == {id(q) for q in st.params}
                     and st.intact() and all(p.grad.data_ptr() == st.grad.data_ptr() + 4 * p._mi_off for p in params))
            if whole:
                for p in params:
                    flat = self._momentum_view(p, st)
                if self.device_hyper is not None:
                    if not torch.cuda.is_current_stream_capturing():     # an eager step between graph replays: refresh the device copy
                        self.push_hyper()
                    K.sgd_step_dev(st.data, st.grad, flat, self.device_hyper)
                else:
                    K.sgd_step(st.data, st.grad, flat, lr, mu, wd)       # one launch for the whole module
                st.generation += 1
            else:
                for p in params:
                    state = self.state[p]
                    if "momentum_buffer" not in state:
                        state["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    K.sgd_step(p.data, g, state["momentum_buffer"], lr, mu, wd)
                    ps = getattr(p, "_mi_store", None)
                    if ps is not None:
                        ps.generation += 1
        return loss
