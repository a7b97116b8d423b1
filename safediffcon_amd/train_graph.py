"""One fine-tuning step -- ``loss = (weight * diffusion(state, mean=False)).mean(); loss.backward()`` (+ ``optimizer.step()``) --
captured ONCE in a hipGraph and replayed.

The reference's inference-time fine-tuning loops run this step hundreds of times between two sampling passes
(1D/inference/inference_ft.py:183-226, tokamak/inference/pipeline.py:238-263, 2d/inference_2d.py:267-279).  On the 1-D nets it is
~1 150 launches of ~25 us: the host (Python + autograd bookkeeping) takes as long as the GPU.  Every node of the training graph
launches libsdc_hip.so kernels on torch's current stream with caller-owned buffers (safediffcon_amd/autograd.py), so the whole
step can be recorded by torch's graph capture (plumbing: HIP stream capture + a private memory pool) and replayed with one
launch.  The weights are packed inside the captured step (PackArena's one launch reads the parameters where they live), so
an optimizer step between two replays -- captured or eager -- is seen by the next one without any notification.
"""
import torch

__all__ = ["GraphedLossStep"]


class GraphedLossStep:
    """``step = GraphedLossStep(diffusion, state, weight=None, t=None, noise=None, optimizer=None)``, then per iteration
    ``loss = step(state, weight, t, noise)``: the inputs are copied into the captured buffers, the graph is replayed, the
    parameters' ``.grad`` tensors hold the new gradients (the same tensors every time) and ``loss`` is the captured loss tensor.

    * ``t`` / ``noise`` None: drawn inside the captured step like ``GaussianDiffusion.forward`` does (torch's Philox state is
      advanced per replay by the graph machinery); given: injected (parity tests).
    * ``optimizer``: a capturable torch optimizer (e.g. ``torch.optim.Adam(..., capturable=True)``) whose ``step()`` is recorded
      behind the backward pass; None: the caller steps its optimizer eagerly between replays.
    * ``per_sample`` after a call: the per-sample losses of the step (``diffusion(state, mean=False)``), as the reference's loops log.
    """

    def __init__(self, diffusion, state, weight=None, t=None, noise=None, optimizer=None, warmup=3):
        if not state.is_cuda:
            raise RuntimeError("safediffcon_amd runs on MI355X only: tensors must be on a cuda (HIP) device; there is no CPU fallback")
        self.gd, self.optimizer = diffusion, optimizer
        dev = state.device
        self.state = state.detach().clone()
        self.weight = torch.ones(state.shape[0], device=dev) if weight is None else weight.detach().clone().to(dev)
        self.t = None if t is None else t.detach().clone().to(dev)
        self.noise = None if noise is None else noise.detach().clone().to(dev)
        params = [p for p in diffusion.model.parameters() if p.requires_grad]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                # warm-up off the default stream: kernel modules, PackArena table, LDS opt-ins
            for _ in range(max(1, warmup)):
                for p in params:
                    p.grad = None
                self._body()
            if optimizer is not None:
                # torch optimizers create their state (moments, step counters) lazily in the first step(): inside the capture
                # that would record the zero-fills and replay them every step.  One step here creates the state; the parameters
                # are put back and the state THIS step created is zeroed in place (the fresh state of Adam / AdamW / SGD with
                # momentum), so the first replay is the optimizer's first step.  State the optimizer already had (it has
                # trained before) is restored from clones: its moments and step counts carry on.
                with torch.no_grad():
                    saved = [p.detach().clone() for p in params]
                    had = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for p, st in optimizer.state.items()}
                    optimizer.step()
                    for p, q in zip(params, saved):
                        p.copy_(q)
                    for p, st in optimizer.state.items():
                        old = had.get(p)
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                if old is not None and torch.is_tensor(old.get(k)):
                                    v.copy_(old[k])
                                else:
                                    v.zero_()
                            elif old is not None and k in old:
                                st[k] = old[k]
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for p in params:
            p.grad = None                            # captured backward allocates the .grad tensors in the graph's pool
        # The captured sdc_pack_batch_run bakes in the addresses of the net's PackArena buffer and pointer table (allocated in the
        # warm-up, outside the graph's pool).  Freeze the arena for this object's lifetime and hold both tensors: forward-only
        # calls on the same net between replays (validation under no_grad, sample(enable_grad=True)) would otherwise let
        # PackArena.begin() rebuild and free them under the graph.
        self._arena = diffusion.model._trainer().arena
        self._pinned = self._arena.freeze()
        self.params = params
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self.loss, self.per_sample = self._body()
                if optimizer is not None:
                    optimizer.step()
        except BaseException:
            self._arena.thaw()
            self._arena = None
            raise
        self.grads = [p.grad for p in params]

    def close(self):
        """drop the graph and let the net's PackArena rebuild again"""
        if getattr(self, "_arena", None) is not None:
            self._arena.thaw()
            self._arena = None
        self.graph = self._pinned = None

    def __del__(self):
        try:
            self.close()
        except Exception:                            # noqa: BLE001  (interpreter shutdown)
            pass

    def _body(self):
        gd = self.gd
        if self.t is None:
            per = gd(self.state, mean=False) if self.noise is None else None
            if per is None:
                raise ValueError("noise without t: pass both or neither")
        else:
            per = gd.p_losses(self.state, self.t, noise=self.noise, mean=False)
        loss = (self.weight * per).mean()
        loss.backward()
        return loss.detach(), per.detach()

    def __call__(self, state=None, weight=None, t=None, noise=None):
        if state is not None:
            self.state.copy_(state)
        if weight is not None:
            self.weight.copy_(weight)
        if t is not None:
            if self.t is None:
                raise ValueError("this step was captured drawing its own timesteps")
            self.t.copy_(t)
        if noise is not None:
            if self.noise is None:
                raise ValueError("this step was captured drawing its own noise")
            self.noise.copy_(noise)
        if self.graph is None:
            raise RuntimeError("GraphedLossStep was closed")
        self.graph.replay()
        # an eager optimizer's zero_grad() (set_to_none=True is torch's default, and the reference's loops call it) drops the
        # captured .grad tensors from the parameters: the replay has just written them, seat them again so step() sees them
        for p, g in zip(self.params, self.grads):
            if p.grad is not g:
                p.grad = g
        return self.loss
