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
                # are put back and the state tensors zeroed in place (the fresh state of Adam / AdamW / SGD with momentum), so
                # the first replay is the optimizer's first step.
                with torch.no_grad():
                    saved = [p.detach().clone() for p in params]
                    optimizer.step()
                    for p, q in zip(params, saved):
                        p.copy_(q)
                    for st in optimizer.state.values():
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for p in params:
            p.grad = None                            # captured backward allocates the .grad tensors in the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.per_sample = self._body()
            if optimizer is not None:
                optimizer.step()
        self.grads = [p.grad for p in params]

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
        self.graph.replay()
        return self.loss
