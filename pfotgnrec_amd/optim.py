"""Adam over the TGN's flat parameter buffer: one kernel per step (torch.optim.Adam semantics, main.py:123,389)."""
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, tgn, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.tgn = tgn
        super().__init__(list(tgn.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._m = None
        self._v = None
        self._t = 0

    @torch.no_grad()
    def step(self, closure=None):
        tgn = self.tgn
        if tgn.flat_grad is None or any(p.grad is None for p in tgn.hot_parameters()):
            return None
        _lib.require_gpu(tgn.flat_parameters.device)
        if self._m is None or self._m.device != tgn.flat_parameters.device:
            self._m = torch.zeros_like(tgn.flat_parameters)
            self._v = torch.zeros_like(tgn.flat_parameters)
        g = self.param_groups[0]
        self._t += 1
        _lib.call("pfo_adam_step", tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr(), self._m.data_ptr(),
                  self._v.data_ptr(), tgn.flat_parameters.numel(), float(g["lr"]), float(g["betas"][0]),
                  float(g["betas"][1]), float(g["eps"]), self._t, _lib.stream_ptr())
        return None
