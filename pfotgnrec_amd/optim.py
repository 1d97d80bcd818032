"""Adam over the TGN's flat parameter buffer: one kernel per step (torch.optim.Adam semantics, main.py:123,389).

torch.optim.Adam keeps one step counter per parameter tensor and skips tensors whose ``.grad`` is None.  The
reference hits that on the first batch after every ``__init_memory__`` (main.py:153): no message is pending, the GRU
is never called (memory_updater.py:38-40) and its four tensors stay one step behind for the rest of the run.  The same
bookkeeping is done here on the host; the device work is still ONE launch (``pfo_adam_step_ranges``: contiguous runs of
tensors with equal step counts).
"""
import ctypes

import torch

from . import _lib


class _StepCounts(dict):
    """parameter -> steps taken, stored under id(parameter) (hashing a tensor is a Python-level call: ~100 of them per step);
    tensors and ids are both accepted as keys."""
    @staticmethod
    def _k(key):
        return key if isinstance(key, int) else id(key)

    def __getitem__(self, key):
        return dict.__getitem__(self, self._k(key))

    def __setitem__(self, key, value):
        dict.__setitem__(self, self._k(key), value)

    def __contains__(self, key):
        return dict.__contains__(self, self._k(key))

    def get(self, key, default=None):
        return dict.get(self, self._k(key), default)

    def copy(self):
        return _StepCounts(self)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, tgn, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, zero_grads_in_step=False, overlap_backward=False):
        """``overlap_backward``: for the reference's own loop (main.py:160-394), which reads ``loss.item()`` after every batch.
        ``loss.backward()`` leaves the native backward on a stream of its own and ``step()`` queues the optimizer's kernel behind
        it there, so that ``loss.item()`` waits for the forward only and the host prepares the next batch (candidate draw,
        uploads, launches) while the device is still differentiating; the next forward - and ``state_dict()``, ``tgn.join()`` -
        wait for the step.  Gradients and parameters are IN FLIGHT after ``backward()`` / ``step()`` until then: read ``.grad`` or
        a parameter only behind ``tgn.join()``.  One rank, eager mode; anything else takes the serial order.
        ``zero_grads_in_step``: a side-stream step (``step(side=True)``, i.e. ``bpr_step(..., optimizer=)``) also CLEARS the
        gradients it consumed - ``optimizer.zero_grad()`` folded into the optimizer's kernel, for loops that zero the gradients
        right after the step anyway (main.py:388-390 does): the next native backward then clears nothing on its critical path.
        ``.grad`` reads zero after such a step."""
        self.tgn = tgn
        self.zero_grads_in_step = bool(zero_grads_in_step)
        if overlap_backward:
            tgn.overlap_backward = True
        super().__init__(list(tgn.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._m = None
        self._v = None
        self._steps = _StepCounts() # id(parameter) -> steps taken (torch.optim.Adam's state[p]["step"]; ids: hashing a tensor is a Python call)
        self._sorted_views = None   # (the model's views in flat-buffer order, sorted once)

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            self.tgn.join()                  # (the clear is a torch kernel on the caller's stream)
        return super().zero_grad(set_to_none=set_to_none)

    # tests / checkpoint restore: one step count for every tensor
    @property
    def _t(self):
        return max(self._steps.values(), default=0)

    @_t.setter
    def _t(self, value):
        self._steps = _StepCounts({id(p): int(value) for p in self.tgn.hot_parameters()})

    def sync_steps(self, taken):
        """Adds ``taken`` steps to every tensor that has a gradient (after replaying a captured step ``taken`` times)."""
        for p in self.tgn.hot_parameters():
            if p.grad is not None:
                self._steps[id(p)] = self._steps.get(id(p), 0) + int(taken)

    def set_steps(self, steps_by_name):
        """Per-tensor step counts (``{parameter name: steps taken}``), e.g. from a torch.optim.Adam state dict."""
        names = dict(self.tgn.named_parameters())
        for k, t in steps_by_name.items():
            self._steps[id(names[k])] = int(t)

    @torch.no_grad()
    def step(self, closure=None, step_dev=None, side=False):
        """``step_dev`` (1-element int32 device tensor): the step counts used are the host's plus that device word and the
        host counters are NOT advanced - for steps captured into a HIP graph, which advances the word itself
        (``sync_steps`` folds it back into the host counters afterwards).
        ``side``: the kernel goes to the library's side stream, behind a backward that ran with ``defer_join``
        (``functional.bpr_step(..., optimizer=self)``); otherwise anything pending there is joined first."""
        tgn = self.tgn
        if (not side and step_dev is None and getattr(tgn, "_bwd_event", None) is not None and tgn._overlap_ok()
                and _lib.stream_ptr() != tgn._backward_stream().cuda_stream):
            # behind the backward on ITS stream (overlap_backward); the event the next forward waits for moves behind the step
            bwd = tgn._backward_stream()
            with _lib.on_stream(bwd):
                out = self._step(closure, None, False)
                ev = torch.cuda.Event()
                ev.record(bwd)
            tgn._set_backward_event(ev)
            return out
        return self._step(closure, step_dev, side)

    def _step(self, closure, step_dev, side):
        tgn = self.tgn
        if side and (self._m is None or self._m.device != tgn.flat_parameters.device):
            side = False            # the moments are allocated (and cleared) on the caller's stream below: this one step runs there
        if not side:
            tgn.join()
            comm = getattr(tgn, "_comm_pending", None)
            if comm is not None:                     # (a serial step behind allreduce_flat_grad_ordered: an empty shard's route)
                tgn._comm_pending = None
                torch.cuda.current_stream(tgn.flat_parameters.device).wait_stream(comm)
        if tgn.flat_grad is None:
            return None
        _lib.require_gpu(tgn.flat_parameters.device)
        if self._m is None or self._m.device != tgn.flat_parameters.device:
            self._m = torch.zeros_like(tgn.flat_parameters)
            self._v = torch.zeros_like(tgn.flat_parameters)
        g = self.param_groups[0]
        lo, hi, st = [], [], []
        sv = self._sorted_views
        if sv is None or sv[0] is not tgn._views or len(sv[1]) != len(tgn._views):
            sv = self._sorted_views = (tgn._views, [(p, off, n, id(p)) for p, off, n, _ in sorted(tgn._views, key=lambda v: v[1])])
        steps, inc = self._steps, (1 if step_dev is None else 0)
        for p, off, n, key in sv[1]:
            if p.grad is None:                       # torch.optim.Adam: skipped entirely (no moment decay, no step)
                continue
            t = steps.get(key, 0) + inc
            if inc:
                steps[key] = t
            if lo and hi[-1] == off and st[-1] == t:
                hi[-1] = off + n
            else:
                lo.append(off); hi.append(off + n); st.append(t)
        MAXR = 16
        # the kernel may clear what it read when the ranges cover the whole flat buffer (every tensor has a gradient)
        covered = bool(lo) and lo[0] == 0 and hi[-1] == tgn.flat_parameters.numel() and all(hi[j] == lo[j + 1] for j in range(len(lo) - 1))
        zero_all = bool(side and self.zero_grads_in_step and step_dev is None and covered and len(lo) <= MAXR)
        ordered = side and getattr(tgn, "dp_ordered", False) and step_dev is None and 0 < tgn.grad_split < tgn.flat_parameters.numel()
        if ordered:
            # Two buckets in order of first use: [0, split) = time encoder, GRU, layer 1 - the next forward reads them on the
            # caller's stream ~70 us into the step and waits for THIS kernel alone; [split, total) = the top layer's block, whose
            # all-reduce ran beside the backward - stepped behind it, met by the next forward through the side stream's order.
            split = tgn.grad_split
            cut_lo, cut_hi, cut_st = [], [], []
            for a, b, t in zip(lo, hi, st):
                if a < split < b:
                    cut_lo += [a, split]; cut_hi += [split, b]; cut_st += [t, t]
                else:
                    cut_lo.append(a); cut_hi.append(b); cut_st.append(t)
            first = [j for j in range(len(cut_lo)) if cut_hi[j] <= split]
            later = [j for j in range(len(cut_lo)) if cut_lo[j] >= split]
            args = (tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr())
            hyp = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))

            zflag = 4 if zero_all else 0

            def run(idx, last_bucket):
                for i in range(0, len(idx), MAXR):
                    part = idx[i:i + MAXR]
                    k = len(part)
                    bucket = last_bucket if i + MAXR >= len(idx) else (0 if last_bucket == 1 else 2)
                    _lib.call("pfo_tgn_adam_side_bucket", *args, k, (ctypes.c_int64 * k)(*[cut_lo[j] for j in part]),
                              (ctypes.c_int64 * k)(*[cut_hi[j] for j in part]), (ctypes.c_int32 * k)(*[cut_st[j] for j in part]), *hyp, bucket | zflag)
            if first and later:
                run(first, 1)
                tgn.wait_comm_stream()
                run(later, 2)
                tgn.parameters_changed(refresh=False)
                tgn._grad_zeroed = zero_all
                return None
            tgn.wait_comm_stream()                    # (nothing to cut: the plain side step below, behind the top block's all-reduce)
        elif side and getattr(tgn, "_comm_pending", None) is not None:
            tgn.wait_comm_stream()
        for i in range(0, len(lo), MAXR):
            k = min(MAXR, len(lo) - i)
            if step_dev is not None:
                _lib.call("pfo_adam_step_ranges_dev", tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr(), self._m.data_ptr(),
                          self._v.data_ptr(), k, (ctypes.c_int64 * k)(*lo[i:i + k]), (ctypes.c_int64 * k)(*hi[i:i + k]),
                          (ctypes.c_int32 * k)(*st[i:i + k]), step_dev.data_ptr(), float(g["lr"]), float(g["betas"][0]),
                          float(g["betas"][1]), float(g["eps"]), _lib.stream_ptr())
                continue
            if side and zero_all:
                _lib.call("pfo_tgn_adam_side_bucket", tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr(), self._m.data_ptr(),
                          self._v.data_ptr(), k, (ctypes.c_int64 * k)(*lo[i:i + k]), (ctypes.c_int64 * k)(*hi[i:i + k]),
                          (ctypes.c_int32 * k)(*st[i:i + k]), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                          float(g["eps"]), 4)
                tgn._grad_zeroed = True
                continue
            if side:
                _lib.call("pfo_tgn_adam_side", tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr(), self._m.data_ptr(),
                          self._v.data_ptr(), k, (ctypes.c_int64 * k)(*lo[i:i + k]), (ctypes.c_int64 * k)(*hi[i:i + k]),
                          (ctypes.c_int32 * k)(*st[i:i + k]), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                          float(g["eps"]))
                continue
            _lib.call("pfo_adam_step_ranges", tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr(), self._m.data_ptr(),
                      self._v.data_ptr(), k, (ctypes.c_int64 * k)(*lo[i:i + k]), (ctypes.c_int64 * k)(*hi[i:i + k]),
                      (ctypes.c_int32 * k)(*st[i:i + k]), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                      float(g["eps"]), _lib.stream_ptr())
        if lo:
            # the kernel above wrote the parameters behind torch's back: the model's parameter cache (composite weights, weight
            # images) is rebuilt right here, on the library's side stream behind this kernel - beside the next batch's sampling
            # phase - so that the next forward launches none of it.  (A step being captured rebuilds inside its own forward.)
            tgn.parameters_changed(refresh=tgn.refresh_after_step and step_dev is None and not side)
        return None


def overlap_backward(tgn, optimizer):
    """``FusedAdam(tgn, overlap_backward=True)`` for ANY torch optimizer over ``tgn.parameters()`` (main.py:123 builds
    ``torch.optim.Adam``): ``loss.backward()`` leaves the native backward on a stream of its own, and ``optimizer.step()`` -
    wrapped here - runs its kernels on that stream behind it, so the loop's ``loss.item()`` (main.py:390) waits for the forward
    only and the host prepares the next batch beside the backward.  The next forward, ``state_dict()`` and ``tgn.join()`` wait
    for the step; read ``.grad`` or a parameter from torch only behind ``tgn.join()``.  ``optimizer.zero_grad(set_to_none=
    False)`` joins first (its kernels run on the caller's stream).  Returns the optimizer."""
    tgn.overlap_backward = True
    if isinstance(optimizer, FusedAdam) or getattr(optimizer, "_pfo_overlap", None) is tgn:
        return optimizer
    plain_step, plain_zero = optimizer.step, optimizer.zero_grad

    def step(*args, **kwargs):
        if getattr(tgn, "_bwd_event", None) is None or not tgn._overlap_ok():
            tgn.join()
            return plain_step(*args, **kwargs)
        bwd = tgn._backward_stream()
        with _lib.on_stream(bwd):
            out = plain_step(*args, **kwargs)
            ev = torch.cuda.Event()
            ev.record(bwd)
        tgn._set_backward_event(ev)
        return out

    def zero_grad(set_to_none=True):
        if not set_to_none:
            tgn.join()
        return plain_zero(set_to_none=set_to_none)

    optimizer.step, optimizer.zero_grad, optimizer._pfo_overlap = step, zero_grad, tgn
    return optimizer
