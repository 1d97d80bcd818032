"""A whole training step captured into a HIP graph (``torch.cuda.CUDAGraph``) - for the launch-bound regime.

A step queues ~85 kernels; below a few hundred interactions per batch the host's launch rate, not the GPU, sets the
step time (C1: 0.48 ms, C2 at batch 128: 1.12 ms eager).  Capturing the step replays all of them with one call.
What makes the step capturable:

* every kernel argument that changes from step to step lives in device memory: the batch is copied into static
  tensors, the Philox stream position (``pfo_tgn_batch.offset_dev``, ``pfo_neg_draw_dev``) and Adam's step count
  (``pfo_adam_step_ranges_dev``) are device words the graph advances itself;
* the library allocates nothing and never synchronises; its internal side stream forks from and re-joins the
  capturing stream by events, so it is captured with it;
* gradients stay attached (the captured backward clears the flat buffer itself instead of ``zero_grad(set_to_none=True)``).

Same arithmetic as the eager step (``tests/test_gpu_round2.py::test_graphed_step_equals_eager_step``).  Single GPU;
the baseline path (``compute_temporal_embeddings`` + BPR, main.py:345-394) and the ``ours`` path (MV selection) of
``bench.py``.
"""
import torch

from . import _lib
from .functional import bpr_step


class GraphedTrainStep:
    def __init__(self, tgn, optimizer, sampler, batch, n_neighbors, n_neg=3, port_width=8, mv_sampler=None, n_cand=20):
        _lib.require_gpu(tgn.device)
        if tgn.dp_world != 1:
            raise ValueError("graph capture covers the single-GPU step")
        dev = tgn.device
        self.tgn, self.opt, self.sampler, self.mvs = tgn, optimizer, sampler, mv_sampler
        self.B, self.K, self.n_neg, self.n_cand = int(batch), int(n_neighbors), int(n_neg), int(n_cand)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.src, self.dst = z(self.B, torch.int32), z(self.B, torch.int32)
        self.ts, self.eidx = z(self.B, torch.float64), z(self.B, torch.int32)
        self.port_idx, self.port_len = z((self.B, port_width), torch.int32), z(self.B, torch.int32)
        self.day = z(self.B, torch.int32)
        self.rng_pos = z(1, torch.int64)          # added to the Philox offsets on the device; +2^36 per step
        self.adam_t = z(1, torch.int32)           # steps taken inside the graph
        self.loss = z((), torch.float32)
        self.graph = None
        self.replays = 0
        self._keep = None          # strong references to every buffer whose device pointer the capture froze
        self._guard = None         # (neighbour finder, its version, workspace capacities) at capture time

    def _body(self):
        tgn = self.tgn
        tgn.request_zero_grad()            # the captured backward clears the gradient buffer itself (gradients stay attached)
        self.rng_pos.add_(1 << 36)
        self.adam_t.add_(1)
        if self.mvs is None:
            neg = self.sampler.sample(self.port_idx, self.port_len, self.n_neg, offset=0, offset_dev=self.rng_pos)
            emb, b = tgn.embed_device(self.src, self.dst, [neg.reshape(-1)], [self.n_neg], self.ts, self.eidx, self.K,
                                      offset_dev=self.rng_pos)
            loss = bpr_step(tgn, emb, b, self.n_neg, pos_block=1)
        else:
            cand_neg = self.sampler.sample(self.port_idx, self.port_len, self.n_cand, offset=0, offset_dev=self.rng_pos)
            cand = torch.cat([self.dst.unsqueeze(1), cand_neg], 1).contiguous()
            p_pos, p_neg = self.mvs.select_device(self.day, cand, self.port_idx, self.port_len)
            emb, b = tgn.embed_device(self.src, self.dst, [p_pos.reshape(-1), p_neg.reshape(-1)], [1, self.n_neg], self.ts,
                                      self.eidx, self.K, offset_dev=self.rng_pos)
            loss = bpr_step(tgn, emb, b, self.n_neg, pos_block=2)
        self.opt.step(step_dev=self.adam_t)
        self.loss.copy_(loss)

    def _load(self, src, dst, ts, eidx, port_idx, port_len, day):
        self.src.copy_(src); self.dst.copy_(dst); self.ts.copy_(ts); self.eidx.copy_(eidx)
        self.port_idx.copy_(port_idx); self.port_len.copy_(port_len)
        if day is not None:
            self.day.copy_(day)

    def capture(self, src, dst, ts, eidx, port_idx, port_len, day=None, warmup=3):
        """Runs ``warmup`` eager steps on a side stream (allocations, workspace, Adam moments), then captures one step.
        The steps taken here are real training steps on the given batch."""
        tgn = self.tgn
        tgn.train()
        self._load(src, dst, ts, eidx, port_idx, port_len, day)
        tgn._attach_grads(True)
        if tgn.use_memory and not tgn.memory._any_msg:
            tgn.memory._any_msg = bool(tgn.memory.has_msg.any())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._fold_steps()
        tgn._attach_grads(True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        # The capture froze raw device pointers.  Everything they name is kept alive here (the workspace lives in the TGN's
        # pool, which is purged when a later forward needs larger capacities; the CSR is replaced by NeighborFinder.append),
        # and a replay is refused once the model would no longer run the same step on the same buffers.
        nf = tgn.neighbor_finder
        mem = tgn.memory
        self._keep = (tgn._last_ws, tgn._adj_cache, tgn._pcache, tgn.flat_parameters, tgn.flat_grad, self.opt._m, self.opt._v,
                      tgn.node_raw_features, tgn.edge_raw_features,
                      None if mem is None else (mem.memory, mem.last_update, mem.msg_table, mem.msg_time, mem.has_msg))
        self._guard = (nf, getattr(nf, "_version", 0), tgn._ws_caps, tgn.flat_parameters.data_ptr(), tgn.flat_grad.data_ptr())
        return self

    def _check_guard(self):
        if self.graph is None:
            raise RuntimeError("no captured step: capture() first (finish() ends a captured step's life)")
        tgn = self.tgn
        nf = tgn.neighbor_finder
        now = (nf, getattr(nf, "_version", 0), tgn._ws_caps, tgn.flat_parameters.data_ptr(),
               None if tgn.flat_grad is None else tgn.flat_grad.data_ptr())
        if now[0] is not self._guard[0] or now[1:] != self._guard[1:]:
            raise RuntimeError("the captured step is stale (neighbour finder changed / appended to, workspace capacities grew, "
                               "or the parameter buffers moved): capture() again")

    def __call__(self, src, dst, ts, eidx, port_idx, port_len, day=None):
        """One training step on the given device-resident batch: six small copies and ONE graph launch."""
        self._check_guard()
        self._load(src, dst, ts, eidx, port_idx, port_len, day)
        self.graph.replay()
        self.replays += 1
        self.tgn.parameters_changed()      # (the replayed Adam kernel wrote them; the replayed forward rebuilt its own composites)
        return self.loss

    def eager(self, src, dst, ts, eidx, port_idx, port_len, day=None):
        """The same step, same device-side counters, queued kernel by kernel instead of replayed (profiling brackets, debugging)."""
        self._load(src, dst, ts, eidx, port_idx, port_len, day)
        self.tgn._attach_grads(True)
        self._body()
        return self.loss

    def _fold_steps(self):
        torch.cuda.synchronize()
        taken = int(self.adam_t.item())
        self.opt.sync_steps(taken)
        self.adam_t.zero_()
        return taken

    def finish(self):
        """Ends the captured step's life: folds the steps taken on the device-side counter back into the optimizer's host-side
        counters, drops the graph (its Adam launch has the OLD host counts frozen into its arguments: replaying it after
        this point would step the bias corrections backwards) and detaches the gradients, so that the next eager step
        starts from a cleared buffer like after ``optimizer.zero_grad(set_to_none=True)``.  Returns the steps folded."""
        taken = self._fold_steps()
        self.graph, self._keep, self._guard = None, None, None
        self.opt.zero_grad(set_to_none=True)
        return taken
