"""Data-parallel step on a single GPU: two ranks (processes) share cuda:0 and exchange gradients over gloo.
Checks SURVEY §8e's equality criterion: the 2-rank result equals the 1-rank result at the same global batch
(parameters after Adam, memory, last_update, pending-message tables), and both replicas stay identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


def _run(rank, world, port, out_dir, n_steps, B=48, buckets=False, det=False, fused=False, ordered=False):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import pfotgnrec_amd as P
    from pfotgnrec_amd.distributed import (init_from_env, allreduce_flat_grad, allreduce_flat_grad_buckets, allreduce_flat_grad_ordered,
                                           broadcast_parameters)
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.manual_seed(5)                         # same initial parameters on every rank
    cfg = SyntheticConfig("dp", 300, 25, 5000, 32, 2, 6, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, dev, n_layers=2, n_heads=2, dropout=0.0,
                use_memory=True, memory_dimension=32, message_function="identity")
    tgn.set_data_parallel(rank, world)
    tgn.dp_bucketed = buckets or ordered
    tgn.dp_ordered = ordered                                    # reduce + optimizer step per bucket, first-use bucket first (round 6)
    tgn.deterministic = det                                     # bitwise run-to-run reproducible backward (round 3)
    broadcast_parameters(tgn.flat_parameters, world)
    opt = P.FusedAdam(tgn, lr=1e-3)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    rs = np.random.RandomState(0)
    tgn.train()
    for step in range(n_steps):
        s = 2500 + step * B
        neg = t(rs.randint(301, 326, size=B * 3), np.int32)
        emb, b = tgn.embed_device(t(d.sources[s:s + B], np.int32), t(d.destinations[s:s + B], np.int32), [neg], [3],
                                  t(d.timestamps[s:s + B], np.float64), t(d.edge_idxs[s:s + B], np.int32), 6)
        assert b == (rank + 1) * B // world - rank * B // world         # balanced shards; empty when B < world
        assert emb.shape[0] == 5 * b
        if fused:
            # backward + all-reduce + Adam as ONE call: the end of the backward, the collective and the optimizer's kernel stay
            # on the library's side stream (functional.bpr_step); an empty shard takes the serial order inside the same call
            P.bpr_step(tgn, emb, b, 3, optimizer=opt,
                       collective=(lambda: allreduce_flat_grad_ordered(tgn, world)) if ordered else
                                  ((lambda: allreduce_flat_grad_buckets(tgn, world)) if buckets else (lambda: allreduce_flat_grad(tgn.flat_grad, world))))
            if step == 0:
                tgn.join()
                grad0 = tgn.flat_grad.cpu().numpy().copy()
                mem0 = tgn.memory.memory.cpu().numpy().copy()
            opt.zero_grad(set_to_none=True)
            continue
        loss = P.bpr_loss(emb, b, 3, grad_scale=tgn.dp_grad_scale)      # local mean * (b / B): shard sums = global mean
        loss.backward()
        if buckets:
            assert tgn._bucket_event_fresh == (b > 0)               # the backward recorded the "top layer final" event
            allreduce_flat_grad_buckets(tgn, world)
        else:
            allreduce_flat_grad(tgn.flat_grad, world)
        if step == 0:
            grad0 = tgn.flat_grad.cpu().numpy().copy()
            mem0 = tgn.memory.memory.cpu().numpy().copy()
        opt.step()
        opt.zero_grad(set_to_none=True)
    tgn.join()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "w%d_r%d_B%d%s.npz" % (world, rank, B, ("_buckets" if buckets else "") + ("_ordered" if ordered else "") + ("_fused" if fused else ""))), params=tgn.flat_parameters.cpu().numpy(), grad0=grad0, mem0=mem0,
             memory=tgn.memory.memory.cpu().numpy(), last_update=tgn.memory.last_update.cpu().numpy(),
             msg=tgn.memory.msg_table.cpu().numpy(), msg_t=tgn.memory.msg_time.cpu().numpy(), has=tgn.memory.has_msg.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, port, tmp_path, n_steps, B, buckets=False, det=False, fused=False, ordered=False):
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_run, args=(r, world, port, str(tmp_path), n_steps, B, buckets, det, fused, ordered)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0


@pytest.mark.parametrize("world,B", [(3, 7), (4, 2)])
def test_uneven_and_empty_shards(tmp_path, world, B):
    """B % world != 0 and B < world: every rank completes the step (an empty shard still runs the state update
    and joins the all-reduce with a zero gradient), replicas stay bit-identical, and the summed shard gradients equal the
    single-rank gradient of the global-batch mean loss."""
    port = 29650 + (os.getpid() % 100) + 7 * world
    _spawn(1, port, tmp_path, 2, B)
    _spawn(world, port + 1, tmp_path, 2, B)
    one = np.load(tmp_path / ("w1_r0_B%d.npz" % B))
    rk = [np.load(tmp_path / ("w%d_r%d_B%d.npz" % (world, r, B))) for r in range(world)]
    for r in range(1, world):
        for k in one.files:
            assert np.array_equal(rk[0][k], rk[r][k]), "replicas diverged: " + k
    rel = lambda a, b: np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-12)
    D = 32
    assert rel(rk[0]["grad0"][2 * D:], one["grad0"][2 * D:]) < 1e-4
    assert rel(rk[0]["grad0"][:2 * D], one["grad0"][:2 * D]) < 3e-3
    assert rel(rk[0]["mem0"], one["mem0"]) < 1e-5
    assert np.array_equal(rk[0]["last_update"], one["last_update"]) and np.array_equal(rk[0]["has"], one["has"])


def test_two_ranks_equal_one_rank(tmp_path):
    port = 29800 + (os.getpid() % 150)
    for world in (1, 2):
        _spawn(world, port + world, tmp_path, 3, 48)
    one = np.load(tmp_path / "w1_r0_B48.npz")
    r0, r1 = np.load(tmp_path / "w2_r0_B48.npz"), np.load(tmp_path / "w2_r1_B48.npz")
    for k in one.files:
        assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k          # bit-identical replicas
    rel = lambda a, b: np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-12)
    # step 1 (identical inputs): summed shard gradients == full-batch gradient, same persisted memory.
    # (Later steps are compared only between replicas: Adam turns noise-level gradient differences into +-lr steps,
    #  SURVEY §7 hard part 5.)  The first 2*D entries are the time-encoder parameters (cancellation-heavy sums).
    D = 32
    assert rel(r0["grad0"][2 * D:], one["grad0"][2 * D:]) < 1e-4
    assert rel(r0["grad0"][:2 * D], one["grad0"][:2 * D]) < 3e-3
    assert rel(r0["mem0"], one["mem0"]) < 1e-5
    assert np.array_equal(r0["last_update"], one["last_update"]) and np.array_equal(r0["has"], one["has"])
    assert np.array_equal(r0["msg_t"], one["msg_t"])
    assert rel(r0["params"], one["params"]) < 1e-2


def test_two_bucket_allreduce_equals_the_single_one(tmp_path):
    """SURVEY 8e / DESIGN 6: the top layer's gradient block reduced on a communication stream as soon as the backward's event
    says it is final, the rest after the backward - element-wise sums do not depend on how the buffer is cut, so gradients,
    parameters after Adam and the memory state are BIT-identical to the one-piece all-reduce (two ranks, gloo, one GPU)."""
    port = 29950 + (os.getpid() % 40)
    # (both runs with the deterministic backward: the default float-atomic scatter differs in its last bits from run to run,
    #  which would hide - or fake - a difference between the two collectives)
    _spawn(2, port, tmp_path, 3, 48, det=True)
    _spawn(2, port + 1, tmp_path, 3, 48, buckets=True, det=True)
    for r in (0, 1):
        one, two = np.load(tmp_path / ("w2_r%d_B48.npz" % r)), np.load(tmp_path / ("w2_r%d_B48_buckets.npz" % r))
        for k in one.files:
            assert np.array_equal(one[k], two[k]), (r, k)


def test_bucketed_allreduce_with_an_empty_shard(tmp_path):
    """world 2, global batch 1, two-bucket all-reduce.  Rank 1's shard is empty (no backward, no event): it must
    join BOTH pieces rank 0 issues, with a zero gradient - one whole-buffer collective against two pieces hangs or sums
    wrongly.  Result equals the single all-reduce bit for bit, replicas stay identical."""
    port = 29890 + (os.getpid() % 40)
    _spawn(2, port, tmp_path, 2, 1, det=True)
    _spawn(2, port + 1, tmp_path, 2, 1, buckets=True, det=True)
    for r in (0, 1):
        one, two = np.load(tmp_path / ("w2_r%d_B1.npz" % r)), np.load(tmp_path / ("w2_r%d_B1_buckets.npz" % r))
        for k in one.files:
            assert np.array_equal(one[k], two[k]), (r, k)
    a, b = np.load(tmp_path / "w2_r0_B1_buckets.npz"), np.load(tmp_path / "w2_r1_B1_buckets.npz")
    for k in a.files:
        assert np.array_equal(a[k], b[k]), "replicas diverged: " + k
    assert np.abs(a["grad0"]).max() > 0


@pytest.mark.parametrize("B,buckets", [(48, False), (1, False), (48, True), (1, True)])
def test_fused_step_with_the_collective_on_the_side_stream_equals_the_serial_order(tmp_path, B, buckets):
    """bpr_step(..., optimizer=, collective=) on a data-parallel rank: the end of the backward, the gradient all-reduce and the
    Adam kernel stay on the library's side stream while the caller's stream goes on to the next batch.  Same arithmetic in the
    same order per element as backward -> all-reduce -> step: gradients of step 1, parameters, memory and message tables are
    BIT-identical (two ranks, gloo, one GPU, deterministic backward); B = 1: rank 1's shard is empty - it joins the same single
    collective through the serial path inside the call."""
    port = 29730 + (os.getpid() % 40) + 3 * B + (11 if buckets else 0)
    _spawn(2, port, tmp_path, 3, B, det=True)
    _spawn(2, port + 1, tmp_path, 3, B, det=True, fused=True, buckets=buckets)        # (buckets: the two-piece exchange inside the fused call)
    for r in (0, 1):
        one = np.load(tmp_path / ("w2_r%d_B%d.npz" % (r, B)))
        two = np.load(tmp_path / ("w2_r%d_B%d%s_fused.npz" % (r, B, "_buckets" if buckets else "")))
        for k in one.files:
            assert np.array_equal(one[k], two[k]), (r, k)


def test_fused_step_on_three_ranks_with_uneven_shards(tmp_path):
    """world 3, global batch 7 (shards of 2, 2, 3): the fused backward + all-reduce + Adam call against the serial order, bit for
    bit on every rank (gloo, one GPU, deterministic backward)."""
    port = 29560 + (os.getpid() % 60)
    _spawn(3, port, tmp_path, 3, 7, det=True)
    _spawn(3, port + 1, tmp_path, 3, 7, det=True, fused=True)
    for r in range(3):
        one, two = np.load(tmp_path / ("w3_r%d_B7.npz" % r)), np.load(tmp_path / ("w3_r%d_B7_fused.npz" % r))
        for k in one.files:
            assert np.array_equal(one[k], two[k]), (r, k)


@pytest.mark.parametrize("B", [48, 1])
def test_fused_step_with_buckets_in_order_of_first_use_equals_the_serial_order(tmp_path, B):
    """Round 6: the fused call with the exchange AND the optimizer cut in two - the top layer's block reduced
    beside the backward, [time encoder | GRU | layer 1] reduced and stepped first behind its end (the next forward waits for that
    kernel alone), the top block stepped behind it.  Same arithmetic per element as backward -> all-reduce -> step: gradients
    of step 1, parameters after three steps, memory and message tables BIT-identical on both ranks (gloo, one GPU,
    deterministic backward); B = 1: rank 1's shard is empty - same two collectives through the serial route."""
    port = 29330 + (os.getpid() % 40) + 2 * B
    _spawn(2, port, tmp_path, 3, B, det=True)
    _spawn(2, port + 1, tmp_path, 3, B, det=True, fused=True, ordered=True)
    for r in (0, 1):
        one = np.load(tmp_path / ("w2_r%d_B%d.npz" % (r, B)))
        two = np.load(tmp_path / ("w2_r%d_B%d_ordered_fused.npz" % (r, B)))
        for k in one.files:
            assert np.array_equal(one[k], two[k]), (r, k)
