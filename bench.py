#!/usr/bin/env python3
"""Throughput of the TGN training step on MI355X (BASELINE.json metric: interactions/s, TGN fwd + BPR step).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic interactions, the loop body of
main.py:160-394: candidate-negative draw -> temporal neighbour sampling -> lazy memory update (GRU)
-> L-layer temporal graph attention -> BPR loss -> backward -> Adam -> memory persist + raw-message
store.  Inputs are resident in HBM before the timed region.

N = 1: BASELINE.json configs[1] (C2: 50k users x 500 items, 1M edges, 2-layer attention, 20 neighbours,
dim 172, batch 512, 3 negatives).
N > 1: one process per GPU over RCCL.  When the ranks are not already there (no WORLD_SIZE in the
environment, i.e. plain ``python bench.py --gpus N``) this process starts N child ranks itself - before
touching any GPU - and relays rank 0's line; under ``python -m torch.distributed.run`` it IS a rank.
Default workload for N > 1 is the SAME headline graph (C2) with 512 interactions per GPU - a global batch of 512 N cut
into N shards, one all-reduce of the flat gradient buffer per step (weak scaling: value(N) / (N value(1)) is the
efficiency, on one workload).  Every N > 1 line reports the collective's own time (``config.collective_ms_per_step``,
event-bracketed on the caller's stream) and carries, under ``secondary``, a 1 s run of the same workload with the other
all-reduce form (one piece / two buckets); at N = 8 (or with ``--secondary``) also BASELINE.json configs[3] (C4: 500k
users, 10M edges) at a FIXED global batch of 4096 interactions (strong scaling) with the same batch on ONE of
those GPUs.  ``--no-secondary`` drops all of that; ``--config`` / ``--scaling`` override (``--config C4`` makes the
strong-scaling case the main line).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     live HIP-event timing of the dominant kernel family over the timed region
  "cpu_baseline": the oracle (numpy restatement of the reference) timed on this host, bounded sample
The timed region is K steps; when K steps take less than --min-seconds (2 s) the K-step block is repeated
(each block bracketed by barrier + synchronize) and the line reports the mean over all blocks.
"""
import argparse
import json
import os
import sys
import time

# The step overlaps three internal side streams with the caller's stream.  The HIP runtime multiplexes ALL streams of a
# process onto GPU_MAX_HW_QUEUES hardware queues (default 4): as soon as a process group exists (RCCL and c10d bring their own
# streams) the library's side streams share a hardware queue with the caller's stream, and "beside" silently becomes "behind"
# (measured on the rank path at world 1: 1.65 ms per step against 1.40 with eight queues; the single-GPU path, with three
# streams in all, is unaffected).  Read by the runtime when it initialises - i.e. before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the sparsity figure is never used)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=["C1", "C2", "C3", "C4", "C5"],
                    help="default: C2 on one GPU, C4 (global batch 4096 fixed) on several")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: weak = --batch interactions per GPU, strong = the config's batch cut into N shards "
                         "(default: strong for C4, weak otherwise)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step timed block until this much time is covered")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1: no secondary figures at all (only the main line's workload runs)")
    ap.add_argument("--secondary", action="store_true",
                    help="N > 1: also run the other scaling case (C4 at a fixed global batch of 4 096 with its one-GPU reference, or the "
                         "weak-scaling C2 figure when the main line is C4) - a second 10 M-edge graph per rank.  ON by default at N = 8 "
                         "(BASELINE.json configs[3] names exactly that machine), 1 s budget; every N > 1 line also carries a 1 s run of "
                         "the main workload with each other all-reduce form (secondary.allreduce_single / _buckets / _fused / _fused_buckets)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a captured HIP graph (one launch per step instead of ~85).  auto (one GPU, per-GPU "
                         "batch <= 256): capture, time ten steps either way during set-up and keep the faster form - the graph "
                         "wins where the host's launch rate bounds the step (C1), the eager queue where the GPU does")
    ap.add_argument("--allreduce", default="fused", choices=["single", "buckets", "fused", "fused_buckets", "fused_ordered"],
                    help="N > 1: 'fused_ordered': the fused call with exchange and optimizer cut in two buckets in order of first use - "
                         "the top layer's block reduced beside the backward and stepped last, [time encoder | GRU | layer 1] reduced and "
                         "stepped first behind the backward's end: the next forward's caller's stream waits for that kernel alone (rehearsed "
                         "with stub collectives: the same step time as 'fused' - what is exposed behind the exchange is the side stream's "
                         "composite-weight chain, which needs every bucket: profiles/r6_experiments.txt 9); "
                         "'fused' (default): backward + all-reduce + Adam as one call whose end - the collective included - "
                         "stays on the library's side stream while the caller's stream starts the next batch "
                         "(bpr_step(..., optimizer=, collective=)): the exchange of step n runs beside step n+1's sampling; "
                         "'single': one all-reduce of the flat gradient on the caller's stream after the backward; 'buckets': two "
                         "pieces, the top layer's block reduced on a communication stream while the lower layers are still being "
                         "differentiated; 'fused_buckets': the fused call with the two-piece exchange (the top layer's block "
                         "is reduced beside the rest of the backward, only the lower block waits for its end).  Every N > 1 line "
                         "carries 1 s runs of the other forms (secondary.allreduce_*)")
    ap.add_argument("--emulate-ranks", default=None,
                    help="ONE process, no collective: time rank 0's share of the step (its shard's roots + the global state "
                         "update) for each listed world size, e.g. 1,2,4,8 - the compute-side ceiling of the scaling curve, "
                         "measurable on one GPU.  Prints one JSON line with a per-world-size table")
    ap.add_argument("--emulate-sleep", type=int, default=1, choices=[0, 1],
                    help="--emulate-ranks: 1 (default) the stub collective spins on the side stream for the predicted ring all-reduce "
                         "time of the flat gradient over xGMI (predicted_ring_allreduce_us); 0 it returns at once")
    ap.add_argument("--deterministic", action="store_true",
                    help="bitwise run-to-run reproducible backward (fixed-point level-0 gradient rows, slab-folded time partials)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="start the N ranks, rendezvous, all-reduce the rank ids and print a line; no GPU work (CPU test of the launcher)")
    ap.add_argument("--batch", type=int, default=0, help="interactions per GPU per step (default: the config's)")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=512, help="interactions per oracle step in the CPU baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=4, help="timed oracle steps (one more runs first, untimed)")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event bracketing of kernel launches")
    ap.add_argument("--no-drop-in", action="store_true",
                    help="one GPU: skip secondary.drop_in_surface (the literal main.py:160-394 loop on the drop-in classes, ~1 s per optimizer)")
    ap.add_argument("--overlap-tail", type=int, default=1, choices=[0, 1],
                    help="1 (default, one rank): backward + optimizer step as ONE fused call whose end - the last side-stream launches "
                         "of the backward and the Adam kernel - stays on the library's side stream while the caller's stream goes on to "
                         "the next batch's candidate draw and neighbour sampling (bpr_step(..., optimizer=)); 0: the caller's stream "
                         "waits for the whole backward, then runs Adam (the reference's loss.backward(); optimizer.step() order)")
    ap.add_argument("--prefetch", type=int, default=0, choices=[0, 1, 2, 3],
                    help="1: the next batch's negatives / frontier / compaction / packed rows are issued on a second stream right "
                         "after the current forward (TGN.prefetch); 0 (default): inside the step, as the reference's call order "
                         "has it.  Measured on C2: 1.52 ms with, 1.50 without - the ~70 us of small launches cost as much beside "
                         "the backward's first kernels as they do at the head of the step (DESIGN 4.2)")
    ap.add_argument("--marks", type=int, default=0,
                    help="after the timed region run this many extra steps with the library's milestones on (timing events on the "
                         "caller's stream at named points of the step: the critical path as the GPU ran it, no tracer) and print the "
                         "per-segment means to stderr")
    ap.add_argument("--prof-every", type=int, default=8,
                    help="bracket the kernel launches of every Nth timed step with HIP events (the brackets cost ~0.2 ms "
                         "per step at C2, so the roofline sample is taken on a subset of the timed steps)")
    return ap.parse_args()


# 256 CUs x 4 SIMD-32, one wave64 vector instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md:54,473) = 1 228.8 G
# wave-instructions/s.  Measured on this pool: tools/probes/mfma4x4_probe.hip, independent v_fma_f32 at four wavefronts per SIMD
# = 1.96 cycles per instruction per SIMD (profiles/r5_probe_mfma4x4.txt); a single wavefront issues one per ~7 cycles.
VALU_ISSUE_PEAK_GINST = 1024 * 2.4 / 2

# family (one PFO_PROF_* kind) -> the kernel(s) of that kind as named in the rocprofv3 summaries of profiles/ (template
# arguments differ between configurations: matched by the name in front of them)
FAMILY_KERNEL = {"gemm_bx": "gemm_bx_areg_kernel", "gemm_tn_bx": "gemm_tn_group_bx_kernel", "gemm_tn_bx8": "gemm_tn_group_bx_kernel", "gru_fused": "gru_fused_kernel",
                 "gemm_bx_skinny": "gemm_bx_skinny_kernel", "gemm_nt": "gemm_f32_kernel", "gemm_nn": "gemm_f32_kernel",
                 "gemm_devm": "gemm_f32_kernel", "gemm_tn": "gemm_tn_group_kernel", "gemm_multi": "gemm_multi_kernel",
                 "attn_fwd": "attn_fwd_kernel", "attn_bwd": "attn_bwd_kernel", "attn_bwd_runs": "attn_bwd_runs_kernel",
                 "sampler": "tnbr_sample_kernel", "segsum": "segsum_chunk_kernel", "tn_reduce": "tn_group_reduce_kernel",
                 "gru_gates_bwd": "gru_gates_bwd_vec_kernel"}
WORKLOAD_KEY = [None]      # "C2@512": set by main() - the committed counter passes are keyed by (configuration, batch per GPU)


def _kernel_base(name):
    n = name.split("(")[0].strip()
    n = n[5:] if n.startswith("void ") else n
    return n.split("<")[0].split("::")[-1]


def _family_has(family, kernel_name):
    """Is this kernel (a name of the rocprofv3 tables) one of the family's?  The per-instance attention backward has three
    entry points (attn_bwd_kernel, _none, _direct, _det) behind one PFO_PROF kind; the run-merged kernel is a family of its own."""
    base = _kernel_base(kernel_name)
    if family == "attn_bwd":     # per-instance backward: register form (four entry points) and LDS key-ring form (two)
        return base.startswith("attn_bwd_kernel") or base.startswith("attn_bwd_ring_kernel")
    if family == "gemm_bx":      # three forms behind PFO_PROF_GEMM_BX: four / eight wavefronts per workgroup, A-stationary
        return base in ("gemm_bx_areg_kernel", "gemm_bx_areg8_kernel", "gemm_bx_astat_kernel")
    if family in ("gemm_tn_bx", "gemm_tn_bx8"):   # two template instances = two kernels of the trace: <FMT, 8> is the 256-row form
        eight = kernel_name.split("(")[0].replace(" ", "").endswith(",8>")
        return base == "gemm_tn_group_bx_kernel" and eight == (family == "gemm_tn_bx8")
    if family == "attn_fwd":     # the per-instance register form and the LDS key-ring / pipeline forms (PFO_PROF_ATTN_FWD)
        return base in ("attn_fwd_kernel", "attn_fwd_ring_kernel", "attn_fwd_pipe_kernel")
    return base == FAMILY_KERNEL.get(family, family)


PROFILE_ROUND = "r6"       # the committed counter passes / kernel traces this line is priced from: profiles/<round>_*; a workload
                           # without files of THIS round falls back to the newest older round and says so (``profiles_round``)


def _profile_file(pattern, key):
    """profiles/<PROFILE_ROUND>_<pattern % key> if it exists, else the newest older round's file of the same name."""
    import glob
    if not key:
        return None
    f = os.path.join(REPO, "profiles", ("%s_" + pattern) % (PROFILE_ROUND, key))
    if os.path.exists(f):
        return f
    files = sorted(glob.glob(os.path.join(REPO, "profiles", ("r*_" + pattern) % key)))
    return files[-1] if files else None


def _summary_for(key):
    """The committed profile summary of this workload: profiles/<round>_summary_<key>.json (tools/profile_round.sh +
    tools/summarize_profile.py), the CURRENT round's when it exists."""
    return _profile_file("summary_%s.json", key)


def _trace_for(key):
    return _profile_file("kernel_stats_bench_%s.csv", key)


def pmc_counters(family):
    """Counters per launch of the family's kernel(s) from the newest committed PMC passes of THIS workload (separate --pmc
    runs; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950; ``sq`` = the SQ instruction counts of the same launches).
    Several template instances of one family are averaged by their launch counts.
    (bytes or None, {counter: value} or {}, file name, kernel names)."""
    f = _summary_for(WORKLOAD_KEY[0]) if WORKLOAD_KEY[0] else None
    if not f:
        return None, {}, None, []
    try:
        d = json.load(open(f))
        rows = [(k, v) for k, v in d["pmc"]["kernels"].items() if _family_has(family, k)]
        if not rows:
            return None, {}, os.path.basename(f), []
        w = [max(1, d.get("kernels", {}).get(k, {}).get("launches", 1)) for k, _ in rows]
        tot = float(sum(w))
        hbm = sum(v["hbm_bytes_per_launch_corrected"] * wi for (_, v), wi in zip(rows, w)) / tot
        sq = {}
        for c in set().union(*[set(v.get("sq", {})) for _, v in rows]):
            sq[c] = sum(v.get("sq", {}).get(c, 0.0) * wi for (_, v), wi in zip(rows, w)) / tot
        return int(hbm), sq, os.path.basename(f), [k for k, _ in rows]
    except Exception:
        return None, {}, None, []


def pmc_traffic(family):
    return pmc_counters(family)[0]


def trace_dominant(prof, n_prof_steps):
    """The kernel with the largest summed device time in the committed kernel trace of this workload (profiles/
    r*_kernel_stats_bench_<key>.csv: every launch as the step runs it, co-scheduled launches included), with the family's
    live algorithmic work over that resident time - the rate the kernel sustains in production, beside the bracketed-alone
    figures of ``roofline``."""
    import csv
    import glob
    key = WORKLOAD_KEY[0]
    tf = _trace_for(key)
    files = [tf] if tf else []
    if not files:
        return None
    try:
        steps = 25
        sf = _summary_for(key)
        if sf:
            steps = int(json.load(open(sf)).get("steps_in_trace", 25))
        rows = [r for r in csv.DictReader(open(files[-1])) if not r["Name"].startswith("__amd_rocclr")]
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        fam = next((k for k in FAMILY_KERNEL if _family_has(k, r["Name"]) and k in prof and prof[k]["count"] > 0), None)
        out = {"kernel_name": r["Name"].split("(")[0], "launches_per_step": round(int(r["Calls"]) / steps, 2),
               "avg_launch_us_resident": round(float(r["AverageNs"]) / 1e3, 2),
               "ms_per_step_resident": round(float(r["TotalDurationNs"]) / steps / 1e6, 4),
               "share_of_kernel_time": round(float(r["TotalDurationNs"]) / tot, 4), "from": os.path.basename(files[-1])}
        if fam:
            # (a family is ONE kernel of the trace - the two tile forms of the grouped weight gradients are families of their
            #  own, PFO_PROF_GEMM_TN_BX / _BX8 - so this kernel's live work over this kernel's resident time is one division)
            v = prof[fam]
            unit = 1e12 if (fam.startswith("gemm") or fam == "gru_fused") else 1e9
            work_per_step = v["work"] / max(1, n_prof_steps)
            out["family"] = fam
            out["work_per_step"] = round(work_per_step / unit, 4)
            out["work_unit"] = "TFLOP" if unit == 1e12 else "GB"
            out["co_scheduled_rate"] = round(work_per_step / (out["ms_per_step_resident"] * 1e-3) / unit, 2)
            out["co_scheduled_rate_unit"] = "TFLOP/s" if unit == 1e12 else "GB/s"
            out["alone_rate"] = round(v["work"] / max(1e-12, v["ms"] * 1e-3) / unit, 2)
            out["alone_avg_launch_us"] = round(v["ms"] / max(1, v["count"]) * 1e3, 2)
            if fam in ("gemm_tn_bx", "gemm_tn_bx8"):
                # both tile forms together (the four grouped weight-gradient launches of a step), same two views
                both = [prof[k] for k in ("gemm_tn_bx", "gemm_tn_bx8") if k in prof and prof[k]["count"] > 0]
                res = sum(float(q["TotalDurationNs"]) for q in rows if _kernel_base(q["Name"]) == "gemm_tn_group_bx_kernel") / steps / 1e6
                w = sum(b["work"] for b in both) / max(1, n_prof_steps)
                out["tile_forms_together"] = {"ms_per_step_resident": round(res, 4), "co_scheduled_rate": round(w / max(1e-12, res * 1e-3) / unit, 2),
                                              "alone_rate": round(sum(b["work"] for b in both) / max(1e-12, sum(b["ms"] for b in both) * 1e-3) / unit, 2)}
        return out
    except Exception as e:
        return {"error": repr(e)[:200]}


def steady_state_init(tgn, rs):
    """Every node holds a pending message and a non-zero memory, the state the reference reaches a few
    hundred batches into an epoch (#pending == #seen nodes -> n)."""
    import torch
    mem = tgn.memory
    with torch.no_grad():
        g = torch.Generator(device="cpu").manual_seed(1234)
        mem.memory.copy_((torch.randn(mem.memory.shape, generator=g) * 0.1).to(mem.memory.device))
        mem.msg_table.copy_((torch.randn(mem.msg_table.shape, generator=g) * 0.1).to(mem.memory.device))
        mem.msg_time.zero_()
        mem.last_update.zero_()
        mem.has_msg.fill_(1)
        mem.has_msg[0] = 0


def cpu_baseline(cfg, graph, batch, steps):
    """The oracle's full training step (same op list as the GPU step) on this host; numpy + BLAS threads."""
    from oracle import tgn_oracle as T
    from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency
    d = graph.data
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps), uniform=cfg.uniform)
    P = T.init_params(cfg.dim, cfg.edge_dim, cfg.n_layers, seed=0, use_memory=cfg.use_memory)
    ref = T.OracleTGN(onf, graph.node_features, graph.edge_features, P, cfg.n_layers, cfg.n_heads, cfg.use_memory)
    rs = np.random.RandomState(7)
    if cfg.use_memory:
        M = 3 * cfg.dim + cfg.edge_dim
        msgs = (rs.randn(graph.n_nodes, M) * 0.1).astype(np.float32)
        for v in range(1, graph.n_nodes):
            ref.messages[v] = [(msgs[v], np.float32(0))]
        ref.memory = (rs.randn(graph.n_nodes, cfg.dim) * 0.1).astype(np.float32)
    m = {k: np.zeros_like(v) for k, v in ref.P.items()}
    v2 = {k: np.zeros_like(v) for k, v in ref.P.items()}
    s = cfg.n_edges // 2
    times, phases = [], []
    for it in range(steps + 1):
        sl = slice(s, s + batch)
        sb, db, tb, eb = d.sources[sl], d.destinations[sl], d.timestamps[sl], d.edge_idxs[sl]
        t0 = time.perf_counter()
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=batch * 3)
        se, de, ne = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, cfg.n_neighbors)
        t1 = time.perf_counter()
        loss, cache = T.bpr_loss(se, de.reshape(batch, 1, -1), ne.reshape(batch, 3, -1))
        ds, dp, dn = T.bpr_loss_backward(cache)
        t2 = time.perf_counter()
        grads = ref.backward(np.concatenate([ds, dp.reshape(batch, -1), dn.reshape(3 * batch, -1)]))
        t3 = time.perf_counter()
        for k in ref.P:                                   # Adam, main.py:123
            g = grads[k]
            m[k] = 0.9 * m[k] + 0.1 * g
            v2[k] = 0.999 * v2[k] + 0.001 * g * g
            ref.P[k] = ref.P[k] - 1e-4 * (m[k] / (1 - 0.9 ** (it + 1))) / (np.sqrt(v2[k] / (1 - 0.999 ** (it + 1))) + 1e-8)
        t4 = time.perf_counter()
        if it > 0:                                        # first step pays page-in / BLAS warm-up
            times.append(t4 - t0)
            phases.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        s += batch
    ph = np.median(np.array(phases), axis=0)
    cpu_baseline.phases = {"sampling+memory+forward_s": round(float(ph[0]), 3), "loss_s": round(float(ph[1]), 4),
                           "backward_s": round(float(ph[2]), 3), "adam_s": round(float(ph[3]), 4)}
    return batch / float(np.median(times)), float(np.sum(times))


def launch_ranks(n, poll_s=0.2, grace_s=10.0):
    """``python bench.py --gpus N`` without a launcher: start N ranks as CHILD processes (one per GPU, RCCL rendezvous
    on 127.0.0.1) before this process has touched a GPU, relay rank 0's stdout, exit with the worst child status.
    A watchdog polls every child: on the first non-zero exit the rest are terminated (killed after a grace period) - a rank
    that dies before or inside a collective would otherwise leave the others blocked in RCCL until its timeout."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out_file = tempfile.TemporaryFile()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PFO_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out_file if r == 0 else subprocess.DEVNULL))
    rc = 0
    while True:
        codes = [q.poll() for q in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            t_end = time.time() + grace_s
            for q in procs:
                try:
                    q.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    q.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    for q in procs:
        q.wait()
    out_file.seek(0)
    sys.stdout.write(out_file.read().decode())
    sys.stdout.flush()
    if rc:
        sys.stderr.write("bench.py: a rank exited with status %d; the remaining ranks were stopped\n" % rc)
    return rc


def launcher_selftest(args):
    if os.environ.get("PFO_SELFTEST_FAIL_RANK") == os.environ.get("RANK"):
        sys.exit(7)                                          # watchdog test: this rank dies before the rendezvous
    import torch
    import torch.distributed as dist
    from pfotgnrec_amd.distributed import init_from_env
    rank, world, _ = init_from_env(backend=os.environ.get("PFO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(rank)], device=dev)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "world_size": dist.get_world_size(),
                          "backend": dist.get_backend(), "rank_sum": float(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


class Workload:
    """One configuration resident on this rank's GPU: graph, model, optimizer, device-side batch sources."""

    def __init__(self, args, cfg_name, dev, rank, world, scaling, emulate=False):
        import torch
        import pfotgnrec_amd as P
        from pfotgnrec_amd.distributed import broadcast_parameters
        from pfotgnrec_amd.synthetic import CONFIGS, make_graph
        t_build0 = time.perf_counter()
        from pfotgnrec_amd.rand_edge_sampler import item_availability, DeviceNegativeSampler
        self.P, self.torch, self.args, self.dev, self.rank, self.world = P, torch, args, dev, rank, world
        cfg = self.cfg = CONFIGS[cfg_name]
        self.scaling = scaling
        self.per_gpu = args.batch or cfg.batch
        # weak: every GPU brings per_gpu interactions (global batch grows); strong: the config's batch is cut into shards
        self.B = self.per_gpu * world if scaling == "weak" else self.per_gpu
        self.ours = cfg.name == "C3"                          # C3 = C2 + MV-efficient sampler (main.py `ours` path)
        graph = self.graph = make_graph(cfg, with_prices=self.ours)
        d = graph.data
        # the CSR is built on the device (two stable sorts) - the host build takes ~10 s per 10 M edges and every rank needs one
        nf = P.NeighborFinder.from_arrays(d.sources, d.destinations, d.edge_idxs, d.timestamps, uniform=cfg.uniform, device=dev)
        tgn = self.tgn = P.TGN(nf, graph.node_features, graph.edge_features, dev, n_layers=cfg.n_layers, n_heads=cfg.n_heads,
                               dropout=args.dropout, use_memory=cfg.use_memory, memory_dimension=cfg.dim,
                               message_function="identity", n_neighbors=cfg.n_neighbors)
        tgn.set_data_parallel(rank, world)
        tgn.deterministic = bool(args.deterministic)
        tgn.dp_bucketed = (world > 1 or os.environ.get("PFO_DIST_FORCE") == "1" or emulate) and args.allreduce in ("buckets", "fused_buckets", "fused_ordered")
        tgn.dp_ordered = tgn.dp_bucketed and args.allreduce == "fused_ordered"
        self.emulate = emulate                                # --emulate-ranks: a rank's compute without the collective
        import torch.distributed as _dist
        self.dist_on = (not emulate) and _dist.is_available() and _dist.is_initialized()   # world 1 with PFO_DIST_FORCE=1: the rank path on one GPU
        self.force_dist = os.environ.get("PFO_DIST_FORCE") == "1"
        self.allreduce_mode = args.allreduce
        self.coll_events, self.time_collective = [], False
        if not emulate:
            broadcast_parameters(tgn.flat_parameters, world)
        if cfg.use_memory:
            steady_state_init(tgn, None)
        # (the loop below zeroes the gradients right after every step, as main.py:388-390 does: the optimizer's kernel does it)
        self.opt = P.FusedAdam(tgn, lr=args.lr, zero_grads_in_step=os.environ.get("PFO_BENCH_ZERO_IN_STEP", "1") == "1")
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        self.src_all, self.dst_all = t(d.sources, np.int32), t(d.destinations, np.int32)
        self.ts_all, self.eidx_all = t(d.timestamps, np.float64), t(d.edge_idxs, np.int32)
        self.port_idx_all, self.port_len_all = t(graph.portfolio_idx, np.int32), t(graph.portfolio_len, np.int32)
        self.sampler = DeviceNegativeSampler(item_availability(d.destinations, graph.upper_u, cfg.n_items), graph.upper_u, dev, seed=1)
        self.n_neg = 3
        self._next = None
        self.overlap_tail = bool(getattr(args, "overlap_tail", 1))
        self.prefetch = int(getattr(args, "prefetch", 0))
        tgn.record_mid_event = self.prefetch in (2, 3)
        tgn.mid_event_late = self.prefetch == 3
        self.mvs = None
        if self.ours:
            self.mvs = P.MVSampler(graph.prices, graph.upper_u, dev, gamma=2.0, lambda_mv=0.5, p_pos_num=1, p_neg_num=3)
            self.day_all = t(graph.day_of(d.timestamps), np.int32)
        self.start = cfg.n_edges // 2                         # neighbourhoods are populated
        self.span = cfg.n_edges - self.start - self.B         # batches wrap inside the second half of the edge list
        assert self.span > 0, "batch larger than the timed half of the graph"
        tgn.train()
        torch.cuda.synchronize()
        self.build_s = time.perf_counter() - t_build0          # synthetic graph (host) + device CSR + model + state upload
        # a HIP graph of the whole step for the launch-bound regime (pfotgnrec_amd/graph.py)
        self.gstep, self.graph_note = None, "off"
        want = args.graph == "on" or (args.graph == "auto" and world == 1 and self.B <= 256)
        if want and world == 1:
            try:
                sl = slice(self.start, self.start + self.B)
                self.gstep = P.GraphedTrainStep(tgn, self.opt, self.sampler, self.B, cfg.n_neighbors, n_neg=self.n_neg,
                                                port_width=self.port_idx_all.shape[1], mv_sampler=self.mvs)
                self.gstep.capture(*self._batch(sl))
                self.graph_note = "on"
                if args.graph == "auto":
                    ms = {}
                    for mode in ("graph", "eager"):
                        fn = self.gstep if mode == "graph" else self.gstep.eager
                        for k in range(13):                       # three untimed steps (allocator, caches), then ten timed
                            if k == 3:
                                torch.cuda.synchronize()
                                t0 = time.perf_counter()
                            lo = self.start + (k * self.B) % self.span
                            fn(*self._batch(slice(lo, lo + self.B)))
                        torch.cuda.synchronize()
                        ms[mode] = 1e2 * (time.perf_counter() - t0)
                    self.graph_note = "auto: graph %.3f ms/step, eager %.3f -> %s" % (ms["graph"], ms["eager"],
                                                                                       "graph" if ms["graph"] < ms["eager"] else "eager")
                    if ms["graph"] >= ms["eager"]:
                        self.gstep.finish()
                        self.gstep = None
            except Exception as e:          # capture is an optimisation: the eager step is always available
                self.gstep, self.graph_note = None, "failed (%s): eager" % (str(e)[:80],)

    def _batch(self, sl):
        return (self.src_all[sl], self.dst_all[sl], self.ts_all[sl], self.eidx_all[sl], self.port_idx_all[sl], self.port_len_all[sl],
                self.day_all[sl] if self.mvs is not None else None)

    def set_world(self, rank, world):
        """Re-targets the resident model at a different data-parallel layout (the single-GPU reference of an N-rank run)."""
        self.rank, self.world = rank, world
        self.tgn.set_data_parallel(rank, world)

    def step(self, i):
        from pfotgnrec_amd.distributed import allreduce_flat_grad, allreduce_flat_grad_buckets, allreduce_flat_grad_ordered
        P, torch, tgn, cfg, B, n_neg = self.P, self.torch, self.tgn, self.cfg, self.B, self.n_neg
        lo = self.start + (i * B) % self.span
        sl = slice(lo, lo + B)
        from pfotgnrec_amd import _lib as _lm
        _lm.mark("step.begin")
        if self.gstep is not None:
            from pfotgnrec_amd import _lib as _l
            return (self.gstep.eager if _l.prof_is_on() else self.gstep)(*self._batch(sl))   # bracketed steps run kernel by kernel
        if self.mvs is None:
            if self._next is not None and self._next[0] == i:
                neg = self._next[1]                               # drawn (and its neighbourhood prepared) beside the previous backward
            else:
                neg = self.sampler.sample(self.port_idx_all[sl], self.port_len_all[sl], n_neg, offset=i).reshape(-1)   # utils.py:86-114
            self._next = None
            emb, b = tgn.embed_device(self.src_all[sl], self.dst_all[sl], [neg], [n_neg], self.ts_all[sl],
                                      self.eidx_all[sl], cfg.n_neighbors)                                     # tgn.py:219-327
            pos_block = 1                                                                                      # main.py:364-381
            if self.prefetch == 1:
                # the next batch's negatives, frontier, compaction and packed memory rows on the model's second stream: the
                # reference's loop does this between batches on the host (main.py:190-207 + the neighbour finder); here it
                # runs beside this batch's backward.  Every step still does exactly one step's worth of it.
                lo2 = self.start + ((i + 1) * B) % self.span
                s2 = slice(lo2, lo2 + B)
                with tgn.prefetching():
                    neg2 = self.sampler.sample(self.port_idx_all[s2], self.port_len_all[s2], n_neg, offset=i + 1).reshape(-1)
                    if tgn.prefetch(self.src_all[s2], self.dst_all[s2], [neg2], [n_neg], self.ts_all[s2], self.eidx_all[s2],
                                    cfg.n_neighbors):
                        self._next = (i + 1, neg2)
        else:
            cand_neg = self.sampler.sample(self.port_idx_all[sl], self.port_len_all[sl], 20, offset=i)          # main.py:194-195
            cand = torch.cat([self.dst_all[sl].unsqueeze(1), cand_neg], 1).contiguous()                        # main.py:207
            p_pos, p_neg = self.mvs.select_device(self.day_all[sl], cand, self.port_idx_all[sl], self.port_len_all[sl])   # main.py:209-304
            emb, b = tgn.embed_device(self.src_all[sl], self.dst_all[sl], [p_pos.reshape(-1), p_neg.reshape(-1)], [1, 3],
                                      self.ts_all[sl], self.eidx_all[sl], cfg.n_neighbors)                    # tgn.py:102-217
            pos_block = 2                                                                                      # main.py:321-337
        # loss + loss.backward() (main.py:337,388) as two native calls: the loss kernel hands its gradient rows straight to
        # the TGN backward (P.bpr_loss(...).backward() is the autograd spelling of the same thing, tests/test_gpu_round2.py)
        _lm.mark("step.embedded")
        ranks = self.dist_on and (self.world > 1 or self.force_dist)
        fused_coll = ranks and self.allreduce_mode in ("fused", "fused_buckets", "fused_ordered") and self.overlap_tail and not self.prefetch
        fused_opt = self.overlap_tail and (not ranks or fused_coll) and not self.prefetch
        coll = None
        if self.emulate and self.world > 1 and fused_opt and self.allreduce_mode in ("fused", "fused_buckets", "fused_ordered"):
            # an emulated rank keeps the ranks' step schedule: the collective is a stub that holds the side stream for the
            # predicted ring time (--emulate-sleep 0: returns at once).  The bucketed forms hold the communication stream for the
            # top block's time from the "top layer final" event on, and the side stream for the rest's.
            if getattr(self, "sleep_coll", None) is None or self.sleep_coll_world != self.world or self.sleep_coll_mode != self.allreduce_mode:
                sleep = getattr(self.args, "emulate_sleep", 1)
                total, split = tgn.flat_parameters.numel(), tgn.grad_split
                if self.allreduce_mode == "fused" or not (0 < split < total):
                    self.sleep_coll = SleepCollective(predicted_ring_allreduce_us(total * 4, self.world) if sleep else 0.0)
                else:
                    self.sleep_coll = SleepCollectiveBuckets(tgn, predicted_ring_allreduce_us((total - split) * 4, self.world) if sleep else 0.0,
                                                             predicted_ring_allreduce_us(split * 4, self.world) if sleep else 0.0,
                                                             ordered=self.allreduce_mode == "fused_ordered")
                self.sleep_coll_world, self.sleep_coll_mode = self.world, self.allreduce_mode
            coll = self.sleep_coll
        if fused_coll:
            def coll():                                           # runs on the library's side stream (bpr_step): bracketed THERE
                ev = None
                if self.time_collective:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                if self.allreduce_mode == "fused_ordered":
                    allreduce_flat_grad_ordered(tgn, self.world, force=True)    # (no join of the communication stream: FusedAdam steps per bucket)
                elif self.allreduce_mode == "fused_buckets":
                    allreduce_flat_grad_buckets(tgn, self.world, force=True)    # (the caller's stream of that call = the side stream)
                else:
                    allreduce_flat_grad(tgn.flat_grad, self.world, force=True)
                if ev is not None:
                    ev[1].record()
                    self.coll_events.append(ev)
        loss = P.bpr_step(tgn, emb, b, n_neg, pos_block=pos_block, optimizer=self.opt if fused_opt else None, collective=coll)
        _lm.mark("step.backward_done")
        if self.prefetch in (2, 3) and self.mvs is None:
            # the next batch's negatives, frontier, compaction and packed rows: queued behind the backward's "attention backward
            # is next" event on the model's second stream - beside the longest kernel of the step.  The reference's loop does
            # this work on the host between batches (main.py:190-207 + the neighbour finder); every step still does exactly one
            # step's worth of it.
            lo2 = self.start + ((i + 1) * B) % self.span
            s2 = slice(lo2, lo2 + B)
            with tgn.prefetching(beside_attention_backward=True):
                neg2 = self.sampler.sample(self.port_idx_all[s2], self.port_len_all[s2], n_neg, offset=i + 1).reshape(-1)
                if tgn.prefetch(self.src_all[s2], self.dst_all[s2], [neg2], [n_neg], self.ts_all[s2], self.eidx_all[s2],
                                cfg.n_neighbors):
                    self._next = (i + 1, neg2)
        if ranks and not fused_coll:                                # (set_world(0, 1): rank 0 alone, no collective)
            ev = None
            if self.time_collective:                          # sampled steps: the collective bracketed by events on the caller's stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self.allreduce_mode == "buckets":
                allreduce_flat_grad_buckets(tgn, self.world, force=True)
            else:
                allreduce_flat_grad(tgn.flat_grad, self.world, force=True)
            if ev is not None:
                ev[1].record()
                self.coll_events.append(ev)
        if not fused_opt:
            self.opt.step()                                                                                    # main.py:389
        _lm.mark("step.adam")
        self.opt.zero_grad(set_to_none=True)
        return loss

    def collective_ms(self):
        """Mean device time of the bracketed all-reduce calls (ms), number of samples: from the first event - queued behind
        the backward's last kernel on the caller's stream - to the point where the reduced buffer is usable there.  It
        includes waiting for the slowest rank's backward; with the two-bucket form only the exposed remainder."""
        if not self.coll_events:
            return None, 0
        self.torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in self.coll_events]
        return float(np.mean(t)), len(t)

    def timed(self, steps, warmup, min_seconds, prof_every=0, first_step=0, collective=True, coll_every=4):
        """W warm-up steps, then blocks of exactly K steps, each bracketed by barrier + synchronize on both sides, until
        min_seconds are covered; per block the MAX over ranks is taken.  Returns (seconds, timed steps, blocks, loss, sampled)."""
        import torch.distributed as dist
        from pfotgnrec_amd import _lib
        torch = self.torch
        multi = collective and self.dist_on and (self.world > 1 or self.force_dist)
        # what exists now (graph arrays, the model, a process group's set-up garbage) is collected once and taken out of the
        # cycle collector's later passes: a full collection inside a timed block holds the host for tens of ms (experiment 22)
        import gc
        gc.collect()
        gc.freeze()
        i = first_step
        for _ in range(warmup):
            self.step(i)
            i += 1
        total, n_steps, blocks, n_prof = 0.0, 0, 0, 0
        self.block_ms, self.host_ms, self.coll_events = [], [], []
        loss = None
        while True:
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                sampled = prof_every > 0 and (k % prof_every == 0) and blocks == 0
                if sampled:
                    _lib.prof_enable(True)
                    n_prof += 1
                # the collective's own time: bracketed on every 4th step of every block (two event records, no kernel brackets)
                self.time_collective = self.dist_on and coll_every > 0 and (k % coll_every == 1 % coll_every) and not sampled
                loss = self.step(i)
                i += 1
                if sampled:
                    _lib.prof_enable(False)
            self.host_ms.append(1e3 * (time.perf_counter() - t0) / steps)      # host time to ENQUEUE a step (no wait)
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if multi:
                tmax = torch.tensor([el], device=self.dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            total += el
            self.block_ms.append(1e3 * el / steps)
            n_steps += steps
            blocks += 1
            if total >= min_seconds or blocks >= 10000:
                break
        return total, n_steps, blocks, float(loss.detach().reshape(-1)[0]), n_prof

    def describe(self):
        cfg = self.cfg
        return ("%s: %d users x %d items, %d edges, TGN L%d K%d D%d H%d %s, %d negatives, batch %s"
                % (cfg.name, cfg.n_users, cfg.n_items, cfg.n_edges, cfg.n_layers, cfg.n_neighbors, cfg.dim, cfg.n_heads,
                   "memory+GRU" if cfg.use_memory else "no memory, uniform sampling", self.n_neg,
                   "%d/GPU" % self.per_gpu if self.scaling == "weak" else "%d global (fixed), %d shards" % (self.B, self.world)))


DROP_IN_ENTRIES = ("torch.optim.Adam", "FusedAdam", "FusedAdam + bpr_loss_blocks", "overlap_backward(tgn, torch.optim.Adam)", "FusedAdam(overlap_backward=True)",
                   "FusedAdam(overlap_backward=True) + bpr_loss_blocks", "ours branch, FusedAdam(overlap_backward=True)")


def drop_in_surface(wl, budget_s=0.6):
    """Throughput through the LITERAL drop-in surface - what a maintainer gets after the four-import swap of INTEGRATION.md, with
    the reference's loop otherwise untouched (main.py:160-394, baseline branch): numpy batch slices, ``RandEdgeSampler(...)``
    constructed per batch (main.py:347-348), ``compute_temporal_embeddings`` on numpy arrays, the BPR loss as the reference's
    torch expression (main.py:364-381), ``loss.backward()``, ``optimizer.step()``, ``loss.item()`` (main.py:388-390: a device
    sync per batch), ``detach_memory()`` (main.py:394) - once with torch.optim.Adam (main.py:123), once with the package's
    FusedAdam; then with the optional swaps of INTEGRATION.md (DROP_IN_ENTRIES).  Wall time per step with the host part itemised
    (the host cannot run ahead: the loop synchronises twice per batch)."""
    import torch
    P, cfg, g, tgn = wl.P, wl.cfg, wl.graph, wl.tgn
    d = g.data
    B, K, q = wl.per_gpu, cfg.n_neighbors, 3
    codes = g.codes
    n_batches = 256
    span = (cfg.n_edges - wl.start - B) // B
    # the reference slices ONE prebuilt object array of stock-code lists per batch (train_data.portfolios[s_idx:e_idx],
    # main.py:186; utils/data.py:18-72): built here for the range of edges the loop visits, sliced below the same way
    n_rows = min(n_batches, span) * B
    portfolios_all = np.empty(n_rows, dtype=object)
    pl_all, pi_all = g.portfolio_len[wl.start:wl.start + n_rows], g.portfolio_idx[wl.start:wl.start + n_rows]
    for r in range(n_rows):
        portfolios_all[r] = [codes[j] for j in pi_all[r, :pl_all[r]]] if pl_all[r] > 0 else [""]
    n_vis = n_rows // B
    out = {"workload": wl.describe(), "loop": "main.py:160-394 (baseline branch) on pfotgnrec_amd's drop-in classes, numpy batches"}
    was_training = tgn.training
    mvs = None
    for name in DROP_IN_ENTRIES:
        # (third entry: the loop with ONE more line swapped - main.py:364-381's ten-kernel torch expression replaced by
        #  ``loss = pfotgnrec_amd.bpr_loss_blocks(source_embedding, destination_embedding, negative_embedding)``, INTEGRATION.md;
        #  fourth: the ``ours`` branch, main.py:190-337 - the inline mean-variance block as ``MVSampler.select`` on numpy batches,
        #  ``compute_temporal_embeddings_p`` with one p_pos and three p_neg per interaction: 3 072 roots per batch instead of
        #  2 560, the device-resident comparator is the C3 line of profiles/r*_bench_all_configs.jsonl)
        #  ``overlap_backward``: the native backward and the optimizer's kernel on a stream of their own - the loop's
        #  ``loss.item()`` waits for the forward only and the host prepares the next batch beside the backward)
        native_bpr = name.endswith("bpr_loss_blocks")
        ours = name.startswith("ours")
        overlap = "overlap_backward" in name
        if ours and mvs is None:
            prs = np.random.RandomState(7)
            prices = 100.0 * np.exp(np.cumsum(prs.randn(cfg.n_days, cfg.n_items, 30) * 0.02, axis=2))
            mvs = P.MVSampler(prices, g.upper_u, tgn.device, gamma=2.0, lambda_mv=0.5, p_pos_num=1, p_neg_num=3, day_of=g.day_of)
        tgn.join()
        torch.cuda.synchronize()
        tgn.overlap_backward = False
        if "torch.optim.Adam" in name:
            opt = torch.optim.Adam(tgn.parameters(), lr=wl.args.lr)
            if overlap:
                P.overlap_backward(tgn, opt)
        else:
            opt = P.FusedAdam(tgn, lr=wl.args.lr, overlap_backward=overlap)
        if ours:
            out.setdefault("loop_ours", "main.py:160-394 ('ours' branch, main.py:190-337) - MVSampler.select in place of the inline block")
        t_sampler = t_embed = t_loss = t_bwd = t_opt = t_item = 0.0
        n, i, t_all0 = 0, 0, None
        while True:
            if i == 3:                                           # three untimed steps (allocator, caches)
                torch.cuda.synchronize()
                t_sampler = t_embed = t_loss = t_bwd = t_opt = t_item = 0.0
                n, t_all0 = 0, time.perf_counter()
            s = wl.start + (i % n_vis) * B
            t0 = time.perf_counter()
            optimizer = opt
            optimizer.zero_grad()
            sources_batch, destinations_batch = d.sources[s:s + B], d.destinations[s:s + B]
            edge_idxs_batch, timestamps_batch = d.edge_idxs[s:s + B], d.timestamps[s:s + B]
            portfolios_batch = portfolios_all[s - wl.start:s - wl.start + B]
            train_rand_sampler = P.RandEdgeSampler(sources_batch, d.destinations, portfolios_batch, g.upper_u, g.map_item_id)
            negatives_batch = train_rand_sampler.sample(size=q)
            if ours:
                p_pos_batch, p_neg_batch = mvs.select(destinations_batch, negatives_batch, timestamps_batch,
                                                      train_rand_sampler.port_idx, train_rand_sampler.port_len)
            t1 = time.perf_counter()
            tgn = tgn.train()
            if ours:
                source_embedding, _, destination_embedding, negative_embedding = tgn.compute_temporal_embeddings_p(
                    sources_batch, destinations_batch, p_pos_batch, p_neg_batch, timestamps_batch, edge_idxs_batch, K)
            else:
                source_embedding, destination_embedding, negative_embedding = tgn.compute_temporal_embeddings(
                    sources_batch, destinations_batch, negatives_batch.flatten(), timestamps_batch, edge_idxs_batch, K)
            t2 = time.perf_counter()
            bsbs = source_embedding.shape[0]
            source_embedding = source_embedding.view(bsbs, 1, -1)
            destination_embedding = destination_embedding.view(bsbs, 1, -1)          # (ours: the p_pos block, p_pos_num = 1)
            negative_embedding = negative_embedding.view(bsbs, q, -1)                # (ours: the p_neg block, p_neg_num = 3)
            if native_bpr:
                loss = P.bpr_loss_blocks(source_embedding, destination_embedding, negative_embedding)
            else:
                pos_scores = torch.sum(source_embedding * destination_embedding, dim=2)
                neg_scores = torch.matmul(source_embedding, negative_embedding.transpose(1, 2)).squeeze()
                score_diff = pos_scores - neg_scores
                score_diff_mean = torch.mean(score_diff, dim=1)
                log_and_sigmoid = torch.log(torch.sigmoid(score_diff_mean))
                loss = -torch.mean(log_and_sigmoid)
            t3 = time.perf_counter()
            loss.backward()
            t4 = time.perf_counter()
            optimizer.step()
            t5 = time.perf_counter()
            loss_value = loss.item()
            tgn.memory.detach_memory() if tgn.memory is not None else None
            t6 = time.perf_counter()
            t_sampler += t1 - t0; t_embed += t2 - t1; t_loss += t3 - t2; t_bwd += t4 - t3; t_opt += t5 - t4; t_item += t6 - t5
            n += 1
            i += 1
            if t_all0 is not None and time.perf_counter() - t_all0 >= budget_s and n >= 10:
                break
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_all0
        ms = lambda x: round(1e3 * x / n, 4)
        out[name] = {"value": round(n * B / wall, 1), "unit": "interactions/s", "ms_per_step": ms(wall), "timed_steps": n,
                     "final_loss": round(float(loss_value), 5),
                     "host_ms_per_step": {("RandEdgeSampler(...) + sample() + MVSampler.select() [two device syncs]" if ours else
                                           "RandEdgeSampler(...) + sample() [incl. its device sync]"): ms(t_sampler),
                                          "compute_temporal_embeddings [enqueue]": ms(t_embed), "BPR expression [enqueue]": ms(t_loss),
                                          "loss.backward() [enqueue]": ms(t_bwd), "optimizer.step() [enqueue]": ms(t_opt),
                                          "loss.item() + detach_memory() [wait for the device]": ms(t_item)}}
        if isinstance(opt, P.FusedAdam):
            opt.zero_grad(set_to_none=True)
        tgn.join()
        tgn.overlap_backward = False
    tgn.train(was_training)
    tgn.join()
    torch.cuda.synchronize()
    return out


def family_roofline(fam, v, profiled_workload=True):
    """Roofline record of ONE kernel family (= one kernel, every launch of it over the sampled steps).
    Split contractions: MFMA, against the dense 16-bit peak / the piece products one fp32 product costs (three for the
    two-piece fp16 split, six for bf16x3); fp32-MFMA kernels against the fp32 MFMA peak.  The attention kernels gather rows
    that sit in L2 / Infinity Cache and are bound by their vector instruction stream: ``bound: "valu"`` - achieved = vector
    wave-instructions per launch (SQ_INSTS_VALU of the committed counter pass) / the live launch time, peak = 1024 SIMD-32 x
    2.4 GHz / 2 cycles (VALU_ISSUE_PEAK_GINST); the HBM view (counter bytes / time) and the no-reuse algorithmic bytes of SURVEY 8(d) ride along.  The
    sampler is priced against HBM."""
    per_launch_s = v["ms"] / v["count"] * 1e-3
    traffic, sq, src, names = pmc_counters(fam) if profiled_workload else (None, {}, None, [])
    if fam.startswith("gemm") or fam == "gru_fused":
        products = bx_products(fam)
        peak = MFMA_BF16_PEAK_TF / products if products else MFMA_F32_PEAK_TF
        achieved = v["work"] / v["count"] / per_launch_s / 1e12
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic}
        if products == 6:
            roof["arithmetic"] = ("fp32 contraction as a 3-way bf16 operand split: 6 v_mfma_f32_16x16x32_bf16 per fp32 "
                                  "product block; peak = dense bf16 MFMA peak / 6")
        elif products == 3:
            roof["arithmetic"] = ("fp32 contraction as a 2-way scaled fp16 operand split: 3 v_mfma_f32_16x16x32_f16 per fp32 "
                                  "product block; peak = dense f16 MFMA peak / 3")
    elif fam.startswith("attn") and sq.get("SQ_INSTS_VALU"):
        ginst = sq["SQ_INSTS_VALU"] / per_launch_s / 1e9
        roof = {"bound": "valu", "achieved": round(ginst, 1), "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-inst/s",
                "frac": round(ginst / VALU_ISSUE_PEAK_GINST, 4), "traffic": traffic,
                "valu_insts_per_launch": sq["SQ_INSTS_VALU"], "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"),
                "algorithmic_gbs_no_reuse": round(v["work"] / v["count"] / per_launch_s / 1e9, 1)}
    else:
        achieved = v["work"] / v["count"] / per_launch_s / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic}
        if fam.startswith("attn"):
            # (no committed counter pass of this workload: the gathered rows sit in L2 / Infinity Cache, so the no-reuse
            #  algorithmic bytes of SURVEY 8(d) over the HBM peak are NOT a bound and would print above 1 - no fraction then)
            roof["frac"] = None
            roof["note"] = ("no committed counter pass for this workload (profiles/r*_summary_%s.json): the kernel is bound by its "
                            "vector instruction stream, and the no-reuse algorithmic bytes are not an HBM bound" % WORKLOAD_KEY[0])
    if traffic:
        roof["hbm_gbs_from_traffic"] = round(traffic / per_launch_s / 1e9, 1)
        roof["hbm_frac_from_traffic"] = round(traffic / per_launch_s / 1e9 / HBM_PEAK_GBS, 4)
    if fam == "gemm_multi":
        roof["note"] = ("grouped 32 x 64-tile fp32-MFMA products of the composite-weight builds and their gradient chain: ~40 MFLOP each, "
                        "launch- and k-chain-latency bound, queued on the library's side streams - their brackets and their resident "
                        "time include waiting for workgroup slots beside the caller's stream's kernels; none is on the critical path")
    roof["counters_from"] = src
    roof["kernel"] = fam
    roof["kernel_name"] = ", ".join(names) if names else FAMILY_KERNEL.get(fam, fam)
    roof["avg_launch_us"] = round(per_launch_s * 1e6, 2)
    roof["launches"] = int(v["count"])
    return roof


def trace_family_ms(prof):
    """Resident device time per step of every family in the committed kernel trace of this workload ({} without one)."""
    import csv
    import glob
    key = WORKLOAD_KEY[0]
    tf = _trace_for(key)
    files = [tf] if tf else []
    if not files:
        return {}, None
    try:
        sf = _summary_for(key)
        steps = int(json.load(open(sf)).get("steps_in_trace", 25)) if sf else 25
        out = {}
        for r in csv.DictReader(open(files[-1])):
            base = _kernel_base(r["Name"])
            for fam in FAMILY_KERNEL:
                if not _family_has(fam, r["Name"]) or fam not in prof or prof[fam]["count"] <= 0:
                    continue
                if base == "gemm_f32_kernel":
                    continue            # (three families share this kernel name: they keep their bracketed times)
                out[fam] = out.get(fam, 0.0) + float(r["TotalDurationNs"]) / steps / 1e6
        return out, os.path.basename(files[-1])
    except Exception:
        return {}, None


def roofline_of(prof, n_prof_steps, profiled_workload=True):
    """Dominant KERNEL of the step, priced by ``family_roofline`` (a family is ONE kernel name, every launch of it; the library
    reads the device-side row counts of the touched-table launches back, so every launch carries its work).  The families
    are ranked by the device time they hold in the committed kernel trace of this workload (resident time, co-scheduled
    launches as the step runs them: ``ranked_by`` names the file) and, without a trace, by the summed live brackets; the rate
    that is priced is always the live bracketed one (the kernel alone).  The runner-up rides along in full."""
    fam = {k: v for k, v in prof.items() if v["count"] > 0}
    if not fam:
        return None
    resident, trace_file = trace_family_ms(prof) if profiled_workload else ({}, None)
    rank_ms = {k: resident.get(k, fam[k]["ms"] / max(1, n_prof_steps)) for k in fam}
    order = sorted(fam, key=lambda k: -rank_ms[k])
    dom = order[0]
    roof = family_roofline(dom, fam[dom], profiled_workload)
    v = fam[dom]
    roof["launches_per_step"] = round(v["count"] / max(1, n_prof_steps), 2)
    roof["ms_per_step"] = round(v["ms"] / max(1, n_prof_steps), 4)
    roof["ms_per_step_resident"] = round(rank_ms[dom], 4)
    roof["ranked_by"] = ("resident time in profiles/%s" % trace_file) if trace_file else "summed live brackets (no committed trace of this workload)"
    roof["sampled_steps"] = n_prof_steps
    if len(order) > 1:
        ru = family_roofline(order[1], fam[order[1]], profiled_workload)
        ru["launches_per_step"] = round(fam[order[1]]["count"] / max(1, n_prof_steps), 2)
        ru["ms_per_step"] = round(fam[order[1]]["ms"] / max(1, n_prof_steps), 4)
        ru["ms_per_step_resident"] = round(rank_ms[order[1]], 4)
        roof["runner_up"] = ru
    roof["families_ms_per_step_resident"] = {k: round(x, 4) for k, x in resident.items()}
    roof["families_ms_per_step"] = {k: round(x["ms"] / max(1, n_prof_steps), 4) for k, x in prof.items() if x["count"] > 0}
    roof["families_launches_per_step"] = {k: round(x["count"] / max(1, n_prof_steps), 2) for k, x in prof.items() if x["count"] > 0}
    is_flop = lambda k: k.startswith("gemm") or k == "gru_fused"
    rate = lambda x, unit: round(x["work"] / max(1e-12, x["ms"] * 1e-3) / unit, 2)
    roof["families_rate"] = {k: rate(x, 1e12 if is_flop(k) else 1e9) for k, x in prof.items() if x["count"] > 0}
    tot = lambda ks: (sum(prof[k]["work"] for k in ks if k in prof) / max(1e-9, sum(prof[k]["ms"] for k in ks if k in prof) * 1e-3))
    roof["gemm_all_tflops"] = round(tot([k for k in prof if is_flop(k)]) / 1e12, 2)
    roof["attn_all_gbs"] = round(tot([k for k in prof if k.startswith("attn")]) / 1e9, 1)
    roof["bracketed_ms_per_step"] = round(sum(x["ms"] for x in prof.values()) / max(1, n_prof_steps), 4)
    td = trace_dominant(prof, n_prof_steps) if profiled_workload else None
    if td:
        roof["trace_dominant"] = td
    # The peaks above are priced at the 2.4 GHz of MI355X_MICROARCH.md; the shader clock the product kernels actually hold is
    # stamped in three of them on every launch (pfo_shader_clock: s_memtime against the 100 MHz counter).  A second fraction at
    # the measured clock rides along for the compute-bound families - the first one is never switched silently.
    try:
        from pfotgnrec_amd import _lib as _l
        clk = {k: round(v, 3) for k, v in _l.shader_clock().items() if v > 0}
        if clk:
            roof["shader_clock_ghz_measured"] = clk
            roof["shader_clock_ghz_priced"] = 2.4
            own = clk.get({"gemm_tn_bx8": "gemm_tn_bx"}.get(dom, dom)) or (clk.get("gemm_tn_bx") if dom.startswith("gemm") else None)
            if own and roof.get("bound") in ("mfma", "valu") and roof.get("frac") is not None:
                roof["frac_at_measured_clock"] = round(roof["frac"] * 2.4 / own, 4)
                roof["frac_at_measured_clock_note"] = "frac x 2.4 / %.3f GHz (the clock stamped inside %s)" % (own, "this kernel" if dom in clk or dom == "gemm_tn_bx8" else "gemm_tn_bx")
    except Exception as e:
        roof["shader_clock_ghz_measured"] = {"error": repr(e)[:120]}
    sf = _summary_for(WORKLOAD_KEY[0]) if profiled_workload else None
    roof["profiles_round"] = os.path.basename(sf).split("_")[0] if sf else None
    if sf:
        try:
            roof["profiles_same_build_bench_line"] = json.load(open(sf)).get("bench_line_same_build")
        except Exception:
            pass
    return roof


def cfg_layers(cfg_name):
    from pfotgnrec_amd.synthetic import CONFIGS
    return CONFIGS[cfg_name].n_layers


BX_FAMILIES = ("gemm_bx", "gemm_tn_bx", "gemm_tn_bx8", "gemm_bx_skinny", "gru_fused")


def bx_products(family):
    """MFMA instructions per fp32 product block of a split-contraction family (0: not one).  The image kernels take the
    two-piece fp16 format (3 products) unless PFO_BX_FMT=0 keeps them on bf16x3; the weight-gradient tile likewise (PFO_TN_FMT)."""
    if family not in BX_FAMILIES:
        return 0
    if family in ("gemm_tn_bx", "gemm_tn_bx8"):
        return 6 if os.environ.get("PFO_TN_FMT", "1") == "0" else 3
    return 6 if os.environ.get("PFO_BX_FMT", "1") == "0" else 3


XGMI_LINK_GBS = 153.0      # per-link, per direction (task statement: 7 links x ~153 GB/s per GPU)


def predicted_ring_allreduce_us(n_bytes, world, hop_us=6.0):
    """Ring all-reduce of n_bytes over xGMI at `world` ranks: 2 (N-1)/N of the buffer crosses each rank's link in each
    direction + 2 (N-1) hop latencies.  A prediction, never measured on this pool."""
    if world <= 1:
        return 0.0
    return 2.0 * (world - 1) / world * n_bytes / (XGMI_LINK_GBS * 1e3) + 2.0 * (world - 1) * hop_us


class SleepCollective:
    """--emulate-ranks: a stand-in for the gradient all-reduce that holds the library's side stream for the predicted ring
    time (a device-side spin, torch.cuda._sleep, calibrated once against events), so that the fused schedule's join behind a
    ~100 us exchange is rehearsed on one GPU: what it costs the NEXT step's head, not what the exchange itself costs."""

    def __init__(self, us):
        import torch
        self.us = float(us)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        a.record(); torch.cuda._sleep(2_000_000); b.record()
        torch.cuda.synchronize()
        self.cycles_per_us = 2_000_000 / (a.elapsed_time(b) * 1e3)
        self.calls = 0

    def __call__(self):
        import torch
        if self.us > 0:
            torch.cuda._sleep(int(self.us * self.cycles_per_us))     # (runs on the current stream = the side stream inside bpr_step)
        self.calls += 1


class SleepCollectiveBuckets(SleepCollective):
    """The two-piece exchange rehearsed: the top layer's block holds a communication stream for its predicted ring time from the
    backward's "top layer final" event on (beside the rest of the backward), the rest holds the side stream behind the
    backward's end.  ``ordered``: the communication stream is not joined here - FusedAdam steps the first-use bucket at once and
    joins in front of the top block's step (distributed.allreduce_flat_grad_ordered); otherwise as allreduce_flat_grad_buckets."""

    def __init__(self, tgn, us_top, us_rest, ordered):
        super().__init__(us_rest)
        import torch
        self.tgn, self.us_top, self.ordered = tgn, float(us_top), ordered
        # ONE communication stream per device for the life of the process (the one the real exchange uses): every further
        # stream of the normal priority class ends up sharing a hardware queue with the caller's stream
        from pfotgnrec_amd.distributed import _COMM_STREAMS
        dev = tgn.flat_parameters.device
        if _COMM_STREAMS.get(dev) is None:
            _COMM_STREAMS[dev] = torch.cuda.Stream(device=dev)
        self.comm = _COMM_STREAMS[dev]

    def __call__(self):
        import torch
        tgn = self.tgn
        main = torch.cuda.current_stream()
        fresh, tgn._bucket_event_fresh = tgn._bucket_event_fresh, False
        if fresh:
            self.comm.wait_event(tgn._bucket_event)
        else:
            self.comm.wait_stream(main)
        with torch.cuda.stream(self.comm):
            if self.us_top > 0:
                torch.cuda._sleep(int(self.us_top * self.cycles_per_us))
        if self.us > 0:
            torch.cuda._sleep(int(self.us * self.cycles_per_us))
        if self.ordered:
            tgn._comm_pending = self.comm
        else:
            main.wait_stream(self.comm)
        self.calls += 1


def multi_gpu_line_fields(ms_per_step, coll_ms, coll_n, allreduce_mode):
    """The fields a rank-path line carries about its collective (also built from synthetic numbers by the CPU tests)."""
    fused = allreduce_mode.startswith("fused")
    return {"collective_ms_per_step": round(coll_ms, 4), "collective_on_callers_stream": not fused,
            "compute_ms_per_step": round(ms_per_step - (0.0 if fused else coll_ms), 4), "collective_samples": coll_n}


def strong_scaling_entry(value, ms_per_step, world, workload, global_batch, collective_ms, one_gpu_value=None, one_gpu_ms=None):
    e = {"n_gpus": world, "value": round(value, 1), "ms_per_step": round(ms_per_step, 4),
         "collective_ms_per_step": None if collective_ms is None else round(collective_ms, 4),
         "workload": workload, "global_batch": global_batch, "scaling": "strong"}
    if one_gpu_value is not None:
        e["one_gpu_reference"] = {"value": round(one_gpu_value, 1), "ms_per_step": round(one_gpu_ms, 4)}
        e["efficiency_vs_one_gpu"] = round(value / (world * one_gpu_value), 4)
    return e


def check_line_schema(line, n_gpus):
    """What the driver's N-GPU record must hold (tests/test_host_logic.py builds a synthetic N = 8 line and runs this)."""
    need = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"]
    miss = [k for k in need if k not in line]
    if line.get("n_gpus") != n_gpus:
        miss.append("n_gpus == %d" % n_gpus)
    if n_gpus > 1:
        for k in ("collective_ms_per_step", "compute_ms_per_step", "collective_on_callers_stream", "allreduce", "collective"):
            if k not in line.get("config", {}):
                miss.append("config." + k)
        sec = line.get("secondary", {})
        if n_gpus == 8:
            sc = sec.get("strong_scaling_C4", {})
            for k in ("value", "one_gpu_reference", "efficiency_vs_one_gpu", "collective_ms_per_step"):
                if k not in sc:
                    miss.append("secondary.strong_scaling_C4." + k)
    return miss


def release_workload_memory():
    """After ``del workload``: the workload's objects form reference cycles (model <-> optimizer <-> closures), so the
    memory goes only when the cycle collector runs - and it used to run INSIDE the next workload's timed region (a full
    collection over the dead workload: the second workload of a process measured 8 % slow, 1.31 against 1.22 ms per step,
    whatever its configuration).  Collect here, then hand the cached blocks back."""
    import gc
    import torch
    gc.unfreeze()                     # (Workload.timed froze what existed then: the dead workload is among it)
    gc.collect()
    torch.cuda.empty_cache()


def emulate_ranks(args, dev):
    """Compute-side ceiling of the data-parallel scaling curve on ONE GPU: for each world size N the step of rank 0 -
    sampler / forward / loss / backward for its B/N interactions, the state update and the lazy GRU rows of ALL B global
    positives, Adam - without the all-reduce.  Strong scaling (the config's batch is fixed), default C4 at 4096."""
    import torch
    worlds = [int(x) for x in args.emulate_ranks.split(",")]
    args.graph = "off"
    if args.scaling == "weak":
        # the default N-GPU line: the per-GPU batch is fixed, the global batch (and with it the replicated state update and
        # lazy GRU rows of every rank) grows with N
        cfg_name = args.config or "C2"
        rows = []
        for n in worlds:
            wl = Workload(args, cfg_name, dev, 0, n, "weak", emulate=True)
            el, nt, _, _, _ = wl.timed(args.steps, args.warmup, min(args.min_seconds, 1.0), 0, first_step=1000 * n, collective=False)
            ms = 1e3 * el / nt
            rows.append({"world": n, "rank": 0, "local_batch": wl.per_gpu, "global_batch": wl.B, "ms_per_step": round(ms, 4),
                         "stub_collective_us_on_side_stream": round(getattr(wl, "sleep_coll", None).us, 1) if getattr(wl, "sleep_coll", None) else 0.0,
                         "stub_collective_us_on_comm_stream": round(getattr(getattr(wl, "sleep_coll", None), "us_top", 0.0), 1),
                         "interactions_per_s_if_all_ranks_like_this": round(wl.B / (ms * 1e-3), 1),
                         "host_enqueue_ms_per_step": round(float(np.median(wl.host_ms)), 4)})
            desc = wl.describe()
            if args.marks > 0:                                # milestones of this emulated rank (stderr)
                from pfotgnrec_amd import _lib
                torch.cuda.synchronize()
                _lib.marks_enable(True)
                for k in range(args.marks):
                    wl.step(900005 + k)
                torch.cuda.synchronize()
                _lib.marks_enable(False)
                sys.stderr.write("milestones (emulated rank 0 of %d, mean over %d steps):\n%s" % (n, args.marks, _lib.marks_dump()))
            del wl
            release_workload_memory()
        for r in rows:
            r["compute_scaling_efficiency_vs_first"] = round(rows[0]["ms_per_step"] / r["ms_per_step"], 4)
        print(json.dumps({"emulated_ranks": True, "scaling": "weak", "workload": desc, "collective": ("stub: device-side spins of the predicted ring times, form '%s' (buckets: the top block on a communication stream from the 'top layer final' event on, the rest on the library's side stream)" % args.allreduce) if args.emulate_sleep else "none (stubbed)",
                          "device": torch.cuda.get_device_name(dev), "table": rows}), flush=True)
        return
    cfg_name = args.config or "C4"
    wl = Workload(args, cfg_name, dev, 0, 1, "strong", emulate=True)
    rows = []
    for n in worlds:
        wl.set_world(0, n)
        el, nt, _, _, _ = wl.timed(args.steps, args.warmup, min(args.min_seconds, 1.0), 0, first_step=1000 * n, collective=False)
        ms = 1e3 * el / nt
        rows.append({"world": n, "rank": 0, "local_batch": wl.B // n, "ms_per_step": round(ms, 4),
                     "stub_collective_us_on_side_stream": round(wl.sleep_coll.us, 1) if getattr(wl, "sleep_coll", None) and n > 1 else 0.0,
                     "interactions_per_s_if_all_ranks_like_this": round(wl.B / (ms * 1e-3), 1),
                     "host_enqueue_ms_per_step": round(float(np.median(wl.host_ms)), 4)})
    base = rows[0]["ms_per_step"] * rows[0]["world"]
    for r in rows:
        r["compute_scaling_efficiency_vs_first"] = round(base / (r["ms_per_step"] * r["world"]), 4)
    print(json.dumps({"emulated_ranks": True, "workload": wl.describe(), "global_batch": wl.B, "collective": "none (stubbed)",
                      "device": torch.cuda.get_device_name(dev), "table": rows}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))                   # parent: nothing below runs here, no GPU is ever touched
    if args.launcher_selftest:
        return launcher_selftest(args)
    import torch
    import torch.distributed as dist
    from pfotgnrec_amd import _lib
    from pfotgnrec_amd.distributed import init_from_env

    rank, world, local = init_from_env()
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if os.environ.get("PFO_FORCE_DEVICE") is not None:      # test hook: several ranks on one GPU
        local = int(os.environ["PFO_FORCE_DEVICE"])
    assert local < torch.cuda.device_count(), "rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.emulate_ranks:
        return emulate_ranks(args, dev)
    cfg_name = args.config or "C2"       # the headline graph at every N (BASELINE.json: "1M-edge graph at 1/2/4/8 MI355X")
    scaling = args.scaling or ("strong" if (cfg_name == "C4" and world > 1) else "weak")
    wl = Workload(args, cfg_name, dev, rank, world, scaling)
    prof_every = 0 if args.no_prof else max(1, args.prof_every)
    elapsed, n_timed, blocks, final_loss, n_prof_steps = wl.timed(args.steps, args.warmup, args.min_seconds, prof_every)
    prof = _lib.prof_collect() if not args.no_prof else None
    _lib.prof_enable(False)
    if args.marks > 0:
        for k in range(5):
            wl.step(900000 + k)
        torch.cuda.synchronize()
        _lib.marks_enable(True)
        for k in range(args.marks):
            wl.step(900005 + k)
        torch.cuda.synchronize()
        _lib.marks_enable(False)
        if rank == 0:
            sys.stderr.write("milestones (mean over %d steps, caller's stream):\n%s" % (args.marks, _lib.marks_dump()))
    B = wl.B
    value = n_timed * B / elapsed
    out = {
        "metric": "interactions/sec (TGN fwd+BPR step)", "value": round(value, 1), "unit": "interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / n_timed, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl.describe(), "global_batch": B, "parallelism": "dp%d" % world, "dropout": args.dropout,
                   "final_loss": round(final_loss, 5), "timed_blocks": blocks, "timed_steps": n_timed,
                   "timed_seconds": round(elapsed, 3),
                   "block_ms_per_step": {"first": round(wl.block_ms[0], 4), "min": round(min(wl.block_ms), 4),
                                         "median": round(float(np.median(wl.block_ms)), 4), "last": round(wl.block_ms[-1], 4)},
                   **({"block_ms_all": [round(x, 4) for x in wl.block_ms]} if os.environ.get("PFO_BENCH_BLOCKS") else {}),
                   "host_enqueue_ms_per_step": round(float(np.median(wl.host_ms)), 4), "workload_build_s": round(wl.build_s, 2), "hip_graph": wl.graph_note, "deterministic_backward": bool(args.deterministic),
                   "next_batch_prepared_beside_backward": bool(wl.prefetch and wl.gstep is None and wl.mvs is None),
                   "collective": ("%s all-reduce of the flat fp32 gradient, world %d"
                                  % ("rccl" if dist.get_backend() == "nccl" else dist.get_backend(), dist.get_world_size())) if dist.is_initialized() else None},
    }
    if prof is not None:
        WORKLOAD_KEY[0] = "%s@%d" % (cfg_name, wl.per_gpu if scaling == "weak" else wl.B // world)
        roof = roofline_of(prof, n_prof_steps, profiled_workload=_summary_for(WORKLOAD_KEY[0]) is not None)
        if roof:
            out["roofline"] = roof

    wl_gone = False
    coll_ms, coll_n = wl.collective_ms()
    if coll_ms is not None:
        # ('fused': the collective runs on the library's side stream beside the next batch's sampling - its time is not part
        #  of the caller's stream's step; the other forms hold the caller's stream for it)
        out["config"].update(multi_gpu_line_fields(1e3 * elapsed / n_timed, coll_ms, coll_n, wl.allreduce_mode))
    if world > 1 or wl.force_dist:
        out["config"]["allreduce"] = wl.allreduce_mode
        out["config"]["predicted_scaling_efficiency"] = {"weak_C2_512_per_gpu": {"2": 0.96, "4": 0.94, "8": 0.90},
                                                         "strong_C4_4096_global": {"2": 0.91, "4": 0.79, "8": 0.58},
                                                         "source": "DESIGN.md 6: emulated-rank compute table (round 4), collective on the "
                                                                   "side stream beside the next batch's head ('fused')"}

    if (world > 1 or wl.force_dist) and not args.no_secondary:
        # Secondary figures ride in the same line.  Nothing here may cost the main line: every part is bounded (1 s of timed
        # steps) and an exception is recorded instead of raised.
        sec = {}
        try:
            # (0) the same workload with the OTHER all-reduce form (one piece after the backward / two pieces, the top layer's
            #     block on a communication stream beside the backward): single vs. bucketed decided by data
            for other in [m for m in ("single", "buckets", "fused", "fused_buckets", "fused_ordered") if m != wl.allreduce_mode]:
                if (other.endswith("buckets") or other.endswith("ordered")) and cfg_layers(cfg_name) < 2:
                    continue
                wl.allreduce_mode, wl.tgn.dp_bucketed, wl.tgn.dp_ordered = other, other.endswith("buckets") or other.endswith("ordered"), other.endswith("ordered")
                el, nb, _, _, _ = wl.timed(args.steps, 5, min(args.min_seconds, 1.0), 0, first_step=50000)
                cms, cn = wl.collective_ms()
                sec["allreduce_" + other] = {"n_gpus": world, "value": round(nb * B / el, 1), "ms_per_step": round(1e3 * el / nb, 4),
                                            "collective_ms_per_step": None if cms is None else round(cms, 4),
                                            "workload": "the main line's workload, all-reduce form '%s'" % other}
                wl.tgn.join()
                wl.allreduce_mode, wl.tgn.dp_bucketed, wl.tgn.dp_ordered = args.allreduce, args.allreduce.endswith("buckets") or args.allreduce.endswith("ordered"), args.allreduce.endswith("ordered")
            want_other_case = world > 1 and (args.secondary or world == 8)
            # (1) the same workload on ONE of these GPUs (rank 0 alone, the others wait): the denominator of the strong-scaling
            #     figure.  (2) the other scaling case.
            if want_other_case and scaling == "strong":
                dist.barrier()
                if rank == 0:
                    wl.set_world(0, 1)
                    el, n1, _, _, _ = wl.timed(args.steps, 2, min(args.min_seconds, 1.0), 0, first_step=100000, collective=False)
                    sec["strong_scaling_reference"] = {"n_gpus": 1, "value": round(n1 * B / el, 1), "ms_per_step": round(1e3 * el / n1, 4),
                                                       "workload": "the same global batch on one GPU"}
                    wl.set_world(0, world)
                dist.barrier()
            if want_other_case and not (cfg_name == "C2" and scaling == "weak"):
                del wl
                release_workload_memory()
                w2 = Workload(args, "C2", dev, rank, world, "weak")
                el, n2, _, _, _ = w2.timed(args.steps, args.warmup, min(args.min_seconds, 1.0), 0)
                cms, _ = w2.collective_ms()
                sec["weak_scaling_C2"] = {"n_gpus": world, "value": round(n2 * w2.B / el, 1), "ms_per_step": round(1e3 * el / n2, 4),
                                          "collective_ms_per_step": None if cms is None else round(cms, 4),
                                          "workload": w2.describe(), "global_batch": w2.B}
                del w2
            elif want_other_case:
                # BASELINE.json configs[3] - the SURVEY 8(d) strong-scaling case - beside the headline: C4 (10 M edges) at a
                # FIXED global batch of 4096 cut into N shards, and the same batch on one of these GPUs (rank 0 alone, the
                # others wait)
                del wl
                release_workload_memory()
                w4 = Workload(args, "C4", dev, rank, world, "strong")
                el, n4, _, _, _ = w4.timed(args.steps, args.warmup, min(args.min_seconds, 1.0), 0)
                cms, _ = w4.collective_ms()
                v4, ms4 = n4 * w4.B / el, 1e3 * el / n4
                sec["strong_scaling_C4"] = strong_scaling_entry(v4, ms4, world, w4.describe(), w4.B, cms)
                sec["strong_scaling_C4"]["workload_build_s"] = round(w4.build_s, 2)
                dist.barrier()
                if rank == 0:
                    w4.set_world(0, 1)
                    el, n1, _, _, _ = w4.timed(args.steps, 2, min(args.min_seconds, 1.0), 0, first_step=100000, collective=False)
                    sec["strong_scaling_C4"] = dict(strong_scaling_entry(v4, ms4, world, w4.describe(), w4.B, cms, n1 * w4.B / el, 1e3 * el / n1),
                                                    workload_build_s=round(w4.build_s, 2))
                dist.barrier()
                del w4
        except Exception as e:                                  # the secondary figures never cost the main line
            sec["error"] = repr(e)[:300]
        out["secondary"] = sec

    if world == 1 and not wl_gone and not args.no_drop_in and not (wl.force_dist or args.deterministic) and wl.mvs is None and wl.cfg.use_memory and wl.gstep is None and not args.emulate_ranks:
        try:
            out.setdefault("secondary", {})["drop_in_surface"] = drop_in_surface(wl)
            di = out["secondary"]["drop_in_surface"]
            for k in DROP_IN_ENTRIES[:6]:
                di[k]["vs_device_resident_step"] = round(di[k]["ms_per_step"] / out["ms_per_step"], 3)
        except Exception as e:                                  # a secondary figure never costs the main line
            out.setdefault("secondary", {})["drop_in_surface"] = {"error": repr(e)[:300]}
    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    if not args.no_cpu_baseline and world == 1:
        from pfotgnrec_amd.synthetic import CONFIGS, make_graph
        try:
            threads = os.cpu_count() or 1
            try:                                            # threads the BLAS behind numpy actually uses
                from threadpoolctl import threadpool_info
                threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
            except Exception:
                pass
            cfg = CONFIGS[cfg_name]
            v, spent = cpu_baseline(cfg, make_graph(cfg, with_prices=False), args.cpu_batch, args.cpu_steps)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "interactions/s", "cores": threads, "kind": "port",
                                   "sample": "%d step(s) of %d interactions of the same workload after one untimed step "
                                             "(oracle/tgn_oracle.py: numpy fp32 + BLAS + C fmaf/cosf helper; the attention layers' "
                                             "53 760-instance calls run as 2 048-row chunks on a thread pool of min(64, cores) "
                                             "workers with single-threaded BLAS inside, everything else on %d BLAS threads; "
                                             "steady-state memory), median step, %.1f s timed.  Per thread: %.2f interactions/s - SURVEY 8(d)'s "
                                             "anchor is the REFERENCE itself at 34-39 interactions/s on 8 vCPU = 4.3-4.9 per core (torch CPU, build "
                                             "container); this port is faster in absolute terms but does not scale with the box's threads (numpy "
                                             "gathers and the Python glue are serial), so per core it sits well below the anchor: a stated baseline, "
                                             "not a tuned one"
                                             % (args.cpu_steps, args.cpu_batch, threads, spent, v / max(1, threads)),
                                   "per_thread": round(v / max(1, threads), 3), "survey_anchor_reference_per_core": [4.3, 4.9],
                                   "phases_median_s": getattr(cpu_baseline, "phases", None)}
        except Exception as e:  # the baseline never blocks the measurement
            out["cpu_baseline"] = {"value": None, "unit": "interactions/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
