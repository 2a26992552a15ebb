#!/usr/bin/env python3
"""bench.py — headline benchmark of the Smith-Waterman hot path on MI355X.

Metric (BASELINE.json): GCUPS = sum |x_k|*|y| / t * 1e-9 for 150 bp synthetic reads against a 50 Mbp
synthetic reference (configs[2], "100k x 150 bp vs 50 Mbp"), whole job: score pass + argmax +
traceback + results on the host, inputs resident in HBM when the timed region starts.

One "step" = one pass of the hot path over one batch of --reads reads per GPU (default 2048; the
full 100 000-read set is 49 such batches — 7.5e14 cells — and the per-batch rate is size-normalised).

Launch:  python bench.py --gpus N --steps K --warmup W
  * N = 1: runs in this process.
  * N > 1: this process touches no GPU; it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD (never exec), relays rank 0's JSON line and exits with the child's code.
  * already under torch.distributed.run (RANK in the environment — the way the round-end driver starts the N > 1
    runs): this process is one of the N ranks.
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI), reads sharded across ranks,
reference replicated, no data-path collective; one 8-byte all-reduce(MAX) per step merges the per-rank best
(score, read) — `value` is weak scaling (per-GPU work fixed); `strong_scaling` times a fixed total (config 3 reads
split over the ranks; config 5's reference pieces dealt to the ranks, src/aligner/plocalaligner.cpp:110-129).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (packed op = 1 lane-op)


def load_package():
    name = "parallel_genomeseq_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "parallel-genomeseq_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def host_cores():
    """Host cores this job may use: the cgroup CPU quota when there is one, otherwise the affinity
    mask capped at the GPU box's per-GPU CPU share (16 per visible GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        import torch
        share = 16 * max(1, torch.cuda.device_count())
    except Exception:
        share = 16
    return max(1, min(n, share))


def cpu_baseline(pgs, ref, read_len, seconds_hint=20.0):
    """The reference's own OpenMP path (oracle/_ref/ref_driver_omp = unmodified reference sources
    behind our driver, src/sw_solve_big.cpp:78-92 loop) on this box's host cores, bounded sample.
    Falls back to the oracle's scalar port when the reference build did not travel."""
    cores = host_cores()
    sys.path.insert(0, ROOT)
    from oracle import refproc
    sub_len = min(len(ref), 5_000_000)
    nreads = 96
    sub = ref[:sub_len]
    reads, _ = pgs.synth.fast_reads_from_ref(sub, 77, nreads, read_len)
    if os.access(refproc.DRIVER_OMP, os.X_OK):
        with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
            f.write(sub.tobytes() + b"\n")
            for r in reads:
                f.write(r.tobytes() + b"\n")
            path = f.name
        env = dict(os.environ, OMP_PLACES="cores", OMP_PROC_BIND="close", OMP_DYNAMIC="false",
                   OMP_THREAD_LIMIT=str(cores))
        try:
            t0 = time.time()
            out = refproc.run(["bench %s %d 1" % (path, 2 * cores)], omp=True, env=env, timeout=600)[0].split()
            wall = time.time() - t0
        finally:
            os.unlink(path)
        kv = dict(zip(out[0::2], out[1::2]))
        return {"value": float(kv["gcups_iterate"]), "unit": "GCUPS", "cores": cores, "kind": "reference",
                "sample": "%d reads x %d bp vs first %d bp of the reference; OMPParallelLocalAligner<Skewed,SWAligner<Skewed>> "
                          "with %d pieces, overlap 2.0, OMP_PLACES=cores OMP_PROC_BIND=close; value = reference's own metric "
                          "(iterate() time only, sw_solve_big.cpp:92-99); end-to-end incl. matrix alloc/zero + winner re-run = %.3f GCUPS; %.1f s wall"
                          % (nreads, read_len, sub_len, 2 * cores, float(kv["gcups_wall"]), wall)}
    from oracle import binding as ob
    n = min(len(ref), 2_000_000)
    t0 = time.time()
    for r in reads[:4]:
        ob.score_only(r.tobytes(), ref[:n].tobytes(), ob.U8SAT)
    dt = time.time() - t0
    return {"value": 4 * read_len * n / dt * 1e-9, "unit": "GCUPS", "cores": 1, "kind": "port",
            "sample": "4 reads x %d bp vs first %d bp, scalar uint8 score-only port (oracle/sw_oracle.c)" % (read_len, n)}


def measure_traffic(args, kernel_substr="sw_score_kernel", timeout=240):
    """HBM bytes per launch of the dominant score kernel, MEASURED in this run: two child passes of this very script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the TCC block has no room for both, and no trace
    flag rides along — MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots), one step each, nothing else in the child.
    FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B; x2.0 also calibrated for
    this kernel's byte-wide reference loads by tools/ubench/fetch_calib.hip).  Returns (bytes per launch, description) or
    (None, reason)."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--reads", str(args.reads),
             "--read-len", str(args.read_len), "--ref-len", str(args.ref_len), "--semantics", args.semantics,
             "--match", str(args.match), "--mismatch", str(args.mismatch), "--gap", str(args.gap),
             "--no-cpu-baseline", "--no-extras", "--no-strong", "--no-traffic", "--no-parity"]
    per = {}
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        try:
            p = subprocess.run([exe, "--pmc", counter, "-d", out, "--output-format", "csv", "--"] + child, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (exit %d)" % (counter, p.returncode)
            tot, ids = 0.0, set()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        tot += float(r["Counter_Value"])
                        ids.add(r["Dispatch_Id"])
            if not ids:
                return None, "no %s dispatch in the %s pass" % (kernel_substr, counter)
            per[counter] = tot / len(ids)
        except subprocess.TimeoutExpired:
            return None, "rocprofv3 --pmc %s timed out" % counter
        finally:
            shutil.rmtree(out, ignore_errors=True)
    nbytes = per["FETCH_SIZE"] * 1024.0 * 2.0 + per["WRITE_SIZE"] * 1024.0
    return nbytes, ("measured: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child passes of `bench.py --steps 1 --warmup 0` in this run "
                    "(%.0f s); FETCH_SIZE %.0f KiB x 1024 x 2.0 + WRITE_SIZE %.0f KiB x 1024 per launch"
                    % (time.time() - t0, per["FETCH_SIZE"], per["WRITE_SIZE"]))


def lds_model(ki, kern_cells_per_s):
    """LDS traffic of the score kernel's inner loop from the instance the library reports: per step and lane ceil(R / 4)
    16-byte profile reads (ds_read_b128) + 1 code byte, for R rows of 2 (packed) or 1 cells."""
    R = ki["rows_per_lane"]
    packed = ki["cell"] in (0, 1, 4, 5)
    bytes_per_lane_step = 16.0 * ((R + 3) // 4) + 1.0
    cells_per_lane_step = R * (2 if packed else 1)
    bpc = bytes_per_lane_step / cells_per_lane_step
    peak = 256 * 256 * 2.4e9 * 1e-12                                      # 256 CUs x 256 B/clk x 2.4 GHz (MI355X_MICROARCH.md §LDS)
    ach = kern_cells_per_s * bpc * 1e-12
    return {"bytes_per_cell": bpc, "achieved": ach, "peak": peak, "unit": "TB/s", "frac": ach / peak,
            "note": "profile reads of the inner loop (ds_read_b128, conflict-free by layout: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE "
                    "= 0.3 % in profiles/); peak = 256 CU x 256 B/clk x 2.4 GHz"}


# ------------------------------------------------------------------------------------------------
# launcher: N > 1 asked for, not yet under torch.distributed.run
# ------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args, argv):
    """Parent of the N ranks: no GPU call, no torch import.  Child = torch.distributed.run with one rank per GPU."""
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True)
    line = None
    for out in child.stdout:                                  # relay; remember the result line
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc != 0 or line is None:
        sys.stderr.write("bench.py: %d-rank run failed (torch.distributed.run exit code %d, result line %s)\n"
                         % (args.gpus, rc, "present" if line else "missing"))
        sys.exit(rc if rc != 0 else 1)
    rec = json.loads(line)
    if rec.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d ranks, the job reports %r\n" % (args.gpus, rec.get("n_gpus")))
        sys.exit(1)
    rec["launcher"] = "bench.py spawned torch.distributed.run --nproc-per-node %d as a child process" % args.gpus
    print(json.dumps(rec))
    sys.exit(0)


# ------------------------------------------------------------------------------------------------
# worker
# ------------------------------------------------------------------------------------------------
class Dist:
    """The process group of this job (or none at N = 1): barrier, max-over-ranks, object gather."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = args.dist_backend
        self.dist = None
        if self.backend == "gloo":
            self.local_rank = self.local_rank % max(1, torch.cuda.device_count())   # rehearsal on fewer GPUs than ranks
        torch.cuda.set_device(self.local_rank)
        self.cdev = "cuda" if self.backend == "nccl" else "cpu"
        if self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # under torch.distributed.run even one rank joins a group
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend="gloo")
            if dist.get_world_size() != self.world:
                raise RuntimeError("process group has %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), self.world))

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync(self):
        self.torch.cuda.synchronize()

    def max_float(self, v):
        if self.dist is None:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def max_key(self, key):
        if self.dist is None:
            return int(key)
        t = self.torch.tensor([key], dtype=self.torch.int64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)           # per-rank best (score, index) over xGMI: 8 bytes
        return int(t.item())

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def describe(self):
        """Per-rank device identity + proof that the collective library saw every rank."""
        torch = self.torch
        p = torch.cuda.get_device_properties(self.local_rank)
        me = {"rank": self.rank, "device": self.local_rank, "name": p.name,
              "pci_bus_id": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0)),
              "host": socket.gethostname(), "pid": os.getpid()}
        ranks = self.gather_objects(me)
        info = {"backend": self.backend if self.dist is not None else "none (single process)", "ranks": ranks}
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            info["rccl_version"] = "unavailable: %r" % (e,)
        if self.dist is not None:
            t = torch.ones(1, dtype=torch.int64, device=self.cdev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            info["allreduce_sum_of_ones"] = int(t.item())            # == world size iff every rank took part
            info["distinct_devices"] = len({(r["host"], r["pci_bus_id"]) for r in ranks})
        return info

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(D, fn, steps):
    """barrier + synchronize, `steps` calls, synchronize + barrier; max over ranks of the wall time."""
    D.barrier(); D.sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = fn()
    D.sync(); D.barrier()
    return D.max_float(time.perf_counter() - t0), out


def engine_rate(pgs, ctx, D, nreads, read_len, ref_len, steps, **kw):
    """Whole-job rate of another engine / cell type on the resident workload (1 warm-up + `steps` timed)."""
    ctx.batch_run(raw=True, **kw)
    dt, _ = timed(D, lambda: ctx.batch_run(raw=True, **kw), steps)
    tm = ctx.last_timings()
    ki = ctx.last_kernel()
    cells = float(nreads) * read_len * ref_len
    return {"gcups": cells * steps / dt * 1e-9, "ms_per_step": dt / steps * 1e3, "kernel": ki["name"],
            "kernel_gcups": cells / (tm["score_us"] / max(1, tm["score_launches"]) * 1e-6) * 1e-9 if tm["score_us"] > 0 else None,
            "valu_ops_per_cell": ki["valu_ops_per_cell"], "steps": steps}


def extra_read_lengths(pgs, device, ref_len):
    """Whole-job rate at other read lengths (other tile shapes / kernel instances), both engines: 512 reads each
    (256 at 2048 bp, 128 at 4096 bp: strip-mined batches), 1 warm-up + 1 timed pass; per leg the counters of the candidate
    filters (queries swept again, candidates re-evaluated, settled by their first candidates)."""
    ref = pgs.synth.dna(3, ref_len)
    ctx = pgs.Context(device)
    out = {}
    try:
        ctx.set_reference(ref)
        for read_len, nreads in ((300, 512), (600, 512), (1000, 512), (2048, 256), (4096, 128)):
            reads, _ = pgs.synth.fast_reads_from_ref(ref, 4, nreads, read_len)
            ctx.batch_upload([r.tobytes() for r in reads])
            for sem, name in ((pgs.F32, "f32"), (pgs.U8SAT, "u8")):
                # the SWEPT rate: the uint8 engine's early exit (a read whose background saturates is decided by the reference's
                # first and last sub-chunks, no sweep) is switched off for it and reported separately below
                if sem == pgs.U8SAT:
                    ctx.set_option("no_u8_early")
                ctx.batch_run(semantics=sem, raw=True)
                t0 = time.perf_counter()
                ctx.batch_run(semantics=sem, raw=True)
                dt = time.perf_counter() - t0
                ki = ctx.last_kernel()
                cnt = ctx.last_counters()
                tm = ctx.last_timings()
                rec = {"gcups": float(nreads) * read_len * ref_len / dt * 1e-9, "reads": nreads,
                       "kernel": ki["name"], "valu_ops_per_cell": ki["valu_ops_per_cell"],
                       "score_kernel_ms": tm["score_us"] * 1e-3, "locate_ms": tm["locate_us"] * 1e-3,
                       "traceback_ms": tm["trace_us"] * 1e-3,
                       "filters": {k: cnt[k] for k in ("requeried", "whole_batch_again", "candidates", "first_settled")}}
                if sem == pgs.U8SAT:
                    ctx.set_option("no_u8_early", None)
                    ctx.batch_run(semantics=sem, raw=True)
                    t0 = time.perf_counter()
                    ctx.batch_run(semantics=sem, raw=True)
                    dt2 = time.perf_counter() - t0
                    rec["default_call"] = {"ms_per_batch": dt2 * 1e3, "early_settled": ctx.last_counters()["early_settled"],
                                           "note": "as the library runs it by default: reads whose background saturates are settled without a sweep (gcups above: option no_u8_early)"}
                out["%s_%dbp" % (name, read_len)] = rec
        return out
    finally:
        ctx.close()


def extra_repeat_rich(pgs, device, args, headline_gcups):
    """The headline workload on a REPEAT-RICH reference (synth.dna_repeats: four 300 bp families x 4000 copies at 3 %
    divergence, 500 microsatellite runs, 500 poly-A runs in a uniform 50 Mbp background) with 1 % of the reads cut from inside
    the repeats: what the candidate filters of the sampled sweep do on genome-like data.  Reports the rate, the fraction of
    reads that exceeded their candidate cap and were swept again exactly, and whether the whole batch had to be."""
    ref, planted = pgs.synth.dna_repeats(33, args.ref_len, families=4, family_len=300, copies=4000, divergence=0.03,
                                         tandem_runs=500, tandem_len=400, polya_runs=500, polya_len=300)
    reads, offs, which = pgs.synth.reads_with_repeats(ref, planted, 34, args.reads, args.read_len, repeat_fraction=0.01)
    ctx = pgs.Context(device)
    out = {"reads": args.reads, "reads_from_repeats": int(len(which)),
           "reference": "50 Mbp uniform background + 4 x 4000 copies of 300 bp families (3 % divergence) + 500 microsatellite runs + 500 poly-A runs"}
    try:
        ctx.set_reference(ref)
        ctx.batch_upload([r.tobytes() for r in reads])
        for sem, name in ((pgs.F32, "f32"), (pgs.U8SAT, "u8")):
            ctx.batch_run(semantics=sem, raw=True)
            t0 = time.perf_counter()
            res = ctx.batch_run(semantics=sem, raw=True)
            dt = time.perf_counter() - t0
            cnt = ctx.last_counters()
            tm = ctx.last_timings()
            g = float(args.reads) * args.read_len * args.ref_len / dt * 1e-9
            found = int(np.sum(np.abs(res["pos"] - (offs + 1)) <= 64))
            out[name] = {"gcups": g, "ms_per_step": dt * 1e3, "vs_headline": g / headline_gcups if headline_gcups else None,
                         "queries_swept_again_exactly": cnt["requeried"], "fraction_swept_again": cnt["requeried"] / float(args.reads),
                         "whole_batch_swept_again": cnt["whole_batch_again"], "candidate_subchunks": cnt["candidates"],
                         "score_kernel_ms": tm["score_us"] * 1e-3, "locate_ms": tm["locate_us"] * 1e-3, "traceback_ms": tm["trace_us"] * 1e-3,
                         "reads_found_at_their_origin": found, "kernel": ctx.last_kernel()["name"]}
            if sem == pgs.U8SAT:
                # the uint8 engine settles reads over their candidate cap by their first candidates in order: check the whole
                # batch once against the path that sweeps every such read a second time on the exact instances
                out[name]["settled_by_first_candidates"] = cnt["first_settled"]
                ctx.set_option("no_first")
                ref_res = ctx.batch_run(semantics=sem, raw=True)
                swept = ctx.last_counters()["requeried"]
                ctx.set_option("no_first", None)
                fields = ("score", "pos", "end_x", "end_y", "cons_len")
                out[name]["parity_vs_second_sweep"] = {"queries_swept_again_there": swept, "fields": list(fields),
                                                       "mismatches": int(sum(int(np.sum(res[f] != ref_res[f])) for f in fields))}
        return out
    finally:
        ctx.close()


def extra_config4(pgs, device, nseq):
    """configs[3] shape on one GPU: nseq UniProt-shaped protein sequences (first argument) against the 144-aa P02232
    query (second), identity scoring 3/-3, gap 2, float engine (src/mpi_sw_solve_uniprot.cpp:120)."""
    lens = pgs.synth.lognormal_lengths(5, nseq)
    tot = int(lens.sum())
    allres = pgs.synth.protein(5, tot)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cells = float(tot) * len(pgs.synth.P02232)
    ctx = pgs.Context(device)
    try:
        ctx.set_reference(pgs.synth.P02232)
        # the database as a multi-FASTA reader holds it: one buffer of residues + offsets (mi355_sw_batch_upload_packed)
        ctx.batch_upload_packed(allres, offs)                               # first call sizes the device buffers
        t0 = time.perf_counter()
        ctx.batch_upload_packed(allres, offs)
        up = time.perf_counter() - t0
        out = {"sequences": nseq, "cells": cells, "residues": tot, "upload_s": up,
               "upload_note": "mi355_sw_batch_upload_packed: one contiguous buffer + offsets, host buffer -> HBM incl. length sort"}
        for flags, name in ((pgs.capi.SCORE_ONLY, "score_argmax"), (0, "with_traceback")):
            ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)          # first call of a mode sizes its staging buffers
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            tm = ctx.last_timings()
            out[name] = {"wall_ms": best * 1e3, "device_ms": tm["total_us"] * 1e-3, "gcups_wall": cells / best * 1e-9,
                         "upload_plus_run_ms": (up + best) * 1e3, "gcups_end_to_end": cells / (up + best) * 1e-9}
        # in-run verification at full size: the device-built job lists against the host-built ones (option no_devlist: another
        # pipeline — windowed locate + traceback per alignment), every (score, pos, end_x, end_y, cons_len) of all nseq alignments
        a = ctx.batch_run(semantics=pgs.F32, flags=0, raw=True)
        ctx.set_option("no_devlist")
        try:
            b = ctx.batch_run(semantics=pgs.F32, flags=0, raw=True)
        finally:
            ctx.set_option("no_devlist", None)
        fields = ("score", "pos", "end_x", "end_y", "cons_len")
        bad = np.zeros(nseq, dtype=bool)
        for f in fields:
            bad |= np.asarray(a[f]) != np.asarray(b[f])
        out["parity_check"] = {"vs": "host-built job lists (option no_devlist)", "alignments": nseq, "fields": list(fields),
                               "mismatches": int(bad.sum())}
        out["rank_share"] = rank_share_config4(pgs, ctx, allres, offs, lens)
        return out
    finally:
        ctx.close()


def rank_share_config4(pgs, ctx, allres, offs, lens):
    """One rank's share of configs[3] at world = 1 / 2 / 4 / 8, measured on ONE GPU: the database LPT-partitioned by cell count
    (dist.shard_lpt, the query-sharding of src/mpi_sw_solve_uniprot.cpp:86-138 without its writer rank), the share of the rank
    with the greatest load — the job's critical path — uploaded as one packed buffer and run (score + argmax; with traceback),
    best of three.  predicted_speedup = T(world = 1) / T(share): what the sharded job reaches when every rank is as fast as this
    GPU and the 8-byte all-reduce of the best (score, index) key is free."""
    dist = importlib.import_module("parallel_genomeseq_amd.dist")
    out = {"note": "LPT shard (weights |x| * 144) of the rank with the greatest load; run = mi355_sw_batch_run_view on the resident shard, "
                   "best of five; *_ms = the C-ABI call, *_python_ms = through the ctypes binding (copies the result arrays)", "worlds": {}}
    base = None
    w = lens.astype(np.float64) * len(pgs.synth.P02232)
    for world in (1, 2, 4, 8):
        parts = dist.shard_lpt(w, world) if world > 1 else [np.arange(len(lens), dtype=np.int64)]
        loads = [float(w[p].sum()) for p in parts]
        mine = parts[int(np.argmax(loads))]
        sl = lens[mine]
        so = np.concatenate([[0], np.cumsum(sl)]).astype(np.int64)
        buf = np.empty(int(sl.sum()), dtype=np.uint8)
        for k, i in enumerate(mine):
            buf[so[k]:so[k + 1]] = allres[offs[i]:offs[i + 1]]
        ctx.batch_upload_packed(buf, so)
        rec = {"sequences": int(len(mine)), "cells": float(max(loads))}
        for flags, name in ((pgs.capi.SCORE_ONLY, "score_argmax"), (0, "with_traceback")):
            ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
            best = best_py = None
            for _ in range(5):
                t0 = time.perf_counter()
                ctx.batch_run(semantics=pgs.F32, flags=flags, raw=True)
                dt = time.perf_counter() - t0
                best_py = dt if best_py is None else min(best_py, dt)
                best = ctx.last_call_s if best is None else min(best, ctx.last_call_s)
            rec[name + "_ms"] = best * 1e3                              # mi355_sw_batch_run_view itself: what a C / C++ caller waits for
            rec[name + "_python_ms"] = best_py * 1e3                    # ... plus this binding's copies of the five result arrays
            rec[name + "_device_ms"] = ctx.last_timings()["total_us"] * 1e-3
        if base is None:
            base = rec
        rec["predicted_speedup_score_argmax"] = base["score_argmax_ms"] / rec["score_argmax_ms"]
        rec["predicted_speedup"] = base["with_traceback_ms"] / rec["with_traceback_ms"]
        rec["predicted_speedup_python"] = base["with_traceback_python_ms"] / rec["with_traceback_python_ms"]
        out["worlds"][str(world)] = rec
    return out


def config5_inputs(pgs, ref_len, qlen):
    """The 10 kbp query of configs[4]: a substring of the seed-6 reference with 1 % substitutions and 0.1 % indels."""
    r0 = int(pgs.synth.splitmix64(7, 1)[0] % np.uint64(max(1, ref_len - (qlen + 16))))
    window = pgs.synth.dna(6, qlen + 16, start=r0)
    # read_from_ref draws its offset from the same first output of SplitMix64(7): give it a stand-in of the right length
    class _Ref:
        def __len__(self): return ref_len
        def __getitem__(self, sl): return window[sl.start - r0: sl.stop - r0]
    q, off = pgs.synth.read_from_ref(_Ref(), 7, qlen, sub_rate=0.01, indel_rate=0.001)
    assert off == r0
    return q.tobytes(), off


def extra_config5(pgs, device, ref_len, qlen):
    """configs[4] shape on one GPU: one 10 kbp query against the whole 250 Mbp reference, end to end."""
    q, off = config5_inputs(pgs, ref_len, qlen)
    ref = pgs.synth.dna(6, ref_len)
    ctx = pgs.Context(device)
    try:
        ctx.set_reference(ref)
        ctx.batch_upload([q])
        out = {"ref_len": ref_len, "query_len": qlen, "planted_at": off + 1}
        for sem, name in ((pgs.F32, "f32"), (pgs.U8SAT, "u8"), (pgs.U8SAT, "u8_swept")):
            # "u8": as the library runs it (a 10 kbp read's background saturates within the reference's first sub-chunk: the uint8
            # engine's answer is decided there, no sweep); "u8_swept": option no_u8_early, the sweep of all 2.5e12 cells
            ctx.set_option("no_u8_early", True if name == "u8_swept" else None)
            ctx.batch_run(semantics=sem)
            t0 = time.perf_counter()
            r = ctx.batch_run(semantics=sem)[0]
            dt = time.perf_counter() - t0
            tm = ctx.last_timings()
            cells = float(qlen) * ref_len
            cnt = ctx.last_counters()
            out[name] = {"wall_ms": dt * 1e3, "gcups": cells / dt * 1e-9, "score_kernel_ms": tm["score_us"] * 1e-3,
                         "locate_ms": tm["locate_us"] * 1e-3, "traceback_ms": tm["trace_us"] * 1e-3,
                         "kernel": ctx.last_kernel()["name"], "score": r["score"], "pos": r["pos"],
                         "finish_from_saved_state": {k: cnt[k] for k in ("saved_locates", "saved_traces", "saved_fallbacks")},
                         "early_settled": cnt["early_settled"]}
        ctx.set_option("no_u8_early", None)
        out["rank_share"] = rank_share_config5(pgs, ctx, q, ref_len, qlen)
        # what a C++ user of the mirror gets: OMPParallelLocalAligner<...>(q, ref, 16, 2.0).calculateScore() = mi355_sw_align_split on
        # the caller's buffers (the reference stays resident between calls: 128-bit content hash, re-checked on helper threads)
        refb = ref.tobytes()
        ctx.align_split(q, refb, 16, 2.0, pgs.F32, pgs.F32)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            rs = ctx.align_split(q, refb, 16, 2.0, pgs.F32, pgs.F32)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        tm = ctx.last_timings()
        out["through_align_split"] = {"wall_ms": best * 1e3, "vs_whole_reference_call": best * 1e3 / out["f32"]["wall_ms"],
                                      "score_launches": tm["score_launches"], "score": rs["score"], "pos": rs["pos"], "piece": rs["piece"],
                                      "note": "mi355_sw_align_split, 16 pieces, overlap 2.0: winner-only sweep, the winner finished from its keys and the saved state"}
        del refb
        # in-run verification at full size: the 16-piece split (best_range + the winner finished from its keys) against the
        # whole-reference alignment — where the serial reference logic says they agree (SURVEY.md §8d cfg 5)
        rs = out["rank_share"]["worlds"]["8"]
        out["parity_check"] = {"vs": "16-piece split on the resident reference (best_range + align_scored_range)",
                               "fields": ["score", "pos"],
                               "mismatches": int(rs["pos"] != out["f32"]["pos"]) + int(rs["score"] != out["f32"]["score"])}
        return out
    finally:
        ctx.close()


def rank_share_config5(pgs, ctx, q, ref_len, qlen):
    """One rank's share of configs[4] at world = 1 / 2 / 4 / 8, measured on ONE GPU (the 8-GPU node is the driver's):
    npiece = 2 * world pieces of _make_string_range(npiece, |q|, |ref|, 2.0) (plocalaligner.cpp:44-67) dealt round-robin
    (piece p -> rank p mod world, plocalaligner.cpp:110-129).  Timed: the share of the rank on the job's critical path — the
    owner of the winning piece: it sweeps its pieces on the resident reference (mi355_sw_best_range: what every rank does, all
    shares are the same size; score_ms includes the exact re-evaluation of the winner's candidate sub-chunks, i.e. its argmax)
    and then finishes the winner (traceback, align_scored_range).
    predicted_speedup = T(world = 1) / T(share): what the sharded job reaches when every rank is as fast as this GPU and the
    8-byte all-reduce is free; predicted_speedup_score_pass: the same for the sweep alone."""
    out = {"note": "the pieces of the rank that owns the winning piece, swept on one GPU against the resident %d bp reference; "
                   "finish = argmax + traceback of the winning piece from the sweep's keys" % ref_len, "worlds": {}}
    base = None
    for world in (1, 2, 4, 8):
        npiece = 2 * world
        ranges = pgs.capi.make_string_range(npiece, qlen, ref_len, 2.0)
        fbest, fwhich, _ = ctx.best_range(ranges, semantics=pgs.F32)
        winner = int(fwhich[0])                                              # first piece with the greatest maximum
        owner = winner % world
        mine = ranges[owner::world]
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            bst, wh, _ = ctx.best_range(mine, semantics=pgs.F32)
            t1 = time.perf_counter()
            tm_score = ctx.last_timings()["score_us"]
            k = int(wh[0])
            r = ctx.align_scored_range(k, semantics=pgs.F32)
            t2 = time.perf_counter()
            tm = ctx.last_timings()
            cnt = ctx.last_counters()
            if best is None or (t2 - t0) < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1, tm_score, tm["locate_us"], tm["trace_us"], r,
                        {k: cnt[k] for k in ("saved_traces", "saved_fallbacks")})
        assert mine[k] == ranges[winner] and best[6]["score"] == float(fbest[0])
        rec = {"pieces": len(mine), "columns": int(sum(b - a for a, b in mine)), "share_ms": best[0] * 1e3,
               "score_ms": best[1] * 1e3, "score_kernel_ms": best[3] * 1e-3, "finish_ms": best[2] * 1e3,
               "finish_locate_ms": best[4] * 1e-3, "finish_traceback_ms": best[5] * 1e-3,
               "winning_piece": winner, "owner_rank": owner, "pos": best[6]["pos"] + ranges[winner][0], "score": best[6]["score"],
               "finish_from_saved_state": best[7]}
        if base is None:
            base = rec
        rec["predicted_speedup"] = base["share_ms"] / rec["share_ms"]
        rec["predicted_speedup_score_pass"] = base["score_ms"] / rec["score_ms"]
        rec["predicted_speedup_score_kernel"] = base["score_kernel_ms"] / rec["score_kernel_ms"]
        out["worlds"][str(world)] = rec
    ctx.best_range(ranges[:1], semantics=pgs.F32)
    out["kernel"] = ctx.last_kernel()["name"]
    return out


def extra_latency(pgs, device):
    """One-by-one calls, the unchanged-driver loop (src/sw_solve_big.cpp:78-92): one 150 bp read per mi355_sw_align."""
    ctx = pgs.Context(device)
    out = {}
    try:
        for n in (1_000_000, 50_000_000):
            refa = pgs.synth.dna(1, n)
            ref = refa.tobytes()
            reads = [pgs.synth.read_from_ref(refa, 2 + k, 150)[0].tobytes() for k in range(8)]
            for sem, name in ((pgs.F32, "f32"), (pgs.U8SAT, "u8")):
                ctx.align(reads[0], ref, sem)
                ctx.align(reads[1], ref, sem)
                t0 = time.perf_counter()
                for k in range(32):
                    ctx.align(reads[k % 8], ref, sem)
                dt = (time.perf_counter() - t0) / 32
                tm = ctx.last_timings()
                out["%s_150bp_x_%dMbp" % (name, n // 1_000_000)] = {"ms_per_align": dt * 1e3, "score_kernel_ms": tm["score_us"] * 1e-3,
                                                                   "gcups": 150.0 * n / dt * 1e-9}
                # the same read through OMPParallelLocalAligner as the reference's drivers construct it (sw_solve_big.cpp:78)
                ctx.align_split(reads[0], ref, 16, 2.0, sem, sem)
                t0 = time.perf_counter()
                for k in range(32):
                    ctx.align_split(reads[k % 8], ref, 16, 2.0, sem, sem)
                out["%s_150bp_x_%dMbp_split16" % (name, n // 1_000_000)] = {"ms_per_align": (time.perf_counter() - t0) / 32 * 1e3}
        return out
    finally:
        ctx.close()


def strong_config3(pgs, ctx, D, ref, args, sem):
    """Fixed total: --strong-reads reads of configs[2] split over the ranks (block partition), one pass."""
    total = args.strong_reads
    reads, _ = pgs.synth.fast_reads_from_ref(ref, 4242, total, args.read_len)     # (100 000 reads: the vectorised generator, substitutions only)
    base, rem = divmod(total, D.world)
    lo = D.rank * base + min(D.rank, rem)
    hi = lo + base + (1 if D.rank < rem else 0)
    ctx.batch_upload([r.tobytes() for r in reads[lo:hi]])

    def step():
        out = ctx.batch_run(semantics=sem, match=args.match, mismatch=args.mismatch, gap=args.gap, raw=True)
        best = 0
        if hi > lo:
            k = int(out["score"].argmax())
            best = int(np.float32(out["score"][k]).view(np.uint32)) << 32 | (0xFFFFFFFF - (lo + k))
        return D.max_key(best)
    dt, best = timed(D, step, 1)
    cells = float(total) * args.read_len * args.ref_len
    return {"workload": "%d x %d bp reads in total (block-partitioned over %d rank(s)) vs %d bp" % (total, args.read_len, D.world, args.ref_len),
            "s": dt, "gcups": cells / dt * 1e-9, "best_read": 0xFFFFFFFF - (best & 0xFFFFFFFF),
            "best_score": float(np.uint32(best >> 32).view(np.float32))}


def strong_config5(pgs, D, args, device):
    """configs[4]: the pieces of _make_string_range(npiece, |q|, |ref|, 2.0) dealt round-robin to the ranks
    (plocalaligner.cpp:110-129).  Every rank generates and holds ONLY its own pieces; per-piece maxima by the score
    kernel; one 8-byte all-reduce(MAX) of (score bits << 32 | ~piece) so that the lowest piece wins ties (:125);
    the owner re-aligns its piece with default scoring (:135) and the result is broadcast."""
    n, m, npiece = args.c5_ref_len, args.c5_query_len, args.c5_pieces
    q, off = config5_inputs(pgs, n, m)
    ranges = pgs.capi.make_string_range(npiece, m, n, 2.0)
    mine = list(range(D.rank, npiece, D.world))
    local, parts, at = [], [], 0
    for p in mine:
        l, r = ranges[p]
        parts.append(pgs.synth.dna(6, r - l, start=l))
        local.append((at, at + (r - l)))
        at += r - l
    ctx = pgs.Context(device)
    try:
        if mine:
            ctx.set_reference(np.concatenate(parts))
            ctx.batch_upload([q])
            ctx.best_range(local[:1], semantics=pgs.F32)                   # warm-up: scratch buffers, code objects

        def step():
            # winner-only sweep of this rank's pieces (mi355_sw_best_range): all the reduction needs is the rank's best piece.
            # The sweep's warm-up margin is optimistic: exact for maxima above `above` (the same value on every rank); when the
            # merged best does not exceed it, every rank sweeps again with the margin that best needs (known_best) — one more
            # 8-byte all-reduce decides, identically on all ranks.
            known = 0.0
            for _ in range(3):
                key, above = 0, -1.0
                if mine:
                    best, which, _, above = ctx.best_range(local, semantics=pgs.F32, known_best=known, want_exact_above=True)
                    key = int(np.float32(best[0]).view(np.uint32)) << 32 | (0xFFFFFFFF - mine[int(which[0])])
                key = D.max_key(key)
                above = D.max_float(above)
                gbest = float(np.uint32(key >> 32).view(np.float32))
                if gbest > above:
                    break
                known = max(gbest, 1.0)
            else:
                raise RuntimeError("config 5: merged best %r not above what the sweeps certify (%r) after three rounds" % (gbest, above))
            piece = 0xFFFFFFFF - (key & 0xFFFFFFFF)
            res = None
            if piece in mine:
                # the owner finishes its piece from the sweep's keys (default scoring in both roles: the per-piece sweep IS the
                # sweep of LAT(x, piece), plocalaligner.cpp:132-137) instead of sweeping the winning piece a second time
                r = ctx.align_scored_range(mine.index(piece), semantics=pgs.F32)
                res = {"score": r["score"], "pos": r["pos"] + ranges[piece][0], "piece": piece, "cons_len": len(r["cons_x"])}
            got = [g for g in D.gather_objects(res) if g is not None]
            return got[0]
        dt, res = timed(D, step, 1)
    finally:
        ctx.close()
    return {"workload": "1 x %d bp query vs %d bp reference in %d pieces (overlap 2.0) dealt to %d rank(s); each rank holds only its pieces"
                        % (m, n, npiece, D.world), "s": dt, "gcups": float(m) * n / dt * 1e-9, "winning_piece": res["piece"],
            "score": res["score"], "pos": res["pos"], "planted_at": off + 1}


def worker(args):
    D = Dist(args)
    rank, world, local_rank = D.rank, D.world, D.local_rank
    if "RANK" in os.environ and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    pgs = load_package()
    sem = pgs.F32 if args.semantics == "f32" else pgs.U8SAT
    ctx = pgs.Context(local_rank)
    ref = pgs.synth.dna(3, args.ref_len)                                   # seed 3 (SURVEY §8d cfg 3)
    # SURVEY §8(d)'s generator: substrings with 1 % substitutions AND 0.1 % single-base indels, so that the timed traceback emits gaps
    reads, _ = pgs.synth.reads_from_ref(ref, 4 + 7919 * rank, args.reads, args.read_len)
    ctx.set_reference(ref)
    ctx.batch_upload([r.tobytes() for r in reads])
    flags = pgs.capi.SCORE_ONLY if args.score_only else 0
    cells_per_step = float(args.reads) * args.read_len * args.ref_len
    group = D.describe()

    def step():
        out = ctx.batch_run(semantics=sem, match=args.match, mismatch=args.mismatch, gap=args.gap, flags=flags, raw=True)
        k = int(out["score"].argmax())
        best = int(np.float32(out["score"][k]).view(np.uint32)) << 32 | (0xFFFFFFFF - (k + rank * args.reads))
        return out, D.max_key(best)

    # PCIe-inclusive diagnostic (never the headline): host buffers handed over per call
    pcie = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        tp = time.perf_counter()
        ctx.set_reference(ref)
        ctx.batch_upload([r.tobytes() for r in reads])
        ctx.batch_run(semantics=sem, match=args.match, mismatch=args.mismatch, gap=args.gap, flags=flags, raw=True)
        pcie = time.perf_counter() - tp
    for _ in range(args.warmup):
        step()
    acc = {"kern_us": 0.0, "launches": 0, "locate_us": 0.0, "trace_us": 0.0}

    def timed_step():
        r = step()
        tm = ctx.last_timings()
        acc["kern_us"] += tm["score_us"]; acc["launches"] += tm["score_launches"]
        acc["locate_us"] += tm["locate_us"]; acc["trace_us"] += tm["trace_us"]
        return r
    dt, last = timed(D, timed_step, args.steps)
    ki = ctx.last_kernel()                                                  # the instance the library chose (not re-derived here)
    # in-run verification (outside the timed region): the same resident batch once more on float32 cells — another kernel
    # instance, another candidate path (option force_f32) — every (score, pos, end_x, end_y, cons_len) must agree
    parity = None
    if not args.no_parity:
        ctx.set_option("force_f32")
        try:
            chk = ctx.batch_run(semantics=sem, match=args.match, mismatch=args.mismatch, gap=args.gap, flags=flags, raw=True)
            kchk = ctx.last_kernel()["name"]
        finally:
            ctx.set_option("force_f32", None)
        fields = ("score", "pos", "end_x", "end_y", "cons_len")
        bad = np.zeros(args.reads, dtype=bool)
        for f in fields:
            bad |= np.asarray(last[0][f]) != np.asarray(chk[f])
        parity = {"vs": "the same resident batch on float32 cells (option force_f32): " + kchk, "reads": args.reads,
                  "fields": list(fields), "mismatches": int(D.max_float(float(bad.sum())))}

    def side(fn, *a, **kw):
        """A side measurement never takes the headline down with it (at N = 1; with several ranks a failure must fail the
        job, or the others would wait in a collective)."""
        if world > 1:
            return fn(*a, **kw)
        try:
            return fn(*a, **kw)
        except Exception as e:
            return {"failed": repr(e)[:300]}

    strong = None
    if not args.no_strong:
        strong = {"config3": side(strong_config3, pgs, ctx, D, ref, args, sem)}
        ctx.batch_upload([r.tobytes() for r in reads])                      # the weak-scaling batch again
    extras = None
    if world == 1 and not args.no_extras:
        extras = {}
        extras["u8_engine"] = side(engine_rate, pgs, ctx, D, args.reads, args.read_len, args.ref_len, 2, semantics=pgs.U8SAT)
        extras["f32_cells_fractional_scoring"] = side(engine_rate, pgs, ctx, D, args.reads, args.read_len, args.ref_len, 2,
                                                      semantics=pgs.F32, match=3.5, mismatch=-3.25, gap=2.0)
        extras["f32_cells_fractional_scoring"]["scoring"] = "3.5 / -3.25 / 2"
    ctx.close()
    del ref
    if strong is not None and not args.no_config5:
        strong["config5"] = side(strong_config5, pgs, D, args, local_rank)
    if extras is not None:
        extras["read_lengths"] = side(extra_read_lengths, pgs, local_rank, args.ref_len)
        extras["repeat_rich"] = side(extra_repeat_rich, pgs, local_rank, args, cells_per_step * args.steps / dt * 1e-9)
        extras["one_by_one_calls"] = side(extra_latency, pgs, local_rank)
        extras["config4_uniprot_shape"] = side(extra_config4, pgs, local_rank, args.c4_sequences)
        if not args.no_config5:
            extras["config5_whole_reference"] = side(extra_config5, pgs, local_rank, args.c5_ref_len, args.c5_query_len)

    if rank == 0:
        total_cells = cells_per_step * args.steps * world
        gcups = total_cells / dt * 1e-9
        avg_launch_s = acc["kern_us"] / max(1, acc["launches"]) * 1e-6
        alg_bytes = float(args.reads) * (args.read_len + args.ref_len + 16)   # |x|+|y|+16 per alignment
        achieved = alg_bytes / avg_launch_s * 1e-9
        traffic, traffic_source = None, None
        if world == 1 and not args.no_traffic:
            ctx_closed_note = None
            traffic, traffic_source = measure_traffic(args)
            if traffic is None:
                ctx_closed_note = traffic_source
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if (rec.get("reads") == args.reads and rec.get("ref_len") == args.ref_len and rec.get("semantics") == args.semantics
                        and rec.get("kernel", ki["name"]) == ki["name"]):
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_source = ("replay:profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command on this kernel; not measured in "
                                      "this run%s)" % ("" if args.no_traffic or world > 1 else ": " + str(ctx_closed_note)))
            except Exception:
                traffic = None
        kern_cells_per_s = cells_per_step / avg_launch_s
        opc = ki["valu_ops_per_cell"]
        line = {
            "metric": "GCUPS (cell updates/s), 150 bp reads vs 50 Mbp reference, whole job (score + argmax + traceback)",
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ki["dtype"] if sem == pgs.F32 else "u8",
            "dtype_note": "f16 = packed 2x float16 cells holding H / 2048 (every H an integer within +-2048: exact); i16 = packed 2x16-bit integer cells; both exact for the float32 engine's integer scores; u8 = the uint8 engine's score pass on the same packed float16 cells, unsaturated, published maxima clamped at 255 (locate / traceback evaluate the saturating rule exactly); f32 = float32 cells (fractional scoring); the instance is the library's choice, reported by mi355_sw_last_kernel",
            "data": "synthetic",
            "config": {"workload": "configs[2]: %d x %d bp reads per GPU per step vs %d bp reference (one batch of the 100k-read set)"
                                   % (args.reads, args.read_len, args.ref_len),
                       "semantics": "Similarity_Matrix (float32)" if sem == pgs.F32 else "Similarity_Matrix_Skewed (uint8 saturating)",
                       "reads_per_gpu_per_step": args.reads, "read_len": args.read_len, "ref_len": args.ref_len,
                       "parallelism": "reads sharded x%d, reference replicated" % world, "score_only": bool(args.score_only)},
            "process_group": group,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": ki["name"],
                         "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                         "tile": {"chunk_len": ki["chunk_len"], "sub_len": ki["sub_len"], "warm": ki["warm"]},
                         "note": "scalar recurrence: VALU-bound, not HBM-bound (DESIGN.md §3.4); see valu"},
            "valu": {"kernel_gcups": kern_cells_per_s * 1e-9, "lane_ops_per_cell": opc,
                     "achieved_lane_ops_per_s": kern_cells_per_s * opc,
                     "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                     "frac": kern_cells_per_s * opc / VALU_PEAK_LANE_OPS,
                     "frac_vop3p_issue": kern_cells_per_s * opc / (VALU_PEAK_LANE_OPS / 2.0),
                     "note": "frac: against 2 cycles per wave64 instruction (what plain 32-bit VOP2 adds reach); frac_vop3p_issue: against 4 "
                             "cycles per wave64 instruction, the issue rate of the packed 16-bit (VOP3P) ops, DPP moves and VOP3 ops this "
                             "kernel consists of (profiles/r01_valu_instruction_rates.txt) — its own ceiling"},
            "lds": lds_model(ki, kern_cells_per_s),
            "phases_ms_per_step": {"score_kernel": acc["kern_us"] / args.steps * 1e-3, "locate": acc["locate_us"] / args.steps * 1e-3,
                                   "traceback": acc["trace_us"] / args.steps * 1e-3},
        }
        if parity is not None:
            line["parity_check"] = parity
        if strong is not None:
            line["strong_scaling"] = strong
        if extras is not None:
            line["extras"] = extras
        if pcie is not None:
            line["pcie_inclusive"] = {"gcups": cells_per_step / pcie * 1e-9, "s_per_step": pcie,
                                      "note": "set_reference + batch_upload from host buffers + batch_run, one cold step"}
        if world == 1 and not args.no_cpu_baseline:          # rank 0 at N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(pgs, pgs.synth.dna(3, min(args.ref_len, 5_000_000)), args.read_len)
            except Exception as e:  # the baseline is reported, never required
                line["cpu_baseline"] = {"value": None, "unit": "GCUPS", "cores": 0, "kind": "reference", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    D.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=2048, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--ref-len", type=int, default=50_000_000)
    ap.add_argument("--semantics", choices=["f32", "u8"], default="f32")
    ap.add_argument("--score-only", action="store_true", help="skip the traceback (diagnostic; not the headline)")
    ap.add_argument("--match", type=float, default=3.0)
    ap.add_argument("--mismatch", type=float, default=-3.0)
    ap.add_argument("--gap", type=float, default=2.0, help="non-integer scoring selects the float32-cell kernel instance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the N = 1 side measurements (other engines, configs 4 / 5, one-by-one calls)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling passes")
    ap.add_argument("--no-config5", action="store_true", help="skip everything that needs the 250 Mbp reference")
    ap.add_argument("--strong-reads", type=int, default=100_000,
                    help="total reads of the strong-scaling pass (all ranks together): all of configs[2] by default")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure HBM traffic with rocprofv3 child passes (replay profiles/pmc_traffic.json)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity checks against other kernel instances")
    ap.add_argument("--c4-sequences", type=int, default=561_356)
    ap.add_argument("--c5-ref-len", type=int, default=250_000_000)
    ap.add_argument("--c5-query-len", type=int, default=10_000)
    ap.add_argument("--c5-pieces", type=int, default=16)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) | gloo (rehearsal: ranks may share GPU 0)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        launch(args, sys.argv[1:])
    worker(args)


if __name__ == "__main__":
    main()
