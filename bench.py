#!/usr/bin/env python3
"""bench.py — headline benchmark of the Smith-Waterman hot path on MI355X.

Metric (BASELINE.json): GCUPS = sum |x_k|*|y| / t * 1e-9 for 150 bp synthetic reads against a 50 Mbp
synthetic reference (configs[2], "100k x 150 bp vs 50 Mbp"), whole job: score pass + argmax +
traceback + results on the host, inputs resident in HBM when the timed region starts.

One "step" = one pass of the hot path over one batch of --reads reads per GPU (default 2048; the
full 100 000-read set is 49 such batches — 7.5e14 cells — and the per-batch rate is size-normalised).
Multi-GPU: one process per GPU (torch.distributed / RCCL), reads sharded across ranks, reference
replicated, no data-path collective; one 8-byte all-reduce per step merges the per-rank best
(score, read) — weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (packed op = 1 lane-op)


def kernel_shape(read_len):
    """(SL, R) the library picks for this read length (mi355_sw.hip pick_shape) and the VALU instructions per
    cell of that sw_score_kernel instance: per step and lane 4R (5R uint8) recurrence ops +
    R/2 max-fold + DPP/mask/address/extract, for 2R cells."""
    if read_len > 2048:     # whole-wavefront tiles in strips: the rows per lane with the fewest padded rows
        best = None
        for r in (20, 24, 32):
            rows = -(-read_len // (64 * r)) * 64 * r
            if best is None or rows <= best[0]:
                best = (rows, r)
        return 64, best[1]
    if read_len > 512:      # one whole-wavefront strip
        return 64, next(r for r in (10, 12, 16, 20, 24, 32) if 64 * r >= read_len)
    r16 = next(r for r in (2, 4, 6, 8, 10, 12, 16, 20, 24, 32) if r >= (read_len + 15) // 16)
    r8 = next((r for r in (7, 10, 13, 16, 19, 26, 32) if r >= (read_len + 7) // 8), 0) if read_len >= 36 else 0
    if r8 and 8 * r8 <= 16 * r16:
        return 8, r8
    return 16, r16


def ops_per_cell(sl, r, sem, f16=False):
    """VALU instructions per cell of the instance that runs: packed ops per step and lane
    plus DPP / border mask / profile address, for 2R cells."""
    over = 4 if sl == 8 else 3
    if sem == "f32cells":  # float32 cells, one query per slot: add(clamp), max3, sub per cell; max3 per two cells
        return (3 * r + (r + 1) // 2 + 1 + over) / float(r)
    if f16 and sem == 0:   # float engine, float16 cells: add(clamp), maximum3, add(-g) per cell; one maximum3 per two
        core = 3 * r + (r + 1) // 2 + 1 + over          # cells for the running maximum; one add for the row above
    elif f16:              # uint8 engine, float16 cells: add(clamp), max, add(-g), maximum3; one maximum3 per two odd rows
        odd = r // 2
        core = 4 * r + odd // 2 + odd % 2 + (r % 2) + over
    else:
        core = (5 if sem == 1 else 4) * r + (r + 1) // 2 + (r % 2) + over
    return core / (2.0 * r)


def uses_f16(args):
    """The library's choice (host_score.h make_buckets): float32 engine, integer scores, every value within the
    exactly representable float16 integers."""
    if os.environ.get("MI355_SW_NO_F16") is not None or args.read_len > 2048:
        return False
    if args.semantics == "u8":       # values never leave 0..255: held as (H + 1) / 256 in float16
        return args.reads >= 2
    ints = all(float(v) == int(v) for v in (args.match, args.mismatch, args.gap))
    return (args.semantics == "f32" and ints and args.gap >= 1 and abs(args.mismatch) <= 2048 and
            args.match * (args.read_len + 1) <= 2040 and args.gap <= 2040 and args.read_len <= 512 and
            os.environ.get("MI355_SW_NO_F16") is None)


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
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) | gloo (rehearsal: ranks may share GPU 0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:          # under torch.distributed.run even one rank joins a group
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    pgs = load_package()
    sem = pgs.F32 if args.semantics == "f32" else pgs.U8SAT
    ctx = pgs.Context(local_rank)
    ref = pgs.synth.dna(3, args.ref_len)                                   # seed 3 (SURVEY §8d cfg 3)
    reads, _ = pgs.synth.fast_reads_from_ref(ref, 4 + 7919 * rank, args.reads, args.read_len)
    ctx.set_reference(ref)
    ctx.batch_upload([r.tobytes() for r in reads])
    flags = pgs.capi.SCORE_ONLY if args.score_only else 0
    cells_per_step = float(args.reads) * args.read_len * args.ref_len

    def step():
        out = ctx.batch_run(semantics=sem, match=args.match, mismatch=args.mismatch, gap=args.gap, flags=flags, raw=True)
        best = int(out["score"].max()) << 32 | (0xFFFFFFFF - (int(out["score"].argmax()) + rank * args.reads))
        if dist is not None:
            t = torch.tensor([best], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # per-rank best (score, read) over xGMI
            best = int(t.item())
        return out, best

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
    kern_us, kern_launches, locate_us, trace_us = 0.0, 0, 0.0, 0.0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, best = step()
        tm = ctx.last_timings()
        kern_us += tm["score_us"]; kern_launches += tm["score_launches"]
        locate_us += tm["locate_us"]; trace_us += tm["trace_us"]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kshape = kernel_shape(args.read_len)
    if rank == 0:
        total_cells = cells_per_step * args.steps * world
        gcups = total_cells / dt * 1e-9
        avg_launch_s = kern_us / max(1, kern_launches) * 1e-6
        alg_bytes = float(args.reads) * (args.read_len + args.ref_len + 16)   # |x|+|y|+16 per alignment
        achieved = alg_bytes / avg_launch_s * 1e-9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("reads") == args.reads and rec.get("ref_len") == args.ref_len and rec.get("semantics") == args.semantics:
                    traffic = rec["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        kern_cells_per_s = cells_per_step / avg_launch_s
        f16 = uses_f16(args)
        fractional = any(float(v) != int(v) for v in (args.match, args.mismatch, args.gap))
        opc = ops_per_cell(kshape[0], kshape[1], "f32cells" if (sem == pgs.F32 and fractional) else sem, f16)
        line = {
            "metric": "GCUPS (cell updates/s), 150 bp reads vs 50 Mbp reference, whole job (score + argmax + traceback)",
            "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if (args.match != int(args.match) or args.mismatch != int(args.mismatch) or args.gap != int(args.gap)) else ("f16" if f16 else "i16")) if sem == pgs.F32 else "u8",
            "dtype_note": "f16 = packed 2x float16 cells holding H / 2048 (every H an integer within +-2048: exact); i16 = packed 2x16-bit integer cells; both exact for the float32 engine's integer scores; u8 = saturating uint8 semantics held in packed float16 lanes as (H + 1) / 256 (exact), in packed 16-bit integer lanes for queries beyond 512 rows; f32 = float32 cells (fractional scoring)",
            "data": "synthetic",
            "config": {"workload": "configs[2]: %d x %d bp reads per GPU per step vs %d bp reference (one batch of the 100k-read set)"
                                   % (args.reads, args.read_len, args.ref_len),
                       "semantics": "Similarity_Matrix (float32)" if sem == pgs.F32 else "Similarity_Matrix_Skewed (uint8 saturating)",
                       "reads_per_gpu_per_step": args.reads, "read_len": args.read_len, "ref_len": args.ref_len,
                       "parallelism": "reads sharded x%d, reference replicated" % world, "score_only": bool(args.score_only)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "sw_score_kernel<R=%d, SL=%d>" % (kshape[1], kshape[0]),
                         "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "scalar recurrence: VALU-bound, not HBM-bound (DESIGN.md §5); see valu"},
            "valu": {"kernel_gcups": kern_cells_per_s * 1e-9, "lane_ops_per_cell": opc,
                     "achieved_lane_ops_per_s": kern_cells_per_s * opc,
                     "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                     "frac": kern_cells_per_s * opc / VALU_PEAK_LANE_OPS,
                     "note": "peak = 2 cycles per wave64 instruction; packed 16-bit (VOP3P) ops issue at 4 (profiles/r01_valu_instruction_rates.txt)"},
            "phases_ms_per_step": {"score_kernel": kern_us / args.steps * 1e-3, "locate": locate_us / args.steps * 1e-3,
                                   "traceback": trace_us / args.steps * 1e-3},
        }
        if pcie is not None:
            line["pcie_inclusive"] = {"gcups": cells_per_step / pcie * 1e-9, "s_per_step": pcie,
                                      "note": "set_reference + batch_upload from host buffers + batch_run, one cold step"}
        if world == 1 and not args.no_cpu_baseline:          # rank 0 at N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(pgs, ref, args.read_len)
            except Exception as e:  # the baseline is reported, never required
                line["cpu_baseline"] = {"value": None, "unit": "GCUPS", "cores": 0, "kind": "reference", "sample": "failed: %r" % (e,)}
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
