#!/usr/bin/env python3
"""Multi-GPU drivers for the two sharding modes of SURVEY.md §8(e), one process per GPU:

  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/run_sharded.py queries [--n 4096]
      config 3/4 shape: reads sharded over ranks (LPT by cell count), reference replicated, results all-gathered,
      ONE 8-byte all-reduce for the batch best (replaces the MPI_Send/MPI_Recv task farm of
      src/mpi_sw_solve_uniprot.cpp:95-181).
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/run_sharded.py split [--m 10000 --n 50000000]
      config 5 shape: pieces of _make_string_range dealt to ranks, per-piece maxima on each rank's GPU, packed-key
      all-reduce (lowest piece wins ties, plocalaligner.cpp:125), owner re-aligns and broadcasts.

--backend gloo lets several ranks share GPU 0 (rehearsal on a one-GPU box); default nccl (= RCCL) needs one GPU per rank.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["queries", "split"])
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--m", type=int, default=10_000)
    ap.add_argument("--reads", type=int, default=4096)
    ap.add_argument("--npiece", type=int, default=16)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    pgs = g._load_package()
    from parallel_genomeseq_amd import dist as pd
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = local % max(1, torch.cuda.device_count()) if a.backend == "gloo" else local
    torch.cuda.set_device(dev)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    ctx = pgs.Context(dev)
    if a.mode == "queries":
        n = a.n or 10_000_000
        ref = pgs.synth.dna(3, n)
        reads, offs = pgs.synth.fast_reads_from_ref(ref, 4, a.reads, 150)
        lens = np.full(a.reads, 150)
        ctx.set_reference(ref)
        t0 = time.time()
        idx, res, gathered = pd.align_queries_sharded(lambda qs: ctx.align_batch(qs, semantics=pgs.F32),
                                                      [r.tobytes() for r in reads], weights=lens * n)
        # a rank with an empty shard (fewer reads than ranks) still joins the all-reduce, with the lowest key
        best = pd.allreduce_best(*((max(r["score"] for r in res), int(idx[int(np.argmax([r["score"] for r in res]))]))
                                   if res else (0.0, 0xFFFFFFFF)))
        dt = time.time() - t0
        ok = float((gathered["end_y"] == offs + 150).mean())
        if rank == 0:
            print("queries: %d reads x 150 bp vs %d bp on %d rank(s): %.2f s, %.1f GCUPS, best (score, read) = %s, end_y==cut+150 for %.3f"
                  % (a.reads, n, world, dt, a.reads * 150 * n / dt * 1e-9, best, ok), flush=True)
        assert ok > 0.97
    else:
        n = a.n or 50_000_000
        ref = pgs.synth.dna(6, n)
        q, off = pgs.synth.read_from_ref(ref, 7, a.m, sub_rate=0.01, indel_rate=0.001)
        qb, refb = q.tobytes(), ref.tobytes()
        ranges = pgs.capi.make_string_range(a.npiece, a.m, n, 2.0)
        ctx.set_reference(refb)
        ctx.batch_upload([qb])
        t0 = time.time()
        res, piece = pd.align_split_sharded(
            ranges,
            lambda pieces: ctx.score_ranges([ranges[p] for p in pieces], semantics=pgs.F32)[:, 0],
            lambda p: ctx.align(qb, refb[ranges[p][0]:ranges[p][1]], pgs.F32))
        dt = time.time() - t0
        if rank == 0:
            print("split: %d bp query vs %d bp in %d pieces on %d rank(s): %.2f s, winner piece %d, score %g, pos %d (planted at %d)"
                  % (a.m, n, a.npiece, world, dt, piece, res["score"], res["pos"], off + 1), flush=True)
        assert abs(res["pos"] - (off + 1)) < 300
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
