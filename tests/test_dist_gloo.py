"""world_size-2 gloo tests of the multi-GPU sharding logic on CPU.  The compute
callables are oracle-backed here (tests may use the oracle); on the GPU box the same functions are
driven by capi.Context methods."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, size, port, fn_name, ret):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    pgs = load_package()
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        ret[rank] = globals()[fn_name](rank, size, pgs, ob)
    finally:
        dist.destroy_process_group()


def _run(fn_name, size=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(size, port, fn_name, ret), nprocs=size, join=True)
    return dict(ret)


def _case_query_sharding(rank, size, pgs, ob):
    from parallel_genomeseq_amd import dist as pd
    ref = pgs.synth.dna(5, 20000)
    reads = [pgs.synth.read_from_ref(ref, 50 + k, [40, 80, 125, 150][k % 4])[0].tobytes() for k in range(11)]
    refb = ref.tobytes()
    calls = []

    def align_fn(qs):
        calls.append(len(qs))
        return [ob.align(q, refb, ob.F32) for q in qs]

    idx, res, g = pd.align_queries_sharded(align_fn, reads, weights=[len(r) * len(refb) for r in reads])
    best = pd.allreduce_best(max(r["score"] for r in res), int(idx[int(np.argmax([r["score"] for r in res]))]))
    return dict(idx=idx.tolist(), score=g["score"].tolist(), pos=g["pos"].tolist(), end_y=g["end_y"].tolist(), best=best,
                expect=[ob.align(q, refb, ob.F32) for q in reads])


def test_query_sharding_gloo():
    out = _run("_case_query_sharding")
    a, b = out[0], out[1]
    assert sorted(a["idx"] + b["idx"]) == list(range(11)) and not set(a["idx"]) & set(b["idx"])
    for o in (a, b):       # every rank holds the full gathered arrays
        assert o["score"] == [e["score"] for e in o["expect"]]
        assert o["pos"] == [e["pos"] for e in o["expect"]]
        assert o["end_y"] == [e["end_y"] for e in o["expect"]]
    scores = [e["score"] for e in a["expect"]]
    assert a["best"] == b["best"] == (int(max(scores)), int(np.argmax(scores)))   # ties -> lowest index


def _case_ref_sharding(rank, size, pgs, ob):
    from parallel_genomeseq_amd import dist as pd
    ref = pgs.synth.dna(6, 30000)
    q = ref[7000:7100].tobytes()
    ref = ref.tobytes()
    ref = ref[:21000] + q + ref[21000 + len(q):]          # second identical hit in a later piece
    out = {}
    for sm, la, npiece in ((ob.F32, ob.F32, 5), (ob.U8SAT, ob.U8SAT, 7)):
        ranges = ob.make_string_range(npiece, len(q), len(ref), 2.0)

        def maxima_fn(pieces):
            return [ob.score_only(q, ref[ranges[p][0]:ranges[p][1]], sm) for p in pieces]

        def final_fn(piece):
            l, r = ranges[piece]
            return ob.align(q, ref[l:r], la)

        res, piece = pd.align_split_sharded(ranges, maxima_fn, final_fn)
        exp = ob.align_split(q, ref, npiece, 2.0, sm, la)
        out[npiece] = (res, piece, exp)
    return out


def test_ref_sharding_gloo():
    out = _run("_case_ref_sharding")
    for rank in (0, 1):
        for npiece, (res, piece, exp) in out[rank].items():
            assert piece == exp["piece"]
            for k in ("score", "pos", "cons_x", "cons_y"):
                assert res[k] == exp[k], (rank, npiece, k)


def test_partitions():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    load_package()
    from parallel_genomeseq_amd import dist as pd
    assert [pd.shard_block(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    bins = pd.shard_lpt([9, 1, 1, 1, 8, 2, 2, 3], 2)
    assert sorted(np.concatenate(bins).tolist()) == list(range(8))
    loads = [sum([9, 1, 1, 1, 8, 2, 2, 3][i] for i in b) for b in bins]
    assert abs(loads[0] - loads[1]) <= 1
    assert pd.unpack_key(max(pd.pack_key(30, 5), pd.pack_key(30, 2), pd.pack_key(29, 0))) == (30, 2)
