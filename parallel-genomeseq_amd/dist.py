"""Multi-GPU sharding of the alignment path: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

* query sharding  — the shape of src/mpi_sw_solve_uniprot.cpp:95-138 (independent alignments farmed to
  ranks) without its MPI_Send/MPI_Recv writer rank: reads are block- or LPT-partitioned, the reference is
  replicated, there is no exchange during compute; results stay rank-local or are all-gathered (16 B per
  alignment); "best over the batch" is ONE 8-byte all-reduce(MAX) of a packed (score, ~index) key.
* reference sharding — the shape of OMPParallelLocalAligner (src/aligner/plocalaligner.cpp:105-143):
  pieces from _make_string_range dealt round-robin to ranks, per-piece maxima computed locally, ONE
  all-reduce(MAX) of (score << 32 | ~piece) so the lowest piece index wins ties (serial rule, :125), the
  owner rank re-aligns the winning piece with default scoring (:135) and broadcasts the result.

The functions take the compute step as a callable so that CPU tests can drive them with the oracle;
the product callables are methods of capi.Context (GPU only).
"""
import numpy as np
import torch
import torch.distributed as dist


def _dev():
    """Where the (tiny) collective payloads live: the rank's GPU under RCCL, the host otherwise (gloo, or no group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_block(n, rank, size):
    """Contiguous block of range(n) for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_lpt(weights, size):
    """Longest-processing-time partition of items by weight (cell count |x|*|y|) — for skewed length
    distributions such as UniProt.  Deterministic; returns one index array per rank."""
    w = np.asarray(weights, dtype=np.float64)
    order = np.argsort(-w, kind="stable")
    load = np.zeros(size)
    bins = [[] for _ in range(size)]
    for i in order:
        r = int(np.argmin(load))
        bins[r].append(int(i))
        load[r] += w[i]
    return [np.array(sorted(b), dtype=np.int64) for b in bins]


def pack_key(score, index):
    """(score, index) -> int64 so that MAX picks the highest score, then the LOWEST index.
    The score travels as its float32 bit pattern (non-negative floats order like their bits — the device keys of
    sw_score_kernel do the same), so fractional scorings (12.5 against 12.0) keep their order."""
    s = np.float32(score)
    if not s >= 0:
        raise ValueError("scores are maxima of cells >= 0")
    return (int(s.view(np.uint32)) << 32) | (0xFFFFFFFF - int(index))


BAD_KEY = (0x7FC00000 << 32) | 0xFFFFFFFF      # above every valid key (float32 NaN pattern): "some rank saw a bad score"


def unpack_key(key):
    return float(np.uint32(key >> 32).view(np.float32)), 0xFFFFFFFF - (key & 0xFFFFFFFF)


def allreduce_best(score, index):
    """One 8-byte all-reduce(MAX): global best (score, index), ties to the lowest index."""
    rank, size = world()
    key = pack_key(score, index)
    if size > 1:
        t = torch.tensor([key], dtype=torch.int64, device=_dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        key = int(t.item())
    return unpack_key(key)


def align_queries_sharded(align_fn, queries, weights=None, gather=True):
    """Each rank aligns its shard with `align_fn(list_of_queries) -> list of result dicts`.
    Returns (local_indices, local_results, gathered) where gathered is None or a dict of full-length
    arrays score/pos/end_x/end_y identical on every rank."""
    rank, size = world()
    n = len(queries)
    # every rank derives EVERY rank's shard (both partitions are deterministic): the gather below needs no index exchange
    if weights is None:
        parts = [np.arange(*shard_block(n, r, size), dtype=np.int64) for r in range(size)]
    else:
        parts = shard_lpt(weights, size)
    idx = parts[rank]
    res = align_fn([queries[i] for i in idx]) if len(idx) else []
    gathered = None
    if gather:
        # ONE all_gather of 16 B per alignment: (score as float32 bits, pos, end_x, end_y) as four int32 —
        # pos is the reference's unsigned int, end_x / end_y are below 2^31 for every reference this engine holds; shards are
        # padded to the longest one (all_gather wants equal shapes), the padding is dropped on arrival
        longest = max(len(p) for p in parts) if parts else 0
        rec = np.zeros((max(1, longest), 4), dtype=np.int32)
        if len(idx):
            rec[:len(idx), 0] = np.array([r["score"] for r in res], dtype=np.float32).view(np.int32)
            rec[:len(idx), 1] = np.array([r["pos"] for r in res], dtype=np.int64).astype(np.uint32).view(np.int32)
            rec[:len(idx), 2] = np.array([r["end_x"] for r in res], dtype=np.int64)
            rec[:len(idx), 3] = np.array([r["end_y"] for r in res], dtype=np.int64)
        mine = torch.from_numpy(rec).to(_dev())
        if size > 1:
            every = [torch.empty_like(mine) for _ in range(size)]
            dist.all_gather(every, mine)
        else:
            every = [mine]
        score = np.zeros(n, dtype=np.float32)
        pos = np.zeros(n, dtype=np.int64)
        end_x = np.zeros(n, dtype=np.int64)
        end_y = np.zeros(n, dtype=np.int64)
        for r in range(size):
            k = len(parts[r])
            if k == 0:
                continue
            a = every[r].cpu().numpy()[:k]
            score[parts[r]] = a[:, 0].copy().view(np.float32)
            pos[parts[r]] = a[:, 1].copy().view(np.uint32)
            end_x[parts[r]] = a[:, 2]
            end_y[parts[r]] = a[:, 3]
        gathered = dict(score=score, pos=pos, end_x=end_x, end_y=end_y)
    return idx, res, gathered


def align_split_sharded(ranges, piece_maxima_fn, final_align_fn):
    """Reference sharding.  `ranges` = [(left, right)] from _make_string_range (identical on all ranks);
    `piece_maxima_fn(list_of_piece_indices) -> maxima` sweeps this rank's pieces;
    `final_align_fn(piece_index) -> result dict` re-aligns one piece (default scoring) and is run by the
    owner rank only.  Returns (result dict with pos already shifted by left, winning piece)."""
    rank, size = world()
    mine = list(range(rank, len(ranges), size))
    maxima = piece_maxima_fn(mine) if mine else []
    key = 0                                                   # "no piece": below every packed key (index < 2^32 - 1)
    # serial rule (plocalaligner.cpp:122-129): max_score_l starts at -1, strict '>' -> first piece with the max
    # a negative / NaN maximum on ONE rank must not leave the others waiting in the collective: it travels as a sentinel key
    # above every valid one, and every rank raises after the all-reduce
    for p, v in zip(mine, maxima):
        key = max(key, pack_key(v, p) if np.float32(v) >= 0 else BAD_KEY)
    if size > 1:
        t = torch.tensor([key], dtype=torch.int64, device=_dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        key = int(t.item())
    if key == BAD_KEY:
        raise ValueError("a rank reported a negative or NaN piece maximum (scores are maxima of cells >= 0)")
    _, piece = unpack_key(key)
    owner = piece % size
    res = None
    if rank == owner:
        res = dict(final_align_fn(piece))
        left = ranges[piece][0]
        res["pos"] = res["pos"] + left
        if res.get("end_y", 0) > 0:
            res["end_y"] = res["end_y"] + left
        res["piece"] = piece
        # every rank returns the same keys: what travels in the broadcast (the owner's device timings stay with the owner)
        res = {k: res.get(k, 0) for k in ("score", "pos", "end_x", "end_y", "cons_x", "cons_y", "piece")}
    if size > 1:
        res = _broadcast_result(res, owner)
    return res, piece


def align_split_sharded_certified(ranges, best_fn, final_align_fn, max_rounds=3):
    """Reference sharding on winner-only sweeps with an optimistic warm-up margin (mi355_sw_best_range).
    `best_fn(list_of_piece_indices, known_best) -> (best_score, position_in_that_list, exact_above)` sweeps this rank's pieces:
    best_score is a lower bound of the rank's true best, exact when it exceeds exact_above (the same value on every rank,
    -1 = everything exact).  The packed (best, ~piece) keys are merged by one 8-byte all-reduce(MAX); when the merged best does
    not exceed the merged exact_above, every rank sweeps again with known_best = the merged best, whose margin makes the second
    merge exact (plocalaligner.cpp:122-129: lowest piece wins ties).  Then as align_split_sharded: the owner finishes its
    piece and the result is broadcast.  Returns (result dict, winning piece, rounds of sweeping)."""
    rank, size = world()
    mine = list(range(rank, len(ranges), size))
    known = 0.0
    piece = 0
    rounds = 0
    for rounds in range(1, max_rounds + 1):
        key, above = 0, -1.0
        if mine:
            best, at, above = best_fn(mine, known)
            key = pack_key(best, mine[int(at)]) if np.float32(best) >= 0 else BAD_KEY
        if size > 1:
            t = torch.tensor([key, int(np.float32(above).view(np.int32))], dtype=torch.int64, device=_dev())
            # exact_above is -1 or a non-negative float: as a signed integer its bit pattern orders the same way
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            key = int(t[0].item())
            above = float(np.int32(int(t[1].item())).view(np.float32))
        if key == BAD_KEY:
            raise ValueError("a rank reported a negative or NaN piece maximum (scores are maxima of cells >= 0)")
        gbest, piece = unpack_key(key)
        if gbest > above:
            break
        known = max(gbest, 1.0)
    else:
        # (every rank sees the same merged values, so every rank raises)
        raise RuntimeError("reference sharding: the merged best %r is still not above what the sweeps certify (%r) after %d rounds"
                           % (gbest, above, max_rounds))
    owner = piece % size
    res = None
    if rank == owner:
        res = dict(final_align_fn(piece))
        left = ranges[piece][0]
        res["pos"] = res["pos"] + left
        if res.get("end_y", 0) > 0:
            res["end_y"] = res["end_y"] + left
        res["piece"] = piece
        res = {k: res.get(k, 0) for k in ("score", "pos", "end_x", "end_y", "cons_x", "cons_y", "piece")}
    if size > 1:
        res = _broadcast_result(res, owner)
    return res, piece, rounds


def _broadcast_result(res, src):
    """The owner's result to every rank as two tensor broadcasts (a fixed 6-word header, then the consensus bytes) —
    no pickling, and on RCCL the payload goes GPU to GPU."""
    dev = _dev()
    head = torch.zeros(6, dtype=torch.int64, device=dev)
    if res is not None:
        cx, cy = res["cons_x"].encode("latin-1"), res["cons_y"].encode("latin-1")
        head = torch.tensor([int(np.float32(res["score"]).view(np.uint32)), res["pos"], res.get("end_x", 0), res.get("end_y", 0),
                             len(cx), res["piece"]], dtype=torch.int64, device=dev)
    dist.broadcast(head, src=src)
    h = head.cpu().tolist()
    n = int(h[4])
    body = torch.zeros(max(1, 2 * n), dtype=torch.uint8, device=dev)
    if res is not None and n:
        body = torch.frombuffer(bytearray(cx + cy), dtype=torch.uint8).to(dev)
    dist.broadcast(body, src=src)
    if res is not None:
        return res
    b = bytes(body.cpu().numpy().tobytes())
    return dict(score=float(np.uint32(h[0]).view(np.float32)), pos=int(h[1]), end_x=int(h[2]), end_y=int(h[3]),
                cons_x=b[:n].decode("latin-1"), cons_y=b[n:2 * n].decode("latin-1"), piece=int(h[5]))
