"""Deterministic synthetic inputs: SplitMix64-seeded DNA / protein sequences
and reads in the shapes the reference's drivers consume (src/sw_solve_big.cpp:30-37,69:
single-line reference, reads = CSV column 2; py/ompfg_data_prep.py:92-116)."""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, count, start=0):
    """Outputs start+1 .. start+count of SplitMix64(seed) as a uint64 array (vectorised:
    the k-th output depends only on seed + k*gamma)."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _chunks(fn, starts):
    """Independent chunks of a long sequence on a few threads (numpy releases the GIL inside its loops)."""
    starts = list(starts)
    if len(starts) < 4:
        for s in starts:
            fn(s)
        return
    import os
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))
    with ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(fn, starts))


_DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
# 20 amino acids, rough UniProt background frequencies (per mille)
_AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
_AA_W = np.array([83, 14, 55, 68, 39, 71, 23, 59, 58, 97, 24, 41, 47, 39, 55, 66, 54, 69, 11, 29], dtype=np.float64)


# public UniProt entry P02232 (LGB1_VICFA), the reference's data/query/P02232.fasta (144 aa), re-typed: the query
# of the UniProt-shaped batch (src/mpi_sw_solve_uniprot.cpp:52-63,120)
P02232 = ("MGFTEKQEALVNSSSQLFKQNPSNYSVLFYTIILQKAPTAKAMFSFLKDSAGVVDSPKLGAHAEKVFGMVRDSAVQLRATGEVVLDGKDGSIHIQKGVLDPHFVVVKEALLKTIKEASGD"
          "KWSEELSAAWEVAYDGLATAIKAA")


def dna(seed, length, chunk=1 << 22, start=0):
    """Uniform ACGT: base = "ACGT"[x >> 62].  `start`: positions [start, start + length) of the same stream, so that
    a rank can generate only its own piece of a sharded reference."""
    out = np.empty(length, dtype=np.uint8)

    def part(s):
        n = min(chunk, length - s)
        out[s:s + n] = _DNA[(splitmix64(seed, n, start + s) >> np.uint64(62)).astype(np.intp)]
    _chunks(part, range(0, length, chunk))
    return out


def protein(seed, length, chunk=1 << 22):
    cdf = np.cumsum(_AA_W) / _AA_W.sum()
    out = np.empty(length, dtype=np.uint8)

    def part(s):                                              # chunked: 200 M residues would take GBs of temporaries
        n = min(chunk, length - s)
        u = (splitmix64(seed, n, s) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        out[s:s + n] = _AA[np.minimum(np.searchsorted(cdf, u, side="right"), 19)]
    _chunks(part, range(0, length, chunk))
    return out


def read_from_ref(ref, seed, length, sub_rate=0.01, indel_rate=0.001):
    """One read: substring of `ref` at a PRNG offset with substitutions and single-base indels
    so the traceback sees gaps.  Returns (read uint8 array, 0-based offset)."""
    r = splitmix64(seed, 4 * length + 8)
    n = len(ref)
    span = length + 16
    off = int(r[0] % np.uint64(max(1, n - span)))
    src = ref[off:off + span]
    out = []
    k = 0
    i = 0
    while len(out) < length and i < len(src):
        u = float(r[1 + k] >> np.uint64(11)) / float(1 << 53)
        v = int(r[2 + k] >> np.uint64(62))
        k += 2
        if u < indel_rate / 2:            # deletion from the read
            i += 1
            continue
        if u < indel_rate:                # insertion into the read
            out.append(int(_DNA[v]))
            continue
        b = int(src[i])
        if u < indel_rate + sub_rate:
            b = int(_DNA[(int(np.searchsorted(_DNA, b)) + 1 + v % 3) % 4]) if b in _DNA else int(_DNA[v])
        out.append(b)
        i += 1
    while len(out) < length:
        out.append(int(_DNA[0]))
    return np.array(out[:length], dtype=np.uint8), off


def reads_from_ref(ref, seed, count, length, sub_rate=0.01, indel_rate=0.001):
    """`count` reads; read k uses seed (seed + 0x1000003 * k)."""
    reads = np.empty((count, length), dtype=np.uint8)
    offs = np.empty(count, dtype=np.int64)
    for k in range(count):
        reads[k], offs[k] = read_from_ref(ref, seed + 0x1000003 * k, length, sub_rate, indel_rate)
    return reads, offs


def fast_reads_from_ref(ref, seed, count, length, sub_rate=0.01):
    """Vectorised variant for large batches (substitutions only)."""
    n = len(ref)
    offs = (splitmix64(seed, count) % np.uint64(max(1, n - length))).astype(np.int64)
    idx = offs[:, None] + np.arange(length, dtype=np.int64)[None, :]
    reads = ref[idx]
    r = splitmix64(seed ^ 0x5DEECE66D, count * length).reshape(count, length)
    u = (r >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    sub = u < sub_rate
    repl = _DNA[(r & np.uint64(3)).astype(np.intp)]
    return np.where(sub, repl, reads).astype(np.uint8), offs


def make_lut(seed, scale=1.0):
    """256x256 float32 scoring table: diagonal in [1,6]*scale, off-diagonal in [-5,5]*scale.
    The same construction is used by oracle/ref_driver.cpp's `alignlut` so that fixtures made
    with the real reference can be replayed."""
    r = splitmix64(seed, 65536).reshape(256, 256)
    off = (r % np.uint64(11)).astype(np.int64) - 5
    dia = 1 + (r % np.uint64(6)).astype(np.int64)
    lut = np.where(np.eye(256, dtype=bool), dia, off).astype(np.float32) * np.float32(scale)
    return np.ascontiguousarray(lut)


def lognormal_lengths(seed, count, median=290.0, sigma=0.66, lo=2, hi=35000):
    """UniProt-like sequence length distribution (SURVEY.md §8d cfg 4)."""
    r = splitmix64(seed, 2 * count)
    u1 = ((r[:count] >> np.uint64(11)).astype(np.float64) + 1.0) / float((1 << 53) + 1)
    u2 = (r[count:] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return np.clip(np.rint(median * np.exp(sigma * z)), lo, hi).astype(np.int64)


def dna_repeats(seed, length, families=4, family_len=300, copies=2000, divergence=0.03, tandem_runs=200, tandem_len=400,
                polya_runs=200, polya_len=300):
    """A repeat-rich reference (what genomes look like and uniform ACGT does not): a uniform background with planted
    * interspersed repeat families: `families` consensus sequences of `family_len` bp, `copies` copies each at PRNG
      positions, every copy with `divergence` substitutions (Alu-like: thousands of near-identical 300 bp elements);
    * tandem repeats: `tandem_runs` microsatellite runs of `tandem_len` bp with a 2-6 bp unit;
    * poly-A runs: `polya_runs` runs of `polya_len` bp.
    Returns (reference uint8 array, dict with the planted intervals per class: 'family' [(start, family id)],
    'tandem' [start], 'polya' [start]).  Deterministic in `seed`."""
    ref = dna(seed, length)
    r = splitmix64(seed ^ 0xA5A5A5A5, 4 * (families * copies + tandem_runs + polya_runs) + 64)
    k = 0
    planted = {"family": [], "tandem": [], "polya": []}
    cons = [dna(seed + 1000 + f, family_len) for f in range(families)]
    for f in range(families):
        for _ in range(copies):
            at = int(r[k] % np.uint64(max(1, length - family_len))); k += 1
            copy = cons[f].copy()
            rr = splitmix64(int(r[k] & np.uint64(0x7FFFFFFF)), 2 * family_len); k += 1
            sub = (rr[:family_len] >> np.uint64(11)).astype(np.float64) / float(1 << 53) < divergence
            copy[sub] = _DNA[(rr[family_len:][sub] >> np.uint64(62)).astype(np.intp)]
            ref[at:at + family_len] = copy
            planted["family"].append((at, f))
    for _ in range(tandem_runs):
        at = int(r[k] % np.uint64(max(1, length - tandem_len))); k += 1
        ulen = 2 + int(r[k] % np.uint64(5)); k += 1
        unit = _DNA[(splitmix64(int(r[k] & np.uint64(0x7FFFFFFF)), ulen) >> np.uint64(62)).astype(np.intp)]; k += 1
        ref[at:at + tandem_len] = np.tile(unit, tandem_len // ulen + 1)[:tandem_len]
        planted["tandem"].append(at)
    for _ in range(polya_runs):
        at = int(r[k] % np.uint64(max(1, length - polya_len))); k += 1
        ref[at:at + polya_len] = ord("A")
        planted["polya"].append(at)
    return ref, planted


def reads_with_repeats(ref, planted, seed, count, length, repeat_fraction=0.01, sub_rate=0.01):
    """`count` reads of `length` bp: fast_reads_from_ref of the reference, with round(repeat_fraction * count) of them cut
    from inside planted repeats instead (family copies, tandem runs and poly-A runs in turn).  Returns (reads, offsets,
    indices of the repeat-derived reads)."""
    reads, offs = fast_reads_from_ref(ref, seed, count, length, sub_rate)
    nrep = int(round(repeat_fraction * count))
    if nrep == 0:
        return reads, offs, np.zeros(0, dtype=np.int64)
    r = splitmix64(seed ^ 0x0F0F0F0F, 3 * nrep + 8)
    which = (r[:nrep] % np.uint64(count)).astype(np.int64)
    which = np.unique(which)
    classes = [c for c in ("family", "tandem", "polya") if planted[c]]
    for j, idx in enumerate(which):
        cls = classes[j % len(classes)]
        item = planted[cls][int(r[nrep + j] % np.uint64(len(planted[cls])))]
        at = int(item[0] if cls == "family" else item)
        at = max(0, min(len(ref) - length, at + int(r[2 * nrep + j] % np.uint64(32))))
        reads[idx] = ref[at:at + length]
        offs[idx] = at
    return reads, offs, which
