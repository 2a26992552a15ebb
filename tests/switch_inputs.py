"""Inputs that reach the alternates behind the A/B switches of the library, shared by tests/test_gpu_round4.py (which asserts
that a switch ENGAGED: mi355_sw_last_path changes as expected) and tools/path_probe.py (which prints the paths)."""
import numpy as np

SWITCHES = ["no_f16", "no_unsat", "no_sample", "no_satflag", "no_solo", "no_wave", "no_comb", "no_twin", "no_wide",
            "no_strip", "no_quant", "no_devlist", "no_strip_groups", "u8_long_twin", "long_twin", "no_long",
            "no_requery", "slot=16", "no_long+strip_r=24", "u8_sample_short", "few_r=5", "long_pipes=2", "no_long_p32", "long_groups=2", "force_f32", "no_opt_margin",
            "no_wave_prof", "no_wave_window", "no_first", "no_long_save", "assume_cus=32", "no_wave_pieces", "no_u8_early", "no_wave_f16", "no_devlist_by_id"]

_cache = {}


def _ref(pgs, seed, n):
    key = (seed, n)
    if key not in _cache:
        _cache[key] = pgs.synth.dna(seed, n)
    return _cache[key]


def _reads(pgs, ref, seed, count, m):
    return [pgs.synth.read_from_ref(ref, seed + k, m, sub_rate=0.02, indel_rate=0.004)[0].tobytes() for k in range(count)]


def build(pgs, name):
    """(kind, queries, reference bytes, semantics, extra) of input `name`."""
    if name in ("batch150_f32", "batch150_u8"):
        ref = _ref(pgs, 9101, 200_000)
        return "batch", _reads(pgs, ref, 9200, 32, 150), ref.tobytes(), 0 if name.endswith("f32") else 1
    if name in ("single150_f32", "single150_u8"):
        ref = _ref(pgs, 9101, 200_000)
        return "single", _reads(pgs, ref, 9300, 1, 150), ref.tobytes(), 0 if name.endswith("f32") else 1
    if name == "batch600_f32":
        ref = _ref(pgs, 9101, 200_000)
        return "batch", _reads(pgs, ref, 9400, 8, 600), ref.tobytes(), 0
    if name in ("batch1000_f32", "batch1000_u8"):
        ref = _ref(pgs, 9101, 200_000)
        return "batch", _reads(pgs, ref, 9500, 8, 1000), ref.tobytes(), 0 if name.endswith("f32") else 1
    if name in ("single1000_f32", "single1000_u8"):
        ref = _ref(pgs, 9101, 200_000)
        return "single", _reads(pgs, ref, 9600, 1, 1000), ref.tobytes(), 0 if name.endswith("f32") else 1
    if name == "single400_long_ref":
        ref = _ref(pgs, 9102, 20_000_000)
        return "single", _reads(pgs, ref, 9650, 1, 400), ref.tobytes(), 0
    if name in ("single3000_f32", "single3000_u8"):
        ref = _ref(pgs, 9103, 300_000)
        return "single", _reads(pgs, ref, 9700, 1, 3000), ref.tobytes(), 0 if name.endswith("f32") else 1
    if name == "single5000_f32":
        ref = _ref(pgs, 9103, 300_000)
        return "single", _reads(pgs, ref, 9750, 1, 5000), ref.tobytes(), 0
    if name == "uniprot_shape":
        lens = pgs.synth.lognormal_lengths(5, 300)
        res = pgs.synth.protein(5, int(lens.sum()))
        offs = np.concatenate([[0], np.cumsum(lens)])
        seqs = [res[offs[k]:offs[k + 1]].tobytes() for k in range(len(lens))]
        seqs[7] = seqs[7][:40] + pgs.synth.P02232[10:120].encode() + seqs[7][40:]          # a real hit: a walk with a window
        long = pgs.synth.protein(77, 6000).tobytes()                                       # long streams: swept in pieces
        seqs[3] = long[:3100] + pgs.synth.P02232[5:140].encode() + long[3100:]             # ... a hit deep inside one
        seqs[4] = pgs.synth.P02232[:100].encode() + long[:2500]                            # ... and at the very start of another
        seqs[5] = long[:1030] + pgs.synth.P02232[20:144].encode() + long[1030:1700]        # ... and across a piece border
        return "batch", seqs, pgs.synth.P02232.encode(), 0
    if name in ("polya_f32", "polya_u8"):
        ref = bytearray(_ref(pgs, 9104, 300_000).tobytes())
        for at in range(5_000, 290_000, 9_000):
            ref[at:at + 700] = b"A" * 700
        reads = [b"A" * 150] * 2 + _reads(pgs, _ref(pgs, 9104, 300_000), 9800, 30, 150)
        return "batch", reads, bytes(ref), 0 if name.endswith("f32") else 1
    raise KeyError(name)


INPUTS = ["batch150_f32", "batch150_u8", "single150_f32", "single150_u8", "batch600_f32", "batch1000_f32", "batch1000_u8",
          "single1000_f32", "single1000_u8", "single400_long_ref", "single3000_f32", "single3000_u8", "single5000_f32", "uniprot_shape", "polya_f32", "polya_u8"]


def run_input(pgs, ctx, name, want_results=False):
    """Runs input `name` on `ctx`; returns the path tags of the call (and the results when asked)."""
    kind, qs, ref, sem = build(pgs, name)
    if kind == "single":
        res = [ctx.align(qs[0], ref, sem)]
    else:
        res = ctx.align_batch(qs, ref, semantics=sem)
    path = ctx.last_path()
    return (path, res) if want_results else path
