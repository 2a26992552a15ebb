"""One 150 bp (or argv[3] bp) read per call against an n-bp reference under a few option sets: ms per call, score-kernel ms, kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pgs = g._load_package()
n = int(sys.argv[1]); sem = int(sys.argv[2]); m = int(sys.argv[3]) if len(sys.argv) > 3 else 150
refa = pgs.synth.dna(1, n); ref = refa.tobytes()
reads = [pgs.synth.read_from_ref(refa, 2 + k, m)[0].tobytes() for k in range(8)]
variants = [{}, {"slot": 16}, {"no_twin": 1}, {"no_twin": 1, "slot": 16}, {"chunk": 1024}, {"chunk": 1024, "slot": 16}, {"chunk": 1536, "slot": 16},
            {"chunk": 3072, "slot": 16}, {"no_comb": 1}, {"no_solo": 1}]
for var in variants:
    ctx = pgs.Context(0)
    for k, v in var.items():
        ctx.set_option(k, v)
    for k in range(4): ctx.align(reads[k], ref, sem)
    t0 = time.perf_counter()
    sk = 0.0
    for k in range(64):
        ctx.align(reads[k % 8], ref, sem)
        sk += ctx.last_timings()["score_us"]
    dt = (time.perf_counter() - t0) / 64 * 1e3
    print("%-34s %.3f ms per call, score kernel %.3f ms, %s" % (var, dt, sk / 64e3, ctx.last_kernel()["name"]), file=sys.stderr)
    ctx.close()
