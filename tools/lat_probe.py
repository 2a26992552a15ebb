import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pgs = g._load_package()
ctx = pgs.Context(0)
n = int(sys.argv[1]); sem = int(sys.argv[2])
refa = pgs.synth.dna(1, n); ref = refa.tobytes()
reads = [pgs.synth.read_from_ref(refa, 2 + k, 150)[0].tobytes() for k in range(8)]
for k in range(4): ctx.align(reads[k], ref, sem)
t0=time.perf_counter()
for k in range(64): ctx.align(reads[k%8], ref, sem)
print("ms per align", (time.perf_counter()-t0)/64*1e3, file=sys.stderr)
