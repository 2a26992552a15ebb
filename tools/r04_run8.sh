set -x
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_fifth.json 2> gpurun_out/r04/c4_fifth.err; echo "c4 rc=$?"
python - <<'PYEOF'
import json
d=json.load(open('gpurun_out/r04/c4_fifth.json'))
for w,r in d['rank_share']['worlds'].items(): print(w, {a:(round(b,3) if isinstance(b,float) else b) for a,b in r.items()})
PYEOF
timeout -k 10 200 python tools/c4_share_trace.py 8 1 > gpurun_out/r04/c4_trace8_score2.log 2>&1; tail -9 gpurun_out/r04/c4_trace8_score2.log
timeout -k 10 700 bash tools/collect_profiles_r04.sh > gpurun_out/r04/collect.log 2>&1; echo "collect rc=$?"; tail -5 gpurun_out/r04/collect.log
