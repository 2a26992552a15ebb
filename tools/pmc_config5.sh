#!/usr/bin/env bash
# PMC counters of config 5's strip-mined score instance (10 kbp x 250 Mbp), separate passes
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c5
mkdir -p $OUT
n=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES"; do
  n=$((n+1))
  rocprofv3 --pmc $set -d $OUT/p$n --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_config5.py 250000000 10000 16 > $OUT/p$n.log 2>&1 || echo "pass $n failed"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections,json
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob('gpurun_out/pmc_c5/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sw_score_kernel' not in r['Kernel_Name']: continue
        k=r['Kernel_Name'][:70]
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in tot:
    per={c: tot[k][c]/len(cnt[k][c]) for c in tot[k]}
    d={}
    if 'GRBM_GUI_ACTIVE' in per and 'SQ_INSTS_VALU' in per:
        simd=per['GRBM_GUI_ACTIVE']/8*1024
        d['valu_busy_frac']=4*per['SQ_INSTS_VALU']/simd
    if 'SQ_LDS_IDX_ACTIVE' in per: d['lds_conflict_frac']=per['SQ_LDS_BANK_CONFLICT']/per['SQ_LDS_IDX_ACTIVE']
    if 'SQ_WAVE_CYCLES' in per: d['wait_any_frac']=per['SQ_WAIT_ANY']/per['SQ_WAVE_CYCLES']; d['wait_inst_any_frac']=per['SQ_WAIT_INST_ANY']/per['SQ_WAVE_CYCLES']; d['wait_inst_lds_frac']=per['SQ_WAIT_INST_LDS']/per['SQ_WAVE_CYCLES']; d['active_valu_frac']=per['SQ_ACTIVE_INST_VALU']/per['SQ_WAVE_CYCLES']
    print(json.dumps({'kernel':k,'launches':{c:len(cnt[k][c]) for c in cnt[k]},'per_launch':per,'derived':d},indent=1))
PY
