cd /tmp && export TMPDIR=/tmp
MSDA_BWD_MODE=split rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python3 $GRAFT_REPO_ROOT/tools/ktime.py cfg4_encoder cfg2_encoder cfg4_decoder > /tmp/s.log 2>&1
grep -v amdgpu /tmp/s.log
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_s/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if 'msda' in r['Name']: print("%6s calls %9.1f us avg  %s"%(r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
