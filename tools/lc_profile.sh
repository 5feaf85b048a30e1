export TMPDIR=/tmp
python bench.py --model lightcnn --steps 20 --no-sustained --no-cpu-baseline --timeline-json gpurun_out/tl_lcnn.json > /dev/null 2>&1
python - <<'PY'
import json
j=json.load(open('gpurun_out/tl_lcnn.json'))
print({k:v for k,v in j.items() if k!='streams'})
for s in j['streams']: print(s)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lc -o lc -- python bench.py --model lightcnn --steps 5 --warmup 2 --no-cpu-baseline --serial --no-sustained --no-profile > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_lc/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/7)
for r in rows[:16]:
    print('%-58s calls/step %6.1f ms/step %7.3f avg_us %8.1f' % (r['Name'].replace('(anonymous namespace)::','')[:58], int(r['Calls'])/7, float(r['TotalDurationNs'])/1e6/7, float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/prof_lc
