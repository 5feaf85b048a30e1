set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_sec
for m in lightcnn resnet50_128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sec/$m -o $m -- python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_sec/bench_$m.json 2> gpurun_out/prof_sec/$m.err
  rm -f gpurun_out/prof_sec/$m/*_kernel_trace.csv
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_sec/$m/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('$m total kernel ms', tot/1e6)
for r in rows[:22]:
    print('%-60s calls %5s tot_ms %8.2f avg_us %8.1f %5s%%' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
