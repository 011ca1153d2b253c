cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt_steps; mkdir -p $R/gpurun_out/kt_steps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_steps -o p -- python3 $R/bench.py --headline-only --no-cpu-baseline --no-live-traffic --steps 200 --warmup 20 --long-steps 200 > $R/gpurun_out/kt_steps/log.txt 2>&1
python3 $R/tools/_trace_analyze.py $R/gpurun_out/kt_steps/p_kernel_trace.csv; exit 0
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/kt_steps/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for i, r in enumerate(rows):
    name = r['Kernel_Name'][:60]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    gap = int(r['Start_Timestamp']) - int(rows[i-1]['End_Timestamp']) if i else 0
    by[(name, r['Grid_Size'])].append((d, gap))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 5: continue
    v2 = v[len(v)//2:]   # the later half: steady state
    ds = sorted(x[0] for x in v2); gs = sorted(x[1] for x in v2)
    print(f"{len(v):6d} x {k[0]:60s} grid {k[1]:>8s}: dur median {ds[len(ds)//2]/1e3:7.2f} us  p10 {ds[len(ds)//10]/1e3:7.2f}  p90 {ds[9*len(ds)//10]/1e3:7.2f}   gap before: median {gs[len(gs)//2]/1e3:6.2f} us")
PY
