cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/sh_prof -o sh -- python $R/tools/sharded_one.py 12 1 2>&1 < /dev/null | grep ms_per_step; cd $R
t=$(find gpurun_out/sh_prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && timeout 60 python - "$t" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'Synth' not in r['Kernel_Name'] and 'Build' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)//4
rows=rows[-n:]            # the last of four identical passes
t0=int(rows[0]['Start_Timestamp']); t1=int(rows[-1]['End_Timestamp'])
agg=collections.defaultdict(lambda:[0,0])
for r in rows:
    k=r['Kernel_Name'][:80]; agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
busy=sum(v[1] for v in agg.values())
print('kernels', len(rows), 'window us per step', (t1-t0)/1e3/12, 'kernel time per step us (sum)', busy/1e3/12)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(round(v[1]/1e3/12,1), 'us/step', v[0]/12, 'calls/step', k)
PY
rm -rf gpurun_out/sh_prof
