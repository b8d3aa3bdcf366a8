"""rocprofv3 --kernel-trace CSV of tools/walk_one.py -> the timeline of the LAST DeepWalk call:
every dispatch with its start (us from the call's first kernel), duration and the gap since
the previous dispatch ended.   python tools/walk_gaps.py <dir with *kernel_trace.csv> [out.txt]"""
import csv, glob, sys
src = sys.argv[1]
rows = []
for f in glob.glob(src + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Kernel_Name', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if 'CwInitKernel' in r[2]]
first = starts[-1]
while first > 0 and rows[first][0] - rows[first - 1][1] < 50_000 and 'Cw' not in rows[first - 1][2] \
        and 'Transpose' not in rows[first - 1][2]:
    first -= 1                                   # the memsets / fills in front of the call
t0 = rows[first][0]
out = []
prev_end = None
busy = 0
for s, e, n in rows[first:]:
    short = n.split('(')[0].replace('euler_gpu::', '').replace('void ', '')[:48]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    out.append('%9.2f %8.2f %8.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, short))
    busy += e - s
    prev_end = e
span = (rows[-1][1] - t0) / 1e3
out.append('span %.2f us, kernels %.2f us, gaps %.2f us over %d dispatches' % (span, busy / 1e3, span - busy / 1e3,
                                                                              len(rows) - first))
txt = '   start    dur_us   gap_us  kernel\n' + '\n'.join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + '\n')
