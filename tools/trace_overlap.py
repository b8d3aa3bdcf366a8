"""rocprofv3 --kernel-trace CSV of tools/two_stream_trace.py -> the dispatches of the step's
kernel with their start / end stamps and how much of the time two of them were in flight.
  python tools/trace_overlap.py <dir with *kernel_trace.csv> <out.csv> [kernel substring]"""
import csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else 'SampleFanout'
files = glob.glob(src + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        if pat in name:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                         r.get('Queue_Id', ''), r.get('Stream_Id', ''), r.get('Dispatch_Id', ''), name[:60]))
rows.sort()
half = rows[len(rows) // 2:]                 # the second pass of the loop
t0 = half[0][0]
with open(out, 'w') as fo:
    fo.write('dispatch_id,queue_id,stream_id,start_us,end_us,duration_us,overlaps_previous_us,kernel\n')
    prev_end = None
    for s, e, q, st, d, n in half:
        ov = max(0, min(prev_end, e) - s) if prev_end is not None else 0
        fo.write('%s,%s,%s,%.2f,%.2f,%.2f,%.2f,%s\n' % (d, q, st, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3,
                                                     ov / 1e3, n))
        prev_end = max(prev_end, e) if prev_end is not None else e
# time with >= 2 dispatches in flight
ev = sorted([(s, 1) for s, e, *_ in half] + [(e, -1) for s, e, *_ in half])
depth, last, busy1, busy2, busy3 = 0, ev[0][0], 0, 0, 0
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    if depth >= 3: busy3 += t - last
    depth += d; last = t
span = half[-1][1] - half[0][0]
print(json.dumps({'dispatches': len(half), 'span_us': span / 1e3, 'us_per_dispatch_by_span': span / 1e3 / len(half),
                  'mean_duration_us': sum(e - s for s, e, *_ in half) / len(half) / 1e3,
                  'frac_time_two_in_flight': busy2 / max(1, busy1), 'frac_time_three_in_flight': busy3 / max(1, busy1), 'frac_time_busy': busy1 / max(1, span)}))
