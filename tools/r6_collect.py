"""gpurun_out/<tag>_* (tools/round4_profile.sh) -> profiles/<tag>_{kernel_stats.csv,pmc_summary.json,
bench.json} and profiles/pmc_latest.json (what bench.py quotes as roofline.traffic and
roofline.read_lines_per_s).  Round 6: the one-stream steps of tools/fl_one.py launch
SampleFanoutPlainKernel (fanout_plain.h); the record names whichever one-kernel build the
counter passes saw.
  python tools/r6_collect.py <tag>"""
import glob, json, os, shutil, sys
tag = sys.argv[1]
os.makedirs('profiles', exist_ok=True)
for f in glob.glob('gpurun_out/%s_trace/**/*kernel_stats.csv' % tag, recursive=True):
    shutil.copy(f, 'profiles/%s_kernel_stats.csv' % tag)
for wl in ('deepwalk', 'hetero'):
    for f in glob.glob('gpurun_out/%s_%s_trace/**/*kernel_stats.csv' % (tag, wl), recursive=True):
        # keep the library's kernels and the few torch ones; drop the graph generator's one-off launches
        rows = [l for l in open(f) if 'Synth' not in l and 'BuildBlocks' not in l and 'BuildPivot' not in l]
        open('profiles/%s_%s_kernel_stats.csv' % (tag, wl), 'w').writelines(rows)
p = 'gpurun_out/pmc_%s.json' % tag
if os.path.exists(p):
    doc = json.load(open(p))
    json.dump(doc, open('profiles/%s_pmc_summary.json' % tag, 'w'), indent=1)
    for k, c in doc.items():
        name = 'SampleFanoutPlainKernel' if 'SampleFanoutPlainKernel' in k else \
            'SampleFanoutLeanKernel' if 'SampleFanoutLeanKernel' in k else None
        if name is not None and 'TCC_EA0_RDREQ_sum' in c:
            wr = c.get('WRITE_SIZE')            # KiB
            rd_bytes = c['TCC_EA0_RDREQ_sum'] * 128.0
            wr_bytes = wr * 1024.0 if wr is not None else None
            latest = {'kernel': name, 'kernel_build': k, 'batch': 131072, 'nodes': 100000000,
                      'read_requests': c['TCC_EA0_RDREQ_sum'], 'read_requests_per_launch': c['TCC_EA0_RDREQ_sum'],
                      'read_bytes': rd_bytes,
                      'write_bytes': wr_bytes,
                      'hbm_bytes_per_launch': rd_bytes + (wr_bytes or 0.0),
                      'mean_us_under_pmc': c.get('mean_us_under_pmc'),
                      'source': 'profiles/%s_pmc_summary.json: TCC_EA0_RDREQ x 128 B (a read request '
                                'moves a 128-byte line) + WRITE_SIZE KiB x 1024' % tag}
            json.dump(latest, open('profiles/pmc_latest.json', 'w'), indent=1)
            print(json.dumps(latest))
b = 'gpurun_out/%s_bench.json' % tag
if os.path.exists(b) and os.path.getsize(b):
    line = open(b).read().strip().splitlines()[-1]
    json.dump(json.loads(line), open('profiles/%s_bench.json' % tag, 'w'), indent=1)
for nm in ('two_stream_trace.csv', 'two_stream_summary.json'):
    f = 'gpurun_out/%s_%s' % (tag, nm)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, 'profiles/%s_%s' % (tag, nm))
for f in glob.glob('gpurun_out/%s_2s/**/*kernel_stats.csv' % tag, recursive=True):
    shutil.copy(f, 'profiles/%s_two_stream_kernel_stats.csv' % tag)
