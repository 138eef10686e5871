"""Per-kernel totals from a rocprofv3 --kernel-trace CSV (runs on the GPU box; the raw trace is too large to bring back).
usage: python tools/trace_stats.py <dir with *_kernel_trace.csv> [top=40] [last_ms: only launches that start in the last so many ms]"""
import sys, os, csv, glob, re
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
t_end = max(int(r['End_Timestamp']) for r in rows) if rows else 0
tot = {}
if True:
    if True:
        for row in rows:
            if last_ms > 0 and int(row['Start_Timestamp']) < t_end - last_ms * 1e6:
                continue
            name = row['Kernel_Name']
            dur = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
            grid = row.get('Grid_Size', '')
            key = re.sub(r'\(.*', '', name)[:90]
            a = tot.setdefault(key, [0, 0])
            a[0] += 1; a[1] += dur
all_ns = sum(v[1] for v in tot.values())
print('total kernel time %.1f ms in %d launches' % (all_ns / 1e6, sum(v[0] for v in tot.values())))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%6.2f %%  %9.3f ms  %7d x  %9.1f us  %s' % (100.0 * v[1] / all_ns, v[1] / 1e6, v[0], v[1] / 1e3 / v[0], k))
