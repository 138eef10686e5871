#!/usr/bin/env python3
"""Condense a gpurun_out/prof_rNN directory (tools/prof.sh) into the tracked profiles/ summaries:
   profiles/<tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats per-kernel table (verbatim)
   profiles/<tag>_pmc.json          per kernel and per kernel class: launches, FETCH_SIZE / WRITE_SIZE (KB, raw) and
                                    HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE.
gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-B requests of
16-B/lane streaming reads at 64 B, i.e. it reports half of the bytes read; WRITE_SIZE is exact for 16-B/lane stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_r1'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r01'
os.makedirs('profiles', exist_ok=True)
ks = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))[0]
shutil.copy(ks, 'profiles/%s_kernel_stats.csv' % tag)

CLASS = [('k_rhs', 'rhs'), ('k_jvp', 'jvp'), ('k_multidot', 'multidot'), ('k_gs_update', 'gs_update'),
         ('k_lincomb', 'lincomb'), ('k_basis_axpy', 'basis_axpy'), ('k_rosw_finish', 'rosw_finish'),
         ('k_reduce_rows', 'reduce'), ('k_gfield', 'gfield'), ('k_jcoef', 'gfield'), ('k_dg_frozen', 'gfield'),
         ('k_velocity', 'velocity'), ('k_velmax', 'velocity'), ('k_spec', 'spectral'), ('k_mg_', 'mg'), ('k_cheb', 'mg'), ('k_restrict', 'mg'), ('k_prolong', 'mg')]


def cls_of(name):
    for pat, c in CLASS:
        if pat in name:
            return c
    return 'misc'


def collect(kind):
    f = glob.glob(os.path.join(src, 'pmc_%s' % kind, '*', '*_counter_collection.csv'))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[n][0] += 1
        agg[n][1] += float(r['Counter_Value'])
    return agg


# average duration per kernel class from the --kernel-trace --stats table (weighted by calls)
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(ks)):
    c = dur[cls_of(r['Name'].replace('void ', ''))]
    c[0] += int(r['Calls'])
    c[1] += float(r['TotalDurationNs'])

# ... and per kernel (template arguments kept, argument list dropped)
kdur = {}
for r in csv.DictReader(open(ks)):
    kdur[r['Name'].split('(')[0].replace('void ', '')] = (int(r['Calls']), float(r['AverageNs']) / 1e3)

fe, wr = collect('fetch'), collect('write')
kern, classes = {}, collections.defaultdict(lambda: dict(launches=0, fetch_KB=0.0, write_KB=0.0))
for n in sorted(set(fe) | set(wr)):
    nf, vf = fe.get(n, [0, 0.0])
    nw, vw = wr.get(n, [0, 0.0])
    if not (nf and nw):
        continue
    kern[n] = dict(launches=nf, FETCH_SIZE_KB_avg=vf / nf, WRITE_SIZE_KB_avg=vw / nw,
                   hbm_bytes_per_launch=(2 * vf / nf + vw / nw) * 1024,
                   avg_us=kdur.get(n, (0, None))[1], trace_calls=kdur.get(n, (0, None))[0])
    c = classes[cls_of(n)]
    c['launches'] += nf
    c['fetch_KB'] += vf
    c['write_KB'] += vw * nf / nw
out = dict(source=src, note='HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; see module docstring of tools/summarize_prof.py',
           kernels=kern,
           classes={k: dict(launches=v['launches'],
                            hbm_bytes_per_launch=(2 * v['fetch_KB'] + v['write_KB']) / v['launches'] * 1024,
                            avg_us=(dur[k][1] / dur[k][0] / 1e3) if dur[k][0] else None, trace_calls=dur[k][0])
                    for k, v in classes.items()})
json.dump(out, open('profiles/%s_pmc.json' % tag, 'w'), indent=1)
print(open('profiles/%s_kernel_stats.csv' % tag).read()[:1500])
print(json.dumps(out['classes'], indent=1))
