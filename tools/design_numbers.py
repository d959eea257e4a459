#!/usr/bin/env python3
"""Rewrites the measured numbers of DESIGN.md section 5 (throughput table, configs[1] statement, one-GPU lines of the multi-GPU
workloads) from profiles/<tag>_bench_*.json and profiles/<tag>_bench_other_configs.txt.  usage: design_numbers.py [tag, default r04]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
def k(v):
    v = float(v)
    if v < 10000: return '%d %03d' % (int(round(v)) // 1000, int(round(v)) % 1000)
    if v < 100000: return '%.1f k' % (v / 1e3)
    if v < 1e6: return '%.0f k' % (v / 1e3)
    return '%.2f M' % (v / 1e6)
t = {}
for l in open(os.path.join(ROOT, 'profiles', tag + '_bench_other_configs.txt')):
    w, r, v = l.split(); t[(w, int(r))] = float(v)
b = lambda n: json.load(open(os.path.join(ROOT, 'profiles', '%s_bench_%s.json' % (tag, n))))
pg, pg8, rm, en = b('proteinG56_7A_R1'), b('proteinG56_7A_R8'), b('remd64_proteinG56'), b('ens512_syn150')
p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
labels = {'Trp-cage, 20 res': 'trpcage20_7A', 'protein G, 56 res': 'proteinG56_7A', '150 res, 10': 'syn150_10A', '300 res, 7': 'syn300_7A', '300 res, 10': 'syn300_10A'}
lines = s.split('\n')
for i, l in enumerate(lines):
    for lab, w in labels.items():
        if l.startswith('| ' + lab):
            c = l.split('|')
            c[2], c[3], c[4] = ' %s ' % k(t[(w, 1)]), ' %s ' % k(t[(w, 8)]), ' %s ' % k(t[(w, 64)])
            c[5] = ' %s (%s at 1024, **%s at 4096**) ' % (k(t[(w, 512)]), k(b('R1024')['value']), k(b('R4096')['value'])) if w == 'syn300_10A' else ' %s ' % k(t[(w, 512)])
            if w == 'proteinG56_7A': c[6] = ' **%s** (`profiles/%s_bench_proteinG56_7A_R1.json`: `cpu_baseline`, 1 core) ' % (k(pg['cpu_baseline']['value']), tag)
            if w == 'syn300_10A': c[6] = ' %d (≈%.1f k on 16 cores) ' % (round(b('R1')['cpu_baseline']['value']), b('R4096')['cpu_baseline']['value'] / 1e3)
            lines[i] = '|'.join(c)
s = '\n'.join(lines)
s = re.sub(r"ONE protein G runs at [\d ]+ steps/s on the MI355X and at [\d ]+ steps/s in the unmodified reference on\nONE host core of the same box — the GPU is [\d.]+× one core there; it draws level at two copies and is [\d.]+× one core at eight\.",
           "ONE protein G runs at %s steps/s on the MI355X and at %s steps/s in the unmodified reference on\nONE host core of the same box — the GPU is %.2f× one core there; it draws level at two copies and is %.1f× one core at eight."
           % (k(pg['value']), k(pg['cpu_baseline']['value']), pg['value'] / pg['cpu_baseline']['value'], pg8['value'] / pg['cpu_baseline']['value']), s)
s = re.sub(r"`remd64_proteinG56` [\d ]+k system-steps/s \(64 temperatures", "`remd64_proteinG56` %s system-steps/s (64 temperatures" % k(rm['value']), s)
s = re.sub(r"`ens512_syn150` [\d ]+k \(512 × 150 residues\)", "`ens512_syn150` %s (512 × 150 residues)" % k(en['value']), s)
s = re.sub(r"measured in the same run — [\d ]+k \([\d.]+× the one-GPU rate\) and [\d ]+k \([\d.]+×\)",
           "measured in the same run — %s (%.1f× the one-GPU rate) and %s (%.1f×)" % (k(rm['config']['projected_8gpu']['value']), rm['config']['projected_8gpu']['value'] / rm['value'],
                                                                                   k(en['config']['projected_8gpu']['value']), en['config']['projected_8gpu']['value'] / en['value']), s)
open(p, 'w').write(s)
print('protein G R=1 %s, R=8 %s; syn150 R=64 %s; syn300 R=1 %s (single_system %s), R=64 %s, R=4096 %s' % (k(t[('proteinG56_7A', 1)]), k(t[('proteinG56_7A', 8)]), k(t[('syn150_10A', 64)]),
      k(t[('syn300_10A', 1)]), k(b('R4096')['config']['single_system_steps_per_s']), k(t[('syn300_10A', 64)]), k(b('R4096')['value'])))
