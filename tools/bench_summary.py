import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'no json', e); continue
    print('%s: value %.0f sys-steps/s  ms/step %.3f' % (f, d['value'], d['ms_per_step']))
    ks = d['roofline']['kernels']
    tot = sum(v['avg_ms'] for v in ks.values())
    print('   sum of profiled kernels %.3f ms' % tot)
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]['avg_ms'])[:18]:
        print('   %-45s %.4f ms  %s' % (k, v['avg_ms'], '' if v['GBps'] is None else '%.1f GB/s' % v['GBps']))
    if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'])
