#!/bin/bash
# GPU box: the two PMC passes of tools/ubench/hbm_counters (separate runs, kernel trace only) -> gpurun_out/calib/counter_calibration.{txt,json}
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/calib
rm -rf "$OUT"; mkdir -p "$OUT"
BIN=$PWD/tools/ubench/hbm_counters
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $BIN > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $BIN > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d "$OUT/req" -o req -- $BIN > "$OUT/req.log" 2>&1
python3 - "$OUT" <<'PY'
import collections, glob, json, os, sqlite3, sys
out = sys.argv[1]
BYTES = float(1 << 30)
def per_kernel(sub):
    dbs = glob.glob(os.path.join(out, sub, '**', '*.db'), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    if not dbs: return acc, dur
    db = sqlite3.connect(dbs[0])
    for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        acc[k.split('(')[0]][c].append(v)
    for n, s, e in db.execute("select name, start, end from kernels"):
        dur[n.split('(')[0]].append((e - s) / 1e3)
    return acc, dur
f, fd = per_kernel('fetch'); w, _ = per_kernel('write'); r, _ = per_kernel('req')
lines = ['counter calibration on known byte counts (tools/ubench/hbm_counters.hip): every launch moves 2^30 bytes, dense, 4x the Infinity Cache',
         '%-34s %12s %12s %10s %10s %10s   %s' % ('kernel (= access shape)', 'FETCH_SIZE', 'WRITE_SIZE', 'read x', 'write x', 'GB/s*', 'raw request counters (avg per launch)')]
table = {}
for k in sorted(set(f) | set(w)):
    fs = f.get(k, {}).get('FETCH_SIZE', [0.]); ws = w.get(k, {}).get('WRITE_SIZE', [0.])
    fb = sum(fs) / len(fs) * 1024.; wb = sum(ws) / len(ws) * 1024.
    is_read = k.startswith('read')
    us = sorted(fd.get(k, [0.]))[len(fd.get(k, [0.])) // 2]
    req = {c: sum(v) / len(v) for c, v in r.get(k, {}).items()}
    table[k] = dict(fetch_bytes_counted=fb, write_bytes_counted=wb, true_bytes=BYTES, us_under_profiler=us,
                    read_factor=(BYTES / fb if is_read and fb else None), write_factor=(BYTES / wb if (not is_read) and wb else None), requests=req)
    lines.append('%-34s %12.4e %12.4e %10s %10s %10.0f   %s' % (k, fb, wb, ('%.3f' % (BYTES / fb)) if is_read and fb else '-', ('%.3f' % (BYTES / wb)) if (not is_read) and wb else '-',
                 BYTES / (us * 1e-6) / 1e9 if us else 0., ' '.join('%s=%.3e' % (c.replace('TCC_EA0_', '').replace('_sum', ''), v) for c, v in sorted(req.items()))))
lines.append('read x / write x = true bytes / counted bytes (the factor to multiply FETCH_SIZE / WRITE_SIZE by for that shape); * under the profiler')
open(os.path.join(out, 'counter_calibration.txt'), 'w').write('\n'.join(lines) + '\n')
json.dump(table, open(os.path.join(out, 'counter_calibration.json'), 'w'), indent=1, sort_keys=True)
print('\n'.join(lines))
PY
find "$OUT" -name "*.db" -delete
