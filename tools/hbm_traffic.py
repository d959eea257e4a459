#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE in one run, WRITE_SIZE in another -- they do not fit one pass on
gfx950) into HBM bytes per launch and merge them into profiles/hbm_traffic.json, the table bench.py reads.

Corrections: the counters are in KiB-sized units (bytes = value * 1024), and on gfx950 FETCH_SIZE tallies every read request at 64
bytes while the L2 fetches 128-byte lines.  Round 6 calibrated this on known byte counts in the access shapes of the kernels here
(tools/ubench/hbm_counters.hip -> profiles/r06_counter_calibration.txt, profiles/counter_calibration.json): 16-, 8- and 4-byte
coalesced reads, 24-byte records read as three 8-byte loads, and half-used 24-byte records ALL read back true bytes / FETCH_SIZE =
2.000; WRITE_SIZE is exact for dense 4- to 16-byte stores (1.000) and 4.6 % high for 4-byte stores into 24-byte records.  So
    bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE
is ONE calibrated number (the factors are read from profiles/counter_calibration.json when present).  What it counts is traffic on
the L2's memory side: Infinity-Cache hits are included (MI355X_MICROARCH.md, HBM).

A third database (the SQ pass) adds `valu_insts_per_launch` = SQ_INSTS_VALU per dispatch and `valu_busy` = the share of the time a SIMD's
vector unit was issuing: SQ_ACTIVE_INST_VALU / SIMDs / (SQ_WAVE_CYCLES / resident waves) -- both counters tick in quad-cycles, and every
wave64 VALU instruction (packed or not) holds its SIMD for one quad-cycle.

`pairs` (optional, label=count,...: pair evaluations per launch as bench.py's profile dump reports them) adds
`valu_insts_per_pair`.

usage: hbm_traffic.py fetch.db write.db workload replicas [out.json] [sq.db] [pairs]"""
import collections, json, os, sqlite3, sys

LABEL = {  # rocprof kernel name prefix -> (bench.py profile label, index of the launch of that kernel within one force pass)
    'k_rotamer_grad': 'igraph_bwd:rotamer',
    'k_rotamer_pair_energy': 'igraph_fwd:rotamer',
    'k_rotamer_bp': 'bp:rotamer',
}


def per_kernel(dbfile, counter):
    db = sqlite3.connect(dbfile)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    acc = collections.defaultdict(list)
    for k, c, v in rows:
        if c == counter:
            acc[k.split('(')[0]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def calibration():
    """(read factor, write factor): true bytes per counted byte, from the committed calibration (all dense read shapes agree)"""
    try:
        t = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'counter_calibration.json')))
        rf = [v['read_factor'] for v in t.values() if v.get('read_factor')]
        wf = [v['write_factor'] for k, v in t.items() if v.get('write_factor') and k.startswith('write_16B')]
        return (sum(rf) / len(rf) if rf else 2.0), (wf[0] if wf else 1.0)
    except (OSError, ValueError, KeyError):
        return 2.0, 1.0


def launch_shapes(dbfile):
    """kernel -> (workgroup size, LDS bytes, workgroups) of its launches (the SQ pass's kernel trace)"""
    db = sqlite3.connect(dbfile)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    need = ['name', 'workgroup_x', 'lds_size', 'grid_x', 'grid_y']
    if not all(c in cols for c in need):
        return {}
    out = {}
    for n, wg, lds, gx, gy in db.execute("select name, workgroup_x, lds_size, grid_x, grid_y from kernels"):
        out.setdefault(n.split('(')[0], (wg, lds, (gx // max(wg, 1)) * max(gy, 1)))
    return out


def kernel_source_stamp():
    """sha256 over the kernel sources the counters were collected on: bench.py compares it with the sources it runs and
    reports the table as stale instead of quoting counters of kernels that have changed since"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'upside-md_amd', 'csrc')
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):
        if fn.endswith(('.hip', '.h')):
            h.update(fn.encode()); h.update(open(os.path.join(root, fn), 'rb').read())
    return h.hexdigest()


def main():
    fetch_db, write_db, workload, replicas = sys.argv[1:5]
    sq_db = sys.argv[6] if len(sys.argv) > 6 else None
    pairs = dict((kv.split('=')[0], float(kv.split('=')[1])) for kv in sys.argv[7].split(',')) if len(sys.argv) > 7 and sys.argv[7] else {}
    out = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'hbm_traffic.json')
    f = per_kernel(fetch_db, 'FETCH_SIZE'); w = per_kernel(write_db, 'WRITE_SIZE')
    q = per_kernel(sq_db, 'SQ_INSTS_VALU') if sq_db else {}
    qa = per_kernel(sq_db, 'SQ_ACTIVE_INST_VALU') if sq_db else {}
    qw = per_kernel(sq_db, 'SQ_WAVE_CYCLES') if sq_db else {}
    shapes = launch_shapes(sq_db) if sq_db else {}
    read_x, write_x = calibration()
    try:
        tab = json.load(open(out))
    except (OSError, ValueError):
        tab = {}
    entry = {}
    total = lambda k: (f.get(k, (0, 0))[0] * f.get(k, (0, 0))[1] + w.get(k, (0, 0))[0] * w.get(k, (0, 0))[1])
    # (template instances of one labelled kernel: lightest first, so the one that moves the most bytes ends up in the table)
    for k in sorted(set(f) | set(w), key=lambda k: (next((v for key, v in LABEL.items() if key in k), k), total(k))):
        fb = f.get(k, (0, 0))[0] * 1024.0; wb = w.get(k, (0, 0))[0] * 1024.0
        label = next((v for key, v in LABEL.items() if key in k), k)   # template instances carry a 'void ...<true>' decoration
        entry[label] = dict(rocprof_kernel=k, launches_sampled=f.get(k, (0, 0))[1], fetch_bytes_counted=fb, fetch_bytes=read_x * fb,
                            write_bytes=write_x * wb, bytes_per_launch=read_x * fb + write_x * wb)
        if k in q:
            entry[label]['valu_insts_per_launch'] = q[k][0]
            if label in pairs and pairs[label] > 0:
                entry[label]['valu_insts_per_pair'] = q[k][0] / pairs[label]
        if k in qa and k in qw and k in shapes:
            wg, lds, n_wg = shapes[k]
            per_cu = max(1, min(2048 // max(wg, 1), (160 * 1024) // max(lds, 1) if lds else 32, 32 * 64 // max(wg, 1)))
            resident = min(256 * per_cu, n_wg) * ((wg + 63) // 64)       # waves in flight while the launch fills the device
            entry[label]['valu_busy'] = qa[k][0] / 1024. / (qw[k][0] / resident)      # 1024 SIMDs
            entry[label]['valu_busy_basis'] = dict(workgroup=wg, lds_bytes=lds, workgroups=n_wg, waves_resident=resident)
    # the whole step: every kernel's counted bytes x its launches per force pass (the belief-propagation solve runs once per pass),
    # allocation-time fills and copies left out
    n_pass = max([v['launches_sampled'] for k, v in entry.items() if k == 'bp:rotamer'] or [1])
    per_step = {k: v['bytes_per_launch'] * v['launches_sampled'] / n_pass for k, v in entry.items() if not k.startswith('__amd_rocclr')}
    entry['_step'] = dict(bytes_per_step=sum(per_step.values()), force_passes_sampled=n_pass, read_factor=read_x, write_factor=write_x,
                          share={k: b / max(sum(per_step.values()), 1.) for k, b in sorted(per_step.items(), key=lambda kv: -kv[1])[:12]})
    entry['_kernel_sources_sha256'] = kernel_source_stamp()
    tab['%s/R%s' % (workload, replicas)] = entry
    json.dump(tab, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in sorted(((k, v) for k, v in entry.items() if isinstance(v, dict) and 'bytes_per_launch' in v), key=lambda kv: -kv[1]['bytes_per_launch'])[:12]:
        print('%-40s read %8.1f MB  write %8.1f MB' % (k, v['fetch_bytes'] / 1e6, v['write_bytes'] / 1e6))
    print('whole step: %.2f GB (2 x FETCH_SIZE + WRITE_SIZE)' % (entry['_step']['bytes_per_step'] / 1e9))


if __name__ == '__main__':
    main()
