#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE in one run, WRITE_SIZE in another -- they do not fit one pass on
gfx950) into HBM bytes per launch and merge them into profiles/hbm_traffic.json, the table bench.py reads.

Corrections (MI355X_MICROARCH.md, section HBM): the counters are in KiB-sized units (bytes = value * 1024) and on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so a wide coalesced read is under-counted by 2x; the
pair kernels here gather 16- and 32-byte rows, for which the factor is uncalibrated, so both the raw figure and
the doubled upper bound are recorded and `bytes_per_launch` uses the raw read + write (a LOWER bound on traffic).

A third database (the SQ pass) adds `valu_insts_per_launch` = SQ_INSTS_VALU per dispatch, the numerator of bench.py's
VALU roofline for the interaction-graph kernel.

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
        entry[label] = dict(rocprof_kernel=k, launches_sampled=f.get(k, (0, 0))[1], fetch_bytes=fb, fetch_bytes_if_wide=2 * fb,
                            write_bytes=wb, bytes_per_launch=fb + wb)
        if k in q:
            entry[label]['valu_insts_per_launch'] = q[k][0]
            if label in pairs and pairs[label] > 0:
                entry[label]['valu_insts_per_pair'] = q[k][0] / pairs[label]
    # the whole step: every kernel's counted bytes x its launches per force pass (the belief-propagation solve runs once per pass),
    # allocation-time fills and copies left out
    n_pass = max([v['launches_sampled'] for k, v in entry.items() if k == 'bp:rotamer'] or [1])
    per_step = {k: v['bytes_per_launch'] * v['launches_sampled'] / n_pass for k, v in entry.items() if not k.startswith('__amd_rocclr')}
    wide_step = {k: (v['fetch_bytes_if_wide'] + v['write_bytes']) * v['launches_sampled'] / n_pass for k, v in entry.items() if not k.startswith('__amd_rocclr')}
    entry['_step'] = dict(bytes_per_step=sum(per_step.values()), bytes_per_step_if_wide=sum(wide_step.values()), force_passes_sampled=n_pass,
                          share={k: b / max(sum(per_step.values()), 1.) for k, b in sorted(per_step.items(), key=lambda kv: -kv[1])[:12]})
    entry['_kernel_sources_sha256'] = kernel_source_stamp()
    tab['%s/R%s' % (workload, replicas)] = entry
    json.dump(tab, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in sorted(((k, v) for k, v in entry.items() if isinstance(v, dict) and 'bytes_per_launch' in v), key=lambda kv: -kv[1]['bytes_per_launch'])[:12]:
        print('%-40s read %8.1f MB  write %8.1f MB' % (k, v['fetch_bytes'] / 1e6, v['write_bytes'] / 1e6))
    print('whole step: %.2f GB counted (%.2f GB if every read is a wide one)' % (entry['_step']['bytes_per_step'] / 1e9, entry['_step']['bytes_per_step_if_wide'] / 1e9))


if __name__ == '__main__':
    main()
