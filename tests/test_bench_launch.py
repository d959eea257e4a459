"""bench.py's launcher contract without a GPU: `python bench.py --gpus N` with no launcher in the environment starts its N ranks
itself (a child torch.distributed.run, rendezvous on 127.0.0.1), rank 0 prints ONE JSON line, the exit code is relayed, and a
WORLD_SIZE that contradicts --gpus is refused.  The unit of work is the host-only `_stub` workload over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
        env.pop(k, None)
    return env


def test_self_launch_two_ranks_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '5', '--warmup', '2', '--workload', '_stub'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=_clean_env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().split('\n') if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['warmup'] == 2 and d['scaling'] == 'weak'
    # whole-job value: the units of BOTH ranks over the slower rank's time
    assert abs(d['value'] - 2 * 64 * 5 / (d['ms_per_step'] * 5e-3)) < 1e-6 * d['value']


def test_self_launch_eight_ranks():
    """the shape the driver's scaling run has (N = 8 ranks of one node), over gloo with the host-only unit of work"""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--steps', '4', '--warmup', '1', '--workload', '_stub'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=_clean_env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().split('\n') if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['steps'] == 4 and d['scaling'] == 'weak'
    assert abs(d['value'] - 8 * 64 * 4 / (d['ms_per_step'] * 4e-3)) < 1e-6 * d['value']


def test_single_rank_stub_and_exit_code_relay():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', '_stub'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=_clean_env())
    assert r.returncode == 0 and json.loads(r.stdout.decode().strip())['n_gpus'] == 1
    # a launcher whose world contradicts --gpus is refused loudly, by every rank
    env = dict(_clean_env(), RANK='0', LOCAL_RANK='0', WORLD_SIZE='4', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', '_stub'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=env)
    assert r.returncode != 0 and b'WORLD_SIZE=4' in r.stderr and not r.stdout.strip()
    # a rank's failure is the parent's exit code
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', '_stub', '--steps', '2', '--warmup', '1'], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, env=dict(_clean_env(), UPSIDE_BENCH_STUB_FAIL='1'))
    assert r.returncode != 0


def test_counter_traffic_is_scaled_per_system_between_large_batches_only(tmp_path, monkeypatch):
    """roofline.traffic of a batch size without its own PMC pass: the per-system bytes of the profiled batch (same kernel variants
    from 256 systems on), never for small batches, never from a table measured on other kernel sources."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', BENCH)
    b = importlib.util.module_from_spec(spec)
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    spec.loader.exec_module(b)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import hbm_traffic
    table = {'syn300_10A/R4096': {'_kernel_sources_sha256': hbm_traffic.kernel_source_stamp(), 'bp:rotamer': {'bytes_per_launch': 4096e6}}}
    (tmp_path / 'profiles').mkdir()
    (tmp_path / 'profiles' / 'hbm_traffic.json').write_text(json.dumps(table))
    monkeypatch.setattr(b, 'ROOT', str(tmp_path))
    assert b.profiled('bp:rotamer', 'syn300_10A', 4096, 'bytes_per_launch') == 4096e6
    assert b.profiled_per_system('bp:rotamer', 'syn300_10A', 1024, 'bytes_per_launch') == (1024e6, 4096)
    assert b.profiled_per_system('bp:rotamer', 'syn300_10A', 64, 'bytes_per_launch') == (None, None)       # cluster solves: other kernels
    assert b.profiled_per_system('bp:rotamer', 'syn150_10A', 512, 'bytes_per_launch') == (None, None)      # another workload
    table['syn300_10A/R4096']['_kernel_sources_sha256'] = 'stale'
    (tmp_path / 'profiles' / 'hbm_traffic.json').write_text(json.dumps(table))
    assert b.profiled_per_system('bp:rotamer', 'syn300_10A', 1024, 'bytes_per_launch') == (None, None)


def test_committed_counter_table_was_measured_on_these_kernel_sources():
    """profiles/hbm_traffic.json (roofline.traffic of the default bench line) carries the sha256 of the kernel sources its PMC passes ran
    on; bench.py drops the counters when they differ.  A kernel edit without `tools/refresh_profiles.sh pmc` shows up here, not in the
    driver's bench line."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import hbm_traffic
    with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as f:
        table = json.load(f)
    assert table, 'empty counter table'
    for key, entry in table.items():
        assert entry.get('_kernel_sources_sha256') == hbm_traffic.kernel_source_stamp(), \
            '%s: counters collected on other kernel sources (rerun tools/refresh_profiles.sh pmc on the GPU box and publish)' % key


def test_counter_calibration_is_what_the_traffic_table_applies():
    """profiles/counter_calibration.json (tools/ubench/hbm_counters.sh on the GPU box: 1 GiB moved in each of the solve's access shapes):
    every read shape counts at half its bytes, dense writes exactly; tools/hbm_traffic.py turns FETCH_SIZE / WRITE_SIZE into bytes with
    those factors, and the committed table says which it used"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import hbm_traffic
    with open(os.path.join(ROOT, 'profiles', 'counter_calibration.json')) as f:
        cal = json.load(f)
    reads = {k: v['read_factor'] for k, v in cal.items() if k.startswith('read_')}
    assert len(reads) >= 5 and all(abs(x - 2.0) < 0.01 for x in reads.values()), reads
    assert abs(cal['write_16B_per_lane_coalesced']['write_factor'] - 1.0) < 0.01
    assert 0.9 < cal['write_24B_records_as_6x4B']['write_factor'] < 1.0          # partial-line stores are counted a few per cent high
    rf, wf = hbm_traffic.calibration()
    assert abs(rf - 2.0) < 0.01 and abs(wf - 1.0) < 0.01
    with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as f:
        table = json.load(f)
    for key, entry in table.items():
        st = entry['_step']
        assert abs(st['read_factor'] - rf) < 1e-6 and abs(st['write_factor'] - wf) < 1e-6, key
        bp = entry.get('bp:rotamer')
        if bp:
            assert abs(bp['bytes_per_launch'] - (rf * bp['fetch_bytes_counted'] + bp['write_bytes'])) < 1e-3 * bp['bytes_per_launch']
