#!/usr/bin/env python3
"""Benchmark of the MD inner loop (force pass + leapfrog/OU integrator) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--replicas R] [--workload NAME]

A "step" is one force evaluation + one leapfrog stage of EVERY replica resident on the GPU (the reference's
"MD step", /root/reference/src/main.cpp:677-682; one integration cycle = 3 steps,
/root/reference/src/deriv_engine.cpp:172-192).  The default workload is BASELINE.json configs[2]: the 300-residue
synthetic protein with full side-chain belief propagation and the 10 A pair list (tests/golden/syn300_10A.up),
held as R independent replicas per GPU (different thermostat seeds), everything resident in HBM.
`value` = system-steps per second summed over all replicas and GPUs -- the reciprocal of the reference's own
"us/systems/step" figure.  Multi-GPU runs of it are weak scaling: every rank owns R more replicas; there is no
data-path collective (replicas are independent).

Other workloads (--workload): any fixture name (tests/golden/<name>.up) held as R replicas, and BASELINE.json's two
multi-GPU configurations, whose total work is fixed (strong scaling):
  remd64_proteinG56  configs[3]: 64 temperatures of the 56-residue protein, a geometric ladder in contiguous blocks per GPU,
                     replica exchange every 5 time units with two alternating neighbour swap sets through the C++ RCCL path
                     (upside_hip_comm_*: one fp32 per replica all-gathered, verdicts on the device, straddling pairs moved
                     by ncclSend/ncclRecv -- no host staging)
  ens512_syn150      configs[4]: 512 independent 150-residue proteins, 512 / N per GPU, no communication

Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as ct
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL's version banner goes to stdout; rank 0 must print ONE JSON line there
os.environ['NCCL_DEBUG'] = os.environ.get('UPSIDE_NCCL_DEBUG', 'WARN')
os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')      # ... and its warnings do not belong there either
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TEMPERATURE = 0.8
DT = 0.009


def bind(lib):
    c = lib.calc
    c.upside_hip_set_device.argtypes = [ct.c_int]
    c.upside_hip_construct.restype = ct.c_void_p
    c.upside_hip_construct.argtypes = [ct.c_int, ct.c_char_p, ct.c_int, ct.c_bool]
    c.upside_hip_set_pos.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_get_pos.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_init_md.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint32, ct.c_float, ct.c_float, ct.c_int]
    c.upside_hip_run_md.argtypes = [ct.c_void_p, ct.c_int]
    c.upside_hip_run_steps.argtypes = [ct.c_void_p, ct.c_int]
    c.upside_hip_compute.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p]
    c.upside_hip_profile_reset.argtypes = [ct.c_void_p, ct.c_int]
    c.upside_hip_profile_dump.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_int]
    c.upside_hip_igraph_bytes_per_system.restype = ct.c_double
    c.upside_hip_igraph_bytes_per_system.argtypes = [ct.c_void_p]
    c.upside_hip_bp_min_bytes.restype = ct.c_double
    c.upside_hip_bp_min_bytes.argtypes = [ct.c_void_p]
    c.upside_hip_last_error.restype = ct.c_char_p
    c.upside_hip_calibrate_valu.argtypes = [ct.c_void_p]
    c.upside_hip_comm_get_unique_id.argtypes = [ct.c_char_p]
    c.upside_hip_comm_init.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_char_p, ct.c_void_p]
    c.upside_hip_comm_replica_swap.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int, ct.c_void_p]
    c.upside_hip_comm_free.argtypes = [ct.c_void_p]
    return c


def _trace(msg):
    if os.environ.get('UPSIDE_BENCH_TRACE'):
        print('[bench] ' + msg, file=sys.stderr, flush=True)


def check(c, rc, what):
    if rc:
        raise RuntimeError('%s failed: %s' % (what, c.upside_hip_last_error().decode()))


def cpu_baseline(fixture, variant, budget_s=float(os.environ.get('UPSIDE_BENCH_CPU_BUDGET_S', '15')), one_core=False, n_atom=900):
    """the unmodified reference (oracle/_ref, kind "reference") timed on this host's cores over a bounded
    sample; falls back to the C restatement (kind "port", 1 core) when the reference binary is absent."""
    exe = os.path.join(ROOT, 'oracle', '_ref', 'upside_' + variant)
    n_cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    if os.path.exists(exe):
        # one thread per system as the reference parallelises (main.cpp:618); one thread per system, at most UPSIDE_BENCH_CPU_THREADS (a one-system
        # workload is compared with ONE host core).  `cores` = threads used, `host_cores_total` = what the host offers this process
        # (16 threads by default: measured on a 256-core GPU host, 64 threads deliver LESS in total -- 2.8 k against 4.1 k system-steps/s --
        #  and take 37 s instead of 7 s of wall time)
        n_sys = 1 if one_core else max(1, min(n_cores, int(os.environ.get('UPSIDE_BENCH_CPU_THREADS', '16'))))
        # ~9 ms per step per core for the 300-residue workload, roughly linear in the atoms: size the sample for `budget_s`
        steps = max(30, int(budget_s / (0.010 * max(n_atom, 60) / 900.)))
        duration = steps * DT
        tmp = tempfile.mkdtemp(prefix='upside_cpu_')
        try:
            files = []
            for i in range(n_sys):
                f = os.path.join(tmp, 'sys%d.up' % i)
                shutil.copyfile(fixture, f)
                files.append(f)
            env = dict(os.environ, OMP_NUM_THREADS=str(n_sys))
            t0 = time.time()
            out = subprocess.run([exe, '--duration', '%g' % duration, '--frame-interval', '%g' % duration,
                                  '--temperature', str(TEMPERATURE), '--seed', '1'] + files,
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600).stdout.decode()
            wall = time.time() - t0
            m = re.search(r'\(([\d.eE+-]+) us/systems/step', out)
            if m:
                us = float(m.group(1))
                return dict(value=1e6 / us * 1.0, unit='system-steps/s', cores=n_sys, host_cores_total=n_cores, kind='reference',
                            sample='%d systems x %d steps of the same .up, one OpenMP thread per system, %.1f s wall; '
                                   'value = 1e6/(us/systems/step) as printed by the reference' % (n_sys, steps, wall))
        except Exception as e:  # pragma: no cover
            sys.stderr.write('reference cpu baseline failed: %r\n' % (e,))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    # port: the oracle's MD loop on one core
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as P
    orc = P.pkg.Upside(fixture, library=P.oracle_library())
    pos = orc.initial_pos.copy()
    mom = np.zeros_like(pos)
    n_round = 10
    t0 = time.time()
    orc.calc.oracle_run_md(orc.engine, pos.ctypes.data, mom.ctypes.data, n_round, DT, TEMPERATURE, 1, 5.0, 1)
    wall = time.time() - t0
    return dict(value=3 * n_round / wall, unit='system-steps/s', cores=1, host_cores_total=n_cores, kind='port',
                sample='%d steps of the C restatement (oracle/upside_oracle.c), 1 core' % (3 * n_round))


PROFILE_TABLE = 'profiles/hbm_traffic.json'
PROFILE_NOTE = ('from the rocprofv3 PMC passes of this command committed under profiles/ (FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in '
                'separate passes, tools/refresh_profiles.sh); not collected in this run')


_TABLE_STATE = {}


def profiled(kernel_label, workload, replicas, field):
    """per-launch counter value of `kernel_label` from the committed PMC passes (tools/hbm_traffic.py applies the guide's
    gfx950 corrections to the byte counters); None when no profile exists for this workload / replica count, or when the
    kernel sources have changed since the counters were collected (the table records their sha256)."""
    try:
        with open(os.path.join(ROOT, PROFILE_TABLE)) as f:
            tab = json.load(f)
        entry = tab['%s/R%d' % (workload, replicas)]
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import hbm_traffic
        stamp = entry.get('_kernel_sources_sha256')
        _TABLE_STATE['stale'] = stamp != hbm_traffic.kernel_source_stamp()
        if _TABLE_STATE['stale']:
            return None
        return entry[kernel_label][field]
    except (OSError, KeyError, ValueError):
        return None


def profiled_per_system(kernel_label, workload, replicas, field):
    """(value, source replica count): as profiled(), from the table's entry of the SAME workload at another batch size, scaled per
    system -- only between batches of at least 256 systems, where the same kernel variants run (one workgroup per system, the
    dense solve) and a system's bytes do not depend on how many systems share the launch; (None, None) otherwise."""
    if replicas < 256:
        return None, None
    try:
        with open(os.path.join(ROOT, PROFILE_TABLE)) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return None, None
    for key in tab:
        w, _, r0 = key.rpartition('/R')
        if w != workload or not r0.isdigit() or int(r0) < 256 or int(r0) == replicas:
            continue
        v = profiled(kernel_label, workload, int(r0), field)
        if v is not None:
            return v * replicas / int(r0), int(r0)
    return None, None


WORKLOADS = {   # name -> (fixture, description, total systems or None (= --replicas per GPU, weak scaling))
    'remd64_proteinG56': ('proteinG56_7A', '64-temperature replica exchange of the 56-residue protein (BASELINE.json configs[3]): geometric '
                          'ladder T=0.50..1.00 in contiguous blocks per GPU, exchange attempt every 5 time units (185 rounds) with two '
                          'alternating neighbour swap sets over RCCL', 64),
    'ens512_syn150': ('syn150_10A', '512 independent 150-residue proteins (BASELINE.json configs[4]), 512 / N per GPU, no communication', 512),
}
REPLICA_INTERVAL = 5.0     # time units between exchange attempts (README.md:189-193 pattern)


PARITY_TOL = 1e-5      # north_star: forces / energies within 1e-5 relative (relative RMS per array, tests/parity_util.py)


def parity_check(pkg, c, eng, fixture, R, n_atom, n_check=4):
    """Forces and energies of the engine that was just timed, for `n_check` of its replicas drawn at random, at the positions the
    replicas have reached: against the oracle (oracle/upside_oracle.c, the checker) and against a fresh ONE-system engine (the
    configuration every golden-vector test runs).  Relative RMS per array as in tests/parity_util.py; the run fails above 1e-5."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as P
    pos = np.empty((R, n_atom, 3), dtype='f4'); energy = np.empty(R, dtype='f4'); deriv = np.empty((R, n_atom, 3), dtype='f4')
    check(c, c.upside_hip_get_pos(eng, pos.ctypes.data), 'get_pos')
    check(c, c.upside_hip_compute(eng, energy.ctypes.data, deriv.ctypes.data), 'compute')
    pick = sorted(np.random.RandomState(20260 + R).choice(R, min(n_check, R), replace=False).tolist())
    orc = pkg.Upside(fixture, library=P.oracle_library())
    one = pkg.Upside(fixture)
    worst = dict(deriv_vs_oracle=0., energy_vs_oracle=0., deriv_vs_one_system_engine=0., one_system_engine_vs_oracle=0.)
    for k in pick:
        x = np.ascontiguousarray(pos[k])
        e_ref = float(orc.energy(x)); d_ref = orc.deriv(x)
        # a total energy is a sum of large terms of both signs: its error is measured against the sum of the |per-node potentials|
        # (the gate of tests/test_gpu_parity.py), not against the total
        e_scale = max(1., sum(abs(float(orc.get_output(nm)[0, 0])) for nm in P.POTENTIAL_NODES))
        d_one = one.deriv(x)
        worst['deriv_vs_oracle'] = max(worst['deriv_vs_oracle'], P.rel_rms(d_ref, deriv[k]))
        worst['energy_vs_oracle'] = max(worst['energy_vs_oracle'], abs(float(energy[k]) - e_ref) / e_scale)
        worst['deriv_vs_one_system_engine'] = max(worst['deriv_vs_one_system_engine'], P.rel_rms(d_one, deriv[k]))
        worst['one_system_engine_vs_oracle'] = max(worst['one_system_engine_vs_oracle'], P.rel_rms(d_ref, d_one))
    finite = bool(np.isfinite(deriv).all() and np.isfinite(energy).all())
    mx = max(worst['deriv_vs_oracle'], worst['energy_vs_oracle'])
    return dict(max_rel_rms=mx, n=len(pick), tol=PARITY_TOL, ok=bool(finite and mx <= PARITY_TOL), replicas=pick,
                all_replicas_finite=finite, systems_in_engine=R, **worst,
                what='forces (relative RMS) and total energy (relative to the sum of |node potentials|) of %d random replicas of the TIMED engine at their current positions vs '
                     'oracle/upside_oracle.c; also vs a fresh one-system engine' % len(pick))


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1), relay their output -- rank 0 prints the one JSON line -- and the exit code.  A child, not an exec:
    the contract of the GPU boxes forbids replacing a process, and the parent has not touched the GPU."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    _trace('self-launch: ' + ' '.join(cmd))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # (dmabuf IPC: RCCL between processes needs it on these hosts)
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


def stub_main(args, rep, rank, world):
    """--workload _stub: the launcher / barrier / max-over-ranks / one-line contract of this file with a host-only unit of work over
    gloo (tests/test_bench_contract.py runs it at N = 2 on a machine without a GPU).  Not a benchmark."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo')
    d = dist if world > 1 else None
    units = 64

    def run(n):
        x = np.arange(1 << 12, dtype='f8')
        for _ in range(n):
            x = np.sqrt(x * x + 1.)
        return n
    run(args.warmup); rep.barrier(d)
    t0 = time.perf_counter(); done = run(args.steps); rep.barrier(d)
    elapsed = time.perf_counter() - t0
    value, elapsed = rep.job_throughput(d, units * done, elapsed)
    if rank == 0:
        print(json.dumps(dict(metric='stub units/s (launcher contract test, not a benchmark)', value=value, unit='units/s', n_gpus=world,
                              steps=done, warmup=args.warmup, ms_per_step=elapsed / done * 1e3, higher_is_better=True, scaling='weak',
                              vs_baseline=None, dtype='f64', data='synthetic', config=dict(workload='_stub: host-only contract test'))), flush=True)
    if d is not None:
        dist.barrier(); dist.destroy_process_group()
    return 3 if (os.environ.get('UPSIDE_BENCH_STUB_FAIL') == str(rank)) else 0      # (test hook: one rank fails, the launcher relays it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--settle', type=int, default=100,
                    help='MD steps run while the workload is set up, before the warm-up: the replicas start from one structure (plus noise), so their '
                         'pair lists are all built on step 0 and fall due together for the first few dozen steps -- a start-up burst, not the '
                         'steady state the metric is about (20 timed steps after 5 / 65 / 125 untimed ones: 209 / 215 / 216 k system-steps/s at 4096 replicas)')
    ap.add_argument('--replicas', type=int, default=int(os.environ.get('UPSIDE_BENCH_REPLICAS', '4096')),
                    help='independent replicas resident per GPU')
    ap.add_argument('--workload', default='syn300_10A')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-check', action='store_true',
                    help='skip the oracle check of the timed engine (profiling passes: keeps its one-system launches out of the averages)')
    ap.add_argument('--no-single-system', action='store_true',
                    help='skip the one-replica latency leg (PMC passes: keeps its small launches out of the per-kernel averages)')
    args = ap.parse_args()

    pkg = load_package()
    rep = pkg.replicas
    rank, local_rank, world = rep.world_from_env()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)           # (nothing has touched the GPU yet: the ranks are CHILD processes, never an exec)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher exports WORLD_SIZE=%d (start it as `python bench.py --gpus N` or with '
                         'torch.distributed.run --nproc-per-node N)' % (args.gpus, world))
    if args.workload == '_stub':
        return stub_main(args, rep, rank, world)

    import torch
    dist = None
    # all ranks on device 0 over gloo + a stand-in collective library: the two-ranks-on-one-GPU test of the launcher path (RCCL
    # cannot put two ranks on one device); a real multi-GPU run has neither variable set
    one_device = os.environ.get('UPSIDE_BENCH_ONE_DEVICE') == '1'
    backend = os.environ.get('UPSIDE_BENCH_DIST_BACKEND', 'nccl')
    dist_device = 'cuda' if backend == 'nccl' else 'cpu'
    if one_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    lib = pkg.default_library()          # raises if the HIP extension is missing: no fallback
    c = bind(lib)
    check(c, c.upside_hip_set_device(local_rank), 'set_device')

    total_systems = None
    if args.workload in WORKLOADS:
        fix_name, describe, total_systems = WORKLOADS[args.workload]
        if total_systems % world:
            raise SystemExit('%s needs a GPU count that divides %d' % (args.workload, total_systems))
        R = total_systems // world
    else:
        fix_name, R = args.workload, args.replicas
        describe = None
    fixture = os.path.join(ROOT, 'tests', 'golden', fix_name + '.up')
    variant = '10A' if fix_name.endswith('10A') else '7A'
    pos0 = pkg.config.read_pos(fixture)
    n_atom = pos0.shape[0]
    free_before = torch.cuda.mem_get_info()[0]
    eng = c.upside_hip_construct(n_atom, fixture.encode(), R, True)
    if not eng:
        raise RuntimeError('engine construction failed: %s' % c.upside_hip_last_error().decode())
    pos = np.tile(pos0[None], (R, 1, 1)).astype('f4')
    first_global, _ = rep.weak_shard(R, world, rank)
    if R > 1:      # every replica starts from its own structure (0.05 A of noise keyed by the GLOBAL replica index), so that the
                   # pair-list rebuilds of the replicas are out of phase from the first timed step on
        for r in range(R):
            pos[r] += np.random.RandomState(977 + first_global + r).normal(0., 0.05, pos0.shape).astype('f4')
    pos = np.ascontiguousarray(pos)
    check(c, c.upside_hip_set_pos(eng, pos.ctypes.data), 'set_pos')
    first, _ = rep.weak_shard(R, world, rank)
    remd = args.workload == 'remd64_proteinG56'
    if remd:
        ladder = rep.geometric_ladder(0.5, 1.0, total_systems)
        temps = np.ascontiguousarray(ladder[first:first + R])
    else:
        ladder = None
        temps = np.full(R, TEMPERATURE, dtype='f4')
    # every replica of every rank gets its own thermostat stream: seed = base + GLOBAL replica index (main.cpp:459)
    check(c, c.upside_hip_init_md(eng, temps.ctypes.data, rep.system_seed(1000, first), 5.0, DT, 1), 'init_md')
    swap_sets, exchange_steps = [], 0
    if remd:
        uid = ct.create_string_buffer(128)
        if rank == 0:
            check(c, c.upside_hip_comm_get_unique_id(uid), 'comm_get_unique_id')
        if world > 1:       # the launcher's own rendezvous hands the id around
            t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).to(dist_device)
            dist.broadcast(t, 0)
            uid = ct.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        check(c, c.upside_hip_comm_init(eng, rank, world, uid, np.ascontiguousarray(ladder).ctypes.data), 'comm_init')
        swap_sets = [np.ascontiguousarray(np.array(x, 'i4')) for x in rep.neighbour_swap_sets(total_systems)]
        exchange_steps = 3 * max(1, int(REPLICA_INTERVAL / (3 * DT)))      # main.cpp:445-447: the interval in rounds

    state = dict(done=0, attempts=0)

    def run_steps(n):
        """exactly n force evaluations (+ the exchange attempts that fall due); returns after the stream has drained"""
        left = n
        while left > 0:
            chunk = left if not exchange_steps else min(left, exchange_steps - state['done'] % exchange_steps)
            check(c, c.upside_hip_run_steps(eng, chunk), 'run_steps')
            state['done'] += chunk; left -= chunk
            if exchange_steps and state['done'] % exchange_steps == 0:
                rnd = state['done'] // 3
                for k, pairs in enumerate(swap_sets):      # nothing comes back to the host: the sets are enqueued behind the MD steps
                    check(c, c.upside_hip_comm_replica_swap(eng, len(pairs), pairs.ctypes.data, 1000, rnd, int(k == 0), None), 'comm_replica_swap')
                state['attempts'] += 1
        return n

    def barrier():
        rep.barrier(dist, torch.cuda.synchronize)

    _trace('engine ready, settling')
    run_steps(args.settle)           # set-up, reported as config.settle_steps: de-phases the replicas' list rebuilds (see --settle)
    _trace('warm-up')
    run_steps(args.warmup)
    barrier()
    engine_bytes = free_before - torch.cuda.mem_get_info()[0]      # everything the engine holds for its R systems (lists sized on the first pass included)
    _trace('timed region')
    attempts0 = state['attempts']
    t0 = time.perf_counter()
    steps_done = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    value, elapsed = rep.job_throughput(dist, R * steps_done, elapsed, device=dist_device)   # all ranks' units / slowest rank
    timed_attempts = state['attempts'] - attempts0
    exchange_steps_saved, exchange_steps = exchange_steps, 0      # the profiling legs below run plain MD

    _trace('timed region done: %d steps' % steps_done)
    # ---- parity of what was just timed (outside the timed region): forces of THIS engine -- this batch size, hence these code paths --
    # at the positions its replicas have reached, against the oracle and against a fresh one-system engine
    parity = None
    if not args.no_parity_check:      # EVERY rank checks its own engine (no rank idles in a barrier while rank 0 runs the oracle); rank 0 reports the worst
        parity = parity_check(pkg, c, eng, fixture, R, n_atom)
        if dist is not None:
            every = [None] * world
            dist.all_gather_object(every, parity)
            worst_rank = max(range(world), key=lambda r: (not every[r]['ok'], every[r]['max_rel_rms']))
            parity = dict(every[worst_rank], ok=all(p['ok'] for p in every), rank_reported=worst_rank,
                          max_rel_rms_by_rank=[p['max_rel_rms'] for p in every])
        _trace('parity check: %r' % (parity,))
    # ---- roofline of the dominant kernel: HIP-event timing on the engine's stream, outside the timed region
    roofline = None
    if rank == 0:
        check(c, c.upside_hip_profile_reset(eng, 1), 'profile_reset')
        run_steps(30)
        buf = ct.create_string_buffer(1 << 16)
        check(c, c.upside_hip_profile_dump(eng, buf, len(buf)), 'profile_dump')
        check(c, c.upside_hip_profile_reset(eng, 0), 'profile_reset')
        rows = []
        for ln in buf.value.decode().strip().split('\n'):
            nm, ms, n, by, pr = ln.split()
            rows.append((nm, float(ms), int(n), float(by), float(pr)))
        def entry(r):
            avg_ms = r[1] / r[2]
            bytes_per_launch = r[3] / r[2]
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = profiled(r[0], args.workload, R, 'bytes_per_launch')
            d = dict(bound='hbm', kernel=r[0], achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                     frac=achieved / HBM_PEAK_GBS, traffic=traffic, avg_launch_ms=avg_ms,
                     algorithmic_bytes_per_launch=bytes_per_launch)
            if d['frac'] <= 1.0: d['frac_model'] = d['frac']      # SURVEY 8d bytes (every sweep's re-read of the pair matrices counted) / time / peak; withheld above 1
            if traffic is not None:      # the rate the counters saw is the headline: achieved = HBM bytes of the PMC passes / time
                d['traffic_source'] = PROFILE_NOTE
                d['frac_counter'] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                d['achieved_model'] = d['achieved']
                d['achieved'] = traffic / (avg_ms * 1e-3) / 1e9
                d['frac'] = d['frac_counter']
            elif _TABLE_STATE.get('stale'):
                d['traffic_source'] = 'profiles/hbm_traffic.json is STALE: the kernel sources changed after its PMC passes (rerun tools/refresh_profiles.sh pmc)'
            else:
                # no counter pass of this batch size: the per-system bytes of the profiled batch (same kernel variants from 256 systems on)
                est, r0 = profiled_per_system(r[0], args.workload, R, 'bytes_per_launch')
                if est is not None:
                    d['traffic'] = est
                    d['traffic_source'] = ('per-system HBM bytes of the %d-system PMC passes (%s) x %d systems: the same kernel variant, '
                                           'not counted at this batch size' % (r0, PROFILE_TABLE, R))
                    d['achieved_model'] = d['achieved']
                    d['achieved'] = est / (avg_ms * 1e-3) / 1e9
                    d['frac'] = d['frac_counter'] = d['achieved'] / HBM_PEAK_GBS
                else:
                    d['frac_note'] = ('model bytes only (SURVEY 8d counts every sweep\'s re-read of the pair matrices; the solve keeps them in '
                                      'registers / LDS / L2, so the model rate may exceed the HBM peak); counters exist for the default command only')
            if d['frac'] is not None and d['frac'] > 1.0:      # a model rate above the peak is not a measurement of the memory system: no fraction
                d['frac_note'] = d.get('frac_note', '') + ' -- achieved_model exceeds the HBM peak: bytes served on chip; frac withheld'
                d['achieved_model'] = d.get('achieved_model', d['achieved']); d['achieved'] = None; d['frac'] = None
            return d
        # the dominant kernel of the step (most time): belief propagation, which streams the pair matrices and messages
        # every sweep
        dom = max(rows, key=lambda r: r[1])
        roofline = entry(dom)
        if dom[0].startswith('bp:'):      # what the solve must move at least once (active matrices in, marginals out); counted traffic / this = re-read factor
            mn = c.upside_hip_bp_min_bytes(eng)
            if mn > 0:
                roofline['min_bytes_per_launch'] = mn
                if roofline.get('traffic'): roofline['traffic_over_min'] = roofline['traffic'] / mn
        if roofline.get('traffic') is not None:
            # (ONE number since round 6: traffic = 2 x FETCH_SIZE + WRITE_SIZE, the factors calibrated on known byte counts in the solve's own
            #  access shapes -- tools/ubench/hbm_counters.hip, profiles/r06_counter_calibration.txt: every read shape 2.000, dense writes 1.000)
            roofline['traffic_calibration'] = dict(read_factor=profiled('_step', args.workload, R, 'read_factor'),
                                                   write_factor=profiled('_step', args.workload, R, 'write_factor'),
                                                   source='profiles/r06_counter_calibration.txt',
                                                   counts='requests on the memory side of the L2 (Infinity-Cache hits included)')
        # the whole step: counted HBM bytes of every kernel of a force pass (same PMC passes) / the measured step time
        st = profiled('_step', args.workload, R, 'bytes_per_step')
        if st is not None:
            ms_step = elapsed / steps_done * 1e3
            roofline['step'] = dict(bound='hbm', bytes_per_step=st, achieved=st / (ms_step * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                                    frac=st / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, ms_per_step=ms_step,
                                    share_of_bytes=profiled('_step', args.workload, R, 'share'), traffic_source=PROFILE_NOTE)
        # The interaction-graph kernel north_star names: the side-chain gradient pass.  It is bound by VALU issue, not by HBM
        # (DESIGN.md section 3), so its roofline is arithmetic: achieved = pair evaluations of the launch (counted by the
        # engine) x 350 flop per evaluation / the launch time measured here, against the 157.3 TFLOP/s of the
        # fp32 vector unit WITH packed issue (MI355X_MICROARCH.md).  Beside it the instruction view: wave-level VALU
        # instructions per launch (PMC count of the committed profile, when it belongs to these sources) / the same time,
        # against the issue rates a fp32 chain reaches on THIS device in the same launch shape (upside_hip_calibrate_valu, a
        # known-instruction-count kernel run now): `peak_scalar` for a dependent scalar chain, `peak` (= the packed ceiling,
        # four independent chains compiled to v_pk_fma_f32) for packed issue.  The HBM view of the kernel is kept under "hbm".
        ig_rows = [r for r in rows if r[0].startswith('igraph')]
        ig = next((r for r in ig_rows if r[0] == 'igraph_bwd:rotamer'), max(ig_rows, key=lambda r: r[1]))
        rates = (ct.c_double * 2)()
        check(c, c.upside_hip_calibrate_valu(rates), 'calibrate_valu')
        per_pair = profiled(ig[0], 'syn300_10A', 4096, 'valu_insts_per_pair')
        pairs = ig[4] / ig[2]
        insts = per_pair * pairs if per_pair else None
        ig_ms = ig[1] / ig[2]
        FLOP_PER_PAIR, FP32_VECTOR_PEAK_TF = 350.0, 157.3
        achieved_tf = pairs * FLOP_PER_PAIR / (ig_ms * 1e-3) / 1e12
        achieved_valu = insts / (ig_ms * 1e-3) / 1e9 if insts else None
        isa = None
        try:      # static share of packed instructions in the kernels' loop bodies (tools/isa_summary.py, build container)
            with open(os.path.join(ROOT, 'profiles', 'isa_pk_share.json')) as f:
                isa = json.load(f)
        except (OSError, ValueError):
            pass
        # north_star's own criterion for this kernel, reported as stated: achieved HBM traffic (calibrated counters) / time against the 8 TB/s peak,
        # target 0.5.  The arithmetic view stays beside it as `valu` -- flop rate, instruction rate, and the measured share of the time the
        # vector unit was issuing (PMC pass) -- because that is what bounds the kernel; whether the target is met is
        # for the reader to judge from both.
        ig_hbm = entry(ig)
        valu_busy = profiled(ig[0], 'syn300_10A', 4096, 'valu_busy')
        roofline['igraph'] = dict(bound='hbm', kernel=ig[0], achieved=ig_hbm.get('achieved'), peak=HBM_PEAK_GBS, unit='GB/s', frac=ig_hbm.get('frac'),
                                  traffic=ig_hbm.get('traffic'), traffic_source=ig_hbm.get('traffic_source'), avg_launch_ms=ig_ms,
                                  algorithmic_bytes_per_launch=ig_hbm.get('algorithmic_bytes_per_launch'), frac_model=ig_hbm.get('frac_model'),
                                  target_frac=0.5, target_of='HBM peak (north_star: >= 50 % of the HBM roofline on the interaction-graph kernel)',
                                  pair_evaluations_per_launch=pairs,
                                  valu=dict(bound='valu', achieved=achieved_tf, peak=FP32_VECTOR_PEAK_TF, unit='TFLOP/s', frac=achieved_tf / FP32_VECTOR_PEAK_TF,
                                            flop_per_pair=FLOP_PER_PAIR,
                                            valu_busy=valu_busy,
                                            valu_busy_note='share of the time a SIMD was issuing vector instructions: SQ_ACTIVE_INST_VALU / 1024 SIMDs / '
                                                           '(SQ_WAVE_CYCLES / resident waves), ' + PROFILE_NOTE + '; every wave64 VALU instruction, packed or '
                                                           'not, holds its SIMD for one quad-cycle',
                                            note='22 flop per algorithmic byte: at 100 % of the fp32 vector peak this kernel would move its bytes at ~48 % of the HBM peak',
                                            hbm_frac_at_fp32_peak=(ig[3] / ig[2]) / (pairs * FLOP_PER_PAIR / (FP32_VECTOR_PEAK_TF * 1e12)) / 1e9 / HBM_PEAK_GBS,
                                            issue=dict(achieved=achieved_valu, unit='G wave-instr/s', peak=rates[1] / 1e9, peak_scalar=rates[0] / 1e9,
                                                       frac=(achieved_valu / (rates[1] / 1e9)) if achieved_valu else None,
                                                       frac_of_scalar_ceiling=(achieved_valu / (rates[0] / 1e9)) if achieved_valu else None,
                                                       valu_insts_per_launch=insts, valu_insts_per_pair=per_pair,
                                                       insts_source=('pair evaluations counted in this run x instructions per evaluation ' + PROFILE_NOTE) if insts
                                                       else ('stale or missing profiles/hbm_traffic.json' if _TABLE_STATE.get('stale') else None),
                                                       peak_note='peak = four independent fp32 FMA chains per lane (v_pk_fma_f32), peak_scalar = one dependent '
                                                                 'chain; one 1024-lane workgroup per CU, both measured in this run'),
                                            packed_instruction_share=isa))
        roofline['kernels'] = {r[0]: dict(avg_ms=r[1] / r[2], launches=r[2],
                                          GBps=(r[3] / r[2]) / (r[1] / r[2] * 1e-3) / 1e9 if r[3] else None,
                                          pair_evaluations=(r[4] / r[2]) if r[4] else None) for r in rows}

    _trace('roofline leg done')
    # single-system latency of the same workload (one replica on the GPU), outside the timed region
    single = None
    if rank == 0 and not args.no_single_system:
        eng1 = c.upside_hip_construct(n_atom, fixture.encode(), 1, True)
        if eng1:
            p1 = np.ascontiguousarray(pos0[None].astype('f4')); t1 = np.full(1, TEMPERATURE, dtype='f4')
            check(c, c.upside_hip_set_pos(eng1, p1.ctypes.data), 'set_pos')
            check(c, c.upside_hip_init_md(eng1, t1.ctypes.data, 7, 5.0, DT, 1), 'init_md')
            check(c, c.upside_hip_run_steps(eng1, 150), 'run_steps')
            ts = time.perf_counter()
            check(c, c.upside_hip_run_steps(eng1, 600), 'run_steps')
            single = 600 / (time.perf_counter() - ts)
            lib.calc.free_deriv_engine(ct.c_void_p(eng1))

    # the strong-scaling workloads: what eight GPUs would deliver if each ran its eighth of the systems at the rate ONE GPU reaches
    # with that many (measured here, plain MD) -- so that the first run on an 8-GPU node surprises nobody
    projected = None
    if rank == 0 and total_systems and world == 1 and total_systems % 8 == 0:
        r8 = total_systems // 8
        eng8 = c.upside_hip_construct(n_atom, fixture.encode(), r8, True)
        if eng8:
            p8 = np.ascontiguousarray(pos[:r8]); t8 = np.ascontiguousarray(temps[:r8])
            check(c, c.upside_hip_set_pos(eng8, p8.ctypes.data), 'set_pos')
            check(c, c.upside_hip_init_md(eng8, t8.ctypes.data, rep.system_seed(1000, 0), 5.0, DT, 1), 'init_md')
            check(c, c.upside_hip_run_steps(eng8, 90), 'run_steps')
            ts = time.perf_counter()
            check(c, c.upside_hip_run_steps(eng8, 300), 'run_steps')
            rate8 = r8 * 300 / (time.perf_counter() - ts)
            lib.calc.free_deriv_engine(ct.c_void_p(eng8))
            projected = dict(value=8 * rate8, unit='system-steps/s', systems_per_gpu=r8, one_gpu_rate_at_that_size=rate8,
                             basis='8 x the rate of ONE GPU holding total / 8 systems, measured in this run (plain MD: exchange traffic and '
                                   'straggling not included); NOT a measurement on 8 GPUs')

    _trace('single-system leg done')
    if rank == 0:
        if describe is None:
            describe = ('%s: 300-res synthetic protein, full side-chain BP, 10 A pair list (BASELINE.json configs[2]); %d independent '
                        'replicas per GPU, T=%.1f, dt=%.3f, Langevin thermostat every round' % (args.workload, R, TEMPERATURE, DT)
                        if args.workload == 'syn300_10A' else
                        '%s: %d independent replicas per GPU, T=%.1f, dt=%.3f, Langevin thermostat every round' % (args.workload, R, TEMPERATURE, DT))
        cfg = dict(workload=describe, replicas_per_gpu=R, n_atom=int(n_atom), per_system_steps_per_s=steps_done / elapsed,
                   # the reference's own unit of simulated time (it defines no ns/day, README.md:173-177): steps/s x dt x 86400
                   sim_time_units_per_day_per_system=steps_done / elapsed * DT * 86400.,
                   single_system_steps_per_s=single,
                   engine_hbm_gib=engine_bytes / 2.**30, engine_hbm_mib_per_system=engine_bytes / 2.**20 / R,
                   settle_steps=args.settle)
        if remd:
            cfg.update(exchange_every_steps=exchange_steps_saved, exchange_attempts_timed=timed_attempts, swap_sets=len(swap_sets),
                       exchange='RCCL: ncclAllGather of one fp32 per replica, device Metropolis, ncclSend/ncclRecv of straddling pairs')
        res = dict(metric='MD steps/sec (force evals/sec) per 300-res protein' if fix_name.startswith('syn300') else
                   'MD steps/sec (force evals/sec) per protein', value=value, unit='system-steps/s',
                   n_gpus=world, steps=steps_done, warmup=args.warmup, ms_per_step=elapsed / steps_done * 1e3,
                   higher_is_better=True, scaling='strong' if total_systems else 'weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=cfg, roofline=roofline)
        if parity is not None:
            res['parity_check'] = parity
        if projected is not None:      # (outside `config`: nothing downstream should take it for a scaling measurement)
            res['not_measured_projection_8gpu'] = projected
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N=1 only (rank 0 has the host to itself)
            res['cpu_baseline'] = cpu_baseline(fixture, variant, one_core=(R == 1), n_atom=int(n_atom))
        # (flushed at once: tearing down the RCCL communicator below has been seen to end the process without running Python's
        #  exit-time flush of a buffered stdout)
        sys.stdout.flush()
        print('\n' + json.dumps(res), flush=True)      # (own line even if a C library left an unterminated line on fd 1)
    if remd:
        c.upside_hip_comm_free(eng)
    lib.calc.free_deriv_engine(ct.c_void_p(eng))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity['ok']:
        sys.stderr.write('bench.py: PARITY FAILURE of the timed engine: %r\n' % (parity,))
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
