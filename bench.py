#!/usr/bin/env python3
"""Benchmark of the MD inner loop (force pass + leapfrog/OU integrator) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--replicas R] [--workload NAME]

A "step" is one force evaluation + one leapfrog stage of EVERY replica resident on the GPU (the reference's
"MD step", /root/reference/src/main.cpp:677-682; one integration cycle = 3 steps,
/root/reference/src/deriv_engine.cpp:172-192).  The workload is BASELINE.json configs[2]: the 300-residue
synthetic protein with full side-chain belief propagation and the 10 A pair list (tests/golden/syn300_10A.up),
held as R independent replicas per GPU (different thermostat seeds), everything resident in HBM.
`value` = system-steps per second summed over all replicas and GPUs -- the reciprocal of the reference's own
"us/systems/step" figure.  Multi-GPU runs are weak scaling: every rank owns R more replicas; there is no
data-path collective (replicas are independent).

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
    c.upside_hip_last_error.restype = ct.c_char_p
    return c


def check(c, rc, what):
    if rc:
        raise RuntimeError('%s failed: %s' % (what, c.upside_hip_last_error().decode()))


def cpu_baseline(fixture, variant, budget_s=float(os.environ.get('UPSIDE_BENCH_CPU_BUDGET_S', '15'))):
    """the unmodified reference (oracle/_ref, kind "reference") timed on this host's cores over a bounded
    sample; falls back to the C restatement (kind "port", 1 core) when the reference binary is absent."""
    exe = os.path.join(ROOT, 'oracle', '_ref', 'upside_' + variant)
    n_cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    if os.path.exists(exe):
        n_sys = max(1, min(n_cores, 16))
        # ~9 ms per step per core for this workload: size the sample for `budget_s`
        steps = max(30, int(budget_s / 0.010))
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
                return dict(value=1e6 / us * 1.0, unit='system-steps/s', cores=n_sys, kind='reference',
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
    return dict(value=3 * n_round / wall, unit='system-steps/s', cores=1, kind='port',
                sample='%d steps of the C restatement (oracle/upside_oracle.c), 1 core' % (3 * n_round))


def measured_traffic(kernel_label, workload, replicas):
    """HBM bytes per launch of `kernel_label` from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this very command; tools/hbm_traffic.py applies the guide's gfx950
    corrections).  None when no profile exists for this workload / replica count."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
        with open(path) as f:
            tab = json.load(f)
        return tab['%s/R%d' % (workload, replicas)][kernel_label]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--replicas', type=int, default=int(os.environ.get('UPSIDE_BENCH_REPLICAS', '4096')),
                    help='independent replicas resident per GPU')
    ap.add_argument('--workload', default='syn300_10A')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    pkg = load_package()
    rep = pkg.replicas
    rank, local_rank, world = rep.world_from_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    lib = pkg.default_library()          # raises if the HIP extension is missing: no fallback
    c = bind(lib)
    check(c, c.upside_hip_set_device(local_rank), 'set_device')

    fixture = os.path.join(ROOT, 'tests', 'golden', args.workload + '.up')
    variant = '10A' if args.workload.endswith('10A') else '7A'
    pos0 = pkg.config.read_pos(fixture)
    n_atom = pos0.shape[0]
    R = args.replicas
    eng = c.upside_hip_construct(n_atom, fixture.encode(), R, True)
    if not eng:
        raise RuntimeError('engine construction failed: %s' % c.upside_hip_last_error().decode())
    pos = np.ascontiguousarray(np.tile(pos0[None], (R, 1, 1)).astype('f4'))
    check(c, c.upside_hip_set_pos(eng, pos.ctypes.data), 'set_pos')
    temps = np.full(R, TEMPERATURE, dtype='f4')
    # every replica of every rank gets its own thermostat stream: seed = base + GLOBAL replica index (main.cpp:459)
    first, _ = rep.weak_shard(R, world, rank)
    check(c, c.upside_hip_init_md(eng, temps.ctypes.data, rep.system_seed(1000, first), 5.0, DT, 1), 'init_md')

    def run_steps(n):
        check(c, c.upside_hip_run_steps(eng, n), 'run_steps')     # exactly n force evaluations; returns after the stream has drained
        return n

    def barrier():
        rep.barrier(dist, torch.cuda.synchronize)

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    steps_done = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    value, elapsed = rep.job_throughput(dist, R * steps_done, elapsed, device='cuda')   # all ranks' units / slowest rank

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
            nm, ms, n, by = ln.split()
            rows.append((nm, float(ms), int(n), float(by)))
        def entry(r):
            avg_ms = r[1] / r[2]
            bytes_per_launch = r[3] / r[2]
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            return dict(bound='hbm', kernel=r[0], achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(r[0], args.workload, R), avg_launch_ms=avg_ms,
                        algorithmic_bytes_per_launch=bytes_per_launch)
        # the dominant kernel of the step (most time): belief propagation, which streams the pair matrices and messages
        # every sweep and IS bandwidth bound; the dominant interaction-graph pair kernel (VALU bound) is
        # reported beside it because north_star's target names it
        dom = max(rows, key=lambda r: r[1])
        roofline = entry(dom)
        roofline['igraph'] = entry(max([r for r in rows if r[0].startswith('igraph')], key=lambda r: r[1]))
        roofline['kernels'] = {r[0]: dict(avg_ms=r[1] / r[2], launches=r[2],
                                          GBps=(r[3] / r[2]) / (r[1] / r[2] * 1e-3) / 1e9 if r[3] else None) for r in rows}

    if rank == 0:
        res = dict(metric='MD steps/sec (force evals/sec) per 300-res protein', value=value, unit='system-steps/s',
                   n_gpus=world, steps=steps_done, warmup=args.warmup, ms_per_step=elapsed / steps_done * 1e3,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='%s: 300-res synthetic protein, full side-chain BP, 10 A pair list '
                                        '(BASELINE.json configs[2]); %d independent replicas per GPU, T=%.1f, dt=%.3f, '
                                        'Langevin thermostat every round' % (args.workload, R, TEMPERATURE, DT),
                               replicas_per_gpu=R, n_atom=int(n_atom), per_system_steps_per_s=steps_done / elapsed),
                   roofline=roofline)
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N=1 only (rank 0 has the host to itself)
            res['cpu_baseline'] = cpu_baseline(fixture, variant)
        print(json.dumps(res))
    lib.calc.free_deriv_engine(ct.c_void_p(eng))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
