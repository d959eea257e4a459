"""Print the phase clocks of the belief-propagation kernel (needs a GPU). Usage: python tools/bp_trace.py [fixture] [n_system] [md]   (md, with n_system > 1: the clocks of an MD step instead of an energy evaluation)"""
import os, sys
os.environ["UPSIDE_HIP_BP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from upside_md_amd import engine as E, config as C

name = sys.argv[1] if len(sys.argv) > 1 else "syn300_10A"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
up = os.path.join(root, name + ".up")
pos = np.load(os.path.join(root, name + ".coords.npy")).astype(np.float32)
lib = E.default_library()
if S == 1:
    e = E.Upside(up, library=lib)
    for _ in range(3): e.energy(pos)
    t = e.get_value_by_name((32,), "rotamer", "bp_trace")
else:
    c = lib.calc
    import ctypes as ct
    c.upside_hip_construct.restype = ct.c_void_p
    c.upside_hip_construct.argtypes = [ct.c_int, ct.c_char_p, ct.c_int, ct.c_bool]
    for f in (c.upside_hip_set_pos,): f.argtypes = [ct.c_void_p, ct.c_void_p]
    c.upside_hip_compute.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p]
    eng = c.upside_hip_construct(pos.shape[0], up.encode(), S, True)
    allpos = np.ascontiguousarray(np.tile(pos[None], (S, 1, 1)))
    c.upside_hip_set_pos(eng, allpos.ctypes.data)
    en = np.zeros(S, np.float32)
    if len(sys.argv) > 3 and sys.argv[3] == "md":      # the clocks of an MD step (forces only: the marginal loops skip the energy terms)
        c.upside_hip_run_steps.argtypes = [ct.c_void_p, ct.c_int]
        c.upside_hip_init_md.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_uint32, ct.c_float, ct.c_float, ct.c_int]
        temps = np.full(S, 0.8, np.float32)
        assert c.upside_hip_init_md(eng, temps.ctypes.data, 7, 5.0, 0.009, 1) == 0
        assert c.upside_hip_run_steps(eng, 7) == 0
    else:
        for _ in range(3): c.upside_hip_compute(eng, en.ctypes.data, None)
    t = np.zeros(32, np.float32)
    c.get_value_by_name(32, t.ctypes.data, eng, b"rotamer", b"bp_trace")
names = ["prologue", "loop", "epilogue", "edge_phase", "node_phase"]
for n_, v in zip(names, t[:5]): print("%-12s %8.1f us" % (n_, v * 0.01))
print("sweeps %d  n_slot %d  inbox rows %d  class starts %s" % (t[5], t[6], t[7], t[8:14].astype(int)))
print("per sweep: edge %.2f us  node %.2f us" % (t[3] * 0.01 / (t[5] + 1), t[4] * 0.01 / (t[5] + 1)))

sub = ["setup", "layout+pack", "msg init", "fold+nb init", "resident load", "resident marginals", "packed marginals", "energy+beliefs out"]
print("stamps since kernel start (us): " + ", ".join("%s %.1f" % (n_, v * 0.01) for n_, v in zip(sub, t[16:24])))
print("node phase of wavefront 0, per sweep: rows + products %.2f us, combine %.2f us, finish %.2f us (rest of the node phase: waiting at the barrier)" % tuple(t[24 + i] * 0.01 / (t[5] + 1) for i in range(3)))
print("loop end at %.1f us, kernel end at %.1f us" % ((t[0] + t[1]) * 0.01, (t[0] + t[1] + t[2]) * 0.01))
