import os, sys, ctypes as ct
sys.path.insert(0, '/root/repo')
os.environ['UPSIDE_HIP_PRINT_SCHEDULE'] = '1'
import bench
from __graft_entry__ import load_package
pkg = load_package(); c = bench.bind(pkg.default_library())
c.upside_hip_set_device(0)
fixture = os.path.join('/root/repo', 'tests', 'golden', 'syn300_10A.up')
e = c.upside_hip_construct(900, fixture.encode(), 1024, True)
print('engine', bool(e))
