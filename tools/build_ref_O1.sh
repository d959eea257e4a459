#!/bin/bash
# Throw-away builds of the UNMODIFIED reference at -O1 without fast-math (build container only; outputs under /tmp), used by
# tools/noise_floor.py to measure how far the reference is from ITSELF between builds -- the floor under every comparison
# with its golden vectors.  Same recipe as oracle/Makefile except for the optimisation flags.
set -e
REF=/root/reference/src
SRC="main environment hbond rotamer placement rama_map_pot spline deriv_engine sidechain_radial backbone_steric bonds eig membrane_potential timing thermostat h5_support state_logger monte_carlo_sampler engine_c_library"
for v in 7A 10A; do
  out=/tmp/refO1_$v; mkdir -p $out
  for s in $SRC; do
    g++ -c -fPIC -fopenmp -std=c++11 -DR123_NO_SINCOS -Drestrict=__restrict__ -DNDEBUG -O1 -fno-fast-math -march=x86-64-v3 -w \
        -isystem $REF/include -I/opt/conda/include -DPARAM_${v}_CUTOFF -DPARAM_DERIV -o $out/$s.o $REF/$s.cpp &
  done
  wait
  g++ -shared -fopenmp -o $out/libupside_O1.so $out/*.o /opt/conda/lib/libhdf5.so.103 /opt/conda/lib/libz.so.1 -Wl,-rpath,/opt/conda/lib -Wl,--allow-shlib-undefined
done
ls -la /tmp/refO1_*/libupside_O1.so
