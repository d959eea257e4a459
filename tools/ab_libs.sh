#!/bin/bash
# A/B of two builds of libupside_hip.so inside ONE gpurun call (boxes differ by up to 10 %, so numbers from different
# calls do not compare).  Before the call: build the baseline elsewhere (e.g. `git archive HEAD | tar -x -C /tmp/old &&
# make -C /tmp/old/upside-md_amd/csrc`) and copy its library to upside-md_amd/csrc/libupside_hip.old (untracked; it
# travels with the snapshot).  Usage on the box:  bash tools/ab_libs.sh [bench.py arguments]
L=upside-md_amd/csrc
[ -f $L/libupside_hip.old ] || { echo "no $L/libupside_hip.old"; exit 1; }
cp $L/libupside_hip.so $L/libupside_hip.new
for i in 1 2; do
  for v in old new; do
    cp $L/libupside_hip.$v $L/libupside_hip.so
    echo -n "$v: "
    python bench.py --steps 45 --warmup 15 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'system-steps/s, roofline.frac', round(d['roofline']['frac'], 3))"
  done
done
cp $L/libupside_hip.new $L/libupside_hip.so
