#!/bin/bash
# GPU box: belief-propagation kernel times (rocprofv3 kernel trace) of the library variants under upside-md_amd/csrc/exp/*.so
L=upside-md_amd/csrc
cp $L/libupside_hip.so $L/exp/_keep.so
for f in $L/exp/*.so; do t=$(basename $f .so); [ "$t" = "_keep" ] && continue; cp $f $L/libupside_hip.so; echo "== $t"; bash tools/bp_kernels_time.sh $t 2>&1 | tail -6; done
cp $L/exp/_keep.so $L/libupside_hip.so
