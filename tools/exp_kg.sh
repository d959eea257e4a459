#!/bin/bash
# variant of the library that differs in kernels_graph.o only (pair / list / belief-propagation kernels):
#   tools/exp_kg.sh <tag> [flags...]  ->  upside-md_amd/csrc/exp/<tag>.so     (compare on the box with tools/exp_cmp.sh)
set -e
L=upside-md_amd/csrc
tag=$1; shift
mkdir -p $L/exp
make -C $L -s -j8 2>&1 | grep -E "error" || true
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -I/opt/conda/include "$@" -c $L/kernels_graph.hip -o /tmp/kg_$tag.o 2>&1 | grep -E "error" || true
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/exp/$tag.so $L/kernels_basic.o /tmp/kg_$tag.o $L/engine.o $L/nodes.o $L/engine_c_api.o $L/main_cli.o $L/comm_rccl.o -L/opt/conda/lib -lhdf5 -ldl -Wl,-rpath,/opt/conda/lib
echo "built $tag"
