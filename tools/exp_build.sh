#!/bin/bash
# build a variant of the library with extra compiler flags: tools/exp_build.sh <tag> [flags...]  ->  upside-md_amd/csrc/exp/<tag>.so
set -e
L=upside-md_amd/csrc
tag=$1; shift
mkdir -p $L/exp
make -C $L -s clean
make -C $L -s -j8 EXTRA="$*" 2>&1 | grep -E "error" || true
cp $L/libupside_hip.so $L/exp/$tag.so
echo "built $tag"
