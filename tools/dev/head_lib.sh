#!/bin/bash
# Developer tool (build container): ab_libs/lib_head.so = the library as a committed revision builds it (default HEAD), for
# same-box A/Bs of changes that span several files (tools/ab_bench.sh swaps ONE file).  The library travels to the GPU box
# with the snapshot; there: DMZ_HIP_LIB=$PWD/ab_libs/lib_head.so python tools/dev/pipe_ab.py ...
# usage: tools/dev/head_lib.sh [revision]
set -e
cd "$(dirname "$0")/../.."
REV=${1:-HEAD}
T=$(mktemp -d /tmp/headlib.XXXXXX)
git archive $REV card.io-dmz_amd/Makefile card.io-dmz_amd/csrc include | tar -x -C $T
mkdir -p $T/card.io-dmz_amd/weights ab_libs
cp card.io-dmz_amd/csrc/weights_blob.o $T/card.io-dmz_amd/csrc/   # (embeds the in-tree path of the weights file)
touch $T/card.io-dmz_amd/weights/dmz_models.bin $T/card.io-dmz_amd/csrc/weights_blob.o
make -s -C $T/card.io-dmz_amd -j4 libdmz_hip.so 2>&1 | grep -v "hip-link\|^make" || true
cp $T/card.io-dmz_amd/libdmz_hip.so ab_libs/lib_head.so
rm -rf $T
echo ab_libs/lib_head.so "($REV)"
