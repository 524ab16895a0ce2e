#!/bin/bash
# A second build of libcastrec.so in which ONE source is compiled with other flags (diagnostics: tools/diag_repro2.py CASTREC_LIB=...).
#   tools/probes/build_variant.sh <name> <source.hip> [extra hipcc flags...]   ->  tools/probes/variants/libcastrec_<name>.so
# The other objects are the production ones (csrc/build/*.o: run python -m castrec_amd.build first).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$root/context-aware-sequential-recommendation_amd/csrc
out=$root/tools/probes/variants; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I $root/include -I $csrc -Wall -Wno-unused-function "$@" -x hip -c $csrc/$src -o $out/$src.$name.o
objs=$(ls $csrc/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcastrec_$name.so $objs $out/$src.$name.o -lpthread
echo $out/libcastrec_$name.so
