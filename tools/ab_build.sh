#!/bin/bash
# usage: tools/ab_build.sh <file.hip> [git-rev]   -> _lib/libmodex_A.so = current tree with <file.hip> taken from git-rev (default HEAD)
# A/B runs on the SAME box: MODEX_HIP_LIB=$PWD/mod_extraction_amd/_lib/libmodex_A.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
f=$1; rev=${2:-HEAD}
tmp=/tmp/ab_$(basename $f)
git show $rev:mod_extraction_amd/csrc/$f > mod_extraction_amd/csrc/.ab_tmp.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I include -c mod_extraction_amd/csrc/.ab_tmp.hip -o /tmp/ab_obj.o
rm -f mod_extraction_amd/csrc/.ab_tmp.hip
objs=$(ls mod_extraction_amd/_lib/obj/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mod_extraction_amd/_lib/libmodex_A.so $objs /tmp/ab_obj.o
echo built mod_extraction_amd/_lib/libmodex_A.so
