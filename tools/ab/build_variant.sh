#!/bin/bash
# build a variant of the library with extra compiler flags:   tools/ab/build_variant.sh NAME [flags...]   -> tools/ab/lib_NAME.so
R=$(cd $(dirname $0)/../.. && pwd); N=$1; shift
cd $R/strling_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o $R/tools/ab/lib_$N.so score.hip pair.hip sort.hip cluster.hip bgzf.hip front.hip comm.hip host_logic.cpp call_logic.cpp nim_tables.cpp -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" ; ls -la $R/tools/ab/lib_$N.so
