#!/bin/bash
# tools/exp/build_variant.sh NAME [-DMACRO=..]...: a scratch build of libmipgen_accel.so with extra macros on kernels_svr.hip / accel*.hip
# (tools/exp/scratch/libmipgen_accel_NAME.so), for A/B timing with tools/exp/kernel_time.py.  The product objects are reused for the rest.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
S=$R/tools/exp/scratch; mkdir -p $S
cd $R/mipgen_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c kernels_svr.hip -o $S/kernels_svr_$N.o -save-temps=obj 2> $S/build_$N.log || { tail -30 $S/build_$N.log; exit 1; }
for f in accel accel_tiles accel_score accel_kmer; do /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o $S/${f}_$N.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $S/libmipgen_accel_$N.so $S/accel_$N.o $S/accel_tiles_$N.o $S/accel_score_$N.o $S/accel_kmer_$N.o $S/kernels_svr_$N.o kernels_logistic.o kernels_misc.o kernels_replay.o kernels_kmer.o \
    kernels_logistic_dense.o kernels_format.o kernels_svr_gemm.o kernels_window.o kernels_skip.o
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" $S/kernels_svr-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - | sed -e 's/_Z11k_svr_denseILi\([0-9]*\)E[A-Za-z0-9_]*/dense<\1>/' -e 's/  */ /g'
rm -f $S/*.bc $S/*.hipi $S/*.out* $S/*.hipfb $S/*-host-*
