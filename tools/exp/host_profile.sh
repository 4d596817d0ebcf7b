#!/bin/bash
# tools/exp/host_profile.sh [N = 200000] [logistic|svr] [extra mipgen flags]: the HOST side of an exome-scale silent design (input stage, block dealing, selection
# stage) on a machine WITHOUT a GPU: the command line of the uninstrumented stub build (make -C tests/stub_accel SAN=none) with STUB_ACCEL_FAKE=1 -
# the "accelerator" fabricates a survivor per scan position and strand instead of scoring.  Timings from -gpu_timing on (stage by stage down to the five parts of the pick stage).
# DEVICES=n: n stub devices + -gpus n in the extra flags exercises the block dealing.
R=$(cd "$(dirname "$0")/../.." && pwd)
N=${1:-200000}; M=${2:-logistic}; shift; shift
make -s -j4 -C $R/tests/stub_accel SAN=none || exit 1
STUB_ACCEL_FAKE=1 STUB_ACCEL_DEVICES=${DEVICES:-1} MIPGEN_CLI_BIN=$R/tests/stub_accel/_build/none/mipgen python3 $R/tools/cli_exome.py $N /tmp/mipgen_host_profile exome $M "$@"
