#!/bin/bash
# tools/exp/mp_exome.sh [N] [exome|regions5k] [logistic|svr|mixed]: a synthetic design through the in-process command line (tools/cli_exome.py writes the inputs
# and runs `mipgen`), then the same design through mipgen_amd/mp_design.py with two ranks sharing the box's GPU over gloo: same picked file?
N=${1:-20000}; CFG=${2:-exome}; METHOD=${3:-logistic}
R=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/mp_exome
python3 $R/tools/cli_exome.py $N $W $CFG $METHOD 2>&1 | grep -E "wall|picked"
mv $W/out.picked_mips.txt /tmp/picked_one.txt
if [ $CFG = regions5k ]; then CAP="-min_capture_size 120 -max_capture_size 250"; else CAP="-min_capture_size 150 -max_capture_size 170"; fi
cd $W
T0=$(date +%s.%N); python3 -m mipgen_amd.mp_design --gpus 2 --backend gloo --share-gpus --mipgen-path $W/mipgen -- -regions_to_scan $W/exome.bed -project_name out \
    $CAP -bwa_genome_index $W/genome/index.fa -genome_dir $W/genome -score_method $METHOD -silent_mode on -gpu_copy_counter on 2>&1 | grep -E "^\{|rror" | cut -c1-700; T1=$(date +%s.%N); python3 -c "print('mp_design wall %.1f s' % ($T1 - $T0))"
cmp /tmp/picked_one.txt $W/out.picked_mips.txt && echo "picked files identical ($(wc -l < $W/out.picked_mips.txt) lines)"
