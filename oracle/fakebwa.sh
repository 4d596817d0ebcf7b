#!/bin/sh
# Stand-in for the external `bwa` aligner, used ONLY to drive the reference binary
# (oracle/_ref/mipgen_ref) in tests and in bench.py's cpu_baseline leg.  Test infrastructure.
#
# What the reference expects of bwa (citations into /root/reference/mipgen.cpp):
#   * `system(bwa)` with no arguments must return 256 (exit status 1)          :146-151
#   * `bwa aln ...  > x.sai` may produce anything                               :560,841
#   * `bwa samse idx x.sai x.fq > x.sam` must print one SAM line per read whose first
#     field is the read name and which carries X0:i:<n> / X1:i:<m> tags          :566-594,850-868
#
# Copy numbers are a deterministic function of the read name so that the Python side
# (mipgen_amd/synth.py: shim_copy / shim_unmappable) can rebuild the same tables:
#   arm oligo  "chr<c>:<start>-<stop>"  -> copy = f(start, stop-start+1)
#   capture    "<size>_<c>_<pos>"       -> unique unless g(pos,size) == 0
# FAKEBWA_MODE=unique (default) gives copy 1 / always mappable; FAKEBWA_MODE=hashed varies them;
# FAKEBWA_MODE=blocks puts a 300-base zone of copy-500 oligos into every 3000 bases.
if [ $# -eq 0 ]; then exit 1; fi
case "$1" in
  aln) exit 0 ;;
  samse)
    fq="$4"
    mode="${FAKEBWA_MODE:-unique}"
    awk -v mode="$mode" '
      BEGIN { print "@SQ\tSN:synthetic\tLN:1" }
      NR % 4 == 1 {
        name = substr($0, 2)
        if (name ~ /^chr/) {
          # arm oligo: chr<c>:<start>-<stop>
          i = index(name, ":"); rest = substr(name, i + 1)
          j = index(rest, "-"); start = substr(rest, 1, j - 1) + 0; stop = substr(rest, j + 1) + 0
          len = stop - start + 1
          copy = 1; tag = 1
          if (mode == "hashed") {
            h = (start * 7919 + len * 104729) % 1000
            if (h < 940) copy = 1
            else if (h < 970) copy = 2 + (h % 19)
            else if (h < 985) copy = 21 + (h % 60)
            else if (h < 995) copy = 101 + (h % 400)
            else if (h < 998) tag = 0
            else copy = 0
          }
          if (mode == "blocks") {
            # dead zones: every oligo starting in [1200, 1500) of each 3000-base block is highly repetitive (no MIP can be placed
            # there: the copy product exceeds -max_arm_copy_product, so the pick stage has to open gaps)
            if ((start % 3000) >= 1200 && (start % 3000) < 1500) copy = 500
          }
          if (tag) printf "%s\t0\tsynthetic\t1\t37\t%dM\t*\t0\t0\t*\t*\tXT:A:U\tX0:i:%d\tX1:i:0\n", name, len, copy
          else     printf "%s\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n", name
        } else {
          # capture window: <size>_<c>_<pos>
          n = split(name, f, "_"); size = f[1] + 0; pos = f[n] + 0
          x0 = 1
          if (mode == "hashed" && ((pos * 31 + size * 17) % 211) == 0) x0 = 2
          printf "%s\t0\tsynthetic\t1\t37\t%dM\t*\t0\t0\t*\t*\tXT:A:U\tX0:i:%d\tX1:i:0\n", name, size, x0
        }
      }' "$fq"
    ;;
  *) exit 0 ;;
esac
