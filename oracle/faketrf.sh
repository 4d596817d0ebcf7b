#!/bin/sh
# Stand-in for Tandem Repeats Finder, used ONLY to drive the reference binary.  Test infrastructure.
#   * `system(trf)` with no arguments must return 65280 (exit status 255)    (/root/reference/mipgen.cpp:152-160)
#   * `trf <project>.feature_sequences.fa 2 7 7 80 10 14 100 -m -h` must leave
#     `<basename>.feature_sequences.fa.2.7.7.80.10.14.100.mask` in the CWD     (/root/reference/mipgen.cpp:1049-1054)
# Masking rule (replicated by mipgen_amd/synth.py: shim_mask): in record r (0-based), base offset o is
# replaced by N when ((o / 8) * 131 + r * 17) % 23 == 0, i.e. 8-base blocks.
if [ $# -eq 0 ]; then exit 255; fi
fa="$1"
out="$(basename "$fa").2.7.7.80.10.14.100.mask"
awk '
  /^>/ { r++; base = 0; print; next }
  { s = $0; o = ""; n = length(s)
    for (i = 1; i <= n; i++) {
      off = base + i - 1
      if (((int(off / 8)) * 131 + (r - 1) * 17) % 23 == 0) o = o "N"; else o = o substr(s, i, 1)
    }
    base += n
    print o }' "$fa" > "$out"
exit 0
