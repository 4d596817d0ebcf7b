#!/bin/sh
# Stand-in for `tabix`, used ONLY to drive the reference binary.  Test infrastructure.
#   * `system(tabix)` with no arguments must return 256 (exit status 1)  (/root/reference/mipgen.cpp:919-924)
#   * `tabix <vcf> <queries...> > project.local_snp_data.vcf`           (/root/reference/mipgen.cpp:927)
#     -> we ignore the region queries and stream the whole (plain-text) VCF.
if [ $# -eq 0 ]; then exit 1; fi
cat "$1"
