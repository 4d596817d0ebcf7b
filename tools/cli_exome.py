#!/usr/bin/env python3
"""End-to-end run of the drop-in `mipgen` command line on the synthetic exome of BASELINE configs[3]: writes the 24
chromosome FASTA files (300 Mb) and a BED with the first N of the 200,000 exon-like intervals, then times

    mipgen -regions_to_scan exome.bed -genome_dir genome/ -min_capture_size 150 -max_capture_size 170 -score_method svr
           -silent_mode on -gpu_copy_counter on ...

(no bwa: the arm copy numbers come from the GPU k-mer counter; a 1,024-SV synthetic model beside the executable).

    python3 tools/cli_exome.py [N] [workdir] [exome|regions5k] [svr|logistic|mixed] [extra mipgen flags ...]      e.g. -gpu_gather rccl
"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mipgen_amd import synth, workloads  # noqa: E402


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/mipgen_cli_exome"
    config = sys.argv[3] if len(sys.argv) > 3 else "exome"          # or "regions5k": configs[2], 1,000 x 5 kb, capture 120-250, mixed scoring
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "genome"))
    t0 = time.time()
    if config == "regions5k":
        g5 = workloads.regions5k_genome()
        ivs = workloads.regions5k_intervals(min(n, 1000))
        chroms = ["1"]
        synth.write_fasta(os.path.join(work, "genome", "chr1.fa"), "chr1", g5)
        capture, method = ("120", "250"), "mixed"
    else:
        chrom_len, ivs = workloads.exome_layout()
        ivs = ivs[:n]
        chroms = sorted({iv.chrom for iv in ivs})
        for c in chroms:
            synth.write_fasta(os.path.join(work, "genome", f"chr{c}.fa"), "chr" + c, workloads.exome_chromosome(c, chrom_len[c]))
        capture, method = ("150", "170"), "svr"
    if len(sys.argv) > 4:
        method = sys.argv[4]
    synth.write_bed(os.path.join(work, "exome.bed"), ivs)
    exe = os.path.join(work, "mipgen")
    # MIPGEN_CLI_BIN: another build of the command line (tools/exp/host_profile.sh: the uninstrumented stub build of tests/stub_accel with STUB_ACCEL_FAKE=1,
    # which times the HOST side of an exome-scale design on a machine without a GPU)
    os.symlink(os.environ.get("MIPGEN_CLI_BIN", os.path.join(ROOT, "mipgen_amd", "mipgen")), exe)
    model = workloads.svr_model_path(os.path.join(work, "cache"), workloads.practice62()[0], 1024, rho=workloads.MODEL_RHO["regions5k" if config == "regions5k" else "exome"])
    shutil.copy(model, os.path.join(work, "mipgen_svr.model"))
    print(f"inputs: {len(ivs)} intervals on {len(chroms)} chromosomes written in {time.time() - t0:.1f} s", flush=True)
    argv = [exe, "-regions_to_scan", os.path.join(work, "exome.bed"), "-project_name", "out", "-min_capture_size", capture[0], "-max_capture_size", capture[1],
            "-bwa_genome_index", os.path.join(work, "genome", "index.fa"), "-genome_dir", os.path.join(work, "genome"), "-score_method", method,
            "-silent_mode", "on", "-gpu_copy_counter", "on"]
    t1 = time.time()
    p = subprocess.run(argv + ["-gpu_timing", "on"] + sys.argv[5:], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t1
    print("rc", p.returncode, f"wall {dt:.1f} s")
    print("".join(l + "\n" for l in p.stderr.decode().split("\n") if "timing" in l or "[mipgen_accel]" in l))
    if p.returncode != 0:
        print(p.stderr.decode()[-2000:])
        raise SystemExit(1)
    picked = sum(1 for _ in open(os.path.join(work, "out.picked_mips.txt"))) - 1
    print(f"{picked} picked MIPs for {len(ivs)} intervals; progress tail:")
    print("".join(open(os.path.join(work, "out.progress.txt")).readlines()[-3:]))


if __name__ == "__main__":
    main()
