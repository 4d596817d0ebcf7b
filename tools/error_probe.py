#!/usr/bin/env python3
"""Error-behaviour probe: malformed command lines and inputs through the REAL reference (oracle/_ref/mipgen_ref) and through the drop-in front end, exit
status and the tail of stderr / stdout side by side.  Most cases fail before tile_regions and compare without a GPU; the rest through --save here and --check on the GPU box.

    python3 tools/error_probe.py            # side by side, here (the cases that reach tile_regions need a GPU for the front end)
    python3 tools/error_probe.py --save     # the reference's results -> tests/golden/error_cases.json
    gpurun -- python3 tools/error_probe.py --check
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mipgen_amd import synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def cases(base):
    ok = ["-regions_to_scan", "ok.bed", "-min_capture_size", "150", "-max_capture_size", "160"]
    rng = ["-min_capture_size", "150", "-max_capture_size", "160"]
    return {
        "no_args": [],
        "doc": ["-doc"],
        "missing_required": ["-regions_to_scan", "ok.bed"],
        "unknown_option": base + ok + ["-bogus", "1"],
        "bad_method": base + ok + ["-score_method", "foo"],
        "min_gt_max": base + ["-regions_to_scan", "ok.bed", "-min_capture_size", "170", "-max_capture_size", "160"],
        "no_bed_file": base + ["-regions_to_scan", "nope.bed"] + rng,
        "bad_cols": base + ["-regions_to_scan", "bad_cols.bed"] + rng,
        "bad_num": base + ["-regions_to_scan", "bad_num.bed"] + rng,
        "reversed_interval": base + ["-regions_to_scan", "rev.bed"] + rng,
        "unknown_chromosome": base + ["-regions_to_scan", "nochr.bed"] + rng,
        "empty_bed": base + ["-regions_to_scan", "empty.bed"] + rng,
        "nonint_capture": base + ["-regions_to_scan", "ok.bed", "-min_capture_size", "abc", "-max_capture_size", "160"],
        "bad_bwa": [a if a != os.path.join(ROOT, "oracle", "fakebwa.sh") else "/nonexistent/bwa" for a in base] + ok,
        "odd_args": base + ["-regions_to_scan", "ok.bed", "-min_capture_size", "150", "-max_capture_size"],
        "bad_arm_lengths": base + ok + ["-arm_lengths", "20-22"],
        "bad_tag": base + ok + ["-tag_sizes", "5"],
        "svr_no_model": base + ok + ["-score_method", "svr"],
        "capture_below_arms": base + ["-regions_to_scan", "ok.bed", "-min_capture_size", "40", "-max_capture_size", "44"],
        "no_snp_file": base + ok + ["-snp_file", "nope.vcf.gz", "-tabix", os.path.join(ROOT, "oracle", "faketabix.sh")],
        "no_params_file": base + ok + ["-file_of_parameters", "nope.txt"],
        # real-valued options through boost::lexical_cast<double>: no hex floats, overflow refused, underflow / inf / a leading '+' taken; the masked-arm
        # threshold is only cast at the first design_mip (mipgen.cpp:626): the output files exist by then and stay empty
        "hex_float_score": base + ok + ["-logistic_optimal_score", "0x1p-1"],
        "overflow_score": base + ok + ["-logistic_priority_score", "1e400"],
        "hex_float_threshold": base + ok + ["-masked_arm_threshold", "0x1p-1"],
        "threshold_trailing_space": base + ok + ["-masked_arm_threshold", "0.5 "],
        "denormal_threshold": base + ok + ["-masked_arm_threshold", "1e-320"],
        "inf_plus_scores": base + ok + ["-logistic_optimal_score", "INFINITY", "-logistic_priority_score", "+.5"],
        "hex_int": base + ok + ["-max_mip_overlap", "0x10"],
    }


FILES = ("all_mips.txt", "collapsed_mips.txt", "picked_mips.txt", "snp_mips.txt", "coverage_failed.bed", "double_tile_failed.bed",
         "minus_strand_failed.bed", "minus_strand_double_tile_failed.bed")
STORE = os.path.join(ROOT, "tests", "golden", "error_cases.json")      # committed: the reference's exit status, stderr tail, output line counts + hashes per case


NEEDS_DEVICE = ("reversed_interval", "bad_tag", "no_snp_file", "denormal_threshold", "inf_plus_scores")           # cases that reach tile_regions with candidates to score


def lay_out(work: str) -> list:
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "genome"))
    g = H.golden_genome("genome2_chr1.fa.gz")
    synth.write_fasta(os.path.join(work, "genome", "chr1.fa"), "chr1", g)
    for name, text in (("ok.bed", "chr1\t20000\t20100\ta\n"), ("bad_cols.bed", "chr1\t20000\n"), ("bad_num.bed", "chr1\tabc\t20100\ta\n"),
                       ("rev.bed", "chr1\t20100\t20000\ta\n"), ("nochr.bed", "chr9\t20000\t20100\ta\n"), ("empty.bed", "")):
        with open(os.path.join(work, name), "w") as fh:
            fh.write(text)
    return ["-project_name", "out", "-bwa_genome_index", "genome/index.fa", "-genome_dir", "genome", "-bwa", os.path.join(ROOT, "oracle", "fakebwa.sh")]


def run_one(exe: str, args: list, work: str) -> dict:
    import hashlib
    for f in os.listdir(work):
        if f.startswith("out."):
            os.remove(os.path.join(work, f))
    env = dict(os.environ, FAKEBWA_MODE="unique")
    try:
        p = subprocess.run([exe] + args, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
        rc, out, err = p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    except subprocess.TimeoutExpired:
        rc, out, err = "timeout", "", ""
    files = {}
    for f in FILES:
        path = os.path.join(work, "out." + f)
        if os.path.exists(path):
            with open(path, "rb") as fh:
                data = fh.read()
            if f == "all_mips.txt":
                data = H.normalise_all_mips(data)
            files[f] = [data.count(b"\n"), hashlib.sha256(data).hexdigest()]
    return {"rc": rc, "stderr_tail": [l for l in err.split("\n") if l.strip()][-3:], "stdout_tail": out[-200:], "files": files}


def main() -> None:
    """(no flag): reference and front end side by side, here.  --save: the reference's results into tests/golden/error_cases.json (travels to the
    GPU box).  --check: the front end against that file (on the GPU box: the cases that reach tile_regions need the device)."""
    import json
    mode = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].startswith("--") else ""
    work = "/tmp/mipgen_error_probe"
    base = lay_out(work)
    ours = os.path.join(ROOT, "mipgen_amd", "mipgen")
    skip = {"no_args", "doc", "missing_required", "no_params_file",      # the usage text is this front end's own wording
            "svr_no_model",                                               # the reference segfaults in svm_predict (no model file)
            "capture_below_arms"}                                         # the reference's scan size goes negative: records whose scan target is the rest of the region
    if mode == "--save":
        shutil.copy(os.path.join(ROOT, "oracle", "_ref", "mipgen_ref"), os.path.join(work, "mipgen_ref"))
        res = {name: run_one("./mipgen_ref", args, work) for name, args in cases(base).items() if name not in skip}
        os.makedirs(os.path.dirname(STORE), exist_ok=True)
        with open(STORE, "w") as fh:
            json.dump(res, fh, indent=1)
        print(f"{len(res)} cases saved to {STORE}")
        return
    if mode == "--check":
        with open(STORE) as fh:
            want = json.load(fh)
        bad = 0
        for name, args in cases(base).items():
            if name not in want:
                continue
            got = run_one(ours, args, work)
            same = got == want[name]
            bad += not same
            print(f"== {name}: {'identical (exit status ' + str(got['rc']) + ', stderr tail, files ' + str(sorted(got['files'])) + ')' if same else 'DIFFERS'}")
            if not same:
                for k in ("rc", "stderr_tail", "stdout_tail", "files"):
                    if got[k] != want[name][k]:
                        print(f"   {k}: ref {want[name][k]}\n   {' ' * len(k)}  ours {got[k]}")
        print(f"{bad} of {len(want)} cases differ")
        raise SystemExit(1 if bad else 0)
    shutil.copy(os.path.join(ROOT, "oracle", "_ref", "mipgen_ref"), os.path.join(work, "mipgen_ref"))
    n_diff = 0
    for name, args in cases(base).items():
        r, o = run_one("./mipgen_ref", args, work), run_one(ours, args, work)
        n_diff += r != o
        print(f"== {name}: exit status ref {r['rc']}, ours {o['rc']}{'' if r == o else '    <<< differs'}")
        for k in ("stderr_tail", "stdout_tail", "files"):
            if r[k] != o[k]:
                print(f"   {k}: ref  {r[k]}\n   {' ' * len(k)}  ours {o[k]}")
    print(f"{n_diff} of {len(cases(base))} cases differ")


if __name__ == "__main__":
    main()
