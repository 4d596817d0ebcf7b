#!/usr/bin/env python3
"""Differential probe: N random designs (seeded) run through the REAL reference here (oracle/_ref/mipgen_ref, built where /root/reference exists),
their inputs + the reference's output files laid out like the committed goldens under tests/golden_probe/ (git-ignored scratch that travels to the GPU
box with the snapshot), where tests/test_gpu_probe.py runs the drop-in command line on every one of them - with random device-worker / result-window /
gather settings - and compares the files byte for byte.  A way to look for differences the fixed goldens do not reach; what it finds becomes a golden.

    python3 tools/diff_probe.py [N = 40] [seed = 1] [long|extreme|hard|silent]   # (combinable: hardsilent; silent = every design with -silent_mode on) hard = on the hard genome (ambiguity codes, lower case, '-', homopolymers, microsatellites, GC 20 / 70 %); ~ N x 3 s of reference time; extreme = option values at the edges; long = logistic designs of up to ten regions of 0.3-3 kb
                                                             # (the selection stage at length), ~ N x 30 s
    gpurun -- 'MIPGEN_PROBE=1 python -m pytest tests/test_gpu_probe.py -q -n 6'
"""
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402
from mipgen_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden_probe")


def random_design(rng: np.random.Generator, k: int, multi: bool, long_regions: bool = False, extreme: bool = False, hard: bool = False, all_silent: bool = False) -> dict:
    method = "logistic" if long_regions else str(rng.choice(["logistic", "logistic", "svr", "mixed"]))
    inc = int(rng.choice([1, 2, 3, 5, 5, 5, 10]))
    lo = int(rng.integers(100, 200))
    hi = lo + inc * int(rng.choice([0, 1, 2, 2, 4, 6, 10]))
    if rng.random() < 0.6:
        sums, arm_lengths = sorted(int(x) for x in rng.choice(np.arange(38, 49), size=int(rng.integers(1, 5)), replace=False)), None
        n_pairs = sum(max(0, min(s - 18, 30) - max(16, s - 30) + 1) for s in sums)
    else:
        n = int(rng.integers(1, 9))
        pairs = {(int(rng.integers(16, 31)), int(rng.integers(18, 31))) for _ in range(n)}
        pairs = list(pairs)
        rng.shuffle(pairs)
        sums, arm_lengths = None, ",".join(f"{e}:{l}" for e, l in pairs)
        n_pairs = len(pairs)
    arm_extra = []
    if arm_lengths is not None and rng.random() < 0.3:        # both options at once (the lists merge, mipgen.cpp:222-261), a pair given twice
        more = sorted(int(x) for x in rng.choice(np.arange(38, 49), size=int(rng.integers(1, 3)), replace=False))
        arm_extra = ["-arm_length_sums", ",".join(map(str, more))]
        n_pairs += sum(max(0, min(s - 18, 30) - max(16, s - 30) + 1) for s in more)
        if rng.random() < 0.5:
            arm_lengths += "," + arm_lengths.split(",")[0]
            n_pairs += 1
    elif sums is not None and rng.random() < 0.2:             # sums whose lists are empty: keys below the minimum arm lengths / above 30 + 30
        sums = sorted(set(sums) | {int(x) for x in rng.choice([20, 30, 33, 61, 62, 70], size=int(rng.integers(1, 3)), replace=False)})
    n_sizes = (hi - lo) // inc + 1
    # budget: the reference scores ~5e3 SVR candidates / s (64 SVs), ~1e5 logistic ones
    budget = 120000 if method != "logistic" else (6000000 if long_regions else 1500000)
    per_base = 2 * n_sizes * max(n_pairs, 1)
    n_iv = int(rng.integers(1, 11 if long_regions else 7))
    ivs = []
    bed_lines = []
    chroms = ["2", "10", "X"] if multi else (["4"] if hard else ["1"])
    glen = 40000 if multi else (36000 if hard else 80000)      # hard: the zones of synth.hard_genome end at 34,000
    used = 0
    for j in range(n_iv):
        room = (budget - used) // per_base - hi
        if room < 2:
            break
        length = int(min(room, rng.choice([300, 700, 1500, 3000] if long_regions else [1, 3, 20, 60, 150, 400, 1200])))
        if length < 1:
            break
        c = str(rng.choice(chroms))
        start = int(rng.integers(1500, glen - 1500 - length))
        if ivs and rng.random() < 0.25:                       # overlapping / adjacent / duplicate starts on the previous interval's chromosome
            pc, ps, pe, _ = ivs[-1]
            c = pc
            start = int(max(1500, min(glen - 1500 - length, ps + rng.integers(-30, pe - ps + 40))))
        ivs.append((c, start, start + length, f"r{j}"))
        prefix = "chr" if rng.random() < 0.5 else ""
        bed_lines.append(f"{prefix}{c}\t{start}\t{start + length}\tr{j}" if rng.random() < 0.85 else f"{prefix}{c}\t{start}\t{start + length}")
        used += (length + hi) * per_base
    if not multi:
        ivs.sort(key=lambda t: t[1])
    extra = list(arm_extra)
    if inc != 5:
        extra += ["-capture_increment", str(inc)]

    def maybe(p, *opt):
        if rng.random() < p:
            extra.extend(opt)
    maybe(0.15, "-double_tile_strand_unaware", "on")
    maybe(0.15, "-double_tile_strands_separately", "on")
    maybe(0.15, "-seal_both_strands", "on")
    maybe(0.10, "-half_seal_both_strands", "on")
    maybe(0.25, "-max_mip_overlap", str(int(rng.choice([0, 10, 40, 80]))))
    maybe(0.25, "-starting_mip_overlap", str(int(rng.choice([0, 5, 10, 25]))))
    maybe(0.20, "-masked_arm_threshold", str(rng.choice(["0.1", "0.2", "0.75", "1.0"])))
    maybe(0.10, "-logistic_heuristic", "off")
    maybe(0.10, "-check_copy_number", "off")
    maybe(0.15, "-target_arm_copy", str(int(rng.choice([1, 5, 50]))))
    maybe(0.15, "-max_arm_copy_product", str(int(rng.choice([4, 20, 400]))))
    maybe(0.15, "-ext_min_length", str(int(rng.choice([16, 18, 20]))))
    maybe(0.15, "-lig_min_length", str(int(rng.choice([18, 20, 22]))))
    maybe(0.15, "-logistic_priority_score", str(rng.choice(["0.5", "0.8", "0.95"])))
    maybe(0.15, "-logistic_optimal_score", str(rng.choice(["0.9", "0.95", "0.99"])))
    maybe(0.15, "-svr_priority_score", str(rng.choice(["1.0", "1.4", "1.8"])))
    maybe(0.15, "-svr_optimal_score", str(rng.choice(["1.9", "2.4", "3.0"])))
    maybe(0.10, "-stop_optimizing_scores_above", str(rng.choice(["0.9", "1.6"])))
    if extreme:
        # option values at and beyond the edges of what anybody would type
        extra = list(arm_extra) + (["-capture_increment", str(inc)] if inc != 5 else [])
        maybe(0.3, "-max_mip_overlap", str(int(rng.choice([0, 1, 200, 1000]))))
        maybe(0.3, "-starting_mip_overlap", str(int(rng.choice([0, 60, 150]))))
        maybe(0.3, "-masked_arm_threshold", str(rng.choice(["0", "-1", "0.01", "5"])))
        maybe(0.3, "-target_arm_copy", str(int(rng.choice([0, 1, 1000000]))))
        maybe(0.3, "-max_arm_copy_product", str(int(rng.choice([0, 1, 100000000]))))
        maybe(0.3, "-logistic_priority_score", str(rng.choice(["0", "0.999", "1.5", "-3"])))
        maybe(0.3, "-logistic_optimal_score", str(rng.choice(["0", "0.5", "1", "2"])))
        maybe(0.3, "-svr_priority_score", str(rng.choice(["-10", "0", "1.49", "9"])))
        maybe(0.3, "-svr_optimal_score", str(rng.choice(["-10", "0", "1.5", "9"])))
        maybe(0.2, "-capture_increment", str(int(rng.choice([50, 1000]))))
        maybe(0.2, "-seal_both_strands", "on")
        maybe(0.2, "-half_seal_both_strands", "on")
        maybe(0.15, "-logistic_heuristic", "off")
        maybe(0.15, "-check_copy_number", "off")
    silent = rng.random() < 0.15 or all_silent
    if silent:
        extra += ["-silent_mode", "on"]
    d = dict(name=f"probe{k:03d}", method=method, minC=lo, maxC=hi, sums=sums, arm_lengths=arm_lengths,
             flank=int(rng.choice([0, 200, 1000])) if extreme else int(rng.choice([0, 0, 3, 25])),
             tags=str(rng.choice(["30,30", "0,0", "1,60"])) if extreme else str(rng.choice(["5,0", "4,4", "0,8", "0,0"])), snps=bool(rng.random() < 0.5), trf=bool(rng.random() < 0.3),
             bwa=str(rng.choice(["hashed", "hashed", "unique", "blocks"])), model="svr_syn_64.model" if method != "logistic" else None, extra=extra)
    if multi:
        d["bed_text"] = "\n".join(bed_lines) + "\n"
    else:
        d["ivs"] = ivs
    if hard:
        d["chrom"] = "4"
    if d["snps"] and rng.random() < 0.5:
        # VCF records beyond biallelic SNVs: insertions (ALT longer than REF), several ALT alleles, a position listed twice, a `chr` prefix on the
        # chromosome column (parse_vcf keys its table by that column as it stands, mipgen.cpp:945-947)
        wild_seed = int(rng.integers(0, 1 << 30))

        def hook(snps, wild_seed=wild_seed):
            r2 = np.random.default_rng(wild_seed)
            out = []
            for s in snps:
                u = r2.random()
                if u < 0.12:
                    s = synth.Snp(s.chrom, s.pos, s.ref, s.alt + "ACGT"[int(r2.integers(0, 4))])
                elif u < 0.24:
                    s = synth.Snp(s.chrom, s.pos, s.ref, s.alt + "," + "ACGT"[int(r2.integers(0, 4))])
                elif u < 0.30:
                    out.append(synth.Snp(s.chrom, s.pos, s.ref, "ACGT"[int(r2.integers(0, 4))]))
                elif u < 0.36:
                    s = synth.Snp("chr" + s.chrom, s.pos, s.ref, s.alt)
                out.append(s)
            return out
        d["snp_hook"] = hook
    return d


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if seed == 1:
        shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT, exist_ok=True)
    genome2 = synth.random_genome(80000, 202, n_run_frac=0.002, n_run_len=8)                     # = tests/golden/genome2_chr1.fa.gz
    multi = {c: synth.random_genome(nb, sd, n_run_frac=0.002, n_run_len=7) for c, nb, sd in mg.MULTI_CHROMS}   # = genome3_chr*.fa.gz
    hard = len(sys.argv) > 3 and "hard" in sys.argv[3]                                           # (combinable: hardlong, hardextreme)
    genome4 = synth.hard_genome()                                                                # = tests/golden/genome4_chr4.fa.gz
    rng = np.random.default_rng(seed)
    t0 = time.time()
    made = 0
    for k in range(n):
        is_multi = bool(rng.random() < 0.4) and not hard
        d = random_design(rng, seed * 1000 + k, is_multi, long_regions=len(sys.argv) > 3 and "long" in sys.argv[3],
                          extreme=len(sys.argv) > 3 and "extreme" in sys.argv[3], hard=hard, all_silent=len(sys.argv) > 3 and "silent" in sys.argv[3])
        if not (d.get("ivs") or d.get("bed_text", "").strip()):
            continue
        try:
            if hard:
                mg.gen_design(genome4, d, "genome4_chr4.fa.gz", out_root=OUT)
            else:
                mg.gen_design(multi if is_multi else genome2, d, "genome3" if is_multi else "genome2_chr1.fa.gz", out_root=OUT)
            made += 1
            shutil.rmtree("/tmp/mipgen_golden_" + d["name"], ignore_errors=True)     # (the reference's FASTQ / SAM / all_mips files: gigabytes per long design)
        except (AssertionError, FileNotFoundError) as ex:  # the reference itself refuses the parameter set or ends without its files (a std::exception: exit
            # status 0, mipgen.cpp:2033-2036) - error behaviour is tools/error_probe.py's subject, not probed here
            print("reference failed on", d["name"], str(ex)[-300:].replace("\n", " | "))
            shutil.rmtree(os.path.join(OUT, "design_" + d["name"]), ignore_errors=True)
    print(f"{made} designs under {OUT} in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
