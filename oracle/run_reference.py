"""Drive the REAL reference binary (oracle/_ref/mipgen_ref, built by oracle/Makefile from the sources
under /root/reference) end-to-end on a synthetic design.  Test infrastructure only: used by
tests/golden/make_golden.py (fixture generation, this container) and by bench.py's cpu_baseline leg.

The reference shells out to bwa / tabix / trf (/root/reference/mipgen.cpp:146,154,560-561,841-842,919,927,1049);
the stand-ins in this directory satisfy those calls deterministically.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import time
from typing import Dict, List, Optional, Sequence

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "mipgen_ref")
REF_BIN_O0 = os.path.join(HERE, "_ref", "mipgen_ref_O0")


def have_reference(o0: bool = False) -> bool:
    return os.path.exists(REF_BIN_O0 if o0 else REF_BIN)


def run_reference(workdir: str, genome_dir: str, bed: str, project: str, min_capture: int, max_capture: int,
                  score_method: str = "logistic", model_path: Optional[str] = None, bwa_mode: str = "unique",
                  snp_file: Optional[str] = None, use_trf: bool = False, silent: bool = False,
                  extra: Sequence[str] = (), o0: bool = False, timeout: Optional[float] = None,
                  hot_marker: bool = False) -> Dict[str, object]:
    """Run the reference in `workdir` (it writes <project>.* there).  Returns timing + paths.

    hot_marker: additionally report "hot_seconds" = wall time from the reference's own stderr line
    "[mipgen] bwa copy number analysis finished" (/root/reference/mipgen.cpp:349, printed right before tile_regions) to process exit,
    i.e. tile_regions alone (enumeration + scoring + selection + output) without the input stage / FASTQ / stand-in bwa I/O.

    The SVR model is looked up next to argv[0] as `mipgen_svr.model` (/root/reference/mipgen.cpp:137-138,409),
    so the binary is copied into workdir and the model placed beside it."""
    os.makedirs(workdir, exist_ok=True)
    src = REF_BIN_O0 if o0 else REF_BIN
    exe = os.path.join(workdir, "mipgen_ref")
    shutil.copy2(src, exe)
    if model_path is not None:
        shutil.copy2(model_path, os.path.join(workdir, "mipgen_svr.model"))
    cmd: List[str] = [exe,
                      "-regions_to_scan", os.path.abspath(bed),
                      "-project_name", project,
                      "-min_capture_size", str(min_capture),
                      "-max_capture_size", str(max_capture),
                      "-bwa_genome_index", os.path.join(os.path.abspath(genome_dir), "index.fa"),
                      "-genome_dir", os.path.abspath(genome_dir),
                      "-bwa", os.path.join(HERE, "fakebwa.sh"),
                      "-score_method", score_method]
    if snp_file is not None:
        cmd += ["-snp_file", os.path.abspath(snp_file), "-tabix", os.path.join(HERE, "faketabix.sh")]
    if use_trf:
        cmd += ["-trf", os.path.join(HERE, "faketrf.sh")]
    if silent:
        cmd += ["-silent_mode", "on"]
    cmd += list(extra)
    env = dict(os.environ)
    env["FAKEBWA_MODE"] = bwa_mode
    t0 = time.perf_counter()
    base = os.path.join(workdir, project)
    if hot_marker:
        t_mark = None
        err_lines: List[str] = []
        with subprocess.Popen(cmd, cwd=workdir, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) as pr:
            for raw in pr.stderr:                                # cerr is unbuffered: the marker arrives when it is printed
                line = raw.decode(errors="replace")
                err_lines.append(line)
                if t_mark is None and "bwa copy number analysis finished" in line:
                    t_mark = time.perf_counter()
                if timeout is not None and time.perf_counter() - t0 > timeout:
                    pr.kill()
            rc = pr.wait()
        t1 = time.perf_counter()
        return {"returncode": rc, "seconds": t1 - t0, "hot_seconds": (t1 - t_mark) if t_mark is not None else None,
                "stderr": "".join(err_lines[-50:]), "stdout": "",
                "all_mips": base + ".all_mips.txt", "collapsed_mips": base + ".collapsed_mips.txt",
                "picked_mips": base + ".picked_mips.txt", "snp_mips": base + ".snp_mips.txt",
                "progress": base + ".progress.txt", "cmd": cmd}
    proc = subprocess.run(cmd, cwd=workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    dt = time.perf_counter() - t0
    return {"returncode": proc.returncode, "seconds": dt, "stderr": proc.stderr.decode(errors="replace"),
            "stdout": proc.stdout.decode(errors="replace"),
            "all_mips": base + ".all_mips.txt", "collapsed_mips": base + ".collapsed_mips.txt",
            "picked_mips": base + ".picked_mips.txt", "snp_mips": base + ".snp_mips.txt",
            "progress": base + ".progress.txt", "cmd": cmd}
