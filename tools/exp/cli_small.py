#!/usr/bin/env python3
"""Wall time of the drop-in command line on a golden design (small designs: where process start, HIP initialisation and code-object loading are the run).
    python3 tools/exp/cli_small.py [design = practice62_config1] [repeats = 5] [extra mipgen flags ...]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "practice62_config1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
meta = H.load_design(name)
for k in range(reps):
    with tempfile.TemporaryDirectory() as work:
        argv = H.prepare_cli_workdir(meta, work) + ["-gpu_timing", "on"] + sys.argv[3:]
        t0 = time.time()
        p = subprocess.run(argv, cwd=work, env=dict(os.environ, FAKEBWA_MODE=meta["bwa"]), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.time() - t0
        print(f"{name} run {k}: rc {p.returncode} wall {dt:.3f} s")
        if k == reps - 1:
            print("".join(l + "\n" for l in p.stderr.decode().split("\n") if "timing" in l)[:4000])
