#!/usr/bin/env python3
"""Run the drop-in command line on ONE probe design (tests/golden_probe/design_<name>) and keep its all_mips file under gpurun_out/ for a diff against the
reference's (tools/diff_probe.py leaves that in /tmp/mipgen_golden_<name>/ of the build container).   python3 tools/exp/probe_one.py probe1050 [flags]"""
import gzip
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402
from tests.test_gpu_cli import run_cli  # noqa: E402

name = sys.argv[1]
meta = H.load_design(name, root=os.path.join(ROOT, "tests", "golden_probe"))
work = "/tmp/probe_one_" + name
shutil.rmtree(work, ignore_errors=True)
p = run_cli(meta, work, extra=sys.argv[2:])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(work, "out.all_mips.txt"), "rb") as fh, gzip.open(os.path.join(ROOT, "gpurun_out", name + ".all_mips.txt.gz"), "wb") as gz:
    gz.write(fh.read())
print("ok")
