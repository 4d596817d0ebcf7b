"""GPU: the differential probe (tools/diff_probe.py).  Where tests/golden_probe/ holds designs the real reference was run on in the build container
(random parameter sets, BEDs and options; git-ignored scratch), the drop-in command line must write the same files - with one or two device workers,
forced result windows and either gather route.  Runs with MIPGEN_PROBE=1 where the directory exists (a fresh clone has none: the committed goldens are
tests/golden/)."""
import os
import shutil
import zlib

import pytest

from tests import helpers as H
from tests.test_gpu_cli import run_cli

pytestmark = pytest.mark.gpu
PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_probe")
# opt-in (MIPGEN_PROBE=1): hundreds of designs - a campaign, not part of the default suite
NAMES = sorted(d[len("design_"):] for d in os.listdir(PROBE) if d.startswith("design_")) if os.path.isdir(PROBE) and os.environ.get("MIPGEN_PROBE") == "1" else []


@pytest.mark.skipif(not NAMES, reason="no probe designs / MIPGEN_PROBE != 1 (python3 tools/diff_probe.py generates them where the reference is built)")
@pytest.mark.parametrize("name", NAMES or ["none"])
def test_probe_design_matches_the_reference(name, tmp_path):
    meta = H.load_design(name, root=PROBE)
    h = zlib.crc32(name.encode())
    extra = []
    if h % 3 == 1:
        extra += ["-gpus", str(2 + h % 3), "-gpu_window_candidates", str(20000 + h % 50000)]     # 2-4 device workers: the regions dealt in blocks
    elif h % 3 == 2:
        extra += ["-gpu_window_candidates", str(5000 + h % 100000), "-gpu_gather", "rccl"]
    run_cli(meta, str(tmp_path), extra=extra)
    if "-silent_mode" in meta.get("extra", []):
        H.compare_outputs(meta, str(tmp_path), keys=("picked_mips", "snp_mips"), check_all=False)
    else:
        H.compare_outputs(meta, str(tmp_path))
    shutil.rmtree(str(tmp_path), ignore_errors=True)          # (a campaign of thousands of designs: FASTQ / SAM / all_mips files of a passed design are not kept)
