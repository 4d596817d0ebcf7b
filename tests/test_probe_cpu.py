"""CPU: the differential probe (tools/diff_probe.py) through the host library on the oracle-backed stand-in of the accelerator (tests/stub_accel/, built
with AddressSanitizer + UBSan): every probe design must reproduce the reference's files - this pins the ORACLE (the checker of the HIP path) and the host
side on hundreds of random parameter sets without a GPU.  Opt-in like its GPU twin: MIPGEN_PROBE=1 where tests/golden_probe/ exists."""
import os
import shutil
import zlib

import pytest

from tests import helpers as H
from tests.test_host_threads_cpu import _compare, _run, san  # noqa: F401  (the fixture builds the sanitizer binaries)

PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_probe")
NAMES = sorted(d[len("design_"):] for d in os.listdir(PROBE) if d.startswith("design_")) if os.path.isdir(PROBE) and os.environ.get("MIPGEN_PROBE") == "1" else []


@pytest.mark.skipif(not NAMES, reason="no probe designs / MIPGEN_PROBE != 1")
@pytest.mark.parametrize("name", NAMES or ["none"])
def test_probe_design_on_the_oracle_backed_stub(san, name, tmp_path):  # noqa: F811
    if san != "address":
        pytest.skip("one sanitizer build is enough for the campaign")
    meta = H.load_design(name, root=PROBE)
    h = zlib.crc32(name.encode())
    workers = 1 + h % 3
    p = _run(san, meta, str(tmp_path), workers, extra=["-gpu_window_candidates", str(5000 + h % 100000)])
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    _compare(meta, str(tmp_path))
    shutil.rmtree(str(tmp_path), ignore_errors=True)          # (thousands of designs: the files of a passed one are not kept)
