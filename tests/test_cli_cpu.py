"""CPU: the drop-in command line without a HIP device - there is no CPU path for the hot path, so the run fails the way the reference
fails (`unable to tile sequences due to circumstance N`, exit status 1; /root/reference/mipgen.cpp:2029-2035) and, like the reference,
does NOT announce "mip picking complete" (mipgen.cpp:532-533 is only reached when tile_regions returns)."""
import os
import subprocess

import pytest

from mipgen_amd import capi
from tests import helpers as H


def test_failed_run_does_not_announce_completion(tmp_path):
    if capi.load_library().mipgen_accel_device_count() > 0:
        pytest.skip("a GPU is present: the run would succeed")
    meta = H.load_design("logistic_default_arms")
    work = str(tmp_path / "w")
    os.makedirs(work)
    argv = H.prepare_cli_workdir(meta, work)
    p = subprocess.run(argv, cwd=work, env=dict(os.environ, FAKEBWA_MODE=meta["bwa"]), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode()
    assert p.returncode == 1
    assert "unable to tile sequences due to circumstance 17" in err
    assert "mip picking complete" not in err
    assert "mip picking complete" not in open(os.path.join(work, "out.progress.txt")).read()


def test_doc_is_printed_in_full():
    """`mipgen -doc` prints the whole option documentation (the reference: mipgen.cpp:1291-1310), extensions and this build's limits included."""
    p = subprocess.run([H.CLI_BIN, "-doc"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    out = p.stderr.decode() + p.stdout.decode()
    assert p.returncode == 1
    for needle in ("-feature_flank", "-score_method", "-gpu_gather pcie|rccl", "limits of this build"):
        assert needle in out, needle
