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


def _error_cases():
    import json
    import sys
    sys.path.insert(0, os.path.join(H.ROOT, "tools"))
    import error_probe as ep
    with open(ep.STORE) as fh:
        return ep, json.load(fh)


@pytest.mark.parametrize("name", ["unknown_option", "bad_method", "min_gt_max", "no_bed_file", "bad_cols", "bad_num", "unknown_chromosome", "empty_bed",
                                  "nonint_capture", "bad_bwa", "odd_args", "bad_arm_lengths", "hex_float_score", "overflow_score", "hex_float_threshold",
                                  "threshold_trailing_space", "hex_int"])
def test_error_behaviour_matches_the_reference(name, tmp_path):
    """Malformed command lines and inputs (tools/error_probe.py; expectations in tests/golden/error_cases.json = what the REAL reference did): the exit
    status, the last lines of stderr and the files written are the reference's - `throw <int>` paths exit with 1 and a circumstance number
    (mipgen.cpp:2029-2032); a std::exception (boost::lexical_cast on a bad integer, vector::at on a BED line of two fields, std::string(NULL) for an
    option without its value) prints "unable to tile sequences" + the exception text and exits with 0 (:2033-2036, main() falls off its end); a BED
    without intervals and -min_capture_size above -max_capture_size complete with header-only files.  Real-valued options follow the reference's
    boost::lexical_cast<double> (no hex floats, overflow refused); a malformed -masked_arm_threshold is only met at the first design_mip (:626): the
    four output files exist by then and stay empty.  None of these needs the device."""
    ep, want = _error_cases()
    work = str(tmp_path / "w")
    base = ep.lay_out(work)
    got = ep.run_one(H.CLI_BIN, ep.cases(base)[name], work)
    assert got == want[name], (got, want[name])
