"""CPU: the accelerated tile_regions driver of libmipgen_host.so (mipgen_amd/host/design.cpp: one worker thread + one accelerator handle per device,
result windows through channels to the sequential selection thread, abort paths) under ThreadSanitizer and AddressSanitizer + UBSan on a machine
without a GPU.  The product sources are compiled with the sanitizer where they lie (tests/stub_accel/Makefile) against a STUB of libmipgen_accel.so that
hands out what the oracle computes (tests/stub_accel/stub_accel.cpp: test infrastructure) - so the whole command line runs, with 1, 2 and 4 device
workers and forced result windows, its files are compared byte for byte with the reference's goldens, and an injected accelerator failure in every
call position of every worker ends the run the way the reference ends (/root/reference/mipgen.cpp:2029-2035: message, exit status 1, no completion
line) - without a hang, a race or a leak of a worker."""
import os
import subprocess
import sys

import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_accel")
SAN_ENV = {"thread": {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66 second_deadlock_stack=1"},
           "address": {"ASAN_OPTIONS": "detect_leaks=1 exitcode=67 abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1 print_stacktrace=1"}}


@pytest.fixture(scope="module", params=["thread", "address"])
def san(request):
    import fcntl
    os.makedirs(os.path.join(STUB, "_build"), exist_ok=True)
    with open(os.path.join(STUB, "_build", ".lock"), "w") as lock:          # (pytest-xdist workers share the build directory)
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-s", "-j4", "-C", STUB, f"SAN={request.param}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return request.param


def _run(san, meta, work, workers, extra=(), env_extra=None, timeout=600):
    os.makedirs(work, exist_ok=True)
    argv = H.prepare_cli_workdir(meta, work)
    exe = os.path.join(work, "mipgen")
    os.remove(exe)                                                    # (prepare_cli_workdir links the product binary: this test runs the sanitizer build)
    os.symlink(os.path.join(STUB, "_build", san, "mipgen"), exe)
    env = dict(os.environ, FAKEBWA_MODE=meta["bwa"], STUB_ACCEL_DEVICES=str(max(workers, 1)))
    env.update(SAN_ENV[san])
    env.update(env_extra or {})
    return subprocess.run(argv + ["-gpus", str(workers)] + list(extra), cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def _compare(meta, work):
    if "-silent_mode" in meta.get("extra", []):
        H.compare_outputs(meta, work, keys=("picked_mips", "snp_mips"), check_all=False)
    else:
        H.compare_outputs(meta, work)


@pytest.mark.parametrize("name,workers", [("logistic_default_arms", 1), ("logistic_default_arms", 2), ("mixed_12_regions", 4), ("mixed_12_regions", 2),
                                          ("merge_flank_tags", 2), ("svr_small", 1), ("multichr_mixed", 4), ("multichr_logistic_snps", 2),
                                          ("long_capture_logistic", 1), ("long_capture_svr", 2), ("empty_sum_lists", 2), ("no_arm_pairs", 1),
                                          ("both_arm_options", 2), ("wild_vcf_mixed", 2), ("edge_options", 1), ("edge_options_svr", 1),
                                          ("hard_mixed", 2), ("hard_logistic", 4), ("hard_svr", 2), ("hard_lowcomplexity_svr_silent", 4), ("hard_saturated_logistic", 2), ("hard_saturated_mixed", 2)])
@pytest.mark.parametrize("gather", ["pcie", "rccl"])
def test_threaded_driver_under_sanitizers_writes_the_reference_files(san, name, workers, gather, tmp_path):
    """gather = rccl: the same designs through `-gpu_gather rccl` (mipgen_amd/host/gather.cpp) with TWO and FOUR communicator ranks - its RCCL / HIP
    entry points are the memcpy-backed, stream-threaded stand-ins of tests/stub_accel/stub_rccl.cpp, which the file binds at run time exactly as it
    binds librccl on a GPU box: grouped send / receive per window into two receive slots, one D2H copy, selection of the window before meanwhile."""
    if gather == "rccl":
        workers = 4 if workers in (1, 4) else 2
    if san == "address" and (name, workers) not in (("mixed_12_regions", 4), ("merge_flank_tags", 2), ("svr_small", 1), ("multichr_logistic_snps", 2),
                                                    ("long_capture_logistic", 1), ("long_capture_svr", 2), ("no_arm_pairs", 1), ("edge_options", 1), ("wild_vcf_mixed", 2),
                                                    ("hard_mixed", 2), ("hard_logistic", 4), ("hard_svr", 2), ("hard_lowcomplexity_svr_silent", 4),
                                                    ("hard_saturated_logistic", 2), ("hard_saturated_mixed", 2)):
        pytest.skip("sixteen under ThreadSanitizer, fourteen under AddressSanitizer + UBSan")
    if san == "thread" and name in ("hard_logistic", "hard_svr", "hard_lowcomplexity_svr_silent", "hard_saturated_logistic"):
        pytest.skip("the hard genome's large designs (a million records): AddressSanitizer; hard_mixed under both")
    if gather == "rccl" and not os.environ.get("MIPGEN_SAN_FULL") and name in ("hard_logistic", "hard_svr", "hard_lowcomplexity_svr_silent", "hard_saturated_logistic",
                                                                                 "long_capture_svr", "merge_flank_tags", "multichr_logistic_snps"):
        pytest.skip("once is enough for the large designs (the CPU suite is run sequentially by the driver: minutes, not tens of them)")
    if san == "thread" and name == "edge_options":
        pytest.skip("190,000 records under ThreadSanitizer take a minute: AddressSanitizer only")
    if san == "thread" and name in ("long_capture_logistic", "long_capture_svr") and not os.environ.get("MIPGEN_SAN_FULL"):
        pytest.skip("the 1,100-base captures (string lengths of the host side): AddressSanitizer (45 s each under ThreadSanitizer; MIPGEN_SAN_FULL=1 runs them)")
    meta = H.load_design(name)
    biggest = max(r[2] - r[1] for r in meta["intervals"])
    p = _run(san, meta, str(tmp_path), workers, extra=["-gpu_window_candidates", str(max(1000, biggest * 2000)), "-gpu_gather", gather])
    err = p.stderr.decode()
    assert p.returncode == 0, err[-3000:]
    assert "ThreadSanitizer" not in err and "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    _compare(meta, str(tmp_path))


CALLS = ["create", "load_model_file", "long_range_content_batch", "upload_regions", "score_window", "replay_condense", "collapse", "download_collapsed",
         "download_replay", "format_all_mips", "download_text", "rescore_survivors"]
POSITIONS = [(0, 1), (3, 1), (1, 2)]                                   # (worker, n-th call): first / last worker's first call, a middle worker's second window
# every call once, the positions in rotation (MIPGEN_SAN_FULL=1: every call in every position, ~4 minutes)
SWEEP = [(c, d, n) for c in CALLS for d, n in POSITIONS] if os.environ.get("MIPGEN_SAN_FULL") else [(c,) + POSITIONS[i % 3] for i, c in enumerate(CALLS)]


@pytest.mark.parametrize("call,device,nth", SWEEP)
def test_injected_accelerator_failure_ends_the_run_cleanly(san, call, device, nth, tmp_path):
    """An accelerator call of a worker (first / last worker, first / second result window) fails once: the run ends with the reference's error
    convention - exit status 1, `unable to tile sequences due to circumstance N`, no `mip picking complete` -, no worker is left behind (the process
    exits: no hang), and neither sanitizer reports anything on the abort paths."""
    if san == "address" and not os.environ.get("MIPGEN_SAN_FULL") and CALLS.index(call) % 3 != 1:
        pytest.skip("the abort paths are swept under ThreadSanitizer; AddressSanitizer takes every third")
    meta = H.load_design("mixed_12_regions")                          # mixed + SNPs, 12 regions, non-silent: every call of the worker is on its path
    p = _run(san, meta, str(tmp_path), 4, extra=["-gpu_window_candidates", "30000"], env_extra={"STUB_ACCEL_FAIL": f"{device}:{call}:{nth}"}, timeout=300)
    err = p.stderr.decode()
    assert "ThreadSanitizer" not in err and "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    if f"injected failure of {call}" not in err:
        # a call a handle makes once (create, model, long-range content, upload) has no second occurrence: the run completes
        assert nth > 1 and call in ("create", "load_model_file", "long_range_content_batch", "upload_regions") and p.returncode == 0, err[-2000:]
        return
    assert p.returncode == 1, (p.returncode, err[-2000:])
    assert "unable to tile sequences due to circumstance" in err and "mip picking complete" not in err
    assert "mip picking complete" not in open(os.path.join(str(tmp_path), "out.progress.txt")).read()


@pytest.mark.parametrize("device,nth", [(0, 1), (1, 2)])
def test_injected_failure_of_the_silent_route_ends_the_run_cleanly(san, device, nth, tmp_path):
    """A silent design scores, replays and condenses a window in one call (mipgen_accel_score_condense_window, ABI 6): the same convention when that call
    fails on the first worker's first window or on a middle worker's second."""
    if san == "address" and not os.environ.get("MIPGEN_SAN_FULL"):
        pytest.skip("swept under ThreadSanitizer")
    meta = H.load_design("mixed_12_regions")                            # (the design of the sweep above, run silent)
    p = _run(san, meta, str(tmp_path), 4, extra=["-gpu_window_candidates", "30000", "-silent_mode", "on"],
             env_extra={"STUB_ACCEL_FAIL": f"{device}:score_condense_window:{nth}"}, timeout=300)
    err = p.stderr.decode()
    assert "ThreadSanitizer" not in err and "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert "injected failure of score_condense_window" in err and p.returncode == 1, (p.returncode, err[-2000:])
    assert "unable to tile sequences due to circumstance" in err and "mip picking complete" not in err


RCCL_SWEEP = [("accel", "window_views", 1, 2), ("accel", "synchronize", 3, 1), ("accel", "format_all_mips", 2, 2), ("rccl", "ncclCommInitAll", 0, 1),
              ("rccl", "ncclGroupEnd", 0, 1), ("rccl", "ncclGroupEnd", 0, 5), ("rccl", "hipMemcpyAsync", 0, 3), ("rccl", "hipEventSynchronize", 0, 2),
              ("rccl", "hipMalloc", 0, 2)]


@pytest.mark.parametrize("where,call,device,nth", RCCL_SWEEP)
def test_injected_failure_on_the_rccl_route_ends_the_run_cleanly(san, where, call, device, nth, tmp_path):
    """The abort paths of `-gpu_gather rccl` with four ranks: an accelerator call of a worker on that route (the device views of a window, the
    synchronisation before they are published, the text of a later window) or a call of the gather itself (communicator set-up, the n-th grouped
    send / receive, the D2H copy, the wait for a slot, a receive buffer) fails once.  Same convention as on the PCIe route: exit status 1, the
    reference's message, no completion line; no worker is left waiting for a transfer that will never be marked, nothing is freed under a copy."""
    if san == "address" and RCCL_SWEEP.index((where, call, device, nth)) % 3 != 0 and not os.environ.get("MIPGEN_SAN_FULL"):
        pytest.skip("swept under ThreadSanitizer; AddressSanitizer takes every third")
    meta = H.load_design("mixed_12_regions")
    env = {"STUB_ACCEL_FAIL": f"{device}:{call}:{nth}"} if where == "accel" else {"STUB_RCCL_FAIL": f"{call}:{nth}"}
    p = _run(san, meta, str(tmp_path), 4, extra=["-gpu_window_candidates", "30000", "-gpu_gather", "rccl"], env_extra=env, timeout=300)
    err = p.stderr.decode()
    assert "ThreadSanitizer" not in err and "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert p.returncode == 1, (p.returncode, err[-2000:])
    assert ("injected" in err or "rccl gather" in err) and "unable to tile sequences due to circumstance" in err and "mip picking complete" not in err, err[-2000:]
    assert "mip picking complete" not in open(os.path.join(str(tmp_path), "out.progress.txt")).read()


def _waiting_seconds(err):
    import re
    m = re.search(r"waiting for the device workers ([0-9.e+-]+) s, selection stage ([0-9.e+-]+) s", err)
    assert m, err[-2000:]
    return float(m.group(1)), float(m.group(2))


_ONE_WORKER_WAIT = {}


@pytest.mark.parametrize("gather", ["pcie", "rccl"])
def test_device_workers_overlap_with_many_windows_per_worker(gather, tmp_path):
    """`mipgen -gpus 4` with MANY result windows per device must overlap its device workers: the regions are dealt to the devices in blocks (block b
    to device b mod N) and consumed in design order, so no device waits for the devices before it to be consumed completely (round 5 handed out
    contiguous shards: a device slept after three windows; four workers x 15 windows measured 3.17 s against 2.98 s for one).  The time the selection
    thread spends waiting for the device workers with four of them (>= 12 windows each) is at most 0.4 x the one-worker time.  The oracle-backed stub
    is the 'device' (its scoring time per window is CPU time of the worker thread), so this needs four cores."""
    if (os.cpu_count() or 1) < 4:
        pytest.skip("needs four cores: the stub accelerator scores on the worker threads")
    san = "address"
    import fcntl
    os.makedirs(os.path.join(STUB, "_build"), exist_ok=True)
    with open(os.path.join(STUB, "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-s", "-j4", "-C", STUB, f"SAN={san}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    meta = H.load_design("practice62_config1")                         # 62 regions, logistic, silent
    extra = ["-gpu_window_candidates", "1", "-gpu_timing", "on", "-gpu_gather", gather]     # one region per window: 62 windows, 15-16 per worker of four
    waits = {}
    for workers in (1, 4):
        if workers == 1 and "one" in _ONE_WORKER_WAIT:                  # (the one-worker run is the same for both gather routes: there is no gather to speak of)
            waits[1] = _ONE_WORKER_WAIT["one"]
            continue
        work = str(tmp_path / f"w{workers}")
        p = _run(san, meta, work, workers, extra=extra, timeout=900)
        err = p.stderr.decode()
        assert p.returncode == 0, err[-3000:]
        assert "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
        _compare(meta, work)
        waits[workers] = _waiting_seconds(err)
        if workers == 1:
            _ONE_WORKER_WAIT["one"] = waits[1]
        m = __import__("re").search(r"in (\d+) region blocks on (\d+) device", err)
        assert m and int(m.group(2)) == workers and (workers == 1 or int(m.group(1)) >= 12 * workers), err[-2000:]
    assert waits[4][0] <= 0.4 * waits[1][0], f"device workers do not overlap: waiting {waits[4][0]:.2f} s with four workers, {waits[1][0]:.2f} s with one"


def test_host_profile_harness_runs(tmp_path):
    """tools/exp/host_profile.sh: the uninstrumented stub build with STUB_ACCEL_FAKE=1 (fabricated survivors, nothing scored) times the HOST side of an
    exome-scale silent design on a machine without a GPU - DESIGN.md section 7's selection-stage numbers come from it.  Here: 400 exons, two stub
    devices; the run ends well, picks MIPs and prints the stage-by-stage timing lines the notebook quotes.  (Timing harness only: no file of it is compared with anything.)"""
    import fcntl
    os.makedirs(os.path.join(STUB, "_build"), exist_ok=True)
    with open(os.path.join(STUB, "_build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-s", "-j4", "-C", STUB, "SAN=none"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    env = dict(os.environ, STUB_ACCEL_FAKE="1", STUB_ACCEL_DEVICES="2", MIPGEN_CLI_BIN=os.path.join(STUB, "_build", "none", "mipgen"))
    p = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "cli_exome.py"), "400", str(tmp_path / "w"), "exome", "logistic", "-gpus", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert "pick stage: position sets" in out and "on 2 device worker(s)" in out, out[-3000:]
    picked = int(__import__("re").search(r"(\d+) picked MIPs for 400 intervals", out).group(1))
    assert picked >= 400
