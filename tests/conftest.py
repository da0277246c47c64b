import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The rank processes write their results (whole flat parameter buffers: some 200 MB) outside the tree; their logs
# and exit codes are copied to gpurun_out/dp_children when the results are collected.
DP_OUTDIR = os.path.join(tempfile.gettempdir(), f"ried_dp_children_{os.getpid()}")
DP_LOGDIR = os.path.join(ROOT, "gpurun_out", "dp_children")
_dp_launcher = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_session(config) -> bool:
    """The session may run GPU tests (not `-m "not gpu"`) on a machine that has a GPU.  device_count() does not
    initialise the GPU runtime."""
    if "not gpu" in (config.option.markexpr or ""):
        return False
    import torch
    return torch.cuda.device_count() > 0


def pytest_sessionstart(session):
    """GPU sessions: (1) native backtraces on fatal signals, written to the real stderr (pytest redirects descriptor
    2 per test) and to gpurun_out/crash_backtrace.log; (2) the rank processes of the multi-rank tests are started
    NOW, before this process makes its first GPU call (a process that has initialised the GPU starts no children),
    and run beside the first test files; tests/test_g8_multirank_gpu.py collects their results."""
    global _dp_launcher
    config = session.config
    if not _gpu_session(config):
        return
    os.makedirs(DP_OUTDIR, exist_ok=True)
    for f in os.listdir(DP_OUTDIR):
        os.remove(os.path.join(DP_OUTDIR, f))
    _dp_launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_child.py"), "launcher", DP_OUTDIR],
                                    cwd=ROOT, stdout=open(os.path.join(DP_OUTDIR, "launcher.log"), "w"),
                                    stderr=subprocess.STDOUT)
    fd = -1
    try:            # the descriptor pytest saved the real stderr in, if fd-level capture is on
        capman = config.pluginmanager.getplugin("capturemanager")
        fd = int(capman._global_capturing.err.targetfd_save)
    except Exception:
        fd = -1
    # (diagnostics only: nothing here may stop the session - e.g. a tree without a writable gpurun_out/)
    try:
        logdir = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(logdir, exist_ok=True)
        except OSError:
            logdir = tempfile.gettempdir()
        from reid_gan_amd import ops
        ops.install_crash_backtrace(os.path.join(logdir, "crash_backtrace.log"), fd)
    except Exception as e:          # noqa: BLE001
        print(f"crash backtraces not installed: {e}", file=sys.stderr)


def pytest_sessionfinish(session, exitstatus):
    if _dp_launcher is not None and _dp_launcher.poll() is None:
        try:
            _dp_launcher.wait(timeout=5)
        except subprocess.TimeoutExpired:
            _dp_launcher.kill()           # this exact child (its rank processes end at their own limits)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dp_results():
    """Directory with the rank processes' results, once the launcher has finished (None if it was never started)."""
    if _dp_launcher is None:
        return None
    try:
        _dp_launcher.wait(timeout=900)
    except subprocess.TimeoutExpired:
        _dp_launcher.kill()
        pytest.fail("the multi-rank launcher did not finish within 900 s: " + DP_OUTDIR)
    try:                             # (copies for the record; the tests read DP_OUTDIR)
        os.makedirs(DP_LOGDIR, exist_ok=True)
        for f in os.listdir(DP_OUTDIR):
            if not f.endswith(".npz"):
                shutil.copy(os.path.join(DP_OUTDIR, f), os.path.join(DP_LOGDIR, f))
    except OSError:
        pass
    return DP_OUTDIR


@pytest.fixture(autouse=True)
def _settle_gpu_between_tests(request):
    """After every GPU test: drain the device, let Python drop the test's plans now (with the device idle, not at
    some allocation inside the next test), drain again.  Captured graphs are parked, not destroyed, by default
    (reid_gan_amd.train_step._RETAINED_GRAPHS)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
