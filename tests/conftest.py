import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-style-transfer-gan_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# A GPU-clean helper process for tests that must start other programs (the two-rank data-parallel rehearsal): once this session
# has initialised the GPU it may not exec anything, whatever the order the tests run in (tools/clean_launcher.py).
_LAUNCHER = None


def pytest_sessionstart(session):
    global _LAUNCHER
    try:
        import torch
        have_gpu = torch.cuda.device_count() >= 1 and not torch.cuda.is_initialized()  # device_count() does not initialise it
    except Exception:
        have_gpu = False
    if have_gpu:
        import subprocess
        _LAUNCHER = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "clean_launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, bufsize=1)


def pytest_sessionfinish(session, exitstatus):
    global _LAUNCHER
    if _LAUNCHER is not None:
        try:
            _LAUNCHER.stdin.close()
            _LAUNCHER.wait(timeout=10)
        except Exception:
            _LAUNCHER.kill()
        _LAUNCHER = None


def launch_clean(cmd, env=None, cwd=None, timeout=600):
    """Run ``cmd`` from the GPU-clean helper; returns a dict(returncode, stdout, stderr).  Raises if the helper is missing."""
    import json
    if _LAUNCHER is None or _LAUNCHER.poll() is not None:
        raise RuntimeError("no GPU-clean launcher process (tests/conftest.py starts it at session start when a GPU is present)")
    _LAUNCHER.stdin.write(json.dumps({"cmd": cmd, "env": env, "cwd": cwd, "timeout": timeout}) + "\n")
    _LAUNCHER.stdin.flush()
    line = _LAUNCHER.stdout.readline()
    if not line:
        raise RuntimeError("the GPU-clean launcher process died")
    return json.loads(line)


def usable_cpus() -> int:
    """Cores this process may really use (the GPU box shows 256 logical CPUs and grants 16 through its cgroup): torch sizes its
    thread pool by the former, and an oracle run then crawls under 16x oversubscription."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError, IndexError):
        pass
    return n


@pytest.fixture(scope="session", autouse=True)
def _torch_threads():
    import torch
    torch.set_num_threads(max(1, min(usable_cpus(), 16)))
    yield


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


def rel_l2(a, b):
    import torch
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _fresh_library_switches(request, monkeypatch):
    """libmstg_hip.so reads the MSTG_* switches once per load; a test that sets one through ``setenv_refresh`` (or plain
    monkeypatch + ops.refresh_env()) must not leak it into the next test: re-read after every gpu test."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        monkeypatch.undo()
        from mstg_hip import _lib
        if _lib._lib is not None:
            _lib._lib.mstg_env_refresh()


@pytest.fixture
def setenv_refresh(monkeypatch):
    """setenv/delenv that the HIP library sees at once."""
    from mstg_hip import ops

    class _S:
        def set(self, k, v):
            monkeypatch.setenv(k, v)
            ops.refresh_env()

        def unset(self, k):
            monkeypatch.delenv(k, raising=False)
            ops.refresh_env()
    return _S()
