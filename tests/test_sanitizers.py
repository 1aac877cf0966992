"""The CPU-side C of the repo under sanitizers (the reference is not clean — :115 heap overflow for an
out-of-domain particle, :145 stack overflow beyond 48 neighbours; the restatement and the host helpers must be).

* ASan + UBSan: the oracle (`make -C oracle asan`) under tests/test_oracle_golden.py, the host helpers
  (`make -C pi-sph-fluid_amd host-asan`) under tests/test_host.py — each in a child pytest with the sanitizer runtime
  preloaded into python (the libraries are loaded through ctypes).
* TSan: the shared-memory transport of the multi-GPU host (host/sph_shm.h: barrier, double-buffered collectives and halo
  mailboxes) driven by threads (`make -C pi-sph-fluid_amd host-tsan`).
No GPU involved."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip("gcc has no " + name)
    return path


def _child_pytest(files, extra_env):
    env = dict(os.environ, **extra_env)
    # leaks of the interpreter itself are not ours; a UBSan report aborts (-fno-sanitize-recover), an ASan report exits 1
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=66"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    return r.stdout


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    out = _child_pytest(["tests/test_oracle_golden.py"], {"LD_PRELOAD": _runtime("libasan.so"), "ORC_SANITIZE": "asan"})
    assert " passed" in out, out[-500:]


def test_host_helpers_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pi-sph-fluid_amd"), "host-asan"])
    lib = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "libsph_host_asan.so")
    out = _child_pytest(["tests/test_host.py"], {"LD_PRELOAD": _runtime("libasan.so"), "SPH_HOST_LIB": lib})
    assert " passed" in out, out[-500:]


def test_shared_memory_transport_under_tsan():
    _runtime("libtsan.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pi-sph-fluid_amd"), "host-tsan"])
    exe = os.path.join(ROOT, "pi-sph-fluid_amd", "host", "test_shm_comm_tsan")
    for ranks in (2, 4, 8):
        r = subprocess.run([exe, str(ranks), "300"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert "ok" in r.stdout
    # a rank whose callback fails in front of the barrier: every rank returns an error, nobody is left waiting (the run ENDS)
    for ranks, bad, step in ((2, 0, 0), (4, 2, 17), (8, 7, 5)):
        r = subprocess.run([exe, str(ranks), "50", str(bad), str(step)], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "ThreadSanitizer" not in r.stderr and "reached every rank" in r.stdout, (r.stdout + r.stderr)[-3000:]
