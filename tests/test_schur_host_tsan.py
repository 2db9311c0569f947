"""The helper team of the host window kernel under ThreadSanitizer (CPU suite).

`csrc/schur_host.hip` is host-only code: compiled as plain C++ with -fsanitize=thread together with
`tests/tsan/aed_team_tsan.cpp`, it reduces the captured AED windows of
`tests/golden/aed_windows_lcg20000.npz` serially and with the team.  The sanitizer must stay silent
(every hand-over between the calling thread and the helpers goes through the published log position
or a helper's tail) and the two results must be identical to the last bit."""
import os, shutil, struct, subprocess
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with a ThreadSanitizer runtime in this image")
    out = tmp_path_factory.mktemp("tsan")
    exe = str(out / "aed_team_tsan")
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-DSN_NO_TARGET_CLONES",
           "-march=x86-64-v3", "-pthread", os.path.join(HERE, "tsan", "aed_team_tsan.cpp"),
           os.path.join(ROOT, "starneig_amd", "csrc", "schur_host.hip"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("ThreadSanitizer build not possible here: " + r.stderr[-300:])
    g = np.load(os.path.join(HERE, "golden", "aed_windows_lcg20000.npz"))
    raw = str(out / "windows.bin")
    with open(raw, "wb") as f:
        for k in range(len(g["subs"])):
            W = np.asfortranarray(g["windows"][k])
            f.write(struct.pack("<idd", W.shape[0], float(g["subs"][k]), float(g["thres"][k])))
            f.write(W.tobytes(order="F"))
    return exe, raw


@pytest.mark.parametrize("helpers", [3, 5])
def test_team_is_race_free_and_bit_identical_to_the_serial_kernel(harness, helpers):
    exe, raw = harness
    cmd = [exe, raw, str(helpers)]
    if shutil.which("setarch"):                 # the sanitizer's shadow mapping and a randomised layout do not always agree
        cmd = ["setarch", "x86_64", "-R"] + cmd
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if r.returncode < 0 and "ThreadSanitizer" not in r.stderr and "window" not in r.stdout:
        pytest.skip(f"the sanitizer runtime does not start on this kernel (signal {-r.returncode} before main)")
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-2000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert "2 windows, 0 mismatches" in r.stdout
