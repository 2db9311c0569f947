"""CPU suite: a C program compiled against include/starneig/starneig.h and linked with
-lstarneig_amd reproduces the reference's argument-check table (no GPU: the node is never
initialised, every check precedes the STARNEIG_NOT_INITIALIZED test)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_compiles_links_and_checks_arguments(tmp_path):
    import starneig_amd as S
    S.lib.load()                                    # the library must have been built
    libdir = os.path.join(ROOT, "starneig_amd")
    exe = str(tmp_path / "argcheck")
    rocm = "/opt/rocm/lib"
    subprocess.check_call([
        "gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "c_caller", "argcheck.c"), "-o", exe,
        "-L", libdir, "-lstarneig_amd", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath-link,{rocm}",
        f"-Wl,-rpath,{rocm}"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":" + rocm + ":" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "argcheck ok" in out.stdout


def _compile(tmp_path, source, name, extra=()):
    libdir = os.path.join(ROOT, "starneig_amd")
    exe = str(tmp_path / name)
    rocm = "/opt/rocm/lib"
    subprocess.check_call([
        "gcc", "-std=c99", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "c_caller", source), "-o", exe,
        "-L", libdir, "-lstarneig_amd", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath-link,{rocm}",
        f"-Wl,-rpath,{rocm}", *extra])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":" + rocm + ":" + env.get("LD_LIBRARY_PATH", "")
    return exe, env


def test_full_chain_c_program_compiles_and_links(tmp_path):
    """CPU suite: the computing C caller builds against the headers and the product library."""
    import starneig_amd as S
    S.lib.load()
    exe, _ = _compile(tmp_path, "full_chain.c", "full_chain")
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_full_chain_c_program_runs_the_chain_on_the_gpu(tmp_path):
    """A C program -- no Python, no torch in the process -- initialises the node, runs Hessenberg, Schur, Select and
    ReorderSchur through the C-ABI on host arrays and applies the acceptance checks of the reference's example
    (examples/sep_sm_full_chain.c:55-134, examples/validate.c:63-130: residual and orthogonality < 1000 u)."""
    exe, env = _compile(tmp_path, "full_chain.c", "full_chain")
    out = subprocess.run([exe, "1200"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "full_chain ok" in out.stdout, out.stdout + out.stderr


def test_gep_chain_c_program_compiles_and_links(tmp_path):
    import starneig_amd as S
    S.lib.load()
    exe, _ = _compile(tmp_path, "gep_chain.c", "gep_chain")
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_gep_chain_c_program_runs_the_generalized_chain_on_the_gpu(tmp_path):
    """The generalized twin: starneig_GEP_SM_HessenbergTriangular + starneig_GEP_SM_Schur from a C program on host
    arrays (examples/gep_sm_full_chain.c:55-130 without the generalized reordering, which is not built), the
    example's checks on both matrices and both orthogonal factors (< 1000 u), the generalized Schur form entry by
    entry."""
    exe, env = _compile(tmp_path, "gep_chain.c", "gep_chain")
    out = subprocess.run([exe, "900"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gep_chain ok" in out.stdout, out.stdout + out.stderr
