"""CPU suite: a C program compiled against include/starneig/starneig.h and linked with
-lstarneig_amd reproduces the reference's argument-check table (no GPU: the node is never
initialised, every check precedes the STARNEIG_NOT_INITIALIZED test)."""
import os
import subprocess

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
