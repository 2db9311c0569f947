"""Builds libstarneig_amd.so (HIP, gfx950) in-tree with the committed Makefile."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstarneig_amd.so")
TEST_LIB = os.path.join(HERE, "libstarneig_amd_test.so")     # product objects + test hooks


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(TEST_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(TEST_LIB))
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h", "Makefile")) and os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    inc = os.path.join(HERE, "..", "include")
    for root, _, files in os.walk(inc):
        for f in files:
            if os.path.getmtime(os.path.join(root, f)) > t:
                return True
    return False


def build(force=False, jobs=6):
    if force or needs_build():
        subprocess.check_call(["make", "-C", CSRC, "-j", str(jobs)] + (["-B"] if force else []))
    return LIB


if __name__ == "__main__":
    print(build(force=True))
