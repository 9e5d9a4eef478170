"""Builds libfluid_amd.so in-tree with hipcc for gfx950 (csrc/Makefile)."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(PKG, "libfluid_amd.so")


def build(force=False, quiet=True):
    srcdir = os.path.join(PKG, "csrc")
    if not os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists(LIB) and not force:
        return LIB
    args = ["make", "-C", srcdir] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL if quiet else None)
    return LIB
