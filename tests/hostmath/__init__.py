"""Builds tests/hostmath/hostmath.cpp (a host compile of houv_amd/csrc/houv_math.h) with g++ and
loads it through ctypes.  Test infrastructure only."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    out = os.path.join(_HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhostmath.so")
    src = os.path.join(_HERE, "hostmath.cpp")
    hdr = os.path.join(_HERE, "..", "..", "houv_amd", "csrc", "houv_math.h")
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)
