"""Loader of probes/libseeme_probes.so (micro-benchmarks behind the design decisions in DESIGN.md; not product code)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib() -> C.CDLL:
    path = os.path.join(_HERE, "libseeme_probes.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run probes/build.sh")
    return C.CDLL(path)


def check(rc, what="probe"):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc})")
