"""Shared pytest plumbing: markers, library loaders.

`-m "not gpu"` covers the oracle against golden vectors / the reference build and the
host logic; `-m gpu` tests are the parity tests proper and call through the C-ABI.
"""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _ensure_oracle():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith("_oracle.c") or f.endswith(".h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"),
                               os.path.join(ROOT, "oracle", "liboracle.so")])
    return so


@pytest.fixture(scope="session")
def oracle():
    import vp9ref
    return vp9ref.load_oracle(_ensure_oracle())


@pytest.fixture(scope="session")
def ref():
    """The reference's own object code (oracle/_ref), when it has been built."""
    import vp9ref
    so = os.path.join(ROOT, "oracle", "_ref", "libvpxref.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/libvpx"):
            subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libvpxref.so not built (reference tree absent)")
    return vp9ref.load_ref(so)


@pytest.fixture(scope="session")
def hip():
    """The product: libvp9hip.so through its C-ABI.  Fails loudly when missing."""
    import vp9ref
    return vp9ref.load_hip()
