import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fasta(path):
    """Minimal FASTA reader for the tests: list of (name, uint8 codes A=0,C=1,G=2,T=3,N=4)."""
    lut = np.full(256, 4, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
        lut[ch + 32] = i
    seqs, name, parts = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    seqs.append((name, lut[np.frombuffer(b"".join(parts), np.uint8)]))
                name, parts = line[1:].decode(), []
            elif line:
                parts.append(line)
    if name is not None:
        seqs.append((name, lut[np.frombuffer(b"".join(parts), np.uint8)]))
    return seqs


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return tmp_path_factory.mktemp("rsq")


@pytest.fixture(scope="session", autouse=True)
def kernel_cache_in_the_session_directory(tmp_path_factory):
    """read kernels compiled for a profile (rsq_spec.h) keep their code objects in this session's directory, not under the user's ~/.cache: every profile of the
    suite is compiled once per session, and a test run leaves nothing behind"""
    from reseq_amd import api
    api.set_kernel_cache_dir(str(tmp_path_factory.mktemp("kernel_cache")))
    yield
    api.set_kernel_cache_dir(None)


@pytest.fixture(scope="session")
def tiny_profile_path(workdir):
    from reseq_amd import synth
    path = workdir / "tiny.rsqp"
    synth.write_profile(path, synth.make_profile(synth.TINY, seed=5))
    return str(path)


@pytest.fixture(scope="session")
def tiny_profile_arrays():
    from reseq_amd import synth
    return synth.make_profile(synth.TINY, seed=5)


@pytest.fixture(scope="session")
def p0_profile_path(workdir):
    from reseq_amd import synth
    path = workdir / "p0.rsqp"
    synth.write_profile(path, synth.make_profile(synth.P0, seed=103741084))
    return str(path)


OPTION_DEFAULTS = {"fill_mode": -1, "chain_warmup": -1, "specialize": 1}          # every other option: 0


@pytest.fixture
def rsq_options():
    """set(name, value): an option of the product library and of the host emulation (rsq_set_option; reseq_amd/csrc/rsq_host.h Options),
    back to its default when the test ends"""
    from backends import set_option_everywhere
    changed = []

    def set_(name, value):
        set_option_everywhere(name, value)
        changed.append(name)

    yield set_
    for name in changed:
        set_option_everywhere(name, OPTION_DEFAULTS.get(name, 0))
