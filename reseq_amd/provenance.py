"""Identity of the kernel sources a measurement was taken with: the counter collections under profiles/ carry it, bench.py compares it with the
sources it runs from and says `counters_stale` when they differ (there is no git on the GPU box, so a content hash stands in for the commit)."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the device-side sources: what decides the kernels' code"""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_CSRC, "*.hip"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
