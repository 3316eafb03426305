"""RSQP named-array container: the flat on-disk form of a PREPARED ReSeq profile.

The reference keeps a fitted profile in two Boost text archives (`.reseq`:
DataStats.h:180-212, `.reseq.ipf`: ProbabilityEstimates.h:1475-1483).  This
build's primary format is a flat, little-endian sequence of named arrays that
C (oracle/oracle_base.c), C++ (reseq_amd/csrc/rsq_host.cpp, written by
rsq_profile_archive.cpp) and numpy read with the same
20-line loop.  Field names follow the reference's member names.

Layout
------
    8s   magic  "RSQPROF1"
    u32  version (1)
    u32  number of records
    then per record, each field starting on an 8-byte boundary:
        u16  name length, name bytes (no NUL)
        u8   dtype code, u8 ndim, u64 dims[ndim]
        raw little-endian data
"""
import struct

import numpy as np

MAGIC = b"RSQPROF1"
VERSION = 1

_DTYPES = [
    np.dtype("<u1"), np.dtype("<u2"), np.dtype("<u4"), np.dtype("<u8"),
    np.dtype("<i4"), np.dtype("<i8"), np.dtype("<f8"),
]
_CODE = {dt: i for i, dt in enumerate(_DTYPES)}


def _pad8(n):
    return (-n) % 8


def write_container(path, arrays):
    """Write `arrays` (dict name -> ndarray / scalar) in insertion order."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, len(arrays)))
        for name, value in arrays.items():
            a = np.ascontiguousarray(value)
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            dt = np.dtype(dt.str.replace("=", "<").replace("|", "<"))
            if dt not in _CODE:
                raise TypeError(f"{name}: unsupported dtype {a.dtype}")
            a = a.astype(dt, copy=False)
            nb = name.encode()
            head = struct.pack("<H", len(nb)) + nb + struct.pack("<BB", _CODE[dt], a.ndim)
            f.write(head)
            f.write(b"\0" * _pad8(len(head)))
            f.write(struct.pack(f"<{a.ndim}Q", *a.shape))
            data = a.tobytes()
            f.write(data)
            f.write(b"\0" * _pad8(len(data)))


def read_container(path):
    """Read a container back into an ordered dict name -> ndarray."""
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC:
        raise ValueError(f"{path}: not an RSQP container")
    version, n = struct.unpack_from("<II", buf, 8)
    if version != VERSION:
        raise ValueError(f"{path}: unsupported version {version}")
    pos = 16
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", buf, pos)
        name = buf[pos + 2:pos + 2 + ln].decode()
        code, ndim = struct.unpack_from("<BB", buf, pos + 2 + ln)
        head = 2 + ln + 2
        pos += head + _pad8(head)
        dims = struct.unpack_from(f"<{ndim}Q", buf, pos)
        pos += 8 * ndim
        dt = _DTYPES[code]
        count = int(np.prod(dims)) if ndim else 1
        nbytes = count * dt.itemsize
        out[name] = np.frombuffer(buf, dtype=dt, count=count, offset=pos).reshape(dims).copy()
        pos += nbytes + _pad8(nbytes)
    return out
