"""Output files ending in .gz are written by several threads as gzip members of 1 MB of text each (reseq_amd/csrc/rsq_textio.h ParallelGzip): the decompressed bytes
are the plain file's, whatever the number of members.  Run here through the host emulation's --writeSysError file (the same Writer the command line's FASTQ
outputs go through; tests/test_parity_gpu.py::test_cli_gzip_output_and_input does those on the GPU)."""
import gzip
import zlib

import parity_cases as P
from backends import EmuBackend
from reseq_amd import synth


def members(data):
    """the gzip members of a file, decompressed one by one"""
    out = []
    while data:
        d = zlib.decompressobj(16 + zlib.MAX_WBITS)
        out.append(d.decompress(data))
        assert d.eof
        data = d.unused_data
    return out


def test_gzip_members_hold_the_plain_text(workdir):
    ppath, fpath, seqs = P.make_inputs(workdir, "gzout", synth.TINY, [700000, 300, 150000])
    b = EmuBackend(ppath, fpath)
    plain, packed = workdir / "sys.fq", workdir / "sys.fq.gz"
    b.create_sys_error_profile(9, plain)
    b.create_sys_error_profile(9, packed)
    b.close()
    text = plain.read_bytes()
    assert len(text) > 3 << 20
    parts = members(packed.read_bytes())
    assert len(parts) == -(-len(text) // (1 << 20)) and all(len(p) == 1 << 20 for p in parts[:-1])
    assert b"".join(parts) == text == gzip.decompress(packed.read_bytes())


def test_gzip_of_nothing_is_one_empty_member(workdir):
    import ctypes as C
    import os
    from backends import emu_lib
    L = emu_lib()
    L.emu_write_text_file.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    for n in (0, 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 5 * (1 << 20)):
        path = workdir / f"t{n}.gz"
        data = os.urandom(64) * (n // 64) + b"x" * (n % 64)
        assert L.emu_write_text_file(str(path).encode(), data, n) == 0
        assert gzip.decompress(path.read_bytes()) == data
        assert len(members(path.read_bytes())) == max(1, -(-n // (1 << 20)))
