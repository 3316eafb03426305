"""gzip on the device (reseq_amd/csrc/rsq_deflate.h; the reference writes .gz through SeqAn's stream from Simulator::Flush, Simulator.cpp:150-182).
CPU suite: the kernels' per-thread walk run by tests/hostemu with the workgroup's threads taken in turn -- every member must inflate (zlib, an independent
implementation) to its piece of the text, carry the BGZF frame, and the whole must stay within 1.3 x the size of zlib level 1 on FASTQ text.
-m gpu: the kernels themselves through the C ABI (rsq_gzip_device), byte-equal to the emulation."""
import ctypes as C
import gzip
import struct
import zlib

import numpy as np
import pytest

from backends import emu_lib
from reseq_amd import synth

PIECE = 65280


def emu_gzip(text, force_stored=False):
    L = emu_lib()
    L.emu_gzip.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    need = C.c_uint64(0)
    assert L.emu_gzip(text, len(text), int(force_stored), None, 0, C.byref(need)) == 0
    out = C.create_string_buffer(max(1, need.value))
    assert L.emu_gzip(text, len(text), int(force_stored), out, need.value, C.byref(need)) == 0
    return out.raw[:need.value]


def members_of(blob):
    """(member bytes, inflated bytes) per member, walking the BGZF frames"""
    at, out = 0, []
    while at < len(blob):
        magic, method, flags, _mtime, _xfl, _os, xlen = struct.unpack_from("<HBBIBBH", blob, at)
        assert (magic, method, flags, xlen) == (0x8b1f, 8, 4, 6), (at, hex(magic), method, flags, xlen)
        si1, si2, slen, bsize = struct.unpack_from("<BBHH", blob, at + 12)
        assert (si1, si2, slen) == (ord("B"), ord("C"), 2)
        member = blob[at:at + bsize + 1]
        d = zlib.decompressobj(-15)
        text = d.decompress(member[18:-8]) + d.flush()
        assert d.eof and not d.unused_data, "one final deflate block fills the member exactly"
        crc, isize = struct.unpack("<II", member[-8:])
        assert crc == zlib.crc32(text) and isize == len(text) and len(text) <= PIECE
        out.append((member, text))
        at += bsize + 1
    assert at == len(blob)
    return out


def fastq_text(n_records, seed=1, read_len=150, quality_values=40):
    rng = np.random.default_rng(seed)
    recs = []
    pos = 1000
    for i in range(n_records):
        pos += int(rng.integers(0, 5))
        seq = bytes(b"ACGT"[c] for c in rng.integers(0, 4, read_len))
        q = rng.integers(0, quality_values, read_len)
        q[rng.random(read_len) < 0.6] = quality_values - 1                     # long runs of the best quality, as real reads have
        qual = bytes(int(x) + 35 for x in q)
        recs.append(b"@ReseqRead%d_%d:%d:synthEcoli0:%d:0:1337:1337 %dM E%d\n%s\n+\n%s\n" % (1 + i // 1000, i % 1000, pos, pos + 350 + int(rng.integers(0, 40)), read_len, int(rng.integers(0, 3)), seq, qual))
    return b"".join(recs)


def test_members_inflate_to_the_text_and_carry_the_bgzf_frame():
    text = fastq_text(1500)                                                          # 8 pieces
    assert len(text) > 7 * PIECE
    blob = emu_gzip(text)
    members = members_of(blob)
    assert b"".join(t for _, t in members) == text and [len(t) for _, t in members[:-1]] == [PIECE] * (len(members) - 1)
    assert gzip.decompress(blob) == text                                             # and as one gzip stream of many members
    level1 = len(zlib.compress(text, 1))
    assert len(blob) <= 1.3 * level1, (len(blob), level1)
    print("device gzip", len(blob), "zlib -1", level1, "zlib -6", len(zlib.compress(text, 6)), "ratio to level 1", round(len(blob) / level1, 3))


@pytest.mark.parametrize("name,text", [
    ("empty", b""),
    ("one byte", b"A"),
    ("three bytes", b"AAA"),
    ("a run longer than a match", b"G" * 5000),
    ("exactly a piece", bytes(range(256)) * 255),
    ("a piece and a byte", b"ACGT" * (PIECE // 4) + b"N"),
    ("a period longer than a group of positions", (b"0123456789abcdefghijklmnopqrstuvwxyz" * 10)[:300] * 400),
    ("text shorter than a hash window at a piece's end", b"x" * (PIECE - 2) + b"yz" + b"END"),
])
def test_edge_cases(name, text):
    blob = emu_gzip(text)
    if not text:
        assert blob == b""
        return
    assert b"".join(t for _, t in members_of(blob)) == text, name
    if len(set(text)) < 40 and len(text) > 1000:
        assert len(blob) < len(text) / 4, (name, len(blob))


def test_text_the_sample_does_not_suit_is_stored():
    """the call's code comes from a sample of its pieces (every 2nd piece of 65 here): a piece of random bytes between pieces of text would need more than its
    slot under that code, and is stored; every member stays within its bound"""
    rng = np.random.default_rng(3)
    pieces = []
    for i in range(130):
        pieces.append(rng.integers(0, 256, PIECE, dtype=np.uint8).tobytes() if i == 1 else fastq_text(180, seed=i)[:PIECE].ljust(PIECE, b"\n"))
    text = b"".join(pieces)
    blob = emu_gzip(text)
    members = members_of(blob)
    assert b"".join(t for _, t in members) == text
    assert len(members[1][0]) == 18 + 5 + PIECE + 8 and members[1][0][18] == 1      # BFINAL, BTYPE 00
    assert all(len(m) <= 65536 for m, _ in members)
    assert b"".join(t for _, t in members_of(emu_gzip(text[:3 * PIECE + 17], force_stored=True))) == text[:3 * PIECE + 17]


def test_the_code_of_a_call():
    """Kraft's sum of both codes is exactly one (a complete code: inflate refuses an over-subscribed one and zlib an incomplete literal code), no code is longer
    than 15 bits, frequent symbols are not longer than rare ones, and symbols the sample never saw still have a code"""
    L = emu_lib()
    L.emu_gzip_code.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(7)
    for trial in range(30):
        sample = np.zeros(320, np.uint32)
        kind = trial % 3
        if kind == 0:                                   # FASTQ-like: a few heavy symbols
            for s, c in ((65, 9000), (67, 9100), (71, 8900), (84, 9050), (10, 400), (256, 8)):
                sample[s] = c
            sample[35:75] = rng.integers(1, 3000, 40)
            sample[257:286] = rng.integers(0, 50, 29)
            sample[288:318] = rng.integers(0, 40, 30)
        elif kind == 1:                                 # counts that force lengths beyond 15 bits before the repair (Fibonacci-like)
            a, b = 1, 1
            for s in range(40):
                sample[s] = min(a, 2 ** 31)
                a, b = b, a + b
        else:
            sample[:286] = rng.integers(0, 2 ** 20, 286)
            sample[288:318] = rng.integers(0, 2 ** 20, 30)
        lengths = np.zeros(316, np.uint8)
        bits = C.c_uint32(0)
        assert L.emu_gzip_code(sample.ctypes.data, lengths.ctypes.data, C.byref(bits)) == 0
        for part, counts in ((lengths[:286], sample[:286]), (lengths[286:], sample[288:318])):
            assert part.min() >= 1 and part.max() <= 15
            assert sum(2.0 ** -int(l) for l in part) == 1.0
            order = np.argsort(counts, kind="stable")
            assert all(part[a] >= part[b] for a, b in zip(order[:-1], order[1:]) if counts[a] < counts[b])
        assert 17 + 4 * 3 <= bits.value <= 128 * 32


# ------------------------------------------------------------------------------------------------------------------------- on the device
@pytest.fixture(scope="module")
def sim(tiny_profile_path):
    from reseq_amd import api
    prof = api.Profile(tiny_profile_path)
    s = api.Simulator(prof, None, 0)
    yield s
    s.close()
    prof.close()


@pytest.mark.gpu
def test_the_kernels_write_what_the_emulation_writes(sim):
    """k_gzip_pieces / k_gzip_stored / k_gzip_compact through rsq_sim_gzip_device: the same bytes as the threads taken in turn on the host (the walk is deterministic:
    candidates by atomicMax, a scan for the bit offsets), members that inflate to the text; texts of one piece, of many, with a stored piece among them"""
    rng = np.random.default_rng(5)
    fastq = fastq_text(2600)
    mixed = b"".join(rng.integers(0, 256, PIECE, dtype=np.uint8).tobytes() if i == 1 else fastq_text(180, seed=i)[:PIECE].ljust(PIECE, b"\n") for i in range(70))
    for name, text in (("fastq", fastq), ("short", fastq[:1000]), ("one byte", b"A"), ("a piece and a byte", fastq[:PIECE + 1]), ("a stored piece among text", mixed),
                       ("runs", b"G" * 200000)):
        got = sim.gzip(text)
        assert b"".join(t for _, t in members_of(got)) == text, name
        assert got == emu_gzip(text), name
    assert sim.gzip(b"") == b""


@pytest.mark.gpu
def test_too_small_an_output_is_refused_with_the_size_needed(sim):
    from reseq_amd import api
    text = fastq_text(900)
    src = api.DeviceArray.from_numpy(0, np.frombuffer(text, np.uint8))
    out = api.DeviceArray(0, 1000)
    n, rc = sim.gzip_device(src, len(text), out, 1000)
    assert rc == api.RSQ_ENOSPC and n == len(emu_gzip(text))
    assert api.lib().rsq_gzip_bound(len(text)) >= n
    src.free()
    out.free()


@pytest.mark.gpu
def test_throughput_and_size_on_fastq_text_of_the_simulator(workdir):
    """the FASTQ text of P0 pairs as the simulator writes it (about 250 MB): every member inflates to its piece, the size is within 1.3 x zlib's level 1 (measured on a
    16 MB sample of the same text) and the kernels pass 20 GB/s of text"""
    import time
    from reseq_amd import api
    import parity_cases as P
    ppath, fpath, _ = P.make_inputs(workdir, "gz_p0", synth.P0, [400000], prof_seed=103741084, ref_seed=2)
    prof, ref = api.Profile(ppath), api.Reference(fpath, 0)
    s = api.Simulator(prof, ref, 0)
    info = s.prepare(11, 700000)
    n, l1, l2, _ = s.pairs_device(1, info.total_blocks + 1, None, None)
    r1, r2 = api.DeviceArray(0, l1 + 64), api.DeviceArray(0, l2 + 64)
    n, l1, l2, rc = s.pairs_device(1, info.total_blocks + 1, r1, r2)
    assert rc == api.RSQ_OK and l1 > 200 << 20
    out = api.DeviceArray(0, l1 // 2)
    s.gzip_device(r1, l1, out, out.nbytes)
    t0 = time.perf_counter()
    size, rc = s.gzip_device(r1, l1, out, out.nbytes)
    wall = time.perf_counter() - t0
    assert rc == api.RSQ_OK
    kernel_s = s.last_kernel_ms("gzip") / 1e3
    text = r1.to_numpy(np.uint8, l1).tobytes()
    assert gzip.decompress(out.to_numpy(np.uint8, size).tobytes()) == text
    sample = text[:16 << 20]
    level1 = len(zlib.compress(sample, 1)) / len(sample)
    print(f"\\ndevice gzip: {l1} bytes of FASTQ text -> {size} ({l1 / size:.2f} x; zlib -1 {1 / level1:.2f} x, -6 {len(sample) / len(zlib.compress(sample, 6)):.2f} x), "
          f"{l1 / kernel_s / 1e9:.1f} GB/s by kernel time, {l1 / wall / 1e9:.1f} GB/s by the call")
    assert size <= 1.3 * level1 * l1
    assert l1 / kernel_s > 20e9
    for d in (r1, r2, out):
        d.free()
    s.close()
    ref.close()
    prof.close()


@pytest.mark.gpu
def test_gz_outputs_of_the_command_line_are_made_on_the_device(workdir):
    """`reseq illuminaPE -1 x.fq.gz` and `reseq seqToIllumina -o y.fq.gz`: members framed by the device's kernels (BGZF's extra field), the inflated files equal to the
    plain outputs and to what zlib on host threads writes with --rsqOption host_gzip:1; a job's kept text compressed on the device spans several arrays"""
    import os
    import subprocess
    import parity_cases as P
    from reseq_amd import api
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "reseq")
    ppath, fpath, _ = P.make_inputs(workdir, "gz_cli", synth.TINY, [60000, 80, 21000])
    common = [exe, "illuminaPE", "-R", fpath, "-s", ppath, "--numReads", "90000", "--seed", "9"]
    names = {k: [str(workdir / f"gzcli_{k}_{m}.fq{ext}") for m in (1, 2)] for k, ext in (("plain", ""), ("device", ".gz"), ("host", ".gz"))}
    subprocess.run(common + ["-1", names["plain"][0], "-2", names["plain"][1]], check=True, capture_output=True)
    subprocess.run(common + ["-1", names["device"][0], "-2", names["device"][1]], check=True, capture_output=True)
    subprocess.run(common + ["-1", names["host"][0], "-2", names["host"][1], "--rsqOption", "host_gzip:1"], check=True, capture_output=True)
    for m in (0, 1):
        plain, device, host = (open(names[k][m], "rb").read() for k in ("plain", "device", "host"))
        assert len(plain) > 5 * PIECE and b":0:Adapter:0:" in plain
        assert b"".join(t for _, t in members_of(device)) == plain == gzip.decompress(host)
        assert host[3] == 0 and device[3] == 4                                    # FLG: zlib writes no extra field, the device's members carry BGZF's
        assert len(device) < 1.35 * len(host)             # TINY's text (30-base reads, five quality values) takes the dense route (gz::dense_pays): 1.07 x zlib level 1, 1.29 x level 6; by FASTQ lines it would be 1.55 x
    # seqToIllumina from file to file
    arrays = synth.make_profile(synth.TINY, seed=5)
    rec = synth.make_error_model_input(9, 30000, 30, arrays, zero_frac=0.7)
    fa = workdir / "gzcli_records.fa"
    fa.write_bytes(P.fasta_of_records(rec, [f"r{i}" for i in range(30000)], wrap_every=3))
    outs = [str(workdir / f"gzcli_records{ext}") for ext in (".fq", ".fq.gz")]
    for o in outs:
        subprocess.run([exe, "seqToIllumina", "-i", str(fa), "-o", o, "-s", ppath, "--seed", "13", "--blockKB", "64"], check=True, capture_output=True)
    packed = open(outs[1], "rb").read()
    assert b"".join(t for _, t in members_of(packed)) == open(outs[0], "rb").read() and len(members_of(packed)) > 3
    # a rank's kept text in arrays of 1 MB: every array becomes members, written at an offset behind other bytes
    api.set_option("job_chunk_bytes", 1 << 20)
    try:
        prof, ref = api.Profile(ppath), api.Reference(fpath, 0)
        s = api.Simulator(prof, ref, 0)
        info = s.prepare(9, 90000)
        n, b1, b2 = s.job_generate(1, info.total_blocks + 1, 7)
        c1, c2 = s.job_compress()
        assert 0 < c1 < b1 / 2 and 0 < c2 < b2 / 2
        g1, g2 = workdir / "gzjob_1.fq.gz", workdir / "gzjob_2.fq.gz"
        g1.write_bytes(gzip.compress(b"in front\n"))
        front = g1.stat().st_size
        g2.write_bytes(b"")
        s.job_write(g1, front, g2, 0)
        s.job_free()
        s.close()
        ref.close()
        prof.close()
    finally:
        api.set_option("job_chunk_bytes", 0)
    plain1 = open(names["plain"][0], "rb").read()
    adapter_at = plain1.index(b":0:Adapter:0:")
    body = plain1[:plain1.rindex(b"@", 0, adapter_at)]                                # the fragments' pairs: what a job holds (adapter-only pairs come behind)
    assert gzip.decompress(g1.read_bytes()) == b"in front\n" + body and g1.stat().st_size == front + c1 and g2.stat().st_size == c2
