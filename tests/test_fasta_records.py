"""seqToIllumina's input parsed by the device code (reseq_amd/csrc/rsq_fasta.h: record_start, parse_record), run here on the CPU through the host emulation
and compared with a restatement in Python of what the reference does with such a file: SeqAn reads records (id line, sequence lines joined), then
Simulator::ApplyErrorsAndQualityToFastaInput (reseq/Simulator.cpp:2416-2477) takes the id line apart from its END -- two entries as long as the sequence,
semicolons, fragment length, template segment, a blank, the id.  Well-formed files, every complaint of the reference, line ends of both kinds, wrapped
sequences, blocks cut in the middle of a record, and files damaged at random."""
import os
import random

import numpy as np
import pytest

from backends import emu_parse_fasta
from reseq_amd import synth

KINDS = {"too short": 1, "separators": 2, "no id": 3, "segment": 4, "segment separator": 5, "fragment length": 6, "contains N": 7}
CODE = {c: i for i, c in enumerate(b"ACGT")}
CODE.update({c: i for i, c in enumerate(b"acgt")})


def records_of(text):
    """(offset, header, sequence) per record: a record begins at a '>' that begins a line; its lines lose their line end (\\n, or \\r\\n)"""
    starts = [p for p in range(len(text)) if text[p:p + 1] == b">" and (p == 0 or text[p - 1:p] == b"\n")]
    out = []
    for k, p in enumerate(starts):
        body = text[p + 1:starts[k + 1] if k + 1 < len(starts) else len(text)]
        lines = [ln[:-1] if ln.endswith(b"\r") else ln for ln in body.split(b"\n")]
        out.append((p, lines[0], b"".join(lines[1:])))
    return out, (starts[0] if starts else len(text))


def take_apart(header, seq):
    """Simulator.cpp:2423-2477; returns a complaint's name or the fields"""
    L = len(seq)
    if len(header) <= 2 * L + 2:
        return "too short"
    end = len(header) - 2 * L - 3
    if header[end + 1:end + 2] != b";" or header[end + 2 + L:end + 3 + L] != b";":
        return "separators"
    dom, rate = header[end + 2:end + 2 + L], header[len(header) - L:]
    while end and header[end:end + 1] != b" ":
        end -= 1
    if not end:
        return "no id"
    if header[end + 1:end + 2] not in (b"1", b"2"):
        return "segment"
    if header[end + 2:end + 3] != b";":
        return "segment separator"
    number = header[end + 3:len(header) - 2 * L - 2]
    if not number or any(c not in b"0123456789" for c in number):
        return "fragment length"
    if any(c not in CODE for c in seq):
        return "contains N"
    r = np.frombuffer(rate, np.uint8).astype(np.int64) - 33
    r = np.where(r < 0, r + 256, r)
    r = np.where(r > 86, 2 * r - 86, r) & 0xFF
    return {"id_len": end, "seg": int(header[end + 1:end + 2]) - 1, "frag_len": int(number) % 2 ** 32, "seqs": np.array([CODE[c] for c in seq], np.uint8),
            "dom": np.array([CODE.get(c, 4) for c in dom], np.uint8), "rate": r.astype(np.uint8)}


def expect(text, final=True):
    recs, lead_end = records_of(text)
    lead = any(c not in b"\r\n" for c in text[:lead_end])
    n = len(recs) if final or not recs else len(recs) - 1
    consumed = len(text) if final or not recs else recs[-1][0]
    fields, bad = [], None
    for i, (_p, header, seq) in enumerate(recs[:n]):
        f = take_apart(header, seq)
        if isinstance(f, str):
            bad = (i, KINDS[f])
            break
        fields.append(f)
    return n, consumed, lead, bad, fields


def check(text, final=True):
    n, consumed, lead, bad, fields = expect(text, final)
    got = emu_parse_fasta(text, final)
    assert (got["n"], got["consumed"], got["lead"]) == (n, consumed, lead), (text[:200], final)
    if bad:
        assert (got["bad"], got["bad_kind"]) == bad, (text[:300], bad, got["bad"], got["bad_kind"])
        return got
    assert got["bad"] is None, (got["bad"], got["bad_kind"], text[got["at"][got["bad"]]:][:300])
    for i, f in enumerate(fields):
        assert got["len"][i] == len(f["seqs"]) and got["id_len"][i] == f["id_len"] and got["seg"][i] == f["seg"] and got["frag_len"][i] == f["frag_len"], i
        for k in ("seqs", "dom", "rate"):
            assert np.array_equal(got[k][i], f[k]), (i, k)
    # nothing is written outside the records' own stretches [at, at + len) of the three arrays
    mask = np.ones(len(got["packed"]), bool)
    for i in range(n):
        mask[got["at"][i]:got["at"][i] + got["len"][i]] = False
    for arr in got["arrays"]:
        assert np.all(arr[mask] == 0xEE)
    # the device's layout (a half-word per base at the same offsets) holds the same codes
    seqs, dom, rate = got["arrays"]
    want = seqs.astype(np.uint16) | (dom.astype(np.uint16) << 2) | (rate.astype(np.uint16) << 8)
    assert np.array_equal(got["packed"][~mask], want[~mask]) and np.all(got["packed"][mask] == 0xEEEE)
    return got


def fasta_text(seed, n, read_len, wrap=0, crlf=False, ids_with_blanks=True, lower=False, last_newline=True):
    arrays = synth.make_profile(synth.TINY, seed=5)
    rec = synth.make_error_model_input(seed, n, read_len, arrays, zero_frac=0.5)
    rng = random.Random(seed)
    rec["rate"][0, :min(4, read_len)] = [100, 94, 87, 86][:min(4, read_len)]
    lines = []
    for i in range(n):
        rid = f"read {i}/x y" if ids_with_blanks and i % 5 == 0 else f"r{i}"
        seq = "".join("ACGT"[b] for b in rec["seqs"][i])
        if lower and i % 2:
            seq = seq.lower()
        dom = "".join("ACGTN"[b] for b in rec["dom"][i])
        rate = synth.encode_sys_rate(rec["rate"][i]).tobytes().decode()
        lines.append(f">{rid} {int(rec['seg'][i]) + 1};{int(rec['frag_len'][i])};{dom};{rate}")
        if wrap and i % 3 == 0:
            w = rng.randint(1, wrap)
            lines += [seq[k:k + w] for k in range(0, len(seq), w)]
            if i % 6 == 0:
                lines.append("")                 # a blank line inside the file
        else:
            lines.append(seq)
    nl = "\r\n" if crlf else "\n"
    return (nl.join(lines) + (nl if last_newline else "")).encode()


def test_eight_bytes_at_a_time_equal_one_at_a_time():
    """base_codes / rate_percents / zero_bytes of rsq_fasta.h on every byte value in every place of a word"""
    from backends import emu_lib
    assert emu_lib().emu_fasta_words_check() == 0


@pytest.mark.parametrize("read_len", [1, 7, 8, 9, 30, 75, 151])
def test_well_formed_records(read_len):
    """lengths around the 8-byte groups the codes are written in; ids with blanks, rates above 86 percent (stored halved, Simulator.cpp:2439-2442)"""
    got = check(fasta_text(read_len, 40, read_len))
    assert got["n"] == 40


def test_line_ends_wrapped_sequences_and_case():
    for kw in (dict(wrap=20), dict(crlf=True), dict(wrap=9, crlf=True), dict(lower=True), dict(last_newline=False), dict(crlf=True, last_newline=False),
               dict(wrap=5, crlf=True, last_newline=False)):
        got = check(fasta_text(3, 31, 30, **kw))
        assert got["n"] == 31, kw
    text = b"\n\r\n" + fasta_text(4, 5, 12)            # line ends in front of the first record
    assert check(text)["n"] == 5


def test_two_lengths_and_a_record_without_bases():
    text = fasta_text(1, 20, 30) + fasta_text(2, 7, 75, wrap=11) + b">x 1;5;;\n" + fasta_text(3, 9, 30)
    got = check(text)
    assert got["n"] == 37 and sorted(set(got["len"].tolist())) == [0, 30, 75]


def test_a_block_that_ends_inside_a_record():
    """final=0: the block's last record is left to the caller, whatever is there of it; a block without a second record start is not consumed at all"""
    text = fasta_text(8, 12, 30, wrap=7)
    whole = check(text)
    for cut in (len(text), len(text) - 1, len(text) - 40, int(whole["at"][5]) + 3, int(whole["at"][1]), int(whole["at"][1]) - 1, 5, 1):
        got = check(text[:cut], final=False)
        assert got["consumed"] == max(int(a) for a in whole["at"][:-1] if a < cut)
        assert got["n"] == sum(1 for a in whole["at"][:-1] if a < cut) - 1
    # in pieces: what a caller does with `consumed`
    for piece in (100, 333, 1000):
        fields, pos, rest = [], 0, b""
        while pos < len(text) or rest:
            block = rest + text[pos:pos + piece]
            pos += piece
            final = pos >= len(text)
            got = check(block, final=final)
            fields += [(int(got["len"][i]), int(got["frag_len"][i]), got["seqs"][i].tobytes()) for i in range(got["n"])]
            rest = block[got["consumed"]:]
            if final:
                break
        assert fields == [(int(whole["len"][i]), int(whole["frag_len"][i]), whole["seqs"][i].tobytes()) for i in range(whole["n"])]


def test_every_complaint_of_the_reference():
    good = fasta_text(5, 6, 10)
    cases = {">r 3;40;NNNN;!!!!\nACGT\n": "segment", ">r1;40;NNNN;!!!!\nACGT\n": "no id", ">r 1;4x;NNNN;!!!!\nACGT\n": "fragment length", ">r 1;;NNNN;!!!!\nACGT\n": "fragment length",
             ">r 1;40;NNN;!!!!\nACGT\n": "separators", ">r 1;40;NNNN!!!!!\nACGT\n": "separators", ">r\nACGT\n": "too short", ">r 1;40;NNNN;!!!!\nACNT\n": "contains N",
             ">r 1,40;NNNN;!!!!\nACGT\n": "segment separator", ">r 1;40;NNNN;!!!!\n": "separators", "> 1;40;NNNN;!!!!\nACGT\n": "no id", ">\n": "too short", ">": "too short",
             ">r 1;40;NNNN;!!!!\nAC GT\n": "separators"}
    for bad, kind in cases.items():
        for text in (bad.encode(), good + bad.encode()) + ((good + bad.encode() + good, good + bad.encode() + b">r\nACGT\n") if bad.endswith("\n") else ()):
            got = check(text)
            assert got["bad_kind"] == KINDS[kind], (bad, got["bad_kind"])
    assert check(b"ACGT\n" + good)["lead"] and check(b"ACGT\n")["lead"] and not check(b"\n\n")["lead"]


def test_files_damaged_at_random():
    """bytes replaced, dropped and inserted at random places of a well-formed file: the first complaint and its record, or every field, agree"""
    rounds = int(os.environ.get("RSQ_FUZZ", "1")) * 150
    rng = random.Random(2024)
    base = [fasta_text(11, 14, 20, wrap=6), fasta_text(12, 9, 33, crlf=True), fasta_text(13, 30, 8)]
    complaints = set()
    for r in range(rounds):
        text = bytearray(base[r % 3])
        for _ in range(rng.randint(1, 3)):
            p = rng.randrange(len(text))
            what = rng.random()
            c = rng.choice(b";; >>\n\r12ACGTN!x0")
            if what < 0.5:
                text[p] = c
            elif what < 0.75:
                del text[p]
            else:
                text.insert(p, c)
        got = check(bytes(text), final=bool(r % 4))
        complaints.add(got["bad_kind"])
    assert len(complaints) >= 6, complaints
