"""GPU parity: the product (libreseq_amd.so through the C ABI, reseq_amd.api) against the CPU oracle on the same
seeded inputs.  Same cases as tests/test_parity_hostemu.py (tests/parity_cases.py)."""
import pytest

import parity_cases as P
from backends import GpuBackend

pytestmark = pytest.mark.gpu


def test_device_visible():
    from reseq_amd import api
    assert api.device_count() >= 1


def test_reference_packing(workdir):
    P.case_reference_packing(GpuBackend, workdir)


def test_prepass(workdir):
    P.case_prepass(GpuBackend, workdir)


def test_sieve_and_reads_tiny(workdir):
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)


def test_sieve_own_thresholds(workdir):
    P.case_sieve_own_thresholds(GpuBackend, workdir)


def test_profile_from_reseq_archive(workdir):
    P.case_profile_from_reseq_archive(GpuBackend, workdir)


def test_coverage_driven(workdir):
    P.case_coverage_driven(GpuBackend, workdir)


@pytest.mark.parametrize("kind", ["subs", "indels", "meth"])
def test_p0_variants(workdir, kind):
    P.case_p0_variants(GpuBackend, workdir, kind)


def test_dense_coverage(workdir):
    P.case_dense_coverage(GpuBackend, workdir)


def test_adapter_only(workdir):
    P.case_adapter_only(GpuBackend, workdir)


def test_more_quality_values_than_the_screen_is_built_for(workdir):
    P.case_more_quality_values_than_the_screen_is_built_for(GpuBackend, workdir)
    from reseq_amd import api
    assert "double precision" in api.last_warning()                      # rsq_sim_create said why this profile takes the slow route


def test_p0_reads(workdir):
    P.case_p0_reads(GpuBackend, workdir)


def test_indel_draw_decided_by_the_word_alone_with_no_indel_in_a_middle_column(workdir):
    P.case_indel_columns_shuffled(GpuBackend, workdir)


def test_tables_over_their_own_value_ranges(workdir):
    from reseq_amd import synth
    P.case_ragged_tables(GpuBackend, workdir)
    P.case_ragged_tables(GpuBackend, workdir, cfg_base=synth.P0, lengths=(20000,), num_pairs=1000)


def test_packed_reference_of_another_simulator(workdir):
    P.case_packed_reference(GpuBackend, workdir)


def test_profile_edits(workdir):
    P.case_profile_edits(GpuBackend, workdir)


def test_error_model_tiny(workdir):
    P.case_error_model_tiny(GpuBackend, workdir)


def test_error_model_long_templates(workdir):
    P.case_error_model_long_templates(GpuBackend, workdir)


def test_error_model_templates_beyond_the_staging(workdir):
    P.case_error_model_templates_beyond_the_staging(GpuBackend, workdir)


def test_fasta_records_parsed_where_they_lie_in_hbm(workdir, rsq_options):
    """k_fasta_records has two instantiations of parse_record: on the workgroup's stretch of text staged in LDS, and on the text where it lies in HBM (records too
    long for the staging).  hipcc 7.2 once lost a field of one of them (rsq_fasta.h, the note above parse_record's fields): every FASTA case through BOTH, the second
    forced with option fasta_no_stage"""
    rsq_options("fasta_no_stage", 1)
    P.case_error_model_tiny(GpuBackend, workdir)
    P.case_error_model_p0(GpuBackend, workdir)
    P.case_error_model_long_templates(GpuBackend, workdir)


def test_a_block_of_fasta_text_without_a_record_start(workdir, tiny_profile_path):
    """include/reseq_amd.h: a block that is not the last and holds no record start consumes nothing and yields no record (the caller hands in more); as the last block
    it is consumed whole"""
    from reseq_amd import api
    prof = api.Profile(tiny_profile_path)
    s = api.Simulator(prof, None, 0)
    s.prepare(5)
    try:
        assert s.error_model_fasta(b"\n\n\n", final=False) == (b"", 0, 0)
        assert s.error_model_fasta(b"\n\n\n", final=True) == (b"", 0, 3)
    finally:
        s.close()
        prof.close()


def test_error_model_fasta_on_damaged_files(workdir):
    """rsq_sim_error_model_fasta on files damaged at random (bytes replaced, dropped, inserted): the reference's complaint about the first malformed record as the
    restatement in tests/test_fasta_records.py expects it -- or, if the file is still well-formed, the text of rsq_sim_error_model_fastq on the fields that
    restatement reads (runs of equal template length; that entry point is pinned to the oracle by the cases above)"""
    import os
    import random
    import re
    import numpy as np
    import test_fasta_records as F
    from reseq_amd import synth
    ppath, _, _ = P.make_inputs(workdir, "em_tiny", synth.TINY, [100], prof_seed=5)
    b = GpuBackend(ppath, None)
    b.prepare(19)
    words = {1: "too short to contain", 2: "not separated by a semicolon from themselves", 3: "No sequence id found", 4: "not 1 or 2", 5: "template segment and fragment length are not separated",
             6: "is not a pure integer", 7: "must not contain N"}
    rng = random.Random(77)
    base = [F.fasta_text(11, 300, 20, wrap=6), F.fasta_text(12, 90, 33, crlf=True), F.fasta_text(13, 600, 8), F.fasta_text(14, 40, 330)]
    seen, well_formed = set(), 0
    try:
        for r in range(int(os.environ.get("RSQ_FUZZ", "1")) * 120):
            text = bytearray(base[r % 4])
            for _ in range(rng.randint(0, 2)):
                at = rng.randrange(len(text))
                what, c = rng.random(), rng.choice(b";; >>\n\r12ACGTN!x0")
                if what < 0.5:
                    text[at] = c
                elif what < 0.75:
                    del text[at]
                else:
                    text.insert(at, c)
            text = bytes(text)
            n, _consumed, lead, bad, fields = F.expect(text)
            if lead or bad:
                with pytest.raises(Exception) as e:
                    b.error_model_fasta(text, first_index=5)
                assert ("without a header" if lead else words[bad[1]]) in str(e.value), (r, bad, str(e.value))
                seen.add("lead" if lead else bad[1])
                continue
            recs, _ = F.records_of(text)
            try:
                got, k, used = b.error_model_fasta(text, first_index=5)
            except Exception as e:                        # a damaged fragment length outside the profile's tables: the words of the reference's Vect::at (Vect.hpp:196-221)
                m = re.search(r"record (\d+) of the call: Called index (\d+) range is from (\d+) to (\d+)", str(e))
                assert m, str(e)
                i, fl, lo, hi = (int(v) for v in m.groups())
                assert fields[i]["frag_len"] == fl and not lo <= fl < hi and all(lo <= f["frag_len"] < hi for f in fields[:i]), str(e)
                seen.add("fragment length outside the profile")
                continue
            assert (k, used) == (n, len(text))
            want, i = [], 0
            while i < n:                                      # runs of one template length
                j = i
                while j < n and len(fields[j]["seqs"]) == len(fields[i]["seqs"]):
                    j += 1
                L = len(fields[i]["seqs"])
                if L:
                    rec = {key: np.stack([fields[q][key] for q in range(i, j)]) for key in ("seqs", "dom", "rate")}
                    rec["seg"] = np.array([fields[q]["seg"] for q in range(i, j)], np.uint8)
                    rec["frag_len"] = np.array([fields[q]["frag_len"] for q in range(i, j)], np.uint32)
                    want.append(b.error_model_fastq(rec, [recs[q][1][:fields[q]["id_len"]] for q in range(i, j)], first_index=5 + i))
                else:
                    want = None                               # a record without bases: the array entry point has no shape for it
                    break
                i = j
            if want is not None:
                assert got == b"".join(want), r
                well_formed += 1
    finally:
        b.close()
    assert len(seen) >= 5 and well_formed >= 20, (seen, well_formed)


def test_error_model_p0(workdir):
    P.case_error_model_p0(GpuBackend, workdir)


def test_error_model_p0_at_the_borders_of_the_eight_byte_groups(workdir):
    P.case_error_model_p0_groups(GpuBackend, workdir)


@pytest.mark.parametrize("n_tiles", [3, 96])
def test_p0_with_tiles(workdir, n_tiles):
    """--tiles profiles: one tile's tables per workgroup image, reads binned by tile (k_fill_reads<MASK, VAR, true>, k_fill_records<MASK, true>)"""
    P.case_p0_tiles(GpuBackend, workdir, n_tiles, num_pairs=1500 if n_tiles == 3 else 6000)


def test_tiles_binned_although_they_fit(workdir, rsq_options):
    """TINY's three tiles fit one image; option image_tiles = 1 bins them all the same: the scheduler (bins chosen by the workgroups, re-staged
    images, bins shorter than a chunk) on profiles with adapters, variable read lengths and indels"""
    rsq_options("image_tiles", 1)
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_dense_coverage(GpuBackend, workdir)
    P.case_adapter_only(GpuBackend, workdir)
    P.case_error_model_tiny(GpuBackend, workdir)
    P.case_variants_indels(GpuBackend, workdir)
    P.case_methylation(GpuBackend, workdir)


def test_job_text_kept_on_the_device_and_written_in_place(workdir, rsq_options):
    """rsq_sim_job_generate / rsq_sim_job_write (a rank's share of a multi-GPU job): the text of a block range generated in several calls, kept in device arrays
    (here small ones, so that it spans many and calls outgrow their array), written by several threads per file at an offset of files that hold other bytes"""
    import numpy as np
    from parity_cases import make_inputs
    from reseq_amd import synth
    ppath, fpath, _ = make_inputs(workdir, "tiny_e2e", synth.TINY, [5000, 80, 3210])
    b = GpuBackend(ppath, fpath)
    info = b.prepare(7, num_pairs=3000)
    lo, hi = 2, info["total_blocks"] + 1
    frags, t1, t2 = b.pairs(lo, hi)
    import pytest
    from reseq_amd import api
    with pytest.raises(api.RsqError) as e:                                # nothing generated yet: refused, not two files of zeros
        b.sim.job_write(workdir / "none_1.fq", 0, workdir / "none_2.fq", 0, 0)
    assert e.value.code == api.RSQ_ESTATE and "no generated text" in str(e.value)
    with pytest.raises(api.RsqError) as e:                                # an inverted range is refused before anything is sized from it
        b.sim.job_generate(hi, lo, 0)
    assert e.value.code == api.RSQ_EINVAL
    for chunk_bytes, batch, threads, direct in ((0, 0, 0, 0), (30_000, 2, 3, 0), (5_000, 1, 5, 0), (30_000, 2, 2, 1), (0, 0, 1, 1)):
        rsq_options("job_chunk_bytes", chunk_bytes)
        rsq_options("job_write_direct", direct)                               # whole 4 KB blocks around the page cache where the file system allows it, head and tail buffered
        b.sim.take_options()                                                  # (a simulator keeps the switches it was created with until told otherwise)
        n, n1, n2 = b.sim.job_generate(lo, hi, batch)
        assert (n, n1, n2) == (len(frags), len(t1), len(t2))
        f1, f2 = workdir / "job_1.fq", workdir / "job_2.fq"
        f1.write_bytes(b"x" * (len(t1) + 300))
        f2.write_bytes(b"y" * 50)                                     # shorter than offset + text: pwrite extends it
        b.sim.job_write(f1, 100, f2, 70, threads)
        assert f1.read_bytes() == b"x" * 100 + t1 + b"x" * 200
        assert f2.read_bytes() == b"y" * 50 + bytes(20) + t2
        b.sim.job_free()
        with pytest.raises(api.RsqError):                                 # freed: nothing to write
            b.sim.job_write(f1, 100, f2, 70, threads)
    # rsq_sim_job_compress: the kept text as gzip members, written at offsets like the plain text (a rank's share of .gz outputs) -- made on the device and kept there
    # (the arrays of text become arrays of members, which rsq_sim_job_read serves as well), or with option host_gzip by zlib on host threads into host memory
    import gzip
    rsq_options("job_chunk_bytes", 30_000)
    rsq_options("job_write_direct", 0)
    for host_gzip in (0, 1):
        rsq_options("host_gzip", host_gzip)
        b.sim.take_options()
        b.sim.job_generate(lo, hi, 2)
        c1, c2 = b.sim.job_compress()
        assert 0 < c1 < len(t1) // 2 and 0 < c2 < len(t2) // 2
        with pytest.raises(api.RsqError) as e:                                # once only
            b.sim.job_compress()
        assert e.value.code == api.RSQ_ESTATE
        scratch = api.DeviceArray(0, c1)
        if host_gzip:                                                         # in host memory: nothing on the device to read
            with pytest.raises(api.RsqError) as e:
                b.sim.job_read(0, 0, 16, scratch.ptr.value)
            assert e.value.code == api.RSQ_ESTATE and "compressed" in str(e.value)
        else:
            b.sim.job_read(0, 0, c1, scratch.ptr.value)
            assert gzip.decompress(scratch.to_numpy(np.uint8, c1).tobytes()) == t1
        scratch.free()
        g1, g2 = workdir / f"job{host_gzip}_1.fq.gz", workdir / f"job{host_gzip}_2.fq.gz"
        g1.write_bytes(gzip.compress(b"in front\n"))
        front = len(g1.read_bytes())
        b.sim.job_write(g1, front, g2, 0, 0)
        assert gzip.decompress(g1.read_bytes()) == b"in front\n" + t1 and len(g1.read_bytes()) == front + c1
        assert gzip.decompress(g2.read_bytes()) == t2 and len(g2.read_bytes()) == c2
    b.sim.job_free()
    b.close()


@pytest.mark.parametrize("parts", [2, 3, 7])
def test_pipelined_sub_ranges(workdir, parts, rsq_options):
    """rsq_sim_pairs over a large block range runs as sub-ranges whose sieve / reads / text stages overlap on three streams (two workspaces, offsets
    of a part continuing where the part before ended); option overlap = n forces n parts on these small cases: same fragments, same bytes,
    also with parts that hold no pair at all (the 80-base sequence has no block with fragments), with variants, methylation and tiles"""
    rsq_options("overlap", parts)
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)
    P.case_dense_coverage(GpuBackend, workdir)
    P.case_variants_indels(GpuBackend, workdir)
    P.case_variants_substitutions(GpuBackend, workdir)
    P.case_methylation(GpuBackend, workdir)
    P.case_p0_tiles(GpuBackend, workdir, 3)


@pytest.mark.parametrize("mode", [0])
def test_double_precision_path(workdir, mode, rsq_options):
    """k_fill_reads<0>: every draw in double precision from HBM, the reference's recipe itself; the default of the other tests is the
    screened path (single-precision draws on the LDS image, double precision only where the screen cannot decide)"""
    rsq_options("fill_mode", mode)
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)


@pytest.mark.gpu
def test_every_draw_through_the_route_behind_the_screen(workdir, rsq_options):
    """option force_exact: the screen decides nothing, so every draw of the read kernel (and of seqToIllumina's) takes the route of an undecided one,
    the call of the double-precision recipe (exact_draw_call)"""
    rsq_options("force_exact", 1)
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)
    P.case_indel_columns_shuffled(GpuBackend, workdir)
    P.case_error_model_p0(GpuBackend, workdir)
    P.case_adapter_only(GpuBackend, workdir)


@pytest.mark.gpu
def test_read_kernel_compiled_for_the_profile_and_the_library_instantiation(workdir, rsq_options):
    """Every other test runs the read kernels compiled for the loaded profile at run time (hiprtc: rsq_sim_specialize says so).  Here: that this IS what runs, that a
    second simulator of the same profile takes the code object from the kernel cache, and -- option specialize 0 -- the same cases through the library's own
    instantiations (any profile's shapes as run-time values)"""
    from reseq_amd import api, synth
    api.set_kernel_cache_dir(str(workdir / "kernel_cache"))
    try:
        ppath, fpath, _ = P.make_inputs(workdir, "spec", synth.TINY, [5000, 3210])
        notes = []
        for _ in range(2):
            prof, ref = api.Profile(ppath), api.Reference(fpath, 0)
            sim = api.Simulator(prof, ref, 0)
            sim.prepare(7, 3000)
            done, note = sim.specialize(0)
            assert done and "rsq_spec_fill_reads compiled for this profile" in note, note
            done, note1 = sim.specialize(1)
            assert done and "rsq_spec_fill_records" in note1, note1
            notes.append(note)
            sim.close(), ref.close(), prof.close()
        assert " ms)" in notes[0] and "kernel cache" in notes[1], notes
    finally:
        api.set_kernel_cache_dir(str(workdir / "kernel_cache_session"))
    rsq_options("specialize", 0)
    prof, ref = api.Profile(ppath), api.Reference(fpath, 0)
    sim = api.Simulator(prof, ref, 0)
    sim.prepare(7, 3000)
    done, note = sim.specialize(0)
    assert not done and "specialize is 0" in note
    sim.close(), ref.close(), prof.close()
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)
    P.case_ragged_tables(GpuBackend, workdir)
    P.case_p0_tiles(GpuBackend, workdir, 3)
    P.case_error_model_p0(GpuBackend, workdir)
    P.case_variants_indels(GpuBackend, workdir)
    P.case_adapter_only(GpuBackend, workdir)


@pytest.mark.gpu
def test_error_rate_rows_fall_back_to_hbm(workdir, rsq_options):
    """only row 0 of the error-rate margins staged: every position with a systematic error rate takes the HBM branch"""
    rsq_options("rate_rows", 1)
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)


def test_sys_error_profile_round_trip(workdir):
    P.case_sys_error_profile_round_trip(GpuBackend, workdir)


def test_sys_error_profile_rejects_wrong_reference(workdir):
    P.case_sys_error_profile_rejects_wrong_reference(GpuBackend, workdir)


def test_ref_bias_modes(workdir):
    P.case_ref_bias_modes(GpuBackend, workdir)


def test_bias_sums_in_windows_are_the_sums_at_once(workdir, rsq_options):
    """the bias sums run over the reference in windows whose surrounding tracks reuse two buffers (rsq_sim.hip bias_partials); a chunk's sum does
    not depend on the window it is computed in: windows of 700 positions (several per sequence, borders inside chunks' reach) give the same
    doubles as one window, also for a share of a sharded pre-pass"""
    import numpy as np
    from parity_cases import make_inputs
    from reseq_amd import synth
    lengths = [5200, 90, 3100, 2048]
    ppath, fpath, _ = make_inputs(workdir, "biaswin", synth.TINY, lengths)
    got = []
    for window in (0, 700, 64):
        rsq_options("bias_window", window)
        b = GpuBackend(ppath, fpath)
        info = b.prepare(5, num_pairs=3000)
        thr = np.array(b.thresholds())
        b.close()
        b = GpuBackend(ppath, fpath)
        b.prepare_plan(5, num_pairs=3000)
        sums, maxes = b.bias_partials(5, b.info()["total_blocks"] + 1)        # from block 5 of the first sequence on: the chunks of the later sequences
        b.close()
        got.append((info["bias_normalization"], thr, np.array(sums), np.array(maxes)))
    for norm, thr, sums, maxes in got[1:]:
        assert norm == got[0][0]
        assert np.array_equal(thr, got[0][1])
        assert np.array_equal(sums, got[0][2]) and np.array_equal(maxes, got[0][3])
    assert 0 < np.count_nonzero(got[0][2]) < len(got[0][2])


def test_cli_write_then_read_sys_error_profile(workdir):
    """reseq illuminaPE --writeSysError f simulates from the profile it just wrote (main.cpp:389), so a second run with
    --readSysError f and the same seed must produce the same FASTQ files; --refBias no is accepted"""
    import os
    import subprocess
    from reseq_amd import synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reseq_amd", "reseq")
    ppath, fpath, _ = P.make_inputs(workdir, "cli_sys", synth.TINY, [4100, 2999])
    prof = workdir / "cli_sys.fq"
    common = [exe, "illuminaPE", "-R", fpath, "-s", ppath, "--numReads", "1500", "--seed", "5", "--refBias", "no"]
    a1, a2, b1, b2 = (str(workdir / n) for n in ("a1.fq", "a2.fq", "b1.fq", "b2.fq"))
    subprocess.run(common + ["-1", a1, "-2", a2, "--writeSysError", str(prof)], check=True, capture_output=True)
    subprocess.run(common + ["-1", b1, "-2", b2, "--readSysError", str(prof)], check=True, capture_output=True)
    assert open(a1, "rb").read() == open(b1, "rb").read() and open(a2, "rb").read() == open(b2, "rb").read()
    assert open(a1, "rb").read().count(b"\n") >= 4 * 1000
    assert prof.read_bytes().count(b"\n") == 16
    r = subprocess.run(common + ["-1", a1, "-2", a2, "--writeSysError", str(prof), "--readSysError", str(prof)], capture_output=True)
    assert r.returncode != 0 and b"mutually exclusive" in r.stderr


def test_cli_seq_to_illumina_equals_the_oracle(workdir):
    """reseq seqToIllumina (Simulator::ApplyErrorsAndQualityToFastaInput, Simulator.cpp:2403-2512): FASTA records
    "{id} {1|2};{fragment length};{dominant errors};{error rates}" -> FASTQ "@{id} {CIGAR} E{errors}"; ids with blanks, error rates above
    86 % (stored halved, Simulator.cpp:2439-2442), wrapped sequence lines, two template lengths in one file, input order kept"""
    import os
    import subprocess
    import numpy as np
    import oracle_lib as O
    from reseq_amd import synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reseq_amd", "reseq")
    ppath, _, _ = P.make_inputs(workdir, "cli_s2i", synth.TINY, [100])
    arrays = synth.make_profile(synth.TINY, seed=5)
    parts = [synth.make_error_model_input(21, 700, 30, arrays, zero_frac=0.6), synth.make_error_model_input(22, 60, 75, arrays, zero_frac=0.5),
             synth.make_error_model_input(23, 300, 30, arrays, zero_frac=0.9)]
    parts[0]["rate"][3, 5] = 94                     # stored as 90 + 33: decompressed to 94 again
    parts[0]["rate"][4, 0] = 100                    # the largest: stored as 93 + 33 = '~'
    parts[0]["rate"][5, 7] = 87                     # odd rates above 86 lose their last bit in the file
    fasta, ids = [], []
    for rec in parts:
        for i in range(len(rec["seg"])):
            rid = f"read {len(ids)}/x" if len(ids) % 7 == 0 else f"r{len(ids)}"
            ids.append(rid)
            seq = "".join("ACGT"[b] for b in rec["seqs"][i])
            dom = "".join("ACGTN"[b] for b in rec["dom"][i])
            rate = synth.encode_sys_rate(rec["rate"][i]).tobytes().decode()
            fasta.append(f">{rid} {int(rec['seg'][i]) + 1};{int(rec['frag_len'][i])};{dom};{rate}")
            fasta += [seq[k:k + 20] for k in range(0, len(seq), 20)] if len(ids) % 3 == 0 else [seq]
    inp, out = workdir / "s2i.fa", workdir / "s2i.fq"
    inp.write_text("\n".join(fasta) + "\n")
    subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", ppath, "--seed", "77"], check=True, capture_output=True)
    oprof = O.Profile(ppath)
    want, first = [], 0
    for rec in parts:
        rec = dict(rec)
        r = rec["rate"].astype(np.int64)            # what survives the file: odd percents above 86 become the even one below
        rec["rate"] = np.where(r > 86, r - r % 2, r).astype(np.uint8)
        for seq, qual, cigar, nerr, _tile in O.error_model_only(oprof, 77, rec, first_index=first):
            want.append(f"@{ids[len(want)]} {cigar} E{nerr}\n" + "".join("ACGTN"[b] for b in seq) + "\n+\n" + qual.decode() + "\n")
        first += len(rec["seg"])
    oprof.close()
    got = out.read_text()
    assert got == "".join(want)
    # replaceQuals is the same mode (BASELINE.json's name for it); stdin / stdout without -i / -o
    r = subprocess.run([exe, "replaceQuals", "-s", ppath, "--seed", "77"], input=inp.read_bytes(), check=True, capture_output=True)
    assert r.stdout.decode() == got
    # blocks far smaller than the file (the pipeline's default is 48 MB): several readers at offsets, records across block ends put together on the device, one
    # block or several in a call (records longer than a block among them); a gzip file, which one reader reads in sequence
    for extra in (["--blockKB", "4", "--readThreads", "3"], ["--blockKB", "1", "--batchBlocks", "1", "--readThreads", "5"], ["--blockKB", "7", "--batchBlocks", "3"]):
        subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", ppath, "--seed", "77"] + extra, check=True, capture_output=True)
        assert out.read_text() == got, extra
    r = subprocess.run([exe, "replaceQuals", "-s", ppath, "--seed", "77", "--blockKB", "3"], input=inp.read_bytes(), check=True, capture_output=True)
    assert r.stdout.decode() == got
    import gzip
    gz = workdir / "s2i.fa.gz"
    gz.write_bytes(gzip.compress(inp.read_bytes()))
    subprocess.run([exe, "seqToIllumina", "-i", str(gz), "-o", str(workdir / "s2i.fq.gz"), "-s", ppath, "--seed", "77", "--blockKB", "16"], check=True, capture_output=True)
    assert gzip.decompress((workdir / "s2i.fq.gz").read_bytes()).decode() == got
    crlf = workdir / "s2i_crlf.fa"
    crlf.write_bytes(inp.read_bytes().replace(b"\n", b"\r\n"))
    subprocess.run([exe, "seqToIllumina", "-i", str(crlf), "-o", str(out), "-s", ppath, "--seed", "77", "--blockKB", "5"], check=True, capture_output=True)
    assert out.read_text() == got
    # the reference's complaints about malformed headers
    for bad, msg in ((">r 3;40;NNNN;!!!!\nACGT\n", "Template segment is 3"), (">r1;40;NNNN;!!!!\nACGT\n", "No sequence id found"), (">r 1;4x;NNNN;!!!!\nACGT\n", "not a pure integer"),
                     (">r 1;40;NNN;!!!!\nACGT\n", "not separated by a semicolon"), (">r\nACGT\n", "too short"), (">r 1;40;NNNN;!!!!\nACNT\n", "must not contain N"),
                     # a fragment length the profile's tables do not hold: the reference's Vect::at prints this and throws (Vect.hpp:196-221, from ReadLength, Simulator.h:185-198)
                     (">r 1;70000;NNNN;!!!!\nACGT\n", "Called index 70000 range is from")):
        inp.write_text(bad)
        r = subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", ppath, "--seed", "77"], capture_output=True)
        assert r.returncode != 0 and msg.encode() in r.stderr, (bad, r.stderr)
        assert not out.exists()
    # a malformed record far into the file, and text in front of the first record
    inp.write_text("\n".join(fasta) + "\n>r 3;40;NNNN;!!!!\nACGT\n" + "\n".join(fasta[:40]) + "\n")
    r = subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", ppath, "--seed", "77", "--blockKB", "4"], capture_output=True)
    assert r.returncode != 0 and b"Template segment is 3 not 1 or 2: r 3;40;NNNN;!!!!" in r.stderr and not out.exists()
    inp.write_text("ACGT\n" + "\n".join(fasta) + "\n")
    r = subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", ppath, "--seed", "77"], capture_output=True)
    assert r.returncode != 0 and b"without a header" in r.stderr and not out.exists()
    inp.write_text("\n")
    r = subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(out), "-s", ppath, "--seed", "77"], capture_output=True)
    assert r.returncode != 0 and b"does not contain any sequences" in r.stderr and not out.exists()


def test_cli_rejects_bad_numbers_and_contradicting_options(workdir):
    """main.cpp:783-786 (numReads and coverage exclude each other) and values that are not numbers; a gzip reference that ends early"""
    import gzip
    import os
    import subprocess
    from reseq_amd import synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reseq_amd", "reseq")
    ppath, fpath, _ = P.make_inputs(workdir, "cli_bad", synth.TINY, [4100])
    base = [exe, "illuminaPE", "-R", fpath, "-s", ppath, "-1", str(workdir / "b1.fq"), "-2", str(workdir / "b2.fq")]
    for extra, msg in ((["--numReads", "100", "--coverage", "3"], "mutually exclusive"), (["--numReads", "12x"], "is invalid"), (["--coverage", "abc"], "is invalid"),
                       (["--numReads", "100", "--seed", "1e3"], "is invalid")):
        r = subprocess.run(base + extra, capture_output=True)
        assert r.returncode != 0 and msg.encode() in r.stderr, (extra, r.stderr)
    whole = gzip.compress(open(fpath, "rb").read())
    cut = str(workdir / "cut.fa.gz")
    open(cut, "wb").write(whole[:len(whole) // 2])
    r = subprocess.run([exe, "illuminaPE", "-R", cut, "-s", ppath, "-1", str(workdir / "b1.fq"), "-2", str(workdir / "b2.fq"), "--numReads", "100"], capture_output=True)
    assert r.returncode != 0 and b"truncated" in r.stderr, r.stderr


def test_methylation(workdir):
    P.case_methylation(GpuBackend, workdir)


def test_cli_gzip_output_and_input(workdir):
    """-1/-2 ending in .gz are written gzip-compressed and hold the same text; a .gz reference and a .gz sys-error profile are read"""
    import gzip
    import os
    import subprocess
    from reseq_amd import synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reseq_amd", "reseq")
    ppath, fpath, _ = P.make_inputs(workdir, "cli_gz", synth.TINY, [4100, 2999])
    fgz = str(workdir / "cli_gz.fa.gz")
    with open(fpath, "rb") as f, gzip.open(fgz, "wb") as g:
        g.write(f.read())
    prof = str(workdir / "cli_gz_sys.fq.gz")
    a1, a2, b1, b2 = (str(workdir / n) for n in ("g1.fq", "g2.fq", "g1.fq.gz", "g2.fq.gz"))
    subprocess.run([exe, "illuminaPE", "-R", fpath, "-s", ppath, "--numReads", "1200", "--seed", "9", "-1", a1, "-2", a2, "--writeSysError", prof], check=True, capture_output=True)
    subprocess.run([exe, "illuminaPE", "-R", fgz, "-s", ppath, "--numReads", "1200", "--seed", "9", "-1", b1, "-2", b2, "--readSysError", prof], check=True, capture_output=True)
    assert gzip.open(b1, "rb").read() == open(a1, "rb").read() and gzip.open(b2, "rb").read() == open(a2, "rb").read()
    assert open(b1, "rb").read()[:2] == b"\x1f\x8b" and gzip.open(prof, "rb").read().count(b"\n") == 16
    # the same with bzip2: reference and systematic-error profile in, FASTQ out
    import bz2
    fbz, pbz = str(workdir / "cli_gz.fa.bz2"), str(workdir / "cli_gz_sys.fq.bz2")
    open(fbz, "wb").write(bz2.compress(open(fpath, "rb").read()))
    open(pbz, "wb").write(bz2.compress(gzip.open(prof, "rb").read()))
    c1, c2 = str(workdir / "g1.fq.bz2"), str(workdir / "g2.fq.bz2")
    subprocess.run([exe, "illuminaPE", "-R", fbz, "-s", ppath, "--numReads", "1200", "--seed", "9", "-1", c1, "-2", c2, "--readSysError", pbz], check=True, capture_output=True)
    assert bz2.decompress(open(c1, "rb").read()) == open(a1, "rb").read() and bz2.decompress(open(c2, "rb").read()) == open(a2, "rb").read()


def test_simulate_module_equals_cli(workdir):
    """python -m reseq_amd.simulate (the multi-GPU launcher, here with one rank) writes the files of reseq illuminaPE"""
    import os
    import subprocess
    import sys
    from reseq_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reseq_amd", "reseq")
    ppath, fpath, _ = P.make_inputs(workdir, "cli_sim", synth.TINY, [5000, 80, 3210])
    a1, a2, b1, b2 = (str(workdir / n) for n in ("s1.fq", "s2.fq", "t1.fq", "t2.fq"))
    args = ["-R", fpath, "-s", ppath, "--numReads", "30000", "--seed", "13", "--refBias", "no"]
    subprocess.run([exe, "illuminaPE"] + args + ["-1", a1, "-2", a2], check=True, capture_output=True)
    env = dict(os.environ, PYTHONPATH=root)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    subprocess.run([sys.executable, "-m", "reseq_amd.simulate"] + args + ["-1", b1, "-2", b2, "--batchBlocks", "3"], check=True, capture_output=True, env=env, cwd=root)
    assert open(a1, "rb").read() == open(b1, "rb").read() and open(a2, "rb").read() == open(b2, "rb").read()
    assert b":0:Adapter:0:" in open(a1, "rb").read()
    # --gatherOutput (one rank here: the slices go from the kept text through rsq_sim_job_read into tensors and through rsq_dev_pwrite into the files)
    c1, c2 = str(workdir / "u1.fq"), str(workdir / "u2.fq")
    subprocess.run([sys.executable, "-m", "reseq_amd.simulate"] + args + ["-1", c1, "-2", c2, "--batchBlocks", "3", "--gatherOutput", "--gatherSliceMB", "1"], check=True, capture_output=True,
                   env=env, cwd=root)
    assert open(a1, "rb").read() == open(c1, "rb").read() and open(a2, "rb").read() == open(c2, "rb").read()
    # .gz outputs: the rank's share compressed by the library's threads (rsq_sim_job_compress), the adapter-only pairs as a member behind it
    import gzip
    g1, g2 = str(workdir / "v1.fq.gz"), str(workdir / "v2.fq.gz")
    subprocess.run([sys.executable, "-m", "reseq_amd.simulate"] + args + ["-1", g1, "-2", g2, "--batchBlocks", "3"], check=True, capture_output=True, env=env, cwd=root)
    assert gzip.decompress(open(g1, "rb").read()) == open(a1, "rb").read() and gzip.decompress(open(g2, "rb").read()) == open(a2, "rb").read()


def test_seq_to_illumina_in_shares_equals_the_single_run(workdir):
    """seqToIllumina over several GPUs (shards by input record ranges), the ranks one after the other on this GPU: rsq_fasta_count_records per
    stretch of the file, sharding.record_share, rsq_sim_error_model_file with keep_text on the share, rsq_sim_job_write at the offset -- the bytes of
    `reseq seqToIllumina`; the launcher itself (python -m reseq_amd.simulate seqToIllumina) with one rank, over RCCL; a malformed record in a share"""
    import os
    import subprocess
    import sys
    from reseq_amd import api, sharding, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reseq_amd", "reseq")
    ppath, _, _ = P.make_inputs(workdir, "s2i_shares", synth.TINY, [100])
    arrays = synth.make_profile(synth.TINY, seed=5)
    parts = [synth.make_error_model_input(31, 900, 30, arrays, zero_frac=0.6), synth.make_error_model_input(32, 80, 75, arrays, zero_frac=0.5)]
    text = b"".join(P.fasta_of_records(rec, [f"read {k}/{i} x" if i % 5 == 0 else f"r{k}_{i}" for i in range(len(rec["seg"]))], wrap_every=4) for k, rec in enumerate(parts))
    inp, single = workdir / "shares.fa", workdir / "single.fq"
    inp.write_bytes(text)
    subprocess.run([exe, "seqToIllumina", "-i", str(inp), "-o", str(single), "-s", ppath, "--seed", "21"], check=True, capture_output=True)
    want = single.read_bytes()
    assert want.count(b"\n") == 4 * 980
    prof = api.Profile(ppath)
    sim = api.Simulator(prof, None, 0)
    sim.prepare(21)
    try:
        # the whole file through the library call, to a file and kept
        whole = workdir / "whole.fq"
        n, nbytes, trace = sim.error_model_file(inp, whole, block_kb=8, batch_blocks=2, read_threads=3, trace=True)
        assert (n, nbytes) == (980, len(want)) and whole.read_bytes() == want and "device calls" in trace
        for world in (2, 3, 7):
            counts = []
            for r in range(world):
                lo, hi = sharding.record_stretch(len(text), r, world)
                counts.append(api.count_fasta_records(inp, lo, hi))
            out = workdir / f"shares{world}.fq"
            out.write_bytes(b"")
            offset, records = 0, 0
            for r in range(world):
                begin, end, first = sharding.record_share(counts, len(text), r)
                assert first == records
                if end <= begin:
                    continue
                n, nbytes = sim.error_model_file(inp, None, from_=begin, to=end, first_record=first, keep_text=True, block_kb=16)
                sim.job_write(out, offset, None, 0)
                sim.job_free()
                offset += nbytes
                records += n
            assert records == 980 and out.read_bytes() == want, world
        with pytest.raises(Exception, match="no generated text"):
            sim.job_write(workdir / "nothing.fq", 0, None, 0)
        bad = workdir / "bad.fa"
        bad.write_bytes(text + b">r 3;40;NNNN;!!!!\nACGT\n")
        with pytest.raises(Exception, match="Template segment is 3 not 1 or 2"):
            sim.error_model_file(bad, None, keep_text=True)
        with pytest.raises(Exception, match="no generated text"):         # a failed run keeps nothing
            sim.job_write(workdir / "nothing.fq", 0, None, 0)
    finally:
        sim.close()
        prof.close()
    env = dict(os.environ, PYTHONPATH=root, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    launched = workdir / "launched.fq"
    subprocess.run([sys.executable, "-m", "reseq_amd.simulate", "seqToIllumina", "-i", str(inp), "-o", str(launched), "-s", ppath, "--seed", "21"], check=True, capture_output=True,
                   env=env, cwd=root)
    assert launched.read_bytes() == want
    import gzip
    packed = workdir / "launched.fq.gz"
    subprocess.run([sys.executable, "-m", "reseq_amd.simulate", "seqToIllumina", "-i", str(inp), "-o", str(packed), "-s", ppath, "--seed", "21"], check=True, capture_output=True,
                   env=env, cwd=root)
    assert gzip.decompress(packed.read_bytes()) == want


def test_sharded_pre_passes(workdir):
    P.case_sharded_prepare(GpuBackend, workdir)


def test_sharded_pre_passes_with_variants(workdir):
    P.case_sharded_prepare(GpuBackend, workdir, world=4, variants=True)


def test_draws_without_the_bounds_on_the_random_word(workdir, rsq_options):
    """option no_indel_skip: every indel draw of the reads and every error-rate draw of the chains reads its rows (normally the random word alone decides most of
    them, rsq_pack.h certain_column / chain_sure); the results are the same"""
    rsq_options("no_indel_skip", 1)
    P.case_prepass(GpuBackend, workdir)
    P.case_variants_indels(GpuBackend, workdir, density=9, seed=47, tag="nobounds", lengths=(5300, 2600), samples=2)


@pytest.mark.parametrize("chunk,warmup", [(1024, 300), (64, 0)])
def test_sharded_pre_passes_with_other_chunks(workdir, rsq_options, chunk, warmup):
    """shard borders fall inside chunks of any length; the ranks exchange the states at the chunk borders next to them"""
    rsq_options("chain_chunk", chunk)
    rsq_options("chain_warmup", warmup)
    P.case_sharded_prepare(GpuBackend, workdir, world=4, variants=True)


def test_sieve_with_dense_thresholds(workdir):
    P.case_sieve_dense_thresholds(GpuBackend, workdir)


def test_variants_substitutions(workdir):
    P.case_variants_substitutions(GpuBackend, workdir)


def test_variants_insertions_and_deletions(workdir):
    P.case_variants_indels(GpuBackend, workdir)


def test_variants_insertions_and_deletions_dense_four_alleles(workdir):
    P.case_variants_indels(GpuBackend, workdir, density=9, seed=47, tag="indels4", lengths=(5300, 2600), samples=2)


def test_variants_sparse_call_set(workdir):
    """most fragments hold no variant at all (the usual case of a real call set: one variant in several hundred bases); the walk over the variants finds nothing in them"""
    P.case_variants_indels(GpuBackend, workdir, density=150, seed=31, tag="sparse", lengths=(5300, 2600), samples=1)
    P.case_variants_indels(GpuBackend, workdir, density=60, seed=53, tag="sparseb", lengths=(5300, 2600), samples=2)


def test_variants_crowding_the_sequence_ends(workdir):
    """start and end surroundings that wrap around a sequence end while variants sit in them: the wrapped part is the plain reference"""
    P.case_variants_indels(GpuBackend, workdir, density=30, seed=71, tag="ends71", lengths=(3300, 2100), ends=45)
    P.case_variants_indels(GpuBackend, workdir, density=30, seed=78, tag="ends78", lengths=(3300, 2100), ends=45)


def test_variants_systematic_errors_in_strand_windows(workdir, rsq_options):
    """the host pass over the variants' systematic errors cuts long strands into windows that start from the chain's state in front of them
    (8.4 M positions each; here two chunks of 256, so that these short sequences are cut as well)"""
    rsq_options("window_chunks", 2)
    P.case_variants_indels(GpuBackend, workdir, density=9, seed=47, tag="windows", lengths=(5300, 2600), samples=2)


@pytest.mark.parametrize("chunk,warmup", [(64, 0), (64, 17), (1024, 100), (4096, -1), (256, 0)])
def test_chains_in_chunks_of_any_length_with_any_run_up(workdir, rsq_options, chunk, warmup):
    """the systematic-error chains are a fixed point of passes over chunks: the tracks (and the variants' own errors, which start from the chain state in
    front of them) do not depend on the chunk length or on the run-up the first pass guesses a chunk's entering state from"""
    rsq_options("chain_chunk", chunk)
    rsq_options("chain_warmup", warmup)
    P.case_prepass(GpuBackend, workdir)
    P.case_variants_indels(GpuBackend, workdir, density=9, seed=47, tag=f"chunk{chunk}_{warmup}", lengths=(5300, 2600), samples=2)


def test_variants_complex(workdir):
    P.case_variants_complex(GpuBackend, workdir)


def test_variants_walk_off_sequence_is_reported(workdir):
    P.case_variants_walk_off_sequence(GpuBackend, workdir)


def test_variants_with_loaded_sys_errors(workdir):
    P.case_variants_with_loaded_sys_errors(GpuBackend, workdir)


def test_variants_with_methylation_in_regions_far_apart(workdir):
    P.case_variants_methylation_far_regions(GpuBackend, workdir)


def test_variants_with_methylation(workdir):
    P.case_variants_with_methylation(GpuBackend, workdir)


def test_variants_many_alleles(workdir):
    """ten alleles (five samples) and the maximum of 128 (64 samples): ChooseAlleles over up to 256 (allele, strand) slots"""
    P.case_variants_indels(GpuBackend, workdir, density=25, seed=53, tag="alleles10", lengths=(3300, 2100), samples=5)
    P.case_variants_indels(GpuBackend, workdir, density=40, seed=59, tag="alleles128", lengths=(3100,), samples=64)


def test_variants_more_alleles_than_the_reference_supports_are_refused(workdir):
    P.case_variants_rejected(GpuBackend, workdir)


def test_variants_every_staging_mode(workdir, rsq_options):
    """the two instantiations of the read kernel with variants: every table from HBM (mode 0) and every table staged"""
    for mode in (0, -1):
        rsq_options("fill_mode", mode)
        P.case_variants_substitutions(GpuBackend, workdir)
        P.case_variants_indels(GpuBackend, workdir)


def test_cli_with_variants_equals_the_oracle(workdir):
    """reseq illuminaPE -V: the files of the command line are the oracle's text for the same seed (own pre-pass: thresholds from the
    device's tree reduction are given to the oracle), and python -m reseq_amd.simulate -V writes the same files"""
    import ctypes as C
    import os
    import subprocess
    import sys
    import numpy as np
    import oracle_lib as O
    from reseq_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reseq_amd", "reseq")
    lengths = [5000, 80, 3210]
    ppath, fpath, seqs = P.make_inputs(workdir, "cli_var", synth.TINY, lengths)
    vcf = workdir / "cli_var.vcf"
    P.write_vcf(vcf, seqs, P._mixed_variant_set(seqs, np.random.default_rng(3), 50, [0, 999, 1000, 1001]))      # substitutions, insertions, deletions
    a1, a2, b1, b2 = (str(workdir / n) for n in ("v1.fq", "v2.fq", "w1.fq", "w2.fq"))
    args = ["-R", fpath, "-s", ppath, "-V", str(vcf), "--numReads", "20000", "--seed", "13", "--refBias", "no"]
    subprocess.run([exe, "illuminaPE"] + args + ["-1", a1, "-2", a2], check=True, capture_output=True)
    env = dict(os.environ, PYTHONPATH=root)
    subprocess.run([sys.executable, "-m", "reseq_amd.simulate"] + args + ["-1", b1, "-2", b2], check=True, capture_output=True, env=env)
    assert open(a1, "rb").read() == open(b1, "rb").read() and open(a2, "rb").read() == open(b2, "rb").read()
    p = P.Pair(GpuBackend, workdir, "cli_var", synth.TINY, lengths, seed=13, num_pairs=20000, ref_bias_mode=1, vcf=vcf)
    try:
        p.osim.set_normalization(p.info["bias_normalization"], p.b.thresholds())
        ofr = p.osim.sieve_var(1, p.info["total_blocks"] + 1)
        o1, o2 = p.osim.create_reads_var(ofr)
        ao1, ao2 = p.osim.adapter_only()
        assert open(a1, "rb").read() == o1 + ao1 and open(a2, "rb").read() == o2 + ao2
    finally:
        p.close()


# ------------------------------------------------------------------------------------------------------- the reference's own known answers through the C ABI
def test_reference_sequence_known_answers_on_the_device(workdir, tiny_profile_path):
    """ReferenceTest.cpp:277-326 through rsq_sim_reference_sequence: the templates the KERNELS compute (allele_template on the allele's coordinate map, the function
    k_variant_templates runs per mate) for the 17 calls of the reference's test -- both alleles, forward and reversed, starts inside inserted bases, fragments shorter
    than the variant they start in -- from a VCF that encodes the test's hand-built variants; and the two calls without variants"""
    import json
    import os
    from conftest import GOLDEN
    from reseq_amd import api
    from test_abi import write_known_answer_vcf
    ka = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))["reference_sequence_with_variants"]
    prof = api.Profile(tiny_profile_path)
    fasta = os.path.join(GOLDEN, "reference-test.fa")
    # without variants
    ref = api.Reference(fasta, 0)
    sim = api.Simulator(prof, ref, 0)
    for start, length, reversed_, want in ka["plain"]:
        assert sim.reference_sequence(ka["seq"], start, length, reversed_) == want
    with pytest.raises(api.RsqError):
        sim.reference_sequence(ka["seq"], 495, 10)                     # leaves the sequence
    sim.close()
    ref.close()
    # with the variants of the test; the fourth variant (position 499) lies behind everything the first ten calls read, so one simulator serves all 17
    vcf = workdir / "reference_sequence_known_answers.vcf"
    write_known_answer_vcf(vcf, ka)
    ref = api.Reference(fasta, 0)
    assert ref.read_variants(vcf) == 2
    sim = api.Simulator(prof, ref, 0)
    n = 0
    for start, length, reversed_, vid, vpos, allele, want in ka["calls"] + ka["calls_with_added_variant"]:
        assert sim.reference_sequence(ka["seq"], start, length, reversed_, (vid, vpos), allele) == want, (start, length, reversed_, vid, vpos, allele)
        n += 1
    assert n == 17
    assert sim.reference_sequence(1, 0, 12, False, (0, 0), 1) == "GATTGCGCTGGC"        # the sequence without variants, on an allele
    with pytest.raises(api.RsqError):
        sim.reference_sequence(0, 0, 10, False, (0, 0), 2)             # no such allele
    with pytest.raises(api.RsqError):
        sim.reference_sequence(0, 4, 3, False, (1, 4), 1)              # variant 1 has three bases
    sim.close()
    ref.close()
    # a substitution-only set is packed as a copy of the reference per allele (rsq_pack.h variants_mode_for): the other route of the same entry point
    sub = dict(ka, vcf_records=[r for r in ka["vcf_records"] if len(r[2]) == len(r[3]) == 1])
    write_known_answer_vcf(workdir / "reference_sequence_substitution.vcf", sub)
    ref = api.Reference(fasta, 0)
    ref.read_variants(workdir / "reference_sequence_substitution.vcf")
    sim = api.Simulator(prof, ref, 0)
    assert sim.reference_sequence(0, 0, 12, False, (0, 0), 0) == "AGCTTTTCACTC" and sim.reference_sequence(0, 0, 12, False, (0, 0), 1) == "AGCTTTTCATTC"
    assert sim.reference_sequence(0, 12, 12, True, (0, 0), 0) == "GAGTGAAAAGCT"
    sim.close()
    ref.close()
    prof.close()


def test_update_ref_seq_bias_known_answers(workdir):
    """FragmentDistributionStatsTest.cpp:1020-1048 through rsq_sim_set_ref_bias_file / rsq_sim_prepare / rsq_sim_get_ref_seq_bias, on the reference's own
    reference-test.fa and ref-bias-test.txt (byte-identical copies): keep falls back to ones when the stored vector has another size, keep keeps, no gives ones,
    the file gives {2.0, 1.0}"""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    from reseq_amd import api, synth
    g = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))["update_ref_seq_bias"]
    ref = api.Reference(os.path.join(GOLDEN, "reference-test.fa"), 0)
    for k, (mode, stored, want) in enumerate(g["cases"]):
        arrays = synth.make_profile(synth.TINY, seed=5, n_ref_seqs=len(stored))
        arrays["frag.ref_seq_bias"] = np.array(stored, np.float64)
        path = workdir / f"update_ref_seq_bias_{k}.rsqp"
        synth.write_profile(path, arrays)
        prof = api.Profile(path)
        sim = api.Simulator(prof, ref, 0)
        if mode == "file":
            sim.set_ref_bias_file(os.path.join(GOLDEN, g["file"]))
        sim.prepare(3, 100, 0.0, {"keep": 0, "no": 1, "draw": 2, "file": 3}[mode])
        assert sim.ref_seq_bias(2).tolist() == want, (mode, stored)
        sim.close()
        prof.close()
    ref.close()
