"""CPU checks of the kernels' logic: the per-lane __host__ __device__ functions of reseq_amd/csrc (state
machine, draws, packing, Philox counter layout, chunked systematic-error chains, sieve cells, FASTQ formatting)
looped on the host by tests/hostemu and compared with the oracle.  The GPU run of the same cases through the
C ABI is tests/test_parity_gpu.py."""
import numpy as np
import pytest

import oracle_lib as O
import parity_cases as P
from backends import EmuBackend, emu_lib


def test_philox_matches_oracle():
    rng = np.random.default_rng(1)
    for _ in range(200):
        seed = int(rng.integers(0, 2 ** 63))
        c = [int(x) for x in rng.integers(0, 2 ** 32, size=4)]
        out = np.zeros(4, np.uint32)
        emu_lib().emu_philox(seed, *c, out.ctypes.data)
        assert out.tolist() == list(O.lib().orc_philox4x32_10(seed, *c).w)


def test_single_draws_match_oracle(p0_profile_path, tiny_profile_path):
    import ctypes as C
    rng = np.random.default_rng(2)
    for path in (tiny_profile_path, p0_profile_path):
        oprof = O.Profile(path)
        b = EmuBackend(path)
        fams = [("quality", 4, lambda: (int(rng.integers(0, 2)), 0, int(rng.integers(0, 4)), 0)),
                ("seq_quality", 3, lambda: (int(rng.integers(0, 2)), 0, 0, 0)),
                ("base_call", 4, lambda: (int(rng.integers(0, 2)), 0, int(rng.integers(0, 4)), int(rng.integers(0, 5)))),
                ("dom_error", 3, lambda: (int(rng.integers(0, 4)), int(rng.integers(0, 5)), int(rng.integers(0, 5)), 0)),
                ("error_rate", 3, lambda: (int(rng.integers(0, 4)), int(rng.integers(0, 5)), 0, 0)),
                ("indels", 3, lambda: (int(rng.integers(0, 2)), int(rng.integers(0, 6)), 0, 0))]
        nt = 3 if path == tiny_profile_path else 1
        for fam, nm, pick in fams:
            for _ in range(300):
                a, bb, c, d = pick()
                if fam in ("quality", "seq_quality", "base_call"):
                    bb = int(rng.integers(0, nt))
                flat = {"quality": (a * nt + bb) * 4 + c, "seq_quality": a * nt + bb, "base_call": ((a * nt + bb) * 4 + c) * 5 + d,
                        "dom_error": (a * 5 + bb) * 5 + c, "error_rate": a * 5 + bb, "indels": a * 6 + bb}[fam]
                idx = [int(x) for x in rng.integers(0, 200, size=nm)]       # far outside the limits too: clamping
                u = float(rng.random()) if rng.random() < 0.9 else float(rng.choice([0.0, 1.0 - 2 ** -32]))
                ps = C.c_double()
                ii = np.asarray(idx, np.uint32)
                exp = O.lib().orc_draw(oprof.table(fam, a, bb, c, d), O._ptr(ii, O.u32p), u, C.byref(ps))
                got, gps = b.draw(fam, flat, idx, u)
                assert (got, gps) == (exp, ps.value), (fam, a, bb, c, d, idx, u)
        b.close()
        oprof.close()


def test_reference_packing(workdir):
    P.case_reference_packing(EmuBackend, workdir)


def test_prepass(workdir):
    P.case_prepass(EmuBackend, workdir)


def test_sieve_and_reads_tiny(workdir):
    P.case_sieve_and_reads_tiny(EmuBackend, workdir)


def test_sieve_own_thresholds(workdir):
    P.case_sieve_own_thresholds(EmuBackend, workdir)


def test_profile_from_reseq_archive(workdir):
    P.case_profile_from_reseq_archive(EmuBackend, workdir)


def test_coverage_driven(workdir):
    P.case_coverage_driven(EmuBackend, workdir)


@pytest.mark.parametrize("kind", ["subs", "indels", "meth"])
def test_p0_variants(workdir, kind):
    P.case_p0_variants(EmuBackend, workdir, kind)


def test_dense_coverage(workdir):
    P.case_dense_coverage(EmuBackend, workdir)


def test_adapter_only(workdir):
    P.case_adapter_only(EmuBackend, workdir)


def test_more_quality_values_than_the_screen_is_built_for(workdir):
    P.case_more_quality_values_than_the_screen_is_built_for(EmuBackend, workdir)


def test_p0_reads(workdir):
    P.case_p0_reads(EmuBackend, workdir)


def test_every_draw_through_the_route_behind_the_screen(workdir, rsq_options):
    rsq_options("force_exact", 1)
    P.case_p0_reads(EmuBackend, workdir)


@pytest.mark.parametrize("lag", [1, 2, 5])
def test_a_read_that_lags_its_wave_finds_its_quality_rows(workdir, lag):
    """A lane whose read lost steps to deletions is behind its wave's step counter: up to kRingLag steps its position's rows are still in the ring, beyond that the
    wave stages them in the slot behind the ring (fill_wave_reads).  The emulated lane runs `lag` steps behind: same reads as the oracle's, and the screen still decides
    nearly every quality draw (without the slot, every draw of such a read went to the double-precision call)."""
    import ctypes as C
    import numpy as np
    from backends import emu_lib
    stats = np.zeros(8, np.uint64)
    emu_lib().emu_set_ring_lag(lag)
    try:
        emu_lib().emu_screen_stats(C.c_void_p(stats.ctypes.data))             # reset
        P.case_p0_reads(EmuBackend, workdir)
        emu_lib().emu_screen_stats(C.c_void_p(stats.ctypes.data))
    finally:
        emu_lib().emu_set_ring_lag(0)
    draws, left = stats.reshape(4, 2)[0].tolist()
    assert draws > 10_000 and left < 0.01 * draws, (draws, left)


def test_indel_draw_decided_by_the_word_alone_with_no_indel_in_a_middle_column(workdir):
    import ctypes as C
    import numpy as np
    from backends import emu_lib
    stats = np.zeros(8, np.uint64)
    emu_lib().emu_screen_stats(C.c_void_p(stats.ctypes.data))             # reset
    P.case_indel_columns_shuffled(EmuBackend, workdir)
    emu_lib().emu_screen_stats(C.c_void_p(stats.ctypes.data))
    w, w_left = stats.reshape(4, 2)[3].tolist()
    assert w > 200_000 and 1e-3 * w < w_left < 0.05 * w, (w, w_left)        # the frequent insertion and the bound's slack: about 1 %


def test_tables_over_their_own_value_ranges(workdir):
    P.case_ragged_tables(EmuBackend, workdir)


def test_packed_reference_of_another_simulator(workdir):
    P.case_packed_reference(EmuBackend, workdir)


def test_profile_edits(workdir):
    P.case_profile_edits(EmuBackend, workdir)


def test_error_model_tiny(workdir):
    P.case_error_model_tiny(EmuBackend, workdir)


def test_error_model_long_templates(workdir):
    P.case_error_model_long_templates(EmuBackend, workdir)


def test_error_model_p0(workdir):
    P.case_error_model_p0(EmuBackend, workdir)


def test_error_model_p0_at_the_borders_of_the_eight_byte_groups(workdir):
    P.case_error_model_p0_groups(EmuBackend, workdir, n=600)


def test_p0_with_tiles(workdir):
    """--tiles profiles: the LDS plan holds one tile per image (lds_stage_descriptors / ScreenTables with a tile's image_qbase)"""
    P.case_p0_tiles(EmuBackend, workdir, 3, num_pairs=700)


def test_tiles_binned_although_they_fit(workdir, rsq_options):
    rsq_options("image_tiles", 1)
    b = EmuBackend(str(P.make_inputs(workdir, "tiny_e2e", P.synth.TINY, [5000, 80, 3210])[0]))
    assert b.fill_plan() == {"mask": 3, "image_tiles": 1}
    b.close()
    P.case_sieve_and_reads_tiny(EmuBackend, workdir)
    P.case_error_model_tiny(EmuBackend, workdir)


@pytest.mark.parametrize("mode", [0])
def test_double_precision_path(workdir, mode):
    """k_fill_reads<0>: every draw in double precision from HBM, the reference's recipe itself; the default of the other tests is the
    screened path (single-precision draws on the LDS image, double precision only where the screen cannot decide)"""
    class Capped(EmuBackend):
        fill_mode = mode
    P.case_sieve_and_reads_tiny(Capped, workdir)
    P.case_p0_reads(Capped, workdir)


def test_error_rate_rows_fall_back_to_hbm(workdir, rsq_options):
    """only row 0 of the error-rate margins staged: every position with a systematic error rate takes the HBM branch"""
    rsq_options("rate_rows", 1)
    P.case_sieve_and_reads_tiny(EmuBackend, workdir)
    P.case_p0_reads(EmuBackend, workdir)


def test_sys_error_profile_round_trip(workdir):
    P.case_sys_error_profile_round_trip(EmuBackend, workdir)


def test_sys_error_profile_rejects_wrong_reference(workdir):
    P.case_sys_error_profile_rejects_wrong_reference(EmuBackend, workdir)


def test_ref_bias_modes(workdir):
    P.case_ref_bias_modes(EmuBackend, workdir)


def test_methylation(workdir):
    P.case_methylation(EmuBackend, workdir)


def test_sharded_pre_passes(workdir):
    P.case_sharded_prepare(EmuBackend, workdir)


def test_sharded_pre_passes_with_variants(workdir):
    P.case_sharded_prepare(EmuBackend, workdir, world=4, variants=True)


def test_draws_without_the_bounds_on_the_random_word(workdir, rsq_options):
    """option no_indel_skip: every indel draw of the reads and every error-rate draw of the chains reads its rows (normally the random word alone decides most of
    them, rsq_pack.h certain_column / chain_sure); the results are the same"""
    rsq_options("no_indel_skip", 1)
    P.case_prepass(EmuBackend, workdir)
    P.case_variants_indels(EmuBackend, workdir, density=9, seed=47, tag="nobounds", lengths=(5300, 2600), samples=2)


@pytest.mark.parametrize("chunk,warmup", [(1024, 300), (64, 0)])
def test_sharded_pre_passes_with_other_chunks(workdir, rsq_options, chunk, warmup):
    """shard borders fall inside chunks of any length; the ranks exchange the states at the chunk borders next to them"""
    rsq_options("chain_chunk", chunk)
    rsq_options("chain_warmup", warmup)
    P.case_sharded_prepare(EmuBackend, workdir, world=4, variants=True)


def test_sieve_with_dense_thresholds(workdir):
    P.case_sieve_dense_thresholds(EmuBackend, workdir)


def test_variants_substitutions(workdir):
    P.case_variants_substitutions(EmuBackend, workdir)


def test_variants_insertions_and_deletions(workdir):
    P.case_variants_indels(EmuBackend, workdir)


def test_variants_insertions_and_deletions_four_alleles(workdir):
    P.case_variants_indels(EmuBackend, workdir, density=30, seed=47, tag="indels4", lengths=(4300, 2600), samples=2)


def test_variants_sparse_call_set(workdir):
    """most fragments hold no variant at all (the usual case of a real call set: one variant in several hundred bases); the walk over the variants finds nothing in them"""
    P.case_variants_indels(EmuBackend, workdir, density=150, seed=31, tag="sparse", lengths=(5300, 2600), samples=1)


def test_variants_crowding_the_sequence_ends(workdir):
    """start and end surroundings that wrap around a sequence end while variants sit in them: the wrapped part is the plain reference"""
    P.case_variants_indels(EmuBackend, workdir, density=30, seed=71, tag="ends71", lengths=(3300, 2100), ends=45)
    P.case_variants_indels(EmuBackend, workdir, density=30, seed=78, tag="ends78", lengths=(3300, 2100), ends=45)


def test_variants_systematic_errors_in_strand_windows(workdir, rsq_options):
    """the host pass over the variants' systematic errors cuts long strands into windows that start from the chain's state in front of them
    (8.4 M positions each; here two chunks of 256, so that these short sequences are cut as well)"""
    rsq_options("window_chunks", 2)
    P.case_variants_indels(EmuBackend, workdir, density=9, seed=47, tag="windows", lengths=(5300, 2600), samples=2)


@pytest.mark.parametrize("chunk,warmup", [(64, 0), (64, 17), (1024, 100), (4096, -1), (256, 0)])
def test_chains_in_chunks_of_any_length_with_any_run_up(workdir, rsq_options, chunk, warmup):
    """the systematic-error chains are a fixed point of passes over chunks: the tracks (and the variants' own errors, which start from the chain state in
    front of them) do not depend on the chunk length or on the run-up the first pass guesses a chunk's entering state from"""
    rsq_options("chain_chunk", chunk)
    rsq_options("chain_warmup", warmup)
    P.case_prepass(EmuBackend, workdir)
    P.case_variants_indels(EmuBackend, workdir, density=9, seed=47, tag=f"chunk{chunk}_{warmup}", lengths=(5300, 2600), samples=2)


def test_variants_complex(workdir):
    P.case_variants_complex(EmuBackend, workdir)


def test_variants_walk_off_sequence_is_reported(workdir):
    P.case_variants_walk_off_sequence(EmuBackend, workdir)


def test_variants_with_loaded_sys_errors(workdir):
    P.case_variants_with_loaded_sys_errors(EmuBackend, workdir)


def test_variants_with_methylation_in_regions_far_apart(workdir):
    P.case_variants_methylation_far_regions(EmuBackend, workdir)


def test_variants_with_methylation(workdir):
    P.case_variants_with_methylation(EmuBackend, workdir)


def test_variants_many_alleles(workdir):
    """ten alleles (five samples) and the maximum of 128 (64 samples): ChooseAlleles over up to 256 (allele, strand) slots"""
    P.case_variants_indels(EmuBackend, workdir, density=25, seed=53, tag="alleles10", lengths=(3300, 2100), samples=5)
    P.case_variants_indels(EmuBackend, workdir, density=40, seed=59, tag="alleles128", lengths=(3100,), samples=64)


def test_variants_more_alleles_than_the_reference_supports_are_refused(workdir):
    P.case_variants_rejected(EmuBackend, workdir)


def test_the_screen_decides_almost_every_draw(workdir):
    """screened draws (rsq_core.h): single precision decides unless u*S lies within the error bound of a cumulative boundary -- a few
    draws in ten thousand; everything else is repeated in double precision.  The outputs equal the oracle's either way (the cases
    above); this checks that the fast path is the one that runs."""
    import ctypes as C
    import numpy as np
    from backends import emu_lib
    stats = np.zeros(8, np.uint64)
    emu_lib().emu_screen_stats(C.c_void_p(stats.ctypes.data))             # reset
    P.case_p0_reads(EmuBackend, workdir)
    emu_lib().emu_screen_stats(C.c_void_p(stats.ctypes.data))
    (q, q_left), (b, b_left), (i, i_left), (w, w_left) = stats.reshape(4, 2).tolist()
    assert q > 400_000 and b > 400_000 and w > 400_000
    assert 0 < q_left < 1e-3 * q, (q, q_left)                               # K = 40: about 2e-4
    assert b_left < 2e-4 * b and i_left <= 2e-4 * w, (b, b_left, i, i_left)
    # the indel draw: the random word alone says "no indel" (DevTable::sure_range) except for a few draws in a thousand; only those read rows
    assert i == w_left and 0 < w_left < 5e-3 * w, (w, w_left)
