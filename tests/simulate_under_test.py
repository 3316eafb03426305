"""Test infrastructure: `reseq_amd.simulate.main` started with the test suite's hooks -- what the product's own command line and environment cannot reach.

    python -m torch.distributed.run --nproc-per-node N ... tests/simulate_under_test.py [--emulate] <the launcher's arguments>

--emulate            the host emulation of the kernels (tests/hostemu through tests/emu_ranks.py) stands where the device would be and the ranks' small exchanges run
                     over gloo on the CPU: the launcher's N-rank path in a container without a GPU.  Says so on stderr.
RSQ_FAULT_INJECT     =<step>:<rank>: the named rank dies on the spot (SIGKILL: no exception, no goodbye) when it reaches the named step of run_rank (generate, write)
                     -- with or without --emulate, i.e. also with the real kernels behind the ranks (tests/test_multi_gpu.py, mode `shared`).
"""
import os
import pathlib
import signal
import sys

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))

from reseq_amd import simulate  # noqa: E402


def _die_at(step, rank):
    if os.environ.get("RSQ_FAULT_INJECT") == f"{step}:{rank}":
        os.kill(os.getpid(), signal.SIGKILL)


class RealKernels(simulate.Hooks):
    at_step = staticmethod(_die_at)


class Emulated(simulate.Hooks):
    on_cpu = True
    banner = ">>> EMULATED on the CPU (tests/hostemu): the launcher's test, not the product path and not a measurement"
    at_step = staticmethod(_die_at)

    @staticmethod
    def make_backend(a, seed, packed_from):
        import emu_ranks
        assert not a.ipf and a.ipfPrecision == 5.0, "ReSeq's archives are read by the product library only; the emulation takes RSQP containers"
        edits = dict(error_multiplier=a.errorMutliplier, no_substitutions=a.noSubstitutionErrors, no_indels=a.noInDelErrors)
        return emu_ranks.EmuRankBackend(a.profile, a.ref, seed, a.vcf, a.methylation, a.readSysError, packed_from=packed_from, edits=edits, ref_bias_file=a.refBiasFile)

    @staticmethod
    def records_sim(a, seed):
        import emu_ranks
        return emu_ranks.EmuRecordsSim(a.profile, seed, dict(error_multiplier=a.errorMutliplier, no_substitutions=a.noSubstitutionErrors, no_indels=a.noInDelErrors))


def main():
    argv = sys.argv[1:]
    emulate = "--emulate" in argv
    if emulate:
        argv.remove("--emulate")
        if "--backend" not in argv:
            argv += ["--backend", "gloo"]
    simulate.main(argv, Emulated if emulate else RealKernels)


if __name__ == "__main__":
    main()
