"""bench.py's launch path on CPU: `--gpus N` started bare must start N ranks itself (the driver's multi-GPU command is `bench.py --gpus N` under
torch.distributed.run; a bare `python bench.py --gpus 2` used to run ONE rank and print n_gpus 1), and a launcher's world size that is not --gpus is refused.
The ranks run the host emulation of the kernels over gloo (--emulate: a test switch whose line says so)."""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _bench(*flags, env=None, check=True):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], capture_output=True, text=True, timeout=600, env=e, cwd=str(ROOT))
    if check:
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr                      # ONE JSON line, printed by rank 0
    return json.loads(lines[0])


def test_bare_gpus_2_starts_two_ranks():
    one = _line(_bench("--gpus", "1", "--emulate", "--backend", "gloo", "--steps", "2", "--warmup", "1"))
    two = _line(_bench("--gpus", "2", "--emulate", "--backend", "gloo", "--steps", "2", "--warmup", "1"))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["emulated"] and "EMULATED" in two["metric"]
    assert len(two["ms_per_step_per_rank"]) == 2 and two["ms_per_step"] >= max(two["ms_per_step_per_rank"]) * 0.999
    # one job of two sequences: twice the reference, about twice the pairs, rank 0 holds the first half of the blocks
    assert two["config"]["reference_bp"] == 2 * one["config"]["reference_bp"]
    assert 1.6 < two["config"]["pairs_per_step"] / one["config"]["pairs_per_step"] < 2.4
    assert two["config"]["blocks_of_rank_0"][1] - 1 < two["config"]["total_blocks"]
    assert two["scaling"] == "weak" and two["steps"] == 2 and two["warmup"] == 1


def test_one_rank_with_a_process_group():
    out = _line(_bench("--gpus", "1", "--emulate", "--backend", "gloo", "--dist-single", "--steps", "1", "--warmup", "0"))
    assert out["n_gpus"] == 1 and len(out["ms_per_step_per_rank"]) == 1


def test_world_size_that_is_not_gpus_is_refused():
    r = _bench("--gpus", "2", "--emulate", "--backend", "gloo", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, check=False)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "1 rank" in r.stderr
    r = _bench("--gpus", "2", "--backend", "gloo", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"}, check=False)
    assert r.returncode != 0 and "gloo" in r.stderr
