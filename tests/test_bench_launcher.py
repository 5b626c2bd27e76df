"""`python bench.py --gpus N` starts its own ranks (the driver's plain command line): the launcher of bench.py is run
here with a stub rank (tests/helpers/stub_rank_worker.py: gloo, no GPU) - environment as torch.distributed.run sets
it, only rank 0's stdout is relayed, the worst exit code comes back, a dead rank does not leave the others hanging."""
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = [sys.executable, os.path.join(ROOT, "tests", "helpers", "stub_rank_worker.py")]


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_runs_two_gloo_ranks_and_relays_one_line():
    out = io.StringIO()
    rc = _bench().launch_ranks(2, ["--gpus", "2", "--steps", "6", "--warmup", "2", "--seqs-per-gpu", "2"], worker=STUB,
                               timeout_s=240, out=out)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, out.getvalue()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2
    assert d["gathered_trajectories"] == {"sequences": 4, "ok": True}
    assert abs(d["value"] - 6 * 2 * 2 / 0.02) < 1e-9  # max-over-ranks clock: rank 1's 0.02 s


def test_launcher_runs_eight_gloo_ranks():
    """the shape of the driver's 8-GPU run: eight ranks, three sequences each, sharded s % 8 == r, one gathered line"""
    out = io.StringIO()
    rc = _bench().launch_ranks(8, ["--gpus", "8", "--steps", "4", "--warmup", "1", "--seqs-per-gpu", "3"], worker=STUB,
                               timeout_s=480, out=out)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, out.getvalue()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["gathered_trajectories"] == {"sequences": 24, "ok": True}
    assert abs(d["value"] - 4 * 3 * 8 / 0.08) < 1e-9  # max-over-ranks clock: rank 7's 0.08 s


def test_launcher_reports_a_dead_rank():
    out = io.StringIO()
    rc = _bench().launch_ranks(2, ["--gpus", "2", "--fail-rank", "1"], worker=STUB, timeout_s=240, out=out)
    assert rc == 7 or rc > 0
    assert rc != 0


def test_bench_does_not_touch_torch_or_hip_before_launching():
    """the launcher path must run before anything initialises the GPU: importing bench.py pulls in neither torch nor
    the HIP library, and `--gpus 2` with a stub interpreter reaches launch_ranks"""
    code = ("import sys, importlib.util; spec = importlib.util.spec_from_file_location('b', sys.argv[1]); "
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); "
            "assert 'torch' not in sys.modules and 'ptudes_lab_amd' not in sys.modules; print('clean')")
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr
