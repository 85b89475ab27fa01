"""bench.py's control path at world_size 2 on CPU: `python bench.py --gpus 2` must start two ranks ITSELF
(torch.distributed.run, gloo here / RCCL on GPUs), refuse any mismatch between --gpus and the ranks that
joined, take the MAX over ranks of the timed region and report the all-reduced return curve.  The engine is
bench.StubEngine (--stub-engine): no GPU, no HIP library -- this file tests the launcher, not the kernels."""
import json
import os
import subprocess
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_BENCH = os.path.join(_ROOT, "bench.py")


def _run(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, _BENCH] + argv, capture_output=True, text=True, env=env, timeout=600, cwd=_ROOT)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, stdout          # exactly ONE line, printed by rank 0 only
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-engine", "--workload", "cfg1_batched", "--seeds-per-gpu", "4"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["comm"] == {"backend": "gloo", "world_size": 2, "rccl_version": None}
    # whole-job aggregate: both ranks' seeds x agents x env steps over the slowest rank's time
    agent_steps = 2 * 4 * 5 * 1000 * 3
    assert abs(out["value"] - agent_steps / (out["ms_per_step"] * 3e-3)) <= 1e-6 * out["value"]
    assert out["ms_per_step"] >= 10.0                               # the stub sleeps 10 ms per block
    # the stub's return of seed s is -s; ranks hold seeds 1000..1003 and 1004..1007: the all-reduced mean is over all 8
    assert abs(out["mean_team_return_last_block"] + 1003.5) < 1e-9
    assert "STUB" in out["data"]


def test_single_rank_stub_no_group():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--stub-engine", "--workload", "cfg1_batched", "--seeds-per-gpu", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["comm"] is None
    assert abs(out["mean_team_return_last_block"] + 1000.5) < 1e-9


def test_world_size_mismatch_is_refused():
    # a launcher started 1 rank but the command line says 2 GPUs (and the other way round): abort, never mislabel
    r = _run(["--gpus", "2", "--stub-engine"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--stub-engine"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_more_gpus_than_visible_is_refused():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
