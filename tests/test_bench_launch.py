"""bench.py's launch path: `python bench.py --gpus N` must start its own N ranks when no launcher is around it
(the driver's N = 1 and N > 1 commands differ only in the number), deliver exactly ONE JSON line on stdout, and
every start-up / collective stage is named and bounded (exit code 3 with the stage's name, never a silent hang)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    return env


def test_bench_gpus_2_starts_its_own_ranks_and_prints_one_json_line():
    """No GPU touched: --launch-check brings the two ranks and their gloo group up and stops."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # the contract: one JSON line on stdout, whatever gloo / RCCL print
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1]
    assert "starting 2 ranks" in r.stderr
    for rank in (0, 1):  # every rank names its stages on stderr
        assert f"[rank {rank}] stage: init_process_group(gloo, world 2)" in r.stderr


def test_bench_under_a_launcher_does_not_launch_again():
    """With WORLD_SIZE in the environment (the driver's torch.distributed.run command line) bench.py is a rank."""
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting" not in r.stderr
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_stage_watchdog_ends_a_stalled_stage_with_its_name(tmp_path):
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from raytracingmin_amd.distributed import StageWatchdog
        dog = StageWatchdog(limit_s=30, rank=5, stage_file={str(tmp_path / 'stage.txt')!r})
        dog.enter("quick stage")
        dog.enter("a collective that never returns", limit_s=1)
        time.sleep(60)
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    assert "[rank 5] stage: quick stage" in r.stderr
    assert "[rank 5] WATCHDOG: stage 'a collective that never returns' has not finished after 1 s" in r.stderr
    assert (tmp_path / "stage.txt").read_text() == "WATCHDOG: a collective that never returns"


def test_stage_watchdog_done_disarms(tmp_path):
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from raytracingmin_amd.distributed import StageWatchdog
        dog = StageWatchdog(limit_s=1, quiet=True)
        dog.enter("short")
        dog.done()
        time.sleep(2.5)
        print("alive")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "alive" and r.stderr == ""


@pytest.mark.gpu
def test_bench_gpus_2_self_launch_renders_on_the_gpu():
    """The real thing on the one GPU of this box: two ranks share cuda:0 (gloo gather), started by bench.py itself."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "2",
                        "--warmup", "1", "--width", "256", "--height", "144", "--samples", "4", "--super-samples", "2",
                        "--no-extras", "--cpu-rows", "0"], capture_output=True, text=True, timeout=600, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0
    assert j["config"]["collective"] == "gloo gather, 2 rank(s)"
    assert "timed step(s)" in r.stderr
    # the N-rank line proves itself: the gathered frame equals rank 0's own single-launch render bit for bit, every rank's
    # kernel time / rows / counters are in the line, the gather alone is timed
    assert j["frame_matches_single_gpu"] is True and j["gather_ms"] > 0.0
    per_rank = j["config"]["per_rank"]
    assert [p["rank"] for p in per_rank] == [0, 1] and sum(p["rows"] for p in per_rank) == 144
    assert all(p["kernel_ms"] > 0 and p["casts"] >= p["samples"] > 0 for p in per_rank)
    assert sum(p["samples"] for p in per_rank) == 256 * 144 * 16
