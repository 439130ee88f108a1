"""bench.py's launcher contract (ADVICE r01 / VERDICT r01 next #6): `--gpus N` must never silently time
fewer than N GPUs.  CPU-only checks: no GPU exists here, so both forms must exit non-zero with a message
that says why, before anything touches a device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_without_launcher_refuses_when_devices_are_missing():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this check is for machines with fewer than 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], drop=("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert '"n_gpus"' not in r.stdout            # no figure was reported


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr
    assert '"n_gpus"' not in r.stdout
