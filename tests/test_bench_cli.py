"""bench.py launch contract on CPU: --gpus must agree with a launcher's WORLD_SIZE, and without a launcher --gpus N spawns N ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_must_agree_with_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def test_gpus_spawns_one_process_per_rank(tmp_path):
    """No launcher: `bench.py --gpus 2` starts two children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (here they fail at the
    first device call -- there is no GPU -- and the parent reports the failure instead of printing a one-GPU line)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "local_ba"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0 and "rank exit codes" in r.stderr and '"n_gpus": 1' not in r.stdout
