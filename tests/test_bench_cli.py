"""bench.py's launch modes resolve without a GPU (`--dry-run`): a plain `--gpus N` must select the
product's single-process row-sharded engine, never exit asking for torch.distributed.run."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def dry(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run", *args], capture_output=True, text=True, env=e,
                          timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


def test_plain_multi_gpu_launch_selects_the_product_path():
    for n in (2, 8):
        plan = dry(["--gpus", str(n)])
        assert plan["mode"].startswith("node: one process, mi355rec_create_placed (sharded)") and plan["gpus"] == n
    plan = dry(["--virtual-shards", "8"])
    assert "virtual shards" in plan["mode"]
    plan = dry(["--gpus", "8", "--placement", "replicated"])
    assert plan["mode"].startswith("node: one process, mi355rec_create_placed (replicated)")


def test_single_and_rank_modes_and_labels():
    assert dry([])["mode"].startswith("single")
    assert dry(["--gpus", "4"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})["mode"].startswith("rank")
    # the metric label follows --rows / --topn
    assert dry(["--rows", "1000000", "--topn", "10"])["metric"] == "queries/sec, cosine top-10 over a 1M x 12 fp32 catalogue"
    assert dry([])["metric"] == "queries/sec, cosine top-100 over a 10M x 12 fp32 catalogue"
