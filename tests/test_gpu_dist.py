"""The RCCL branch of leetcuda_amd/dist.py executed on ONE GPU (round-2 verdict: with WORLD_SIZE <= 1 init() returned
before init_process_group, so the nccl + device_id + all_gather_into_tensor-on-device path had never run anywhere).
LC_DIST_FORCE=1 creates the one-rank process group; the same code then serves the 8-GPU node.
Reference idiom: others/pytorch/distributed/test_dist_all.py:22-37 (init_process_group("nccl") + set_device),
:189-234 (mp.spawn)."""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from leetcuda_amd import dist as lcd
    w = lcd.init()
    assert w.backend == "nccl" and w.group and w.is_dist and w.size == 1, w
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    lcd.barrier(w)
    rows = lcd.gather_row(w, [3.5, 7.0, w.rank, 11.25])           # device tensor through all_gather_into_tensor
    mx = lcd.max_over_ranks(w, 0.125)
    x = torch.arange(8, dtype=torch.float32, device="cuda")
    dist.all_reduce(x)                                             # one more RCCL collective on the device
    torch.cuda.synchronize()
    print("RESULT " + json.dumps({"rows": rows.tolist(), "max": mx, "sum": float(x.sum())}))
    lcd.shutdown(w)
""") % str(ROOT)


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LC_DIST_FORCE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LC_DIST_BACKEND", "LC_DIST_INIT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_rccl_single_rank_group_barrier_and_device_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    p = subprocess.run([sys.executable, str(script)], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["rows"] == [[3.5, 7.0, 0.0, 11.25]] and res["max"] == 0.125 and res["sum"] == 28.0


def test_bench_line_through_the_rccl_branch():
    """bench.py's timed region (barrier + synchronize on both sides, MAX over ranks gathered over RCCL) with the group forced."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--quick"], env=_env(),
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and "backend nccl" in out["config"]["parallelism"]
    assert out["roofline"]["frac"] > 0.2


def test_spawn_refuses_more_nccl_ranks_than_gpus():
    import torch
    from leetcuda_amd import dist as lcd
    for k in ("LC_DIST_BACKEND",):
        os.environ.pop(k, None)
    with pytest.raises(RuntimeError, match="one process per GPU"):
        lcd.spawn(print, torch.cuda.device_count() + 1, ())
