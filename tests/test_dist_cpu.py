"""world_size-2 gloo test (CPU) of the N>1 path of bench.py: rendezvous over 127.0.0.1, sharding of the
independent attention units, barrier-bracketed timing with MAX over ranks, and the timing gather."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, %r)
    from leetcuda_amd import dist as lcd, host
    w = lcd.init("gloo")
    B, H = 4, 6
    b, h, first = host.attn_shard(B, H, w.size, w.rank)
    lcd.barrier(w)
    t0 = time.perf_counter()
    time.sleep(0.05 * (w.rank + 1))            # rank 1 is the slow shard
    lcd.barrier(w)
    dt = time.perf_counter() - t0
    rows = lcd.gather_row(w, [w.rank, b * h, first, dt])
    mx = lcd.max_over_ranks(w, 0.05 * (w.rank + 1))
    if w.rank == 0:
        print("RESULT " + json.dumps({"rows": rows.tolist(), "max": mx, "size": w.size}))
    lcd.shutdown(w)
""") % str(ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0]
    import json
    res = json.loads(line[7:])
    assert res["size"] == 2
    rows = res["rows"]
    assert [r[0] for r in rows] == [0.0, 1.0]
    assert sum(r[1] for r in rows) == 24 and rows[0][2] == 0 and rows[1][2] == 12   # units partitioned
    assert abs(res["max"] - 0.10) < 1e-9                                             # MAX over ranks
    assert min(r[3] for r in rows) >= 0.095        # barrier-bracketed: both ranks waited for the slow one


def _spawned_worker(outdir):
    """module-level (picklable) body of one self-spawned rank: the N > 1 plumbing of bench.py on gloo / CPU."""
    import json
    from leetcuda_amd import dist as lcd, host
    w = lcd.init("gloo")
    b, h, first = host.attn_shard(32, 32, w.size, w.rank)       # config 4 batch shard
    lcd.barrier(w)
    mx = lcd.max_over_ranks(w, 1.0 + w.rank)
    rows = lcd.gather_row(w, [w.rank, b, h, first])
    if w.rank == 0:
        Path(outdir, "out.json").write_text(json.dumps({"size": w.size, "max": mx, "rows": rows.tolist()}))
    lcd.shutdown(w)


def test_self_spawn_two_ranks_gloo(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without torch.distributed.run self-spawns its ranks through dist.spawn()
    (torch.multiprocessing.spawn, rendezvous on 127.0.0.1): same partition and MAX-over-ranks as the torchrun path."""
    import json
    from leetcuda_amd import dist as lcd
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    lcd.spawn(_spawned_worker, 2, (str(tmp_path),))
    res = json.loads((tmp_path / "out.json").read_text())
    assert res["size"] == 2 and res["max"] == 2.0
    assert res["rows"] == [[0.0, 16.0, 32.0, 0.0], [1.0, 16.0, 32.0, 512.0]]


def test_forced_single_rank_group_gloo(tmp_path):
    """LC_DIST_FORCE=1: a ONE-rank process group is created so that the collective branch (barrier, all_gather_into_tensor)
    executes without a second rank — on the GPU box the same switch drives the RCCL branch (tests/test_gpu_dist.py)."""
    import json
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import json, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from leetcuda_amd import dist as lcd
        w = lcd.init("gloo")
        assert w.group and w.is_dist and w.size == 1 and dist.is_initialized()
        lcd.barrier(w)
        print("RESULT " + json.dumps({"rows": lcd.gather_row(w, [1.5, 2.5]).tolist(), "max": lcd.max_over_ranks(w, 4.0)}))
        lcd.shutdown(w)
    """) % str(ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LC_DIST_INIT")}
    env["LC_DIST_FORCE"] = "1"
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res == {"rows": [[1.5, 2.5]], "max": 4.0}
