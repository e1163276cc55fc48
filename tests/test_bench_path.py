"""bench.py's own code path — self-spawn under torch.distributed.run, weight
broadcast, weak-scaling headline region, BASELINE config 3 (LPT shards + ordered
gather) — at world sizes 2 AND 8 (the size the driver's scaling run ends at) over gloo on the CPU emulator build with
shrunk hyper-parameters.  The same script, flags aside, is what the driver runs on GPUs."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def _run(emu_library, gpus, utterances=7, extra=()):
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--ids", "12",
           "--concurrency", "2", "--repeats", "2", "--config3-utterances", str(utterances), "--device", "cpu", "--library", str(emu_library),
           "--tiny", "--no-cpu-baseline", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=str(REPO))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_line_single_gpu_with_the_half_mode_legs(emu_library_path):
    """The path the driver runs first (N = 1: no process group, no all_reduce), with the secondary legs the emulator runs skip by
    default: `half_mode` (the native fp16 vocoder) and `bf16x3_mode`; and the shape of `roofline.by_kernel` (per kernel NAME and
    launch sub-key: launches, average raw-event microseconds) that lets a driver record be compared kernel by kernel."""
    out = _run(emu_library_path, 1, 5, extra=("--tiny-half",))
    assert out["n_gpus"] == 1 and out["process_group"] is None and out["config"]["parallelism"] == "utterance-dp1"
    assert out["value"] > 0 and abs(out["value"] - 3 / (out["ms_per_step"] * 3 / 1e3)) < 1e-6 * out["value"]
    bk = out["roofline"]["by_kernel"]
    assert set(bk) >= {"conv_mfma.hifigan_resblock", "conv_mfma.hifigan_upsample", "glow_top"}
    for cls, rows in bk.items():
        for name, r in rows.items():
            assert "/" in name and r["launches"] > 0 and r["avg_us"] > 0 and r["total_ms"] > 0, (cls, name, r)
    hm = out["half_mode"]
    assert hm["utterances_per_sec"] > 0 and hm["ms_per_step"] > 0 and hm["latency_ms_single_stream"] > 0
    assert hm["dtype"].startswith("f16") and hm["roofline"]["bound"] == "mfma" and hm["roofline"]["launches"] > 0
    assert any(k.startswith("conv_f16") or k.startswith("pair_f16") for k in hm["roofline"]["by_kernel"]["conv_mfma.hifigan_resblock"])
    assert out["bf16x3_mode"]["utterances_per_sec"] > 0


@pytest.mark.parametrize("gpus,utterances", [(2, 7), (8, 19)])
def test_bench_line_at_world_size(emu_library_path, gpus, utterances):
    out = _run(emu_library_path, gpus, utterances)
    assert out["n_gpus"] == gpus and out["steps"] == 3 and out["warmup"] == 1
    assert out["metric"] == "utterances_per_sec" and out["unit"] == "utterances/s" and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - gpus * 3 / (out["ms_per_step"] * 3 / 1e3)) < 1e-6 * out["value"]
    assert out["config"]["parallelism"] == f"utterance-dp{gpus}" and out["config"]["batch"] == 1
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["launches"] > 0 and "grouped launch" in rf["schedule"]
    c3 = out["config3"]
    assert c3["utterances"] == utterances and c3["scaling"] == "strong" and sum(c3["shard_sizes"]) == utterances
    assert len(c3["shard_sizes"]) == gpus and min(c3["shard_sizes"]) >= 1
    assert len(out["per_rank"]["utterances_per_sec"]) == gpus
    assert c3["utterances_per_sec"] > 0 and c3["audio_seconds"] > 0
    assert out["denoiser_on"] is None  # the emulator's tiny vocoder (hop 8) is below the denoiser's STFT size
