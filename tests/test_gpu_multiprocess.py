"""The multi-process path of bench.py with the real renderer: ranks that share the one GPU of the box
(gloo gathers through host memory; with nccl one rank owns one GPU) must assemble the single-process frame."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

COMMON = ["--steps", "3", "--warmup", "1", "--width", "256", "--height", "200", "--spp", "4", "--no-cpu-baseline",
          "--no-side-measurements"]


def _line(cmd, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_two_and_three_ranks_assemble_the_single_process_frame():
    """bands dealt to 2 and to 3 ranks (a height that is no multiple of the band count), several frames per
    call, three calls in flight, the gather and the assembly: the frame of seed 0 has the single process's
    checksum bit for bit, the whole job counted the same rays, and the JSON line keeps its contract."""
    one = _line([sys.executable, "bench.py"] + COMMON, 29541)
    assert one["n_gpus"] == 1 and one["config"]["split"] == "single device"
    for world, port in ((2, 29543), (3, 29547)):
        many = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                      "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(world),
                      "--backend", "gloo"] + COMMON, port)
        assert many["n_gpus"] == world and many["steps"] == 3 and many["scaling"] == "strong"
        assert many["config"]["frame_checksum_seed0"] == one["config"]["frame_checksum_seed0"]
        assert many["config"]["rays_per_step"] == one["config"]["rays_per_step"]
        # every frame of the timed run, those that were the second or third frame of a call included (200 rows over 3
        # ranks is an uneven split: ranks own 72, 64 and 64 rows), equals the single process's frame of that index
        assert many["config"]["frames_per_call"] == world
        assert many["config"]["frame_checksums_timed"] == one["config"]["frame_checksums_timed"]
        assert sorted(one["config"]["frame_checksums_timed"]) == ["0", "1", "2"]
        assert many["metric"] == one["metric"] and many["unit"] == "Mrays/s" and many["value"] > 0


@pytest.mark.timeout(300)
def test_one_rank_through_rccl():
    """the nccl backend with the one rank this box allows: process group on the device, all-gather on the call's
    stream, barrier and reductions on the device; same frame."""
    one = _line([sys.executable, "bench.py"] + COMMON, 29551)
    forced = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                    "--master-addr", "127.0.0.1", "--master-port", "29553", "bench.py", "--gpus", "1", "--force-dist"] + COMMON, 29553)
    assert forced["config"]["split"].startswith("8-row bands")
    assert forced["config"]["frame_checksum_seed0"] == one["config"]["frame_checksum_seed0"]
