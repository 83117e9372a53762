"""bench.py's configuration table and launcher, checked without a GPU."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_configurations_are_renderable_and_priced():
    flops = json.loads((ROOT / "profiles" / "flop_per_ray.json").read_text())
    for name, cfg in bench.CONFIGS.items():
        lanes = cfg["lanes"]
        assert lanes & (lanes - 1) == 0 and 1 <= lanes <= 64, name
        assert cfg["spp"] % lanes == 0, name
        assert cfg["spp"] // lanes >= 8, f"{name}: too few samples per lane for the compacted launch sequence"
        # sample-sharded over 8 GPUs the composition check renders lanes * 8 streams per pixel on one GPU
        assert lanes * 8 <= 64, name
        assert name in flops and flops[name]["flop_per_ray"] > 0, f"{name}: no measured flop-per-ray terms"
        assert set(flops[name]["terms"]) >= {"planes", "spheres", "rectangles", "hit_fraction", "rays_per_sample"}
    # the headline is priced with the terms counted by the oracle; the closed formula is the fallback
    assert bench.FLOP_PER_RAY_CORNELL_UPPER == 887 and abs(flops["cornell"]["flop_per_ray"] - 872.3) < 1


def test_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks of itself under torch.distributed.run on 127.0.0.1."""
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    args = bench.parse_args()
    assert bench.launch_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def _args(**kw):
    class Args:
        width = height = spp = lanes_per_pixel = spp_total = 0
        shard, scaling = "samples", "strong"
    a = Args()
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_multi_gpu_job_is_a_fixed_workload():
    """`bench.py --gpus N` as typed: the Cornell box at 1920x1080 and 1024 samples per pixel IN TOTAL, split into
    sample batches over the ranks (strong scaling), with enough samples per lane left for the compacted launch
    sequence; --scaling weak keeps the configuration's spp on every rank; --shard rows splits the rows of the
    configuration itself."""
    cfg = bench.CONFIGS["cornell"]
    one = bench.plan_job(_args(), cfg, 1, False)
    assert (one["spp_rank"], one["spp_total"], one["lanes"]) == (256, 256, 4)
    for n in (2, 4, 8):
        job = bench.plan_job(_args(), cfg, n, True)
        assert job["scaling"] == "strong" and job["spp_total"] == 1024 and job["spp_rank"] * n == 1024
        assert job["spp_rank"] % job["lanes"] == 0 and job["spp_rank"] // job["lanes"] >= 8
        assert job["lanes"] * n <= 64        # the composition check renders all batches as streams of one render
    weak = bench.plan_job(_args(scaling="weak"), cfg, 8, True)
    assert weak["scaling"] == "weak" and weak["spp_rank"] == 256 and weak["spp_total"] == 2048
    rows = bench.plan_job(_args(shard="rows"), bench.CONFIGS["cornell4k"], 8, True)
    assert rows["scaling"] == "strong" and rows["rows_mode"] and rows["spp_rank"] == rows["spp_total"] == 4096
    # few samples per rank: the lanes per pixel are narrowed so that a lane keeps 8 samples
    narrow = bench.plan_job(_args(spp_total=128), cfg, 8, True)
    assert narrow["spp_rank"] == 16 and narrow["lanes"] == 2
