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
    # the headline is priced with the survey's formula for 18 rectangles
    assert bench.FLOP_PER_RAY_CORNELL == 887


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
