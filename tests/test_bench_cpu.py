"""Host logic of bench.py that needs no GPU: argument handling, the BASELINE-config presets, the self-launch decision
(`python bench.py --gpus N` starts its own ranks unless a launcher already did) and the machine-filling tile grids."""
import importlib.util
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_default_line_is_the_headline_workload(bench):
    a = bench.parse_args([])
    w = bench.workload_of(a)
    assert a.gpus == 1 and a.scaling == "weak"
    assert w["vit"] == "ViT-L-14" and w["precision"] == "bf16" and w["model_type"] == "Experimental" and w["outlier"] == 30 and w["sim"]


@pytest.mark.parametrize("cfg,vit,prec,extra", [(2, "ViT-B-16", "bf16", {}), (3, "ViT-L-14", "bf16", {"model_type": "GEM", "outlier": 30}),
                                               (4, "ViT-L-14", "bf16", {"upsampler": "jbu_one"}), (5, "ViT-H-14", "fp8", {"ctf": True})])
def test_baseline_config_presets(bench, cfg, vit, prec, extra):
    w = bench.workload_of(bench.parse_args(["--config", str(cfg)]))
    assert w["vit"] == vit and w["precision"] == prec
    for k, v in extra.items():
        assert w[k] == v
    assert os.path.exists(os.path.join(ROOT, "configs", w["names"]))
    assert len(bench.query_idx_of(w["names"])) == {2: 8, 3: 9, 4: 16, 5: 2}[cfg]
    # overrides compose with a preset
    w2 = bench.workload_of(bench.parse_args(["--config", str(cfg), "--precision", "f16x2", "--cross-tile-fusion"]))
    assert w2["precision"] == "f16x2" and w2["ctf"]


def test_self_launch_only_without_a_launcher(bench, monkeypatch):
    calls = []
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, env=None: calls.append((cmd, env)) or type("R", (), {"returncode": 7})())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.self_launch(bench.parse_args(["--gpus", "1"]))                   # one GPU: nothing to start
    assert not calls
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(bench.parse_args(["--gpus", "4", "--steps", "3"]))
    assert e.value.code == 7                                               # exits with the children's return code
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")                                  # under torchrun: this process IS a rank
    bench.self_launch(bench.parse_args(["--gpus", "4"]))
    assert not calls


def test_tile_grids_fill_whole_rounds_of_the_persistent_gemm(bench):
    """rows x cols tiles per rank: the 256-row output tiles of every linear of a block should fill >= 98 % of whole rounds of 256 CUs."""
    shapes = {"ViT-L-14": (1370, 1024), "ViT-B-16": (1025, 768), "ViT-H-14": (1370, 1280)}
    for vit, (rows, cols) in bench.TILE_GRID.items():
        N, D = shapes[vit]
        m_tiles = math.ceil(rows * cols * N / 256)
        for n_out in (D, 3 * D, 4 * D):
            t = m_tiles * math.ceil(n_out / 256)
            eff = t / (math.ceil(t / 256) * 256)
            assert eff > 0.98, (vit, n_out, eff)
