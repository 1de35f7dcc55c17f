"""wq CLI: BASELINE configs[0]-style plumbing on the host backend (CPU), the rank-sharded path with a
world_size-2 gloo job on CPU, and the hip backend on the GPU."""
import json
import os
import re
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import mtq_oracle as orc
from quantization_analysis_amd import cli, model_source

ROOT = Path(__file__).resolve().parent.parent
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


def write_cfg(tmp_path, algo="mixed-tile-greedy", seed=123, params=None):
    p = tmp_path / "cfg.json"
    cfg = {"algorithm": algo, "quantization_formats": ["bf16", "bfp8", "bfp4", "bfp2", "fp0"],
           "params": params or {"metric": "pcc", "threshold": 0.999}}
    if seed is not None:
        cfg["seed"] = seed
    p.write_text(json.dumps(cfg))
    return str(p)


def run_dir(results_root: Path) -> Path:
    runs = sorted(results_root.glob("*/*/*"))
    assert len(runs) == 1, runs
    return runs[0]


def strip_time(text: str) -> list[str]:
    """Table text without the TIME(s) column values (wall times differ run to run)."""
    out = []
    for line in text.splitlines():
        m = re.match(r"^(  \S+\s+\S+\s+\S+\s+\S+\s+\S+\s+)(\S+)(.*)$", line)
        out.append((m.group(1) + "T" + m.group(3)) if m and not line.strip().startswith("COMP") else line)
    return out


def check_maps_against_oracle(rdir: Path, algo_dir: str, names, index, oracle_fn):
    for name in names:
        x = index.load(name).float().numpy()
        a = np.load(rdir / algo_dir / cli._slug(name) / "assignment.npy")
        want = oracle_fn(x)
        assert a.dtype == np.int8 and np.array_equal(a, want), name
        mapping = json.loads((rdir / algo_dir / cli._slug(name) / "assignment_mapping.json").read_text())
        assert mapping["int_to_format"] == ALL and mapping["assignment_shape"] == list(a.shape) and mapping["tile_hw"] == 32


def test_name_filters_and_sharding():
    idx = model_source.build_model_index("synthetic:tiny")
    names = model_source.resolve_selected_tensors(idx, None)
    assert "model.layers.1.mlp.up.weight_scale_inv" not in names and "lm_head.bias" not in names  # weight-like only (:290-301)
    assert model_source.resolve_selected_tensors(idx, "model.layers.1") == ["model.layers.1.attn.q.weight", "model.layers.1.mlp.up.weight"]
    assert model_source.resolve_selected_tensors(idx, "NORM") == ["model.layers.0.norm.weight"]
    assert model_source.resolve_selected_tensors(idx, "bias") == ["lm_head.bias"]  # falls back to all names
    with pytest.raises(RuntimeError):
        model_source.resolve_selected_tensors(idx, "nothing-matches")
    assert model_source.resolve_format_list(["BFP8", "all"], ["bf16", "bfp8"]) == ["bfp8", "bf16"]
    with pytest.raises(ValueError):
        model_source.resolve_format_list(["mxfp4"], ["bf16", "bfp8"])
    big = model_source.build_model_index("synthetic:llama3-8b")
    names = model_source.resolve_selected_tensors(big, "model.layers")
    assert len(names) == 224
    shards = model_source.lpt_shards(names, big.numel, 8)
    loads = [sum(big.numel(names[i]) for i in s) for s in shards]
    assert sorted(i for s in shards for i in s) == list(range(224)) and max(loads) / min(loads) < 1.01


def test_wq_emulation_single_process(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    rc = cli.run(["synthetic:tiny", "--compression-config", write_cfg(tmp_path), "--backend", "emulation", "--summary",
                  "--results-dir", str(tmp_path / "results")])
    assert rc == 0
    rdir = run_dir(tmp_path / "results")
    used = json.loads((rdir / "compression_config.used.json").read_text())
    assert used["seed"] == 123 and used["seed_source"] == "config" and "seed" not in used["params"]
    idx = model_source.build_model_index("synthetic:tiny")
    names = model_source.resolve_selected_tensors(idx, None)
    check_maps_against_oracle(rdir, "mixed_tile_greedy", names, idx, lambda x: orc.greedy(x, ALL, "pcc", 0.999, 123)[0])
    table = (rdir / "table.txt").read_text()
    assert "Summary (mean across matched tensors)" in table and table.count("mixed-tile-greedy  MIXED") == len(names)
    # PNG artifacts next to every map (wq:282, :412)
    for name in names:
        d = rdir / "mixed_tile_greedy" / cli._slug(name)
        for png in (d / f"{cli._slug(name)}_assignment.png", d / "size_vs_accuracy.png"):
            assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n", png
    # no match → exit code 1 (wq:599-601); unknown source → error
    assert cli.run(["synthetic:tiny", "zzz", "--results-dir", str(tmp_path / "r2")]) == 1


def test_config0_wq_gpt2_c_attn_emulation(tmp_path, monkeypatch):
    """BASELINE.json configs[0] as written: `wq gpt2 h.0.attn.c_attn.weight`, mixed-tile-greedy pcc >= 0.999, --backend emulation on the
    CPU (offline: the synthetic:gpt2 preset holds the tensor's shape and storage type, 768 x 2304 float32).  The map is the oracle's
    for seed 123 and the counts are the pinned 0 / 1461 / 267 / 0."""
    monkeypatch.chdir(tmp_path)
    rc = cli.run(["synthetic:gpt2", "h.0.attn.c_attn.weight", "--compression-config", str(ROOT / "compression_configs" / "greedy_seed123.json"),
                  "--backend", "emulation", "--results-dir", str(tmp_path / "results"), "--no-plots"])
    assert rc == 0
    rdir = run_dir(tmp_path / "results")
    idx = model_source.build_model_index("synthetic:gpt2")
    names = model_source.resolve_selected_tensors(idx, "h.0.attn.c_attn.weight")
    assert names == ["h.0.attn.c_attn.weight"]
    check_maps_against_oracle(rdir, "mixed_tile_greedy", names, idx, lambda x: orc.greedy(x, ALL, "pcc", 0.999, 123)[0])
    a = np.load(rdir / "mixed_tile_greedy" / cli._slug(names[0]) / "assignment.npy")
    assert a.shape == (24, 72) and [int((a == i).sum()) for i in range(4)] == [0, 1461, 267, 0]
    table = (rdir / "table.txt").read_text()
    row = [ln for ln in table.splitlines() if ln.strip().startswith("mixed-tile-greedy")]
    assert len(row) == 1 and row[0].split()[7:11] == ["0", "1461", "267", "0"] and row[0].split()[-1] == "1,764,687"   # BYTES = mixed_tile_total_bytes of the counts


def test_wq_random_search_artifacts(tmp_path, monkeypatch):
    """mixed-tile-random through the CLI: table row, per-tensor CSV of the samples, the selected map and its mapping
    (wq:151-218); the map is the oracle's (i.e. the reference's) map for the same seed."""
    monkeypatch.chdir(tmp_path)
    cfg = write_cfg(tmp_path, algo="mixed-tile-random", seed=5, params={"metric": "pcc", "threshold": 0.97, "iters": 6})
    assert cli.run(["synthetic:tiny", "model.layers.1", "--compression-config", cfg, "--results-dir", str(tmp_path / "results")]) == 0
    rdir = run_dir(tmp_path / "results")
    assert rdir.parent.name == "mixed-tile-random"
    idx = model_source.build_model_index("synthetic:tiny")
    table = (rdir / "table.txt").read_text()
    for name in ("model.layers.1.attn.q.weight", "model.layers.1.mlp.up.weight"):
        x = idx.load(name).float().numpy()
        want, _counts, samples = orc.random_search(x, ALL, "pcc", 0.97, 6, 5)
        slug = cli._slug(name)
        d = rdir / "mixed_tile_random"
        assert np.array_equal(np.load(d / f"{slug}_assignment.npy"), want), name
        assert json.loads((d / f"{slug}_assignment_mapping.json").read_text())["assignment_shape"] == list(want.shape)
        rows = (d / f"{slug}.csv").read_text().splitlines()
        assert rows[0] == "sample_id,bf16_tiles,bfp8_tiles,bfp4_tiles,bfp2_tiles,total_gb,pcc,mae,atol" and len(rows) == 7
        for line, s in zip(rows[1:], samples):
            f = line.split(",")
            assert [int(v) for v in f[:5]] == [s["id"], *[s["counts"][k] for k in ALL]]
            assert abs(float(f[6]) - s["pcc"]) <= 1e-6 and abs(float(f[7]) - s["mae"]) <= 1e-6 and abs(float(f[8]) - s["atol"]) <= 1e-6
        assert (d / f"{slug}.png").read_bytes()[:4] == b"\x89PNG"
    assert table.count("mixed-tile-random  MIXED") == 2 and "BYTES" in table
    # --no-plots: every non-image artifact, no PNG
    assert cli.run(["synthetic:tiny", "model.layers.1", "--compression-config", cfg, "--results-dir", str(tmp_path / "r3"), "--no-plots"]) == 0
    assert not list((tmp_path / "r3").rglob("*.png")) and len(list((tmp_path / "r3").rglob("*.csv"))) == 2


def test_wq_random_seed_is_recorded(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    assert cli.run(["synthetic:tiny", "norm", "--compression-config", write_cfg(tmp_path, seed=0), "--results-dir", str(tmp_path / "results")]) == 0
    used = json.loads((run_dir(tmp_path / "results") / "compression_config.used.json").read_text())
    assert used["seed_source"] == "random" and 0 < used["seed"] < 2**31


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_wq_two_ranks_gloo_matches_single_process(tmp_path):
    cfg = write_cfg(tmp_path, algo="mixed-tile-threshold", seed=None, params={"metric": "pcc", "threshold": 0.99})
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, str(ROOT / "wq"), "synthetic:tiny", "--compression-config", cfg, "--results-dir", str(tmp_path / "r1")],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), str(ROOT / "wq"), "synthetic:tiny", "--compression-config", cfg,
                          "--results-dir", str(tmp_path / "r2")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    d1, d2 = run_dir(tmp_path / "r1"), run_dir(tmp_path / "r2")
    assert strip_time((d1 / "table.txt").read_text()) == strip_time((d2 / "table.txt").read_text())
    f1 = sorted(p.relative_to(d1) for p in d1.rglob("assignment.npy"))
    f2 = sorted(p.relative_to(d2) for p in d2.rglob("assignment.npy"))
    assert f1 == f2 and len(f1) == 5
    for rel in f1:
        assert np.array_equal(np.load(d1 / rel), np.load(d2 / rel)), rel


def test_wq_two_ranks_one_fails_gloo(tmp_path):
    """A failure on one rank must not leave the other in a collective: both exit, the job returns non-zero well inside the
    collective timeout, and the error names the rank.  The fault is injected by this test's launcher script (rank 1's
    cli._evaluate_shard is replaced by one that raises): the package carries no test hook."""
    cfg = write_cfg(tmp_path, algo="mixed-tile-threshold", seed=None, params={"metric": "pcc", "threshold": 0.99})
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    launcher = tmp_path / "wq_faulty.py"
    launcher.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from quantization_analysis_amd import cli\n"
        "real = cli._evaluate_shard\n"
        "def faulty(*a, **k):\n"
        "    rank = int(os.environ.get('RANK', '0'))\n"
        "    if rank == 1:\n"
        "        raise RuntimeError(f'injected fault on rank {rank}')\n"
        "    return real(*a, **k)\n"
        "cli._evaluate_shard = faulty\n"
        "cli.main()\n")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), str(launcher), "synthetic:tiny", "--compression-config", cfg,
                          "--results-dir", str(tmp_path / "r2"), "--no-plots"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert two.returncode != 0
    assert "injected fault on rank 1" in two.stderr and "Timeout" not in two.stderr and "timed out" not in two.stderr.lower()
    assert not list((tmp_path / "r2").rglob("table.txt"))


def test_bench_gather_two_ranks_gloo(tmp_path):
    """bench.py's multi-rank tail (one gather of summary rows + MAX of the timed region) on a world_size-2 gloo job."""
    script = tmp_path / "g.py"
    script.write_text(
        "import os, sys, json\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import torch, torch.distributed as dist\n"
        "import bench\n"
        "dist.init_process_group('gloo')\n"
        "rank, world = dist.get_rank(), dist.get_world_size()\n"
        "rows = torch.full((3, 11), float(rank), dtype=torch.float64); rows[:, 0] = torch.arange(3) + 10 * rank\n"
        "allr, dt = bench.gather_summary(rows, 1.0 + rank, dist, rank, world)\n"
        "assert dt == 2.0\n"
        "if rank == 0:\n"
        "    assert allr.shape == (6, 11) and allr[:, 0].tolist() == [0, 1, 2, 10, 11, 12] and allr[3:, 1].tolist() == [1, 1, 1]\n"
        "    print('GATHER_OK')\n"
        "else:\n"
        "    assert allr is None\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, r.stderr[-2000:]
    import bench

    assert bench.cpu_budget() >= 1 and bench.default_workers() >= 4
    lone, dt = bench.gather_summary(__import__("torch").ones((2, 11), dtype=__import__("torch").float64), 0.5, None, 0, 1)
    assert lone.shape == (2, 11) and dt == 0.5


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (before any GPU call) and rank 0 prints one JSON
    line; --dry-run keeps the job on the CPU (gloo, made-up rows, no measurement)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["max_seconds"] == 2.0 and "x2" in out["config"]["sharding"]
    assert out["dry_run"] is True and out["value"] is None and out["steps"] == 3
    # N = 1 never goes through the launcher
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-run"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1, r.stderr[-2000:]


def test_bench_model_workload_shards_over_two_ranks(tmp_path):
    """`bench.py --workload llama3-8b` (BASELINE configs[3], strong scaling): the 224 tensors are LPT-sharded over the ranks without
    communication, every tensor lands on exactly one rank, the ranks' differently sized sets of summary rows arrive at rank 0 through
    the job's one gather (gloo here, RCCL on the GPUs) — the --dry-run of that mode on a world of 2 and of 3."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for world in (2, 3):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--workload", "llama3-8b", "--dry-run"],
                           cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        assert out["n_gpus"] == world and out["scaling"] == "strong" and out["tensors"] == 224 and out["tensors_seen_once"] is True
        assert sum(out["tensors_per_rank"]) == 224 and out["imbalance"] < 1.05 and out["max_seconds"] == float(world)
    from bench import llama_shard

    index, names, mine, groups = llama_shard(1, 8)
    assert len(names) == 224 and sum(len(v) for v in groups.values()) == len(mine) and set(groups) <= {(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)}


@pytest.mark.gpu
def test_wq_hip_backend(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    rc = cli.run(["synthetic:tiny", "--compression-config", write_cfg(tmp_path), "--backend", "hip", "--results-dir", str(tmp_path / "results")])
    assert rc == 0
    rdir = run_dir(tmp_path / "results")
    idx = model_source.build_model_index("synthetic:tiny")
    names = model_source.resolve_selected_tensors(idx, None)
    check_maps_against_oracle(rdir, "mixed_tile_greedy", names, idx, lambda x: orc.greedy(x, ALL, "pcc", 0.999, 123)[0])
    # hip columns agree with the host backend's literal float32 columns to 1e-6 (small tensors)
    rc = cli.run(["synthetic:tiny", "--compression-config", write_cfg(tmp_path), "--backend", "emulation", "--results-dir", str(tmp_path / "results_emu")])
    assert rc == 0

    def numbers(text):
        return [float(v) for line in text.splitlines() if line.startswith("  none") or line.startswith("  mixed") for v in line.split()[2:5]]

    a = numbers((rdir / "table.txt").read_text())
    b = numbers((run_dir(tmp_path / "results_emu") / "table.txt").read_text())
    assert len(a) == len(b) > 0 and max(abs(x - y) for x, y in zip(a, b)) <= 2e-5  # table prints 5 decimals / 3 significant digits


@pytest.mark.gpu
def test_wq_hip_threshold_and_random(tmp_path, monkeypatch):
    """The other two mixed-tile algorithms through `wq --backend hip` (records stay on the device: K4 / column sums there):
    maps equal the oracle's, artifacts and PNGs are written."""
    monkeypatch.chdir(tmp_path)
    idx = model_source.build_model_index("synthetic:tiny")
    names = model_source.resolve_selected_tensors(idx, None)
    cfg = write_cfg(tmp_path, algo="mixed-tile-threshold", seed=None, params={"metric": "pcc", "threshold": 0.99})
    assert cli.run(["synthetic:tiny", "--compression-config", cfg, "--backend", "hip", "--results-dir", str(tmp_path / "rt")]) == 0
    rdir = run_dir(tmp_path / "rt")
    check_maps_against_oracle(rdir, "mixed_tile_threshold", names, idx, lambda x: orc.threshold(x, ALL, "pcc", 0.99)[0])
    assert len(list(rdir.rglob("size_vs_accuracy.png"))) == len(names)
    cfg = write_cfg(tmp_path, algo="mixed-tile-random", seed=5, params={"metric": "pcc", "threshold": 0.97, "iters": 6})
    assert cli.run(["synthetic:tiny", "model.layers.1", "--compression-config", cfg, "--backend", "hip", "--results-dir", str(tmp_path / "rr")]) == 0
    rdir = run_dir(tmp_path / "rr")
    for name in ("model.layers.1.attn.q.weight", "model.layers.1.mlp.up.weight"):
        want = orc.random_search(idx.load(name).float().numpy(), ALL, "pcc", 0.97, 6, 5)[0]
        assert np.array_equal(np.load(rdir / "mixed_tile_random" / f"{cli._slug(name)}_assignment.npy"), want), name
