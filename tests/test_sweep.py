"""Threshold sweep against rows produced by the reference's own sweep pieces (tests/golden/f6_sweep.npz:
step, threshold, size_bytes, pcc, mae, atol, per-format tile counts) and its pareto mask."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer
from quantization_analysis_amd.sweep import pareto_mask, sweep_tensor
from tests.inputs import gen

ROOT = Path(__file__).resolve().parent.parent
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]
CASES = (("pcc", "normal_bf16", (256, 256), "pcc", 0.9, 10), ("mae", "heavy_f32", (160, 224), "mae", 0.02, 8),
         ("atol", "heavy_f32", (96, 128), "atol", 0.3, 6))


def _check(golden_dir, quantizer, to_input=lambda x: x):
    d = np.load(golden_dir / "f6_sweep.npz")
    for tag, kind, shape, metric, lowest, steps in CASES:
        x = gen(kind, 61, shape)
        rows, _baselines, thr = sweep_tensor(to_input(x), ALL, metric, lowest, steps, quantizer)
        want = d[f"{tag}_rows"]
        got = np.asarray([[r["step"], r["threshold"], r["size_bytes"], r["pcc"], r["mae"], r["atol"], *[r[f"{f}_tiles"] for f in ALL]] for r in rows])
        assert got.shape == want.shape
        assert np.array_equal(got[:, 1], want[:, 1]), (tag, "thresholds")          # same float32-derived start, same linspace
        assert np.array_equal(got[:, 6:], want[:, 6:]), (tag, "tile counts per step")
        assert np.array_equal(got[:, 2], want[:, 2]), (tag, "size_bytes")
        # float64 moments vs the reference's float32 columns: mae/atol 1e-6; the float32 BLAS pcc of the reference is itself
        # only good to a few 1e-6 on heavy-tailed tensors of this size
        assert np.max(np.abs(got[:, 4:6] - want[:, 4:6])) <= 1e-6, tag
        assert np.max(np.abs(got[:, 3] - want[:, 3])) <= 3e-6, tag
        k = 3 if metric == "pcc" else (4 if metric == "mae" else 5)
        pts = [{"size": r[2], "metric": r[k]} for r in want]
        assert pareto_mask(pts, metric) == list(d[f"{tag}_pareto"])


def test_sweep_emulation_backend(golden_dir):
    _check(golden_dir, Quantizer("emulation"))


def test_sweep_script_cli(tmp_path):
    out = tmp_path / "sweep"
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "sweep_mixed_tile_threshold.py"), "synthetic:tiny", r"layers\.1\..*weight$",
                        "--steps", "6", "--lowest-metric-val", "0.95", "--out-dir", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    csvs = sorted(out.rglob("sweep_results.csv"))
    assert len(csvs) == 2
    lines = csvs[0].read_text().splitlines()
    assert lines[0] == "step,threshold,size_bytes,pcc,mae,atol,bf16_tiles,bfp8_tiles,bfp4_tiles,bfp2_tiles" and len(lines) == 7
    pngs = sorted(p.relative_to(out).as_posix() for p in out.rglob("*.png"))  # per-tensor frontier + the two overlays (:796, :818-835)
    assert len(pngs) == 4 and "weight_overlays.png" in pngs and "layer_overlays.png" in pngs
    assert sum(p.endswith("size_vs_metric.png") for p in pngs) == 2
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "sweep_mixed_tile_threshold.py"), "synthetic:tiny", "layers", "--list-matches"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Matched 5 tensor(s)" in r.stdout


def test_sweep_script_two_ranks_gloo(tmp_path):
    """The sweep under torch.distributed.run with 2 CPU ranks (gloo): tensors are sharded over the ranks, every rank writes
    the CSVs of its tensors, the Pareto frontiers are gathered to rank 0 for the overlay figures; same CSVs as one process."""
    import os
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = str(ROOT / "scripts" / "sweep_mixed_tile_threshold.py")
    common = ["synthetic:tiny", "layers", "--steps", "5", "--lowest-metric-val", "0.95"]
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, script, *common, "--out-dir", str(tmp_path / "one")], capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), script, *common, "--out-dir", str(tmp_path / "two")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    csv1 = sorted(p.relative_to(tmp_path / "one").as_posix() for p in (tmp_path / "one").rglob("sweep_results.csv"))
    csv2 = sorted(p.relative_to(tmp_path / "two").as_posix() for p in (tmp_path / "two").rglob("sweep_results.csv"))
    assert csv1 == csv2 and len(csv1) == 5
    for rel in csv1:
        assert (tmp_path / "one" / rel).read_text() == (tmp_path / "two" / rel).read_text(), rel
    assert (tmp_path / "two" / "weight_overlays.png").exists() and (tmp_path / "two" / "layer_overlays.png").exists()
    assert "[rank 1]" in two.stdout and "[rank 0]" in two.stdout


def test_sweep_script_two_ranks_one_fails_gloo(tmp_path):
    """One rank's tensor fails: it still takes part in the gather / verdict broadcast / barrier, and every rank exits non-zero
    instead of the job hanging in a collective.  The fault is injected by this test's launcher (rank 0's loader raises): the
    script carries no test hook."""
    import os
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = str(ROOT / "scripts" / "sweep_mixed_tile_threshold.py")
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    launcher = tmp_path / "sweep_faulty.py"
    launcher.write_text(
        "import os, runpy, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from quantization_analysis_amd import model_source\n"
        "real = model_source.ModelIndex.load\n"
        "def faulty(self, *a, **k):\n"
        "    rank = int(os.environ.get('RANK', '0'))\n"
        "    if rank == 0:\n"
        "        raise RuntimeError(f'injected fault on rank {rank}')\n"
        "    return real(self, *a, **k)\n"
        "model_source.ModelIndex.load = faulty\n"
        f"sys.argv[0] = {script!r}\n"
        f"runpy.run_path({script!r}, run_name='__main__')\n")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(launcher), "synthetic:tiny", "layers", "--steps", "4", "--lowest-metric-val", "0.95", "--no-plots",
                          "--out-dir", str(tmp_path / "two")], capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode != 0
    assert "injected fault on rank 0" in two.stdout and "[rank 1]" in two.stdout and "Wrote sweep results" not in two.stdout


def test_reconstruct_script(tmp_path):
    from oracle import mtq_oracle as orc
    from quantization_analysis_amd import model_source

    idx = model_source.build_model_index("synthetic:tiny")
    name = "model.layers.1.mlp.up.weight"
    x = idx.load(name).float().numpy()
    a = np.random.default_rng(0).integers(0, 4, size=(5, 7)).astype(np.int8)
    np.save(tmp_path / "a.npy", a)
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "reconstruct_mixed_tile_assignment.py"), "synthetic:tiny", name, str(tmp_path / "a.npy")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    y = np.load(tmp_path / "a_recon.npy")
    assert np.array_equal(y.view(np.uint32), orc.apply_assignment(x, a).view(np.uint32))


@pytest.mark.gpu
def test_sweep_hip_backend(golden_dir):
    import torch

    _check(golden_dir, Quantizer("hip"))
    _check(golden_dir, Quantizer("hip"), to_input=lambda x: torch.from_numpy(x).cuda())


@pytest.mark.gpu
def test_sweep_threaded_literal_chunks_equal_one_chunk(monkeypatch):
    """The literal re-scoring in many small chunks on pool threads (each worker makes the tensor's device current: a new thread starts
    on device 0 with its own current stream) gives the rows of the single-chunk, single-thread run."""
    import torch

    from quantization_analysis_amd import sweep

    x = torch.from_numpy(gen("normal_bf16", 7, (512, 768))).cuda().to(torch.bfloat16)
    q = Quantizer("hip")
    monkeypatch.setattr(sweep, "LITERAL_CHUNK_TILES", 1 << 20)
    want = sweep.sweep_tensor(x, ALL, "pcc", 0.9, 12, q)
    monkeypatch.setattr(sweep, "LITERAL_CHUNK_TILES", 16)
    monkeypatch.setenv("MTQ_SWEEP_THREADS", "6")
    from quantization_analysis_amd.settings import settings
    settings(refresh=True)
    got = sweep.sweep_tensor(x, ALL, "pcc", 0.9, 12, q)
    assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])


@pytest.mark.gpu
def test_sweep_and_reconstruct_scripts_hip(tmp_path):
    """The two scripts with --backend hip produce the same CSV rows / reconstruction as with the host backend (counts,
    thresholds and sizes exactly; float columns to 1e-6)."""
    import csv as _csv

    outs = {}
    for backend in ("emulation", "hip"):
        out = tmp_path / backend
        r = subprocess.run([sys.executable, str(ROOT / "scripts" / "sweep_mixed_tile_threshold.py"), "synthetic:tiny", r"layers\.1\..*weight$",
                            "--steps", "6", "--lowest-metric-val", "0.95", "--backend", backend, "--out-dir", str(out), "--no-plots"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[backend] = {p.parent.name: list(_csv.reader(p.open())) for p in out.rglob("sweep_results.csv")}
    assert outs["emulation"].keys() == outs["hip"].keys() and len(outs["hip"]) == 2
    for k in outs["hip"]:
        a, b = outs["emulation"][k], outs["hip"][k]
        assert a[0] == b[0] and len(a) == len(b) == 7
        for ra, rb in zip(a[1:], b[1:]):
            assert ra[:3] == rb[:3] and ra[6:] == rb[6:]                                    # step, threshold, size_bytes, tile counts
            assert all(abs(float(x) - float(y)) <= 1e-6 for x, y in zip(ra[3:6], rb[3:6]))  # pcc, mae, atol
