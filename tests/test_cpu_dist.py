"""The N>1 path on CPU: two processes over gloo run the same shard -> forward -> all-gather
plumbing bench.py uses on RCCL (vit.cpp_amd/dist.py).  The forward itself is GPU-only, so the
per-rank forward here is a deterministic stand-in (a function of the image content) -- what
is under test is the sharding arithmetic and that the gathered block is in global image order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_forward(block):
    """Stand-in for the HIP engine: per-image 'probabilities' that identify the image."""
    import torch
    m = block.reshape(block.shape[0], -1).mean(1, keepdim=True)
    logits = m * torch.arange(1, 8, dtype=torch.float32)[None, :]
    return torch.softmax(logits, 1)


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import _pkg
    pkg = _pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    images = torch.rand((n_total, 8, 8, 3), generator=g)
    got = pkg.dist.predict_sharded(_fake_forward, images)
    want = _fake_forward(images)
    ok = torch.allclose(got, want) and got.shape == want.shape
    lo, hi = pkg.dist.shard_bounds(n_total, world, rank)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([int(ok), lo, hi]))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 2])
def test_two_rank_shard_and_gather(tmp_path, n_total):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, str(tmp_path))) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    for p in procs:
        assert p.exitcode == 0
    res = [np.load(tmp_path / f"r{r}.npy") for r in range(2)]
    assert all(r[0] == 1 for r in res)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_total      # contiguous cover


def test_shard_bounds_cover_and_balance(pkg):
    for n in (0, 1, 5, 256, 2048, 2049):
        for w in (1, 2, 3, 8):
            b = [pkg.dist.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.dist.shard_bounds(4, 2, 2)


@pytest.mark.parametrize("world", [8, 2])
def test_bench_plumbing_under_torchrun(world):
    """bench.py exactly as the driver launches it for the scaling run -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with the engine stubbed (--stub-engine: gloo, a
    stand-in forward that encodes the rank), so that a rank / port / barrier / gather-order bug cannot be the first thing an 8-GPU node
    finds.  Checked: one JSON line from rank 0 only, whole-job aggregate value over all N ranks, every rank's block of the gathered
    tensor is THAT rank's (asserted inside bench.py), and the line is stamped invalid (it is not a measurement)."""
    import json
    import subprocess
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if not k.startswith("VITX_")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--stub-engine", "--model", "vit_tiny_patch16_224", "--batch", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last == lines[0] and len(last.encode()) < 8192            # the driver parses the LAST stdout line: compact, one object
    d = json.loads(last)
    ranks = d["config"]["ranks"]                                     # a scaling line says which ranks the process group saw
    assert ranks["world_size"] == world and ranks["ranks_seen"] == list(range(world)) and ranks["backend"] == "gloo"
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 4 * world
    assert abs(d["value"] - 4 * world * 3 / (d["ms_per_step"] * 3e-3)) <= 0.02 * d["value"]          # whole-job aggregate, not per rank
    assert "stub" in d["invalid"]


def test_bench_refuses_development_overrides():
    """A VITX_* variable (VITX_LIB would load another library, the laboratory build honours VITX_SKIP ...) makes bench.py refuse to
    measure -- before it touches torch or the GPU."""
    import subprocess
    env = dict(os.environ, VITX_SKIP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "development overrides" in (r.stderr + r.stdout)


def test_bench_gate_and_roofline_arithmetic():
    """The pure-Python pieces of bench.py that decide a line's validity and its roofline object (r05): the parity gate taken against the
    same-semantics oracle for a quantised file (the reference's q8_0-activation figure is reported, not gated), and the per-kernel clock
    with the measured bracket cost subtracted from every launch."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rng = np.random.default_rng(0)
    ref = rng.dirichlet(np.ones(10) * 0.3, size=6).astype(np.float32)
    same = ref + 1e-4                      # what the engine's own semantics gives
    got = same + 3e-4
    far = np.roll(ref, 1, axis=1)          # "the reference's block semantics": far away
    par = bench.parity_of(np, got, far, 2e-2, {"row_ids": [0]}, gate=("max_dprob_vs_dequantised_oracle", same, 1e-3))
    assert par["passed"] and par["gated_on"] == "max_dprob_vs_dequantised_oracle" and par["bound"] == 1e-3
    assert abs(par["max_dprob_vs_dequantised_oracle"] - 3e-4) < 1e-6 and par["max_dprob_vs_ref"] > 0.05 and par["row_ids"] == [0]
    par = bench.parity_of(np, got, far, 2e-2, None, gate=("max_dprob_vs_dequantised_oracle", same, 1e-4))
    assert not par["passed"]
    par = bench.parity_of(np, got, same, 1e-3)
    assert par["passed"] and "gated_on" not in par
    prof = [dict(name="gemm_fc2_resid", launches=24, total_ms=3.30, busy_ms=3.30, flops=2.856e12, bytes=8.4e9),
            dict(name="gemm_qkv_bias", launches=24, total_ms=2.2, busy_ms=2.2, flops=2.14e12, bytes=3.7e9),
            dict(name="attention", launches=24, total_ms=1.0, busy_ms=1.0, flops=3.7e11, bytes=3.7e9)]
    roof, table = bench.roofline_of(prof, 1, {"gemm_fc2_resid": 0.417}, "pmc", bracket_us=5.0)
    assert roof["kernel"] == "gemm_fc2_resid" and roof["bracket_us"] == 5.0
    assert abs(roof["avg_launch_ms"] - (3.30 - 24 * 0.005) / 24) < 1e-4 and abs(roof["avg_launch_ms_raw_event_interval"] - 3.30 / 24) < 1e-4
    assert abs(roof["achieved"] - 2.856e12 / ((3.30 - 0.12) * 1e-3) / 1e12) < 0.1 and roof["achieved"] > roof["achieved_raw_event_interval"]
    assert abs(table["attention"]["us_per_launch"] - (1.0 - 0.12) / 24 * 1e3) < 0.01 and "HIP events" in roof["clock"]


def test_bench_last_line_is_compact():
    """r05's final line was 28 KB and the driver recorded `parsed: null`.  compact_line() reduces a FULL record -- here
    the committed r05 one, the very record that broke the parse -- to one json.loads-able line below 6 KiB that still carries the contract's keys,
    `roofline`, `cpu_baseline`, every gate's verdict and one short summary per other configuration; and it can never outgrow the limit
    (optional summaries are dropped first)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "r05", "cls_tail", "boxF_57ab21f", "bench_default.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) <= bench.COMPACT_LIMIT < 8192
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in c and c[k] == (full[k] if k != "config" else c[k]), k
    assert "model" not in c["config"] and c["config"]["workload"].startswith("vit_base_patch16_224")
    r = c["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] == full["roofline"]["traffic"]
    assert c["cpu_baseline"] == full["cpu_baseline"] and c["probe_tflops"] > 1000
    assert c["parity"]["passed"] is True and c["parity"]["bound"] > c["parity"]["max_dprob_vs_ref"] > 0
    assert c["parity_mode"]["parity"]["passed"] is True and c["parity_mode"]["value"] == full["parity_mode"]["value"]
    assert len(c["other_configs"]) == 3 and all("value" in v and "frac" in v and v["parity"]["passed"] for v in c["other_configs"].values())
    # a record with absurdly many configurations still fits: the optional blocks go first
    fat = dict(full, other_configs={f"cfg{i}": v for i in range(6) for v in full["other_configs"].values()})
    assert len(bench.compact_line(fat).encode()) <= 8192
    # the q4_0 deviation from the reference's own semantics beside that semantics' self-noise (item 6)
    v = bench.vs_reference_semantics(6.6e-2, 2.3e-2)
    assert v["ratio"] == 2.87 and v["ratio"] < v["ratio_limit"] == bench.Q_REF_RATIO_LIMIT and v["gated"] is False
