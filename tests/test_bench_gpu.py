"""bench.py's output contract on a one-GPU box (-m gpu): the single JSON line the driver parses, for N = 1 and -- over gloo
with every rank on device 0 (MSPLAT_BENCH_ONE_DEVICE=1, a debug aid: it exercises the N > 1 control flow, not xGMI) -- N = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "6", "--warmup", "2", "--prewarm", "24", "--serial-frames", "8", "--profile-frames", "1"]


def _line(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check_contract(d, n):
    assert d["metric"] == "frames_per_sec" and d["unit"] == "frames/s"
    assert d["n_gpus"] == n and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0 and abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-3 * 1e3
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "BASELINE configs[1]" in cfg["workload"] and cfg["splats"] == 1000000 and (cfg["width"], cfg["height"]) == (1920, 1080)
    assert "model" not in cfg
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["avg_launch_ms"] > 0 and (r["traffic"] is None or r["traffic"] > 0)
    assert d["serial"]["ms_per_frame"] > 0
    # no fraction in the line can exceed 1 (r3 printed SURVEY 8d's 52 D formula as a "fraction": 1.9 on the scene-like cloud)
    assert set(r["stages"]) >= {"sort", "project", "binning", "composite"}
    for k in ("sort", "project", "binning", "composite"):
        st = r["stages"][k]
        assert st["bytes"] > 0 and st["us"] > 0 and 0.0 < st["frac"] <= 1.0, (k, st)
    assert 0.0 < d["frame_moved_frac"] <= 1.0 and 0.0 < d["frame_moved_frac_serial"] <= 1.0
    assert "formula_frac" not in r and "frame_hbm_frac" not in d
    # r6: the VALU view reports ACHIEVED flops (14.5 executed per evaluation) beside the nominal 20, and the share of the
    # evaluations whose weight survives the discard; the timed phase is long enough to be seen from outside
    v = r["valu"]
    assert 0.0 < v["frac_of_fp32_vector_peak"] < v["frac_of_fp32_vector_peak_nominal_20_flop"] < 1.0 and v["flop_per_eval_executed"] == 14.5
    assert 0.05 < v["useful_eval_frac"] < 1.0
    # r6: what the box's HBM delivers to a plain copy, measured in the same process (context for the fractions of the nominal peak)
    assert cfg["cu_partition"] in ([1, 2, 1, 2], [0, 0, 0, 0]), cfg["cu_partition"]       # r6: four frames in flight, two per half of the CUs ([0, 0, 0, 0]: not a 256-CU device)
    hd = r["hbm_delivered"]
    assert (hd is None) if d["n_gpus"] > 1 else (hd["copy_GBps"] is not None and 1000.0 < hd["copy_GBps"] < 8000.0), hd
    assert (d["timed_seconds"] >= 1.5 or n > 1) and d["timed_blocks"] >= 2 and d["block_ms"]["min"] <= d["block_ms"]["median"] <= d["block_ms"]["max"]
    tp = cfg["two_pass"]                                   # r4: what the latest two-pass frame did (None: the frames ran in one pass)
    assert tp["mode"] in ("auto", "on", "off") and set(tp) == {"mode", "timed_region", "serial_frames"}


def test_bench_single_gpu_json_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--cpu-frames", "1"] + QUICK,
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    _check_contract(d, 1)
    assert d["rccl_ranks"] == 1 and d["process_group"] is None and "gather_check" not in d
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port-tiled" and cb["value"] > 0 and 1 <= cb["cores"] <= cb["threads"] and cb["unit"] == "frames/s" and cb["sample"]
    lit = d["cpu_baseline_literal"]                        # the one-to-one restatement of the shaders, kept beside it
    assert lit["kind"] == "port" and lit["value"] > 0 and lit["cores"] >= 1
    assert d["value"] > 10 * cb["value"]                   # sanity only: the ratio says nothing about the kernels


def test_bench_two_ranks_one_device_control_flow():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MSPLAT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--no-cpu-baseline", "--also", ""] + QUICK,
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    d = _line(p.stdout)
    _check_contract(d, 2)
    assert d["rccl_ranks"] == 0                            # gloo stand-in: no RCCL ranks are claimed
    pg = d["process_group"]                                # what the live process group reports
    assert pg["backend"] == "gloo" and pg["world_size"] == 2 and len(pg["ranks"]) == 2
    assert sorted(x["rank"] for x in pg["ranks"]) == [0, 1] and pg["distinct_devices"] == 1      # both ranks on device 0 here
    gc = d["gather_check"]                                 # rank 0's gathered frame == the unbanded single-context frame
    assert gc["bit_exact"] is True and gc["values_different"] == 0 and gc["values_compared"] >= 2 * 1920 * 1080 * 4
    assert gc["poses"] == [5, 37] and gc["exchange"] == "gloo"
    per = gc["exchange_ms_per_gather_per_rank"]            # r5: what the row exchange costs each rank (events on the gathering stream)
    assert isinstance(per, list) and len(per) == 2 and all(x is not None and x >= 0.0 for x in per), per
    assert "gloo" in gc["exchange_call"] and "c_abi_exchange" not in gc
    assert "over 2 ranks" in d["config"]["sharding"] and "layout" in d["config"]["sharding"]
    assert d["gather"]["bytes_into_rank0_per_frame"] > 0
    # r6: --layout auto calibrates msplat_band_root_weight's cost model on rank 0 and says what it found; the timed frames' exchange
    # is named (torch's gather here: the C-ABI exchange needs RCCL, i.e. one device per rank)
    lm = d["config"]["layout_model"]
    assert lm["root_weight_percent"] >= 1 and lm["ms_per_bin_row"] > 0 and lm["row_bytes"] == 32 * 1920 * 16
    assert "torch.distributed" in d["config"]["exchange"]
    # the weighted contiguous bands through the same control flow: rank 0 owns 2.5x the rows of rank 1, gathered bit-exactly
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--no-cpu-baseline", "--also", "", "--layout", "weighted:250"] + QUICK,
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    d = _line(p.stdout)
    assert d["gather_check"]["bit_exact"] is True and "weighted:250" in d["config"]["sharding"]
    assert d["gather"]["bytes_into_rank0_per_frame"] == 10 * 32 * 1920 * 16          # 34 bin rows: 24 for rank 0, 10 for rank 1


def test_bench_peer_store_check_child_mode():
    """what rank 0 of an N > 1 bench runs in a child process: the one-process device group against a single context, bit for bit
    (one device here: the mode, its JSON and the comparison; distinct devices: tests/test_gpu_parity.py, multi-GPU boxes only)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--peer-store-check", "--gpus", "1", "--workload", "tiny"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["error"] is None and d["bit_exact"] is True and d["devices"] == [0] and d["poses"] == [5, 37]
    ex = d["exchanges"]                                    # r5: every exchange form the group has, compared and timed (one device: one entry)
    assert list(ex) == ["peer_store"] and ex["peer_store"]["bit_exact"] is True and ex["peer_store"]["ms_per_frame"] > 0


def test_bench_two_pass_workload_reports_what_the_passes_did():
    """a 6 M-splat workload with two-pass frames forced on: the line says what pass 1 and pass 2 did, and no stage fraction exceeds 1"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cfg3", "--two-pass", "on",
                        "--no-cpu-baseline", "--steps", "6", "--warmup", "2", "--prewarm", "12", "--serial-frames", "16", "--profile-frames", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    tp = d["config"]["two_pass"]
    assert tp["mode"] == "on"
    for part in ("timed_region", "serial_frames"):
        t = tp[part]
        assert t is not None and 0 < t["splats_pass1"] < t["visible"] and t["pairs_pass1"] > 0 and t["bins"] == 60 * 34
        assert 0.0 < t["share_pass1"] <= 0.75
    for k in ("sort", "project", "binning", "composite"):
        st = d["roofline"]["stages"][k]
        assert st["bytes"] > 0 and st["us"] > 0 and 0.0 < st["frac"] <= 1.0, (k, st)
    assert 0.0 < d["frame_moved_frac"] <= 1.0
    tc = d["two_pass_check"]               # r5: two fresh contexts, two passes (the share the timed frames ended on) against one pass
    assert tc["bit_exact"] is True and tc["values_different"] == 0 and tc["values_compared"] >= 2 * 1920 * 1080 * 4
    assert tc["poses"] == [5, 37] and tc["two_pass_frames"] == 2 and 0.0 < tc["share"] <= 0.75


def test_bench_ply_names_the_file_and_the_camera_source(tmp_path):
    """`bench.py --ply` (BASELINE configs[2]: a real Inria scene when one is on the box) -- an Inria-style directory
    (point_cloud/iteration_30000/point_cloud.ply with cameras.json two levels up, found like app.cpp:418-461 finds it): the line's
    config.workload names the file, its SH degree and the camera source; without a cameras.json it says the orbit was used"""
    sys.path.insert(0, ROOT)
    from splatapult_amd import camera, synthetic
    ply = tmp_path / "scene" / "point_cloud" / "iteration_30000" / "point_cloud.ply"
    os.makedirs(ply.parent)
    synthetic.write_ply(str(ply), synthetic.generate(30000, seed=5, pos_sigma=1.5, log_scale_mean=-3.2))
    quick = ["--steps", "6", "--warmup", "2", "--prewarm", "12", "--serial-frames", "8", "--profile-frames", "1", "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tiny", "--ply", str(ply)] + quick,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    w = d["config"]["workload"]
    assert str(ply) in w and "30000 splats, SH3" in w and "no cameras.json found" in w and d["data"] == "file" and d["config"]["splats"] == 30000
    synthetic.write_cameras_json(str(tmp_path / "scene" / "cameras.json"), synthetic.scene_cameras(12), 640, 360, camera.FOVY)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tiny", "--ply", str(ply)] + quick,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    w = d["config"]["workload"]
    assert "cameras.json (12 poses, CamerasConfig::ImportJson)" in w and d["config"]["cameras"] == "cameras.json, 12 poses"
