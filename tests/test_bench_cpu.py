"""Host-side pieces of bench.py that run without a GPU: the synthetic clip, the parity summary read from profiles/, the rocm-smi power trace (with a
stand-in for the tool) and the argument surface the driver's contract names."""
import json
import os
import subprocess
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_synthetic_clip_is_deterministic_and_shaped():
    fr, mk, pr = bench.synth_clip(5, 48, 64, seed=7, t0=3)
    fr2, mk2, pr2 = bench.synth_clip(5, 48, 64, seed=7, t0=3)
    assert fr.shape == (5, 48, 64, 3) and fr.dtype == np.uint8 and mk.shape == (5, 48, 64) and pr.shape == fr.shape
    assert np.array_equal(fr, fr2) and np.array_equal(mk, mk2) and np.array_equal(pr, pr2)
    assert 0 < (mk > 0).mean() < 0.5                                   # a mask that covers part of the frame
    _, mk_all, _ = bench.synth_clip(8, 48, 64, seed=7, t0=0)           # the drifting rectangle follows the ABSOLUTE frame index (a rank's slice starts at t0)
    assert np.array_equal(mk_all[3:8], mk)
    f1, m1, p1 = bench.synth_frame(4, 48, 64)                          # per-frame form (strong-scaling mode): prior = frame with the masked pixels at the frame mean
    assert f1.shape == (48, 64, 3) and (p1[m1 == 0] == f1[m1 == 0]).all() and len(np.unique(p1[m1 > 0].reshape(-1, 3), axis=0)) == 1


def test_parity_summary_reads_the_committed_log():
    p = bench.parity_summary()
    assert p and p["source"].startswith("profiles/r") and "c1_full_width[fp16,precise-decoder,10 steps]" in p["per_pixel_max_abs_vs_fp32_oracle"]
    assert all(0 < v <= 1.0e-3 for v in p["per_pixel_max_abs_vs_fp32_oracle"].values())      # the default plan's asserted bound


def test_power_trace_parses_rocm_smi_and_never_raises(monkeypatch):
    calls = {"n": 0}

    def fake_run(cmd, **kw):
        calls["n"] += 1
        assert cmd[0] == "rocm-smi" and "-d" in cmd
        if "--showmaxpower" in cmd:
            out = "GPU[1]\t\t: Max Graphics Package Power (W): 1400.0\n"
        else:
            out = ("GPU[1]\t\t: sclk clock level: 1: (%dMhz)\nGPU[1]\t\t: Current Socket Graphics Package Power (W): %.1f\n" % (1800 + calls["n"], 1300.0 + calls["n"]))
        return types.SimpleNamespace(stdout=out)
    monkeypatch.setattr(subprocess, "run", fake_run)
    with bench.PowerTrace(1) as pt:
        import time
        t0 = time.time()
        while len(pt.samples) < 5 and time.time() - t0 < 20:
            time.sleep(0.05)
    s = pt.summary()
    assert s["samples"] >= 5 and s["cap_w"] == 1400.0 and 1300 < s["mean_w"] < 1400 and s["share_of_samples_at_or_above_1200_w"] == 1.0 and 1800 < s["mean_sclk_mhz"] < 1900
    json.dumps(s)
    # a machine without the tool / with odd output: no exception, no object
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(FileNotFoundError("rocm-smi")))
    with bench.PowerTrace(0) as pt2:
        pass
    assert pt2.summary() is None
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(stdout="no numbers here"))
    with bench.PowerTrace(0) as pt3:
        import time
        time.sleep(0.4)
    assert pt3.summary() is None


def test_bench_flags_of_the_driver_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120).stdout
    for flag in ("--gpus", "--steps", "--warmup", "--frames", "--denoise-steps", "--no-power-trace", "--lanes", "--prior", "--dilate", "--reference-defaults", "--dry-run"):
        assert flag in out


def test_kernel_table_groups_profile_records():
    """bench.kernel_table: hip.PROFILE records (key, flops, bytes, event pair) -> launches / seconds / flops / bytes per key (the c5 line's `prior` table)."""
    class Ev:
        def __init__(self, t): self.t = t
        def elapsed_time(self, other): return other.t - self.t
    prof = [("prior:corr_lookup", 0.0, 100.0, Ev(0.0), Ev(2.0)), ("prior:corr_lookup", 0.0, 100.0, Ev(5.0), Ev(6.0)), ("conv", 8.0, 0.0, Ev(0.0), Ev(4.0))]
    tab = bench.kernel_table(prof)
    assert tab["prior:corr_lookup"] == [2, 3.0e-3, 0.0, 200.0] and tab["conv"] == [1, 4.0e-3, 8.0, 0.0]
