"""CLI contract of the drop-in module (reference diffuerase.py:121-155): flags, default output name (:132), the
--prior_video branch (:142, inverted in the reference; a supplied prior is USED here), the size asserts (:145,147) and the
call into run_infill_on_frames with all other defaults (:150).  Runs on CPU: frame I/O and the hot path are stubbed."""
import sys
import types

import numpy as np
import pytest


@pytest.fixture()
def cli(monkeypatch, tmp_path):
    import diffuerase
    calls = {"load": [], "write": [], "run": []}
    videos = {}

    def load(path, start=0, max_frames=-1):
        calls["load"].append((path, start, max_frames))
        return [f.copy() for f in videos[path]], 24.0

    def write(out, frames, fps, H0, W0):
        calls["write"].append((out, len(frames), fps, H0, W0))

    tools = types.ModuleType("tools")
    tools.load_video_frames_from_path, tools.write_video_frames_to_path = load, write
    monkeypatch.setitem(sys.modules, "tools", tools)

    def fake_run(frames, masks, **kw):
        calls["run"].append((len(frames), len(masks), kw))
        return [f.copy() for f in frames]

    monkeypatch.setattr(diffuerase, "run_infill_on_frames", fake_run)
    color = tmp_path / "in.mkv"
    color.write_bytes(b"x")
    videos[str(color)] = [np.zeros((16, 24, 3), np.uint8)] * 3
    videos["mask.mkv"] = [np.zeros((16, 24, 3), np.uint8)] * 3
    videos["prior.mkv"] = [np.ones((16, 24, 3), np.uint8)] * 3
    videos["small.mkv"] = [np.zeros((8, 24, 3), np.uint8)] * 3
    return diffuerase, calls, str(color), monkeypatch


def test_cli_defaults(cli):
    d, calls, color, mp = cli
    mp.setattr(sys, "argv", ["diffuerase.py", "--color_video", color, "--mask_video", "mask.mkv"])
    d.main()
    assert calls["load"] == [(color, 0, -1), ("mask.mkv", 0, -1)]
    assert calls["run"] == [(3, 3, {"propainer_frames": None})]                 # every other argument left at its default
    assert calls["write"] == [(color + "_vanished.mkv", 3, 24.0, 16, 24)]        # reference :132


def test_cli_prior_and_range(cli):
    d, calls, color, mp = cli
    mp.setattr(sys, "argv", ["diffuerase.py", "--color_video", color, "--mask_video", "mask.mkv", "--prior_video", "prior.mkv",
                             "--start_frame", "5", "--max_frames", "3", "--out", "o.mkv"])
    d.main()
    assert calls["load"] == [(color, 5, 3), ("mask.mkv", 5, 3), ("prior.mkv", 5, 3)]
    n, m, kw = calls["run"][0]
    assert (n, m) == (3, 3) and len(kw["propainer_frames"]) == 3 and int(kw["propainer_frames"][0][0, 0, 0]) == 1
    assert calls["write"][0][0] == "o.mkv"


def test_cli_errors(cli):
    d, calls, color, mp = cli
    mp.setattr(sys, "argv", ["diffuerase.py", "--color_video", color, "--mask_video", "small.mkv"])
    with pytest.raises(AssertionError, match="mask and color video"):
        d.main()
    mp.setattr(sys, "argv", ["diffuerase.py", "--color_video", color, "--mask_video", "mask.mkv", "--prior_video", "small.mkv"])
    with pytest.raises(AssertionError, match="prior and color video"):
        d.main()
    mp.setattr(sys, "argv", ["diffuerase.py", "--color_video", "/nonexistent.mkv", "--mask_video", "mask.mkv"])
    with pytest.raises(AssertionError, match="input video missing"):
        d.main()
    mp.setattr(sys, "argv", ["diffuerase.py", "--mask_video", "mask.mkv"])
    with pytest.raises(SystemExit):
        d.main()


def test_configure_reference_defaults_is_one_switch():
    """diffuerase.configure(reference_defaults=True) selects what the reference app computes by default (its third-party pipeline's temporal windows +
    the complete ProPainter prior; the 2-step TCD schedule is this module's default already); explicit fields still win; configure() resets."""
    import diffuerase
    from videovanish_amd.config import RunConfig
    try:
        diffuerase.configure(reference_defaults=True)
        assert diffuerase._run_config.windowing == "reference" and diffuerase._prior_stages == {"flow_completion": True, "generator": True}
        diffuerase.configure(run=RunConfig(steps=7, dtype="bf16"), prior={"generator": False}, reference_defaults=True)
        assert (diffuerase._run_config.windowing, diffuerase._run_config.steps, diffuerase._run_config.dtype) == ("reference", 7, "bf16")
        assert diffuerase._prior_stages == {"flow_completion": True, "generator": False}
    finally:
        diffuerase.configure()
    assert diffuerase._run_config is None and diffuerase._prior_stages == {}
