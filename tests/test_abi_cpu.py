"""CPU tests of the C-ABI library: it loads and exports every symbol include/vvhip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "videovanish_amd", "csrc", "libvvhip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def _declared():
    src = open(os.path.join(ROOT, "include", "vvhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vv_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vvhip.h but not exported"


def test_binding_lists_match_header(lib):
    from videovanish_amd import hip
    assert sorted(hip.EXPORTS) == _declared()
    assert lib.vv_abi_version() == hip.ABI_VERSION == 10


def test_integration_md_binding_snippet_version_check(lib):
    """The reference-side ctypes snippet of INTEGRATION.md section 2 is executed up to (and including) its ABI assert, against the library as built:
    a stale version number in the document fails here."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", md, flags=re.S) if "vv_abi_version" in b)
    head = block.split("def dilate_masks")[0]
    assert "ctypes.CDLL" in head and "assert lib.vv_abi_version() ==" in head
    cwd = os.getcwd()
    os.chdir(ROOT)      # the snippet loads the library by its path relative to the app directory
    try:
        exec(compile(head, "INTEGRATION.md", "exec"), {})
    finally:
        os.chdir(cwd)
    hdr = open(os.path.join(ROOT, "include", "vvhip.h")).read()
    ver = re.search(r"#define VV_ABI_VERSION (\d+)", hdr).group(1)
    assert "vv_abi_version() == " + ver in head


def test_vvio_header_symbols_are_exported():
    """include/vvio.h (frame codec, libvvio.so): every declared entry point is exported and the ABI versions agree."""
    src = open(os.path.join(ROOT, "include", "vvio.h")).read()
    ver = int(re.search(r"#define VVIO_ABI_VERSION (\d+)", src).group(1))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(vvio_[a-z0-9_]+)\s*\(", src)))
    assert names == ["vvio_abi_version", "vvio_ffv1_config_record", "vvio_ffv1_decode_frame", "vvio_ffv1_decode_frame_yuv", "vvio_ffv1_decoder_close",
                     "vvio_ffv1_decoder_decode", "vvio_ffv1_decoder_info", "vvio_ffv1_decoder_open", "vvio_ffv1_encode_frame", "vvio_ffv1_stream_info",
                     "vvio_ycbcr_to_rgb"]
    io = ctypes.CDLL(os.path.join(ROOT, "videovanish_amd", "csrc", "libvvio.so"))
    for n in names:
        assert hasattr(io, n), f"{n} declared in include/vvio.h but not exported"
    assert io.vvio_abi_version() == ver


def test_argument_validation_without_gpu(lib):
    """Launchers validate arguments before touching the device: bad args -> negative code + message."""
    from videovanish_amd import hip
    lib.vv_last_error.restype = ctypes.c_char_p
    p = hip.ConvParams()
    assert lib.vv_conv_gemm(ctypes.byref(p), 0, None) == -1 and b"null tensor" in lib.vv_last_error()
    a = hip.AttnParams()
    assert lib.vv_attention(ctypes.byref(a), 0, None) == -1
    assert lib.vv_layernorm(None, 1, 4, None, None, None, 1, None, 0, None) == -1
    assert lib.vv_groupnorm_nsplit(14400, 320) >= 1


def test_product_path_has_no_cpu_fallback():
    """hip wrappers refuse CPU tensors; the product package never imports oracle/."""
    import torch
    from videovanish_amd import hip
    with pytest.raises(RuntimeError):
        hip.axpby(torch.zeros(4), torch.zeros(4), 1.0, 1.0)
    pkg = os.path.join(ROOT, "videovanish_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
    txt = open(os.path.join(ROOT, "diffuerase.py")).read() if os.path.isfile(os.path.join(ROOT, "diffuerase.py")) else ""
    assert "oracle" not in txt


def test_chain_stream_layout_is_part_of_the_abi(lib):
    """ABI 10: vv_spatial_chain_c320 refuses a weight stream whose `layout` is not the order its kernel consumes -- the orders have equal slab
    and parameter counts, so only the id can tell a stream packed the pre-round-5 way from the current one.  Checked before anything touches a device."""
    from videovanish_amd import hip, packing
    lib.vv_last_error.restype = ctypes.c_char_p
    assert hip.CHAIN_LAYOUT_IDS[packing.CHAIN_LAYOUT] == 1
    hdr = open(os.path.join(ROOT, "include", "vvhip.h")).read()
    for name, val in (("TOKENS", 0), ("ROWSPLIT", 1), ("COLUMNS", 2)):
        assert re.search(rf"#define VV_CHAIN_LAYOUT_{name}\s+{val}\b", hdr)
    buf = (ctypes.c_char * 64)()
    addr = ctypes.addressof(buf)
    base = dict(o=addr, t_in=addr, x=addr, res1=0, out=addr, out_dtype=hip.F32, stream=addr, params=addr, M=128, C=320, heads=8, text_len=77, n_slabs=462, n_params=5120)
    for stale in (0, 2, 7):
        cp = hip.ChainParams(layout=stale, **base)
        assert lib.vv_spatial_chain_c320(ctypes.byref(cp), hip.F16, None) == -1
        msg = lib.vv_last_error()
        assert b"layout" in msg and str(stale).encode() in msg, msg


def test_product_sources_carry_no_timing_probes():
    """VERDICT r5 hygiene 10: switches that make a kernel compute wrong results (timing probes) live in lab headers / tools/lab/*.patch, never in a
    translation unit of the product library."""
    csrc = os.path.join(ROOT, "videovanish_amd", "csrc")
    product = [f for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".c")) and "_lab" not in f and f != "vv_conv3.hip"]
    assert "vv_chain.hip" in product and "vv_motion.hip" in product
    for f in product:
        txt = open(os.path.join(csrc, f)).read()
        for tok in ("VV_PROBE_", "VV_CHAIN_PROBE_", "VV_MOTION_NO_PIN"):
            assert tok not in txt, (f, tok)
    assert os.path.isfile(os.path.join(ROOT, "tools", "lab", "r5_chain_probes.patch"))


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of the three parameter structs as gcc compiles include/vvhip.h == the ctypes mirrors in hip.py."""
    import subprocess
    from videovanish_amd import hip
    probes = {"vv_conv_params": (hip.ConvParams, ["weight", "bias", "out", "ldo", "act", "split_heads", "split_tokens", "tile_hint", "act_slope", "sc_ox", "gn_partials"]),
              "vv_deform_params": (hip.DeformParams, ["x_dtype", "offset", "flow", "max_residue", "col", "B", "deform_groups", "Wo"]),
              "vv_attn_params": (hip.AttnParams, ["o", "q_rs", "D", "scale", "q_hs", "v_hs", "q_prescaled", "lse", "o_hs"]),
              "vv_groupnorm_params": (hip.GroupNormParams, ["groups", "eps", "gamma", "stats_ws", "out_dtype"]),
              "vv_chain_params": (hip.ChainParams, ["out_dtype", "stream", "M", "text_len", "n_params", "layout", "o_hw"])}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "vvhip.h"', "int main(void) {"]
    for name, (_, fields) in probes.items():
        src.append(f'  printf("{name} %zu", sizeof({name}));')
        for f in fields:
            src.append(f'  printf(" %zu", offsetof({name}, {f}));')
        src.append('  printf("\\n");')
    src += ["  return 0;", "}"]
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out:
        parts = line.split()
        cls, fields = probes[parts[0]]
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f in fields]
        assert [int(x) for x in parts[1:]] == want, (parts[0], parts[1:], want)


def test_tcd_timesteps_match_diffusers_schedule():
    """diffusers TCDScheduler.set_timesteps (original_inference_steps 50): floor(linspace(0, 50, n, endpoint=False)) into [999, 979, ...]."""
    from oracle import model_ref as M
    from videovanish_amd.pipeline import ddim_timesteps, tcd_timesteps
    assert tcd_timesteps(2) == [999, 499] and tcd_timesteps(4) == [999, 759, 499, 259] and tcd_timesteps(3) == [999, 679, 339]
    assert tcd_timesteps(1) == [999] and tcd_timesteps(50) == [20 * k - 1 for k in range(50, 0, -1)]
    for n in (1, 2, 3, 4, 5, 8, 10, 25, 50):
        assert tcd_timesteps(n) == M.tcd_timesteps(n) and ddim_timesteps(n) == M.ddim_timesteps(n)


def test_normalize_device():
    """str / torch.device / int are all accepted and keep their index (one process per GPU passes cuda:<LOCAL_RANK>)."""
    import pytest
    import torch
    from videovanish_amd.nn import normalize_device
    assert normalize_device("cuda:3") == torch.device("cuda", 3)
    assert normalize_device(torch.device("cuda", 5)) == torch.device("cuda", 5)
    assert normalize_device(2) == torch.device("cuda", 2)
    with pytest.raises(RuntimeError):
        normalize_device("cpu")
