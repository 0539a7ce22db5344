#!/usr/bin/env python3
"""bench.py with class-level switches of the product flipped for an A/B on one box:  python tools/bench_with.py ResBlock.H16_MID=0 -- --steps 4 --warmup 2
(each `Class.ATTR=value` names an attribute of a class in videovanish_amd.nn / videovanish_amd.unet; the rest goes to bench.main())."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
from videovanish_amd import hip as _hip, nn as vnn, unet as vunet      # noqa: E402
if os.environ.get("VV_LIB_PATH"):
    _hip._LIB_PATH = os.environ["VV_LIB_PATH"]      # (lab) another build of the library
for item in args[:split]:
    path, val = item.split("=")
    cls, attr = path.split(".")
    obj = getattr(vnn, cls, None) or getattr(vunet, cls)
    cur = getattr(obj, attr)
    setattr(obj, attr, type(cur)(int(val)) if isinstance(cur, (bool, int)) else type(cur)(val))
sys.argv = [os.path.join(ROOT, "bench.py")] + args[split + 1:]
import bench      # noqa: E402
bench.main()
