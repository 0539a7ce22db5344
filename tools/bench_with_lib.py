#!/usr/bin/env python3
"""Lab: bench.py on another build of libvvhip.so (A/B of two builds on one box):  VV_LIB_PATH=/path/to/lib.so python tools/bench_with_lib.py [bench.py flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videovanish_amd import hip
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]
import bench
bench.main()
