#!/usr/bin/env python3
"""Lab: run pytest against another build of libvvhip.so (the product never reads the library path from the environment):
    VV_LIB_PATH=videovanish_amd/csrc/ab/libvvhip_x.so python tools/pytest_with_lib.py tests/test_kernels_gpu.py -x -q -k tile_forms"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videovanish_amd import hip
if os.environ.get("VV_LIB_PATH"):
    hip._LIB_PATH = os.environ["VV_LIB_PATH"]
import pytest
sys.exit(pytest.main(sys.argv[1:]))
