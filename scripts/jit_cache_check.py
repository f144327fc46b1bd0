#!/usr/bin/env python
"""Does a box find the code objects build() put into the in-tree cache?  DSH_JIT_DEBUG=1 python scripts/jit_cache_check.py  (prints hit / MISS per request and the headers fingerprint)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsol_amd
from diffsol_amd import _ffi
dev = _ffi.load_device_lib()
twin = dev.dsh_model_lane_twin(diffsol_amd.MODELS["spm"], 20)
print("twin", twin, "precompile rc", dev.dsh_model_precompile(twin, 2), "compiled in this process:", dev.dsh_jit_compile_count())
