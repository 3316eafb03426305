#!/usr/bin/env python3
"""Runs another tool of this directory with options of the library set first (rsq_set_option; for the product library and the host emulation):
    python tools/with_options.py chain_chunk=64,chain_warmup=23 tools/stress_variants.py 24 gpu tiny"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from backends import set_option_everywhere  # noqa: E402

for item in sys.argv[1].split(","):
    name, value = item.split("=")
    set_option_everywhere(name, int(value))
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
