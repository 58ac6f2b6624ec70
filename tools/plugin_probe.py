#!/usr/bin/env python3
"""GPU box: the plugin_path leg of bench.py alone (48 tiles of one 12 MP grid through the decoder-plugin ABI from 8
threads).  HM_PLUGIN_DEBUG=1 prints the phases of every batch of the shared device worker; HM_PLUGIN_LINGER_US sets how
long the worker keeps collecting."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package(test_knobs=True)
tiles = [d for d, _ in bench.make_streams(pkg.capi, range(1200000, 1200048))]
print(json.dumps(bench.plugin_path(pkg, tiles)))
