#!/bin/bash
# k_tailf on the HDR class with its stages switched off in turn (tools/tailf_probe.py)
python3 tools/tailf_probe.py 2>&1 | tail -6
