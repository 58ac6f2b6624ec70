#!/bin/bash
HM_PLUGIN_DEBUG=1 python3 tools/plugin_probe.py 2>&1 | grep -E "plugin worker|plugin request|ms_per" | sort | uniq -c | sort -rn | head -40
