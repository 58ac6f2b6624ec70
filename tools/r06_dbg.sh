#!/bin/bash
HM_CLASS_ONLY=8bit_420_ctb16 HM_CHAIN_DEBUG=1 timeout 600 python3 tools/check_launcher.py 1536 2>&1 | grep -E "k_chain\]|tiles:" | sort | uniq -c | sort -rn | head -12
HM_CHECK_TILE=1024 HM_CHAIN_DEBUG=1 timeout 600 python3 tools/check_launcher.py 1536 2>&1 | grep -E "k_chain\]|tiles:" | sort | uniq -c | sort -rn | head -12
