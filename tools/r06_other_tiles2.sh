#!/bin/bash
HM_CHECK_TILE=256 HM_CHAIN_DEBUG=1 timeout 600 python3 tools/check_launcher.py 1536 768 3072 > /tmp/o256.txt 2> /tmp/e256.txt; grep "tiles:" /tmp/o256.txt; grep "k_chain\]" /tmp/e256.txt | head -4
python3 - <<'PY'
import json
for line in open('/tmp/o256.txt'):
    if line.startswith('{'):
        d=json.loads(line)
        for r in d['rows']:
            a=r['all']; print(d['tile'], r['tiles'], 'auto', a['auto'], sorted(a.items(), key=lambda kv: kv[1])[:6])
PY
HM_CHECK_TILE=1024 HM_CHAIN_DEBUG=1 timeout 900 python3 tools/check_launcher.py 1536 768 > /tmp/o1024.txt 2> /tmp/e1024.txt; grep "tiles:" /tmp/o1024.txt; grep "k_chain\]" /tmp/e1024.txt | head -4
python3 - <<'PY'
import json
for line in open('/tmp/o1024.txt'):
    if line.startswith('{'):
        d=json.loads(line)
        for r in d['rows']:
            a=r['all']; print(d['tile'], r['tiles'], 'auto', a['auto'], sorted(a.items(), key=lambda kv: kv[1])[:6])
PY
