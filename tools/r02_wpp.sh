timeout 1500 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -4
python - <<'PY'
import importlib, sys, json
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module('heif-decoder-lib_amd')
print(json.dumps(bench.wpp_parse_rates(pkg), indent=1))
PY
