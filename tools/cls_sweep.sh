for n in 24 48 96 192 384; do echo "images $n: $(python bench.py --quick --no-parity --images $n --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})")"; done
HM_CLASS_TILES=1536 python tools/bench_classes.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print(' '.join(f\"{k}:{v['k_recon_ms']}\" for k,v in d.items()))"
