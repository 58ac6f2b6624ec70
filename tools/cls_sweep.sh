for n in 1536 6144 18432; do echo "tiles $n: $(HM_CLASS_TILES=$n python tools/bench_classes.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print(' '.join(f\"{k}:{v['k_recon_ms']}\" for k,v in d.items()))")"; done
