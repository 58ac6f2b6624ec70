# picture classes outside the headline: kernel times per class and batch size (tools/bench_classes.py)
for n in 1536 6144; do echo "tiles $n: $(HM_CLASS_TILES=$n python tools/bench_classes.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print(' '.join(f\"{k}:{v['k_recon_ms']}/{v['GP_per_s_kernels']}\" for k,v in d.items()))")"; done
python tools/bench_classes.py > gpurun_out/r02_class_sweep.json 2>/dev/null
