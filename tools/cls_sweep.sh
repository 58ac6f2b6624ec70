#!/bin/bash
# Picture classes outside the headline: reconstruction kernel time per class and batch size (tools/bench_classes.py),
# and few large pictures by shape (tools/shape_probe.py).  A/B knobs, read once per process:
#   HM_QUAD_CLASS=1|0   every / no picture without rare syntax goes to k_recon_quad (default: hevc_syntax.h quad_class)
#   HM_QUAD_WAVES=n     waves per picture of k_recon_quad;  HM_RECON_WAVES=n  of k_recon;  HM_QUAD_DEBUG=1 prints the choice
for n in 1536 6144 18432; do echo "tiles $n: $(HM_CLASS_TILES=$n python tools/bench_classes.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print(' '.join(f\"{k}:{v['k_recon_ms']}\" for k,v in d.items()))")"; done
python tools/shape_probe.py 2>/dev/null | tail -1
