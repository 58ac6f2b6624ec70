#!/bin/bash
# r06: the pixel-store patterns of the fused tail as a micro-benchmark (tools/ubench/store_patterns.hip)
cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip 2>/dev/null && ./store_patterns 96 && ./store_patterns 384
