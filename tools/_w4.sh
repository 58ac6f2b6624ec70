mkdir -p gpurun_out
{
echo "== chain modes + decode"; timeout 1200 python3 -m pytest tests/test_chain_modes_gpu.py tests/test_decode_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== few pictures"; timeout 600 python3 tools/few_pictures_probe.py 2>&1 | tail -1
echo "== plugin"; HM_PLUGIN_DEBUG=1 timeout 300 python3 tools/plugin_probe.py 2>/tmp/plug.err | tail -1 | cut -c1-400; grep "plugin worker" /tmp/plug.err | tail -12; grep "plugin decode_image" /tmp/plug.err | tail -5
echo "== cfg5"; timeout 600 python3 -c "
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, json, __graft_entry__ as g, bench
pkg=g.load_package(); torch.cuda.set_device(0); dev=torch.device('cuda',0); st=torch.cuda.current_stream().cuda_stream
r=bench.config5_single(torch,pkg,dev,st); print(r['MP_per_s'], {k:v['ms_per_step'] for k,v in r['kernels'].items()})
" 2>&1 | tail -1
} > gpurun_out/r04h_w4.log 2>&1
cat gpurun_out/r04h_w4.log
