#!/usr/bin/env python3
"""k_tailf on the class of 10-bit HDR photographs (bench.py: hdr10_420_grid_rgb24, 96 x 12 MP grids of 10-bit 4:2:0 tiles -> RGB24) with its
stages switched off in turn: 3 = deblocking + SAO, 2 = SAO only, 1 = deblocking only, 0 = neither (loads, LDS, colour, stores).
r05: 5.26 / 3.07 / 4.56 / 2.63 ms - deblocking (the general one-line-at-a-time filters on 16-bit samples, edge parameters from memory) is
42 % of the kernel.  usage (repo root, GPU box): python3 tools/tailf_probe.py"""
import sys, json, ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import __graft_entry__ as g
pkg = g.load_package(test_knobs=True)  # (HM_TAIL_HDR16=0: k_tailf instead of k_tail420's 16-bit instantiation)
capi, L = pkg.capi, pkg.lib()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
W,H,cols,rows,tile = bench.OUT_W, bench.OUT_H, bench.GRID_COLS, bench.GRID_ROWS, bench.TILE
n=96
datas=[bench.tile_stream(9100000+k, bit_depth=10) for k in range(cols*rows)]
blobs=[capi.parse_hevc(d) for d in datas]
ys,cs,os_=L.hm_plane_stride(W,2),L.hm_plane_stride((W+1)//2,2),L.hm_plane_stride(W,3)
ch=(H+1)//2
batch=capi.Batch(); ims=[]
for _ in range(n):
    im=(torch.zeros((H,ys),dtype=torch.uint8,device=dev),torch.zeros((max(64,ch),cs),dtype=torch.uint8,device=dev),torch.zeros((max(64,ch),cs),dtype=torch.uint8,device=dev),torch.zeros((H,os_),dtype=torch.uint8,device=dev))
    for t in range(cols*rows):
        d=capi.TileDest(); d.plane[0],d.plane[1],d.plane[2]=im[0].data_ptr(),im[1].data_ptr(),im[2].data_ptr(); d.pitch[0],d.pitch[1],d.pitch[2]=ys,cs,cs
        d.canvas_width,d.canvas_height=W,H; d.x0,d.y0=(t%cols)*tile,(t//cols)*tile; d.tile_has_nclx,d.tile_full_range,d.tile_matrix=1,1,6
        batch.add(blobs[t],d)
    ims.append(im)
batch.upload(st)
desc=capi.ColourDesc(W,H,10,1,0,6,1,1,capi.HM_OUT_RGB,ys,cs,cs,os_)
PtrArr=C.c_void_p*n
ptrs=[PtrArr(*[im[k].data_ptr() for im in ims]) for k in range(4)]
batch.set_colour(desc,n,*ptrs,0)
for stages in ([int(a) for a in sys.argv[1:]] or [3,2,1,0]):  # (tools/r06_tailf_counters.sh: "3" only)
    batch.execute(stages,st); torch.cuda.synchronize()
    batch.set_profiling(3)
    for _ in range(3): batch.execute(stages,st)
    torch.cuda.synchronize()
    ms=[batch.timings5_ms(i) for i in range(3)]
    print("stages",stages,"k_tailf ms", round(sum(m[2] for m in ms)/3,3), "fused", batch.tail_fused())
batch.check(); batch.close()
