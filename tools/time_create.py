"""Host-side set-up times of the engines on the GPU box (synthetic weights, weight packing + upload): python tools/time_create.py
round 3: depth ViT-L split 4.3 s (f16 2.0 s), flow_raft 0.13 s, mask R-101 1.1 s, flow_gmflow 0.1 s; synth ViT-L weights 2.4 s."""
import time, sys, os
sys.path.insert(0, os.getcwd())
t=time.time(); import torch; from prisma_amd import engine, synth; print("import", round(time.time()-t,2))
t=time.time(); dw=synth.depth_anything_weights("vitl", seed=1234); print("synth depth", round(time.time()-t,2))
t=time.time(); rw=synth.raft_weights(seed=4321); print("synth raft", round(time.time()-t,2))
torch.zeros(1).cuda(); 
for prec in (1,0):
    t=time.time(); dn=engine.DepthAnything(dw,"vitl",max_batch=32,precision=prec); print("create depth prec",prec, round(time.time()-t,2))
    t=time.time(); fn=engine.FlowRaft(rw,precision=prec); print("create flow prec",prec, round(time.time()-t,2))
    dn.close(); fn.close()
t=time.time(); mw=synth.solov2_weights(synth.MASK_CFGS["r101"]); print("synth mask", round(time.time()-t,2))
t=time.time(); m=engine.MaskMMDet(mw, synth.MASK_CFGS["r101"], max_batch=8); print("create mask", round(time.time()-t,2))
t=time.time(); gw=synth.gmflow_weights(seed=2468); g=engine.FlowGMFlow(gw); print("synth+create gmflow", round(time.time()-t,2))
