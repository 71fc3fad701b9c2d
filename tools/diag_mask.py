import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from prisma_amd import engine, synth
from oracle import solov2_oracle as SO
cfg = synth.MASK_CFGS["r101"]; w = synth.solov2_weights(cfg)
net = engine.MaskMMDet(w, cfg, max_batch=2); net.set_profiling(True, True)
frames = synth.frames(2, 720, 1280, seed=2)
KEEP=[synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]
out = net.infer_batch(frames, 0.5, KEEP)
for l in range(5):
    c = net.stage(f"cls_logit{l}"); k = net.stage(f"kernel_pred{l}")
    print(l, 'cls mean %.2f std %.2f max %.2f | band mean %.2f max %.2f | kp std %.2f' % (c.mean(), c.std(), c.max(), c[:,KEEP].mean(), c[:,KEEP].max(), k.std()))
mf = net.stage("mask_feats"); print('mf', mf.mean(), mf.std())
for n in ("c2","c3","c4","c5","p2","p5"):
    t = net.stage(n); print(n, t.mean(), t.std(), np.abs(t).max())
sc, lb, mk, cand = net.instances(1, with_masks=True)
print(cand, len(sc), sc[:8], lb[:8], mk.reshape(len(sc),-1).mean(1)[:8])
kps = [torch.from_numpy(net.stage(f"kernel_pred{l}")).half().float() for l in range(5)]
cps = [torch.from_numpy(net.stage(f"cls_logit{l}")) for l in range(5)]
x, meta = SO.preprocess(frames[1], cfg)
o_sc, o_lb, o_mk, dbg = SO.get_results(cfg, kps, cps, torch.from_numpy(mf), meta["img_shape"], meta["ori_shape"], img_id=1, return_debug=True)
d = np.abs(sc - o_sc.numpy())/o_sc.numpy(); print('max rel score diff', d.max(), d.argmax(), sc[d.argmax()], o_sc[d.argmax()])
print('pre-nms top', dbg['pre_nms_scores'].sort(descending=True)[0][:8])
st = {s["name"]: s for s in net.kernel_stats()}
print({k: (round(v["ms"],2), round(v["flops"]/1e9,1), v["launches"]) for k,v in st.items()})
