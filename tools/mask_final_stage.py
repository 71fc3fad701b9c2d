"""VERDICT r4 missing #1 / next-round 5a: is the id image's distance from the fp32 oracle's the LAST stage's doing?

For every frame three id images are compared (R channel = (255 x drawn instances) mod 256):
  A  the engine's (HIP path end to end)
  B  the fp32 oracle's (oracle/solov2_oracle.py end to end, torch CPU)
  C  the oracle's FINAL STAGE - solov2_head.py:721-760: dynamic convolution, sigmoid, the two bilinear resizes, `> mask_thr`, Matrix NMS and
     the band accumulation, torch fp32 on the CPU in the oracle's own operation order - run on the ENGINE's soft outputs (kernel predictions,
     class logits, mask features read back through pb_mask_get_stage).
|A != C| is what a final stage "in fp32, in the oracle's order" could still change on the engine's side; |C != B| is what no final stage can
repair, because the inputs already differ (stage maxima 4-7e-6 of the range).  python tools/mask_final_stage.py [tiny|r101] ...
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import solov2_oracle as SO
from prisma_amd import engine, synth

KEEP = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]
CASES = {"tiny": [("tiny", 180, 300, 5), ("tiny", 180, 300, 8), ("tiny", 180, 300, 21)],
         "r101": [("r101", 720, 1280, 2), ("r101", 1080, 1920, 9)]}
tot = [0, 0, 0]
for which in (sys.argv[1:] or ["tiny", "r101"]):
    for arch, H, W, seed in CASES[which]:
        cfg = synth.MASK_CFGS[arch]
        w = synth.solov2_weights(cfg)
        net = engine.MaskMMDet(w, cfg, max_batch=2)
        net.set_profiling(True, True)
        frames = synth.frames(2, H, W, seed=seed)
        A = net.infer_batch(frames, 0.5, KEEP)
        kps = [torch.from_numpy(net.stage(f"kernel_pred{l}")) for l in range(5)]
        cps = [torch.from_numpy(net.stage(f"cls_logit{l}")) for l in range(5)]
        mf = torch.from_numpy(net.stage("mask_feats"))
        for b in range(2):
            x, meta = SO.preprocess(frames[b], cfg)
            okps, ocps, omf = SO.network(w, cfg, x)
            sc, lb, mk, dbg = SO.get_results(cfg, okps, ocps, omf, meta["img_shape"], meta["ori_shape"], return_debug=True)
            B = SO.band_mask(sc, lb, mk, synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5, meta["ori_shape"])
            sc2, lb2, mk2, dbg2 = SO.get_results(cfg, kps, cps, mf, meta["img_shape"], meta["ori_shape"], img_id=b, return_debug=True)
            C = SO.band_mask(sc2, lb2, mk2, synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5, meta["ori_shape"])
            ab, cb, ac = [int((p[..., 0] != q[..., 0]).sum()) for p, q in ((A[b], B), (C, B), (A[b], C))]
            tot = [tot[0] + ab, tot[1] + cb, tot[2] + ac]
            drawn = int(((sc.numpy() > 0.5) & np.array([synth.COCO_CLASSES[int(c)] in synth.BAND_CLASSES for c in lb.numpy()], bool)).sum())
            soft = dbg["soft"].numpy()
            near = int((np.abs(soft - cfg.mask_thr) < 1e-5).sum())
            print(f"{arch} {W}x{H} seed {seed} frame {b}: {H * W} pixels, {drawn} instances drawn | engine vs oracle {ab} | oracle final stage on "
                  f"engine inputs vs oracle {cb} | engine vs oracle final stage on engine inputs {ac} | oracle soft-mask values within 1e-5 of "
                  f"mask_thr (all {soft.shape[0]} instances): {near}", flush=True)
        net.close()
print(f"totals: engine vs oracle {tot[0]}, oracle final stage on engine inputs vs oracle {tot[1]}, engine vs that {tot[2]}")
