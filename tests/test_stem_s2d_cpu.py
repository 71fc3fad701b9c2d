"""CPU: the identity behind the engines' stems (raft_engine.hip / mask_engine.hip load()): a 7x7 / stride-2 / pad-3 convolution equals a
3x3 / stride-1 / pad-1 convolution on the 4 x 4 space-to-depth image whose 4 x O output channels are the 2 x 2 output pixels of a block,
with  W2[(sy, sx, o)][(ty, tx)][(dy, dx, c)] = w[o][c][ky][kx],  ky = 4 (ty - 1) + dy - 2 sy + 3,  kx likewise, zero outside 0..6."""
import numpy as np
import torch
import torch.nn.functional as F


def s2d_weights(w):
    O, C = w.shape[:2]
    w2 = np.zeros((4 * O, 3, 3, 64), np.float32)
    for sy in range(2):
        for sx in range(2):
            for ty in range(3):
                for tx in range(3):
                    for dy in range(4):
                        for dx in range(4):
                            ky, kx = 4 * (ty - 1) + dy - 2 * sy + 3, 4 * (tx - 1) + dx - 2 * sx + 3
                            if 0 <= ky < 7 and 0 <= kx < 7:
                                w2[(sy * 2 + sx) * O:(sy * 2 + sx + 1) * O, ty, tx, (dy * 4 + dx) * 4:(dy * 4 + dx) * 4 + C] = w[:, :, ky, kx]
    return w2


def test_stem_as_space_to_depth_conv():
    rng = np.random.default_rng(0)
    O, C, H, W = 8, 3, 24, 32
    w = rng.standard_normal((O, C, 7, 7)).astype(np.float32)
    x = rng.standard_normal((H, W, C)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(w), stride=2, padding=3)[0].permute(1, 2, 0).numpy()
    x2 = np.zeros((H // 4, W // 4, 64), np.float32)
    for dy in range(4):
        for dx in range(4):
            x2[:, :, (dy * 4 + dx) * 4:(dy * 4 + dx) * 4 + C] = x[dy::4, dx::4, :]
    w2 = s2d_weights(w)
    y2 = F.conv2d(torch.from_numpy(x2).permute(2, 0, 1)[None], torch.from_numpy(w2).permute(0, 3, 1, 2), padding=1)[0].permute(1, 2, 0).numpy()
    out = np.zeros_like(ref)
    for sy in range(2):
        for sx in range(2):                                   # pixel shuffle (EPI_PIXSHUF, ps_s = 2, ps_co = O)
            out[sy::2, sx::2, :] = y2[:, :, (sy * 2 + sx) * O:(sy * 2 + sx + 1) * O]
    assert np.abs(out - ref).max() < 1e-4 * np.abs(ref).max()
    assert abs(float((w2 != 0).mean()) - 49 * 3 / (9 * 64)) < 1e-6     # 147 of the 576 K entries per output channel are real taps
