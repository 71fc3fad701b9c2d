"""CPU: the C-ABI library builds, loads, exports every declared symbol, and refuses to run
without a GPU (no silent fallback).  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from oracle import depth_oracle as O
from prisma_amd import _lib, synth


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(os.path.dirname(_lib._HERE), "include", "prisma_bands.h")).read()
    declared = set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)


def test_net_size_matches_oracle(lib):
    for (w, h) in [(1280, 720), (1920, 1080), (934, 440), (518, 518), (128, 96), (120, 90), (640, 481), (333, 777)]:
        nh, nw = C.c_int(), C.c_int()
        assert lib.pb_depth_net_size(h, w, C.byref(nh), C.byref(nw)) == 0
        assert (nw.value, nh.value) == O.net_size(w, h)


def test_no_gpu_fails_loudly(lib):
    if lib.pb_device_count() > 0:
        pytest.skip("a GPU is visible")
    ctx = C.c_void_p()
    rc = lib.pb_create(C.byref(ctx), 0, b"ops", None, 0, None, 0)
    assert rc == -2 and not ctx.value
    assert b"no HIP device" in lib.pb_last_error()


def test_synth_weights_are_deterministic():
    a = synth.depth_anything_weights("vits", seed=1234)
    b = synth.depth_anything_weights("vits", seed=1234)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert a["pretrained.blocks.0.attn.qkv.weight"].shape == (1152, 384)
    assert a["depth_head.resize_layers.0.weight"].shape == (48, 48, 4, 4)
    c = synth.depth_anything_weights("vits", seed=1)
    assert not np.array_equal(a["pretrained.cls_token"], c["pretrained.cls_token"])
