"""INTEGRATION.md's reference-side binding is a real file (integration/depth_anything_stub.py): the document quotes it
verbatim, its ctypes structs match the C header (checked with gcc), and on a GPU the stub runs against the built library."""
import ctypes as C
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "integration", "depth_anything_stub.py")


def header_layout(tmp_path, structs):
    """{struct: (sizeof, {field: offset})} as gcc sees include/prisma_bands.h."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "prisma_bands.h"', 'int main(void) {']
    for s, fields in structs.items():
        lines.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return dict(zip(out[::2], map(int, out[1::2])))


def check_struct(cls, name, lay):
    assert C.sizeof(cls) == lay[name], (name, C.sizeof(cls), lay[name])
    for f, _ in cls._fields_:
        assert getattr(cls, f).offset == lay[f"{name}.{f}"], (name, f)


def load_stub(lib_path):
    os.environ["PRISMA_BANDS_LIB"] = lib_path
    spec = importlib.util.spec_from_file_location("depth_anything_stub", STUB)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_document_quotes_the_stub_verbatim():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "```python\n" + open(STUB).read() + "```" in md


def test_struct_layouts_match_the_header(tmp_path):
    import __graft_entry__ as entry
    from prisma_amd import _lib
    entry.build()
    fields = lambda cls: [f for f, _ in cls._fields_]
    stub = load_stub(_lib.LIB_PATH)
    lay = header_layout(tmp_path, {"pb_tensor": fields(_lib.pb_tensor), "pb_depth_cfg": fields(_lib.pb_depth_cfg),
                                   "pb_flow_cfg": fields(_lib.pb_flow_cfg), "pb_mask_cfg": fields(_lib.pb_mask_cfg),
                                   "pb_kernel_stat": fields(_lib.pb_kernel_stat)})
    for cls, name in ((_lib.pb_tensor, "pb_tensor"), (_lib.pb_depth_cfg, "pb_depth_cfg"), (_lib.pb_flow_cfg, "pb_flow_cfg"),
                      (_lib.pb_mask_cfg, "pb_mask_cfg"), (_lib.pb_kernel_stat, "pb_kernel_stat"),
                      (stub.pb_tensor, "pb_tensor"), (stub.pb_depth_cfg, "pb_depth_cfg")):
        check_struct(cls, name, lay)
    # the header's field lists are complete: no member the bindings do not know
    hdr = open(os.path.join(ROOT, "include", "prisma_bands.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} pb_depth_cfg;", hdr).group(1)
    assert len(re.findall(r"int32_t\s+\w+", body)) == len(_lib.pb_depth_cfg._fields_) == len(stub.pb_depth_cfg._fields_)


@pytest.mark.gpu
def test_stub_runs_against_the_library():
    from oracle import depth_oracle as O
    from prisma_amd import _lib, synth
    stub = load_stub(_lib.LIB_PATH)
    cfg = synth.DEPTH_CFGS["vits"]
    w = synth.depth_anything_weights(cfg, seed=1234)
    stub.init_model(w, "vits")
    frame = synth.frames(1, 96, 128, seed=11)[0]
    d = stub.infer(frame)
    ref = O.infer(w, frame, cfg.depth, cfg.heads)
    assert float(np.abs(d - ref).max() / np.abs(ref).max()) < 1e-3
    n = stub.infer(frame, normalize=True)
    assert n.min() == 0.0 and abs(n.max() - 1.0) < 1e-6
